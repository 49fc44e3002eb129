// Persistent stream-K GEMM for the prefill / adapter regime (M in the hundreds to tens of thousands, fragment-packed W).
//
// Why not plain output tiling: at the path's shapes (M = 7 calls x ~170 tokens ~ 1200) a 128x128 tiling gives 320 tiles
// for the N = 4096 projections on 256 CUs x 2 slots - 64 CUs run two tiles while 192 run one - and each K-step (64)
// of one tile is only ~0.2 us of MFMA work, less than one HBM/L2 round trip, so a double-buffered block spends most
// of its time waiting at the barrier.  This kernel fixes both:
//   * ONE persistent 8-wave workgroup per CU (grid = #CUs).  The (tile, k-step) iteration space is cut into equal
//     contiguous ranges, one per CU (stream-K), so every CU does the same number of MFMA k-steps whatever M, N are.
//     A tile whose k-range is shared by several CUs is finished by the CU that owns its FIRST k-step; the others
//     publish fp32 partials (register-ordered, 128 KiB, fully coalesced) and a flag.  Contributors handle their
//     shared tile at the START of their range and the finisher at the END of its own, so the wait is normally over
//     before it begins.  Hand-off follows the agent-scope release / acquire recipe (vector L1s are not coherent
//     across CUs): stores -> vmcnt(0) -> barrier -> one lane: release fence + drained flag store; consumer: relaxed
//     poll -> one acquire fence -> barrier -> plain loads.  Flags are self-cleaning (the consumer zeroes them).
//     Summation order is fixed (ascending k-range), so results are deterministic.
//   * 128 x 256 x 64 tiles in a 3-stage LDS ring (3 x 48 KiB) filled by LDS-DMA (global_load_lds, 16 B/lane) with a
//     COUNTED vmcnt: two stages stay in flight across the single raw s_barrier per k-step.
// A is row-major bf16 (XOR-swizzled through the per-lane source address, LDS destinations are lane-linear), W is
// fragment-packed so its LDS image is already in ds_read_b128 order.  Epilogues as in gemm.hip.
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// Two geometries (NW = waves per workgroup; every wave owns a 64 x 64 sub-tile = 4 x 4 MFMA fragments):
//   NW = 8: 128 x 256 tile, 3-stage ring (144 KiB), ONE workgroup per CU  - two stages in flight across the barrier
//   NW = 4: 128 x 128 tile, 2-stage ring (64 KiB),  TWO workgroups per CU - the two workgroups drift out of phase, so one's
//           MFMA burst overlaps the other's LDS-DMA / ds_read phase (a single workgroup is phase-locked by its barrier)
constexpr int BM = 128, BK = 64;
constexpr int A_BYTES = BM * BK * 2;
template <int NW> struct Geo {
    static constexpr int BN = NW * 32, W_BYTES = BN * BK * 2, STAGE_BYTES = A_BYTES + W_BYTES;
    static constexpr int STAGES = NW == 8 ? 3 : 2, A_LOADS = 16 / NW, PARTIAL_F4 = NW * 16 * 64, WCOLS = NW / 2;
};

template <int NW>
__device__ __forceinline__ void stage_load(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wp, int M, int K,
                                           int m0, int n0, int k0, char* slot, int wave, int lane) {
    constexpr int AL = Geo<NW>::A_LOADS;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int r8 = (wave * AL + i) * 8;
        const int row = r8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int grow = m0 + row;
        grow = grow < M ? grow : M - 1;
        glds16(A + (int64_t)grow * lda + k0 + c * 8, slot + r8 * (BK * 2));
    }
    const int kfr = K >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = wave * 4 + i;
        const int nt = (n0 >> 4) + (f >> 1);
        glds16(Wp + (((int64_t)nt * kfr + (k0 >> 5) + (f & 1)) * 64 + lane) * 8, slot + A_BYTES + f * 1024);
    }
}

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

template <int OUT_BF16, int ACT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_sk(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wp,
                                               const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                               int64_t ldc, int M, int N, int K, int tiles_m, int ksteps, int64_t total_units,
                                               f32x4* partial, int* flags, int* status) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BN = Geo<NW>::BN, STAGES = Geo<NW>::STAGES, STAGE_BYTES = Geo<NW>::STAGE_BYTES, PARTIAL_F4 = Geo<NW>::PARTIAL_F4;
    const int wr = wave / Geo<NW>::WCOLS, wc = wave % Geo<NW>::WCOLS;
    const int fr = lane & 15, kg = lane >> 4;
    // Blocks land on XCD (blockIdx % 8), each XCD with a private L2.  Rank the blocks XCD-major so that the CUs of one
    // XCD own one contiguous eighth of the unit space (= whole W column panels, walked m-tile by m-tile at similar k):
    // the panel is then fetched from HBM once per XCD instead of once per CU.  Flags / partials are indexed by rank.
    const int G = gridDim.x;
    const int X = (G & 7) == 0 ? 8 : 1;
    const int b = (blockIdx.x % X) * (G / X) + blockIdx.x / X;
    const int64_t u_begin = (int64_t)b * total_units / G, u_end = (int64_t)(b + 1) * total_units / G;

    for (int64_t u = u_begin; u < u_end;) {
        const int tile = (int)(u / ksteps);
        const int ks0 = (int)(u % ksteps);
        const int nks = (int)min((int64_t)(ksteps - ks0), u_end - u);
        const int m0 = (tile % tiles_m) * BM, n0 = (tile / tiles_m) * BN;

        f32x4 acc[4][4];  // [ni][mi]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        stage_load<NW>(A, lda, Wp, M, K, m0, n0, ks0 * BK, smem, wave, lane);
        if (STAGES == 3 && nks > 1) stage_load<NW>(A, lda, Wp, M, K, m0, n0, (ks0 + 1) * BK, smem + STAGE_BYTES, wave, lane);
        int slot = 0;
        for (int i = 0; i < nks; ++i) {
            // stage i must have landed; with 3 stages, stage i+1 (6 loads per lane) may stay in flight across the barrier
            if (STAGES == 3 && i + 1 < nks) wait_vmcnt<6>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (i + STAGES - 1 < nks) {
                int s2 = slot + STAGES - 1;   // the slot read in iteration i-1: every wave is past it after this barrier
                s2 = s2 >= STAGES ? s2 - STAGES : s2;
                stage_load<NW>(A, lda, Wp, M, K, m0, n0, (ks0 + i + STAGES - 1) * BK, smem + s2 * STAGE_BYTES, wave, lane);
            }
            const char* a_s = smem + slot * STAGE_BYTES;
            const char* w_s = a_s + A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 wf[4], af[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    wf[q] = *(const bf16x8*)(w_s + ((wc * 4 + q) * 2 + ks) * 1024 + lane * 16);
                    const int row = wr * 64 + q * 16 + fr;
                    af[q] = *(const bf16x8*)(a_s + row * (BK * 2) + (((ks * 4 + kg) ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
                        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], af[mi], acc[ni][mi], 0, 0, 0);
            }
            slot = slot + 1 == STAGES ? 0 : slot + 1;
        }
        __builtin_amdgcn_s_barrier();  // every wave is done with the ring before the next segment's prologue refills it

        const bool head = ks0 == 0, whole = head && nks == ksteps;
        if (!head) {
            // contributor: publish the partial accumulators of this segment, then go on with the next tile
            f32x4* P = partial + (int64_t)b * PARTIAL_F4 + wave * 16 * 64 + lane;
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) P[(ni * 4 + mi) * 64] = acc[ni][mi];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(flags + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            u += nks;
            continue;
        }
        if (!whole) {
            // finisher: add the partials of the blocks that own the rest of this tile's k-range, in ascending order
            const int64_t tile_end = (int64_t)(tile + 1) * ksteps;
            for (int bp = b + 1; bp < G && (int64_t)bp * total_units / G < tile_end; ++bp) {
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(flags + bp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1u << 26)) {
                            *status = 1;
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const f32x4* P = partial + (int64_t)bp * PARTIAL_F4 + wave * 16 * 64 + lane;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[ni][mi] += P[(ni * 4 + mi) * 64];
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + bp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // epilogue: lane owns row m = ..+fr, columns n = ..+kg*4 .. +3 of each 16x16 fragment
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr * 64 + mi * 16 + fr;
            if (m >= M) continue;
            if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
                for (int ni = 0; ni < 4; ni += 2) {
                    const int n = n0 + wc * 64 + ni * 16;
                    const int no = (n >> 1) + kg * 4;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                    if (OUT_BF16) *(u32x2*)((bf16_t*)Cv + (int64_t)m * ldc + no) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
                }
            } else {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const int n = n0 + wc * 64 + ni * 16 + kg * 4;
                    f32x4 v = acc[ni][mi];
                    if (bias) v += *(const f32x4*)(bias + n);
                    if (ACT == RV_ACT_RELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (res) v += *(const f32x4*)(res + (int64_t)m * ldr + n);
                    if (OUT_BF16) *(u32x2*)((bf16_t*)Cv + (int64_t)m * ldc + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
                }
            }
        }
        u += nks;
    }
}

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

template <int OUT_BF16, int ACT, int NW>
int launch(const bf16_t* A, int64_t lda, const bf16_t* Wp, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
           int M, int N, int K, void* ws, hipStream_t st) {
    static bool attr_set = false;
    constexpr int BN = Geo<NW>::BN;
    const int smem = Geo<NW>::STAGES * Geo<NW>::STAGE_BYTES;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_sk<OUT_BF16, ACT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) {
            rv_set_error("gemm_sk: cannot reserve %d bytes of LDS", smem);
            return RV_ERR_HIP;
        }
        attr_set = true;
    }
    const int tiles_m = (int)cdiv(M, BM), tiles_n = N / BN, ksteps = K / BK;
    const int64_t total = (int64_t)tiles_m * tiles_n * ksteps;
    int G = num_cus() * (NW == 8 ? 1 : 2);
    if (total < G) G = (int)total;
    int* flags = (int*)ws;
    int* status = flags + 1023;
    f32x4* partial = (f32x4*)((char*)ws + 4096);
    hipLaunchKernelGGL((gemm_sk<OUT_BF16, ACT, NW>), dim3(G), dim3(NW * 64), smem, st, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, tiles_m,
                       ksteps, total, partial, flags, status);
    return RV_OK;
}

}  // namespace

size_t gemm_sk_ws_bytes() { return 4096 + (size_t)num_cus() * Geo<8>::PARTIAL_F4 * sizeof(f32x4); }  // = 2*CUs*Geo<4>

// Measured on MI355X at the path's shapes (tools/kbench.py, M ~ 1.1-1.2k): stream-K removes the tile-quantisation loss
// on the N = 4096 projections (down: 150 -> 129 us) but loses the L2 coincidence the output-tiled kernel gets from
// launching all m-tiles of a W panel on one XCD at the same k, and pays ~10 us of partial-tile hand-off per launch, so
// it is slower on the wide projections (qkv 125 -> 185 us).  It is therefore OFF by default (geometry 0) and kept as an
// opt-in knob for shapes with few tiles per CU.
int g_sk_geometry = 0;  // 0 = output-tiled kernel only; 4 / 8 = stream-K with 4- / 8-wave workgroups (see Geo<>)
extern "C" void rv_set_gemm_geometry(int32_t waves) { g_sk_geometry = (waves == 8 || waves == 4) ? waves : 0; }

bool gemm_sk_supported(int w_layout, int64_t M, int64_t N, int64_t K) {
    return g_sk_geometry != 0 && w_layout == 1 && M > 16 && N % (g_sk_geometry * 32) == 0 && K % BK == 0 && num_cus() <= 511;
}

int gemm_sk_launch(const void* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr, void* C,
                   int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws, hipStream_t st) {
    const bf16_t* a = (const bf16_t*)A;
    const bf16_t* w = (const bf16_t*)Wp;
    const int ob = out_dtype == RV_BF16;
    int rc;
#define SK(OB, AC)                                                                                        \
    rc = g_sk_geometry == 8 ? launch<OB, AC, 8>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, ws, st) \
                            : launch<OB, AC, 4>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, ws, st)
    if (ob && act == RV_ACT_NONE) SK(1, RV_ACT_NONE);
    else if (ob && act == RV_ACT_RELU) SK(1, RV_ACT_RELU);
    else if (ob && act == RV_ACT_SILU_MUL) SK(1, RV_ACT_SILU_MUL);
    else if (!ob && act == RV_ACT_NONE) SK(0, RV_ACT_NONE);
    else if (!ob && act == RV_ACT_RELU) SK(0, RV_ACT_RELU);
    else SK(0, RV_ACT_SILU_MUL);
#undef SK
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_sk");
    return RV_OK;
}
