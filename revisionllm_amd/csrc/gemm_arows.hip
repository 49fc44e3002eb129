// A-resident GEMM for the short-K, many-row family (the "feature scan": nn.Linear(768 -> 4096) over [segments x frames]
// rows, vtimellm_arch.py:42,125; and the K = 768 projections of the ClipEncoder, transformer.py:188-337):
//   C[M,N] = act(A[M,K] . W[N,K]^T + bias) + residual,   K <= 1024, M >> N-tile, W fragment-packed.
//
// Why another kernel.  A 128x128 output tile moves (128 + 128) * 2 B of operands per k through the CU's 64 B/clk vector-memory
// path for 128 * 128 MACs, which at 2048 MAC/clk is exactly that path's rate: the tile kernels are load-path-bound, and with
// M = 25.6k rows every A tile is re-fetched by each of the N / 128 column tiles (PMC: 1.5 GB of fabric traffic for 255 MB of
// algorithmic bytes on the dense projector).  Here a workgroup OWNS a block of rows for the whole launch:
//   * its A rows (all K) are copied ONCE into LDS (<= 104 rows x 768 x 2 B = 156 KiB of the CU's 160 KiB) by LDS-DMA,
//     XOR-swizzled through the per-lane source address so that the MFMA operand reads (ds_read_b128) are conflict-free;
//   * the 8 waves then walk the N dimension independently - wave w takes the 64- (or 32-) column slices w, w + 8, ... - with
//     the W fragments going L2 -> VGPR directly (fragment-packed W: one coalesced 1 KiB load per 16 x 32 fragment; a fragment is
//     used by exactly one wave, so an LDS round trip would be overhead) through a 4-deep rotating register pipeline that runs
//     across slice boundaries;
//   * there is NO barrier after the A copy: the waves drift apart, one wave's epilogue (bias / activation / residual / stores)
//     and load latency hide under the other waves' MFMAs.
// Per k-step of 32 a wave issues MF LDS reads (its whole row block) + NW W-fragment loads for MF * NW MFMAs: with MF = 7, NW = 4
// that is 25 % of the LDS read rate and 57 % of the vector-memory path at full MFMA rate.  HBM traffic = A once + C once (+ W
// once per XCD): the algorithmic bytes.
// Rows are dealt evenly: rows_per_wg = ceil(M / (#CU * rounds)); the last 16-row fragment of a block is partly padding
// (M = 25600 on 256 CUs: 100 rows in 7 fragments, 89 % useful MFMA work).
#include <atomic>
#include "kernels.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int AR_LDS_MAX = 160 * 1024;
constexpr int AR_PAD = 2;   // pad chunks (16 B) per LDS row: row stride = K * 2 + 32 B puts the 16 lanes of every ds_read_b128 lane
                            // group ({rows 0-3, 12-15} at k-chunk kg with {rows 4-11} at kg + 1, ...) on 16 distinct 16-byte slots

// NW = n-fragments (16 columns) per wave slice.  W fragments sit in a ring of 4 register slots, loaded 2 k-steps ahead with
// the step inside its half-body (4 k-steps = 4 KiB) as the load's immediate offset.
template <int OUT_BF16, int ACT, int MF, int NW>
__global__ __launch_bounds__(512) void gemm_arows_kernel(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp,
                                                         const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                                         int64_t ldc, int M, int N, int K, int rows_per_wg, int probe) {
    // probe (measurement only, wrong results): bit 0 = every slice re-reads the W fragments of the wave's first slice (a small
    // L2-hot region instead of streaming all of W), bit 1 = no stores
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * rows_per_wg;
    int nrows = M - r0;
    nrows = nrows < rows_per_wg ? nrows : rows_per_wg;
    if (nrows <= 0) return;
    const int cpr = K >> 3;            // 16-byte chunks per row
    const int lcpr = cpr + AR_PAD;     // ... in the LDS image
    // ---- A rows -> LDS, once (LDS-DMA: a wave instruction fills 64 consecutive LDS chunks; pad chunks repeat the row's last) ----
    {
        const int total = nrows * lcpr;
        for (int q0 = wave * 64; q0 < total; q0 += 512) {
            int q = q0 + lane;
            q = q < total ? q : total - 1;
            const int r = q / lcpr;
            int c = q - r * lcpr;
            c = c < cpr ? c : cpr - 1;
            __builtin_amdgcn_global_load_lds((gptr_t)(A + (int64_t)(r0 + r) * lda + c * 8), (lptr_t)(smem + q0 * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int fr = lane & 15, kg = lane >> 4;
    const int KF = K >> 5;                              // k-steps (fragments) per slice: a multiple of 8
    const int nslices = N / (NW * 16);
    const int my_slices = (nslices - wave + 7) >> 3;    // slices wave, wave + 8, ...
    if (my_slices <= 0) return;
    // LDS address of this lane's A fragment (mf, kf = 0): row mf*16 + fr (clamped into the block: rows past it repeat the last
    // row, computed but never stored), chunk kg; k-step kf adds kf * 64 bytes - an immediate offset inside a body of 8 steps
    int abase[MF];      // (byte offsets into smem: keeps the reads in the LDS address space - ds_read_b128 with an immediate)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int row = mf * 16 + fr;
        row = row < nrows ? row : nrows - 1;
        abase[mf] = (row * lcpr + kg) * 16;
    }
    // W: fragment (nt, kf) is the 1 KiB block (nt * KF + kf); this lane's 16 bytes of it.  Consumer position = (slice si, kf);
    // the producer runs 4 k-steps ahead and crosses into the wave's next slice (nt + 8 * NW) at the end of a slice.
    const int64_t slice_jump = (probe & 1) ? -(int64_t)KF * 512 : ((int64_t)8 * NW * KF - KF) * 512;     // elements: from the end of a slice to the start of the next one
    const op16_t *wcur[NW], *wnext[NW];       // this lane's 16 bytes of k-step 0 of the consumer's half-body / of the one after it
#pragma unroll
    for (int nf = 0; nf < NW; ++nf) wnext[nf] = Wp + ((int64_t)(wave * NW + nf) * KF * 64 + lane) * 8;
    int next_kf = 0, next_si = 0;             // position of wnext
    auto half_advance = [&]() {               // wcur <- wnext; wnext <- the half-body after it (past the end: stays, never used)
        next_kf += 4;
        int64_t step = 4 * 512;
        if (next_kf == KF) {
            next_kf = 0;
            if (next_si + 1 < my_slices) { ++next_si; step += slice_jump; }
            else { next_kf = KF - 4; step = 0; }
        }
#pragma unroll
        for (int nf = 0; nf < NW; ++nf) { wcur[nf] = wnext[nf]; wnext[nf] += step; }
    };
    half_advance();

    f32x4 acc[MF][NW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NW; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // W ring of 4 slots: k-step j of a body uses slot j & 3, its fragments were loaded 2 steps earlier
    op16x8 wb[4][NW], afA[MF], afB[MF];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int nf = 0; nf < NW; ++nf) wb[j][nf] = *(const op16x8*)(wcur[nf] + j * 512);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) afA[mf] = *(const op16x8*)(smem + abase[mf]);

    // Deferred bf16 stores.  A finished slice is packed to bf16 at once (two 16-column fragments -> one 16-byte piece per lane:
    // v_permlane16_swap trades the odd 16-lane rows of the first fragment's words with the even rows of the second's, after
    // which lane (fr, kg) holds 8 CONSECUTIVE columns, n0 + (kg & 1) * 16 + (kg >> 1) * 8 .. + 7) and written one row fragment
    // per k-step during the NEXT slice: vmcnt is in-order, so a burst of 14 stores in front of the W loads in flight would park
    // the wave until every store is acknowledged; one store between two loads is invisible.
    u32x4 pend[MF];
    op16_t* pend_ptr = nullptr;      // &C[r0 + fr][column of this lane's piece] of the pending slice; nullptr = nothing pending
    const int64_t mf_stride = 16 * ldc;
    const bool row_ok_last = (MF - 1) * 16 + fr < nrows;       // rows of fragments 0 .. MF-2 are always inside the block
    auto store_pending = [&](int mf) {
        if (OUT_BF16 && pend_ptr && (mf < MF - 1 || row_ok_last) && !(probe & 2)) *(u32x4*)(pend_ptr + mf * mf_stride) = pend[mf];
    };

    // one k-step: A fragments of the NEXT step -> NXT, W fragments of step +4 -> the other ring half, MFMAs of this step
#define AR_STEP(CUR, NXT, J, AOFF, ST)                                                                        \
    do {                                                                                                      \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) NXT[mf] = *(const op16x8*)(smem + abase[mf] + (AOFF)); \
        _Pragma("unroll") for (int nf = 0; nf < NW; ++nf)                                                     \
            wb[((J) + 2) & 3][nf] = ((J) & 3) < 2 ? *(const op16x8*)(wcur[nf] + (((J) & 3) + 2) * 512)        \
                                                  : *(const op16x8*)(wnext[nf] + (((J) & 3) - 2) * 512);      \
        if ((ST) >= 0 && (ST) < MF) store_pending(ST);                                                        \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) _Pragma("unroll") for (int nf = 0; nf < NW; ++nf)   \
            acc[mf][nf] = rv_mfma16(wb[(J) & 3][nf], CUR[mf], acc[mf][nf]); \
        /* interleave: one memory instruction in the shadow of each MFMA */                                  \
        _Pragma("unroll") for (int i = 0; i < NW; ++i) {            /* W loads first: the longest latency */  \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                  \
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);                                                 \
        }                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MF; ++i) {                                                      \
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                                  \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                \
        }                                                                                                     \
        __builtin_amdgcn_sched_group_barrier(0x8, MF * NW - MF - NW, 0);                                      \
    } while (0)
    // a body of 8 k-steps; FIRST: the pending stores of the previous slice ride along, one per step
#define AR_BODY(FIRST)                                                                                        \
    do {                                                                                                      \
        AR_STEP(afA, afB, 0, 1 * 64, (FIRST) ? 0 : -1);                                                       \
        AR_STEP(afB, afA, 1, 2 * 64, (FIRST) ? 1 : -1);                                                       \
        AR_STEP(afA, afB, 2, 3 * 64, (FIRST) ? 2 : -1);                                                       \
        AR_STEP(afB, afA, 3, 4 * 64, (FIRST) ? 3 : -1);                                                       \
        half_advance();                                                                                       \
        AR_STEP(afA, afB, 4, 5 * 64, (FIRST) ? 4 : -1);                                                       \
        AR_STEP(afB, afA, 5, 6 * 64, (FIRST) ? 5 : -1);                                                       \
        AR_STEP(afA, afB, 6, 7 * 64, (FIRST) ? 6 : -1);                                                       \
        /* the last step of a body reads the first fragments of the next body (k-step kf0 + 8, or 0 of the next slice) */ \
        const int wrap = kf0 + 8 >= KF ? -(KF - 8) * 64 : 8 * 64;                                            \
        _Pragma("unroll") for (int mf = 0; mf < MF; ++mf) abase[mf] += wrap;                                  \
        AR_STEP(afB, afA, 7, 0, -1);                                                                          \
        half_advance();                                                                                       \
    } while (0)

    for (int si = 0; si < my_slices; ++si) {
        {
            const int kf0 = 0;
            AR_BODY(true);
        }
        for (int kf0 = 8; kf0 < KF; kf0 += 8) AR_BODY(false);
        // ---- epilogue of this slice: lane owns row (mf*16 + fr), columns n0 + nf*16 + kg*4 .. +3 ----
        const int n0 = (wave + 8 * si) * NW * 16;
        if constexpr (OUT_BF16 != 0 && NW == 2) {
            if (!res) {       // pack now, store during the next slice
                f32x4 b0 = f32x4{0.f, 0.f, 0.f, 0.f}, b1 = b0;
                if (bias) { b0 = *(const f32x4*)(bias + n0 + kg * 4); b1 = *(const f32x4*)(bias + n0 + 16 + kg * 4); }
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) {
                    f32x4 v0 = acc[mf][0] + b0, v1 = acc[mf][1] + b1;
                    acc[mf][0] = acc[mf][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v0[r] = rv_act_apply<ACT>(v0[r]); v1[r] = rv_act_apply<ACT>(v1[r]); }
                    }
                    const u32x2 w0 = pack_op16x4(v0), w1 = pack_op16x4(v1);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(w0[0], w1[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(w0[1], w1[1], false, false);
                    pend[mf] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
                pend_ptr = (op16_t*)Cv + (int64_t)(r0 + fr) * ldc + n0 + (kg & 1) * 16 + (kg >> 1) * 8;
                continue;
            }
        }
#pragma unroll
        for (int nf = 0; nf < NW; ++nf) {
            const int n = n0 + nf * 16 + kg * 4;
            f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias) bv = *(const f32x4*)(bias + n);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int row = mf * 16 + fr;
                f32x4 v = acc[mf][nf] + bv;
                acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (row >= nrows || (probe & 2)) continue;
                const int64_t m = r0 + row;
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + m * ldc + n) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + m * ldc + n) = v;
            }
        }
    }
    // the last slice's pieces
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) store_pending(mf);
#undef AR_BODY
#undef AR_STEP
}

int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

struct ArPlan {
    int rows_per_wg = 0, mf = 0, nw = 0, grid = 0;
    size_t lds = 0;
};

// rows per workgroup: even deal over (#CU x rounds) workgroups, bounded by the LDS a block of rows needs
ArPlan plan_for(int64_t M, int64_t N, int64_t K) {
    ArPlan p;
    if (K % 256 != 0 || K > 1024) return p;         // bodies of 8 k-steps
    if ((N / 32) % 8 == 0 && N % 32 == 0) p.nw = 2;  // 8 waves x 32-column slices
    else return p;
    const int cus = device_cus();
    const int rmax = (int)((AR_LDS_MAX - 1024) / (K * 2 + AR_PAD * 16));   // rows whose K fit (the copy rounds up to 1 KiB)
    const int rcap = rmax < 112 ? rmax : 112;
    if (rcap < 16) return p;
    const int64_t rounds = cdiv(M, (int64_t)cus * rcap);
    const int64_t wgs = (int64_t)cus * rounds;
    p.rows_per_wg = (int)cdiv(M, wgs);
    p.mf = (p.rows_per_wg + 15) / 16;
    p.grid = (int)cdiv(M, p.rows_per_wg);
    p.lds = (((size_t)p.rows_per_wg * (K * 2 + AR_PAD * 16)) + 1023) & ~(size_t)1023;
    return p;
}

// (the dynamic-LDS opt-in is a per-DEVICE attribute of the function: remembered per device, bit d of the mask)
template <typename Kern>
int set_lds(Kern k, std::atomic<uint64_t>& have) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(have.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)AR_LDS_MAX) != hipSuccess) {
            rv_set_error("gemm_arows: cannot reserve %d bytes of LDS", AR_LDS_MAX);
            return RV_ERR_HIP;
        }
        have.fetch_or(bit, std::memory_order_relaxed);
    }
    return RV_OK;
}

template <int OUT_BF16, int ACT, int MF, int NW>
int launch(const ArPlan& p, const op16_t* A, int64_t lda, const op16_t* Wp, const float* bias, const float* res, int64_t ldr, void* C,
           int64_t ldc, int M, int N, int K, hipStream_t st) {
    static std::atomic<uint64_t> have{0};
    if (int rc = set_lds(gemm_arows_kernel<OUT_BF16, ACT, MF, NW>, have)) return rc;
    hipLaunchKernelGGL((gemm_arows_kernel<OUT_BF16, ACT, MF, NW>), dim3(p.grid), dim3(512), p.lds, st, A, lda, Wp, bias, res, ldr, C, ldc, M,
                       N, K, p.rows_per_wg, rv_cur_opts().gemm_arows >> 4);
    return RV_OK;
}

template <int OUT_BF16, int ACT, int NW>
int launch_mf(const ArPlan& p, const op16_t* A, int64_t lda, const op16_t* Wp, const float* bias, const float* res, int64_t ldr, void* C,
              int64_t ldc, int M, int N, int K, hipStream_t st) {
    switch (p.mf) {
        case 4: return launch<OUT_BF16, ACT, 4, NW>(p, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, st);
        case 5: return launch<OUT_BF16, ACT, 5, NW>(p, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, st);
        case 6: return launch<OUT_BF16, ACT, 6, NW>(p, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, st);
        case 7: return launch<OUT_BF16, ACT, 7, NW>(p, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, st);
    }
    rv_set_error("gemm_arows: no kernel for %d row fragments", p.mf);
    return RV_ERR_ARG;
}

}  // namespace

// Policy: packed W, no gated epilogue, K in {256, 512, 768, 1024}, N a multiple of 8 wave slices of 32 columns, and enough rows
// that every CU gets at least 4 row fragments (M > 48 rows per CU; measured: with 2 - 3 fragments the W stream per MFMA is too
// large and the ring kernel wins): the [segments x frames] batches of the adapter and the dense projector.
bool gemm_arows_supported(int w_layout, int act, int64_t M, int64_t N, int64_t K) {
    if (w_layout != 1 || act == RV_ACT_SILU_MUL) return false;
    const ArPlan p = plan_for(M, N, K);
    return p.nw != 0 && p.mf >= 4 && p.mf <= 7;
}

int gemm_arows_launch(const void* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
                      int out_dtype, int act, int64_t M, int64_t N, int64_t K, hipStream_t st) {
    const ArPlan p = plan_for(M, N, K);
    const op16_t* a = (const op16_t*)A;
    const op16_t* w = (const op16_t*)Wp;
    const int ob = out_dtype == RV_OP16;
    int rc;
#define AR(OB, AC) rc = launch_mf<OB, AC, 2>(p, a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st)
    if (ob && act == RV_ACT_NONE) AR(1, RV_ACT_NONE);
    else if (ob && act == RV_ACT_RELU) AR(1, RV_ACT_RELU);
    else if (ob) AR(1, RV_ACT_QUICK_GELU);
    else if (act == RV_ACT_NONE) AR(0, RV_ACT_NONE);
    else if (act == RV_ACT_RELU) AR(0, RV_ACT_RELU);
    else AR(0, RV_ACT_QUICK_GELU);
#undef AR
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_arows");
    return RV_OK;
}
