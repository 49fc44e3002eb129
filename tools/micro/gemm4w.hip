// Feasibility probe: a 256 x 256 x 64 bf16 GEMM tile with FOUR waves (one per SIMD, 128 x 128 outputs each, 256 accumulator
// registers + two operand fragment sets in the 512-entry unified file), against the eight-wave ping-pong kernel of gemm_pp.hip
// (two waves per SIMD, 128 x 64 each).  Per k-tile the four waves read 128 KiB of operand fragments from LDS instead of 192 KiB.
// C[M,N] = A[M,K] . W[N,K]^T, A row-major bf16, W fragment-packed (csrc/kernels.h), C bf16.  Output-tiled, M % 256 == N % 256 == 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/gemm4w.hip -o tools/micro/gemm4w ; run: tools/micro/gemm4w [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef uint16_t bf16_t;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }
__device__ __forceinline__ bf16_t f2b(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2, LDS_BYTES = 5 * A_BYTES;   // two A stages + three W stages of 32 KiB

#ifndef INTERLEAVE
#define INTERLEAVE 1
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm4w(const bf16_t* __restrict__ A, int64_t lda,
                                                                                          const bf16_t* __restrict__ Wp, bf16_t* C, int64_t ldc,
                                                                                          int M, int N, int K, int tiles_m, int tiles_n, unsigned long long* dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;
    int tm, tn;
    {   // the m-tiles of one W column panel run on one XCD at about the same time (as gemm_pp)
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        tn = (idx / tiles_m) * 8 + xcd;
        tm = idx % tiles_m;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kfr = K >> 5, nks = K / BK;
    // DMA sources.  A piece p = 8 rows of 128 B (this wave: p = wave * 8 + i); lane -> row (lane >> 3), LDS slot (lane & 7) of the
    // row holds global chunk slot ^ (row & 7).  W piece (nfrag, ks) = one MFMA fragment (this wave: nfrag = wave * 4 + (i >> 1), ks = i & 1).
    const char* Ab = (const char*)A;
    const char* Wb = (const char*)Wp;
    unsigned a_off[8], w_off[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        a_off[i] = (unsigned)(((int64_t)(m0 + row) * lda + chunk * 8) * 2);
        const int nb = (n0 >> 4) + wave * 4 + (i >> 1);
        w_off[i] = (unsigned)(((((int64_t)nb * kfr + (i & 1)) * 64 + lane) * 8) * 2);
    }
    const int a_rd = (wr * 128 + fr) * 128;
    const int a_c[2] = {((kg ^ (fr & 7)) << 4), (((4 + kg) ^ (fr & 7)) << 4)};

    f32x4 acc[8][8];   // [nj][mi]
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int f = 0; f < 8; ++f) acc[j][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa[2][8], fw[2][8];

    // LDS: two A stages (32 KiB each) then three W stages (32 KiB each) = 160 KiB.  A(c + 2) and W(c + 3) are issued in the second half
    // of tile c (after its barrier: A stage c % 2 and W stage c % 3 have been read by every wave by then), i.e. the activations (L2 /
    // MALL) get one tile of latency budget and the weights (HBM) two.
#define A_STAGE(c) (smem + ((c) & 1) * A_BYTES)
#define W_STAGE(ws) (smem + 2 * A_BYTES + (ws) * A_BYTES)
#define ISSUE_A(KT, ST, i) glds16(Ab + (int64_t)(KT) * 128 + a_off[i], (ST) + (wave * 8 + (i)) * 1024)
#define ISSUE_W(KT, ST, i) glds16(Wb + (int64_t)(KT) * 2048 + w_off[i], (ST) + (wave * 8 + (i)) * 1024)
#define READ_A(B, ST, KS, f) fa[B][f] = *(const bf16x8*)((ST) + a_rd + (f) * 2048 + a_c[KS])
#define READ_W(B, ST, KS, j) fw[B][j] = *(const bf16x8*)((ST) + w_rd + (j) * 2048 + (KS) * 1024)
// accumulators pinned to AGPRs, operands to VGPRs (the register allocator otherwise trades them back and forth at this pressure)
#define MFMA1(B, j, f) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][f]) : "v"(fw[B][j]), "v"(fa[B][f]));
#define MFMA2(B, j, f) MFMA1(B, j, f) MFMA1(B, j, (f) + 1)
#define PIN __builtin_amdgcn_sched_barrier(0);
    const int w_rd = wc * 16384 + lane * 16;
    const int last = nks - 1;
    auto kclamp = [&](int k) { return k < last ? k : last; };
    // prologue: A(0), W(0), W(1), A(1), W(2) in that order (a wave's loads retire in order)
#pragma unroll
    for (int i = 0; i < 8; ++i) ISSUE_A(0, A_STAGE(0), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) ISSUE_W(0, W_STAGE(0), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) ISSUE_W(kclamp(1), W_STAGE(1), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) ISSUE_A(kclamp(1), A_STAGE(1), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) ISSUE_W(kclamp(2), W_STAGE(2), i);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 8; ++f) { READ_A(0, A_STAGE(0), 0, f); READ_W(0, W_STAGE(0), 0, f); }
    int ws = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int c = 0; c < nks; ++c) {
        char* a_cur = A_STAGE(c);
        char* a_nxt = A_STAGE(c + 1);
        char* w_cur = W_STAGE(ws);
        const int ws1 = ws == 2 ? 0 : ws + 1;
        char* w_nxt = W_STAGE(ws1);
        const int ka = kclamp(c + 2), kw = kclamp(c + 3);
        PIN
        // ---- first half: MFMAs on (c, ks 0), fragments of (c, ks 1) arrive
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            MFMA2(0, j, 0) PIN
            READ_A(1, a_cur, 1, j); MFMA2(0, j, 2) PIN
            READ_W(1, w_cur, 1, j); MFMA2(0, j, 4) PIN
            MFMA2(0, j, 6) PIN
        }
#ifndef NO_VMWAIT
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // A(c + 1), W(c + 1) landed; W(c + 2) may stay in flight
#endif
#ifndef NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        PIN
        // ---- second half: MFMAs on (c, ks 1); fragments of (c + 1, ks 0) arrive; A(c + 2) -> this tile's A stage, W(c + 3) -> its W stage
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            MFMA2(1, j, 0) PIN
            if (j < 4) { READ_A(0, a_nxt, 0, 2 * j); ISSUE_A(ka, a_cur, 2 * j); } else { READ_W(0, w_nxt, 0, 2 * (j - 4)); ISSUE_W(kw, w_cur, 2 * (j - 4)); }
            MFMA2(1, j, 2) PIN
            if (j < 4) { READ_A(0, a_nxt, 0, 2 * j + 1); } else { READ_W(0, w_nxt, 0, 2 * (j - 4) + 1); }
            MFMA2(1, j, 4) PIN
            if (j < 4) { ISSUE_A(ka, a_cur, 2 * j + 1); } else { ISSUE_W(kw, w_cur, 2 * (j - 4) + 1); }
            MFMA2(1, j, 6) PIN
        }
        ws = ws1;
    }
    asm volatile("s_waitcnt vmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    if (dbg && tid == 0) {
        dbg[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
        dbg[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    // epilogue: lane owns row m = .. + fr, columns n = .. + kg * 4 .. + 3 of every fragment
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const int m = m0 + wr * 128 + f * 16 + fr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wc * 128 + j * 16 + kg * 4;
            const f32x4 v = acc[j][f];
            const uint32_t lo = (uint32_t)f2b(v[0]) | ((uint32_t)f2b(v[1]) << 16), hi = (uint32_t)f2b(v[2]) | ((uint32_t)f2b(v[3]) << 16);
#ifdef NO_EPI
            if (m < M && lo == 0x12345678u) *(u32x2*)(C + (int64_t)m * ldc + n) = u32x2{lo, hi};
#else
            if (m < M) *(u32x2*)(C + (int64_t)m * ldc + n) = u32x2{lo, hi};
#endif
        }
    }
}

static float b2f(bf16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2b_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 4096, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    const int copies = 4;
    std::vector<bf16_t> hA((size_t)M * K), hW((size_t)N * K), hWp((size_t)N * K);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : hA) v = f2b_host(rnd());
    for (auto& v : hW) v = f2b_host(rnd() * 0.1f);
    const int kfr = K / 32;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const int nb = n >> 4, fr = n & 15, kf = k >> 5, kg = (k >> 3) & 3, j = k & 7;
            hWp[((((size_t)nb * kfr + kf) * 64) + kg * 16 + fr) * 8 + j] = hW[(size_t)n * K + k];
        }
    bf16_t *dA, *dW[copies], *dC;
    CK(hipMalloc(&dA, hA.size() * 2));
    CK(hipMalloc(&dC, (size_t)M * N * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < copies; ++i) {
        CK(hipMalloc(&dW[i], hWp.size() * 2));
        CK(hipMemcpy(dW[i], hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
    }
    const int tiles_m = M / BM, tiles_n = N / BN;
    if (M % BM || N % BN || tiles_n % 8 || K % BK) { printf("M, N multiples of 256 (N of 2048), K of 64\n"); return 1; }
    CK(hipFuncSetAttribute((const void*)gemm4w, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    unsigned long long* dDbg;
    CK(hipMalloc(&dDbg, (size_t)tiles_m * tiles_n * 16));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto launch = [&](int i) {
        hipLaunchKernelGGL(gemm4w, dim3(tiles_m * tiles_n), dim3(256), LDS_BYTES, st, dA, (int64_t)K, dW[i % copies], dC, (int64_t)N, M, N, K, tiles_m, tiles_n, dDbg);
    };
    launch(0);
    CK(hipStreamSynchronize(st));
    std::vector<bf16_t> hC((size_t)M * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int t = 0; t < 4000; ++t) {
        s = s * 1664525u + 1013904223u;
        const int m = (s >> 4) % M;
        s = s * 1664525u + 1013904223u;
        const int n = (s >> 4) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)b2f(hA[(size_t)m * K + k]) * b2f(hW[(size_t)n * K + k]);
        const double err = fabs(ref - b2f(hC[(size_t)m * N + n])) / (fabs(ref) + 1.0);
        if (err > worst) worst = err;
    }
    printf("max rel err over 4000 samples: %.3e %s\n", worst, worst < 1e-2 ? "OK" : "WRONG");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < (argc > 4 ? atoi(argv[4]) / 2 : 5); ++i) launch(i);
    const int iters = argc > 4 ? atoi(argv[4]) : 30;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    {
        std::vector<unsigned long long> hd((size_t)tiles_m * tiles_n * 2);
        CK(hipMemcpy(hd.data(), dDbg, hd.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, rt = 0;
        for (size_t i = 0; i < hd.size(); i += 2) { cyc += (double)hd[i]; rt += (double)hd[i + 1]; }
        const double n = hd.size() / 2.0;
        printf("main loop: %.0f shader cycles per k-tile (ideal 2048), %.3f us per k-tile, clock %.0f MHz\n", cyc / n / (K / BK), rt / n / 100.0 / (K / BK),
               cyc / rt * 100.0);
    }
    printf("M=%d N=%d K=%d: %.1f us  %.1f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    return 0;
}
