// Can a subset of the CUs keep HBM busy while the rest do something else?  Streams with CU masks (hipExtStreamCreateWithCUMask):
// (1) streaming-read rate of a 180 MB buffer (rotating over 6 GB) on n of the 256 CUs; (2) the same while an MFMA-bound
// kernel runs on the complementary CUs, and that kernel's rate alone vs alongside.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/cumask.hip -o tools/micro/cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(8))) unsigned short bf16x8;

__global__ __launch_bounds__(512) void rd(const u32x4* __restrict__ p, size_t n16, size_t chunk16, unsigned* sink) {
    u32x4 a = {0, 0, 0, 0};
    for (size_t c = blockIdx.x; c * chunk16 < n16; c += gridDim.x) {   // persistent-ish: blocks loop over chunks
        const size_t base = c * chunk16;
        for (size_t i = threadIdx.x; i < chunk16; i += 512 * 8) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t j = i + (size_t)u * 512;
                v[u] = (j < chunk16 && base + j < n16) ? __builtin_nontemporal_load(p + base + j) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a ^= v[u];
        }
    }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) *sink = 1;
}

// MFMA-bound filler: each wave spins on register-resident MFMAs
__global__ __launch_bounds__(256) void mfma_burn(int iters, float* out) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (unsigned short)(0x3f80 + threadIdx.x % 3); b[i] = (unsigned short)(0x3f00 + i); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 123.456f) *out = s;
}

static hipStream_t masked_stream(int first_cu_per_xcd, int n_cu_per_xcd) {
    // 256 CUs = 8 XCDs x 32; mask bit index = CU id.  Assume CU ids are XCD-interleaved or XCD-major?  Use a mask that takes
    // the same relative CUs in every group of 32 under either convention: bits [first, first + n) of every 32-bit word.
    uint32_t w = 0;
    for (int i = 0; i < n_cu_per_xcd; ++i) w |= 1u << (first_cu_per_xcd + i);
    std::vector<uint32_t> mask(8, w);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask.data()) != hipSuccess) { printf("cu mask stream failed\n"); exit(1); }
    return s;
}

int main() {
    const size_t bytes = (size_t)180 << 20, pool = (size_t)6 << 30;
    const int nbuf = pool / bytes;
    char* d; unsigned* sink; float* fout;
    hipMalloc(&d, pool); hipMalloc(&sink, 4); hipMalloc(&fout, 4);
    hipMemset(d, 1, pool);
    hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    const size_t n16 = bytes / 16, chunk16 = (size_t)(128 << 10) / 16;
    for (int ncu : {32, 16, 12, 8, 6, 4}) {   // CUs per XCD given to the streaming kernel
        hipStream_t sr = masked_stream(0, ncu), sm = ncu < 32 ? masked_stream(ncu, 32 - ncu) : nullptr;
        const unsigned grid = (unsigned)(ncu * 8 * 2);   // two resident blocks per CU, looping over chunks
        auto stream_reads = [&](int reps) {
            for (int r = 0; r < reps; ++r) rd<<<grid, 512, 0, sr>>>((const u32x4*)(d + (size_t)(r % nbuf) * bytes), n16, chunk16, sink);
        };
        stream_reads(4); hipStreamSynchronize(sr);
        hipEventRecord(e0, sr); stream_reads(20); hipEventRecord(e1, sr); hipStreamSynchronize(sr);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("stream on %3d CUs alone: %.2f TB/s", ncu * 8, bytes * 20 / ms / 1e9);
        if (sm) {
            const int mcus = (32 - ncu) * 8, iters = 200000;
            const double flop = (double)mcus * 8 /*blocks per CU*/ * 4 /*waves*/ * iters * 8.0 * 16 * 16 * 32 * 2;
            mfma_burn<<<mcus * 8, 256, 0, sm>>>(1000, fout); hipStreamSynchronize(sm);
            hipEventRecord(f0, sm); mfma_burn<<<mcus * 8, 256, 0, sm>>>(iters, fout); hipEventRecord(f1, sm); hipStreamSynchronize(sm);
            float ms_alone; hipEventElapsedTime(&ms_alone, f0, f1);
            hipEventRecord(f0, sm); mfma_burn<<<mcus * 8, 256, 0, sm>>>(iters, fout); hipEventRecord(f1, sm);
            hipEventRecord(e0, sr); stream_reads(40); hipEventRecord(e1, sr);
            hipStreamSynchronize(sr); hipStreamSynchronize(sm);
            float ms_m, ms_r; hipEventElapsedTime(&ms_m, f0, f1); hipEventElapsedTime(&ms_r, e0, e1);
            printf(" | with MFMA burn on the other %3d CUs: stream %.2f TB/s, MFMA %.0f TF/s (alone %.0f)", mcus, bytes * 40 / ms_r / 1e9,
                   flop / ms_m / 1e9, flop / ms_alone / 1e9);
        }
        printf("\n");
    }
    return 0;
}
