// Synthetic-weight initialiser: the device twin of revisionllm_amd/utils/hashinit.py (bit-identical).
#include "common.h"

namespace {
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// v*step is exact in double (24 x 24 bits), so mul+add and a contracted fma give the same double; one
// final rounding to fp32.  numpy does the identical double arithmetic -> bit-identical on host and device.
__device__ __forceinline__ float hash_value(int v, float step, float base) {
    return (float)((double)v * (double)step + (double)base);
}

template <typename T>
__global__ void init_hash_kernel(T* dst, int64_t n, uint64_t key, float step, float base) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = splitmix64((uint64_t)i + key);
        const int v = (int)(h >> 40) - (1 << 23);
        const float w = hash_value(v, step, base);
        if (sizeof(T) == 2)
            dst[i] = (T)f32_to_op16(w);
        else
            dst[i] = (T)w;
    }
}
template <>
__global__ void init_hash_kernel<float>(float* dst, int64_t n, uint64_t key, float step, float base) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = splitmix64((uint64_t)i + key);
        const int v = (int)(h >> 40) - (1 << 23);
        dst[i] = hash_value(v, step, base);
    }
}
}  // namespace

extern "C" int rv_init_hash(void* dst, int dtype, int64_t n, uint64_t key, float step, float base, void* stream) {
    RV_CHECK_ARG(dst && n >= 0, "rv_init_hash: bad arguments");
    RV_CHECK_ARG(dtype == RV_F32 || dtype == RV_OP16, "rv_init_hash: dtype must be f32 or bf16");
    if (n == 0) return RV_OK;
    const int threads = 256;
    const int blocks = (int)(cdiv(n, threads) < 8192 ? cdiv(n, threads) : 8192);
    if (dtype == RV_F32)
        hipLaunchKernelGGL(init_hash_kernel<float>, dim3(blocks), dim3(threads), 0, as_stream(stream), (float*)dst, n, key, step, base);
    else
        hipLaunchKernelGGL(init_hash_kernel<op16_t>, dim3(blocks), dim3(threads), 0, as_stream(stream), (op16_t*)dst, n, key, step, base);
    RV_CHECK_LAUNCH("rv_init_hash");
    return RV_OK;
}
