// v_cvt_f16_f32 against v_cvt_pk_f16_f32 (gfx950) over a sweep of f32 bit patterns, checked against the host's (_Float16) cast (IEEE RNE, denormals kept):
// do the two instructions round alike - in the fp16 DENORMAL range and on ties in particular?  (Round 6: f32_to_op16 went from the scalar to the packed form
// and test_init_hash_bit_exact[f16] found elements that differ.)    Build: hipcc --offload-arch=gfx950 -O3 tools/micro/cvt_f16.hip -o tools/micro/cvt_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float* in, unsigned short* a, unsigned short* b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned r0, r1;
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(r0) : "v"(in[i]));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r1) : "v"(in[i]));
    a[i] = (unsigned short)r0;
    b[i] = (unsigned short)r1;
}
int main() {
    std::vector<float> h;
    for (unsigned e = 127 - 30; e <= 127 + 17; ++e)            // 2^-30 .. 2^17
        for (unsigned m = 0; m < (1u << 23); m += 4099) {      // sparse mantissas + exact ties
            unsigned u = (e << 23) | m; float f; memcpy(&f, &u, 4); h.push_back(f); h.push_back(-f);
        }
    for (unsigned e = 127 - 26; e <= 127 + 15; ++e)            // exact half-way points at 10 / 11 .. bits
        for (unsigned s = 0; s < 16; ++s)
            for (unsigned j = 0; j < 64; ++j) {
                unsigned m = (j << 17) | (1u << (12 + 0)) ; m = (j << 13) + (1u << 12) + 0; unsigned u = (e << 23) | ((m << (s % 11)) & 0x7fffff); float f; memcpy(&f, &u, 4); h.push_back(f);
            }
    const int n = (int)h.size();
    float* din; unsigned short *da, *db;
    (void)hipMalloc(&din, n * 4); (void)hipMalloc(&da, n * 2); (void)hipMalloc(&db, n * 2);
    (void)hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, din, da, db, n);
    std::vector<unsigned short> a(n), b(n);
    (void)hipMemcpy(a.data(), da, n * 2, hipMemcpyDeviceToHost);
    (void)hipMemcpy(b.data(), db, n * 2, hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0, diff = 0, shown = 0;
    for (int i = 0; i < n; ++i) {
        _Float16 r = (_Float16)h[i]; unsigned short want; memcpy(&want, &r, 2);
        bad_a += a[i] != want; bad_b += b[i] != want; diff += a[i] != b[i];
        if ((a[i] != want || b[i] != want) && shown < 12) { printf("  f32 %.9g (%08x): host %04x  v_cvt_f16_f32 %04x  v_cvt_pk_f16_f32 %04x\n", h[i], *(unsigned*)&h[i], want, a[i], b[i]); ++shown; }
    }
    printf("%d values: v_cvt_f16_f32 differs from the host cast on %d, v_cvt_pk_f16_f32 on %d, the two from each other on %d\n", n, bad_a, bad_b, diff);
    return 0;
}
