// Token selection and segment scores, fused on the device so the decode loop needs no [B,V] host round trip:
//   rv_sample         HF warper chain (temperature -> top-k -> top-p), inverse-CDF draw / argmax, plus the
//                     entropy of the processed and of the raw next-token distribution
//   rv_entropy_stats  get_entropy_statistics (funs_get_feature_X.py:120-146) over stacked per-step scores
//   rv_topk_cosine    column-normalised top-k pooled cosine score (eval_nlq_retrieval_e2e2.py:380-386)
#include "kernels.h"

namespace {

constexpr int TPB = 1024;   // threads per row
constexpr int ITEMS = 32;   // V <= 32768
constexpr int KCAP = 64;

struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax better(ArgMax a, ArgMax b) {  // larger value, then smaller index
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b{__shfl_xor(a.v, o, 64), __shfl_xor(a.i, o, 64)};
        a = better(a, b);
    }
    return a;
}

__device__ __forceinline__ float block_sum(float v, float* sh) {  // sh: 16 floats; all threads get the result
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) t = fmaxf(t, sh[w]);
    return t;
}

// entropy of softmax(x) with the reference's formula H = -sum p*log(p + 1e-10); -inf entries give p = 0
__device__ float block_entropy(const float* x, int V, float* sh) {
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += TPB) mx = fmaxf(mx, x[i]);
    mx = block_max(mx, sh);
    float z = 0.f;
    for (int i = threadIdx.x; i < V; i += TPB) z += expf(x[i] - mx);
    z = block_sum(z, sh);
    const float iz = 1.0f / z;
    float hsum = 0.f;
    for (int i = threadIdx.x; i < V; i += TPB) {
        const float p = expf(x[i] - mx) * iz;
        hsum += p * logf(p + 1e-10f);
    }
    return -block_sum(hsum, sh);
}

__global__ __launch_bounds__(TPB) void sample_kernel(const float* __restrict__ logits, int V, const float* __restrict__ uniforms,
                                                     int do_sample, float temperature, int top_k, float top_p,
                                                     int32_t* __restrict__ out_tok, float* __restrict__ out_hp,
                                                     float* __restrict__ out_hr, int32_t* __restrict__ out_idx,
                                                     float* __restrict__ out_val, int32_t* __restrict__ out_nkeep) {
    __shared__ float sh[16];
    __shared__ float topv[KCAP];
    __shared__ int topi[KCAP];
    __shared__ float e[KCAP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (int64_t)b * V;

    const float h_raw = block_entropy(x, V, sh);
    if (tid == 0) out_hr[b] = h_raw;

    float v[ITEMS];
    const float inv_t = do_sample ? 1.0f / temperature : 1.0f;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = tid + j * TPB;
        v[j] = i < V ? x[i] * inv_t : -INFINITY;
    }
    const int K = do_sample ? top_k : 1;
    // ---- top-K selection: radix select on order-preserving integer keys (2 bits per round, pure register counting +
    // one block reduction per round), then the <= K survivors are rank-sorted by (score desc, index asc).
    __shared__ int cnt_sh[16][4];
    __shared__ int n_list, n_eq_taken;
    __shared__ unsigned list_key[KCAP];
    __shared__ int list_idx[KCAP];
    unsigned key[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const unsigned u = __float_as_uint(v[j]);
        key[j] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        if (tid + j * TPB >= V) key[j] = 0u;   // below every real score (-inf maps to 0x007fffff)
    }
    unsigned prefix = 0u;   // bits of the K-th largest key decided so far
    int need = K;           // how many of the keys matching the prefix are still wanted
    for (int shift = 30; shift >= 0; shift -= 2) {
        const unsigned himask = shift == 30 ? 0u : (0xffffffffu << (shift + 2));
        int c1 = 0, c2 = 0, c3 = 0;   // keys (matching the prefix) whose next 2 bits are >= 1, >= 2, >= 3
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool m = (key[j] & himask) == prefix;
            const unsigned d = (key[j] >> shift) & 3u;
            c1 += (m && d >= 1u);
            c2 += (m && d >= 2u);
            c3 += (m && d >= 3u);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            c1 += __shfl_xor(c1, o, 64);
            c2 += __shfl_xor(c2, o, 64);
            c3 += __shfl_xor(c3, o, 64);
        }
        __syncthreads();
        if (lane == 0) {
            cnt_sh[wave][1] = c1;
            cnt_sh[wave][2] = c2;
            cnt_sh[wave][3] = c3;
        }
        __syncthreads();
        int t1 = 0, t2 = 0, t3 = 0;
#pragma unroll
        for (int w = 0; w < TPB / 64; ++w) {
            t1 += cnt_sh[w][1];
            t2 += cnt_sh[w][2];
            t3 += cnt_sh[w][3];
        }
        // digit of the need-th largest key among the matching ones
        unsigned dsel;
        if (t3 >= need) dsel = 3u;
        else if (t2 >= need) { dsel = 2u; need -= t3; }
        else if (t1 >= need) { dsel = 1u; need -= t2; }
        else { dsel = 0u; need -= t1; }
        prefix |= dsel << shift;
    }
    // prefix = K-th largest key; `need` of the keys equal to it are wanted (the ones with the smallest indices)
    if (tid == 0) {
        n_list = 0;
        n_eq_taken = 0;
    }
    __syncthreads();
    int n_eq_local = 0;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) n_eq_local += (key[j] == prefix);
    const int n_eq_total = (int)(block_sum((float)n_eq_local, sh) + 0.5f);   // <= 32768: exact in fp32
    const bool take_all_eq = n_eq_total == need;   // the usual case: no tie straddles the K-th place
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        if (key[j] > prefix || (take_all_eq && key[j] == prefix)) {
            const int p = atomicAdd(&n_list, 1);
            if (p < KCAP) {
                list_key[p] = key[j];
                list_idx[p] = tid + j * TPB;
            }
        }
    }
    __syncthreads();
    if (!take_all_eq) {
        // ties at the K-th value: keep the `need` smallest indices.  Walk the index space in ascending order: item j of
        // every thread covers indices [j*TPB, (j+1)*TPB) in (wave, lane) order.
        for (int j = 0; j < ITEMS; ++j) {
            const bool eq = key[j] == prefix;
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) cnt_sh[wave][0] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < TPB / 64; ++w) {
                before += w < wave ? cnt_sh[w][0] : 0;
                total += cnt_sh[w][0];
            }
            const int taken = n_eq_taken;
            if (eq) {
                const int r = taken + before + __popcll(bal & ((1ull << lane) - 1ull));
                if (r < need) {
                    const int p = (K - need) + r;
                    list_key[p] = key[j];
                    list_idx[p] = tid + j * TPB;
                }
            }
            __syncthreads();
            if (tid == 0) n_eq_taken = taken + total;
            __syncthreads();
            if (n_eq_taken >= need) break;
        }
        __syncthreads();
    }
    if (tid < 64) {
        // rank sort of the K candidates: (key desc, index asc)
        const unsigned kk = tid < K ? list_key[tid] : 0u;
        const int ii = tid < K ? list_idx[tid] : 0x7fffffff;
        int rank = 0;
        for (int q = 0; q < K; ++q) {
            const unsigned kq = list_key[q];
            const int iq = list_idx[q];
            rank += (kq > kk) || (kq == kk && iq < ii);
        }
        if (tid < K) {
            const unsigned u = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk;
            topv[rank] = __uint_as_float(u);
            topi[rank] = ii;
        }
    }
    __syncthreads();
    if (!do_sample) {
        if (tid == 0) {
            out_tok[b] = topi[0];
            out_hp[b] = h_raw;
            out_nkeep[b] = 0;
        }
        if (tid < KCAP && out_idx && out_val) {
            out_idx[b * KCAP + tid] = -1;
            out_val[b * KCAP + tid] = -INFINITY;
        }
        return;
    }
    if (tid == 0) {
        // sequential, in the order HF's TopPLogitsWarper accumulates (ascending probability)
        const float mx = topv[0];
        float z = 0.f;
        for (int i = 0; i < K; ++i) {
            e[i] = expf(topv[i] - mx);
            z += e[i];
        }
        int keep = K;
        if (top_p < 1.0f) {
            float cum = 0.f;
            keep = 1;
            for (int i = K - 1; i >= 1; --i) {
                cum += e[i] / z;
                if (cum > 1.0f - top_p) {
                    keep = i + 1;
                    break;
                }
            }
        }
        float z2 = 0.f;
        for (int i = 0; i < keep; ++i) z2 += e[i];
        float hp = 0.f, cum = 0.f;
        const float u = uniforms ? uniforms[b] : 0.f;
        int pos = 0;
        for (int i = 0; i < keep; ++i) {
            const float p = e[i] / z2;
            hp += p * logf(p + 1e-10f);
            cum += p;
            if (cum <= u) pos = i + 1;
        }
        pos = pos < keep - 1 ? pos : keep - 1;
        out_tok[b] = topi[pos];
        out_hp[b] = -hp;
        out_nkeep[b] = keep;
    }
    if (tid < KCAP) {
        out_idx[b * KCAP + tid] = tid < K ? topi[tid] : -1;
        out_val[b * KCAP + tid] = tid < K ? topv[tid] : -INFINITY;
    }
}

__global__ __launch_bounds__(TPB) void entropy_stats_kernel(const float* __restrict__ logits, int G, int V, float* __restrict__ out) {
    __shared__ float sh[16];
    extern __shared__ float hs[];  // G entropies
    const int b = blockIdx.x;
    for (int g = 0; g < G; ++g) {
        const float h = block_entropy(logits + ((int64_t)b * G + g) * V, V, sh);
        if (threadIdx.x == 0) hs[g] = h;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mx = -INFINITY, mn = INFINITY, s = 0.f;
        for (int g = 0; g < G; ++g) {
            mx = fmaxf(mx, hs[g]);
            mn = fminf(mn, hs[g]);
            s += hs[g];
        }
        const float mean = s / (float)G;
        float q = 0.f;
        for (int g = 0; g < G; ++g) q += (hs[g] - mean) * (hs[g] - mean);
        out[b * 4 + 0] = mx;
        out[b * 4 + 1] = mn;
        out[b * 4 + 2] = mean;
        out[b * 4 + 3] = G > 1 ? sqrtf(q / (float)(G - 1)) : __int_as_float(0x7fc00000);
    }
}

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }

// one block (1024 threads) per segment: column norms over frames, sims[t] = <f_t / norm, q>, sum of the k largest
// (k <= 0: mean).  Phase 1 splits the frames over 4 thread groups per column block to keep ~T/4 loads per thread.
template <typename T>
__global__ __launch_bounds__(1024) void topk_cosine_kernel(const T* __restrict__ feat, const float* __restrict__ q, int Tn, int d,
                                                           int k, float* __restrict__ out) {
    extern __shared__ float smem[];
    float* qn = smem;            // [d] q / column norm
    float* psum = smem + d;      // [4][d] partial sums of squares (summed in fixed order: deterministic)
    float* sims = smem + 5 * d;  // [Tn]
    const T* f = feat + (int64_t)blockIdx.x * Tn * d;
    {
        const int part = threadIdx.x >> 8, c0 = threadIdx.x & 255;   // 4 frame partitions x 256 column lanes
        for (int c = c0; c < d; c += 256) {
            float s = 0.f;
            for (int t = part; t < Tn; t += 4) {
                const float v = ld<T>(f + (int64_t)t * d + c);
                s += v * v;
            }
            psum[part * d + c] = s;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 1024) qn[c] = q[c] / sqrtf((psum[c] + psum[d + c]) + (psum[2 * d + c] + psum[3 * d + c]));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < Tn; t += 16) {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s += ld<T>(f + (int64_t)t * d + c) * qn[c];
        s = wave_sum(s);
        if (lane == 0) sims[t] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.f;
        if (k <= 0) {
            for (int t = 0; t < Tn; ++t) acc += sims[t];
            acc /= (float)Tn;
        } else {
            for (int r = 0; r < k && r < Tn; ++r) {
                int bi = 0;
                float bv = -INFINITY;
                for (int t = 0; t < Tn; ++t)
                    if (sims[t] > bv) {
                        bv = sims[t];
                        bi = t;
                    }
                acc += bv;
                sims[bi] = -INFINITY;
            }
        }
        out[blockIdx.x] = acc;
    }
}

}  // namespace

extern "C" int rv_sample(const float* logits, int32_t B, int32_t V, const float* uniforms, int32_t do_sample, float temperature,
                         int32_t top_k, float top_p, int32_t* out_tokens, float* out_entropy_proc, float* out_entropy_raw,
                         int32_t* out_topk_idx, float* out_topk_val, int32_t* out_nkeep, void* stream) {
    RV_CHECK_ARG(logits && out_tokens && out_entropy_proc && out_entropy_raw && out_nkeep, "rv_sample: null output");
    RV_CHECK_ARG(B > 0 && V > 0 && V <= TPB * ITEMS, "rv_sample: V=%d exceeds %d", V, TPB * ITEMS);
    if (do_sample) {
        RV_CHECK_ARG(top_k >= 1 && top_k <= KCAP, "rv_sample: top_k=%d must be in [1,%d] when sampling", top_k, KCAP);
        RV_CHECK_ARG(temperature > 0.f && top_p > 0.f, "rv_sample: temperature and top_p must be positive");
        RV_CHECK_ARG(out_topk_idx && out_topk_val, "rv_sample: candidate outputs required when sampling");
    }
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(TPB), 0, as_stream(stream), logits, V, uniforms, do_sample, temperature, top_k,
                       top_p, out_tokens, out_entropy_proc, out_entropy_raw, out_topk_idx, out_topk_val, out_nkeep);
    RV_CHECK_LAUNCH("rv_sample");
    return RV_OK;
}

extern "C" int rv_entropy_stats(const float* logits, int32_t B, int32_t G, int32_t V, float* out, void* stream) {
    RV_CHECK_ARG(logits && out && B > 0 && G > 0 && V > 0, "rv_entropy_stats: bad arguments");
    RV_CHECK_ARG(G <= 8192, "rv_entropy_stats: G=%d too large", G);
    hipLaunchKernelGGL(entropy_stats_kernel, dim3(B), dim3(TPB), G * sizeof(float), as_stream(stream), logits, G, V, out);
    RV_CHECK_LAUNCH("rv_entropy_stats");
    return RV_OK;
}

extern "C" int rv_topk_cosine(const void* feat, int feat_dtype, const float* q_cls, int32_t n, int32_t T, int32_t d, int32_t k,
                              float* out, void* stream) {
    RV_CHECK_ARG(feat && q_cls && out && n > 0 && T > 0 && d > 0, "rv_topk_cosine: bad arguments");
    RV_CHECK_ARG((size_t)(5 * d + T) * 4 <= 64 * 1024, "rv_topk_cosine: 5*d + T too large for LDS");
    const size_t sm = (size_t)(5 * d + T) * sizeof(float);
    if (feat_dtype == RV_BF16)
        hipLaunchKernelGGL(topk_cosine_kernel<bf16_t>, dim3(n), dim3(1024), sm, as_stream(stream), (const bf16_t*)feat, q_cls, T, d, k, out);
    else if (feat_dtype == RV_F32)
        hipLaunchKernelGGL(topk_cosine_kernel<float>, dim3(n), dim3(1024), sm, as_stream(stream), (const float*)feat, q_cls, T, d, k, out);
    else {
        rv_set_error("rv_topk_cosine: dtype must be f32 or bf16");
        return RV_ERR_ARG;
    }
    RV_CHECK_LAUNCH("rv_topk_cosine");
    return RV_OK;
}
