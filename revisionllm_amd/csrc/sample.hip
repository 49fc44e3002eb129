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

__device__ __forceinline__ unsigned block_max_u32(unsigned v, unsigned* shu) {  // shu: 16 words; all threads get the result
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned w = (unsigned)__shfl_xor((int)v, o, 64);
        v = w > v ? w : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) shu[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned t = 0u;
#pragma unroll
    for (int w = 0; w < TPB / 64; ++w) t = shu[w] > t ? shu[w] : t;
    return t;
}

// K-th largest of the block's keys, ONE key per thread, 4 bits per round and one barrier per round: five ballots give lane
// c < 16 of every wave the number of its wave's matching keys whose digit is c; the 16 x 16 wave histograms meet in LDS
// (double-buffered by round parity), every wave adds them up and suffix-scans them redundantly, so the choice of the digit
// is wave-uniform without a second barrier.  -> the want-th largest key; need_out = how many of the keys EQUAL to it are
// among the `want` largest (exact form).  Requires want <= number of threads.
template <bool BOUND_ONLY>
__device__ __forceinline__ unsigned select_kth_per_thread(unsigned k, int want, int (*whist)[16][TPB / 64], int& need_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned prefix = 0u;
    int need = want;
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
        const int shift = 28 - 4 * r;
        const unsigned himask = r == 0 ? 0u : (0xffffffffu << (shift + 4));
        const bool m = (k & himask) == prefix;
        const unsigned d = (k >> shift) & 15u;
        unsigned long long sel = __ballot(m);
        const unsigned long long b0 = __ballot((d & 1u) != 0u), b1 = __ballot((d & 2u) != 0u), b2 = __ballot((d & 4u) != 0u),
                                 b3 = __ballot((d & 8u) != 0u);
        sel &= (lane & 1) ? b0 : ~b0;
        sel &= (lane & 2) ? b1 : ~b1;
        sel &= (lane & 4) ? b2 : ~b2;
        sel &= (lane & 8) ? b3 : ~b3;
        if (lane < 16) whist[r & 1][lane][wave] = __popcll(sel);
        __syncthreads();
        int ge = 0;
        if (lane < 16) {
            const int4* hp = reinterpret_cast<const int4*>(&whist[r & 1][lane][0]);   // this digit's count in the 16 waves
#pragma unroll
            for (int q = 0; q < TPB / 256; ++q) {
                const int4 t = hp[q];
                ge += (t.x + t.y) + (t.z + t.w);
            }
        }
        // suffix sums over lanes 0..15 (one DPP row): ge[c] = matching keys with digit >= c
        ge += __builtin_amdgcn_update_dpp(0, ge, 0x101, 0xf, 0xf, true);   // row_shl:1 (lane i reads lane i + 1, 0 past the row)
        ge += __builtin_amdgcn_update_dpp(0, ge, 0x102, 0xf, 0xf, true);
        ge += __builtin_amdgcn_update_dpp(0, ge, 0x104, 0xf, 0xf, true);
        ge += __builtin_amdgcn_update_dpp(0, ge, 0x108, 0xf, 0xf, true);
        const unsigned ok = (unsigned)(__ballot(lane < 16 && ge >= need) & 0xffffull);   // bit 0 is always set
        const int dsel = 31 - __clz((int)ok);
        const int above = __builtin_amdgcn_readlane(ge, dsel < 15 ? dsel + 1 : 0);
        const int spare = __builtin_amdgcn_readlane(ge, dsel) - need;   // keys with this digit or a larger one beyond the wanted
        need -= dsel < 15 ? above : 0;
        prefix |= (unsigned)dsel << shift;
        // BOUND_ONLY: the caller only needs a lower bound of the want-th largest key with few keys above it - the prefix with
        // zero low bits is one as soon as at most 4 keys more than wanted are >= it
        if (BOUND_ONLY && spare <= 4) break;
    }
    need_out = need;
    return prefix;
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
                                                     float* __restrict__ out_val, int32_t* __restrict__ out_nkeep, float* __restrict__ out_thr, int variant) {
    __shared__ float sh[16];
    __shared__ unsigned shu[16];
    __shared__ float topv[KCAP];
    __shared__ int topi[KCAP];
    __shared__ float e[KCAP];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* x = logits + (int64_t)b * V;
#ifdef RV_SAMPLE_PROBE
    long long tp[12];
    for (int q = 0; q < 12; ++q) tp[q] = 0;
    tp[0] = wall_clock64();
#endif

    // the row is read ONCE into registers (32 independent loads per thread); the raw entropy uses the same per-thread
    // element order and block reductions as block_entropy(), so it is bit-identical to rv_entropy_stats on the same row
    float v[ITEMS];
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) {
        const int i = tid + j * TPB;
        v[j] = i < V ? x[i] : -INFINITY;
    }
    float h_raw;
    {
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) mx = fmaxf(mx, v[j]);
        mx = block_max(mx, sh);
        float ex[ITEMS];
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            ex[j] = expf(v[j] - mx);   // exp(-inf) = 0 for the padding
            z += ex[j];
        }
        z = block_sum(z, sh);
        const float iz = 1.0f / z;
        float hsum = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const float p = ex[j] * iz;
            if (tid + j * TPB < V) hsum += p * logf(p + 1e-10f);
        }
        h_raw = -block_sum(hsum, sh);
    }
    if (tid == 0) out_hr[b] = h_raw;
#ifdef RV_SAMPLE_PROBE
    tp[1] = wall_clock64();
#endif

    const float inv_t = do_sample ? 1.0f / temperature : 1.0f;
#pragma unroll
    for (int j = 0; j < ITEMS; ++j) v[j] = tid + j * TPB < V ? v[j] * inv_t : -INFINITY;
    if (do_sample && (top_k == 0 || top_k > KCAP)) {
        // ---- top_k = 0: HF's "filter disabled" (TopKLogitsWarper is not instantiated) - every token is a candidate, only top-p trims.  No list of
        // candidates can hold V entries, so nothing is sorted: the two places where the reference's order matters (the ascending cumulative
        // sum of TopPLogitsWarper, the inverse-CDF walk in descending order) are answered by a 32-round binary descent over the order-preserving
        // integer keys, each round one block-wide sum of the probabilities on one side of the candidate key.  Sums run in the block's
        // reduction order, not sequentially: a draw whose uniform sits within rounding of a CDF step may differ from the oracle's (as for the
        // K <= 64 path, tests keep uniforms away from the steps).
        unsigned key[ITEMS];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const unsigned u = __float_as_uint(v[j]);
            key[j] = tid + j * TPB < V ? ((u & 0x80000000u) ? ~u : (u | 0x80000000u)) : 0u;
            mx = fmaxf(mx, v[j]);
        }
        mx = block_max(mx, sh);
        // top_k > KCAP (no list can hold the candidates either): TopKLogitsWarper removes every score BELOW the top_k-th largest one (a tie at that
        // place stays whole) - the same descent, over counts: kcut = the largest key that at least top_k keys reach; what lies below it gets no mass
        // and, through `cut` >= kcut, no part in anything that follows.  top_k >= V removes nothing (HF: min(top_k, V)).
        unsigned kcut = 0u;
        if (top_k > 0 && top_k < V) {
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned c2 = kcut | (1u << bit);
                float reach = 0.f;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) reach += key[j] >= c2 ? 1.f : 0.f;       // (padding keys are 0: never >= c2; counts <= 32768 are exact)
                reach = block_sum(reach, sh);
                if (reach >= (float)top_k) kcut = c2;
            }
        }
        float ex[ITEMS];
        float z = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            ex[j] = key[j] >= kcut ? expf(v[j] - mx) : 0.f;
            z += ex[j];
        }
        z = block_sum(z, sh);
        unsigned kmax = 0u;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) kmax = key[j] > kmax ? key[j] : kmax;
        kmax = block_max_u32(kmax, shu);                         // the largest key: never removed (HF keeps at least one token)
        // top-p: remove the ascending prefix whose inclusive cumulative probability stays <= 1 - top_p; keep from key `cut` on
        unsigned cut = 0u;
        if (top_p >= 1.0f) cut = kcut;
        else {               // (the keys below kcut carry no mass: the descent ends at or above it)
            const float t = (1.0f - top_p) * z;                 // compare un-normalised sums
            for (int bit = 31; bit >= 0; --bit) {
                const unsigned c2 = cut | (1u << bit);
                float below = 0.f;
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) below += key[j] < c2 ? ex[j] : 0.f;
                below = block_sum(below, sh);
                if (below <= t && c2 <= kmax) cut = c2;
            }
        }
        float z2 = 0.f, cnt = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool kept = key[j] >= cut && tid + j * TPB < V;
            z2 += kept ? ex[j] : 0.f;
            cnt += kept ? 1.f : 0.f;
        }
        z2 = block_sum(z2, sh);
        const int n_keep = (int)(block_sum(cnt, sh) + 0.5f);
        const float iz2 = 1.0f / z2;
        float hsum = 0.f, thr = INFINITY;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            if (key[j] >= cut && tid + j * TPB < V) {
                const float p = ex[j] * iz2;
                hsum += p * logf(p + 1e-10f);
                thr = fminf(thr, v[j]);
            }
        }
        const float hp = -block_sum(hsum, sh);
        thr = -block_max(-thr, sh);                              // the smallest kept processed score
        // draw: the token whose descending inclusive cumulative probability first exceeds u = the largest key c with sum_{kept, key >= c} p > u
        const float u = (uniforms ? uniforms[b] : 0.f) * z2;
        unsigned pick = 0u;                                        // greedy from the top bit: the largest c whose kept mass at or above c exceeds u
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned c2 = pick | (1u << bit);
            float at_or_above = 0.f;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) at_or_above += (key[j] >= c2 && key[j] >= cut) ? ex[j] : 0.f;
            at_or_above = block_sum(at_or_above, sh);
            if (at_or_above > u) pick = c2;
        }
        // `pick` is an existing kept key (the sum above is a step function of the key) unless rounding left no key with a sum above u: then the
        // LAST kept token (the smallest kept key) is taken, as the oracle's clamp does.  Ties share a key: the reference orders them by index -
        // the n-th of them with n = how many whole tied probabilities fit between the sum above the tie and u
        float above = 0.f, tie_e = 0.f, ties = 0.f;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) ties += (tid + j * TPB < V && key[j] == pick && key[j] >= cut) ? 1.f : 0.f;
        int n_ties = (int)(block_sum(ties, sh) + 0.5f);
        if (n_ties == 0) {
            unsigned lowest = 0xffffffffu;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j)
                if (tid + j * TPB < V && key[j] >= cut) lowest = key[j] < lowest ? key[j] : lowest;
            pick = ~block_max_u32(~lowest, shu);
            ties = 0.f;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) ties += (tid + j * TPB < V && key[j] == pick) ? 1.f : 0.f;
            n_ties = (int)(block_sum(ties, sh) + 0.5f);
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const bool kept = key[j] >= cut && tid + j * TPB < V;
            above += (kept && key[j] > pick) ? ex[j] : 0.f;
            if (kept && key[j] == pick) tie_e = ex[j];
        }
        above = block_sum(above, sh);
        tie_e = block_max(tie_e, sh);
        int nth = 0;
        if (n_ties > 1 && tie_e > 0.f) {
            nth = (int)floorf((u - above) / tie_e);
            nth = nth < 0 ? 0 : (nth > n_ties - 1 ? n_ties - 1 : nth);
        }
        // the nth smallest index among the tokens whose key == pick: nth + 1 block-wide minimum searches (nth = 0 unless scores tie exactly)
        int last = -1, chosen = -1;
        for (int r = 0; r <= nth; ++r) {
            int best = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int i = tid + j * TPB;
                if (i < V && key[j] == pick && key[j] >= cut && i > last) best = i < best ? i : best;
            }
            best = -(int)block_max((float)(-best), sh);           // indices < 2^24: exact in fp32
            chosen = last = best;
        }
        if (tid == 0) {
            out_tok[b] = chosen >= 0 && chosen < V ? chosen : 0;
            out_hp[b] = hp;
            out_nkeep[b] = n_keep;
            if (out_thr) out_thr[b] = thr;
        }
        if (tid < KCAP && out_idx && out_val) {                   // no candidate list in this mode: the kept set is {score >= out_threshold}
            out_idx[b * KCAP + tid] = -1;
            out_val[b * KCAP + tid] = -INFINITY;
        }
        return;
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
    // Fast path (the decode shape: every thread owns a real score, K <= 64): the K-th largest of the 1024 per-thread
    // maxima is a lower bound L of the K-th largest score, and the scores >= L are few (K plus the runners-up that share a
    // thread with a larger one); they are compacted into LDS and the exact K-th largest is selected among them with one key
    // per thread.  Both selections cost 8 one-barrier rounds; the 16 register-counting rounds over all 32 keys per thread
    // below are the general path (short rows, > 1024 candidates, or a tie straddling the K-th place).
    __shared__ __attribute__((aligned(16))) int whist[2][16][TPB / 64];
    __shared__ unsigned cand_key[TPB];
    __shared__ int cand_idx[TPB];
    __shared__ int n_cand;
    bool fast = variant != 0 && V >= TPB && K <= KCAP;
    if (tid == 0) {
        n_list = 0;
        n_eq_taken = 0;
        n_cand = 0;
    }
    if (fast) {
        unsigned tmax = 0u;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) tmax = key[j] > tmax ? key[j] : tmax;
        int unused;
        const unsigned L = select_kth_per_thread<true>(tmax, K, whist, unused);
#ifdef RV_SAMPLE_PROBE
        tp[5] = wall_clock64();
#endif
        int c_local = 0;
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) c_local += (key[j] >= L);
        const int C = (int)(block_sum((float)c_local, sh) + 0.5f);   // <= 32768: exact in fp32
#ifdef RV_SAMPLE_PROBE
        tp[6] = wall_clock64();
#endif
        if (C > TPB) {
            fast = false;
        } else {
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                if (key[j] >= L) {
                    const int p = atomicAdd(&n_cand, 1);
                    cand_key[p] = key[j];
                    cand_idx[p] = tid + j * TPB;
                }
            }
            __syncthreads();
#ifdef RV_SAMPLE_PROBE
            tp[7] = wall_clock64();
#endif
            if (C <= 64) {
                // the usual case: the candidates fit one wave - the rank sort below orders all of them by (score desc, index
                // asc) and keeps the first K, which also settles ties across the K-th place
                if (tid < C) {
                    list_key[tid] = cand_key[tid];
                    list_idx[tid] = cand_idx[tid];
                }
                if (tid == 0) n_list = C;
            } else {
                const unsigned ck = tid < C ? cand_key[tid] : 0u;
                int need;
                const unsigned kth = select_kth_per_thread<false>(ck, K, whist, need);
                const int n_eq = (int)(block_sum(ck == kth ? 1.f : 0.f, sh) + 0.5f);
                if (n_eq != need) {
                    fast = false;   // a tie straddles the K-th place: the general path keeps the smallest indices
                } else if (ck >= kth) {   // (padding keys are 0 < kth: every real key is >= 0x007fffff)
                    const int p = atomicAdd(&n_list, 1);
                    if (p < KCAP) {
                        list_key[p] = ck;
                        list_idx[p] = cand_idx[tid];
                    }
                }
            }
            __syncthreads();
        }
    }
#ifdef RV_SAMPLE_PROBE
    tp[2] = wall_clock64();
#endif
    if (!fast) {
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
#ifdef RV_SAMPLE_PROBE
        tp[2] = wall_clock64();
#endif
        // prefix = K-th largest key; `need` of the keys equal to it are wanted (the ones with the smallest indices)
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
        if (tid == 0) n_list = K;   // (the tie walk places its entries without counting them)
        __syncthreads();
    }
    if (tid < 64) {
        // rank sort of the n >= K listed candidates by (key desc, index asc); the first K are the result
        const int n = n_list < KCAP ? n_list : KCAP;
        const unsigned kk = tid < n ? list_key[tid] : 0u;
        const int ii = tid < n ? list_idx[tid] : 0x7fffffff;
        int rank = 0;
        for (int q = 0; q < n; ++q) {
            const unsigned kq = list_key[q];
            const int iq = list_idx[q];
            rank += (kq > kk) || (kq == kk && iq < ii);
        }
        if (tid < n && rank < K) {
            const unsigned u = (kk & 0x80000000u) ? (kk & 0x7fffffffu) : ~kk;
            topv[rank] = __uint_as_float(u);
            topi[rank] = ii;
        }
    }
    __syncthreads();
#ifdef RV_SAMPLE_PROBE
    tp[3] = wall_clock64();
#endif
    if (!do_sample) {
        if (tid == 0) {
            out_tok[b] = topi[0];
            out_hp[b] = h_raw;
            out_nkeep[b] = 0;
            if (out_thr) out_thr[b] = -INFINITY;
        }
        if (tid < KCAP && out_idx && out_val) {
            out_idx[b * KCAP + tid] = -1;
            out_val[b * KCAP + tid] = -INFINITY;
        }
        return;
    }
    // The transcendental work (exp, divide, log) runs one candidate per lane; the SUMS stay sequential in thread 0, in the
    // order HF's TopPLogitsWarper / multinomial accumulate (ascending index = descending probability).
    __shared__ float pr[KCAP], lg[KCAP];
    __shared__ float z2_sh;
    __shared__ int keep_sh;
    if (tid < K) e[tid] = expf(topv[tid] - topv[0]);
    __syncthreads();
    if (tid == 0) {
        float z = 0.f;
        for (int i = 0; i < K; ++i) z += e[i];
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
        z2_sh = z2;
        keep_sh = keep;
    }
    __syncthreads();
    const int keep = keep_sh;
    if (tid < keep) {
        const float p = e[tid] / z2_sh;
        pr[tid] = p;
        lg[tid] = logf(p + 1e-10f);
    }
    __syncthreads();
    if (tid == 0) {
        float hp = 0.f, cum = 0.f;
        const float u = uniforms ? uniforms[b] : 0.f;
        int pos = 0;
        for (int i = 0; i < keep; ++i) {
            const float p = pr[i];
            hp += p * lg[i];
            cum += p;
            if (cum <= u) pos = i + 1;
        }
        pos = pos < keep - 1 ? pos : keep - 1;
        out_tok[b] = topi[pos];
        out_hp[b] = -hp;
        out_nkeep[b] = keep;
        if (out_thr) out_thr[b] = topv[keep - 1];       // the smallest kept processed score
#ifdef RV_SAMPLE_PROBE
        tp[4] = wall_clock64();
        for (int q = 1; q < 5; ++q) out_val[b * KCAP + 59 + q] = (float)(tp[q] - tp[q - 1]);
        for (int q = 5; q < 9; ++q) out_val[b * KCAP + 50 + q] = (float)(tp[q] - tp[1]);
#endif
    }
    if (tid < KCAP) {
        out_idx[b * KCAP + tid] = tid < K ? topi[tid] : -1;
#ifdef RV_SAMPLE_PROBE
        if (tid < 55)
#endif
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
__device__ __forceinline__ float ld<op16_t>(const op16_t* p) { return op16_to_f32(*p); }

// one block (1024 threads) per segment: column norms over frames, sims[t] = <f_t / norm, q>, sum of the k largest
// (k <= 0: mean).  Phase 1 splits the frames over 4 thread groups per column block to keep ~T/4 loads per thread.
template <typename T>
__global__ __launch_bounds__(1024) void topk_cosine_kernel_generic(const T* __restrict__ feat, const float* __restrict__ q, int Tn, int d,
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


// The same with 16-byte loads (d a multiple of the vector width VEC = 16 B / sizeof(T)): thread t owns column chunk t % (d / VEC)
// and walks the frames of group t / (d / VEC), so a wave reads whole 1 KiB row segments; the per-group partial sums meet in LDS
// and are added in group order (deterministic).  The scores take one wave per frame with the same 16-byte loads, and the
// top-k is k wave-argmax rounds (largest value, then smallest frame index: the serial scan's choice).  39 MB of bf16 features
// for 100 segments: 157 -> ~25 us.
template <typename T>
__global__ __launch_bounds__(1024) void topk_cosine_kernel(const T* __restrict__ feat, const float* __restrict__ q, int Tn, int d,
                                                           int k, float* __restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(T);
    extern __shared__ float smem[];
    const int chunks = d / VEC, groups = 1024 / chunks;
    float* qn = smem;                 // [d]
    float* psum = smem + d;           // [groups][d]
    float* sims = psum + groups * d;  // [Tn]
    const T* f = feat + (int64_t)blockIdx.x * Tn * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto load = [&](const T* p, float (&v)[VEC]) {
        if constexpr (sizeof(T) == 2) {
            const op16x8 r = *(const op16x8*)p;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = op16_to_f32((op16_t)r[e]);
        } else {
            const f32x4 r = *(const f32x4*)p;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = r[e];
        }
    };
    if (tid < chunks * groups) {
        const int cc = tid % chunks, fg = tid / chunks;
        float ss[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) ss[e] = 0.f;
        for (int t = fg; t < Tn; t += groups) {
            float v[VEC];
            load(f + (int64_t)t * d + cc * VEC, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) ss[e] += v[e] * v[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) psum[fg * d + cc * VEC + e] = ss[e];
    }
    __syncthreads();
    for (int c = tid; c < d; c += 1024) {
        float s2 = 0.f;
        for (int g = 0; g < groups; ++g) s2 += psum[g * d + c];
        qn[c] = q[c] / sqrtf(s2);
    }
    __syncthreads();
    for (int t = wave; t < Tn; t += 16) {
        float s1 = 0.f;
        for (int cc = lane; cc < chunks; cc += 64) {
            float v[VEC];
            load(f + (int64_t)t * d + cc * VEC, v);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s1 += v[e] * qn[cc * VEC + e];
        }
        s1 = wave_sum(s1);
        if (lane == 0) sims[t] = s1;
    }
    __syncthreads();
    if (wave != 0) return;
    float acc = 0.f;
    if (k <= 0) {
        for (int t = lane; t < Tn; t += 64) acc += sims[t];
        acc = wave_sum(acc) / (float)Tn;
    } else {
        for (int r = 0; r < k && r < Tn; ++r) {
            ArgMax best{-INFINITY, 0x7fffffff};
            for (int t = lane; t < Tn; t += 64) best = better(best, ArgMax{sims[t], t});
            best = wave_argmax(best);
            acc += best.v;
            if (lane == 0 && best.i < Tn) sims[best.i] = -INFINITY;
            __builtin_amdgcn_s_waitcnt(0);   // the LDS write lands before the next round's reads (one wave: program order)
        }
    }
    if (lane == 0) out[blockIdx.x] = acc;
}

// _topk_pooling (similarity.py:71-94) for one (video, text) pair per block: sims[t] = <f_t, q>, the k frames with the largest
// sims (ties: smaller frame index), pooled[c] = sum over the selected frames of f[t][c], added in descending-sims order.
// 256 threads: one wave per frame for the scores, the whole block over the columns for the sum.
template <typename T>
__global__ __launch_bounds__(256) void topk_pool_kernel(const T* __restrict__ video, const float* __restrict__ text, int Tn, int d,
                                                        int Nt, int k, float* __restrict__ out, int32_t* __restrict__ out_idx) {
    extern __shared__ float smem[];
    float* qs = smem;          // [d]
    float* sims = smem + d;    // [Tn]
    __shared__ int sel[64];
    const int v = blockIdx.x, tx = blockIdx.y;
    const T* f = video + (int64_t)v * Tn * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < d; c += 256) qs[c] = text[(int64_t)tx * d + c];
    __syncthreads();
    for (int t = wave; t < Tn; t += 4) {
        float s = 0.f;
        for (int c = lane; c < d; c += 64) s += ld<T>(f + (int64_t)t * d + c) * qs[c];
        s = wave_sum(s);
        if (lane == 0) sims[t] = s;
    }
    __syncthreads();
    if (wave == 0) {
        for (int r = 0; r < k; ++r) {
            ArgMax best{-INFINITY, 0x7fffffff};
            for (int t = lane; t < Tn; t += 64) best = better(best, ArgMax{sims[t], t});
            best = wave_argmax(best);
            if (lane == 0) {
                sel[r] = best.i;
                sims[best.i] = -INFINITY;
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    __syncthreads();
    float* o = out + ((int64_t)v * Nt + tx) * d;
    for (int c = tid; c < d; c += 256) {
        float a = 0.f;
        for (int r = 0; r < k; ++r) a += ld<T>(f + (int64_t)sel[r] * d + c);
        o[c] = a;
    }
    if (out_idx && tid < k) out_idx[((int64_t)v * Nt + tx) * k + tid] = sel[tid];
}

}  // namespace

extern "C" int rv_sample(const rv_ctx* ctx, const float* logits, int32_t B, int32_t V, const float* uniforms, int32_t do_sample, float temperature,
                         int32_t top_k, float top_p, int32_t* out_tokens, float* out_entropy_proc, float* out_entropy_raw,
                         int32_t* out_topk_idx, float* out_topk_val, int32_t* out_nkeep, float* out_threshold, void* stream) {
    RvOptScope scope(rv_ctx_opts(ctx));
    RV_CHECK_ARG(logits && out_tokens && out_entropy_proc && out_entropy_raw && out_nkeep, "rv_sample: null output");
    RV_CHECK_ARG(B > 0 && V > 0 && V <= TPB * ITEMS, "rv_sample: V=%d exceeds %d", V, TPB * ITEMS);
    if (do_sample) {
        RV_CHECK_ARG(top_k >= 0, "rv_sample: top_k=%d must be 0 (no top-k filter) or positive when sampling", top_k);
        RV_CHECK_ARG(temperature > 0.f && top_p > 0.f, "rv_sample: temperature and top_p must be positive");
        RV_CHECK_ARG(out_topk_idx && out_topk_val, "rv_sample: candidate outputs required when sampling");
    }
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(TPB), 0, as_stream(stream), logits, V, uniforms, do_sample, temperature, top_k,
                       top_p, out_tokens, out_entropy_proc, out_entropy_raw, out_topk_idx, out_topk_val, out_nkeep, out_threshold, rv_cur_opts().sample_variant);
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
    RV_CHECK_ARG(feat_dtype == RV_OP16 || feat_dtype == RV_F32, "rv_topk_cosine: dtype must be f32 or bf16");
    const int vec = feat_dtype == RV_OP16 ? 8 : 4;
    const int chunks = d / vec, groups = chunks > 0 && chunks <= 1024 ? 1024 / chunks : 0;
    const size_t sm_fast = ((size_t)(groups + 1) * d + T) * sizeof(float);
    if (d % vec == 0 && groups > 0 && sm_fast <= 64 * 1024) {
        if (feat_dtype == RV_OP16)
            hipLaunchKernelGGL(topk_cosine_kernel<op16_t>, dim3(n), dim3(1024), sm_fast, as_stream(stream), (const op16_t*)feat, q_cls, T, d, k, out);
        else
            hipLaunchKernelGGL(topk_cosine_kernel<float>, dim3(n), dim3(1024), sm_fast, as_stream(stream), (const float*)feat, q_cls, T, d, k, out);
    } else {
        RV_CHECK_ARG((size_t)(5 * d + T) * 4 <= 64 * 1024, "rv_topk_cosine: 5*d + T too large for LDS");
        const size_t sm = (size_t)(5 * d + T) * sizeof(float);
        if (feat_dtype == RV_OP16)
            hipLaunchKernelGGL(topk_cosine_kernel_generic<op16_t>, dim3(n), dim3(1024), sm, as_stream(stream), (const op16_t*)feat, q_cls, T, d, k, out);
        else
            hipLaunchKernelGGL(topk_cosine_kernel_generic<float>, dim3(n), dim3(1024), sm, as_stream(stream), (const float*)feat, q_cls, T, d, k, out);
    }
    RV_CHECK_LAUNCH("rv_topk_cosine");
    return RV_OK;
}

extern "C" int rv_topk_pool(const void* video, int dtype, const float* text, int32_t Nv, int32_t T, int32_t d, int32_t Nt, int32_t k,
                            float* out, int32_t* out_idx, void* stream) {
    RV_CHECK_ARG(video && text && out && Nv > 0 && T > 0 && d > 0 && Nt > 0, "rv_topk_pool: bad arguments");
    RV_CHECK_ARG(dtype == RV_OP16 || dtype == RV_F32, "rv_topk_pool: dtype must be f32 or bf16");
    RV_CHECK_ARG(k >= 1 && k <= 64 && k <= T, "rv_topk_pool: k=%d must be in [1, min(64, T=%d)]", k, T);
    RV_CHECK_ARG(Nt <= 65535, "rv_topk_pool: at most 65535 texts per launch");
    const size_t sm = (size_t)(d + T) * sizeof(float);
    RV_CHECK_ARG(sm + 64 * sizeof(int) <= 64 * 1024, "rv_topk_pool: d + T too large for LDS (dynamic %zu B + 256 B static)", sm);
    if (dtype == RV_OP16)
        hipLaunchKernelGGL(topk_pool_kernel<op16_t>, dim3(Nv, Nt), dim3(256), sm, as_stream(stream), (const op16_t*)video, text, T, d, Nt, k, out, out_idx);
    else
        hipLaunchKernelGGL(topk_pool_kernel<float>, dim3(Nv, Nt), dim3(256), sm, as_stream(stream), (const float*)video, text, T, d, Nt, k, out, out_idx);
    RV_CHECK_LAUNCH("rv_topk_pool");
    return RV_OK;
}
