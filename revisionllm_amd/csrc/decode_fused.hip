// Decode step (S = 1, M <= 16 rows): attention + o projection in ONE launch.
//
// In a KV-cached decode step the attention of a layer is a pure latency chain (launch -> Q -> K / V^T tiles -> softmax ->
// merge -> store: ~9 us for 20 MB) during which HBM idles, and the o projection that follows is a 34 MB weight stream that
// cannot start before the attention output exists.  Here every workgroup FIRST issues the whole weight stream of its o
// projection tile (16 rows x D, 4 KiB per wave and 128-k block, straight into registers), THEN runs its attention unit
// (one (batch row, head) pair: the 8 waves split the keys 32 at a time and merge (m, l, O) through LDS), publishes the
// head's output, and waits until all units have arrived before it feeds the (already landed) weights to the MFMAs.  The
// attention's latency hides under the weight stream and one launch per layer disappears.
//
// Cross-workgroup hand-off: the attention output a16 is written with sc1 (agent-coherent, write-through) stores and read
// with sc1 loads; arrival is an agent-scope atomic counter polled by one lane.  All workgroups are co-resident (grid <=
// #CUs at 512 threads), spins are bounded, and a flag holds the host-side launch epoch, so nothing is reset.  Epilogue = the decode o projection's: + residual -> h, and the fused-RMSNorm
// producer outputs (bf16 w_next * h, per-workgroup sums of squares; kernels.h GemvNorm).
#include <atomic>

#include "kernels.h"

namespace {

std::atomic<int> g_decode_epoch{0};
constexpr int SC1 = 16;   // cache-policy bit 4 = sc1 on gfx940+
constexpr int DH = 128, NC = DH / 32, ND = DH / 16;
constexpr int MAXKB = 4;  // 128-k blocks per wave held in registers: D <= 8 waves x 4 x 128 = 4096

struct AttnOproj {
    const bf16_t* q16;   // [B, D] RoPE-rotated queries of this step
    const bf16_t* kc;    // this layer's K cache  [B, H, Smax, 128]
    const bf16_t* vtc;   // this layer's V^T cache [B, H, 128, Smax]
    bf16_t* a16;         // [B, D] attention output (scratch, sc1 traffic only)
    const bf16_t* wo;    // fragment-packed [D, D]
    float* h;            // [B, D] residual stream, updated in place
    GemvNorm nrm;        // producer side (xw_out, w_next, out_sumsq)
    int* sync;           // one flag word per workgroup: the epoch of the last launch it arrived in
    int* status;
    int epoch;           // host-side launch counter (> 0; the workspace starts zeroed)
    int B, H, Lk, Smax, D;
    float scale;
};

__global__ __launch_bounds__(512) void attn_oproj_decode(AttnOproj p) {
    __shared__ __attribute__((aligned(16))) float red[8 * 256];   // attention merge (m, l, O) / GEMV cross-wave reduction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4;
    const int D = p.D, nkb = D >> 7, tiles = D >> 4, units = p.B * p.H;
    const bool has_tile = (int)blockIdx.x < tiles;

    // ---- 1. the whole weight stream of this workgroup's o-projection tile.  A wave's loads return in order, so the stream
    // is issued right AFTER the loads of the wave's first attention key block (they are consumed first) and before any of
    // the attention arithmetic: the attention chain then runs while the weights are in flight.
    bf16x8 wf[MAXKB][4];
    f32x4 res_pre = f32x4{0.f, 0.f, 0.f, 0.f}, wn_pre = f32x4{0.f, 0.f, 0.f, 0.f};
    bool w_issued = false;
    auto issue_weights = [&]() {
        if (has_tile) {
            const bf16_t* wp = p.wo + (int64_t)blockIdx.x * (D >> 5) * 512 + lane * 8;
#pragma unroll
            for (int d = 0; d < MAXKB; ++d) {
                const int kb = wave + 8 * d;
                if (kb < nkb) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) wf[d][j] = __builtin_nontemporal_load((const bf16x8*)(wp + (kb * 4 + j) * 512));
                }
            }
        }
        if (has_tile && wave == 0 && fr < p.B) {   // epilogue operands of this tile: independent of the attention, fetched now
            res_pre = *(const f32x4*)(p.h + (int64_t)fr * D + blockIdx.x * 16 + g * 4);
            if (p.nrm.out_sumsq) wn_pre = *(const f32x4*)(p.nrm.w_next + blockIdx.x * 16 + g * 4);
        }
        w_issued = true;
    };

    // ---- 2. attention units (batch row, head): 8 waves split the keys, merge through LDS --------------------------------
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(p.a16, 0, p.B * D * 2, 0x00020000);
    float* sm_m = red;             // [8]
    float* sm_l = red + 8;         // [8]
    float* sm_o = red + 16;        // [8][128]
    for (int u = blockIdx.x; u < units; u += gridDim.x) {
        const int bb = u / p.H, hh = u - bb * p.H;
        const bf16_t* qp = p.q16 + (int64_t)bb * D + hh * DH + g * 8;
        bf16x8 qf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) qf[c] = *(const bf16x8*)(qp + c * 32);
        const bf16_t* kbase = p.kc + ((int64_t)bb * p.H + hh) * p.Smax * DH + g * 8;
        const bf16_t* vbase = p.vtc + (((int64_t)bb * p.H + hh) * DH + fr) * p.Smax + g * 4;
        f32x4 o[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float m_run = -INFINITY, l_run = 0.f;
        for (int k0 = wave * 32; k0 < p.Lk; k0 += 256) {
            bf16x8 kf[2][NC];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int key = min(k0 + t * 16 + fr, p.Lk - 1);
                const bf16_t* kp = kbase + (int64_t)key * DH;
#pragma unroll
                for (int c = 0; c < NC; ++c) kf[t][c] = *(const bf16x8*)(kp + c * 32);
            }
            union VF { bf16x8 v; u32x2 h2[2]; };
            VF vf[ND];
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                const bf16_t* vp = vbase + (int64_t)dt * 16 * p.Smax + k0;
                vf[dt].h2[0] = *(const u32x2*)(vp);
                vf[dt].h2[1] = *(const u32x2*)(vp + 16);
            }
            if (!w_issued) issue_weights();
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < NC; ++c) s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][c], qf[c], s[t], 0, 0, 0);
            }
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = k0 + t * 16 + g * 4 + r;
                    const float v = key >= p.Lk ? -INFINITY : s[t][r] * p.scale;
                    s[t][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __expf(m_run - m_use);
            float psum = 0.f, pe[8];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __expf(s[t][r] - m_use);
                    pe[t * 4 + r] = e;
                    psum += e;
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            union { bf16x8 v; uint32_t u[4]; } pf;
#pragma unroll
            for (int i = 0; i < 4; ++i) pf.u[i] = pack_bf16x2(pe[2 * i], pe[2 * i + 1]);
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) {
                o[dt] *= alpha;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt].v, pf.v, o[dt], 0, 0, 0);
            }
        }
        if (!w_issued) issue_weights();   // a wave without a key block of its own
        l_run += __shfl_xor(l_run, 16, 64);
        l_run += __shfl_xor(l_run, 32, 64);
        // the 16 MFMA "query rows" are copies of the one real query: row fr == 0 carries the result
        if (lane == 0) {
            sm_m[wave] = m_run;
            sm_l[wave] = l_run;
        }
        if (fr == 0) {
#pragma unroll
            for (int dt = 0; dt < ND; ++dt) *(f32x4*)&sm_o[wave * DH + dt * 16 + g * 4] = o[dt];
        }
        __syncthreads();
        if (tid < 64) {   // thread t -> output dims 2t, 2t + 1
            float mm = -INFINITY;
#pragma unroll
            for (int w = 0; w < 8; ++w) mm = fmaxf(mm, sm_m[w]);
            float lt = 0.f, v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                const float sc = sm_m[w] == -INFINITY ? 0.f : __expf(sm_m[w] - mm);
                lt += sm_l[w] * sc;
                v0 += sm_o[w * DH + 2 * tid] * sc;
                v1 += sm_o[w * DH + 2 * tid + 1] * sc;
            }
            const float inv = 1.0f / lt;
            __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(v0 * inv, v1 * inv), a_rs, (int)((((int64_t)bb * D + hh * DH + 2 * tid)) * 2), 0, SC1);
        }
        __syncthreads();   // LDS is reused by the next unit / the GEMV reduction
    }
    if (!w_issued) issue_weights();                     // a workgroup without an attention unit
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this workgroup's a16 stores are acknowledged (and its weights have landed)
    __syncthreads();

    // ---- 3. every unit of the launch has arrived: one flag word per workgroup holding the launch epoch (256 same-address
    // agent-scope atomics would serialise for ~10 us; plain sc1 flag stores do not), polled by the first wave ------------
    if (tid == 0) __hip_atomic_store(p.sync + blockIdx.x, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
            for (int i = lane; i < (int)gridDim.x; i += 64)
                ok = ok && __hip_atomic_load(p.sync + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == p.epoch;
            if (__all(ok)) break;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) {
                if (lane == 0) *p.status = 2;
                break;
            }
        }
    }
    __syncthreads();

    // ---- 4. o projection of this workgroup's 16 output rows: out[b, n] = sum_k Wo[n, k] a[b, k] ---------------------------
    if (has_tile) {
        const int M = p.B;
        const int xr = fr < M ? fr : M - 1;
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < MAXKB; ++d) {
            const int kb = wave + 8 * d;
            if (kb < nkb) {
                bf16x8 xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xf[j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(a_rs, (int)(((int64_t)xr * D + kb * 128 + j * 32 + g * 8) * 2), 0, SC1));
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d][j], xf[j], acc, 0, 0, 0);
            }
        }
        *(f32x4*)(red + (wave * 64 + lane) * 4) = acc;
    }
    __syncthreads();
    if (has_tile && wave == 0) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 8; ++w) s += *(const f32x4*)(red + (w * 64 + lane) * 4);
        const int b = fr, n = blockIdx.x * 16 + g * 4;   // lane owns batch row b, output columns n .. n + 3
        float sq = 0.f;
        if (b < p.B) {
            float* hp = p.h + (int64_t)b * D + n;
            const f32x4 v = s + res_pre;
            *(f32x4*)hp = v;
            if (p.nrm.out_sumsq) {
                const f32x4 wn = wn_pre;
                *(u32x2*)((bf16_t*)p.nrm.xw_out + (int64_t)b * D + n) =
                    u32x2{pack_bf16x2(v[0] * wn[0], v[1] * wn[1]), pack_bf16x2(v[2] * wn[2], v[3] * wn[3])};
                sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
            }
        }
        if (p.nrm.out_sumsq) {
            sq += __shfl_xor(sq, 16, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (g == 0) p.nrm.out_sumsq[blockIdx.x * 16 + b] = b < p.B ? sq : 0.f;
        }
    }
}

}  // namespace

bool attn_oproj_decode_supported(int B, int H, int dh, int64_t D) {
    return dh == DH && D % 128 == 0 && D <= 8 * MAXKB * 128 && B >= 1 && B <= 16 && (int64_t)B * H <= 4096;
}

int attn_oproj_decode_launch(const void* q16, const void* kc, const void* vtc, void* a16, const void* wo, float* h, const GemvNorm& nrm,
                             int* sync, int* status, int B, int H, int Lk, int Smax, int64_t D, float scale, hipStream_t st) {
    RV_CHECK_ARG(q16 && kc && vtc && a16 && wo && h && sync && status, "attn_oproj_decode: null argument");
    RV_CHECK_ARG(attn_oproj_decode_supported(B, H, DH, D) && Lk >= 1 && Lk <= Smax && Smax % 32 == 0, "attn_oproj_decode: bad geometry");
    int cus = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const int tiles = (int)(D >> 4), units = B * H;
    const int grid = tiles > (units < cus ? units : cus) ? tiles : (units < cus ? units : cus);   // all resident: <= one per CU
    RV_CHECK_ARG(tiles <= cus, "attn_oproj_decode: %d tiles do not fit %d CUs", tiles, cus);
    AttnOproj p;
    p.q16 = (const bf16_t*)q16; p.kc = (const bf16_t*)kc; p.vtc = (const bf16_t*)vtc; p.a16 = (bf16_t*)a16; p.wo = (const bf16_t*)wo;
    int epoch = ++g_decode_epoch;
    if (epoch <= 0) { g_decode_epoch = 1; epoch = 1; }
    p.epoch = epoch;
    p.h = h; p.nrm = nrm; p.sync = sync; p.status = status; p.B = B; p.H = H; p.Lk = Lk; p.Smax = Smax; p.D = (int)D; p.scale = scale;
    hipLaunchKernelGGL(attn_oproj_decode, dim3((unsigned)grid), dim3(512), 0, st, p);
    RV_CHECK_LAUNCH("attn_oproj_decode");
    return RV_OK;
}
