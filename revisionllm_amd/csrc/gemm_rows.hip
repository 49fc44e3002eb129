// Decode projections with 33 .. 144 rows: several pools' worth of generates merged into ONE pass over the weights.
//
// The <= 32-row kernel (gemv_stream, gemm.hip) re-reads the activations once per workgroup from L2; with MB = 4 / 8 row blocks
// that is 4 - 8x the weight bytes.  Here the 4 consumer waves of a workgroup own different 16-column tiles (64 columns per
// workgroup) and SHARE the activations: the 128-k slab of all rows (4 * MB KiB, contiguous in the fragment-packed layout,
// kernels.h) is staged into LDS by a producer wave (LDS-DMA), while each consumer streams its own weight fragments into registers
// (weights: HBM -> registers once; activations: L2 -> LDS once per workgroup -> registers once per wave).
//
// Summation order.  gemv_stream deals the k-blocks to 8 virtual waves (kb % 8, ascending kb) and adds the 8 partial sums as the
// balanced tree gemv_tree8.  A workgroup here walks VPW of the virtual waves one after the other and folds them as adjacent
// subtrees of that tree; with S = 8 / VPW workgroups per column group (split-K wherever N / 64 groups alone would not fill the
// CUs) each leaves one partial plane, and the LAST of them to arrive (a monotonic arrival counter per column group and split
// count; nobody waits) adds the planes of the group's (column block, row block) pairs - all loads in flight at once - and runs the
// shared epilogue (gemv_finish.h).  A row's result is therefore bit-identical to what the <= 32-row kernel produces for it, whatever it is batched with
// (tests/test_gpu_merged_decode.py).
//
// Measured (MI355X, 63 rows, isolated): qkv 46 us, o / down 22 us, gate/up 68 us per launch - a fixed ~8 us (launch, first loads,
// finish) plus the weights at 3.2 (K = 4096) .. 4.6 TB/s (long K).  Compile-time probes: without the slab DMA and the barriers the
// consumers' loop alone takes 55 of the gate/up launch's 58 us, the slab pipeline alone 18 us - what bounds it is the weight bytes in
// flight (8 consumer waves x 5 stages x 4 KiB = 160 KiB per CU, all the registers the accumulator stack leaves; gemv_stream at <= 16
// rows keeps 256 KiB in flight and streams at 5.8 TB/s); a 63-row step 6.5 ms vs 4.1 ms for 28 rows on gemv_stream, i.e. 0.72 vs 1.02 ms
// per 7-row generate (112 rows: 8.7 ms, 0.54 ms).
//
// Round 3, measured and NOT kept: (a) FP8 weights (WP = 2, kept as the opt-in fp8 path) halve the bytes of a launch and leave its time
// where it was (70 rows, isolated step: gate/up 67.4 -> 62.7 us, qkv 51.8 -> 49.8, o / down 23.2 -> 24.4): the launch is not bound by
// weight bytes at these row counts.  (b) A PERSISTENT grid (2 workgroups per CU walking their (column group, split) items as one
// stream of stages, weight and slab rings running on across item boundaries, the item's hand-over / finish inside the stream): same
// results, and the same time as one workgroup per item (70-row step 7.23 vs 7.30 ms; both behind this kernel's 6.84 ms because the
// finish must then stay small enough to keep the ring's 64 VGPRs live: pair-by-pair plane loads) - the cold start and tail of a
// 16-stage workgroup are not what bounds the launch either.
// (c) What the RS_PROBE builds say (tools/rows_probe.sh, 70 rows, back-to-back launches, gate/up shape N = 22016, K = 4096: 53.1 us):
// every weight load L1 / L2-hot 37.1 us; no per-stage barriers 54.1; no MFMAs 53.7; no slabs staged 52.2; hot weights + no barriers + no slabs
// 34.0.  And the K sweep (tools/rows_time.py): launch time = FIXED + weights at 5.4 TB/s marginal, with FIXED = 19.9 us (gate/up: 688
// workgroups = two rounds of the 512 resident slots), 14.4 us (QKV: 384 workgroups), 7 - 9 us (o / down: 256).  So: the streaming itself
// runs at what HBM gives; a ROUND of workgroups costs ~10 us whatever it streams (first weights cold, planes out + acknowledged, the
// arrival counter's round trip, the last arriver's plane loads and epilogue), and gate/up pays it twice because an item (column group,
// split) must be whole virtual k-waves of the shared summation tree and 688 items do not deal out over 512 slots.  What would remove the second
// round: a persistent grid whose hand-overs are DEFERRED (planes stored without waiting; one acknowledgement, the counters and the finishes
// once at the end of the workgroup's life).
// (d) BUILT as rows_kernel_p (below; option rows_persistent, on by default): a launch with more (column group, split) items than resident
// workgroups runs as a grid of exactly the resident workgroups, item i + k * grid going to workgroup i; rows_splits picks a finer split
// when it shortens the longest stream by >= 20 %.  Measured (isolated steps, us per launch inside the step): 112 rows (one workgroup per
// CU: 688 / 384 items on 256 slots) gate/up 91.0 -> 67.0, QKV 81.4 -> 70.6, step 9.82 -> 8.62 ms; 70 rows (two per CU) gate/up with
// S = 4: 67.6 -> 65.1, step 6.92 -> 6.81 ms - not the 15 us the model above promised: 1376 planes of 20 KiB out and in again are
// 56 MB of extra traffic next to 180 MB of weights (PMC: 241 MB per launch), which eats most of the second round's fixed cost.
// 129 .. 144 rows (MB = 9: twenty 7-row generates in one step): 10.07 ms per step = 0.50 ms per generate (70 rows: 0.68, 112: 0.54).
// (e) rows_single (8 / 9 row blocks): the fused QKV projection (192 column groups) runs UNSPLIT on 192 of the 256 CUs - 84 -> 64 us per launch at
// 140 rows, step 10.06 -> 9.47 ms: fewer, longer workgroups without planes win.  (Round 3 put that down to LDS reads - "4 x 36 KiB per stage =
// 1150 LDS cycles" - which counted ds_read_b128 at 128 B/clk; it moves 256: a stage's reads are 576 cycles, as long as its 36 MFMAs.)
// (f) Round 5, the probes again at 140 rows (plain epilogues, back-to-back launches, us per launch for N = 12288 / 22016 at K = 4096; regular
// 32.5 / 58.5): weights L1-hot 23.8 / 43.6; no barriers 33.3 / 63.7; no MFMAs 32.1 / 57.7; no slabs 30.0 / 55.8; hot + no barriers + no slabs
// 21.1 / 37.9, ... + no MFMAs 14.0 / 27.1, ... + no LDS reads 19.1 / 36.2, all of them off 11.0 / 26.2.  So a stage of the bare loop costs ~460
// cycles, its MFMAs add ~600 (they do NOT hide under the loop's own issue stream with one wave per SIMD) and its LDS reads ~100; with real
// weights the stage takes 0.9 us whatever the ring depth (RS_DW_BIG = 5 / 6 / 8: N = 4096 shapes unchanged to the 0.1 us, the others slower by
// their pad slots and spills) and whatever order the tiles' k-blocks are walked in (RS_PROBE & 8192: 58.5 -> 55.2, 32.9 -> 33.0): 16 KiB per CU
// per 0.9 us = 4.4 TB/s marginal next to a fixed 8 (N = 4096) .. 17 us (gate/up) per launch.  What a launch at 9 row blocks pays for is the
// memory system's rate at one workgroup per CU and its fixed cost, not the stage's arithmetic.
// (g) ... which the cycle stamps of one consumer wave then corrected (RS_PROBE & 16384, tools/rows_stamps.py; N = 12288, K = 4096, 140 rows, COLD weights, workgroup 8):
// the wave never waits - neither for its weights (the ring's three stages cover the HBM latency) nor at the barrier (the s_memtime pair around each reads
// its own ~112 cycles).  A stage is ~1300 cycles of the wave's OWN instruction stream: the 36 x (read, wait, MFMA) triples take 1010 .. 1170 (27 - 30 cycles
// per MFMA; the MFMAs alone, RS_PROBE & 4096: 984; the reads alone: 552 = 256 B/clk for the CU's four waves), the loop / fold behind them ~200 (fold stages
// + 500).  The compiler's read-2 / MFMA-2 order cost 1484: that was the 400 cycles the prescribed order removed.  Issuing a stage's first eight reads BEFORE
// the next weight loads are set up (RS_EARLY_READS = 1, with or without pinning those loads to the head of the block): 1080 / 1088 vs 1168 under the stamps,
// 8.15 - 8.26 vs 8.15 - 8.17 ms per step without - level, not kept as the default.  So the 9-row-block launches are MFMA-ISSUE-bound inside a stage (16 cycles per
// MFMA would be the pipe's rate; this loop reaches 27, the prefill GEMM's main loop 23) and pay ~12 us per launch outside the stages (launch gap, cold first
// weights, the exchange and the finish): 37 % of a QKV launch.  Eight row blocks (112 rows): 1012 cycles for 32 MFMAs.
#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "kernels.h"
#include "gemv_finish.h"

// Timing probes (tools/rows_probe.sh builds one library per value; results are garbage, only the time means something):
//   RS_PROBE & 1   every weight load re-reads the same 8 KiB (L1 / L2 hot): no HBM latency or bandwidth in the consumers' loop
//   RS_PROBE & 2   no per-stage barrier: consumers and producer run free (slabs may be stale)
//   RS_PROBE & 4   no MFMAs (the accumulators get one add per stage so that the loads stay live)
//   RS_PROBE & 8   the producer stages no slabs at all
//   RS_PROBE & 16384 cycle stamps (s_memtime) of consumer wave 0 of workgroup 8 at four points of every stage of an UNSPLIT launch - before the weight
//                    wait, behind it, behind the barrier, behind the last MFMA - left in the first 2 KiB of the planes workspace (tools/rows_stamps.py)
//   RS_PROBE & 8192  every wave walks its tile's k-blocks from its own rotated start (sums in another order: is the weight stream camping on channels?)
//   RS_PROBE & 4096  one activation fragment per stage instead of 4 * MB (the compiler hoists the read: no LDS traffic in the loop)
#ifndef RS_PROBE
#define RS_PROBE 0
#endif
#ifndef RS_PART
#define RS_PART 0
#endif
#ifndef RS_DW_BIG
#define RS_DW_BIG 4            // weight ring depth at 8 / 9 row blocks (probe builds: tools/rows_ab.sh)
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }

// The partial planes move between workgroups (any XCD) with sc1 = agent-coherent accesses (write-through stores, L2-missing loads)
// and a relaxed agent-scope counter - NOT with release / acquire fences: their buffer_wbl2 / buffer_inv sweep the XCD's whole L2 once
// per workgroup and serialise the launch (measured here: 30 -> 200 us per projection; same finding as gemm_pp.hip).
typedef unsigned int rs_u32x4 __attribute__((ext_vector_type(4)));
constexpr int RS_SC1 = 16;     // cache-policy bit 4 = sc1 on gfx940+
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rs_u32x4, v), r, (int)byte_off, 0, RS_SC1);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, RS_SC1));
}

constexpr int RS_W = 4;        // consumer waves = 16-column tiles per workgroup (64-column groups); wave RS_W is the producer
constexpr int RS_THREADS = (RS_W + 1) * 64;
// Weight ring 4 deep (3 x 4 KiB per consumer wave in flight) everywhere.  Deeper rings were tried because the consumers' loop is bound by
// the weight bytes in flight - and lost: at 6 deep the accumulator stack of the tree fold no longer fits the 168 VGPRs of three waves per
// SIMD and the spills (28 - 196 B of scratch per lane) cost more than the depth gains: 56-row step 6.44 -> 6.08 ms, 70-row step 7.86 -> 6.86 ms
// going from 6 to 4; 8 deep with 8 row blocks: gate/up 80 -> 107 us.  With 4 / 5 row blocks neither the LDS reads nor the MFMAs bound the loop either
// (compile-time probes at 70 rows: a loop-invariant activation fragment instead of the 20 LDS reads per stage - same time; one MFMA per
// weight fragment instead of five - 5-8 % faster): the eight consumer waves of a CU simply stream their weights at 3.4 (K = 4096) .. 4.6
// TB/s (long K) where the sixteen waves of gemv_stream reach 5.8.
// MB = 4 row blocks: 4 slabs of 16 KiB, two workgroups per CU;  MB = 5 (65 .. 80 rows: ten 7-row generates): 3 slabs of 20 KiB, two per CU;
// MB = 8 / 9 (81 .. 128 / 129 .. 144 rows: twenty 7-row generates): 3 slabs of 32 / 36 KiB, one workgroup per CU
// WP = 2: FP8 (e4m3fn) weights in gemv_stream's fp8 fragment packing (gemm.hip: a lane's 16-byte load = its operand of two consecutive
// 32-k blocks), widened to bf16 in registers right before the MFMA, per-output-row scales applied in the shared epilogue
// (GemvNorm::w_scale).  A 128-k stage is then 2 loads of 1 KiB per consumer wave instead of 4: the ring is twice as many STAGES deep for
// the same registers (and the same bytes in flight).
template <int MB, int WP = 1> struct RowsCfg {
    static constexpr int DW = WP == 2 ? 8 : (MB >= 8 ? RS_DW_BIG : 4), LPS = WP == 2 ? 2 : 4, DX = MB == 4 ? 4 : 3, WPE = MB <= 5 ? 3 : 2;
};
typedef unsigned int rs_w8 __attribute__((ext_vector_type(4)));

// Round 4: the rows' sums of squares of a consumed fused RMSNorm.  Every workgroup needs all <= 144 rows' sums: MB x 256 per-block
// partials of 16 rows each (36.8k floats at 9 row blocks).  Rounds 2 - 3 loaded them as 144 conditional four-byte loads per thread
// (a branch and a 64-bit address computation each: ~1700 instructions per wave) in the TAIL of the workgroup - 8 us of every launch
// that consumes a norm at 140 rows (probe without them: fused QKV 57.5 -> 49.2 us, gate/up 73.0 -> 65.9 us).  Now:
//   * a thread owns CHAINS of (row block, q, quad of rows): its 8 partials q, q + 32, ... as 16-byte loads (four rows at once), added in
//     the order gemv_stream's (q, row) threads add them (the finish then adds q = 0 .. 31): the rows stay bit-identical;
//   * 8 / 9 row blocks (one workgroup per CU, 256 registers): the consumer waves do it at the HEAD of the workgroup, right behind the
//     opening loads of the weight ring - the round trip hides under the first (cold, HBM) weights - into an LDS area of their own
//     behind the slab ring; 4 / 5 row blocks (168 registers: the head form spilled): in the tail as before, into the freed ring;
//   * every workgroup of an XCD starts at a different chain: all of them walking the same lines in the same order at the same moment
//     queued up on one L2 channel after the other (2 - 3 us of the 8).
constexpr bool rows_sumsq_at_head(int MB) { return MB >= 8; }
template <int MB, int NTH, int PTB>       // NTH threads take part (tg = index among them); PTB chains per thread in flight at once
__device__ __forceinline__ void rows_sumsq(const GemvNorm& nrm, float* ssq, int tg) {
    if (!nrm.in_sumsq || (RS_PROBE & 16)) return;
    constexpr int CHAINS = MB * 128;                       // (mb, q = 0 .. 31, row quad = 0 .. 3)
    constexpr int PT = (CHAINS + NTH - 1) / NTH;
    const int nblk = nrm.in_nblk;
    const int off = (RS_PROBE & 2048) ? 0 : (int)((blockIdx.x >> 3) * 37u) % CHAINS;     // (workgroup b lands on XCD b % 8)
#pragma unroll
    for (int i0 = 0; i0 < PT; i0 += PTB) {
        f32x4 pj[PTB][8];
        int cc[PTB];
#pragma unroll
        for (int i = 0; i < PTB; ++i) {
            const int idx = tg + (i0 + i) * NTH;
            int c = idx + off;
            c = c >= CHAINS ? c - CHAINS : c;
            cc[i] = (i0 + i < PT && idx < CHAINS) ? c : -1;
            const int mb = c >> 7, q = (c >> 2) & 31, rq = c & 3;
            const float* src = nrm.in_sumsq + ((int64_t)(mb * nblk + q) * 16 + rq * 4);
            // unconditional loads (a chain past the end re-reads a valid chain, a block past nblk re-reads block q: both are dropped by a
            // select) - a branch per load was what made the old form slow
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool have = q + 32 * j < nblk;
                const f32x4 v = *(const f32x4*)(src + (have ? j * 512 : 0));
                pj[i][j] = have ? v : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __builtin_amdgcn_sched_barrier(0);                 // (a batch's loads in flight at once: one round trip)
#pragma unroll
        for (int i = 0; i < PTB; ++i) {
            if (cc[i] < 0) continue;
            const int c = cc[i], mb = c >> 7, q = (c >> 2) & 31, rq = c & 3;
            f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) a += pj[i][j];
            for (int b = q + 256; b < nblk; b += 32) a += *(const f32x4*)(nrm.in_sumsq + ((int64_t)(mb * nblk + b) * 16 + rq * 4));
            *(f32x4*)(ssq + mb * 512 + q * 16 + rq * 4) = a;        // [mb][q][row]
        }
    }
}

// ... then ONE thread per row adds the 32 per-q sums (q = 0 .. 31, the order of gemv_stream's finish) into totl[MB * 16]: the finish of every
// (tile, row block) pair reads one float instead of adding 32 LDS values again (36 pairs x 64 lanes did, with one wave per SIMD to hide nothing).
template <int MB>
__device__ __forceinline__ void rows_sumsq_total(const float* ssq, float* totl, int t) {
    if (t < MB * 16) {
        const int mb = t >> 4, row = t & 15;
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) tot += ssq[(mb * 32 + q) * 16 + row];
        totl[t] = tot;
    }
}

// The fused q/k/v finish of a MERGED decode step (rows at their own positions: qr.row_pos), written for this kernel: a (tile, row block)
// pair's section / head / column offset are wave-uniform (no per-lane integer division), every pair's position, RoPE coefficients and FP8
// scales are loaded BEFORE the first store (one round trip for the wave's <= 9 pairs), and a pair then costs a few dozen instructions - the
// generic path (gemv_finish -> qkv_rope_store) spent ~390 per pair, and with one wave per SIMD that WAS the tail of the launch (round 4,
// DESIGN section 8).  Same arithmetic, same stores: bit-identical caches and Q.
template <int MB, int PPW, int PAIRS, int WP>
__device__ __forceinline__ void rows_qkv_finish(const f32x4 (&sres)[PPW], int wave, int cg, int fr, int kg, int M, const GemvNorm& nrm, const QkvRope& qr,
                                                const float* totl) {
    const int D = qr.H * 128;
    const int n0 = cg * 64;                               // the workgroup's 64 columns lie inside ONE head of ONE section
    const int sec = n0 >= 2 * D ? 2 : n0 >= D ? 1 : 0;
    const int hd0 = n0 - sec * D, head = hd0 >> 7;
    int posv[PPW];
    f32x4 cf[PPW], wsc[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + RS_W * i < PAIRS ? wave + RS_W * i : PAIRS - 1;
        const int tile = p / MB, mb = p % MB, b = mb * 16 + fr;
        const int pc = (hd0 & 127) + tile * 16 + kg * 4;  // column of the head (pair-interleaved order for q / k)
        const int bc = b < M ? b : M - 1;
        posv[i] = b < M ? qr.row_pos[bc] : -1;
        cf[i] = sec < 2 ? *(const f32x4*)(qr.cs + ((int64_t)bc * 64 + (pc >> 1)) * 2) : f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (WP == 2) wsc[i] = *(const f32x4*)(nrm.w_scale + n0 + tile * 16 + kg * 4);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + RS_W * i;
        if (p >= PAIRS) break;
        const int tile = p / MB, mb = p % MB, b = mb * 16 + fr;
        const int pc = (hd0 & 127) + tile * 16 + kg * 4;
        f32x4 v = sres[i];
        if constexpr (WP == 2) v *= wsc[i];
        if (nrm.in_sumsq) v *= rsqrtf(__fmaf_rn(totl[mb * 16 + fr], nrm.inv_d, nrm.eps));
        const int pos = posv[i];
        if (pos < 0 || pos >= qr.Smax) continue;          // inactive row, or past the pool's capacity: nothing is stored
        if (sec < 2) {
            const f32x4 t = cf[i];
            const float a0 = __fmaf_rn(v[0], t[0], -__fmul_rn(v[1], t[1])), b0 = __fmaf_rn(v[1], t[0], __fmul_rn(v[0], t[1]));
            const float a1 = __fmaf_rn(v[2], t[2], -__fmul_rn(v[3], t[3])), b1 = __fmaf_rn(v[3], t[2], __fmul_rn(v[2], t[3]));
            const u32x2 o = pack_op16x4(f32x4{a0, b0, a1, b1});
            if (sec == 0) *(u32x2*)((op16_t*)qr.q16 + (int64_t)b * D + hd0 + tile * 16 + kg * 4) = o;
            else *(u32x2*)((op16_t*)qr.kc + (((int64_t)b * qr.H + head) * qr.Smax + pos) * 128 + pc) = o;
        } else {
            op16_t* dst = (op16_t*)qr.vtc + ((int64_t)b * qr.H + head) * 128 * qr.Smax + rv_vt_index(pc, pos);
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[r * 8] = f32_to_op16(v[r]);
        }
    }
}

// Round 5.  With 8 / 9 row blocks a CU holds ONE workgroup: one consumer wave per SIMD, nothing to hide a wave's LDS latency but its own
// instruction stream - and the compiler's schedule of a stage was "two fragment reads, wait, two MFMAs" eighteen times over (it sinks the 36
// reads to their uses to save registers): every pair paid a full LDS round trip, a stage took the SUM of its LDS time (4 waves x 36 KiB = 1152
// cycles) and its MFMA time (36 x 32 cycles) instead of the larger of the two.  The stage's order is now prescribed: RS_PF fragment reads ahead of
// the MFMA that consumes the oldest (a rotating window of RS_PF x 4 registers; the compiler still inserts the exact lgkmcnt waits).
// Measured (same box, isolated 140-row steps, alternating libraries): 8.26 / 8.27 -> 8.04 / 8.10 ms and 8.42 / 8.36 -> 8.16 / 8.28 ms per step
// (112 rows: 7.39 / 7.34 -> 7.03 / 7.16) - 2 %, not the 2 x the cycle count promised: ds_read_b128 moves 256 B/clk (a stage's 144 KiB = 576
// cycles, as long as its 36 MFMAs of 16 cycles), so a stage was never LDS- or MFMA-bound; what the launch waits for is its weight stream.
// And a DEEPER weight ring at 9 row blocks (RS_DW_BIG = 6 / 8 with this window or a 4-fragment one: the registers are there at one workgroup
// per CU) is slower again: 8.68 / 8.63 / 9.40 ms per step (the persistent form pads an item's 16 stages to the ring depth; 8 deep spills).
#ifndef RS_PF_N
#define RS_PF_N 8
#endif
constexpr int RS_PF = RS_PF_N;
#ifndef RS_EARLY_READS
#define RS_EARLY_READS 0      // 1: a stage's first RS_PF fragment reads are issued right behind the barrier, BEFORE the next stage's weight loads are set up
#endif
constexpr bool rows_early(int MB) { return RS_EARLY_READS && MB >= 8 && !(RS_PROBE & (4 | 4096)); }
template <int MB, int WP>
__device__ __forceinline__ void rows_stage_schedule() {
#ifndef RS_NO_STAGE_SCHEDULE
    if constexpr (MB >= 8 && !(RS_PROBE & 4)) {
        constexpr int R = 4 * MB;                                         // fragment reads = MFMAs of a stage
        if constexpr (!rows_early(MB)) __builtin_amdgcn_sched_group_barrier(0x100, RS_PF, 0);            // DS reads (early form: already issued)
#ifdef RS_EARLY_PIN_VMEM
        else __builtin_amdgcn_sched_group_barrier(0x020, WP == 2 ? 2 : 4, 0);                             // the next stage's weight loads stay at the head of the block
#endif
#pragma unroll
        for (int i = 0; i < R - RS_PF; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA ...
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);            // ... one read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, RS_PF, 0);
    }
#endif
}

// FIN: 0 = f32 out, one tile per block (o / down projections: residual, next-norm prescale + sums of squares)
//      1 = bf16 SILU(gate) * up, tile pairs (gate/up)        2 = f32 out, tile pairs (N >= 16384: lm_head)
//      3 = fused q/k/v + RoPE epilogue                        4 = bf16 out, one tile per block
//
// Roles.  The consumer waves stream ONLY their weight fragments (registers, DW stages deep, vmcnt counts nothing else); the producer
// wave copies the activation slabs into LDS (LDS-DMA, DX slabs deep, its own vmcnt).  vmcnt retires in order, so a wave that issued
// both could not wait for a young slab without also waiting for every older weight load - the weight ring would be no deeper than
// the slab ring.  One raw s_barrier per stage hands slab g to the consumers and the slot of slab g - 1 back to the producer.
template <int MB, int VPW, int FIN, int WP = 1>
__global__ __attribute__((amdgpu_flat_work_group_size(RS_THREADS, RS_THREADS), amdgpu_waves_per_eu(RowsCfg<MB>::WPE, RowsCfg<MB>::WPE))) void
rows_kernel(const op16_t* __restrict__ X, const op16_t* __restrict__ W, float* __restrict__ planes, const float* __restrict__ bias,
            const float* res, int64_t ldr, void* Cv, int64_t ldc, int M, int N, int K, GemvNorm nrm, QkvRope qr) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    constexpr int S = 8 / VPW, LOG = VPW == 8 ? 3 : VPW == 4 ? 2 : VPW == 2 ? 1 : 0;
    constexpr int DW = RowsCfg<MB, WP>::DW, DX = RowsCfg<MB, WP>::DX, LPS = RowsCfg<MB, WP>::LPS;
    constexpr int SLAB = MB * 4096;                   // bytes of one 128-k slab of MB row blocks
    constexpr int XL = MB * 4;                        // LDS-DMA fragments (1 KiB) per slab
    static_assert((DX - 2) * XL <= 63 && (DW - 2) * LPS <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = blockIdx.x / S, sp = blockIdx.x % S;
    const int nkb = K >> 7;
    const int ntile = cg * RS_W + (wave < RS_W ? wave : 0);   // this wave's 16-column tile
    // stage sequence: virtual waves v = sp * VPW + i, i = 0 .. VPW - 1, one after the other; virtual wave v owns k-blocks v, v + 8, ...
    // (cnt(v) of them).  T stages in all; both roles run exactly T + (stages issued past the end as L2-hot dummies) of them.
    auto vcount = [&](int i) -> int { return (nkb - (sp * VPW + i) + 7) >> 3; };
    int T = 0;
#pragma unroll
    for (int i = 0; i < VPW; ++i) T += vcount(i);

    f32x4 acc[MB], stk[LOG ? LOG : 1][MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave == RS_W) {
        // ---------------- producer: slab of stage g -> LDS slot g % DX ----------------
        int pi = 0, pc = 0;     // (virtual wave index, k-block index in it) of the next slab to issue
        auto next_kb = [&]() -> int {
            int kb = -1;
            if (pi < VPW) {
                kb = sp * VPW + pi + 8 * pc;
                if (++pc == vcount(pi)) { pc = 0; ++pi; }
            }
            return kb;
        };
        auto issue = [&](int g) {
            const int kb = next_kb();
            const op16_t* xs = X + (kb >= 0 ? (int64_t)kb * (MB * 2048) : 0) + lane * 8;
            char* dst = rs_smem + (g % DX) * SLAB;
#pragma unroll
            for (int i = 0; i < XL; ++i) glds16(xs + i * 512, dst + i * 1024);
        };
        if constexpr (!(RS_PROBE & 8))
            for (int d = 0; d < DX - 1; ++d) issue(d);
        if constexpr (rows_sumsq_at_head(MB)) {
            if (nrm.in_sumsq && !(RS_PROBE & 16)) __builtin_amdgcn_s_barrier();   // the consumers' head barrier (rows_sumsq)
        }
        for (int g = 0; g < T; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DX - 2) * XL) : "memory");   // slab g has landed
            if constexpr (!(RS_PROBE & 2)) __builtin_amdgcn_s_barrier();           // consumers: slab g is yours, slot of slab g - 1 is mine
            if constexpr (!(RS_PROBE & 8)) issue(g + DX - 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dummy slabs issued past the end must not outlive the ring's reuse below
    } else {
        // ---------------- consumers: weight fragments of stage g in ring slot g % DW (static: the loop is unrolled by DW) ----------------
        const op16_t* wp = W + (int64_t)ntile * (K >> 5) * (WP == 2 ? 256 : 512) + lane * 8;   // (fp8: half the bytes per tile)
        typename std::conditional<WP == 2, rs_w8, op16x8>::type wf[DW][LPS];
        int li = 0, lc = 0;     // next stage to issue
        int ci = 0, cc = 0;     // stage being computed
#define RS_ISSUE_W(slot)                                                                                                             \
    do {                                                                                                                             \
        int kb_ = -1;                                                                                                                \
        if (li < VPW) {                                                                                                              \
            kb_ = sp * VPW + li + 8 * lc;                                                                                            \
            if (++lc == vcount(li)) { lc = 0; ++li; }                                                                                \
        }                                                                                                                            \
        const op16_t* ws_ = kb_ >= 0 ? ((RS_PROBE & 1) ? W + lane * 8 + (kb_ & 1) * 2048 : wp + (int64_t)((RS_PROBE & 8192) ? (kb_ + ((int)blockIdx.x * RS_W + wave) * 5) % nkb : kb_) * (WP == 2 ? 1024 : 2048)) : X + lane * 8; \
        _Pragma("unroll") for (int j_ = 0; j_ < LPS; ++j_)                                                                          \
            wf[slot][j_] = __builtin_nontemporal_load((const typename std::remove_reference<decltype(wf[0][0])>::type*)(ws_ + j_ * 512)); \
    } while (0)
#pragma unroll
        for (int d = 0; d < DW - 1; ++d) RS_ISSUE_W(d);
        if constexpr (rows_sumsq_at_head(MB)) {
            if (nrm.in_sumsq && !(RS_PROBE & 16)) {
                float* ssq_ = (float*)(rs_smem + DX * SLAB);
                rows_sumsq<MB, 256, 3>(nrm, ssq_, tid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                       // head barrier (the producer joins it): every (q, row) sum is in LDS
                rows_sumsq_total<MB>(ssq_, ssq_ + MB * 512, tid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // published by the stage barriers; read in the finish, many barriers later
            }
        }
        unsigned long long* stamps = (unsigned long long*)(rs_smem + DX * SLAB + MB * (2048 + 64));
        const bool stamping = (RS_PROBE & 16384) && S == 1 && wave == 0 && blockIdx.x == 8;
#define RS_STAMP(k_)                                                                       \
    if constexpr (RS_PROBE & 16384) {                                                      \
        if (stamping && g < 64) {                                                          \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                    \
            if (lane == 0) stamps[g * 4 + (k_)] = t_;                                      \
        }                                                                                  \
    }
        if constexpr (RS_PROBE & 16384) {
            if (stamping) {
                for (int i = lane; i < 256; i += 64) stamps[i] = 0ull;
            }
        }
        for (int g0 = 0; g0 < T; g0 += DW) {
#pragma unroll
            for (int u = 0; u < DW; ++u) {
                const int g = g0 + u;
                if (g < T) {
                    RS_STAMP(0)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * LPS) : "memory");   // the weights of stage g have landed
                    RS_STAMP(1)
                    if constexpr (!(RS_PROBE & 2)) __builtin_amdgcn_s_barrier();         // ... and its slab (producer)
                    RS_STAMP(2)
                    op16x8 pre[rows_early(MB) ? RS_PF : 1];
                    if constexpr (rows_early(MB)) {
                        const char* xs0 = rs_smem + (g % DX) * SLAB + lane * 16;
#pragma unroll
                        for (int i = 0; i < RS_PF; ++i) pre[i] = *(const op16x8*)(xs0 + i * 1024);
                    }
                    RS_ISSUE_W((u + DW - 1) % DW);
                    {
                        const char* xs = rs_smem + (g % DX) * SLAB + lane * 16;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            op16x8 wj;
                            if constexpr (WP == 2) wj = fp8x8_to_op16x8(wf[u][j >> 1][(j & 1) * 2], wf[u][j >> 1][(j & 1) * 2 + 1]);
                            else wj = wf[u][j];
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) {
                                const op16x8 xf = (rows_early(MB) && j * MB + mb < RS_PF) ? pre[(j * MB + mb) % RS_PF]
                                                                                           : *(const op16x8*)(xs + ((RS_PROBE & 4096) ? 0 : (j * MB + mb) * 1024));
                                if constexpr (RS_PROBE & 4) {
                                    if (mb == 0) acc[0][0] += __builtin_bit_cast(f32x4, wj)[j & 3] + __builtin_bit_cast(f32x4, xf)[0];
                                } else {
                                    acc[mb] = rv_mfma16(wj, xf, acc[mb]);
                                }
                            }
                        }
                        rows_stage_schedule<MB, WP>();
                    }
                    RS_STAMP(3)
                    if (++cc == vcount(ci)) {     // virtual wave ci is complete: fold it into the tree (binary-counter merge of adjacent subtrees)
                        if constexpr (VPW > 1) {
                            bool placed = false;
#pragma unroll
                            for (int b = 0; b < LOG; ++b) {
                                if (placed) continue;
                                if ((ci >> b) & 1) {
#pragma unroll
                                    for (int mb = 0; mb < MB; ++mb) acc[mb] = stk[b][mb] + acc[mb];
                                } else {
#pragma unroll
                                    for (int mb = 0; mb < MB; ++mb) stk[b][mb] = acc[mb];
                                    placed = true;
                                }
                            }
                            if (placed) {
#pragma unroll
                                for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        }
                        cc = 0;
                        ++ci;
                    }
                }
            }
        }
#undef RS_ISSUE_W
#undef RS_STAMP
        if constexpr (RS_PROBE & 16384) {
            if (stamping) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (int i = lane; i < 256; i += 64) ((unsigned long long*)planes)[i] = stamps[i];
            }
        }
    }
    const unsigned plane = (unsigned)(N >> 4) * MB * 1024;   // bytes per split plane (S planes <= 36 MiB: 32-bit offsets)
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(planes, 0, 0x7ffffff0, 0x00020000);
    // S == 1: the waves' sums change hands through LDS, no plane round trip.  The exchange area lies BEHIND the sums-of-squares area
    // ssq [MB][32][16] f32 = MB * 2 KiB (18 KiB at 9 row blocks: a fixed 16 KiB offset let ssq[8] overlap the sums of tile 0)
    constexpr int XCH_OFF = MB * 2048 > 16384 ? MB * 2048 : 16384;
    static_assert(XCH_OFF + RS_W * MB * 1024 <= DX * MB * 4096, "the exchange area must fit the slab ring");
    char* xch = rs_smem + XCH_OFF;
    if constexpr (S == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the trailing LDS-DMA padding stages are done)
        __syncthreads();                                   // everyone is through with the slab ring
        if (wave < RS_W) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) *(f32x4*)(xch + ((wave * MB + mb) * 64 + lane) * 16) = acc[mb];
        }
    } else {
        if (wave < RS_W) {
            const unsigned off = sp * plane + ((unsigned)ntile * MB * 64 + lane) * 16;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) st_sc1(pr, off + mb * 1024, acc[mb]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the planes are written through (and the trailing LDS-DMA padding stages are done)
    }
    __syncthreads();
    // ---- the LAST of the S workgroups of the column group to arrive finishes it ----
    // Arrival counter: one per (column group, log2 S), never reset - a launch adds exactly S to it (launches of one stream are ordered
    // and every engine slot owns its workspace), so the last arrival of this launch is the one that reads S - 1 (mod S).  Nobody
    // WAITS for anybody: a workgroup that spun for the others could deadlock against the stream-K prefill GEMMs of another stream,
    // whose workgroups spin for each other too (seen: both kernels half resident, each holding the CUs the other needs).
    __shared__ int is_last;
    if constexpr (S > 1) {
        if (tid == 0) {
            unsigned* cnt = (unsigned*)nrm.arrive + (LOG * (RV_ROWS_COUNTERS / 4) + cg);     // (LOG of VPW: 0 .. 2 here)
            is_last = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % S == S - 1;
        }
        __syncthreads();
        if (!is_last) return;
    }
    constexpr int NT = (FIN == 1 || FIN == 2) ? 2 : 1;
    constexpr int OUT_BF16 = (FIN == 1 || FIN == 4) ? 1 : 0;
    constexpr int ACT = FIN == 1 ? RV_ACT_SILU_MUL : RV_ACT_NONE;
    constexpr int ROPE = FIN == 3 ? 1 : 0;
    constexpr int BPG = RS_W / NT;                      // blocks (of NT tiles) per column group
    constexpr int PAIRS = BPG * MB;                     // (block, row block) pairs of the group; pair p = block * MB + mb
    constexpr int PPW = (PAIRS + RS_W - 1) / RS_W;      // pairs per consumer wave: p = wave + RS_W * i  ->  mb = p % MB
    const int fr = lane & 15, kg = lane >> 4;
    const int nblk = N / (16 * NT);
    // (1) the partial planes of this wave's pairs: agent-coherent loads, all in flight at once (one memory latency, not PPW of them)
    f32x4 pl[PPW][NT][S];
    if (wave < RS_W) {
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + RS_W * i < PAIRS ? wave + RS_W * i : PAIRS - 1;      // (ragged last pass: a valid pair, not used)
            const int blk = cg * BPG + p / MB, mb = p % MB;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (S == 1) {
                    pl[i][t][0] = *(const f32x4*)(xch + (((p / MB * NT + t) * MB + mb) * 64 + lane) * 16);
                } else {
                    const unsigned q = (((unsigned)(blk * NT + t) * MB + mb) * 64 + lane) * 16;
#pragma unroll
                    for (int w = 0; w < S; ++w) pl[i][t][w] = ld_sc1(pr, q + w * plane);
                }
            }
        }
    }
    // (2) the rows' sums of squares [MB][32][16]: added up at the head of the workgroup behind the slab ring (8 / 9 row blocks), or here
    //     into the freed ring (4 / 5 row blocks)
    float* ssq = (float*)(rs_smem + (rows_sumsq_at_head(MB) ? DX * SLAB : 0));
    if constexpr (!rows_sumsq_at_head(MB)) {
        if (nrm.in_sumsq && !(RS_PROBE & 16)) {
            rows_sumsq<MB, RS_THREADS, 2>(nrm, ssq, tid);
            __syncthreads();
            rows_sumsq_total<MB>(ssq, ssq + MB * 512, tid);
            __syncthreads();
        }
    }
    const float* totl = ssq + MB * 512;
    if (wave >= RS_W) return;
    if constexpr (ROPE) {
        if (qr.row_pos && qr.G == 1 && !(RS_PROBE & 64)) {      // merged decode step: the finish written for it
            f32x4 sv[PPW];
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                if constexpr (S == 8) sv[i] = gemv_tree8(pl[i][0]);
                else if constexpr (S == 4) sv[i] = (pl[i][0][0] + pl[i][0][1]) + (pl[i][0][2] + pl[i][0][3]);
                else if constexpr (S == 2) sv[i] = pl[i][0][0] + pl[i][0][1];
                else sv[i] = pl[i][0][0];
            }
            rows_qkv_finish<MB, PPW, PAIRS, WP>(sv, wave, cg, fr, kg, M, nrm, qr, totl);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int p = wave + RS_W * i;
        if (p >= PAIRS) break;
        const int blk = cg * BPG + p / MB, mb = p % MB;
        f32x4 sres[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (S == 8) sres[t] = gemv_tree8(pl[i][t]);
            else if constexpr (S == 4) sres[t] = (pl[i][t][0] + pl[i][t][1]) + (pl[i][t][2] + pl[i][t][3]);
            else if constexpr (S == 2) sres[t] = pl[i][t][0] + pl[i][t][1];
            else sres[t] = pl[i][t][0];
        }
        const float tot = nrm.in_sumsq ? totl[mb * 16 + fr] : 0.f;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        gemv_finish<NT, OUT_BF16, ACT, WP, ROPE>(sres, mb, fr, kg, blk, nblk, M, N, bias, res, ldr, Cv, ldc, nrm, qr, tot, z, z, z, z, false);
    }
}

// PERSISTENT launches (round 3).  A work item = (64-column group cg, split sp) = T stages.  When a launch has more items than workgroups fit the
// chip at once (gate/up at 65 .. 80 rows: 688 items on 512 slots) the old one-workgroup-per-item launch ran TWO rounds and each round paid the
// fixed ~10 us of a workgroup's life (cold first weights; planes out and acknowledged; the arrival counter's round trip; the last arriver's plane
// loads and epilogue).  Now such a launch is a grid of resident workgroups that walk items it, it + G, ... (G a multiple of S: the split sp never
// changes) as ONE stream of stages - the weight ring and the slab ring run on across item boundaries - and DEFER the hand-over: at the end of an
// item the consumers only store its partial planes (no wait) and clear their accumulators; the acknowledgement, the arrival counters of all the
// workgroup's items and the finishes of the groups it completed come once, behind the stream.  Finer items (S = 4: 1376 items of 8 stages, <= 3 per
// workgroup) then balance the stream lengths.  An item occupies T rounded up to the ring depth stage SLOTS (pad slots load an L2-hot dummy and
// compute nothing), so the ring slot of a stage is a compile-time constant.  grid == items is the old launch.  Same sums, same finish: same bits.
constexpr int RS_MAXIT = 8;   // items per workgroup at most
// (A separate kernel: folded into rows_kernel, the item bookkeeping cost the 5-row-block S = 2 variant 8 spilled VGPRs per stage.)
template <int MB, int VPW, int FIN, int WP = 1>
__global__ __attribute__((amdgpu_flat_work_group_size(RS_THREADS, RS_THREADS), amdgpu_waves_per_eu(RowsCfg<MB>::WPE, RowsCfg<MB>::WPE))) void
rows_kernel_p(const op16_t* __restrict__ X, const op16_t* __restrict__ W, float* __restrict__ planes, const float* __restrict__ bias,
            const float* res, int64_t ldr, void* Cv, int64_t ldc, int M, int N, int K, GemvNorm nrm, QkvRope qr) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    constexpr int S = 8 / VPW, LOG = VPW == 8 ? 3 : VPW == 4 ? 2 : VPW == 2 ? 1 : 0;
    constexpr int DW = RowsCfg<MB, WP>::DW, DX = RowsCfg<MB, WP>::DX, LPS = RowsCfg<MB, WP>::LPS;
    constexpr int SLAB = MB * 4096;                   // bytes of one 128-k slab of MB row blocks
    constexpr int XL = MB * 4;                        // LDS-DMA fragments (1 KiB) per slab
    static_assert((DX - 2) * XL <= 63 && (DW - 2) * LPS <= 63, "vmcnt is a 6-bit counter");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nitems = (N >> 6) * S, G = (int)gridDim.x;   // (host: G % S == 0 or G == nitems, so this workgroup's split never changes)
    const int sp = blockIdx.x % S;
    const int nkb = K >> 7;
    const int n_it = (nitems - (int)blockIdx.x + G - 1) / G;    // items of this workgroup: blockIdx.x + n * G   (host: <= RS_MAXIT; S == 1: 1)
    // stage sequence of ONE item: virtual waves v = sp * VPW + i, i = 0 .. VPW - 1, one after the other; virtual wave v owns k-blocks
    // v, v + 8, ... (cnt(v) of them): T stages, Tp slots.  Both roles run exactly n_it * Tp slots (+ slots issued past the end as L2-hot dummies).
    auto vcount = [&](int i) -> int { return (nkb - (sp * VPW + i) + 7) >> 3; };
    int T = 0;
#pragma unroll
    for (int i = 0; i < VPW; ++i) T += vcount(i);
    const int Tp = (T + DW - 1) / DW * DW;
    const unsigned plane = (unsigned)(N >> 4) * MB * 1024;   // bytes per split plane (S planes <= 36 MiB: 32-bit offsets)
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(planes, 0, 0x7ffffff0, 0x00020000);

    f32x4 acc[MB], stk[LOG ? LOG : 1][MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (wave == RS_W) {
        // ---------------- producer: slab of slot g -> LDS slot g % DX (the same k-block sequence for every item: sp is fixed) ----------------
        int pi = 0, pc = 0, pp = 0, pn = 0;     // next slab to issue: (virtual wave, k-block in it, slot in the item, item)
        auto issue = [&](int g) {
            int kb = -1;
            if (pn < n_it) {
                if (pp < T) {
                    kb = sp * VPW + pi + 8 * pc;
                    if (++pc == vcount(pi)) { pc = 0; ++pi; }
                }
                if (++pp == Tp) { pp = 0; pi = 0; ++pn; }
            }
            const op16_t* xs = X + (kb >= 0 ? (int64_t)kb * (MB * 2048) : 0) + lane * 8;
            char* dst = rs_smem + (g % DX) * SLAB;
#pragma unroll
            for (int i = 0; i < XL; ++i) glds16(xs + i * 512, dst + i * 1024);
        };
        if constexpr (!(RS_PROBE & 8))
            for (int d = 0; d < DX - 1; ++d) issue(d);
        if constexpr (rows_sumsq_at_head(MB)) {
            if (nrm.in_sumsq && !(RS_PROBE & 16)) __builtin_amdgcn_s_barrier();   // the consumers' head barrier (rows_sumsq)
        }
        const int TT = n_it * Tp;
        for (int g = 0; g < TT; ++g) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DX - 2) * XL) : "memory");   // slab g has landed
            if constexpr (!(RS_PROBE & 2)) __builtin_amdgcn_s_barrier();           // consumers: slab g is yours, slot of slab g - 1 is mine
            if constexpr (!(RS_PROBE & 8)) issue(g + DX - 1);
        }
    } else {
        // ---------------- consumers: weight fragments of slot g in ring slot g % DW = (slot in the item) % DW: a compile-time constant ----------------
        const int64_t tile_stride = (int64_t)(K >> 5) * (WP == 2 ? 256 : 512);     // elements of one 16-column tile (fp8: half the bytes)
        const int64_t item_stride = (int64_t)(G / S) * RS_W * tile_stride;         // the next item of this workgroup: column group + G / S
        const op16_t* wp = W + ((int)blockIdx.x / S * RS_W + wave) * tile_stride + lane * 8;     // this wave's tile of the item being ISSUED
        typename std::conditional<WP == 2, rs_w8, op16x8>::type wf[DW][LPS];
        int li = 0, lc = 0, lp = 0, ln = 0;     // next slot to issue: (virtual wave, k-block in it, slot in the item, item)
#define RS_ISSUE_W(slot)                                                                                                             \
    do {                                                                                                                             \
        const op16_t* ws_ = X + lane * 8;                                                                                            \
        if (ln < n_it) {                                                                                                             \
            if (lp < T) {                                                                                                            \
                const int kb_ = sp * VPW + li + 8 * lc;                                                                              \
                ws_ = (RS_PROBE & 1) ? W + lane * 8 + (kb_ & 1) * 2048 : wp + (int64_t)((RS_PROBE & 8192) ? (kb_ + ((int)blockIdx.x * RS_W + wave) * 5) % (K >> 7) : kb_) * (WP == 2 ? 1024 : 2048);                 \
                if (++lc == vcount(li)) { lc = 0; ++li; }                                                                            \
            }                                                                                                                        \
            if (++lp == Tp) { lp = 0; li = 0; ++ln; wp += item_stride; }                                                             \
        }                                                                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < LPS; ++j_)                                                                          \
            wf[slot][j_] = __builtin_nontemporal_load((const typename std::remove_reference<decltype(wf[0][0])>::type*)(ws_ + j_ * 512)); \
    } while (0)
#pragma unroll
        for (int d = 0; d < DW - 1; ++d) RS_ISSUE_W(d);
        if constexpr (rows_sumsq_at_head(MB)) {
            if (nrm.in_sumsq && !(RS_PROBE & 16)) {
                float* ssq_ = (float*)(rs_smem + DX * SLAB);
                rows_sumsq<MB, 256, 3>(nrm, ssq_, tid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                       // head barrier (the producer joins it): every (q, row) sum is in LDS
                rows_sumsq_total<MB>(ssq_, ssq_ + MB * 512, tid);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // published by the stage barriers; read in the finish, many barriers later
            }
        }
        int g = 0;
        for (int n = 0; n < n_it; ++n) {
            int ci = 0, cc = 0;             // stage being computed: (virtual wave, k-block in it)
            for (int s0 = 0; s0 < Tp; s0 += DW) {
#pragma unroll
                for (int u = 0; u < DW; ++u, ++g) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * LPS) : "memory");   // the weights of slot g have landed
                    if constexpr (!(RS_PROBE & 2)) __builtin_amdgcn_s_barrier();         // ... and its slab (producer)
                    op16x8 pre[rows_early(MB) ? RS_PF : 1];
                    if constexpr (rows_early(MB)) {       // (a pad slot reads its L2-hot dummy slab and uses nothing of it)
                        const char* xs0 = rs_smem + (g % DX) * SLAB + lane * 16;
#pragma unroll
                        for (int i = 0; i < RS_PF; ++i) pre[i] = *(const op16x8*)(xs0 + i * 1024);
                    }
                    RS_ISSUE_W((u + DW - 1) % DW);
                    if (s0 + u < T) {
                        const char* xs = rs_smem + (g % DX) * SLAB + lane * 16;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            op16x8 wj;
                            if constexpr (WP == 2) wj = fp8x8_to_op16x8(wf[u][j >> 1][(j & 1) * 2], wf[u][j >> 1][(j & 1) * 2 + 1]);
                            else wj = wf[u][j];
#pragma unroll
                            for (int mb = 0; mb < MB; ++mb) {
                                const op16x8 xf = (rows_early(MB) && j * MB + mb < RS_PF) ? pre[(j * MB + mb) % RS_PF]
                                                                                           : *(const op16x8*)(xs + ((RS_PROBE & 4096) ? 0 : (j * MB + mb) * 1024));
                                if constexpr (RS_PROBE & 4) {
                                    if (mb == 0) acc[0][0] += __builtin_bit_cast(f32x4, wj)[j & 3] + __builtin_bit_cast(f32x4, xf)[0];
                                } else {
                                    acc[mb] = rv_mfma16(wj, xf, acc[mb]);
                                }
                            }
                        }
                        rows_stage_schedule<MB, WP>();
                        if (++cc == vcount(ci)) {     // virtual wave ci is complete: fold it into the tree (binary-counter merge of adjacent subtrees)
                            if constexpr (VPW > 1) {
                                bool placed = false;
#pragma unroll
                                for (int b = 0; b < LOG; ++b) {
                                    if (placed) continue;
                                    if ((ci >> b) & 1) {
#pragma unroll
                                        for (int mb = 0; mb < MB; ++mb) acc[mb] = stk[b][mb] + acc[mb];
                                    } else {
#pragma unroll
                                        for (int mb = 0; mb < MB; ++mb) stk[b][mb] = acc[mb];
                                        placed = true;
                                    }
                                }
                                if (placed) {
#pragma unroll
                                    for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                                }
                            }
                            cc = 0;
                            ++ci;
                        }
                    }
                }
            }
            if constexpr (S > 1) {          // the item's partial sums: planes out, NOT waited for; the stream goes on
                const unsigned ntile = (unsigned)(((int)blockIdx.x + n * G) / S * RS_W + wave);
                const unsigned off = sp * plane + (ntile * MB * 64 + lane) * 16;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    st_sc1(pr, off + mb * 1024, acc[mb]);
                    acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#undef RS_ISSUE_W
    }
    // ---------------- behind the stream: acknowledge, count arrivals, finish what this workgroup completed ----------------
    // S == 1: the waves' sums change hands through LDS, no plane round trip.  The exchange area lies BEHIND the sums-of-squares area
    // ssq [MB][32][16] f32 = MB * 2 KiB (18 KiB at 9 row blocks: a fixed 16 KiB offset let ssq[8] overlap the sums of tile 0)
    constexpr int XCH_OFF = MB * 2048 > 16384 ? MB * 2048 : 16384;
    static_assert(XCH_OFF + RS_W * MB * 1024 <= DX * MB * 4096, "the exchange area must fit the slab ring");
    char* xch = rs_smem + XCH_OFF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the planes are written through; the trailing LDS-DMA / weight padding loads are done
    __syncthreads();                                   // everyone is through with the slab ring
    if constexpr (S == 1) {
        if (wave < RS_W) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) *(f32x4*)(xch + ((wave * MB + mb) * 64 + lane) * 16) = acc[mb];
        }
        __syncthreads();
    }
    // The LAST of the S workgroups of a column group to arrive finishes it.  Arrival counter: one per (column group, log2 S), never reset - a
    // launch adds exactly S to it (launches of one stream are ordered and every engine slot owns its workspace), so the last arrival of this
    // launch is the one that reads S - 1 (mod S).  Nobody WAITS for anybody: a workgroup that spun for the others could deadlock against the
    // stream-K prefill GEMMs of another stream, whose workgroups spin for each other too.
    __shared__ int is_last[RS_MAXIT];
    if constexpr (S > 1) {
        if (tid < n_it) {
            unsigned* cnt = (unsigned*)nrm.arrive + (LOG * (RV_ROWS_COUNTERS / 4) + ((int)blockIdx.x + tid * G) / S);     // (LOG of VPW: 0 .. 2 here)
            is_last[tid] = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) % S == S - 1;
        }
        __syncthreads();
        bool any = false;
        for (int n = 0; n < n_it; ++n) any = any || is_last[n];
        if (!any) return;
    }
    constexpr int NT = (FIN == 1 || FIN == 2) ? 2 : 1;
    constexpr int OUT_BF16 = (FIN == 1 || FIN == 4) ? 1 : 0;
    constexpr int ACT = FIN == 1 ? RV_ACT_SILU_MUL : RV_ACT_NONE;
    constexpr int ROPE = FIN == 3 ? 1 : 0;
    constexpr int BPG = RS_W / NT;                      // blocks (of NT tiles) per column group
    constexpr int PAIRS = BPG * MB;                     // (block, row block) pairs of the group; pair p = block * MB + mb
    constexpr int PPW = (PAIRS + RS_W - 1) / RS_W;      // pairs per consumer wave: p = wave + RS_W * i  ->  mb = p % MB
    const int fr = lane & 15, kg = lane >> 4;
    const int nblk = N / (16 * NT);
    // (2) the rows' sums of squares [MB][32][16]: added up at the head of the workgroup behind the slab ring (8 / 9 row blocks), or here
    //     into the freed ring (4 / 5 row blocks)
    float* ssq = (float*)(rs_smem + (rows_sumsq_at_head(MB) ? DX * SLAB : 0));
    if constexpr (!rows_sumsq_at_head(MB)) {
        if (nrm.in_sumsq && !(RS_PROBE & 16)) {
            rows_sumsq<MB, RS_THREADS, 2>(nrm, ssq, tid);
            __syncthreads();
            rows_sumsq_total<MB>(ssq, ssq + MB * 512, tid);
            __syncthreads();
        }
    }
    const float* totl = ssq + MB * 512;
    if (wave >= RS_W) return;
    for (int n = 0; n < n_it; ++n) {
        if constexpr (S > 1) {
            if (!is_last[n]) continue;
        }
        const int cg = ((int)blockIdx.x + n * G) / S;
        // (1) the partial planes of this wave's pairs: agent-coherent loads, all in flight at once (one memory latency, not PPW of them)
        f32x4 pl[PPW][NT][S];
        if (wave < RS_W) {
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int p = wave + RS_W * i < PAIRS ? wave + RS_W * i : PAIRS - 1;      // (ragged last pass: a valid pair, not used)
                const int blk = cg * BPG + p / MB, mb = p % MB;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (S == 1) {
                        pl[i][t][0] = *(const f32x4*)(xch + (((p / MB * NT + t) * MB + mb) * 64 + lane) * 16);
                    } else {
                        const unsigned q = (((unsigned)(blk * NT + t) * MB + mb) * 64 + lane) * 16;
#pragma unroll
                        for (int w = 0; w < S; ++w) pl[i][t][w] = ld_sc1(pr, q + w * plane);
                    }
                }
            }
        }
        // (the q/k/v finish written for merged steps, rows_qkv_finish, is used by the one-item-per-workgroup kernel only: inside this item loop
        //  its prefetch arrays cost the persistent instantiations 10 - 100 spilled registers, and no 7B launch with a RoPE epilogue is persistent)
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int p = wave + RS_W * i;
            if (p >= PAIRS) break;
            const int blk = cg * BPG + p / MB, mb = p % MB;
            f32x4 sres[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (S == 8) sres[t] = gemv_tree8(pl[i][t]);
                else if constexpr (S == 4) sres[t] = (pl[i][t][0] + pl[i][t][1]) + (pl[i][t][2] + pl[i][t][3]);
                else if constexpr (S == 2) sres[t] = pl[i][t][0] + pl[i][t][1];
                else sres[t] = pl[i][t][0];
            }
            const float tot = nrm.in_sumsq ? totl[mb * 16 + fr] : 0.f;
            const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
            gemv_finish<NT, OUT_BF16, ACT, WP, ROPE>(sres, mb, fr, kg, blk, nblk, M, N, bias, res, ldr, Cv, ldc, nrm, qr, tot, z, z, z, z, false);
        }
    }
}

int rows_num_cus(int dev) {     // CUs of the device (cached per device id)
    static std::atomic<int> cus[64];
    int n = cus[dev & 63].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev & 63].store(n, std::memory_order_relaxed);
    }
    return n;
}
// workgroups resident at once: two per CU up to 5 row blocks (60 KiB of LDS, 168 VGPRs), one with 8
int rows_slots(int MBp) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return (rows_num_cus(dev) & ~7) * (MBp <= 5 ? 2 : 1);
}

int rows_splits(int64_t N, int MBp) {   // workgroups per 64-column group: the smallest power of two that fills the CUs, <= 8
    const int64_t cg = N / 64;
    int s = MBp >= 5 ? 2 : 1;              // (5 / 8 row blocks x all 8 virtual waves in one workgroup: 4 accumulator sets do not fit the VGPRs:
                                           //  measured with the 68 B of spills it takes, 70-row step 6.85 -> 7.05 ms)
    const int fill = rv_cur_opts().rows_fill;   // workgroups a launch should at least have (tunable; 2 per CU are resident with <= 5 row blocks)
    // 8 / 9 row blocks (one workgroup per CU, 256 registers per wave: all 8 virtual waves fit): a launch whose column groups alone fill
    // >= 3/4 of the CUs runs UNSPLIT - no partial planes, no hand-over, every workgroup finishes its own columns (the QKV projection:
    // 192 groups on 256 CUs)
    if (MBp >= 8 && rv_cur_opts().rows_single && cg <= rows_slots(MBp) && cg * 4 >= (int64_t)rows_slots(MBp) * 3) return 1;
    while (s < 8 && cg * s < fill) s *= 2;
    // More items than resident workgroups: the launch becomes a persistent grid (rows_launch); finer items then balance the streams.
    // Cost = the longest stream in virtual k-waves; a finer split must shorten it by >= 20 % to pay for its extra planes.
    if (rv_cur_opts().rows_persistent) {
        const int64_t slots = rows_slots(MBp);
        auto longest = [&](int s_) { return (cg * s_ + slots - 1) / slots * (8 / s_); };
        if (cg * s > slots)
            for (int s2 = s * 2; s2 <= 8; s2 *= 2)
                if (longest(s2) * 5 <= longest(s) * 4) s = s2;
    }
    return s;
}

#define RS_KERNEL (PERS ? rows_kernel_p<MB, VPW, FIN, WP> : rows_kernel<MB, VPW, FIN, WP>)
template <int MB, int VPW, int FIN, int WP, int PERS>
int rows_launch(const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int M, int N, int K,
                const GemvNorm& nrm, const QkvRope& qr, hipStream_t st) {
    size_t lds = (size_t)RowsCfg<MB, WP>::DX * MB * 4096 + (rows_sumsq_at_head(MB) ? (size_t)MB * (2048 + 64) : 0)      // the slab ring (+ the rows' sums of squares [MB][32][16] f32)
                 + ((RS_PROBE & 16384) ? 2048 : 0);
    // Spreading.  The dispatcher packs workgroups two to a CU (<= 5 row blocks): a launch of 256 workgroups (the o / down projections) then
    // occupies 128 of the 256 CUs (PMC: SQ_BUSY_CU_CYCLES = 0.50 of the launch).  A launch with no more workgroups than `rows_spread`
    // asks for more LDS than two workgroups can share, so each gets a CU of its own.
    const unsigned items = (unsigned)(N / 64 * (8 / VPW));
    unsigned grid = items;
    if constexpr (PERS) {
        const unsigned slots = (unsigned)rows_slots(MB);
        if (slots >= 8 && slots < items) grid = slots;       // a multiple of 8, hence of S: a workgroup's split never changes
    }
    RV_CHECK_ARG(items <= (unsigned)RS_MAXIT * grid, "gemm_rows: %u items for %u workgroups", items, grid);
    constexpr size_t alone = 81 * 1024;
    const bool spread = MB <= 5 && (int)grid <= rv_cur_opts().rows_spread;
    if (spread && lds < alone) lds = alone;
    // the > 64 KiB opt-in is a per-DEVICE attribute of the function: remembered per device (bit d of the mask), not per process
    static std::atomic<uint64_t> opted{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (lds > 65536 && !(opted.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)RS_KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)(MB <= 5 && lds < alone ? alone : lds)) != hipSuccess) {
            rv_set_error("gemm_rows: cannot reserve %zu bytes of LDS", lds);
            return RV_ERR_HIP;
        }
        opted.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((RS_KERNEL), dim3(grid), dim3(RS_THREADS), lds, st, X, W, nrm.planes, bias, res, ldr, C,
                       ldc, M, N, K, nrm, qr);
    return RV_OK;
}
#undef RS_KERNEL
template <int MB, int FIN, int WP>
int rows_by_split(int S, const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int M, int N, int K,
                  const GemvNorm& nrm, const QkvRope& qr, hipStream_t st) {
    // more items than resident workgroups (and a real split): the persistent grid with deferred hand-overs
    // (5 row blocks: only the S = 4 instantiation of the persistent kernel is spill-free - S = 2 needs 8 more VGPRs than three waves per SIMD have)
    const bool pers = S > 1 && rv_cur_opts().rows_persistent && (int64_t)(N / 64) * S > rows_slots(MB) && (MB != 5 || S == 4);
#define RS_LAUNCH(VPW_) (pers ? rows_launch<MB, VPW_, FIN, WP, 1>(X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st) \
                              : rows_launch<MB, VPW_, FIN, WP, 0>(X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st))
    switch (S) {
        case 1: return rows_launch<MB, 8, FIN, WP, 0>(X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st);
        case 2: return RS_LAUNCH(4);
        case 4: return RS_LAUNCH(2);
        default: return RS_LAUNCH(1);
    }
#undef RS_LAUNCH
}
template <int FIN, int WP>
int rows_by_mb(int MBp, int S, const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int M, int N,
               int K, const GemvNorm& nrm, const QkvRope& qr, hipStream_t st) {
#if RS_PART & 2
    return MBp == 8 ? rows_by_split<8, FIN, WP>(S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st)
                    : rows_by_split<9, FIN, WP>(S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st);
#else
    return MBp == 4 ? rows_by_split<4, FIN, WP>(S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st)
                    : rows_by_split<5, FIN, WP>(S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, qr, st);
#endif
}
template <int WP>
int rows_by_fin(int MBp, int S, const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int out_dtype,
                int act, int M, int N, int K, const GemvNorm& nrm, const QkvRope* qr, hipStream_t st) {
    const QkvRope q0{};
    if (qr) return rows_by_mb<3, WP>(MBp, S, X, W, nullptr, nullptr, 0, nullptr, 0, M, N, K, nrm, *qr, st);
    if (act == RV_ACT_SILU_MUL) return rows_by_mb<1, WP>(MBp, S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, q0, st);
    if (N >= 16384 && out_dtype == RV_F32) return rows_by_mb<2, WP>(MBp, S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, q0, st);
    if (N < 16384 && out_dtype == RV_OP16) return rows_by_mb<4, WP>(MBp, S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, q0, st);
    if (N < 16384) return rows_by_mb<0, WP>(MBp, S, X, W, bias, res, ldr, C, ldc, M, N, K, nrm, q0, st);
    rv_set_error("gemm_rows: bf16 output with N >= 16384 is not instantiated");
    return RV_ERR_ARG;
}

}  // namespace

// This file is compiled four times (build.py; gemm_rows_p1 / _p2 / _p3.hip include it with RS_PART set): bit 0 = FP8 weights, bit 1 = the
// 8 / 9 row-block instantiations.  Part 0 (this file itself) also holds the entry points.
#define RS_PART_FN_(n) gemm_rows_part##n
#define RS_PART_FN(n) RS_PART_FN_(n)
int RS_PART_FN(RS_PART)(int MBp, int S, const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
                        int out_dtype, int act, int M, int N, int K, const GemvNorm& nrm, const QkvRope* qr, hipStream_t st) {
    return rows_by_fin<(RS_PART & 1) ? 2 : 1>(MBp, S, X, W, bias, res, ldr, C, ldc, out_dtype, act, M, N, K, nrm, qr, st);
}

#if RS_PART == 0
#define RS_PART_DECL(n)                                                                                                                      \
    int gemm_rows_part##n(int MBp, int S, const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, \
                          int out_dtype, int act, int M, int N, int K, const GemvNorm& nrm, const QkvRope* qr, hipStream_t st)
RS_PART_DECL(1);
RS_PART_DECL(2);
RS_PART_DECL(3);

// S * (N / 16) tiles * MB KiB with S = rows_splits(N, MB): S * N < 2 * 240 * 128 wherever the CU fill asks for S > 2; N <= 32768, S = 2 otherwise
size_t gemm_rows_ws_bytes() { return (size_t)(2 * 32768 / 16) * RV_XP_MAX_BLOCKS * 1024; }
extern "C" size_t rv_gemm_rows_ws_bytes(void) { return gemm_rows_ws_bytes(); }

// X: 33 .. 144 fragment-packed rows (nrm.x_packed row blocks); nrm.planes: zero-initialised workspace of gemm_rows_ws_bytes().
// qr != nullptr: the fused q/k/v + RoPE epilogue.  w_layout 1: bf16 fragment-packed W; 2: FP8 fragment-packed W + nrm.w_scale.
int gemm_rows(const op16_t* X, const op16_t* W, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc, int out_dtype, int act,
              int M, int N, int K, hipStream_t st, const GemvNorm& nrm, const QkvRope* qr, int w_layout) {
    RV_CHECK_ARG(w_layout == 1 || (w_layout == 2 && nrm.w_scale), "gemm_rows: bf16 (1) or fp8 + per-row scales (2) fragment-packed weights");
    RV_CHECK_ARG(M > 32 && M <= RV_ROWS_MAX && nrm.x_packed == rv_xp_blocks(M) && nrm.planes && nrm.arrive,
                 "gemm_rows: 33 .. 144 fragment-packed rows, a plane workspace and arrival counters");
    RV_CHECK_ARG(N / 64 <= RV_ROWS_COUNTERS / 4, "gemm_rows: too many column groups");
    RV_CHECK_ARG(N % 64 == 0 && K % 128 == 0 && K >= 1024 && N <= 32768, "gemm_rows: N %% 64, K %% 128, K >= 1024, N <= 32768");
    RV_CHECK_ARG(act == RV_ACT_NONE || act == RV_ACT_SILU_MUL, "gemm_rows: no activation or SILU_MUL");
    RV_CHECK_ARG(act != RV_ACT_SILU_MUL || out_dtype == RV_OP16, "gemm_rows: SILU_MUL writes bf16");
    const int MBp = nrm.x_packed, S = rows_splits(N, MBp);
    RV_CHECK_ARG((size_t)S * (N / 16) * MBp * 1024 <= gemm_rows_ws_bytes(), "gemm_rows: %d partial planes of N = %d do not fit the plane workspace", S, N);
    auto* part = MBp >= 8 ? (w_layout == 2 ? gemm_rows_part3 : gemm_rows_part2) : (w_layout == 2 ? gemm_rows_part1 : gemm_rows_part0);
    const int rc = part(MBp, S, X, W, bias, res, ldr, C, ldc, out_dtype, act, M, N, K, nrm, qr, st);
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_rows");
    return RV_OK;
}

// Building block (include/revision_hip.h): one projection of a merged decode step on 33 .. 144 fragment-packed rows, with the engine's
// epilogue variants (act = RV_ACT_SILU_MUL + bf16 out: the gate/up launch).
extern "C" int rv_gemm_rows(const void* Xp, const void* Wp, const float* w_scale, void* C, int32_t M, int32_t N, int32_t K, void* planes,
                            int32_t* arrive, int act, int out_dtype, void* stream) {
    RV_CHECK_ARG(Xp && Wp && C && planes && arrive, "rv_gemm_rows: null argument");
    GemvNorm nrm;
    nrm.w_scale = w_scale;
    nrm.x_packed = rv_xp_blocks(M);
    nrm.planes = (float*)planes;
    nrm.arrive = arrive;
    const int ldc = act == RV_ACT_SILU_MUL ? N / 2 : N;
    return gemm_rows((const op16_t*)Xp, (const op16_t*)Wp, nullptr, nullptr, 0, C, ldc, out_dtype, act, M, N, K, (hipStream_t)stream, nrm, nullptr,
                     w_scale ? 2 : 1);
}
#endif   // RS_PART == 0
