// Ping-pong GEMM for the prefill / adapter regime: C[M,N] = A[M,K] . W[N,K]^T, fragment-packed W, 256 x 256 x 64 tiles.
//
// Why a second tiled kernel: the 128 x 128 ring kernel (gemm.hip) moves (128 + 128) x 2 B of operands per 128 x 128 MACs
// and k, which at the MFMA rate is exactly the 64 B/clk a CU's vector-memory path delivers, so it tops out near 0.9
// PFLOP/s whatever the schedule.  A 256 x 256 tile halves the bytes per MAC; what it costs is occupancy (128 KiB of LDS,
// one workgroup per CU, two waves per SIMD), so the overlap has to be built by hand:
//
//   * 8 waves = 2 (M) x 4 (N); every wave owns 128 x 64 outputs = 8 x 4 MFMA fragments (128 accumulator registers).
//     Waves w and w + 4 share a SIMD.  The two M-groups run the same phase sequence ONE BARRIER APART: while group 0
//     issues its LDS reads and LDS-DMA loads (the "memory part" of a phase), group 1 runs its 16-MFMA burst (the "compute
//     part"), and vice versa, so each SIMD always has one wave in its MFMA burst.
//   * a k-tile (64) is four phases, one per 64 x 32 quadrant of the wave's outputs: (m0,n0) (m0,n1) (m1,n1) (m1,n0); the
//     register sub-tiles are loaded once per phase (12 / 4 / 8 / 0 ds_read_b128) and the n0 fragments stay resident.
//   * the operands live in LDS as 16 KiB UNITS ordered by first use: U0 = A rows of the m0 quadrants (both groups),
//     U1 = W columns of the n0 quadrants (all four wave columns), U2 = W n1, U3 = A m1; two A stages and THREE W stages
//     (all 160 KiB).  Every phase issues one unit (2 LDS-DMA pieces of 1 KiB per wave): the A units of tile t+1, the W
//     units of tile t+2 (the weights come cold from HBM and get twice the latency budget).  Loads stay in flight across
//     every barrier: the only waits are COUNTED vmcnt, placed before the barrier that precedes the first read of a unit
//     (schedule and hazard table at pp_mainloop).
//   * A is staged in full 128-byte rows (8 rows per 1 KiB piece) with the 16-byte chunks XOR-swizzled by (row & 7) through
//     the per-lane SOURCE address (LDS-DMA destinations are lane-linear); W pieces are MFMA fragments already.  Both are
//     read back with conflict-free ds_read_b128.
//   * slots = barrier-to-barrier intervals; group 0 runs the memory part of phase k in slot 2k, group 1 in slot 2k+1.  RAW: a
//     unit is read in slot >= 2k only after every wave waited for its own pieces of it before the barrier ending slot 2k-1.
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "kernels.h"

// PP_ABL (compile-time ablation probe, tools/pp_probe.sh; results are garbage, only the time means anything): 1 = the steady state issues no LDS-DMA,
// 2 = no fragment reads, 4 = no barriers, 8 = no MFMAs, 16 = every panel reads the weights of panel 0 (W stays in L2; lda = 0 does the same for A),
// 64 = the plain / gated whole-panel epilogues of the eight-wave form compute but do not store.
#ifndef PP_ABL
#define PP_ABL 0
#endif
#if PP_ABL & 32
// 32 = cycle stamps (s_memtime) behind each of the 8 barriers of a k-tile, summed per slot over the steady-state tiles of workgroup 0 .. 255, waves 0 and 4:
// pp_stamp_buf[(workgroup * 2 + group) * 9 + slot], [.. + 8] = k-tiles counted; read with rv_pp_stamps (probe builds are linked without the export map).
__device__ __attribute__((visibility("default"))) unsigned long long pp_stamp_buf[256 * 2 * 9];
// pp_stamp_ext[workgroup * 4 + i]: 0 = prologue (entry of the k-split loop -> behind its second barrier), 1 = the whole loop, 2 = the whole-panel epilogue, 3 = panels
__device__ __attribute__((visibility("default"))) unsigned long long pp_stamp_ext[3 * 256 * 4];   // [kind: 0 gated, 1 plain, 2 q/k/v][workgroup][i]
#endif

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) { __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0); }
__device__ __forceinline__ float silu(float x) { return rv_silu(x); }

constexpr int PBM = 256, PBN = 256, PBK = 64;
constexpr int UNIT = 128 * 128;      // bytes: 128 rows (or W columns) x 64 k x bf16
constexpr int PP_LDS = 10 * UNIT;    // 160 KiB: two A stages (U0 = A m0 | U3 = A m1) + three W stages (U1 = W n0 | U2 = W n1)

struct PpSrc {
    const char* A;        // uniform bases; the per-lane parts are 32-bit byte offsets (operands are < 4 GiB), which keeps the
    const char* W;        // load addresses in SGPR-base + VGPR-offset form and 8 VGPRs out of the main loop
    unsigned a0[2];       // this wave's two pieces of A unit U0 (rows of the m0 quadrants), k = 0
    unsigned a1[2];       // ... of A unit U3 (m1 quadrants, 64 rows further down)
    unsigned w0[2];       // ... of W unit U1 (n0 quadrants), one per k32 half
    unsigned w1[2];       // ... of W unit U2 (n1 quadrants)
};

// FP8 operands: per-row scales of the quantised activations [M] and per-column scales of the quantised weights [N]; the f32
// accumulator is multiplied by a[m] * w[n] before anything else in the epilogue.  Both null for bf16 operands.
struct PpScale {
    const float* a;
    const float* w;
};
__device__ __forceinline__ f32x4 pp_scaled(const PpScale& sc, int m, int n, f32x4 v) {
    const float sa = sc.a[m];
    const f32x4 sw = *(const f32x4*)(sc.w + n);
    return f32x4{v[0] * (sa * sw[0]), v[1] * (sa * sw[1]), v[2] * (sa * sw[2]), v[3] * (sa * sw[3])};
}

// NF = MFMA fragments per wave along N: 4 -> 256-column panels, 3 -> 192-column panels (n1 quadrant = one fragment)
template <int NF>
__device__ __forceinline__ void pp_sources(PpSrc& s, const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp, int M,
                                           int K, int m0, int n0, int wave, int lane) {
    s.A = (const char*)A;
    s.W = (const char*)Wp;
    // A unit: local row lr in [0,128): group = lr >> 6, tile row = group * 128 + mq * 64 + (lr & 63); piece = 8 rows of 128 B;
    // lane -> row (lane >> 3); LDS slot (lane & 7) of that row holds global chunk slot ^ (row & 7).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int lr = (wave * 2 + i) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (lane >> 3);
        const int trow = (lr >> 6) * 128 + (lr & 63);
        int g0 = m0 + trow, g1 = m0 + trow + 64;
        g0 = g0 < M ? g0 : M - 1;
        g1 = g1 < M ? g1 : M - 1;
        s.a0[i] = (unsigned)(((int64_t)g0 * lda + chunk * 8) * 2);
        s.a1[i] = (unsigned)(((int64_t)g1 * lda + chunk * 8) * 2);
    }
    // W unit n0: piece pc = (wcol * 2 + j) * 2 + ks; this wave issues pc = 2 * wave + ks: wcol = wave >> 1, j = wave & 1.
    // W unit n1, NF = 4: the same two fragments further on; NF = 3: one fragment per wave column -> piece wcol * 2 + ks,
    // one piece per wave: wcol = wave >> 1, ks = wave & 1 (kept in w1[0]).
    const int kfr = K >> 5;
    const int nb0 = ((PP_ABL & 16) ? 0 : (n0 >> 4)) + (wave >> 1) * NF + (wave & 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        s.w0[ks] = (unsigned)(((((int64_t)nb0 * kfr + ks) * 64 + lane) * 8) * 2);
        s.w1[ks] = (unsigned)(((((int64_t)(nb0 + 2) * kfr + ks) * 64 + lane) * 8) * 2);
    }
    if constexpr (NF == 3) {
        const int nb1 = (n0 >> 4) + (wave >> 1) * NF + 2;
        s.w1[0] = (unsigned)(((((int64_t)nb1 * kfr + (wave & 1)) * 64 + lane) * 8) * 2);
    }
}

// Issue unit U (0..3) of the k-tile at element offset k0 into LDS at `dst0` (the unit's base).  Every wave issues 2 pieces of
// 1 KiB (1 piece for the one-fragment n1 unit of NF = 3).
template <int U, int NF>
__device__ __forceinline__ void issue_unit(const PpSrc& s, int k0, char* dst0, int wave) {
    char* dst = dst0 + wave * 2048;
    const char* ab = s.A + (int64_t)k0 * 2;
    const char* wb = s.W + (int64_t)k0 * 32;       // k32 block kb sits at kb * 1024 bytes = k0 * 32
    if constexpr (U == 0) {
        glds16(ab + s.a0[0], dst);
        glds16(ab + s.a0[1], dst + 1024);
    } else if constexpr (U == 3) {
        glds16(ab + s.a1[0], dst);
        glds16(ab + s.a1[1], dst + 1024);
    } else if constexpr (U == 1) {
        glds16(wb + s.w0[0], dst);
        glds16(wb + s.w0[1], dst + 1024);
    } else if constexpr (NF == 4) {
        glds16(wb + s.w1[0], dst);
        glds16(wb + s.w1[1], dst + 1024);
    } else {
        glds16(wb + s.w1[0], dst0 + wave * 1024);   // NF = 3: one piece per wave
    }
}

template <int N_>
__device__ __forceinline__ void wait_vm() {
    static_assert(N_ >= 0 && N_ <= 14, "extend the table");
    if constexpr (N_ == 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
    else if constexpr (N_ == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N_ == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    else if constexpr (N_ == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if constexpr (N_ == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N_ == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// acc += A[tile rows, k-tiles kt0 .. kt0+nks) . W[tile cols, same k]^T.  On entry no LDS access and no load of this
// workgroup is outstanding; the same holds on return (every wave has passed the same number of barriers).
//
// LDS: two A stages (U0 = A m0 | U3 = A m1, 32 KiB each) and THREE W stages (U1 = W n0 | U2 = W n1, 32 KiB each) = all 160 KiB.
// The activation panel is L2 / MALL resident (re-read by every team of the XCD) and stays one tile ahead; the weights come from
// HBM exactly once and are issued TWO tiles ahead, which doubles the latency they may take (6 phases instead of 3: in the
// recursion, where W is cold, the old distance showed as +20 % on the QKV projection against a MALL-warm benchmark loop).
// Round 4: the ACTIVATION units travel 1.5 tiles ahead too.  With 4020 rows (four prefills to a pass) the activation panel is 33 MB: it
// lives in the infinity cache, not in L2, and every XCD re-reads it once per panel group; with U0(c+1) issued in phase 0 of tile c and
// awaited at the end of its phase 3 (0.9 tiles = 1.2 us) those reads stalled the loop - the same launch with lda = 0 (A L2-hot,
// tools/gemm_a_traffic_probe.py) ran 20 % faster (gate/up at 4020 rows: 684 vs 551 us back to back).  A unit's LDS is free as soon as
// its fragments are in registers (U0: read in phase 0, U3: in phase 2), so the two A stages suffice for the longer distance:
// Issue order inside tile c:  ph0 A-U3(c+1) | ph1 W-U1(c+2) | ph2 A-U0(c+2) | ph3 W-U2(c+2).  Waits (counted: a wave's loads retire in
// order; before the barrier that precedes the first read of a unit; n1 / n2 = tile c+1 / c+2 exists, j = loads of unit U2):
//   end of ph0: U2(c) landed     -> may stay in flight: U3(c), U1 U0 U2 (c+1), U3(c+1)          = n1 ? 8 + j : 2
//   end of ph1: U3(c) landed     -> U1 U0 U2 (c+1), U3(c+1), U1(c+2)                            = n2 ? 8 + j : n1 ? 6 + j : 0
//   end of ph3: U0(c+1) landed (and the older U1(c+1))  -> U2(c+1), U3(c+1), U1 U0 U2 (c+2)     = n2 ? 6 + 2j : 2 + j      (n1 only)
// WAR: U3(c+1) overwrites U3(c-1), last read in phase 2 of tile c-1 (>= 3 slots earlier); U0(c+2) overwrites U0(c), read in phase 0 of
// this tile by both groups (slots 0, 1; the issue is in slots 4, 5); a W stage is rewritten >= 1 tile after its last read.
#ifndef PP_ISSUE_MID
#define PP_ISSUE_MID 0
#endif
#ifndef PP_PRIO_MODE
#define PP_PRIO_MODE 1
#endif
#define _PP_C ,
constexpr int PP_A_STAGE = 2 * UNIT, PP_W_BASE = 2 * PP_A_STAGE, PP_W_STAGE = 2 * UNIT;
// F8: the operands are FP8 (e4m3fn) bytes addressed as if they were bf16 matrices of K / 2 columns - a 128-byte staged row is
// then 128 k instead of 64 and a W piece holds, per lane, k = 16 kg .. + 15 of each 64-k half (the bf16 fragment packing of the
// byte pairs).  The lane's two 16-byte groups (ks = 0, 1) of either operand ARE its 32-byte operand of ONE
// v_mfma_f32_16x16x128_f8f6f4: both operands carry the same k permutation and a dot product does not care.  Everything else
// (addresses, LDS layout, waits) is the bf16 kernel at K / 2.
typedef int pp_i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ pp_i32x8 pp_cat(op16x8 lo, op16x8 hi) {
    typedef int i32x4_ __attribute__((ext_vector_type(4)));
    const i32x4_ a = __builtin_bit_cast(i32x4_, lo), b = __builtin_bit_cast(i32x4_, hi);
    return pp_i32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
// ---- round 6: the k-split form of the same tile (16-bit operands, 256-column panels) ---------------------------------------------------
// What the cycle stamps (PP_ABL & 32, tools/pp_stamps.py) showed in the loop above: a k-tile takes ~2560 cycles against 2048 of MFMA issue, and 350 of the
// 510 lost cycles sit in the two slots around a group's phase-0 memory part - its 12 ds_read_b128 are consumed by the compute part right behind the next
// barrier, so their LDS latency (4 waves x 12 KiB through the LDS + the LDS-DMA issues in front of the barrier) is exposed twice per k-tile and SIMD.  With
// the quadrant order all 64 operand registers are live, so nothing can be read ahead.  Here a phase is (M-half, k32-half) instead of a quadrant:
//     phase 0 (m0, k0)   phase 1 (m0, k1)   phase 2 (m1, k0)   phase 3 (m1, k1)      16 MFMAs each: 4 W fragments x 4 A fragments of ONE k32 half
// so a phase needs 4 + 4 operand fragments (32 registers) and the other 32 hold what the NEXT compute part needs: the memory part of phase p reads the
// operands of compute part p + 1 (8 / 4 / 4 / 8 ds_read_b128: 24 per k-tile as before), a whole compute part and two barriers before they are used.
// Every accumulator still sees k ascending (k0 of a tile, then k1): the results are bit-identical to the quadrant loop's.
// LDS: ONE ring of ten 16 KiB units (all 160 KiB), a unit per phase in the order its data is needed:
//     phase 0 issues UA0(c+2) (A rows of the m0 halves) | phase 1 UW1(c+2) (the k1 pieces of the 16 W fragments) | phase 2 UA1(c+2) | phase 3 UW0(c+3)
// and the unit issued in phase t (counted over all tiles) lands in slot t mod 10, i.e. on the unit issued 10 phases earlier.  First / last read (memory
// part of phase, tile): UW0(c) ph3(c-1); UA0(c) ph3(c-1), ph0(c); UW1(c) ph0(c); UA1(c) ph1(c), ph2(c) - every unit is overwritten >= 2 phases (>= 3 barriers)
// after its last read, and travels 7 phases (1.75 k-tiles; UW0: 8) between issue and first read - the quadrant loop gave A 1.5 and W 2 k-tiles.
// Counted waits (a wave's loads retire in order, the issue order is the order of first use; placed as above: group 1 at the end of its memory part, group 0
// at the end of its compute part, before the barrier in front of the first read):  end of ph0: UA1(c) | end of ph2: UW0, UA0 (c+1) | end of ph3: UW1(c+1),
// each with 6 younger units in flight = vmcnt(12) in the steady state; the last three tiles are peeled (e1 / e2 / e3 = tile c+1 / c+2 / c+3 exists).
#ifndef PP_KSPLIT
#define PP_KSPLIT 1      // bit 0: the kernels without a q / k / v epilogue run this loop, bit 1: those too (measured slower inside a pass: 642 -> 672 us)
#endif
#ifndef PP_EARLY
#define PP_EARLY 0       // 1 (probe): the next work item's first loads are issued before the current item's whole-panel epilogue (pp_sk_body) - measured SLOWER
                         // inside a pass of 8040 rows (gate/up 1069 -> 1098 us, o / down 395 -> 430): a wave's vmcnt retires in order, so the epilogue's stores queue
                         // behind the cold loads and the loop's first counted wait then waits for ALL nine units and the stores
#endif
// LDS-DMA as inline asm in the SGPR-base + 32-bit-VGPR-offset form: the builtin makes the compiler (a) rebuild a 64-bit VGPR address per load (v_lshl_add_u64) and
// (b) treat every later wait as "flat pending" - each s_waitcnt it inserts for a ds_read result becomes lgkmcnt(0), which here would wait for the reads of the NEXT
// compute part as well.  The compiler knows nothing about these loads: the counted vmcnt waits and the barriers below are the only ordering.
__device__ __forceinline__ void glds16_sv(const char* sbase, unsigned voff, char* lds_dst) {
    const unsigned l = (unsigned)(uintptr_t)(lptr_t)lds_dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(l) : "memory");   // (M0 is reserved, never allocated: nothing else in a kernel that runs this loop uses it)
}
template <int KIND>   // 0 = UA0, 1 = UA1, 2 = UW0, 3 = UW1 of k-tile kt (absolute) into the 16 KiB unit at dst0
__device__ __forceinline__ void issue_ks(const PpSrc& s, int kt, char* dst0, int wave) {
    char* dst = dst0 + wave * 2048;
    const char* ab = s.A + (int64_t)kt * (PBK * 2);
    const char* wb = s.W + (int64_t)kt * (PBK * 32);
    if constexpr (KIND == 0) {
        glds16_sv(ab, s.a0[0], dst);
        glds16_sv(ab, s.a0[1], dst + 1024);
    } else if constexpr (KIND == 1) {
        glds16_sv(ab, s.a1[0], dst);
        glds16_sv(ab, s.a1[1], dst + 1024);
    } else {   // piece wave * 2 + i = fragment (wave & 1) + 2 i of wave column wave >> 1, k32 half KIND - 2
        glds16_sv(wb, s.w0[KIND - 2], dst);
        glds16_sv(wb, s.w1[KIND - 2], dst + 1024);
    }
}
// The first nine units of a work item in the steady state's order (ring slots 1 .. 9): 18 loads per wave (fewer for items of one or two k-tiles).  A caller may
// issue them BEFORE the previous item's epilogue (PP_EARLY, a probe: nothing reads the LDS then; any wave is past the loop's last barrier) to run the panel's cold
// start - every workgroup of the chip asking HBM for new weights at the same moment, 5 - 8 thousand cycles per item on the stamps - under the epilogue's stores.
// The counted waits stay valid with the epilogue's younger loads / stores in flight (a younger operation only makes vmcnt(n) wait longer) - which is also why it
// does not pay: see PP_EARLY.
__device__ __forceinline__ void ks_issue_prologue(const PpSrc& src, int kt0, int nks, char* smem, int wave) {
    issue_ks<2>(src, kt0, smem + 1 * UNIT, wave);
    issue_ks<0>(src, kt0, smem + 2 * UNIT, wave);
    issue_ks<3>(src, kt0, smem + 3 * UNIT, wave);
    issue_ks<1>(src, kt0, smem + 4 * UNIT, wave);
    if (nks > 1) {
        issue_ks<2>(src, kt0 + 1, smem + 5 * UNIT, wave);
        issue_ks<0>(src, kt0 + 1, smem + 6 * UNIT, wave);
        issue_ks<3>(src, kt0 + 1, smem + 7 * UNIT, wave);
        issue_ks<1>(src, kt0 + 1, smem + 8 * UNIT, wave);
        if (nks > 2) issue_ks<2>(src, kt0 + 2, smem + 9 * UNIT, wave);
    }
}
template <int DUMMY = 0>
__device__ __forceinline__ void pp_mainloop_ks(f32x4 (&acc)[4][8], const PpSrc& src, int kt0, int nks, char* smem, int wave, int lane, [[maybe_unused]] int kind = 1,
                                               bool pre = false) {
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 64 + fr) * 128;                    // A fragment f (rows wr * 64 + f * 16 + fr) of a unit: + f * 2048 + chunk
    const int a_c0 = ((kg ^ (fr & 7)) << 4), a_c1 = (((4 + kg) ^ (fr & 7)) << 4);
    const int w_rd = wc * 4096 + lane * 16;                   // W fragment j of this wave column: piece (wc * 2 + (j & 1)) * 2 + (j >> 1)
    auto slot = [&](int ib, int k) { const int x = ib + k; return smem + (x >= 10 ? x - 10 : x) * UNIT; };
    op16x8 A0[4], A1[4], W0[4], W1[4];
    [[maybe_unused]] uint32_t st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] uint64_t st_prev = 0;
#if PP_ABL & 32
#define PP_STAMP(i) { const uint64_t t_ = __builtin_readcyclecounter(); st_sum[i] += (uint32_t)(t_ - st_prev); st_prev = t_; }
#else
#define PP_STAMP(i)
#endif
#define KS_READ_A(DST, BASE, CH)                                                                            \
    if constexpr (!(PP_ABL & 2)) _Pragma("unroll") for (int f = 0; f < 4; ++f) DST[f] = *(const op16x8*)((BASE) + a_rd + f * 2048 + (CH));
#define KS_READ_W(DST, BASE)                                                                                \
    if constexpr (!(PP_ABL & 2)) _Pragma("unroll") for (int j = 0; j < 4; ++j) DST[j] = *(const op16x8*)((BASE) + w_rd + (j & 1) * 2048 + (j >> 1) * 1024);
#define KS_MFMA(WB, AB, MI0)                                                                                \
    if constexpr (!(PP_ABL & 8)) _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int f = 0; f < 4; ++f) \
        acc[j][(MI0) + f] = rv_mfma16(WB[j], AB[f], acc[j][(MI0) + f]);
#define KS_SYNC_M(WAITN)                                   \
    if constexpr (grp == 1) { WAITN; }                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    if constexpr (!(PP_ABL & 4)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);
#define KS_SYNC_C(WAITN)                                   \
    __builtin_amdgcn_sched_barrier(0);                     \
    if constexpr (grp == 0) { WAITN; }                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    if constexpr (!(PP_ABL & 4)) __builtin_amdgcn_s_barrier(); \
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PP_ABL & 2) {   // (the probe computes on whatever these hold)
        asm volatile("" : "=v"(A0[0]), "=v"(A0[1]), "=v"(A0[2]), "=v"(A0[3]), "=v"(A1[0]), "=v"(A1[1]), "=v"(A1[2]), "=v"(A1[3]));
        asm volatile("" : "=v"(W0[0]), "=v"(W0[1]), "=v"(W0[2]), "=v"(W0[3]), "=v"(W1[0]), "=v"(W1[1]), "=v"(W1[2]), "=v"(W1[3]));
    }

#if PP_ABL & 32
    const uint64_t t_in = __builtin_readcyclecounter();
#endif
    // prologue: what tiles -3 .. -1 of the steady state would have issued (ks_issue_prologue; `pre`: the caller issued it already, under the previous work
    // item's epilogue), then what the memory part of phase 3 of tile -1 does
    if (!pre) ks_issue_prologue(src, kt0, nks, smem, wave);
    if (nks > 2) wait_vm<14>();                  // UW0(0), UA0(0) landed
    else if (nks > 1) wait_vm<12>();
    else wait_vm<4>();
    __builtin_amdgcn_s_barrier();
    KS_READ_A(A0, smem + 2 * UNIT, a_c0)
    KS_READ_W(W0, smem + 1 * UNIT)
    if (nks > 2) wait_vm<12>();                  // UW1(0)
    else if (nks > 1) wait_vm<10>();
    else wait_vm<2>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one slot behind group 0

    auto tile = [&](auto e1_c, auto e2_c, auto e3_c, auto group_c, int c, int ib) {   // ib = 4 c mod 10: the ring slot phase 0 of this tile issues into
        constexpr bool e1 = decltype(e1_c)::value, e2 = decltype(e2_c)::value, e3 = decltype(e3_c)::value;
        constexpr int grp = decltype(group_c)::value;
        const int kt = kt0 + c;
        // ---- phase 0: (m0, k0) on A0 W0; reads (m0, k1) ----
        KS_READ_A(A1, slot(ib, 2), a_c1)
        KS_READ_W(W1, slot(ib, 3))
        if constexpr (e2 && !(PP_ABL & 1)) issue_ks<0>(src, kt + 2, slot(ib, 0), wave);
        KS_SYNC_M(wait_vm<e2 ? 12 : e1 ? 8 : 0>())          // UA1 of this tile
        PP_STAMP(0)
        KS_MFMA(W0, A0, 0)
        KS_SYNC_C(wait_vm<e2 ? 12 : e1 ? 8 : 0>())
        PP_STAMP(1)
        // ---- phase 1: (m0, k1) on A1 W1; reads (m1, k0) ----
        KS_READ_A(A0, slot(ib, 4), a_c0)
        if constexpr (e2 && !(PP_ABL & 1)) issue_ks<3>(src, kt + 2, slot(ib, 1), wave);
        KS_SYNC_M((void)0)
        PP_STAMP(2)
        KS_MFMA(W1, A1, 0)
        KS_SYNC_C((void)0)
        PP_STAMP(3)
        // ---- phase 2: (m1, k0) on A0 W0; reads (m1, k1) ----
        KS_READ_A(A1, slot(ib, 4), a_c1)
        if constexpr (e2 && !(PP_ABL & 1)) issue_ks<1>(src, kt + 2, slot(ib, 2), wave);
        KS_SYNC_M(if constexpr (e1) wait_vm<e2 ? 12 : 4>())  // UW0, UA0 of the next tile
        PP_STAMP(4)
        KS_MFMA(W0, A0, 4)
        KS_SYNC_C(if constexpr (e1) wait_vm<e2 ? 12 : 4>())
        PP_STAMP(5)
        // ---- phase 3: (m1, k1) on A1 W1; reads (m0, k0) of the next tile ----
        if constexpr (e1) {
            KS_READ_A(A0, slot(ib, 6), a_c0)
            KS_READ_W(W0, slot(ib, 5))
        }
        if constexpr (e3 && !(PP_ABL & 1)) issue_ks<2>(src, kt + 3, slot(ib, 3), wave);
        KS_SYNC_M(if constexpr (e1) wait_vm<e3 ? 12 : e2 ? 10 : 2>())   // UW1 of the next tile
        PP_STAMP(6)
        KS_MFMA(W1, A1, 4)
        KS_SYNC_C(if constexpr (e1) wait_vm<e3 ? 12 : e2 ? 10 : 2>())
        PP_STAMP(7)
    };
    auto run = [&](auto group_c) {
        int c = 0, ib = 0;
        auto next = [&]() { ++c; ib = ib >= 6 ? ib - 6 : ib + 4; };
        while (c + 3 < nks) { tile(std::true_type{}, std::true_type{}, std::true_type{}, group_c, c, ib); next(); }
        if (c + 2 < nks) { tile(std::true_type{}, std::true_type{}, std::false_type{}, group_c, c, ib); next(); }
        if (c + 1 < nks) { tile(std::true_type{}, std::false_type{}, std::false_type{}, group_c, c, ib); next(); }
        tile(std::false_type{}, std::false_type{}, std::false_type{}, group_c, c, ib);
    };
#if PP_ABL & 32
    st_prev = __builtin_readcyclecounter();
    const uint64_t t_pro = st_prev;
#endif
    if (wr == 1) __builtin_amdgcn_s_setprio(1);   // (PP_PRIO_MODE 1 of the quadrant loop: static priority for the second-dispatched half)
    if (wr == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    if (wr == 1) __builtin_amdgcn_s_setprio(0);
#if PP_ABL & 32
    if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 8; ++i) atomicAdd(&pp_stamp_buf[(blockIdx.x * 2 + wr) * 9 + i], (unsigned long long)st_sum[i]);
        atomicAdd(&pp_stamp_buf[(blockIdx.x * 2 + wr) * 9 + 8], (unsigned long long)nks);
        if (wr == 0) {
            atomicAdd(&pp_stamp_ext[(kind * 256 + blockIdx.x) * 4 + 0], (unsigned long long)(t_pro - t_in));
        }
    }
#endif
#undef KS_READ_A
#undef KS_READ_W
#undef KS_MFMA
#undef KS_SYNC_M
#undef KS_SYNC_C
#undef PP_STAMP
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
}

template <int NF, int F8 = 0, int KS = 1>
__device__ __forceinline__ void pp_mainloop(f32x4 (&acc)[NF][8], const PpSrc& src, int kt0, int nks, char* smem, int wave, int lane, [[maybe_unused]] int kind = 1,
                                            [[maybe_unused]] bool pre = false) {
    if constexpr (NF == 4 && !F8 && KS && PP_KSPLIT) {   // the k-split loop (above): same operands, same LDS budget, same results
        pp_mainloop_ks(acc, src, kt0, nks, smem, wave, lane, kind, pre);
        return;
    } else {
    constexpr int NJ1 = NF - 2;      // fragments of the n1 quadrant = loads per wave of unit U2
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, kg = lane >> 4;
    // A fragment (f, ks) of unit U0 / U3: row lr = wr * 64 + f * 16 + fr at lr * 128 + (((ks * 4 + kg) ^ (lr & 7)) << 4)
    const int a_rd = (wr * 64 + fr) * 128;
    const int a_c0 = ((kg ^ (fr & 7)) << 4), a_c1 = (((4 + kg) ^ (fr & 7)) << 4);
    // W fragment (j, ks) of unit U1 / U2: piece (wc * 2 + j) * 2 + ks
    const int w_rd = wc * 4096 + lane * 16;
    const int w_rd1 = NF == 4 ? w_rd : wc * 2048 + lane * 16;   // unit U2: pieces (wc * NJ1 + j) * 2 + ks

    // prologue: what tiles "-2" and "-1" of the steady state would have issued, in its order: U1 U0 U2 (0), U3(0), U1 U0 U2 (1)
    issue_unit<1, NF>(src, kt0 * PBK, smem + PP_W_BASE, wave);
    issue_unit<0, NF>(src, kt0 * PBK, smem, wave);
    issue_unit<2, NF>(src, kt0 * PBK, smem + PP_W_BASE + UNIT, wave);
    issue_unit<3, NF>(src, kt0 * PBK, smem + UNIT, wave);
    if (nks > 1) {
        issue_unit<1, NF>(src, (kt0 + 1) * PBK, smem + PP_W_BASE + PP_W_STAGE, wave);
        issue_unit<0, NF>(src, (kt0 + 1) * PBK, smem + PP_A_STAGE, wave);
        issue_unit<2, NF>(src, (kt0 + 1) * PBK, smem + PP_W_BASE + PP_W_STAGE + UNIT, wave);
        wait_vm<6 + 2 * NJ1>();                  // U1(0), U0(0) landed
    } else {
        wait_vm<2 + NJ1>();
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // group 1 runs one slot behind group 0

    op16x8 af[4][2], b0[2][2], b1[2][2];
    [[maybe_unused]] uint32_t st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] uint64_t st_prev = 0;
#if PP_ABL & 32
#define PP_STAMP(i) { const uint64_t t_ = __builtin_readcyclecounter(); st_sum[i] += (uint32_t)(t_ - st_prev); st_prev = t_; }
#else
#define PP_STAMP(i)
#endif
    if constexpr (PP_ABL & 2) {   // (the probe computes on whatever these hold)
        asm volatile("" : "=v"(af[0][0]), "=v"(af[0][1]), "=v"(af[1][0]), "=v"(af[1][1]), "=v"(af[2][0]), "=v"(af[2][1]), "=v"(af[3][0]), "=v"(af[3][1]));
        asm volatile("" : "=v"(b0[0][0]), "=v"(b0[0][1]), "=v"(b0[1][0]), "=v"(b0[1][1]), "=v"(b1[0][0]), "=v"(b1[0][1]), "=v"(b1[1][0]), "=v"(b1[1][1]));
    }
    // One k-tile.  m1 / m2 (tile c+1 / c+2 exists: their units are issued here) and the wave's M-group are compile-time
    // constants: the steady-state loop carries no branches - the last two tiles are peeled, the two groups run separate copies.
    auto tile = [&](auto m1_c, auto m2_c, auto group_c, int c, int ws) {   // ws = c % 3
        constexpr bool m1 = decltype(m1_c)::value, m2 = decltype(m2_c)::value;
        constexpr int grp = decltype(group_c)::value;
        char* acur = smem + (c & 1) * PP_A_STAGE;
        char* anxt = smem + ((c + 1) & 1) * PP_A_STAGE;
        char* wcur = smem + PP_W_BASE + ws * PP_W_STAGE;
        char* wnn = smem + PP_W_BASE + (ws == 0 ? 2 : ws - 1) * PP_W_STAGE;   // (c + 2) % 3
        const int k1 = (kt0 + c + 1) * PBK, k2 = k1 + PBK;

        // PART 0 / 1: the two halves of a phase's MFMAs (bf16: k32 half ks = PART of every fragment pair, i.e. the order every accumulator
        // sees is unchanged; FP8: fragment columns j < NJ / 2 | the rest).  PP_ISSUE_MID: the phase's LDS-DMA unit is issued BETWEEN the halves.
#define PP_MFMA_PART(B, NI, MI0, NJ, PART)                                                                          \
    do {                                                                                                            \
        if constexpr (F8) {                                                                                         \
            _Pragma("unroll") for (int j = (PART) * ((NJ) / 2); j < ((PART) ? (NJ) : (NJ) / 2); ++j)                 \
                _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                        \
                acc[(NI) + j][(MI0) + f] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                        \
                    pp_cat(B[j][0], B[j][1]), pp_cat(af[f][0], af[f][1]), acc[(NI) + j][(MI0) + f], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f); \
        } else {                                                                                                    \
            _Pragma("unroll") for (int j = 0; j < (NJ); ++j)                                                         \
                _Pragma("unroll") for (int f = 0; f < 4; ++f) acc[(NI) + j][(MI0) + f] =                             \
                    rv_mfma16(B[j][(PART)], af[f][(PART)], acc[(NI) + j][(MI0) + f]); \
        }                                                                                                           \
    } while (0)
#define PP_MFMA(B, NI, MI0, NJ, MID_ISSUE)                                                                          \
    if constexpr (!(PP_ABL & 8)) do {                                                                               \
        if constexpr (PP_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(1);                                             \
        PP_MFMA_PART(B, NI, MI0, NJ, 0);                                                                            \
        if constexpr (PP_ISSUE_MID) {                                                                               \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
            MID_ISSUE;                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                      \
        }                                                                                                           \
        PP_MFMA_PART(B, NI, MI0, NJ, 1);                                                                            \
        if constexpr (PP_PRIO_MODE == 0) __builtin_amdgcn_s_setprio(0);                                             \
    } while (0)
#define PP_READ_A(BASE)                                                                                             \
    if constexpr (!(PP_ABL & 2)) _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                                 \
        af[f][0] = *(const op16x8*)((BASE) + a_rd + f * 2048 + a_c0);                                               \
        af[f][1] = *(const op16x8*)((BASE) + a_rd + f * 2048 + a_c1);                                               \
    }
#define PP_READ_W(B, BASE, RD, NJ)                                                                                  \
    if constexpr (!(PP_ABL & 2)) _Pragma("unroll") for (int j = 0; j < (NJ); ++j) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                \
        B[j][ks] = *(const op16x8*)((BASE) + (RD) + j * 2048 + ks * 1024);
        // The memory part of a phase ends at its first barrier, the compute part at its second.  Group 1 waits for its
        // loads at the end of its memory part, group 0 at the end of its compute part: the same slot.
#define PP_SYNC_M(WAITN)                                   \
    if constexpr (grp == 1) { WAITN; }                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    if constexpr (!(PP_ABL & 4)) __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);
#define PP_SYNC_C(WAITN)                                   \
    if constexpr (grp == 0) { WAITN; }                     \
    __builtin_amdgcn_sched_barrier(0);                     \
    if constexpr (!(PP_ABL & 4)) __builtin_amdgcn_s_barrier();                          \
    __builtin_amdgcn_sched_barrier(0);

        // PP_ISSUE_MID = 1: the unit of a phase is issued in the MIDDLE of its compute part (between the two halves of its MFMAs, which run on
        // while the wave issues) instead of in its memory part.  Group 1 waits at the end of its memory part, i.e. BEFORE that issue: its counts
        // exclude the unit (2 loads; NJ1 for U2); group 0 waits behind its compute part: unchanged.
        constexpr int MID = PP_ISSUE_MID;
        // ---- phase 0: quadrant (m0, n0) ----
        PP_READ_W(b0, wcur, w_rd, 2)
        PP_READ_A(acur)
        if constexpr (!MID && m1 && !(PP_ABL & 1)) issue_unit<3, NF>(src, k1, anxt + UNIT, wave);
        PP_SYNC_M(wait_vm<m1 ? 8 + NJ1 - 2 * MID : 2>())          // U2 of this tile
        PP_STAMP(0)
        PP_MFMA(b0, 0, 0, 2, if constexpr (m1) issue_unit<3 _PP_C NF>(src, k1, anxt + UNIT, wave));
        PP_SYNC_C(wait_vm<m1 ? 8 + NJ1 : 2>())
        PP_STAMP(1)
        // ---- phase 1: quadrant (m0, n1) ----
        PP_READ_W(b1, wcur + UNIT, w_rd1, NJ1)
        if constexpr (!MID && m2 && !(PP_ABL & 1)) issue_unit<1, NF>(src, k2, wnn, wave);
        PP_SYNC_M(wait_vm<m2 ? 8 + NJ1 - 2 * MID : m1 ? 6 + NJ1 : 0>())     // U3 of this tile
        PP_STAMP(2)
        PP_MFMA(b1, 2, 0, NJ1, if constexpr (m2) issue_unit<1 _PP_C NF>(src, k2, wnn, wave));
        PP_SYNC_C(wait_vm<m2 ? 8 + NJ1 : m1 ? 6 + NJ1 : 0>())
        PP_STAMP(3)
        // ---- phase 2: quadrant (m1, n1) ----
        PP_READ_A(acur + UNIT)
        if constexpr (!MID && m2 && !(PP_ABL & 1)) issue_unit<0, NF>(src, k2, acur, wave);   // U0(c+2) into the slot U0(c) left in phase 0
        PP_SYNC_M((void)0)
        PP_STAMP(4)
        PP_MFMA(b1, 2, 4, NJ1, if constexpr (m2) issue_unit<0 _PP_C NF>(src, k2, acur, wave));
        PP_SYNC_C((void)0)
        PP_STAMP(5)
        // ---- phase 3: quadrant (m1, n0) ----
        if constexpr (!MID && m2 && !(PP_ABL & 1)) issue_unit<2, NF>(src, k2, wnn + UNIT, wave);
        PP_SYNC_M(if constexpr (m1) wait_vm<m2 ? 6 + 2 * NJ1 - NJ1 * MID : 2 + NJ1>())  // U0 (and the older U1) of the next tile
        PP_STAMP(6)
        PP_MFMA(b0, 0, 4, 2, if constexpr (m2) issue_unit<2 _PP_C NF>(src, k2, wnn + UNIT, wave));
        PP_SYNC_C(if constexpr (m1) wait_vm<m2 ? 6 + 2 * NJ1 : 2 + NJ1>())
        PP_STAMP(7)
    };
    auto run = [&](auto group_c) {
        int c = 0, ws = 0;
        for (; c + 2 < nks; ++c) {
            tile(std::true_type{}, std::true_type{}, group_c, c, ws);
            ws = ws == 2 ? 0 : ws + 1;
        }
        if (c + 1 < nks) {
            tile(std::true_type{}, std::false_type{}, group_c, c, ws);
            ws = ws == 2 ? 0 : ws + 1;
            ++c;
        }
        tile(std::false_type{}, std::false_type{}, group_c, c, ws);
    };
    // PP_PRIO_MODE (compile-time probe): 0 = priority 1 around every MFMA block (rounds 2 - 4); 1 (default) = STATIC priority 1 for the
    // second-dispatched half (waves 4 - 7, the arbitration loser by age) and no per-segment flips; 2 = no priority changes at all.
    // Measured at 4020 rows, alternating builds on one box: gate/up 610.3 / 611.4 (mode 0) vs 606.8 / 605.5 (1) vs 606.8 (2) us,
    // o / down 218.1 / 218.6 vs 217.2 / 216.2, QKV 380.1 / 380.8 vs 380.4 / 378.5: priorities are worth half a percent at most.
    if constexpr (PP_PRIO_MODE == 1) { if (wr == 1) __builtin_amdgcn_s_setprio(1); }
#if PP_ABL & 32
    st_prev = __builtin_readcyclecounter();
#endif
    if (wr == 0) run(std::integral_constant<int, 0>{});   // the two groups run separate copies of the loop
    else run(std::integral_constant<int, 1>{});
#if PP_ABL & 32
    if ((wave & 3) == 0 && lane == 0 && blockIdx.x < 256) {
        for (int i = 0; i < 8; ++i) atomicAdd(&pp_stamp_buf[(blockIdx.x * 2 + wr) * 9 + i], (unsigned long long)st_sum[i]);
        atomicAdd(&pp_stamp_buf[(blockIdx.x * 2 + wr) * 9 + 8], (unsigned long long)nks);
    }
#endif
    if constexpr (PP_PRIO_MODE == 1) { if (wr == 1) __builtin_amdgcn_s_setprio(0); }
#undef PP_MFMA
#undef PP_MFMA_PART
#undef PP_READ_A
#undef PP_READ_W
#undef PP_SYNC_M
#undef PP_SYNC_C
#undef PP_STAMP
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance group 1's extra barrier
    }
}

// ---- the FOUR-wave form of the 256 x 256 x 64 tile (bf16 operands, 256-column panels) --------------------------------------------
// One wave per SIMD, 2 (M) x 2 (N) waves of 128 x 128 outputs: 64 accumulator fragments = 256 registers, pinned to the AGPR half of the
// unified file, plus two operand fragment sets (k32 halves, 2 x 64 VGPRs).  Per k-tile the workgroup then reads 128 KiB of fragments
// from LDS (the eight-wave form above: 192 KiB) and there is no second wave on a SIMD to keep in lock-step: every wave interleaves its
// own ds_reads and LDS-DMA issues between its MFMAs (two MFMAs, one memory instruction, pinned with sched_barrier).  Same LDS budget
// and latency distances as above: two A stages (one tile ahead) + three W stages (two tiles ahead) = 160 KiB; ONE barrier per k-tile,
// in the middle: behind it the first half's operands (k32 half 0 of the next tile) are read and the stages the tile has finished with
// (its A stage and W stage: all of their fragments are in registers by then) are refilled.  The MFMAs are inline asm: the register
// allocator, left to choose, trades accumulators and operand fragments between the two halves of the file inside the loop.
// tools/micro/gemm4w.hip is the stand-alone probe this came from (4096^3 sustained: 98 us against 102 us for the eight-wave form).
struct Pp4Src {
    const char* A;
    const char* W;
    unsigned a[8];   // this wave's 8 pieces (8 rows x 128 B) of the A tile, k = 0
    unsigned w[8];   // this wave's 8 pieces (fragment nfrag = wave * 4 + (i >> 1), k32 half i & 1) of the W tile
};
__device__ __forceinline__ void pp4_sources(Pp4Src& s, const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp, int M, int K,
                                            int m0, int n0, int wave, int lane) {
    s.A = (const char*)A;
    s.W = (const char*)Wp;
    const int kfr = K >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int g = m0 + (wave * 8 + i) * 8 + (lane >> 3);   // LDS slot (lane & 7) of the row holds global chunk slot ^ (row & 7)
        g = g < M ? g : M - 1;
        s.a[i] = (unsigned)(((int64_t)g * lda + (((lane & 7) ^ (lane >> 3)) * 8)) * 2);
        const int nb = (n0 >> 4) + wave * 4 + (i >> 1);
        s.w[i] = (unsigned)(((((int64_t)nb * kfr + (i & 1)) * 64 + lane) * 8) * 2);
    }
}
constexpr int PP4_TILE = 256 * 64 * 2;   // bytes of one operand tile: stages A0 A1 W0 W1 W2
template <int DUMMY = 0>
__device__ __forceinline__ void pp4_mainloop(f32x4 (&acc)[8][8], const Pp4Src& src, int kt0, int nks, char* smem, int wave, int lane) {
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;
    const int a_rd = (wr * 128 + fr) * 128;
    const int a_c[2] = {((kg ^ (fr & 7)) << 4), (((4 + kg) ^ (fr & 7)) << 4)};
    const int w_rd = wc * 16384 + lane * 16;
    op16x8 fa[2][8], fw[2][8];
#define P4_A_STAGE(c) (smem + ((c) & 1) * PP4_TILE)
#define P4_W_STAGE(ws) (smem + (2 + (ws)) * PP4_TILE)
#define P4_ISSUE_A(KT, ST, i) glds16(src.A + (int64_t)(KT) * 128 + src.a[i], (ST) + (wave * 8 + (i)) * 1024)
#define P4_ISSUE_W(KT, ST, i) glds16(src.W + (int64_t)(KT) * 2048 + src.w[i], (ST) + (wave * 8 + (i)) * 1024)
#define P4_READ_A(B, ST, KS, f) fa[B][f] = *(const op16x8*)((ST) + a_rd + (f) * 2048 + a_c[KS])
#define P4_READ_W(B, ST, KS, j) fw[B][j] = *(const op16x8*)((ST) + w_rd + (j) * 2048 + (KS) * 1024)
#define P4_MFMA1(B, j, f) asm volatile(RV_MFMA16_ASM " %0, %1, %2, %0" : "+a"(acc[j][f]) : "v"(fw[B][j]), "v"(fa[B][f]));
#define P4_MFMA2(B, j, f) P4_MFMA1(B, j, f) P4_MFMA1(B, j, (f) + 1)
#define P4_PIN __builtin_amdgcn_sched_barrier(0);
    const int last = kt0 + nks - 1;
    auto kclamp = [&](int k) { return k < last ? k : last; };   // past the range: the last tile again, into a stage nobody reads
    // prologue: A(0), W(0), W(1), A(1), W(2) in that order (a wave's loads retire in order; see the counted wait below)
#pragma unroll
    for (int i = 0; i < 8; ++i) P4_ISSUE_A(kt0, P4_A_STAGE(0), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) P4_ISSUE_W(kt0, P4_W_STAGE(0), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) P4_ISSUE_W(kclamp(kt0 + 1), P4_W_STAGE(1), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) P4_ISSUE_A(kclamp(kt0 + 1), P4_A_STAGE(1), i);
#pragma unroll
    for (int i = 0; i < 8; ++i) P4_ISSUE_W(kclamp(kt0 + 2), P4_W_STAGE(2), i);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 8; ++f) { P4_READ_A(0, P4_A_STAGE(0), 0, f); P4_READ_W(0, P4_W_STAGE(0), 0, f); }
    int ws = 0;
    for (int c = 0; c < nks; ++c) {
        char* a_cur = P4_A_STAGE(c);
        char* a_nxt = P4_A_STAGE(c + 1);
        char* w_cur = P4_W_STAGE(ws);
        const int ws1 = ws == 2 ? 0 : ws + 1;
        char* w_nxt = P4_W_STAGE(ws1);
        const int ka = kclamp(kt0 + c + 2), kw = kclamp(kt0 + c + 3);
        P4_PIN
        // first half: MFMAs on (c, k32 half 0) while the fragments of (c, half 1) arrive
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            P4_MFMA2(0, j, 0) P4_PIN
            P4_READ_A(1, a_cur, 1, j); P4_MFMA2(0, j, 2) P4_PIN
            P4_READ_W(1, w_cur, 1, j); P4_MFMA2(0, j, 4) P4_PIN
            P4_MFMA2(0, j, 6) P4_PIN
        }
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // A(c + 1), W(c + 1) landed; W(c + 2) may stay in flight
        __builtin_amdgcn_s_barrier();
        P4_PIN
        // second half: MFMAs on (c, half 1); fragments of (c + 1, half 0) arrive; A(c + 2) -> this tile's A stage, W(c + 3) -> its W stage
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            P4_MFMA2(1, j, 0) P4_PIN
            if (j < 4) { P4_READ_A(0, a_nxt, 0, 2 * j); P4_ISSUE_A(ka, a_cur, 2 * j); } else { P4_READ_W(0, w_nxt, 0, 2 * (j - 4)); P4_ISSUE_W(kw, w_cur, 2 * (j - 4)); }
            P4_MFMA2(1, j, 2) P4_PIN
            if (j < 4) { P4_READ_A(0, a_nxt, 0, 2 * j + 1); } else { P4_READ_W(0, w_nxt, 0, 2 * (j - 4) + 1); }
            P4_MFMA2(1, j, 4) P4_PIN
            if (j < 4) { P4_ISSUE_A(ka, a_cur, 2 * j + 1); } else { P4_ISSUE_W(kw, w_cur, 2 * (j - 4) + 1); }
            P4_MFMA2(1, j, 6) P4_PIN
        }
        ws = ws1;
    }
    // drain: the clamped loads of the last tiles and the fragment reads behind the last barrier target LDS the next user rewrites;
    // the MFMA results are read by ordinary VALU code from here on (the compiler does not see the asm as an MFMA: cover the hazard)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_nop 15\n s_nop 15" ::: "memory");
    __builtin_amdgcn_s_barrier();
#undef P4_A_STAGE
#undef P4_W_STAGE
#undef P4_ISSUE_A
#undef P4_ISSUE_W
#undef P4_READ_A
#undef P4_READ_W
#undef P4_MFMA1
#undef P4_MFMA2
#undef P4_PIN
}

// One row's NFR fragments (columns slab0 + ni * 16 + kg * 4 .. + 3 of head `head` in section `sec`; slab0 = offset of the wave's slab in the head).
template <int NFR>
__device__ __forceinline__ void pp_rope_row_store(const f32x4 (&v)[NFR], const QkvRope& qr, const QkvRow& row, int sec, int head, int slab0, int kg) {
    const int p0 = slab0 + kg * 4;
    if constexpr ((RS_PROBE_K_ & 256) != 0) { if (sec == 2) return; }      // (timing probes, tools/qkv_store_probe.sh: no V^T / K / Q stores)
    if constexpr ((RS_PROBE_K_ & 512) != 0) { if (sec == 1) return; }
    if constexpr ((RS_PROBE_K_ & 1024) != 0) { if (sec == 0) return; }
    if (sec == 2) {
        for (int bb = row.b0; bb < row.b1; ++bb) {
            op16_t* dst = (op16_t*)qr.vtc + ((int64_t)bb * qr.H + head) * 128 * qr.Smax + rv_vt_index(p0, row.pos);
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni)
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(ni * 16 + e) * 8] = f32_to_op16(v[ni][e]);
        }
        return;
    }
    u32x2 o[NFR];
#pragma unroll
    for (int ni = 0; ni < NFR; ++ni) {
        const f32x4 t = *(const f32x4*)(row.cs + (p0 + ni * 16));   // pairs (p >> 1) of p = p0 + ni * 16: (c0, s0, c1, s1) at cs + (p >> 1) * 2
        const float a0 = __fmaf_rn(v[ni][0], t[0], -__fmul_rn(v[ni][1], t[1])), b0 = __fmaf_rn(v[ni][1], t[0], __fmul_rn(v[ni][0], t[1]));
        const float a1 = __fmaf_rn(v[ni][2], t[2], -__fmul_rn(v[ni][3], t[3])), b1 = __fmaf_rn(v[ni][3], t[2], __fmul_rn(v[ni][2], t[3]));
        o[ni] = pack_op16x4(f32x4{a0, b0, a1, b1});
    }
    if (sec == 0) {
        op16_t* dst = (op16_t*)qr.q16 + (int64_t)row.mrow * (qr.H * 128) + head * 128 + p0;
#pragma unroll
        for (int ni = 0; ni < NFR; ++ni) *(u32x2*)(dst + ni * 16) = o[ni];
    } else {
        for (int bb = row.b0; bb < row.b1; ++bb) {
            op16_t* dst = (op16_t*)qr.kc + (((int64_t)bb * qr.H + head) * qr.Smax + row.pos) * 128 + p0;
#pragma unroll
            for (int ni = 0; ni < NFR; ++ni) *(u32x2*)(dst + ni * 16) = o[ni];
        }
    }
}
// The fused q / k / v epilogue of a WHOLE 256 x 256 panel, staged through LDS (round 6; the "LDS-staged V^T" of VERDICT r4 / r5).  pp_rope_row_store writes what a lane
// holds - 4 consecutive columns of one row per fragment: 8-byte pieces of Q / K rows (32 contiguous bytes per row and instruction) and, for V, sixteen 2-byte stores per row
// 16 bytes apart (the cache keeps V transposed in blocks of 8 positions).  Probes with the stores compiled out (tools/qkv_store_probe.sh, 4 x 1005 rows): 361 us per launch,
// 335 without the V^T stores, 341 without K, 348 without Q, 302 without any - 59 us of a 361 us launch were its stores.  Here a wave first parks its 128 x 64 sub-tile (one
// head slab of one section; RoPE applied, packed to 16 bits: the same values) in a private LDS tile - the operand stages are free behind the main loop's last barrier - with
// the rows' (position, cache rows) next to it, and then writes it out in the order memory wants:
//   Q / K   a lane takes 8 consecutive columns (16 bytes) of a row: a store instruction covers 8 whole 128-byte row slabs;
//   V^T     a lane takes ONE column d and the 8 rows of an aligned group of 8 positions of one sequence: 16 bytes, and the wave's 64 columns are 1 KiB contiguous
//           (element (d, pos) at ((pos >> 3) * 128 + d) * 8 + (pos & 7)); rows whose group of 8 is cut by the tile's edge or a sequence's end go out as 2-byte stores.
// Every byte lands where pp_rope_row_store puts it: caches and Q bit-identical (tests: the batched / shared-prefix / ragged prefills against their separate forms).
template <int F8>
__device__ __forceinline__ void pp_epilogue_rope_lds(const f32x4 (&acc)[4][8], int M, int m0, int n0, int wave, int lane, const QkvRope& qr, const PpScale& sc,
                                                     char* smem) {
    constexpr int RS = 144;                                 // bytes per tile row: 128 + 16 (conflict-free 8-byte writes, 16-byte and 2-byte-column reads)
    constexpr int TILE = 128 * RS, WREG = TILE + 128 * 16;  // + one int4 per row: (pos or -1, b0, b1, mrow)
    static_assert(8 * WREG <= PP_LDS, "eight private wave regions");
    char* my = smem + wave * WREG;
    int4* info = (int4*)(my + TILE);
    const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, kg = lane >> 4;
    const int D = qr.H * 128, nb = n0 + wc * 64;
    const int sec = __builtin_amdgcn_readfirstlane(nb / D), hd0 = nb - sec * D, head = hd0 >> 7, slab0 = hd0 & 127;
    const int mbase = m0 + wr * 128;
    // ---- phase A: rotate / pack what the lane holds into the tile; lanes kg == 0 record their row ----
    // rows first (8 per lane: two integer divisions each), then column block by column block with the NEXT block's (cos, sin) loads in flight: a row's coefficients sit
    // behind its position, and fetched row block by row block they were 8 dependent L2 round trips per epilogue
    // (the 128 rows' records are computed ONCE - two rows per lane, not eight per lane four times over - and read back: the epilogue is bound by its instruction count,
    // and a row costs ~100 instructions)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int r = lane + 64 * hh, m = mbase + r;
        const QkvRow row = qkv_rope_row(qr, m < M ? m : M - 1);
        info[r] = (m < M && row.b1 > row.b0) ? int4{row.pos, row.b0, row.b1, row.mrow} : int4{-1, 0, 0, 0};
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (a dead row - past M - reads a valid table row and writes its tile row like any other: phase B never looks at it.  Its accumulators are those of row M - 1,
    // whose operand rows the main loop's clamped loads gave it: they can raise the saturation report below only together with that live row)
    int posv[8];
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int p_ = ((const int*)(info + mi * 16 + fr))[0];
        posv[mi] = p_ >= 0 ? p_ : qr.cs_pos0;
    }
    uint32_t sat = 0;                                   // one saturation report for the wave's whole sub-tile
    if (sec == 2) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                f32x4 v = acc[ni][mi];
                if constexpr (F8) v = pp_scaled(sc, mbase + mi * 16 + fr, nb + ni * 16 + kg * 4, v);
                *(u32x2*)(my + (mi * 16 + fr) * RS + (ni * 16 + kg * 4) * 2) = u32x2{pack_op16x2_m(v[0], v[1], sat), pack_op16x2_m(v[2], v[3], sat)};
            }
    } else {
        const float* cs0 = qr.cs + (slab0 + kg * 4) - (int64_t)qr.cs_pos0 * 128;      // (row.cs = qr.cs + (pos - cs_pos0) * 128 for prefill rows)
        f32x4 tc[8], tn[8];
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) tc[mi] = *(const f32x4*)(cs0 + (int64_t)posv[mi] * 128);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni < 3) {
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) tn[mi] = *(const f32x4*)(cs0 + (int64_t)posv[mi] * 128 + (ni + 1) * 16);
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                f32x4 v = acc[ni][mi];
                if constexpr (F8) v = pp_scaled(sc, mbase + mi * 16 + fr, nb + ni * 16 + kg * 4, v);
                const f32x4 t = tc[mi];
                const float a0 = __fmaf_rn(v[0], t[0], -__fmul_rn(v[1], t[1])), b0 = __fmaf_rn(v[1], t[0], __fmul_rn(v[0], t[1]));
                const float a1 = __fmaf_rn(v[2], t[2], -__fmul_rn(v[3], t[3])), b1 = __fmaf_rn(v[3], t[2], __fmul_rn(v[2], t[3]));
                *(u32x2*)(my + (mi * 16 + fr) * RS + (ni * 16 + kg * 4) * 2) = u32x2{pack_op16x2_m(a0, b0, sat), pack_op16x2_m(a1, b1, sat)};
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) tc[mi] = tn[mi];
        }
    }
    rv_note_saturation(sat);
    if constexpr ((RS_PROBE_K_ & 2048) != 0) return;         // (timing probe: phase A only - rows, coefficients, rotation, LDS writes; nothing leaves the CU)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (wave-private region: the wave's own LDS writes are complete before its lanes read each other's)
    // ---- phase B ----
    if (sec < 2) {
        const int rl = lane >> 3, ch = lane & 7;
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int r = it * 8 + rl;
            const int4 ri = info[r];
            if (ri.x < 0) continue;
            const u32x4 val = *(const u32x4*)(my + r * RS + ch * 16);
            if (sec == 0) {
                *(u32x4*)((op16_t*)qr.q16 + (int64_t)ri.w * D + head * 128 + slab0 + ch * 8) = val;
            } else {
                for (int bb = ri.y; bb < ri.z; ++bb)
                    *(u32x4*)((op16_t*)qr.kc + (((int64_t)bb * qr.H + head) * qr.Smax + ri.x) * 128 + slab0 + ch * 8) = val;
            }
        }
        return;
    }
    const int64_t vrow = (int64_t)128 * qr.Smax;            // elements per (cache row, head) of V^T
    for (int r = 0; r < 128;) {                             // (wave-uniform walk over the tile's rows)
        const int4 a = info[r];
        const int pos = __builtin_amdgcn_readfirstlane(a.x), b0 = __builtin_amdgcn_readfirstlane(a.y), b1 = __builtin_amdgcn_readfirstlane(a.z);
        if (pos < 0) { ++r; continue; }
        bool full = (pos & 7) == 0 && r + 7 < 128;
        if (full) {
            const int4 z = info[r + 7];
            full = __builtin_amdgcn_readfirstlane(z.x) == pos + 7 && __builtin_amdgcn_readfirstlane(z.y) == b0 && __builtin_amdgcn_readfirstlane(z.z) == b1;
        }
        const char* col = my + r * RS + lane * 2;
        if (full) {     // rows r .. r + 7 = positions pos .. pos + 7 of one sequence: this lane's column as one 16-byte piece
            u32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w[j] = (uint32_t)*(const uint16_t*)(col + (2 * j) * RS) | ((uint32_t)*(const uint16_t*)(col + (2 * j + 1) * RS) << 16);
            for (int bb = b0; bb < b1; ++bb)
                *(u32x4*)((op16_t*)qr.vtc + ((int64_t)bb * qr.H + head) * vrow + rv_vt_index(slab0 + lane, pos)) = w;
            r += 8;
        } else {
            const op16_t val = *(const op16_t*)col;
            for (int bb = b0; bb < b1; ++bb) ((op16_t*)qr.vtc)[((int64_t)bb * qr.H + head) * vrow + rv_vt_index(slab0 + lane, pos)] = val;
            ++r;
        }
    }
}

// Epilogue: lane owns row m = .. + fr, columns n = .. + kg * 4 .. + 3 of every 16 x 16 fragment.
template <int OUT_BF16, int ACT, int ROPE, int NF, int F8 = 0>
__device__ __forceinline__ void pp_epilogue(const f32x4 (&acc_in)[NF][8], const float* __restrict__ bias, const float* res, int64_t ldr,
                                            void* Cv, int64_t ldc, int M, int m0, int n0, int wave, int lane, const QkvRope& qr,
                                            const PpScale& sc = PpScale{nullptr, nullptr}) {
    static_assert(NF == 4 || ACT != RV_ACT_SILU_MUL, "the gated epilogue pairs fragments: 256-column panels only");
    constexpr int WN = NF * 16;   // columns per wave
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wr * 128 + mi * 16 + fr;
        if (m >= M) continue;
        f32x4 acc[NF][1];   // this row block's fragments, dequantised for FP8 operands ([.][0] keeps the indexing below)
#pragma unroll
        for (int ni = 0; ni < NF; ++ni) acc[ni][0] = F8 ? pp_scaled(sc, m, n0 + wc * WN + ni * 16 + kg * 4, acc_in[ni][mi]) : acc_in[ni][mi];
        if constexpr (ROPE && NF == 4) {   // 64-column slab: one head of one section (see pp_rope_row_store)
            const int D = qr.H * 128, nb = n0 + wc * WN;
            const int sec = __builtin_amdgcn_readfirstlane(nb / D), hd0 = nb - sec * D;
            const QkvRow row = qkv_rope_row(qr, m);
            if (row.b1 > row.b0) {
                f32x4 v[4];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) v[ni] = acc[ni][0];
                pp_rope_row_store<4>(v, qr, row, sec, hd0 >> 7, hd0 & 127, kg);
            }
        } else if constexpr (ROPE) {
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) qkv_rope_store(qr, m, n0 + wc * WN + ni * 16 + kg * 4, acc[ni][0]);
        } else if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
            for (int ni = 0; ni + 1 < NF; ni += 2) {
                const int n = n0 + wc * WN + ni * 16;
                const int no = (n >> 1) + kg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][0][r]) * acc[ni + 1][0][r];
                if constexpr ((RS_PROBE_K_ & 32768) != 0) {      // (timing probe: the gated epilogue computed, not stored)
                    const u32x2 o_ = pack_op16x4(v);
                    asm volatile("" ::"v"(o_[0]), "v"(o_[1]));
                    continue;
                }
                if constexpr ((PP_ABL & 64) != 0) {      // (probe: computed, not stored)
                    asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
                    continue;
                }
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < NF; ++ni) {
                const int n = n0 + wc * WN + ni * 16 + kg * 4;
                f32x4 v = acc[ni][0];
                if (bias) v += *(const f32x4*)(bias + n);
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if constexpr ((PP_ABL & 64) != 0) {      // (probe: computed, not stored)
                    asm volatile("" ::"v"(v));
                    continue;
                }
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
            }
        }
    }
}

// Whole-panel epilogue of the four-wave form: wave = (wr, wc) owns rows wr * 128 .. + 127, columns wc * 128 .. + 127; acc[nj][mi].
// The fused QKV form: a wave's 128 (64 in the eight-wave form) columns lie in ONE head of ONE section (q, k or v: D and the slab start are
// multiples of the slab width, which divides 128), so section and head are decided once per panel (uniform) and the row (group, sequence,
// position, table row: two integer divisions) once per row block instead of once per fragment.  Values and addresses are those of
// qkv_rope_store (kernels.h) - with its per-fragment section tests inlined 64 times the register allocator gave up on keeping the
// four-wave form's accumulators in place.
template <int OUT_BF16, int ACT, int ROPE>
__device__ __forceinline__ void pp4_epilogue(const f32x4 (&acc)[8][8], const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                             int64_t ldc, int M, int m0, int n0, int wave, int lane, const QkvRope& qr) {
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, kg = lane >> 4;
    if constexpr (ROPE) {
        const int D = qr.H * 128, nb = n0 + wc * 128;
        const int sec = __builtin_amdgcn_readfirstlane(nb / D), hd0 = nb - sec * D, head = hd0 >> 7;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wr * 128 + mi * 16 + fr;
            if (m >= M) continue;
            const QkvRow row = qkv_rope_row(qr, m);
            if (row.b1 <= row.b0) continue;
            f32x4 v[8];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) v[ni] = acc[ni][mi];
            pp_rope_row_store<8>(v, qr, row, sec, head, 0, kg);
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
        const int m = m0 + wr * 128 + mi * 16 + fr;
        if (m >= M) continue;
        if (ACT == RV_ACT_SILU_MUL) {
#pragma unroll
            for (int ni = 0; ni < 8; ni += 2) {
                const int n = n0 + wc * 128 + ni * 16;
                const int no = (n >> 1) + kg * 4;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[ni][mi][r]) * acc[ni + 1][mi][r];
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
            }
        } else {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int n = n0 + wc * 128 + ni * 16 + kg * 4;
                f32x4 v = acc[ni][mi];
                if (bias) v += *(const f32x4*)(bias + n);
                if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
                }
                if (res) v += rv_residual4(res, ldr, m, n);
                if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = pack_op16x4(v);
                else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
            }
        }
    }
}

// ---- output-tiled launch: one workgroup per 256 x 256 tile ------------------------------------------------------------
template <int OUT_BF16, int ACT, int NF>
__global__ __launch_bounds__(512) void gemm_pp(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp,
                                               const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                               int64_t ldc, int M, int N, int K, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    {   // XCD-aware tile map: the m-tiles of one W column panel run on one XCD (private L2) at about the same time
        const int b = blockIdx.x;
        const int tiles_n8 = tiles_n & ~7;
        if (b < tiles_n8 * tiles_m) {
            const int xcd = b & 7, idx = b >> 3;
            tn = (idx / tiles_m) * 8 + xcd;
            tm = idx % tiles_m;
        } else {
            const int r = b - tiles_n8 * tiles_m;
            tm = r % tiles_m;
            tn = tiles_n8 + r / tiles_m;
        }
    }
    const int m0 = tm * PBM, n0 = tn * (NF * 64);
    PpSrc src;
    pp_sources<NF>(src, A, lda, Wp, M, K, m0, n0, wave, lane);
    f32x4 acc[NF][8];   // [ni = nq * 2 + j][mi = mq * 4 + f]
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    pp_mainloop<NF>(acc, src, 0, K / PBK, smem, wave, lane);
    pp_epilogue<OUT_BF16, ACT, 0, NF>(acc, bias, res, ldr, Cv, ldc, M, m0, n0, wave, lane, QkvRope{});
}

// ---- persistent stream-K launch for few-row problems (up to 32 m-tiles, i.e. the LLM prefill) -------------------------
// One workgroup per CU.  The m-tiles of one W column panel form a TEAM of TS = tiles_m workgroups on ONE XCD (block b
// lands on XCD b % 8) that walk the same (panel, k) range together, so a W piece is fetched from HBM once and the other
// team members hit the XCD's L2 - the coincidence the output-tiled launch gets for free.  The (panel, k-tile) space is cut
// into equal contiguous ranges, one per team (stream-K): every CU runs the same number of k-tiles whatever N is.
//
// A panel whose k-range is shared by c teams is finished by ALL of them together, after their own ranges: every
// participant publishes its fp32 partial tile (register order, 256 KiB, coalesced 16-byte stores) and a flag, and then
// reduces and stores 1/c of the tile: it sums the c partials of its share in ascending-k order (fixed order: results are
// deterministic for a given shape) and runs the epilogue on it.  With the accumulators dead by then, each wave has its
// whole share (32 x 16 B per lane) in flight at once, so a hand-off costs about one memory round trip instead of c - 1
// serial 256 KiB reads by a single finisher.
// Vector L1s and the per-XCD L2s are not coherent, so partials and flags move with sc1 (agent-coherent, write-through /
// L2-missing) stores and loads - NOT with release / acquire fences, whose buffer_wbl2 / buffer_inv sweep the whole L2 of
// the XCD once per hand-off and stall its other 31 workgroups (measured: +150 us on the gate/up projection).  A flag holds
// the launch EPOCH (a host-side counter), so nothing has to be cleaned up and a reader never writes.
constexpr int PARTIAL_F4 = 8 * 32 * 64;   // f32x4 per partial tile; two slots per workgroup (first / last segment)
constexpr int PP_HDR = 8192;              // bytes: 2 flags per workgroup (<= 2046) + status word
// 16-byte agent-coherent (sc1) accesses to the partial-tile slots through a raw buffer descriptor (compiler-tracked vmcnt;
// an offset beyond the descriptor's range reads as zero, which is how absent participants are masked)
typedef unsigned int pp_u32x4 __attribute__((ext_vector_type(4)));
constexpr int PP_SC1 = 16;   // cache-policy bit 4 = sc1 on gfx940+
constexpr unsigned PP_OOB = 0x7fffff00u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pp_slot_rsrc(f32x4* partial, int id) {
    return __builtin_amdgcn_make_buffer_rsrc(partial + (int64_t)id * PARTIAL_F4, 0, PARTIAL_F4 * 16, 0x00020000);
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pp_u32x4, v), r, (int)byte_off, 0, PP_SC1);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, PP_SC1));
}

// Epilogue of one reduction unit = the fragment pair (ni = 2 nip, 2 nip + 1) x mi of wave w's sub-tile.
// (W4: the four-wave form - wave w = (w >> 1, w & 1) owns 128 x 128 outputs, four fragment pairs per row block)
template <int OUT_BF16, int ACT, int ROPE, int F8 = 0, int W4 = 0>
__device__ __forceinline__ void pp_epilogue_unit(f32x4 v0, f32x4 v1, int w, int mi, int nip, const float* __restrict__ bias,
                                                 const float* res, int64_t ldr, void* Cv, int64_t ldc, int M, int m0, int n0, int lane,
                                                 const QkvRope& qr, const PpScale& sc = PpScale{nullptr, nullptr}) {
    const int fr = lane & 15, kg = lane >> 4;
    const int m = m0 + (W4 ? (w >> 1) : (w >> 2)) * 128 + mi * 16 + fr;
    if (m >= M) return;
    const int nf = n0 + (W4 ? (w & 1) * 128 : (w & 3) * 64) + nip * 32;   // first column of fragment ni = 2 nip
    if constexpr (F8) {
        v0 = pp_scaled(sc, m, nf + kg * 4, v0);
        v1 = pp_scaled(sc, m, nf + 16 + kg * 4, v1);
    }
    if constexpr (ROPE) {
        qkv_rope_store(qr, m, nf + kg * 4, v0);
        qkv_rope_store(qr, m, nf + 16 + kg * 4, v1);
    } else if (ACT == RV_ACT_SILU_MUL) {
        const int no = (nf >> 1) + kg * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu(v0[r]) * v1[r];
        if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + no) = pack_op16x4(v);
        else *(f32x4*)((float*)Cv + (int64_t)m * ldc + no) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int n = nf + e * 16 + kg * 4;
            f32x4 v = e ? v1 : v0;
            if (bias) v += *(const f32x4*)(bias + n);
            if (ACT == RV_ACT_RELU || ACT == RV_ACT_QUICK_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = rv_act_apply<ACT>(v[r]);
            }
            if (res) v += rv_residual4(res, ldr, m, n);
            if (OUT_BF16) *(u32x2*)((op16_t*)Cv + (int64_t)m * ldc + n) = pack_op16x4(v);
            else *(f32x4*)((float*)Cv + (int64_t)m * ldc + n) = v;
        }
    }
}

// Reduce + store this workgroup's share of a shared panel: units x = j, j + c, ... of the 128 (wave, mi, ni-pair) units,
// dealt to the 8 waves.  C participants are read per pass (ids[base .. base + C), absent ones masked); C = c for c <= 4.
template <int C, int OUT_BF16, int ACT, int ROPE, int F8 = 0, int W4 = 0>
__device__ __forceinline__ void pp_reduce_share(f32x4* partial, const int* ids, int c, int j, int wave, int lane,
                                                const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv, int64_t ldc,
                                                int M, int m0, int n0, const QkvRope& qr, const PpScale& sc = PpScale{nullptr, nullptr},
                                                float poison = 0.f) {
    // poison: 0 normally; NaN when a participant never published its partial tile (bounded wait expired): the share is then
    // stored as NaN instead of a silently incomplete sum
    constexpr int NW = W4 ? 4 : 8;                                  // waves of the workgroup
    constexpr int UPW = C <= 4 ? (128 + NW * C - 1) / (NW * C) : 1;   // units per wave and pass
    const int per_pass = NW * UPW;
    for (int q0 = 0; j + q0 * c < 128; q0 += per_pass) {
        f32x4 s[UPW][2];
#pragma unroll
        for (int u = 0; u < UPW; ++u) s[u][0] = s[u][1] = f32x4{poison, poison, poison, poison};
        for (int base = 0; base < c; base += C) {
            f32x4 v[UPW][C][2];
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const bool have = base + i < c;
                const __amdgpu_buffer_rsrc_t r = pp_slot_rsrc(partial, __builtin_amdgcn_readfirstlane(ids[have ? base + i : 0]));
#pragma unroll
                for (int u = 0; u < UPW; ++u) {
                    const int x = j + (q0 + wave + NW * u) * c;
                    // unit x -> (wave, mi, fragment pair): the published order is fragment (wave, ni, mi) at ((wave * F + ni * 8 + mi) * 64 + lane) * 16, F = 32 / 64
                    const unsigned fragx = W4 ? (unsigned)((x >> 5) * 64 + (x & 3) * 16 + ((x >> 2) & 7)) : (unsigned)((x >> 4) * 32 + (x & 1) * 16 + ((x >> 1) & 7));
                    const unsigned off = (have && x < 128) ? (fragx * 64 + lane) * 16 : PP_OOB;
                    v[u][i][0] = ld_sc1(r, off);
                    v[u][i][1] = ld_sc1(r, off == PP_OOB ? PP_OOB : off + 8 * 1024);
                }
            }
#pragma unroll
            for (int u = 0; u < UPW; ++u)
#pragma unroll
                for (int i = 0; i < C; ++i) {
                    s[u][0] += v[u][i][0];
                    s[u][1] += v[u][i][1];
                }
        }
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const int x = j + (q0 + wave + NW * u) * c;
            if (x < 128) {
                if constexpr (W4) pp_epilogue_unit<OUT_BF16, ACT, ROPE, F8, 1>(s[u][0], s[u][1], x >> 5, (x >> 2) & 7, x & 3, bias, res, ldr, Cv, ldc, M, m0, n0, lane, qr, sc);
                else pp_epilogue_unit<OUT_BF16, ACT, ROPE, F8>(s[u][0], s[u][1], x >> 4, (x >> 1) & 7, x & 1, bias, res, ldr, Cv, ldc, M, m0, n0, lane, qr, sc);
            }
        }
    }
}

template <int OUT_BF16, int ACT, int ROPE, int NF, int F8, int W4>
__device__ __forceinline__ void pp_sk_body(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp,
                                           const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                           int64_t ldc, int M, int N, int K, int tiles_m, int TS, int nk, int dp_panels,
                                           int total_units, f32x4* partial, int* flags, int* status, int epoch, const QkvRope& qr,
                                           const PpScale& sc, int PGx, int tiles_n) {
    static_assert(!W4 || (NF == 4 && !F8), "the four-wave form: bf16 operands, 256-column panels");
    // PG > 1 (many-row problems: batched prefills): a team is tiles_m x PG workgroups - the m-tiles of PG ADJACENT panels walk the same
    // k-range together, so an activation k-slice is fetched once per PG panels too (with one panel per team the 33 MB activation panel
    // of a 4020-row pass is re-read for every panel: 2.5 GB of fabric traffic per gate/up launch, 20 % of its time).  The unit space,
    // dp_panels and the hand-off then count panel GROUPS; a workgroup whose panel lies past the last one sits its items out.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int teams_per_x = (gridDim.x >> 3) / TS;
    // PGx = PG | MHS << 8.  MHS = 1 (round 4): a team covers HALF the m-tiles (tiles_m of them - the caller passes the per-team count) of twice as
    // many panels: a unit of the panel / stream-K space is then (panel group v >> 1, m-half v & 1).  With 16 m-tiles (4 x 1005 rows) a team is
    // 8 x 4 tiles instead of 16 x 2: per 32 tiles an XCD fetches half the activation panel + 4 weight panels (24.8 MB at K = 4096) instead of all
    // of it + 2 (37.6 MB) - the activation re-reads out of the infinity cache were 6.2x the algorithmic bytes of a gate/up launch.
    const int PG = PGx & 255, MHS = PGx >> 8;
    const int team = xcd * teams_per_x + slot / TS, tsl = slot % TS, tm = tsl % tiles_m, pgi = tsl / tiles_m;
    const int T = 8 * teams_per_x;
    if (pgi >= PG || slot >= teams_per_x * TS) return;
    auto unit_m0 = [&](int v) { return ((v & ((1 << MHS) - 1)) * tiles_m + tm) * PBM; };      // first row of this workgroup's tile of unit v
    auto unit_panel = [&](int v) { return (v >> MHS) * PG + pgi; };                            // ... and its panel
    auto ub = [&](int t) { return (int)((int64_t)t * total_units / T); };   // first unit of team t's range
    const int u_begin = ub(team), u_end = ub(team + 1);
    int shared_panel[2] = {-1, -1};   // stream-K panels this workgroup published a partial of: slot 0 = entered mid-panel, 1 = head

    // Work list: first this team's range of the stream-K unit space (the last `panels % T` panels, cut evenly: always
    // partial pieces), then its whole panels (panel r * T + team) - the pieces are published long before they are needed.
    const int whole_rounds = dp_panels / T;
    // (round 6) the loop looks one work item ahead: the k-split main loop's first loads of item i + 1 are issued before the whole-panel epilogue of item i
    struct Item {
        int panel, ks0, nks, sk_panel, n0, m0;
        bool ok;
    };
    int u_ = u_begin, r_ = 0;
    auto next_item = [&]() {
        Item it{0, 0, 0, -1, 0, 0, false};
        for (;;) {
            it.sk_panel = -1;
            if (u_ < u_end) {
                it.sk_panel = u_ / nk;
                it.ks0 = u_ - it.sk_panel * nk;
                it.nks = min(nk - it.ks0, u_end - u_);
                it.panel = dp_panels + it.sk_panel;
                u_ += it.nks;
            } else if (r_ < whole_rounds) {
                it.panel = r_ * T + team;
                it.ks0 = 0;
                it.nks = nk;
                ++r_;
            } else {
                return it;
            }
            const int tpanel = unit_panel(it.panel);     // this workgroup's panel of the group
            if (tpanel >= tiles_n) continue;
            it.n0 = tpanel * (NF * 64);
            it.m0 = unit_m0(it.panel);
            it.ok = true;
            return it;
        }
    };
    constexpr bool KS_LOOP = !W4 && NF == 4 && !F8 && (PP_KSPLIT & (ROPE ? 2 : 1)) != 0;
    constexpr bool EARLY = KS_LOOP && ROPE != 2 && PP_EARLY;   // (the LDS-staged q / k / v epilogue needs the LDS itself)
    [[maybe_unused]] PpSrc src_pre;
    [[maybe_unused]] bool pre = false;
    for (Item it = next_item(); it.ok;) {
        [[maybe_unused]] const int panel = it.panel;
        const int ks0 = it.ks0, nks = it.nks, sk_panel = it.sk_panel, n0 = it.n0;
        // launder m0 so that the row-dependent address math of the epilogue is not hoisted out of this loop (it would stay live across
        // the main loop and push the accumulators into scratch)
        int m0 = it.m0;
        asm volatile("" : "+s"(m0));
        constexpr int NFW = W4 ? 8 : NF;   // fragments per wave along N
        f32x4 acc[NFW][8];
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if PP_ABL & 32
        const uint64_t t_ml = __builtin_readcyclecounter();
#endif
        if constexpr (W4) {
            Pp4Src src;
            pp4_sources(src, A, lda, Wp, M, K, m0, n0, wave, lane);
            pp4_mainloop(acc, src, ks0, nks, smem, wave, lane);
        } else {
            PpSrc src;
            if (EARLY && pre) src = src_pre;
            else pp_sources<NF>(src, A, lda, Wp, M, K, m0, n0, wave, lane);
            pp_mainloop<NF, F8, (PP_KSPLIT & (ROPE ? 2 : 1)) != 0>(acc, src, ks0, nks, smem, wave, lane, ROPE ? 2 : ACT == RV_ACT_SILU_MUL ? 0 : 1, EARLY && pre);
        }
        it = next_item();
        pre = false;
        if constexpr (EARLY) {
            if (ks0 == 0 && nks == nk && it.ok) {   // the next item's first loads travel under this item's epilogue
                pp_sources<NF>(src_pre, A, lda, Wp, M, K, it.m0, it.n0, wave, lane);
                ks_issue_prologue(src_pre, it.ks0, it.nks, smem, wave);
                pre = true;
            }
        }
#if PP_ABL & 32
        if (tid == 0 && blockIdx.x < 256) atomicAdd(&pp_stamp_ext[((ROPE ? 2 : ACT == RV_ACT_SILU_MUL ? 0 : 1) * 256 + blockIdx.x) * 4 + 1], (unsigned long long)(__builtin_readcyclecounter() - t_ml));
#endif

#if PP_ABL & 32
        const uint64_t t_ep = __builtin_readcyclecounter();
#define PP_EP_STAMP if (tid == 0 && blockIdx.x < 256) { atomicAdd(&pp_stamp_ext[((ROPE ? 2 : ACT == RV_ACT_SILU_MUL ? 0 : 1) * 256 + blockIdx.x) * 4 + 2], (unsigned long long)(__builtin_readcyclecounter() - t_ep)); atomicAdd(&pp_stamp_ext[((ROPE ? 2 : ACT == RV_ACT_SILU_MUL ? 0 : 1) * 256 + blockIdx.x) * 4 + 3], 1ull); }
#else
#define PP_EP_STAMP
#endif
        if (ks0 == 0 && nks == nk) {   // whole panel: finish it from the registers
            if constexpr (W4) pp4_epilogue<OUT_BF16, ACT, ROPE>(acc, bias, res, ldr, Cv, ldc, M, m0, n0, wave, lane, qr);
            else if constexpr (ROPE == 2 && NF == 4) {      // the LDS-staged form (a kernel of its own: with both forms in one kernel the accumulators spilled, 361 -> 393 us);
                pp_epilogue_rope_lds<F8>(acc, M, m0, n0, wave, lane, qr, sc, smem);      // every wave leaves the LDS before the next main loop stages into it
                __syncthreads();
            } else pp_epilogue<OUT_BF16, ACT, ROPE, NF, F8>(acc, bias, res, ldr, Cv, ldc, M, m0, n0, wave, lane, qr, sc);
            PP_EP_STAMP
            continue;
        }
        if constexpr (NF == 4) {   // (192-column panels are launched without a stream-K tail: whole panels only)
            // shared panel: publish the partial accumulators (write-through sc1 stores), then the flag
            const int ps = ks0 == 0 ? 1 : 0, id = (team * TS + tsl) * 2 + ps;
            shared_panel[ps] = sk_panel;
            const __amdgpu_buffer_rsrc_t pr = pp_slot_rsrc(partial, id);
            const unsigned off = (wave * (NFW * 8) * 64 + lane) * 16;
#pragma unroll
            for (int ni = 0; ni < NFW; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) st_sc1(pr, off + (ni * 8 + mi) * 1024, acc[ni][mi]);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flags + id, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if constexpr (NF != 4) return;

    // ---- finish the shared panels: every participant reduces and stores its 1/c share --------------------------------
    int* ids = (int*)smem;   // participants' slot ids, ascending k
    for (int ps = 0; ps < 2; ++ps) {
        const int panel = shared_panel[ps];
        if (panel < 0) continue;
        const int lo = panel * nk, hi = lo + nk;
        int t0 = team, t1 = team;
        while (t0 > 0 && ub(t0) > lo) --t0;
        while (t1 + 1 < T && ub(t1 + 1) < hi) ++t1;
        __syncthreads();   // ids is reused
        int c = 0, j = 0;
        for (int t = t0; t <= t1; ++t) {
            if (ub(t) == ub(t + 1)) continue;   // empty range: took no part
            if (t == team) j = c;
            if (tid == 0) ids[c] = (t * TS + tsl) * 2 + (ub(t) <= lo ? 1 : 0);
            ++c;
        }
        if (tid == 0) {
            int gave_up = 0;
            for (int i = 0; i < c; ++i) {
                unsigned spins = 0;
                while (__hip_atomic_load(flags + ids[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 24)) {
                        *status = 1;
                        gave_up = 1;
                        break;
                    }
                }
            }
            ids[1024] = gave_up;   // (ids[0 .. c) hold the participants, c <= #teams <= 256)
        }
        __syncthreads();
        const float poison = ids[1024] ? __int_as_float(0x7fc00000) : 0.f;   // a missing partial poisons this share (never a silent partial sum)
        const int m0 = unit_m0(dp_panels + panel), n0 = unit_panel(dp_panels + panel) * PBN;
        if (c == 2) pp_reduce_share<2, OUT_BF16, ACT, ROPE, F8, W4>(partial, ids, c, j, wave, lane, bias, res, ldr, Cv, ldc, M, m0, n0, qr, sc, poison);
        else if (c == 3) pp_reduce_share<3, OUT_BF16, ACT, ROPE, F8, W4>(partial, ids, c, j, wave, lane, bias, res, ldr, Cv, ldc, M, m0, n0, qr, sc, poison);
        else if (c == 4) pp_reduce_share<4, OUT_BF16, ACT, ROPE, F8, W4>(partial, ids, c, j, wave, lane, bias, res, ldr, Cv, ldc, M, m0, n0, qr, sc, poison);
        else pp_reduce_share<8, OUT_BF16, ACT, ROPE, F8, W4>(partial, ids, c, j, wave, lane, bias, res, ldr, Cv, ldc, M, m0, n0, qr, sc, poison);
    }
}

template <int OUT_BF16, int ACT, int ROPE, int NF, int F8 = 0>
__global__ __launch_bounds__(512) void gemm_pp_sk(const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp,
                                                  const float* __restrict__ bias, const float* res, int64_t ldr, void* Cv,
                                                  int64_t ldc, int M, int N, int K, int tiles_m, int TS, int nk, int dp_panels,
                                                  int total_units, f32x4* partial, int* flags, int* status, int epoch, QkvRope qr,
                                                  PpScale sc, int PG, int tiles_n) {
    pp_sk_body<OUT_BF16, ACT, ROPE, NF, F8, 0>(A, lda, Wp, bias, res, ldr, Cv, ldc, M, N, K, tiles_m, TS, nk, dp_panels, total_units, partial, flags,
                                               status, epoch, qr, sc, PG, tiles_n);
}
// the four-wave form (pp4_mainloop): one wave per SIMD, all 512 registers of the unified file
template <int OUT_BF16, int ACT, int ROPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_pp4_sk(
    const op16_t* __restrict__ A, int64_t lda, const op16_t* __restrict__ Wp, const float* __restrict__ bias, const float* res, int64_t ldr,
    void* Cv, int64_t ldc, int M, int N, int K, int tiles_m, int TS, int nk, int dp_panels, int total_units, f32x4* partial, int* flags,
    int* status, int epoch, QkvRope qr, PpScale sc, int PG, int tiles_n) {
    pp_sk_body<OUT_BF16, ACT, ROPE, 4, 0, 1>(A, lda, Wp, bias, res, ldr, Cv, ldc, M, N, K, tiles_m, TS, nk, dp_panels, total_units, partial, flags,
                                             status, epoch, qr, sc, PG, tiles_n);
}

int pp_device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}
int pp_num_cus() {
    const int n = pp_device_cus();
    const int want = rv_cur_opts().gemm_cus & ~7;   // 0 = every CU of the device; otherwise the persistent GEMMs use this many
    return (want > 0 && want <= n) ? want : n;
}

// The > 64 KiB dynamic-LDS opt-in is a per-DEVICE attribute of the function: remembered per device (bit d of the mask), not per process.
template <typename Kern>
int reserve_lds(Kern k, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_relaxed) & bit)) {
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS) != hipSuccess) {
            rv_set_error("gemm_pp: cannot reserve %d bytes of LDS", PP_LDS);
            return RV_ERR_HIP;
        }
        done.fetch_or(bit, std::memory_order_relaxed);
    }
    return RV_OK;
}

template <int OUT_BF16, int ACT, int NF>
int launch(const op16_t* A, int64_t lda, const op16_t* Wp, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
           int M, int N, int K, hipStream_t st) {
    static std::atomic<uint64_t> attr_set{0};
    if (int rc = reserve_lds(gemm_pp<OUT_BF16, ACT, NF>, attr_set)) return rc;
    const int tiles_m = (int)cdiv(M, PBM), tiles_n = N / (NF * 64);
    hipLaunchKernelGGL((gemm_pp<OUT_BF16, ACT, NF>), dim3(tiles_m * tiles_n), dim3(512), PP_LDS, st, A, lda, Wp, bias, res, ldr, C, ldc, M, N,
                       K, tiles_m, tiles_n);
    return RV_OK;
}

std::atomic<int> g_epoch{0};

// teams on the chip for a problem of M rows (team = the m-tiles of one panel, all on one XCD)
int pp_teams(int64_t M) {
    const int tiles_m = (int)cdiv(M, PBM), per_x = pp_num_cus() >> 3;
    return tiles_m <= per_x ? 8 * (per_x / tiles_m) : 0;
}

template <int OUT_BF16, int ACT, int ROPE, int NF, int F8 = 0>
int launch_sk(const op16_t* A, int64_t lda, const op16_t* Wp, const float* bias, const float* res, int64_t ldr, void* C, int64_t ldc,
              int M, int N, int K, void* ws, hipStream_t st, const QkvRope& qr, PpScale sc = PpScale{nullptr, nullptr}) {
    static std::atomic<uint64_t> attr_set{0}, attr_set4{0};
    constexpr bool can4 = NF == 4 && !F8;
    const bool w4 = can4 && rv_cur_opts().gemm_waves == 4;
    if (w4) {
        if constexpr (can4) if (int rc = reserve_lds(gemm_pp4_sk<OUT_BF16, ACT, ROPE>, attr_set4)) return rc;
    } else if (int rc = reserve_lds(gemm_pp_sk<OUT_BF16, ACT, ROPE, NF, F8>, attr_set)) return rc;
    const int tiles_m = (int)cdiv(M, PBM), tiles_n = N / (NF * 64), nk = K / PBK;
    const int G = pp_num_cus() & ~7, per_x = G >> 3;
    // team = the m-tiles of one panel; G/8 - (G/8) / TS * TS workgroups per XCD stay idle.  From 8 m-tiles on (batched prefills) with
    // 256-column panels: the m-tiles of PG adjacent panels - one team per XCD (see the kernel)
    int PG = (NF == 4 && tiles_m >= 8 && per_x / tiles_m >= 2) ? per_x / tiles_m : 1;
    int tm_team = tiles_m, MHS = 0;
    // half the m-tiles x twice the panels per team when that fills the XCD at least as well with fewer bytes per tile (see the kernel):
    // 16 m-tiles: 8 x 4 instead of 16 x 2; 12: 6 x 5 = 30 workgroups per XCD instead of 12 x 2 = 24; 10: 5 x 6 instead of 10 x 3
    if (NF == 4 && rv_cur_opts().gemm_mhalf && tiles_m >= 10 && tiles_m % 2 == 0 && tiles_m <= per_x) {
        const int tm2 = tiles_m / 2, pg2 = per_x / tm2;
        if (pg2 <= 255 && (tm2 * pg2 > tiles_m * PG || (tm2 * pg2 == tiles_m * PG && tm2 + pg2 < tiles_m + PG))) {
            tm_team = tm2;
            PG = pg2;
            MHS = 1;
        }
        // ... and a QUARTER of the m-tiles x four times the panels when the halves are still tall (round 6: 8 prefills to a pass = 32 m-tiles: 8 x 4 tiles per team instead
        // of 16 x 2 - the same shape the 16-tile pass has had since round 4; per 32 tiles an XCD then fetches 16.5 + 8.4 MB instead of 33 + 4.2)
        if (MHS == 1 && rv_cur_opts().gemm_mhalf >= 2 && tm_team >= 16 && tm_team % 2 == 0) {
            const int tm4 = tm_team / 2, pg4 = per_x / tm4;
            if (pg4 <= 255 && tm4 * pg4 >= tm_team * PG) {
                tm_team = tm4;
                PG = pg4;
                MHS = 2;
            }
        }
    }
    const int TS = tm_team * PG;
    const int T = (PG > 1 || MHS) ? 8 * (per_x / TS) : pp_teams(M);
    const int groups = ((tiles_n + PG - 1) / PG) << MHS, dp_panels = groups / T * T;      // (in panel groups x m-halves)
    if (NF != 4 && dp_panels != tiles_n) {
        rv_set_error("gemm_pp: 192-column panels need a panel count that is a multiple of the %d teams", T);
        return RV_ERR_ARG;
    }
    int epoch = ++g_epoch;
    if (epoch <= 0) { g_epoch = 1; epoch = 1; }   // 0 = the zero-initialised workspace
    int* flags = (int*)ws;
    int* status = flags + PP_HDR / 4 - 1;
    f32x4* partial = (f32x4*)((char*)ws + PP_HDR);
    if constexpr (can4) {
        if (w4) {
            hipLaunchKernelGGL((gemm_pp4_sk<OUT_BF16, ACT, ROPE>), dim3(G), dim3(256), PP_LDS, st, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K, tm_team, TS,
                               nk, dp_panels, (groups - dp_panels) * nk, partial, flags, status, epoch, qr, sc, PG | MHS << 8, tiles_n);
            return RV_OK;
        }
    }
    if constexpr (ROPE == 1 && NF == 4) {
        // ROPE = 2: the same kernel with the whole-panel q / k / v epilogue staged through LDS (pp_epilogue_rope_lds; option qkv_lds; prefill rows only)
        static std::atomic<uint64_t> attr_set2{0};
        if (rv_cur_opts().qkv_lds && !qr.row_pos) {
            if (int rc = reserve_lds(gemm_pp_sk<OUT_BF16, ACT, 2, NF, F8>, attr_set2)) return rc;
            hipLaunchKernelGGL((gemm_pp_sk<OUT_BF16, ACT, 2, NF, F8>), dim3(G), dim3(512), PP_LDS, st, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K,
                               tm_team, TS, nk, dp_panels, (groups - dp_panels) * nk, partial, flags, status, epoch, qr, sc, PG | MHS << 8, tiles_n);
            return RV_OK;
        }
    }
    hipLaunchKernelGGL((gemm_pp_sk<OUT_BF16, ACT, ROPE, NF, F8>), dim3(G), dim3(512), PP_LDS, st, A, lda, Wp, bias, res, ldr, C, ldc, M, N, K,
                       tm_team, TS, nk, dp_panels, (groups - dp_panels) * nk, partial, flags, status, epoch, qr, sc, PG | MHS << 8, tiles_n);
    return RV_OK;
}

}  // namespace

// The persistent prefill GEMMs may be told to leave CUs free (option "gemm_cus"): a 128 KiB-LDS workgroup owns its CU, so with a
// grid of 192 on a 256-CU device 8 CUs per XCD stay available to whatever another stream launches (the HBM-bound decode
// GEMVs of a second recursion in flight cannot share a CU with these workgroups: both fill the register file).

size_t gemm_pp_ws_bytes() { return PP_HDR + (size_t)2 * pp_device_cus() * PARTIAL_F4 * sizeof(f32x4); }

bool gemm_pp_supported(int w_layout, int64_t M, int64_t N, int64_t K) {
    return w_layout == 1 && M > 16 && N % PBN == 0 && K % PBK == 0;
}

// stream-K form: few-row problems only (a team of tiles_m workgroups must fit on one XCD); needs the zero-initialised workspace
bool gemm_pp_sk_supported(int w_layout, int64_t M, int64_t N, int64_t K) {
    const int cus = pp_num_cus();
    return gemm_pp_supported(w_layout, M, N, K) && pp_teams(M) > 0 && cus <= 1023 && (N / PBN) * (K / PBK) < (1ll << 30);
}

// Persistent-launch plan for a few-row problem: 0 = not worth it, 4 = 256-column panels (whole panels + stream-K tail),
// 3 = 192-column panels when they deal out EXACTLY (panel count a multiple of the teams: no hand-off and no idle CU - the
// fused QKV projection at M ~ 1000: 64 panels of 192 on 64 teams, 130 -> ~100 us against the ring kernel).
// Stream-K pays when a panel is cut at least 4 ways (every workgroup then owns ONE piece of one panel: one 256 KiB
// publish and one shared reduction per launch) and every team still has a few k-tiles of work.  Measured on MI355X at
// M = 1005: down projection 128 -> 92 us, o projection 57 -> 52 us, gate/up (64 whole panels + 22 split ones on 64 teams)
// 200 -> 180 us; a problem with 0.75 panels per team moves 2 pieces per workgroup through HBM and ends up level with the
// ring kernel, so it stays there.
int gemm_pp_sk_plan(int64_t M, int64_t N, int64_t K, bool gated) {
    const int tiles_m = (int)cdiv(M, PBM), per_x = pp_num_cus() >> 3;
    const int64_t T = pp_teams(M), nk = K / PBK;
    if (T == 0 || K < 2048) return 0;
    int ts = per_x / tiles_m * tiles_m;                                   // workgroups per XCD that belong to a team (launch_sk)
    if (rv_cur_opts().gemm_mhalf && tiles_m >= 10 && tiles_m % 2 == 0) ts = std::max(ts, tiles_m / 2 * (per_x / (tiles_m / 2)));
    if ((per_x - ts) * 10 > per_x) return 0;                              // > 10 % of the CUs left without a team
    const int64_t panels = N / PBN, sk_panels = panels % T;
    if (sk_panels == 0 && panels >= T) return 4;                     // 256-column panels deal out exactly (batched prefills: 48 qkv panels on 16 teams)
    if (!gated && N % 192 == 0 && (N / 192) % T == 0) return 3;
    if (sk_panels == 0) return 4;
    if (panels < T) {   // pure split-k: >= 4 pieces per panel with >= 8 k-tiles each, or 2 pieces of >= 64 k-tiles (one hand-off behind
                        // a long k-range: the down projection at ~2000 rows, 167 vs 194 us on the ring kernel)
        if (panels * 4 <= T && panels * nk >= 8 * T) return 4;
        return (panels * 2 <= T && panels * nk >= 64 * T) ? 4 : 0;
    }
    return (sk_panels * 8 >= T && sk_panels * nk >= 8 * T) ? 4 : 0;            // whole panels + tail: one piece (sometimes two) per workgroup
}
bool gemm_pp_sk_profitable(int64_t M, int64_t N, int64_t K) { return gemm_pp_sk_plan(M, N, K, true) != 0; }

// Output-tiled ping-pong pays for long K (the prologue / epilogue of a 256x256 tile is ~3 us) when the tiles fill the
// CUs: 1.2 vs 0.86 PFLOP/s at 4096^3; short-K adapter GEMMs (K = 768) and ragged tile counts stay on the ring kernel.
// -> 4 (256-column tiles), 3 (192-column tiles: N = 768 of the adapter's FFN-2 over 100 x 257 rows gives 101 x 4 = 404 tiles = 79 % of two
// rounds where 256-column tiles give 303 = 59 %), 0 (ring kernel).
int gemm_pp_dp_plan(int64_t M, int64_t N, int64_t K, bool gated) {
    const int64_t cus = pp_num_cus();
    if (K < 2048) return 0;
    const int64_t t4 = cdiv(M, PBM) * (N / PBN), r4 = cdiv(t4, cus);
    if (t4 >= cus && t4 * 100 >= r4 * cus * 85) return 4;
    if (!gated && N % 192 == 0) {
        const int64_t t3 = cdiv(M, PBM) * (N / 192), r3 = cdiv(t3, cus);
        if (t3 >= cus && t3 * 100 >= r3 * cus * 75) return 3;
    }
    return 0;
}
bool gemm_pp_dp_profitable(int64_t M, int64_t N, int64_t K) { return gemm_pp_dp_plan(M, N, K, false) == 4; }

int gemm_pp_launch(const void* A, int64_t lda, const void* Wp, const float* bias, const float* res, int64_t ldr, void* C,
                   int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, void* ws, hipStream_t st) {
    const op16_t* a = (const op16_t*)A;
    const op16_t* w = (const op16_t*)Wp;
    const int ob = out_dtype == RV_OP16;
    const bool sk = ws && gemm_pp_sk_supported(1, M, N, K);
    const bool nf3 = sk && act != RV_ACT_SILU_MUL && gemm_pp_sk_plan(M, N, K, false) == 3;
    int rc;
    if (!sk && gemm_pp_dp_plan(M, N, K, act == RV_ACT_SILU_MUL) == 3) {     // output-tiled with 192-column tiles
        if (ob && act == RV_ACT_NONE) rc = launch<1, RV_ACT_NONE, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st);
        else if (ob && act == RV_ACT_RELU) rc = launch<1, RV_ACT_RELU, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st);
        else if (!ob && act == RV_ACT_NONE) rc = launch<0, RV_ACT_NONE, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st);
        else if (!ob && act == RV_ACT_RELU) rc = launch<0, RV_ACT_RELU, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st);
        else rc = launch<0, RV_ACT_QUICK_GELU, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st);
        if (rc) return rc;
        RV_CHECK_LAUNCH("gemm_pp");
        return RV_OK;
    }
#define PP(OB, AC)                                                                                                       \
    rc = sk ? launch_sk<OB, AC, 0, 4>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, ws, st, QkvRope{})      \
            : launch<OB, AC, 4>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, st)
#define PP3(OB, AC) rc = launch_sk<OB, AC, 0, 3>(a, lda, w, bias, res, ldr, C, ldc, (int)M, (int)N, (int)K, ws, st, QkvRope{})
    if (nf3 && ob && act == RV_ACT_NONE) PP3(1, RV_ACT_NONE);
    else if (nf3 && ob && act == RV_ACT_RELU) PP3(1, RV_ACT_RELU);
    else if (nf3 && ob) PP3(1, RV_ACT_QUICK_GELU);
    else if (nf3 && act == RV_ACT_NONE) PP3(0, RV_ACT_NONE);
    else if (nf3 && act == RV_ACT_RELU) PP3(0, RV_ACT_RELU);
    else if (nf3) PP3(0, RV_ACT_QUICK_GELU);
    else if (ob && act == RV_ACT_NONE) PP(1, RV_ACT_NONE);
    else if (ob && act == RV_ACT_RELU) PP(1, RV_ACT_RELU);
    else if (ob && act == RV_ACT_SILU_MUL) PP(1, RV_ACT_SILU_MUL);
    else if (ob && act == RV_ACT_QUICK_GELU) PP(1, RV_ACT_QUICK_GELU);
    else if (!ob && act == RV_ACT_NONE) PP(0, RV_ACT_NONE);
    else if (!ob && act == RV_ACT_RELU) PP(0, RV_ACT_RELU);
    else if (!ob && act == RV_ACT_QUICK_GELU) PP(0, RV_ACT_QUICK_GELU);
    else PP(0, RV_ACT_SILU_MUL);
#undef PP
#undef PP3
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_pp");
    return RV_OK;
}

// fused QKV projection (RoPE + KV-cache append epilogue) on the persistent kernel
int gemm_pp_qkv_rope(const void* A, int64_t lda, const void* Wp, int64_t M, int64_t N, int64_t K, const QkvRope& r, void* ws,
                     hipStream_t st) {
    int rc = gemm_pp_sk_plan(M, N, K, false) == 3
                 ? launch_sk<0, RV_ACT_NONE, 1, 3>((const op16_t*)A, lda, (const op16_t*)Wp, nullptr, nullptr, 0, nullptr, 0, (int)M, (int)N,
                                                   (int)K, ws, st, r)
                 : launch_sk<0, RV_ACT_NONE, 1, 4>((const op16_t*)A, lda, (const op16_t*)Wp, nullptr, nullptr, 0, nullptr, 0, (int)M, (int)N,
                                                   (int)K, ws, st, r);
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_pp_qkv_rope");
    return RV_OK;
}

// FP8 (e4m3fn) x FP8 prefill GEMM on the persistent kernel: A8 [M, K] row-major bytes with per-row scales sa [M], W8p = the
// bf16 fragment packing of the [N, K] byte matrix taken as [N, K / 2] 16-bit words with per-column scales sw [N].  Only the
// persistent stream-K form exists (plan of the K / 2 problem): returns RV_ERR_ARG when the shape has no plan - the caller
// keeps its bf16 path for those.  r != nullptr: fused RoPE + KV-cache append epilogue (QKV projection).
bool gemm_pp_fp8_supported(int64_t M, int64_t N, int64_t K, bool gated, bool rope) {
    if (K % 128 != 0 || !gemm_pp_sk_supported(1, M, N, K / 2)) return false;
    const int plan = gemm_pp_sk_plan(M, N, K / 2, gated);
    if (rope ? plan != 0 : plan == 4) return true;
    if (rope) return false;
    // Two-way split panels (e.g. the N = 4096 projections at M ~ 2000: 16 panels on 32 teams) lose to the bf16 ring kernel
    // in bf16, which is why the bf16 plan leaves them there - but the FP8 alternative is that SAME bf16 ring kernel.
    const int64_t T = pp_teams(M), panels = N / PBN, nk = K / 2 / PBK;
    return T > 0 && panels < T && panels * 2 <= T && panels * nk >= 8 * T;
}
int gemm_pp_fp8(const void* A8, int64_t lda, const float* sa, const void* W8p, const float* sw, const float* res, int64_t ldr, void* C,
                int64_t ldc, int out_dtype, int act, int64_t M, int64_t N, int64_t K, const QkvRope* r, void* ws, hipStream_t st) {
    const op16_t* a = (const op16_t*)A8;
    const op16_t* w = (const op16_t*)W8p;
    const PpScale sc{sa, sw};
    const int m = (int)M, n = (int)N, k2 = (int)(K / 2);
    const int64_t lda2 = lda / 2;
    int rc;
    if (r) {
        rc = gemm_pp_sk_plan(M, N, K / 2, false) == 3
                 ? launch_sk<0, RV_ACT_NONE, 1, 3, 1>(a, lda2, w, nullptr, nullptr, 0, nullptr, 0, m, n, k2, ws, st, *r, sc)
                 : launch_sk<0, RV_ACT_NONE, 1, 4, 1>(a, lda2, w, nullptr, nullptr, 0, nullptr, 0, m, n, k2, ws, st, *r, sc);
    } else if (act == RV_ACT_SILU_MUL && out_dtype == RV_OP16) {
        rc = launch_sk<1, RV_ACT_SILU_MUL, 0, 4, 1>(a, lda2, w, nullptr, res, ldr, C, ldc, m, n, k2, ws, st, QkvRope{}, sc);
    } else if (act == RV_ACT_NONE && out_dtype == RV_F32) {
        rc = launch_sk<0, RV_ACT_NONE, 0, 4, 1>(a, lda2, w, nullptr, res, ldr, C, ldc, m, n, k2, ws, st, QkvRope{}, sc);
    } else if (act == RV_ACT_NONE && out_dtype == RV_OP16) {
        rc = launch_sk<1, RV_ACT_NONE, 0, 4, 1>(a, lda2, w, nullptr, res, ldr, C, ldc, m, n, k2, ws, st, QkvRope{}, sc);
    } else {
        rv_set_error("gemm_pp_fp8: unsupported epilogue (act %d, out dtype %d)", act, out_dtype);
        return RV_ERR_ARG;
    }
    if (rc) return rc;
    RV_CHECK_LAUNCH("gemm_pp_fp8");
    return RV_OK;
}

#if PP_ABL & 32
extern "C" __attribute__((visibility("default"))) int rv_pp_stamps(unsigned long long* host_out, int reset) {
    if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(pp_stamp_buf), sizeof(pp_stamp_buf)) != hipSuccess) return -1;
    if (host_out && hipMemcpyFromSymbol(host_out + 256 * 2 * 9, HIP_SYMBOL(pp_stamp_ext), sizeof(pp_stamp_ext)) != hipSuccess) return -1;
    if (reset) {
        static unsigned long long zeros[256 * 2 * 9];
        if (hipMemcpyToSymbol(HIP_SYMBOL(pp_stamp_buf), zeros, sizeof(zeros)) != hipSuccess) return -1;
        if (hipMemcpyToSymbol(HIP_SYMBOL(pp_stamp_ext), zeros, sizeof(pp_stamp_ext)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
