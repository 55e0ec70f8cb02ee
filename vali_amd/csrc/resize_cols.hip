// Lanczos-3 / bicubic resize of planes that SHRINK vertically (src_h >= dst_h): columns first.
//
// The reference's only resize filter is Lanczos (src/TC/src/TaskResizeSurface.cpp:67,116,224,273; UDPlanar
// src/TC/src/UDSurface.cpp:45,72); its dominant use is the downscale in front of an encoder or a network.  The
// rows-first kernel (resize_taps.hip) filtered every SOURCE row along x -- at 2:1 that is 2.16 gathered, converted and
// filtered samples per output sample, 44 lane-instructions per output, FP32-VALU-bound at 0.37 of the HBM roofline
// (profiles/r02_lanczos.md).  Columns first (specification: oracle/vali_oracle.c resize_plane_taps, the src_h >= dst_h
// branch) turns the expensive half into streaming work:
//
//   vertical pass   a lane owns 8 ADJACENT source elements (one 8 / 16 / 32-byte load per source row straight from
//                   global memory, no staging) and walks the source rows of its tile ONCE, top to bottom.  Every element is
//                   converted to float once and SCATTERED: each of the <= P dst rows whose window contains this source row
//                   has an accumulator (P x 8 registers) and takes fma(w, f, acc) with a WAVE-UNIFORM weight -- a
//                   scalar operand, no per-lane tap fetch.  P = ceil((TAPS - 1) / scale_y) slots; dst row rr lives in slot
//                   rr mod P (floor((rr + P) s) - floor(rr s) >= TAPS - 1: the slot's next row starts on the row that
//                   completes this one at the earliest -- cols_rows).  The weights of all rows of the tile are computed once (lane r = row r) and scattered through
//                   LDS into one register per slot whose lane t holds the weight the slot applies to the wave's t-th
//                   source row (0.0: none), fetched with v_readlane at the row index; a 64-bit mask says which rows
//                   complete a dst row.
//   horizontal pass two completed dst rows at a time: their columns go to a strip in LDS whose slots hold one column of
//                   BOTH rows (channels of interleaved planes de-interleaved into segments, so every plane is the
//                   1-channel problem; even and odd positions in separate halves: around 2:1 neighbouring lanes then read
//                   neighbouring slots) and the wave filters along x: lane l takes elements l, l + 64, l + 128, l + 192
//                   of the tile; a tap is one ds_read_b64 + one v_pk_fma_f32 for two output samples (the weight's half
//                   broadcast by op_sel), e over the even taps, o over the odd ones, e + o: the specification's order; the
//                   results are transposed through 2 KiB of LDS so a lane stores 4 adjacent elements of both rows.  The
//                   tile's column taps are evaluated once per workgroup (wave w the w-th set).
//
// Per output sample at 2:1: 4 conversions + 6-8 packed FMAs (vertical) + 3 LDS reads + 3 packed FMAs (horizontal) + the
// quantiser, against 13 conversions + 2.16 x (funnel shifts + 3 packed FMAs) + 3 in round 2's rows-first kernel.  Exact
// ratios have leaner forms below: 2:1 along x (only the sampled columns are filtered), 3:2 along x (one weight set, the row
// pass in registers) and 3:2 both ways (two statically scheduled slots).  Planes that grow vertically stay on
// resize_taps.hip (rows first: there the row pass is the smaller half), exactly doubled ones on resize_up2.hip.
#include "resize_common.hpp"
#include "resize_weights.hpp"

#include <type_traits>


namespace vali {

constexpr int kColEl = 8;                    // source elements per lane and row
constexpr int kColSpan = kWave * kColEl;     // 512 source elements per tile row
constexpr int kColPadL = 4, kColPadR = 4;    // replicas of the first / last pixel
constexpr int kColStrip = 560;               // SLOTS (float pairs: one column of two dst rows) of a wave's strip: ES segments, pads included
constexpr int kColLds = 2 * (kColStrip + 256); // floats: + the output transposition (256 elements x 2 rows)
constexpr int kColWave = kColLds + 576;        // floats of a wave: + its program (kColProg)

// A channel segment of the strip is two halves of kColHalf slots: positions (pixels) of even index in front, of odd index
// behind.  Around 2:1 the windows of neighbouring dst pixels start two positions apart: their k-th taps are then
// NEIGHBOURING slots of one half (conflict-free ds_read_b64; in a linear strip they are 16 bytes apart: two lanes per
// bank).  Sizes: >= (positions + pads) / 2; chosen so that the ES channels of one pixel -- read by neighbouring lanes --
// and the two halves sit in different banks.
template <int ES> constexpr int kColHalf = ES == 1 ? 272 : ES == 2 ? 136 : 91;

typedef u32 u32_u __attribute__((aligned(1)));

// (a[HA], b[HB]) in one instruction
template <int HA, int HB> __device__ __forceinline__ v2f32 pk_mov(v2f32 a, v2f32 b) {
  v2f32 r;
  if constexpr (HA == 0 && HB == 0) asm("v_pk_mov_b32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (HA == 1 && HB == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(r) : "v"(a), "v"(b));
  else if constexpr (HA == 1) asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// (a[H], a[H]): its own assembly string, so that the compiler cannot fold it into pk_mov's by selecting the operands
template <int H> __device__ __forceinline__ v2f32 pk_dup(v2f32 a) {
  v2f32 r;
  if constexpr (H == 0) asm("v_pk_mov_b32 %0, %1, %1" : "=v"(r) : "v"(a));
  else asm("v_pk_mov_b32 %0, %1, %1 op_sel:[1,1]" : "=v"(r) : "v"(a));
  return r;
}

// ---- the rows of a wave: a PROGRAM in LDS, one entry per source row the wave walks ----
// entry t = kProgEsz<P> dwords: the weight every slot applies to the wave's t-th source row (0.0: the slot has no use for
// it) and ONE control word:
//   bits 0..23   the source row the walk LOADS while it filters row t (its row t + D, clamped to the plane and to the
//                wave's last row): the load address is v_mad_u32_u24(word, pitch, lane offset) -- the flags above bit 23
//                never reach the product
//   bit 24       row t completes a dst row (rows complete in order, slots in turn)
//   bit 25       ... and is also the FIRST tap of the slot's next row: that weight is first_w[t]
//   bit 26       ... and that dst row closes a PAIR (odd row of the wave, or its last): the row pass runs
// and, float planes only (8-dword entries), a second word: bit j: slot j uses row t (a zero weight on a non-finite sample
// is no no-op).  Everything the walk branches on is a bit of this word: no counters, no compares, no booleans kept in masks.
// The walk fetches an entry with ONE broadcast ds_read_b128 (P <= 3) and uses the weights where they land: the packed
// FMAs take them from the VGPR pair by op_sel.  (Round 3 kept one register per slot, lane t = row t, and paid P + 1
// v_readlane and four scalar instructions of mask tests per source row and wave; its program ended at 56 rows -- a lane
// per row -- so a wave owned 24 dst rows and walked 10 % of its source rows twice with its neighbour below.  Entries in
// LDS have no such end: 112 rows, 48 dst rows per wave at 2:1.)
template <int P, bool ACT> constexpr int kProgEsz = (P <= 3 && !ACT) ? 4 : 8;
template <int P, bool ACT> constexpr int kProgRows = (P <= 3 && !ACT) ? 112 : 64;     // entries (the host keeps ns + D below it)
constexpr int kColProg = 576;                                      // floats: entries + first_w
static_assert(kProgRows<3, false> * (kProgEsz<3, false> + 1) <= kColProg && kProgRows<6, true> * (kProgEsz<6, true> + 1) <= kColProg, "program size");
constexpr u32 kProgDone = 1u << 24, kProgDbl = 1u << 25, kProgPair = 1u << 26, kProgSingle = 1u << 27; // (27: the pair is ONE row, the wave's last)
constexpr int kProgSlotShift = 28; // bits 28 .. 30 (ws_rows): the slot of the dst row that completes -- the walk keeps no counter
struct ColProg {
  u32 lds;            // LDS byte address of entry 0 (kept in a VGPR: the ds_read's address operand)
  const float* first_w;
  int ns;             // source rows the wave walks
  int s_begin;        // the first of them (may be < 0: clamped when loaded)
  int y_first, last_rr;
};

// false: the wave has no rows.  `prog`: kColProg floats of this wave's LDS.  D = the walk's rows in flight.
// BYSLOT: column j of an entry is the weight of SLOT j (dst row rr lives in slot rr mod P) instead of the row of age j: the
// accumulators then never move -- the walk picks the completing set by a scalar switch (cols_walk) -- at the price of P
// copies of take(); the specialised producer, whose take() is a strip write, pays that gladly.
// ytab: the tap table of the plane's rows (tap_table.hip) or null
template <int TAPS> __device__ __forceinline__ LzTap<TAPS> lz_tap_of(const float4* tab, int x, float scale) {
  if (tab) { // (wave-uniform)
    const float4 a = tab[2 * x], b = tab[2 * x + 1];
    LzTap<TAPS> t;
    t.w[0] = a.x; t.w[1] = a.y; t.w[2] = a.z; t.w[3] = a.w;
    if constexpr (TAPS == 6) {
      t.w[4] = b.x; t.w[5] = b.y;
    }
    t.i = __float_as_int(b.z);
    return t;
  }
  return make_lz_tap<TAPS>(x, scale);
}

template <int TAPS, int P, bool ACT, int D, bool BYSLOT = false, int NPROG = kProgRows<P, ACT>>
__device__ __forceinline__ bool cols_rows(int sh, int dh, u32 row_tile, int rps, float* prog, ColProg& r, const float4* ytab = nullptr) {
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int ESZ = kProgEsz<P, ACT>;
  const int lane = threadIdx.x & 63;
  const int rows = P * rps;                                     // dst rows of this wave, <= 64
  r.y_first = (int)row_tile * rows;                             // wave-uniform
  if (r.y_first >= dh)
    return false;
  r.last_rr = min(rows, dh - r.y_first) - 1;
  const float scale_y = (float)sh / (float)dh;
  // Lane rr evaluates the taps of dst row y_first + rr (slot rr mod P) and scatters them into the entries of its source
  // rows.  The host picks P with floor((rr + P) s) - floor(rr s) >= TAPS - 1: the windows of a slot's consecutive rows
  // overlap by ONE source row at most -- the last tap of row rr and the first of row rr + P (at 1.98:1 three slots, not
  // four, and the overlap happens every ~20 rows; at 4:3 four, not six).  That first weight goes to first_w and sets bit
  // 25: the walk completes and emits the old row, then starts the new one from +0 with it.  So every (slot, source row)
  // has one writer at most.  A zero weight is an exact no-op on an accumulator that is never -0, for the finite values
  // integer planes have; float planes skip the slot instead (bits 26 ..).
  const LzTap<TAPS> vy = lz_tap_of<TAPS>(ytab, min(r.y_first + min(lane, rows - 1), dh - 1), scale_y);
  r.s_begin = __builtin_amdgcn_readlane(vy.i, 0) - kBefore;
  r.ns = __builtin_amdgcn_readlane(vy.i, r.last_rr) + TAPS - kBefore - r.s_begin;
  u32* const ent = reinterpret_cast<u32*>(prog);
  float* const first_w = prog + NPROG * ESZ;
  r.first_w = first_w;
  r.lds = (u32)(uintptr_t)(__attribute__((address_space(3))) float*)prog;
#pragma unroll
  for (int t = lane; t < NPROG; t += kWave) {
    // (rows past the wave's last repeat it: the walk's prefetch runs D rows ahead, and rows that belong to the wave
    // below are long gone from the L2 when that wave starts on them)
    const u32 ctl = (u32)clampi(r.s_begin + min(t + D, r.ns - 1), sh - 1);
    if constexpr (ESZ == 4) {
      *reinterpret_cast<uint4*>(ent + 4 * t) = make_uint4(0u, 0u, 0u, ctl);
    } else {
      *reinterpret_cast<uint4*>(ent + 8 * t) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(ent + 8 * t + 4) = make_uint4(0u, 0u, ctl, 0u);
    }
    first_w[t] = 0.0f;
  }
  wave_lds_sync();
  // AGE of a dst row at a source row: how many of the rows before it are still being accumulated there (rows complete
  // in order, one per source row at most, so those are the a rows right before it).  Column a of an entry holds the
  // weight of the row of age a: the row that completes is ALWAYS column 0 -- the walk emits its first accumulator set and
  // moves the others down one place, and nothing in it depends on which dst row it is (a choice among P slots by
  // compare and branch cost more scalar instructions per dst row than the whole row pass has vector ones).
  const int t0 = vy.i - kBefore - r.s_begin;
  const int td = t0 + TAPS - 1;                                  // the source row that completes this lane's dst row
  int age[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k)
    age[k] = 0;
  {
    int prev = td;
#pragma unroll
    for (int m = 1; m <= P; ++m) {
      prev = __builtin_amdgcn_update_dpp(prev, prev, 0x138, 0xf, 0xf, false); // lane l: td of lane l - m
#pragma unroll
      for (int k = 0; k < TAPS; ++k)
        age[k] += (lane >= m && prev >= t0 + k) ? 1 : 0;
    }
  }
  const bool enters = age[0] == P; // (the oldest of the P rows before this one completes on this row's first source row)
  if constexpr (BYSLOT) {
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
      age[k] = lane % P;
  }
  if (lane <= r.last_rr) {
    constexpr int CW = ESZ == 4 ? 3 : 6;                         // dword of the control word
    // age P: the P rows before this one are all still open at its first source row -- the oldest of them completes
    // there (the host's P admits nothing else): this row starts when that one has left
    if (enters) {
      first_w[t0] = vy.w[0];
      __hip_atomic_fetch_or(ent + ESZ * t0 + CW, kProgDbl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    } else {
      prog[ESZ * t0 + age[0]] = vy.w[0];
    }
#pragma unroll
    for (int k = 1; k < TAPS; ++k)
      prog[ESZ * (t0 + k) + age[k]] = vy.w[k];
    if constexpr (ACT) {
#pragma unroll
      for (int k = 0; k < TAPS; ++k)
        if (k > 0 || !enters)
          __hip_atomic_fetch_or(ent + ESZ * (t0 + k) + CW + 1, 1u << age[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    // (at most one dst row completes per source row: scale_y >= 1)
    const u32 fin = kProgDone | (((lane & 1) != 0 || lane == r.last_rr) ? kProgPair : 0u) | (((lane & 1) == 0 && lane == r.last_rr) ? kProgSingle : 0u);
    __hip_atomic_fetch_or(ent + ESZ * (t0 + TAPS - 1) + CW, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  wave_lds_sync();
  return true;
}

// (w[H], w[H]) * f + acc in one instruction: the weight is one half of a VGPR pair, broadcast by op_sel
template <int H> __device__ __forceinline__ void pk_fma_h(v2f32& acc, v2f32 w, v2f32 f) {
  if constexpr (H == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(f));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(f));
}

// The walk over the wave's source rows.  A lane loads ND dwords per row at sp + row * pitch + lane_off (D rows in
// flight), conv() turns them into the 2 NF floats it filters, every slot takes fma(w, f, acc) with the weight of its
// program entry, and when a row completes a dst row take(pair, rr, c) gets that row's column results -- pair: the row
// closes a pair of dst rows (program bit 26).
// The completion block of the BYSLOT walks: set `slot` out (a copy to c) and restarted in place with w f + 0 -- one block of
// assembly, scalar branches included (see cols_walk).  NF = 4, integer planes.
template <int P>
__device__ __forceinline__ void cols_fin(v2f32 (&acc)[P][4], v2f32 (&c)[4], v2f32 w2, const v2f32 (&f)[4], int slot) {
  static_assert(P <= 6, "slots");
#define VALI_FIN1(j)                                                                                                    \
  "v_mov_b64 %[c0], %[a" #j "0]\n\tv_mov_b64 %[c1], %[a" #j "1]\n\tv_mov_b64 %[c2], %[a" #j "2]\n\tv_mov_b64 %[c3], %[a" #j "3]\n\t" \
  "v_pk_fma_f32 %[a" #j "0], %[w], %[f0], 0 op_sel_hi:[0,1,0]\n\tv_pk_fma_f32 %[a" #j "1], %[w], %[f1], 0 op_sel_hi:[0,1,0]\n\t"   \
  "v_pk_fma_f32 %[a" #j "2], %[w], %[f2], 0 op_sel_hi:[0,1,0]\n\tv_pk_fma_f32 %[a" #j "3], %[w], %[f3], 0 op_sel_hi:[0,1,0]\n\t"
#define VALI_FIN_OUT(j) [a##j##0] "+v"(acc[j][0]), [a##j##1] "+v"(acc[j][1]), [a##j##2] "+v"(acc[j][2]), [a##j##3] "+v"(acc[j][3])
#define VALI_FIN_IN [w] "v"(w2), [f0] "v"(f[0]), [f1] "v"(f[1]), [f2] "v"(f[2]), [f3] "v"(f[3]), [s] "s"(slot)
#define VALI_FIN_C [c0] "=&v"(c[0]), [c1] "=&v"(c[1]), [c2] "=&v"(c[2]), [c3] "=&v"(c[3])
  if constexpr (P == 2) {
    asm volatile("s_cmp_eq_u32 %[s], 0\n\ts_cbranch_scc1 0f\n\t" VALI_FIN1(1) "s_branch 9f\n0:\n\t" VALI_FIN1(0) "9:"
                 : VALI_FIN_C, VALI_FIN_OUT(0), VALI_FIN_OUT(1) : VALI_FIN_IN : "scc");
  } else if constexpr (P == 3) {
    asm volatile("s_cmp_eq_u32 %[s], 0\n\ts_cbranch_scc1 0f\n\ts_cmp_eq_u32 %[s], 1\n\ts_cbranch_scc1 1f\n\t" VALI_FIN1(2)
                 "s_branch 9f\n1:\n\t" VALI_FIN1(1) "s_branch 9f\n0:\n\t" VALI_FIN1(0) "9:"
                 : VALI_FIN_C, VALI_FIN_OUT(0), VALI_FIN_OUT(1), VALI_FIN_OUT(2) : VALI_FIN_IN : "scc");
  } else if constexpr (P == 4) {
    asm volatile("s_cmp_eq_u32 %[s], 0\n\ts_cbranch_scc1 0f\n\ts_cmp_eq_u32 %[s], 1\n\ts_cbranch_scc1 1f\n\t"
                 "s_cmp_eq_u32 %[s], 2\n\ts_cbranch_scc1 2f\n\t" VALI_FIN1(3) "s_branch 9f\n2:\n\t" VALI_FIN1(2)
                 "s_branch 9f\n1:\n\t" VALI_FIN1(1) "s_branch 9f\n0:\n\t" VALI_FIN1(0) "9:"
                 : VALI_FIN_C, VALI_FIN_OUT(0), VALI_FIN_OUT(1), VALI_FIN_OUT(2), VALI_FIN_OUT(3) : VALI_FIN_IN : "scc");
  } else {
    // (30 operands at most: the six sets in two blocks of three)
    if (slot < 3) {
      asm volatile("s_cmp_eq_u32 %[s], 0\n\ts_cbranch_scc1 0f\n\ts_cmp_eq_u32 %[s], 1\n\ts_cbranch_scc1 1f\n\t" VALI_FIN1(2)
                   "s_branch 9f\n1:\n\t" VALI_FIN1(1) "s_branch 9f\n0:\n\t" VALI_FIN1(0) "9:"
                   : VALI_FIN_C, VALI_FIN_OUT(0), VALI_FIN_OUT(1), VALI_FIN_OUT(2) : VALI_FIN_IN : "scc");
    } else {
      asm volatile("s_cmp_eq_u32 %[s], 3\n\ts_cbranch_scc1 0f\n\ts_cmp_eq_u32 %[s], 4\n\ts_cbranch_scc1 1f\n\t" VALI_FIN1(2)
                   "s_branch 9f\n1:\n\t" VALI_FIN1(1) "s_branch 9f\n0:\n\t" VALI_FIN1(0) "9:"
                   : VALI_FIN_C, [a00] "+v"(acc[P > 3 ? 3 : 0][0]), [a01] "+v"(acc[P > 3 ? 3 : 0][1]), [a02] "+v"(acc[P > 3 ? 3 : 0][2]),
                     [a03] "+v"(acc[P > 3 ? 3 : 0][3]), [a10] "+v"(acc[P > 4 ? 4 : 0][0]), [a11] "+v"(acc[P > 4 ? 4 : 0][1]),
                     [a12] "+v"(acc[P > 4 ? 4 : 0][2]), [a13] "+v"(acc[P > 4 ? 4 : 0][3]), [a20] "+v"(acc[P - 1][0]),
                     [a21] "+v"(acc[P - 1][1]), [a22] "+v"(acc[P - 1][2]), [a23] "+v"(acc[P - 1][3])
                   : VALI_FIN_IN : "scc");
    }
  }
#undef VALI_FIN1
#undef VALI_FIN_OUT
#undef VALI_FIN_IN
#undef VALI_FIN_C
}

struct NoFlush {
  __device__ __forceinline__ void operator()(bool) const {}
};
template <typename T, int TAPS, int P, int ND, int D, int NF = 4, bool BYSLOT = false, typename Conv, typename Take, typename Flush = NoFlush>
__device__ __forceinline__ void cols_walk(const ColProg& r, const uint8_t* sp, int spitch, int sh, u32 lane_off,
                                          Conv conv, Take take, Flush flush = Flush()) {
  constexpr int EB = (int)sizeof(T);
  constexpr bool ACT = EB == 4;
  constexpr int ESZ = kProgEsz<P, ACT>;
  typedef __attribute__((address_space(3))) const v4f32 lds_v4;
  v2f32 acc[P][NF];
#pragma unroll
  for (int j = 0; j < P; ++j)
#pragma unroll
    for (int i = 0; i < NF; ++i)
      acc[j][i] = (v2f32){0.0f, 0.0f};
  u32 pf[D][ND];
  // A row's loads through a raw buffer descriptor over the plane: address = base + row offset (a SCALAR operand: one s_mul
  // per row) + the lane's offset (the one VGPR of the address) -- no vector instruction per row.  (global_load would do with
  // its scalar base, but the compiler adds the row offset to a 64-bit VECTOR sp + lane_off instead: a v_lshl_add_u64 per row.)
  // Planes stay below 4 GiB (planes_fit_32bit, common.hpp); nothing past a row's last element is ever addressed.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sp), (short)0, (int)0xffffffffu, 0x00020000);
  // (row_off: the scalar offset operand; in the loop it is 0 and the row sits in lane_off: v_mad_u32_u24(control word, pitch,
  // lane offset) -- ONE vector instruction, which also ignores the flag bits above bit 23, instead of s_and + s_mul)
  auto issue = [&](u32 row_off, u32 (&q)[ND], u32 lane_off) {
    if constexpr (ND == 2) {
      const v2u32 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)row_off, 0);
      q[0] = w.x; q[1] = w.y;
    } else if constexpr (ND == 3) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)lane_off, (int)row_off, 0);
      q[0] = w.x; q[1] = w.y; q[2] = w.z;
    } else if constexpr (ND == 6) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w0 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)lane_off, (int)row_off, 0);
      const v3u32 w1 = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)lane_off + 12, (int)row_off, 0);
      q[0] = w0.x; q[1] = w0.y; q[2] = w0.z; q[3] = w1.x; q[4] = w1.y; q[5] = w1.z;
    } else {
#pragma unroll
      for (int c = 0; c < ND / 4; ++c) {
        const v4u32 w = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off + 16 * c, (int)row_off, 0);
        q[4 * c] = w.x; q[4 * c + 1] = w.y; q[4 * c + 2] = w.z; q[4 * c + 3] = w.w;
      }
    }
  };
  // (scheduling barriers: vmcnt retires in order, the rows must be ISSUED in order -- DESIGN.md 5d)
#pragma unroll
  for (int j = 0; j < D; ++j) {
    issue((u32)(clampi(r.s_begin + min(j, r.ns - 1), sh - 1) * spitch), pf[j], lane_off);
    __builtin_amdgcn_sched_barrier(0);
  }
  u32 vprog = r.lds;
  asm volatile("" : "+v"(vprog)); // (a VGPR: entry t0 + d is then an offset field, not a scalar add and a copy per row)
  // (entry 0 by the same assembly as the others: a compiler-visible first load makes the compiler wait for it in EVERY row, in
  // front of the assembly's own wait)
  v4f32 e0, e1 = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};
  asm volatile("ds_read_b128 %0, %1" : "=v"(e0) : "v"(vprog));
  if constexpr (ESZ == 8)
    asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(e1) : "v"(vprog));
  vprog += (u32)(ESZ * 4); // (entry t0 + 1: the entries the trip fetches sit at offsets >= 0 -- ds offsets are unsigned, and behind
  asm volatile("" : "+v"(vprog)); // entry t0 the compiler subtracted a constant from the advanced pointer for every row)
  int emit_rr = 0; // the next dst row to complete
  v2f32 cc[NF];    // BYSLOT: the column results of the row that completes
#pragma unroll 1
  for (int t0 = 0; t0 < r.ns; t0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int t = t0 + d;
      // (the wait for this row's entry in FRONT of the conversions: between the assembly and the v_readfirstlane of the
      // control word the compiler wants a wait state, and finds the conversions there instead of adding an s_nop)
      if constexpr (ESZ == 8)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0), "+v"(e1));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0));
      __builtin_amdgcn_sched_barrier(0);
      v2f32 f[NF];
      conv(pf[d], f);
      // (the load after the conversions, into the registers they just freed: hoisted above them it gets new registers,
      // which the compiler copies back at the end of every trip -- behind a wait for ALL loads in flight)
#pragma unroll
      for (int i = 0; i < NF; ++i)
        asm volatile("" : "+v"(f[i])); // (pins the conversions in front of the barrier: they have no other order)
      __builtin_amdgcn_sched_barrier(0);
      // (__float_as_uint of a vector element: __builtin_bit_cast applied to the element lvalue e1.w read element 0)
      const u32 ctl = __float_as_uint(ESZ == 4 ? e0.w : e1.z);
      const u32 actv = __float_as_uint(e1.w);
      const u32 flags = (u32)__builtin_amdgcn_readfirstlane((int)ctl);
      issue(0u, pf[d], __umul24(ctl, (u32)spitch) + lane_off);
      const v4f32 c0 = e0, c1 = e1;
      // the next row's entry: in flight under this row's arithmetic
      // (assembly: the offset is an immediate of the instruction -- the compiler advanced a second pointer by v_add for every
      // row instead; the wait for it is the lgkmcnt(0) in front of the next row's control word, also assembly: the compiler does
      // not count these reads, which only ever makes its own waits longer -- LDS instructions return in order)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e0) : "v"(vprog), "i"(d * ESZ * 4));
      if constexpr (ESZ == 8)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e1) : "v"(vprog), "i"(d * ESZ * 4 + 16));
      __builtin_amdgcn_sched_barrier(0);
      if (t >= r.ns) // the last trip only: rows past the end are loaded (the issue order stays countable), not used
        continue;
      const v2f32 w01 = (v2f32){c0.x, c0.y}, w23 = (v2f32){c0.z, c0.w}, w45 = (v2f32){c1.x, c1.y};
      u32 act = 0u;
      if constexpr (ACT)
        act = (u32)__builtin_amdgcn_readfirstlane((int)actv);
#pragma unroll
      for (int j = 0; j < P; ++j) {
        if constexpr (ACT) {
          if (((act >> j) & 1u) == 0u) // wave-uniform
            continue;
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          if (j == 0) pk_fma_h<0>(acc[j][i], w01, f[i]);
          else if (j == 1) pk_fma_h<1>(acc[j][i], w01, f[i]);
          else if (j == 2) pk_fma_h<0>(acc[j][i], w23, f[i]);
          else if (j == 3) pk_fma_h<1>(acc[j][i], w23, f[i]);
          else if (j == 4) pk_fma_h<0>(acc[j][i], w45, f[i]);
          else pk_fma_h<1>(acc[j][i], w45, f[i]);
        }
      }
      if constexpr (BYSLOT) {
        if (flags & kProgDone) { // at most one dst row per source row (scale_y >= 1): the row of slot emit_rr mod P
          // One scalar switch over the slot: the set out (a copy) and its restart IN PLACE: w f + 0 with the first weight of
          // the slot's next row when this source row is also its first tap, with w = 0 otherwise (+0 for the finite values
          // of integer planes; float planes restart with +0 itself).  ONE block of assembly, branches included: written in
          // C++ -- with or without tied operands -- the compiler turns the switch into selects over all P sets or copies
          // whole sets at its joins (r04: version G), 16-24 moves per dst row instead of 4.
          // (the slot: three bits of the control word; the first weight of the row that enters: read AND waited for by one block of
          // assembly inside the branch -- a compiler-visible LDS read makes every completion wait for the entry prefetch as well)
          const int slot = (int)((flags >> kProgSlotShift) & 7u);
          v2f32 w2 = (v2f32){0.0f, 0.0f};
          if (flags & kProgDbl) {
            float wx;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(wx) : "v"((u32)(uintptr_t)(__attribute__((address_space(3))) const float*)(r.first_w + t)));
            w2.x = wx;
          }
          auto fin_to = [&](v2f32 (&c)[NF]) {
          if constexpr (NF == 4 && !ACT && (P == 2 || P == 3 || P == 4 || P == 6)) {
            cols_fin<P>(acc, c, w2, f, slot);
          } else {
            auto fin = [&](v2f32 (&a)[NF]) {
#pragma unroll
              for (int i = 0; i < NF; ++i)
                c[i] = a[i];
              if (flags & kProgDbl) {
#pragma unroll
                for (int i = 0; i < NF; ++i)
                  a[i] = __builtin_elementwise_fma((v2f32){w2.x, w2.x}, f[i], (v2f32){0.0f, 0.0f});
              } else {
#pragma unroll
                for (int i = 0; i < NF; ++i)
                  a[i] = (v2f32){0.0f, 0.0f};
              }
            };
            if (slot == 0) fin(acc[0]);
            else if (P > 1 && slot == 1) fin(acc[P > 1 ? 1 : 0]);
            else if (P > 2 && slot == 2) fin(acc[P > 2 ? 2 : 0]);
            else if (P > 3 && slot == 3) fin(acc[P > 3 ? 3 : 0]);
            else if (P > 4 && slot == 4) fin(acc[P > 4 ? 4 : 0]);
            else fin(acc[P - 1]);
          }
          };
          // (ONE instance of the block: a second one in the other arm of a branch -- the pair's first row straight to its
          // own registers -- brings the copies at the joins back, and spills)
          fin_to(cc);
          take(flags, cc);
        }
        continue;
      }
      if constexpr (!BYSLOT) {
      if (flags & kProgDone) { // at most one dst row per source row (scale_y >= 1): the row of age 0
        take((flags & kProgPair) != 0u, emit_rr, acc[0]);
#pragma unroll
        for (int j = 0; j + 1 < P; ++j)
#pragma unroll
          for (int i = 0; i < NF; ++i)
            acc[j][i] = acc[j + 1][i];
        if (flags & kProgDbl) { // this source row is also the first tap of the row that enters (age P - 1 from here on)
          const float w2 = r.first_w[t];
#pragma unroll
          for (int i = 0; i < NF; ++i)
            acc[P - 1][i] = __builtin_elementwise_fma((v2f32){w2, w2}, f[i], (v2f32){0.0f, 0.0f});
        } else {
#pragma unroll
          for (int i = 0; i < NF; ++i)
            acc[P - 1][i] = (v2f32){0.0f, 0.0f};
        }
        ++emit_rr;
      }
      }
    }
    vprog += (u32)(D * ESZ * 4);
  }
}

template <typename T> __device__ __forceinline__ void conv8(const u32 (&d)[2 * sizeof(T)], v2f32 (&f)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if constexpr (sizeof(T) == 1)
      f[i] = (v2f32){(float)((d[i / 2] >> (16 * (i % 2))) & 0xffu), (float)((d[i / 2] >> (16 * (i % 2) + 8)) & 0xffu)};
    else if constexpr (sizeof(T) == 2)
      f[i] = (v2f32){(float)(d[i] & 0xffffu), (float)(d[i] >> 16)};
    else
      f[i] = (v2f32){__uint_as_float(d[2 * i]), __uint_as_float(d[2 * i + 1])};
  }
}

// slot of strip position q: even positions in the front half of a channel segment, odd ones in the back half
template <int ES> __device__ __forceinline__ int col_slot(int q) { return (q >> 1) + (q & 1) * kColHalf<ES>; }

template <typename T, int ES, int TAPS, int P>
__device__ __forceinline__ void cols_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                          int dw, int dh, u32 tx, u32 ty, int N, int rps, float* lds) {
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int EB = (int)sizeof(T);
  constexpr int ND = 2 * EB;                                    // dwords of a lane's 8 elements
  constexpr int D = EB == 4 ? 3 : EB == 2 ? 3 : P >= 4 ? 2 : 4; // source rows in flight (registers: 8 EB bytes per lane and row; the 4- and 6-slot kernels trade two rows for their fourth wave per SIMD)
  constexpr int HALF = kColHalf<ES>, SEG = 2 * HALF;            // slots
  const int lane = threadIdx.x & 63;
  v2f32* const strip = reinterpret_cast<v2f32*>(lds);           // one slot = one column of TWO dst rows
  v2f32* const obuf = strip + kColStrip;
  const u32 lds_base = (u32)(uintptr_t)(__attribute__((address_space(3))) float*)lds; // LDS byte address of the strip
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dwe = dw * ES, row_el = sw * ES;
  const int e0 = (int)tx * N, e_last = min(e0 + N, dwe) - 1;
  const float scale_x = (float)sw / (float)dw;

  // ---- column taps: a lane filters elements lane, lane + 64, lane + 128, lane + 192 of the tile.  The four waves of the
  // workgroup work on the same tile columns (rows ty * 4 + wave): wave w evaluates the w-th set, all four read them back --
  // a tap set is ~130 instructions, four of them per wave were a quarter of the kernel's vector instructions ----
  v2f32 wq[4][TAPS / 2]; // (w0, w1), (w2, w3), ..
  int ci[4];
  if constexpr (EB == 4) { // float planes are bound by their memory stream: no workgroup barriers in front of the walk
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const LzTap<TAPS> c = make_lz_tap<TAPS>(min(e0 + p * kWave + lane, e_last) / ES, scale_x);
#pragma unroll
      for (int k = 0; k < TAPS / 2; ++k)
        wq[p][k] = (v2f32){c.w[2 * k], c.w[2 * k + 1]};
      ci[p] = c.i;
    }
  } else {
    const LzTap<TAPS> c = make_lz_tap<TAPS>(min(e0 + wave * kWave + lane, e_last) / ES, scale_x);
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
      lds[k * kWave + lane] = c.w[k];
    reinterpret_cast<int*>(lds)[TAPS * kWave + lane] = c.i;
    __syncthreads();
    const float* const all = lds - wave * kColWave;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int k = 0; k < TAPS / 2; ++k)
        wq[p][k] = (v2f32){all[p * kColWave + 2 * k * kWave + lane], all[p * kColWave + (2 * k + 1) * kWave + lane]};
      ci[p] = reinterpret_cast<const int*>(all)[p * kColWave + TAPS * kWave + lane];
    }
    __syncthreads();
  }
  ColProg r;
  if (!cols_rows<TAPS, P, EB == 4, D>(sh, dh, ty * kWavesPerBlock + (u32)wave, rps, lds + kColLds, r))
    return;

  // ---- the tile's source span along x (wave-uniform) ----
  const int px_first = e0 / ES, px_last = e_last / ES;
  const int ux0 = (int)__builtin_floorf((float)px_first * scale_x) - kBefore - 1;
  const int ux1 = (int)__builtin_floorf((float)px_last * scale_x) + TAPS + 1 - kBefore;
  const int sx0 = clampi(ux0, sw - 1), sx1 = clampi(ux1, sw - 1);
  const int j_begin = (sx0 * ES) & ~(kColEl - 1);               // first element the wave loads
  const int px_begin = j_begin / ES;
  const int nl = min((((sx1 + 1) * ES - j_begin) + kColEl - 1) / kColEl, kWave); // lanes with data
  // the last chunk of a row slides left to END with the row: no byte outside the row is ever read (borrowed surfaces
  // end where their last row ends); the elements it shares with its neighbour are written twice with the same value
  const bool ragged = j_begin + kColEl * nl > row_el;           // wave-uniform: only a row's last tile
  const int j0 = min(j_begin + kColEl * min(lane, nl - 1), row_el - kColEl);

  // strip slots of this lane's 8 elements.  1- and 2-element pixels: its pixels are consecutive positions q0 .. -- those
  // of q0's parity are one run of adjacent slots (wpos[0] ..), the others another (wpos[1] ..); q0 % 4 == 0 unless the
  // chunk slid (kColPadL = 4): the runs then start on 16 bytes
  int wpos[ES == 3 ? kColEl : 2];
  if constexpr (ES == 3) {
#pragma unroll
    for (int q = 0; q < kColEl; ++q) {
      const int j = j0 + q, px = j / 3;
      wpos[q] = (j - px * 3) * SEG + col_slot<ES>(kColPadL + px - px_begin);
    }
  } else {
    const int q0 = kColPadL + j0 / ES - px_begin;
    wpos[0] = col_slot<ES>(q0);
    wpos[1] = col_slot<ES>(q0 + 1);
  }

  // ---- horizontal pass set-up: the strip slots of the even and odd taps of this lane's 4 elements ----
  u32 ha[4][2]; // LDS byte addresses: taps 0, 2, 4 at ha[p][0] + 0, 8, 16; taps 1, 3, 5 at ha[p][1] + 0, 8, 16
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = min(e0 + p * kWave + lane, e_last);
    const int px = e / ES, ch = e - px * ES;
    const int q = kColPadL + (min(ci[p], sw - 1) - kBefore - px_begin); // >= kColPadL - kBefore
    ha[p][0] = lds_base + 8u * (u32)(ch * SEG + col_slot<ES>(q));
    ha[p][1] = lds_base + 8u * (u32)(ch * SEG + col_slot<ES>(q + 1));
  }
  const bool pad_left = ux0 < 0, pad_right = ux1 > sw - 1;                 // wave-uniform
  const int edge = kColPadL + (sw - 1) - px_begin;                         // the last pixel of the row
  const int eb = e0 + 4 * lane;                                            // store: 4 adjacent elements
  const int n_out = min(4, e_last + 1 - eb);
  // whole dwords from every lane that stores, on 4-byte aligned rows: one store, no per-lane alignment test
  const bool plain_store = EB == 1 && ((e_last + 1 - e0) & 3) == 0 && ((((uintptr_t)dp) | (uintptr_t)dpitch) & 3u) == 0; // wave-uniform

  auto store_row = [&](int rr, float v0, float v1, float v2, float v3) {
    uint8_t* const out = dp + (u32)((r.y_first + rr) * dpitch) + (size_t)eb * EB;
    if (plain_store) {
      u32 q = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0u, 0u);
      q = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1u, q);
      q = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2u, q);
      q = __builtin_amdgcn_cvt_pk_u8_f32(v3, 3u, q);
      gstore_nt<u32>(out, q);
    } else {
      const float res[4][1] = {{v0}, {v1}, {v2}, {v3}};
      store_px4<T, 1>(out, res, (1u << n_out) - 1u);
    }
  };

  // The pass along the rows takes dst rows in PAIRS: the columns of row rr (even) wait in registers until row rr + 1
  // completes, then a strip slot holds one column of BOTH rows, so a tap is one ds_read_b64 + one v_pk_fma_f32 whose
  // weight is the same for both halves -- six of each per two output samples, at any window start (the first r03 form
  // read aligned float pairs of ONE row: eight of each, and an add).  A wave's last row may be single: it runs as a
  // pair with itself.  (Writing the halves of a slot as the rows complete, ds_write_b32 at a stride of 8 dwords between
  // lanes, is an 8-way bank conflict per instruction: measured slower than the r03 form.)
  v2f32 hold[4];      // the columns of the pair's first row
  // the first row of a pair is kept, the second goes to the strip with it
  auto take = [&](bool pair, int rr, v2f32 (&c)[4]) {
    if (!pair) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        hold[i] = c[i];
      return;
    }
    v2f32 lo[4], hi[4]; // (row a, row b) of this lane's columns 2 i / 2 i + 1 (ES = 2: U / V of pixel i)
    if ((rr & 1) == 0) { // the wave's last row is the first of a pair: the pair is the row twice
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = pk_dup<0>(c[i]);
        hi[i] = pk_dup<1>(c[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = pk_mov<0, 0>(hold[i], c[i]);
        hi[i] = pk_mov<1, 1>(hold[i], c[i]);
      }
    }
    // (lanes past the tile's last chunk repeat it -- j0 -- with the same data: no predicate)
    {
      if constexpr (ES == 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          strip[wpos[2 * i]] = lo[i];
          strip[wpos[2 * i + 1]] = hi[i];
        }
      } else if (ragged) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (ES == 2) { // pixel i: (U, V); pixels 0, 2 in the run of wpos[0], 1, 3 in that of wpos[1]
            strip[wpos[i & 1] + (i >> 1)] = lo[i];
            strip[SEG + wpos[i & 1] + (i >> 1)] = hi[i];
          } else {                 // pixels 2 i (run of wpos[0]) and 2 i + 1 (run of wpos[1])
            strip[wpos[0] + i] = lo[i];
            strip[wpos[1] + i] = hi[i];
          }
        }
      } else if constexpr (ES == 2) {
        *reinterpret_cast<float4*>(strip + wpos[0]) = make_float4(lo[0].x, lo[0].y, lo[2].x, lo[2].y);
        *reinterpret_cast<float4*>(strip + wpos[1]) = make_float4(lo[1].x, lo[1].y, lo[3].x, lo[3].y);
        *reinterpret_cast<float4*>(strip + SEG + wpos[0]) = make_float4(hi[0].x, hi[0].y, hi[2].x, hi[2].y);
        *reinterpret_cast<float4*>(strip + SEG + wpos[1]) = make_float4(hi[1].x, hi[1].y, hi[3].x, hi[3].y);
      } else {
        *reinterpret_cast<float4*>(strip + wpos[0]) = make_float4(lo[0].x, lo[0].y, lo[1].x, lo[1].y);
        *reinterpret_cast<float4*>(strip + wpos[0] + 2) = make_float4(lo[2].x, lo[2].y, lo[3].x, lo[3].y);
        *reinterpret_cast<float4*>(strip + wpos[1]) = make_float4(hi[0].x, hi[0].y, hi[1].x, hi[1].y);
        *reinterpret_cast<float4*>(strip + wpos[1] + 2) = make_float4(hi[2].x, hi[2].y, hi[3].x, hi[3].y);
      }
    }
    wave_lds_sync();
    if (pad_left || pad_right) { // image edges: replicas of the first / last pixel of every channel segment
      const int ch = lane >> 3, i = lane & 7;
      if (pad_left && ch < ES && i < kColPadL)
        strip[ch * SEG + col_slot<ES>(kColPadL - 1 - i)] = strip[ch * SEG + col_slot<ES>(kColPadL)];
      if (pad_right && ch < ES && i < kColPadR)
        strip[ch * SEG + col_slot<ES>(edge + 1 + i)] = strip[ch * SEG + col_slot<ES>(edge)];
      wave_lds_sync();
    }
    // Two windows per LDS round trip, every slot with its own ds_read_b64: left to itself the compiler fuses two into a
    // ds_read2_b64, which takes twice the LDS cycles per byte and banks modulo 32 dwords instead of 64
    // (MI355X_MICROARCH.md, LDS).  Hence the assembly; the wait names every loaded register, so nothing reads one early.
    // Two windows per block of assembly: their ds_read_b64s (every slot with its own: left to itself the compiler fuses
    // two into a ds_read2_b64, four times the LDS time -- tools/exp/lds_patterns.hip), the wait for the first window
    // only, its chains (specification order: e over the even taps, o over the odd ones, e + o; one half of a weight
    // pair for both halves of the result by op_sel) while the second window's reads are still in flight.  One block: the
    // compiler pads every boundary between assembly and its own code with an s_nop.
#pragma unroll
    for (int half = 0; half < 4; half += 2) {
      v2f32 t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4, u5;
      if constexpr (TAPS == 6) {
        asm volatile(
            "ds_read_b64 %[t0], %[a0]\n\tds_read_b64 %[t1], %[a1]\n\tds_read_b64 %[t2], %[a0] offset:8\n\t"
            "ds_read_b64 %[t3], %[a1] offset:8\n\tds_read_b64 %[t4], %[a0] offset:16\n\tds_read_b64 %[t5], %[a1] offset:16\n\t"
            "ds_read_b64 %[u0], %[b0]\n\tds_read_b64 %[u1], %[b1]\n\tds_read_b64 %[u2], %[b0] offset:8\n\t"
            "ds_read_b64 %[u3], %[b1] offset:8\n\tds_read_b64 %[u4], %[b0] offset:16\n\tds_read_b64 %[u5], %[b1] offset:16\n\t"
            "s_waitcnt lgkmcnt(6)\n\t"
            "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
            "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
            "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "v_pk_fma_f32 %[t0], %[w2], %[t4], %[t0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[t1], %[w2], %[t5], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_fma_f32 %[u0], %[x0], %[u0], 0 op_sel_hi:[0,1,0]\n\t"
            "v_pk_fma_f32 %[u1], %[x0], %[u1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
            "v_pk_add_f32 %[t0], %[t0], %[t1]\n\t"
            "v_pk_fma_f32 %[u0], %[x1], %[u2], %[u0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[u1], %[x1], %[u3], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "v_pk_fma_f32 %[u0], %[x2], %[u4], %[u0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[u1], %[x2], %[u5], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %[u0], %[u0], %[u1]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3),
              [u4] "=&v"(u4), [u5] "=&v"(u5)
            : [a0] "v"(ha[half][0]), [a1] "v"(ha[half][1]), [b0] "v"(ha[half + 1][0]), [b1] "v"(ha[half + 1][1]),
              [w0] "v"(wq[half][0]), [w1] "v"(wq[half][1]), [w2] "v"(wq[half][TAPS / 2 - 1]), [x0] "v"(wq[half + 1][0]),
              [x1] "v"(wq[half + 1][1]), [x2] "v"(wq[half + 1][TAPS / 2 - 1])
            : "memory");
      } else {
        asm volatile(
            "ds_read_b64 %[t0], %[a0]\n\tds_read_b64 %[t1], %[a1]\n\tds_read_b64 %[t2], %[a0] offset:8\n\t"
            "ds_read_b64 %[t3], %[a1] offset:8\n\t"
            "ds_read_b64 %[u0], %[b0]\n\tds_read_b64 %[u1], %[b1]\n\tds_read_b64 %[u2], %[b0] offset:8\n\t"
            "ds_read_b64 %[u3], %[b1] offset:8\n\t"
            "s_waitcnt lgkmcnt(4)\n\t"
            "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
            "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
            "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_pk_fma_f32 %[u0], %[x0], %[u0], 0 op_sel_hi:[0,1,0]\n\t"
            "v_pk_fma_f32 %[u1], %[x0], %[u1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
            "v_pk_add_f32 %[t0], %[t0], %[t1]\n\t"
            "v_pk_fma_f32 %[u0], %[x1], %[u2], %[u0] op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %[u1], %[x1], %[u3], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
            "s_nop 0\n\t"
            "v_pk_add_f32 %[u0], %[u0], %[u1]"
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
              [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3)
            : [a0] "v"(ha[half][0]), [a1] "v"(ha[half][1]), [b0] "v"(ha[half + 1][0]), [b1] "v"(ha[half + 1][1]),
              [w0] "v"(wq[half][0]), [w1] "v"(wq[half][1]), [x0] "v"(wq[half + 1][0]), [x1] "v"(wq[half + 1][1])
            : "memory");
      }
      obuf[half * kWave + lane] = t0;       // (the chains accumulate in the registers of their first taps)
      obuf[(half + 1) * kWave + lane] = u0;
    }
    wave_lds_sync();
    if (n_out > 0) {
      // (two ds_read_b128: the compiler reads the halves it uses as four ds_read2_b32 at a lane stride of 32 bytes --
      // 8 LDS slots each instead of 2, tools/exp/lds_patterns.hip)
      float4 v0, v1; // (a0, b0, a1, b1), (a2, b2, a3, b3)
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1) : "v"(lds_base + 8u * (u32)kColStrip + 32u * (u32)lane) : "memory");
      store_row(rr - (rr & 1), v0.x, v0.z, v1.x, v1.z); // (a last row on its own: twice)
      store_row(rr, v0.y, v0.w, v1.y, v1.w);
    }
    wave_lds_sync();
  };
  cols_walk<T, TAPS, P, ND, D>(r, sp, spitch, sh, (u32)(j0 * EB), [](const u32 (&d)[ND], v2f32 (&f)[4]) { conv8<T>(d, f); }, take);
}

// ---- the general form on SPECIALISED waves (round 5) ----
// The form above runs both passes in ONE wave: a wave that has completed a pair of dst rows stops walking -- strip write,
// tap reads, chains, transposition write, read, quantise, store: three LDS round trips behind each other -- while its loads
// in flight land and wait; with 118-126 registers at most four such waves share a SIMD, and at any moment most of them wait
// (profiles/r04_lanczos.md, section 6).  Here a workgroup is TWO waves with one tile: the PRODUCER walks the source rows
// (loads, conversions, the scatter into the slots' accumulators -- cols_walk, unchanged) and writes every completed pair of
// dst rows into one of two strips; the CONSUMER runs the pass along the rows on the other strip (the same reads, chains,
// transposition and stores as above, bit for bit).  The producer never waits for an LDS result but its program entries, the
// consumer never for a global load; they meet at ONE s_barrier per pair:
//   producer   [B] W(0) walk [B] W(1) walk ... [B] W(n-1) walk-out [B]        W(k): strip k % 2
//   consumer   [B]      R(0) [B]      R(1) ...  [B]            R(n-1)
// at the barrier in front of W(k) the consumer has finished R(k - 2) (the strip W(k) overwrites) and the producer's W(k - 1)
// has landed (s_waitcnt lgkmcnt(0): by then long done -- the wait sits a whole pair behind the writes it waits for).

constexpr int kWsBlock = 2 * kWave;
constexpr int kWsStripBytes = kColStrip * 8;                  // one strip: kColStrip slots of two floats
constexpr int kWsProg = 2 * 2 * kColStrip;                    // floats: the producer's program, behind the two strips
// NS: sets of 64 dst elements a consumer lane filters -- 4 (256-element tiles), or 5: at ratios below 2:1 the 512 source
// elements a producer row loads cover up to 320 dst elements, and a 256-element tile leaves a quarter of its lanes without data
// (1080p -> 1278x718: 50 of 64).  The transposition of a pair's results (64 NS slots) has no LDS of its own: it goes into the
// head of the strip the results came from, once all of a pair's tap reads have returned -- the producer does not touch that
// strip before the next barrier.
// Program entries: 224 rows of 5 dwords, 160 of 9 (the kernels of 4 and 6 slots and of float planes run five or four waves per
// SIMD by their registers): 110 / 75 dst rows per tile at 2:1, 96 at 3:2.
template <int P, bool ACT> constexpr int kWsProgRows = kProgEsz<P, ACT> == 4 ? 224 : 160;
template <int P, bool ACT> constexpr int kWsLds = kWsProg + kWsProgRows<P, ACT> * (kProgEsz<P, ACT> + 1); // floats of a workgroup

__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Two windows of the pass along the rows: results in ra / rb (both rows of the pair).  OFF: byte offset of the strip.
template <int TAPS, int OFF>
__device__ __forceinline__ void ws_row_taps(v2f32& ra, v2f32& rb, u32 a0, u32 a1, u32 b0, u32 b1, const v2f32 (&wa)[TAPS / 2],
                                            const v2f32 (&wb)[TAPS / 2]) {
  v2f32 t0, t1, t2, t3, t4, t5, u0, u1, u2, u3, u4, u5;
  if constexpr (TAPS == 6) {
    asm volatile(
        "ds_read_b64 %[t0], %[a0] offset:%[o0]\n\tds_read_b64 %[t1], %[a1] offset:%[o0]\n\tds_read_b64 %[t2], %[a0] offset:%[o1]\n\t"
        "ds_read_b64 %[t3], %[a1] offset:%[o1]\n\tds_read_b64 %[t4], %[a0] offset:%[o2]\n\tds_read_b64 %[t5], %[a1] offset:%[o2]\n\t"
        "ds_read_b64 %[u0], %[b0] offset:%[o0]\n\tds_read_b64 %[u1], %[b1] offset:%[o0]\n\tds_read_b64 %[u2], %[b0] offset:%[o1]\n\t"
        "ds_read_b64 %[u3], %[b1] offset:%[o1]\n\tds_read_b64 %[u4], %[b0] offset:%[o2]\n\tds_read_b64 %[u5], %[b1] offset:%[o2]\n\t"
        "s_waitcnt lgkmcnt(6)\n\t"
        "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[t0], %[w2], %[t4], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w2], %[t5], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[u0], %[x0], %[u0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[u1], %[x0], %[u1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_pk_fma_f32 %[u0], %[x1], %[u2], %[u0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[u1], %[x1], %[u3], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[u0], %[x2], %[u4], %[u0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[u1], %[x2], %[u5], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %[u0], %[u0], %[u1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [u0] "=&v"(u0),
          [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), [u4] "=&v"(u4), [u5] "=&v"(u5)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [w0] "v"(wa[0]), [w1] "v"(wa[1]), [w2] "v"(wa[TAPS / 2 - 1]),
          [x0] "v"(wb[0]), [x1] "v"(wb[1]), [x2] "v"(wb[TAPS / 2 - 1]), [o0] "i"(OFF), [o1] "i"(OFF + 8), [o2] "i"(OFF + 16)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64 %[t0], %[a0] offset:%[o0]\n\tds_read_b64 %[t1], %[a1] offset:%[o0]\n\tds_read_b64 %[t2], %[a0] offset:%[o1]\n\t"
        "ds_read_b64 %[t3], %[a1] offset:%[o1]\n\t"
        "ds_read_b64 %[u0], %[b0] offset:%[o0]\n\tds_read_b64 %[u1], %[b1] offset:%[o0]\n\tds_read_b64 %[u2], %[b0] offset:%[o1]\n\t"
        "ds_read_b64 %[u3], %[b1] offset:%[o1]\n\t"
        "s_waitcnt lgkmcnt(4)\n\t"
        "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[u0], %[x0], %[u0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[u1], %[x0], %[u1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_add_f32 %[t0], %[t0], %[t1]\n\t"
        "v_pk_fma_f32 %[u0], %[x1], %[u2], %[u0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[u1], %[x1], %[u3], %[u1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %[u0], %[u0], %[u1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2),
          [u3] "=&v"(u3)
        : [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [w0] "v"(wa[0]), [w1] "v"(wa[1]), [x0] "v"(wb[0]),
          [x1] "v"(wb[1]), [o0] "i"(OFF), [o1] "i"(OFF + 8)
        : "memory");
    (void)t4; (void)t5; (void)u4; (void)u5;
  }
  ra = t0;
  rb = u0;
}

// The program of a specialised producer: cols_rows<BYSLOT> for up to 128 dst rows -- two rounds of "lane rr evaluates row rr" --
// without the exchange between lanes: what a row needs of the row P before it (does that one complete on my first source row?) is
// that row's tap index, a table entry or three instructions.  Twice the rows per tile: the TAPS - 1 source rows two tiles share
// are walked twice per 190 rows instead of per 95 (HBM traffic 1.08 -> 1.04 x at 2160p -> 1936x1088).
template <int TAPS, int P, bool ACT, int D, int NPROG>
__device__ __forceinline__ void ws_rows(int sh, int dh, int y_first, int last_rr, float* prog, ColProg& r, const float4* ytab) {
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int ESZ = kProgEsz<P, ACT>;
  constexpr int CW = ESZ == 4 ? 3 : 6;                          // dword of the control word
  const int lane = threadIdx.x & 63;
  const float scale_y = (float)sh / (float)dh;
  auto idx = [&](int y) { return ytab ? __float_as_int(ytab[2 * y + 1].z) : (int)__builtin_floorf((float)y * scale_y); }; // make_lz_tap's i
  r.y_first = y_first;
  r.last_rr = last_rr;
  r.s_begin = __builtin_amdgcn_readfirstlane(idx(y_first)) - kBefore;
  r.ns = __builtin_amdgcn_readfirstlane(idx(y_first + last_rr)) + TAPS - kBefore - r.s_begin;
  u32* const ent = reinterpret_cast<u32*>(prog);
  float* const first_w = prog + NPROG * ESZ;
  r.first_w = first_w;
  r.lds = (u32)(uintptr_t)(__attribute__((address_space(3))) float*)prog;
#pragma unroll
  for (int t = lane; t < NPROG; t += kWave) {
    const u32 ctl = (u32)clampi(r.s_begin + min(t + D, r.ns - 1), sh - 1); // (see cols_rows)
    if constexpr (ESZ == 4) {
      *reinterpret_cast<uint4*>(ent + 4 * t) = make_uint4(0u, 0u, 0u, ctl);
    } else {
      *reinterpret_cast<uint4*>(ent + 8 * t) = make_uint4(0u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(ent + 8 * t + 4) = make_uint4(0u, 0u, ctl, 0u);
    }
    first_w[t] = 0.0f;
  }
  wave_lds_sync();
#pragma unroll 1
  for (int base = 0; base <= last_rr; base += kWave) {          // (wave-uniform)
    const int rr = base + lane;
    if (rr <= last_rr) {
      const LzTap<TAPS> vy = lz_tap_of<TAPS>(ytab, y_first + rr, scale_y);
      const int t0 = vy.i - kBefore - r.s_begin;
      // the row P before this one is still open on this row's first source row, i.e. completes there (the host's P admits
      // nothing else): this row's first weight goes to first_w, the walk restarts the slot with it
      const bool enters = rr >= P && idx(y_first + rr - P) + TAPS - 1 - kBefore - r.s_begin >= t0;
      const int slot = rr % P;
      if (enters) {
        first_w[t0] = vy.w[0];
        __hip_atomic_fetch_or(ent + ESZ * t0 + CW, kProgDbl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      } else {
        prog[ESZ * t0 + slot] = vy.w[0];
      }
#pragma unroll
      for (int k = 1; k < TAPS; ++k)
        prog[ESZ * (t0 + k) + slot] = vy.w[k];
      if constexpr (ACT) {
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
          if (k > 0 || !enters)
            __hip_atomic_fetch_or(ent + ESZ * (t0 + k) + CW + 1, 1u << slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      const u32 fin = kProgDone | (((rr & 1) != 0 || rr == last_rr) ? kProgPair : 0u) | ((u32)slot << kProgSlotShift);
      __hip_atomic_fetch_or(ent + ESZ * (t0 + TAPS - 1) + CW, fin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
  }
  wave_lds_sync();
}

// One window (the fifth set of a wide tile).
template <int TAPS, int OFF>
__device__ __forceinline__ void ws_row_taps1(v2f32& ra, u32 a0, u32 a1, const v2f32 (&wa)[TAPS / 2]) {
  v2f32 t0, t1, t2, t3, t4, t5;
  if constexpr (TAPS == 6) {
    asm volatile(
        "ds_read_b64 %[t0], %[a0] offset:%[o0]\n\tds_read_b64 %[t1], %[a1] offset:%[o0]\n\tds_read_b64 %[t2], %[a0] offset:%[o1]\n\t"
        "ds_read_b64 %[t3], %[a1] offset:%[o1]\n\tds_read_b64 %[t4], %[a0] offset:%[o2]\n\tds_read_b64 %[t5], %[a1] offset:%[o2]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[t0], %[w2], %[t4], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w2], %[t5], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %[t0], %[t0], %[t1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5)
        : [a0] "v"(a0), [a1] "v"(a1), [w0] "v"(wa[0]), [w1] "v"(wa[1]), [w2] "v"(wa[TAPS / 2 - 1]), [o0] "i"(OFF), [o1] "i"(OFF + 8),
          [o2] "i"(OFF + 16)
        : "memory");
  } else {
    asm volatile(
        "ds_read_b64 %[t0], %[a0] offset:%[o0]\n\tds_read_b64 %[t1], %[a1] offset:%[o0]\n\tds_read_b64 %[t2], %[a0] offset:%[o1]\n\t"
        "ds_read_b64 %[t3], %[a1] offset:%[o1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_pk_fma_f32 %[t0], %[w0], %[t0], 0 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %[t1], %[w0], %[t1], 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]\n\t"
        "v_pk_fma_f32 %[t0], %[w1], %[t2], %[t0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %[t1], %[w1], %[t3], %[t1] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "s_nop 0\n\t"
        "v_pk_add_f32 %[t0], %[t0], %[t1]"
        : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
        : [a0] "v"(a0), [a1] "v"(a1), [w0] "v"(wa[0]), [w1] "v"(wa[1]), [o0] "i"(OFF), [o1] "i"(OFF + 8)
        : "memory");
    (void)t4; (void)t5;
  }
  ra = t0;
}

// The producer of cols_tile_ws: the walk, and what it does with a completed dst row -- the first row of a pair waits in
// `hold`, the second goes to the strip with it (one slot = one column of both rows).  A function of its own, RAGGED a
// template parameter: everything the walk tests per dst row is a bit of the row's control word or a compile-time choice --
// an instruction of ANY kind costs an issue slot of the SIMD, and the scalar tests of round 4's take() were as many as its
// vector work.  (As a generic lambda inside cols_tile_ws, `hold` stayed in scratch memory and the last strip write became
// a flat_store through a pointer select between scratch and LDS.)
template <typename T, int ES, int TAPS, int P, int D, bool RAGGED>
__device__ __forceinline__ void ws_produce(const ColProg& r, const uint8_t* sp, int spitch, int sh, u32 lane_off,
                                           const int (&wpos)[ES == 3 ? kColEl : 2], v2f32* strip) {
  constexpr int ND = 2 * (int)sizeof(T);
  constexpr int SEG = 2 * kColHalf<ES>;
  v2f32 hold[4];                                              // the pair's first row
#pragma unroll
  for (int i = 0; i < 4; ++i)
    hold[i] = (v2f32){0.0f, 0.0f};
  u32 woff = 0u;                                              // slots: 0 / kColStrip
  auto take = [&](u32 flags, v2f32 (&c)[4]) {
    // (a wave's last row may be the FIRST of a pair: it goes out as the second row of a pair whose first is whatever
    // `hold` still holds -- finite values; the consumer stores the second row only)
    if (!(flags & kProgPair)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("v_mov_b64 %0, %1" : "=v"(hold[i]) : "v"(c[i])); // (assembly: see the end of this function)
      return;
    }
    v2f32 lo[4], hi[4]; // (row a, row b) of this lane's columns 2 i / 2 i + 1 (ES = 2: U / V of pixel i)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lo[i] = pk_mov<0, 0>(hold[i], c[i]);
      hi[i] = pk_mov<1, 1>(hold[i], c[i]);
    }
    ws_barrier();
    v2f32* const st = strip + woff;
    woff ^= (u32)kColStrip;
    if constexpr (ES == 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        st[wpos[2 * i]] = lo[i];
        st[wpos[2 * i + 1]] = hi[i];
      }
    } else if constexpr (RAGGED) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (ES == 2) {
          st[wpos[i & 1] + (i >> 1)] = lo[i];
          st[SEG + wpos[i & 1] + (i >> 1)] = hi[i];
        } else {
          st[wpos[0] + i] = lo[i];
          st[wpos[1] + i] = hi[i];
        }
      }
    } else if constexpr (ES == 2) {
      *reinterpret_cast<float4*>(st + wpos[0]) = make_float4(lo[0].x, lo[0].y, lo[2].x, lo[2].y);
      *reinterpret_cast<float4*>(st + wpos[1]) = make_float4(lo[1].x, lo[1].y, lo[3].x, lo[3].y);
      *reinterpret_cast<float4*>(st + SEG + wpos[0]) = make_float4(hi[0].x, hi[0].y, hi[2].x, hi[2].y);
      *reinterpret_cast<float4*>(st + SEG + wpos[1]) = make_float4(hi[1].x, hi[1].y, hi[3].x, hi[3].y);
    } else {
      *reinterpret_cast<float4*>(st + wpos[0]) = make_float4(lo[0].x, lo[0].y, lo[1].x, lo[1].y);
      *reinterpret_cast<float4*>(st + wpos[0] + 2) = make_float4(lo[2].x, lo[2].y, lo[3].x, lo[3].y);
      *reinterpret_cast<float4*>(st + wpos[1]) = make_float4(hi[0].x, hi[0].y, hi[1].x, hi[1].y);
      *reinterpret_cast<float4*>(st + wpos[1] + 2) = make_float4(hi[2].x, hi[2].y, hi[3].x, hi[3].y);
    }
  };
  cols_walk<T, TAPS, P, ND, D, 4, true>(r, sp, spitch, sh, lane_off, [](const u32 (&d)[ND], v2f32 (&f)[4]) { conv8<T>(d, f); }, take);
}

template <typename T, int ES, int TAPS, int P, int NS>
__device__ __forceinline__ void cols_tile_ws(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                             int dw, int dh, u32 tx, u32 ty, int N, int rps, float* lds, const float4* xtab,
                                             const float4* ytab) {
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int EB = (int)sizeof(T);
  constexpr int ND = 2 * EB;
  constexpr int D = EB == 4 ? 3 : EB == 2 ? 3 : 4;              // source rows in flight
  constexpr int HALF = kColHalf<ES>, SEG = 2 * HALF;            // slots
  const int lane = threadIdx.x & 63;
  const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // 0: producer, 1: consumer
  v2f32* const strip = reinterpret_cast<v2f32*>(lds);
  const u32 lds_base = (u32)(uintptr_t)(__attribute__((address_space(3))) float*)lds;
  const int dwe = dw * ES, row_el = sw * ES;
  const int e0 = (int)tx * N, e_last = min(e0 + N, dwe) - 1;
  const float scale_x = (float)sw / (float)dw;
  const int rows = P * rps;
  const int y_first = (int)ty * rows;
  if (y_first >= dh) // (both waves)
    return;
  const int last_rr = min(rows, dh - y_first) - 1;
  // ---- the tile's source span along x (wave-uniform) ----
  const int px_first = e0 / ES, px_last = e_last / ES;
  const int ux0 = (int)__builtin_floorf((float)px_first * scale_x) - kBefore - 1;
  const int ux1 = (int)__builtin_floorf((float)px_last * scale_x) + TAPS + 1 - kBefore;
  const int sx0 = clampi(ux0, sw - 1), sx1 = clampi(ux1, sw - 1);
  const int j_begin = (sx0 * ES) & ~(kColEl - 1);               // first element the wave loads
  const int px_begin = j_begin / ES;

  if (role == 0) {
    ColProg r;
    ws_rows<TAPS, P, EB == 4, D, kWsProgRows<P, EB == 4>>(sh, dh, y_first, last_rr, lds + kWsProg, r, ytab);
    const int nl = min((((sx1 + 1) * ES - j_begin) + kColEl - 1) / kColEl, kWave); // lanes with data
    const bool ragged = j_begin + kColEl * nl > row_el;           // wave-uniform: only a row's last tile
    const int j0 = min(j_begin + kColEl * min(lane, nl - 1), row_el - kColEl);
    int wpos[ES == 3 ? kColEl : 2];
    if constexpr (ES == 3) {
#pragma unroll
      for (int q = 0; q < kColEl; ++q) {
        const int j = j0 + q, px = j / 3;
        wpos[q] = (j - px * 3) * SEG + col_slot<ES>(kColPadL + px - px_begin);
      }
    } else {
      const int q0 = kColPadL + j0 / ES - px_begin;
      wpos[0] = col_slot<ES>(q0);
      wpos[1] = col_slot<ES>(q0 + 1);
    }
    if (ES != 3 && ragged)
      ws_produce<T, ES, TAPS, P, D, true>(r, sp, spitch, sh, (u32)(j0 * EB), wpos, strip);
    else
      ws_produce<T, ES, TAPS, P, D, false>(r, sp, spitch, sh, (u32)(j0 * EB), wpos, strip);
    ws_barrier(); // the last pair has landed
    return;
  }

  // ---- consumer: the pass along the rows ----
  v2f32 wq[NS][TAPS / 2]; // (w0, w1), (w2, w3), ..
  u32 ha[NS][2];          // LDS byte addresses in strip 0: taps 0, 2, 4 at ha[p][0] + 0, 8, 16; taps 1, 3, 5 at ha[p][1] + 0, 8, 16
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    const int e = min(e0 + p * kWave + lane, e_last);
    const int px = e / ES, ch = e - px * ES;
    const LzTap<TAPS> c = lz_tap_of<TAPS>(xtab, px, scale_x);
#pragma unroll
    for (int k = 0; k < TAPS / 2; ++k)
      wq[p][k] = (v2f32){c.w[2 * k], c.w[2 * k + 1]};
    const int q = kColPadL + (min(c.i, sw - 1) - kBefore - px_begin); // >= kColPadL - kBefore
    ha[p][0] = lds_base + 8u * (u32)(ch * SEG + col_slot<ES>(q));
    ha[p][1] = lds_base + 8u * (u32)(ch * SEG + col_slot<ES>(q + 1));
  }
  const u32 obuf_rd = lds_base + 32u * (u32)lane;                          // (+ the strip's offset)
  const bool pad_left = ux0 < 0, pad_right = ux1 > sw - 1;                 // wave-uniform
  const int edge = kColPadL + (sw - 1) - px_begin;                         // the last pixel of the row
  const int eb = e0 + 4 * lane;                                            // store: 4 adjacent elements
  const int n_out = min(4, e_last + 1 - eb);
  const int n_out2 = NS > 4 ? min(4, e_last + 1 - (eb + 256)) : 0;         // ... and of the fifth set, lanes 0 .. 15
  // whole groups of 4 elements from every lane that stores, on rows aligned to a group: one store per lane and row, no
  // per-lane alignment test, and the row's offset a SCALAR operand of a buffer store (as in cols_walk)
  const bool plain_store = ((e_last + 1 - e0) & 3) == 0 && ((((uintptr_t)dp) | (uintptr_t)dpitch) & (4u * EB - 1u)) == 0; // wave-uniform
  // Round 6: a tile's row segment is cols_n elements (248 ...), so its first and last 64-byte sectors are shared with the neighbouring
  // tiles; a non-temporal store of half a sector does not wait in the L2 for the other half (writes 1.15 - 1.26 x the destination,
  // VERDICT r05 #5b).  A lane whose group lies in a sector the segment covers whole stores non-temporally, the others plain.
  const bool rows64 = ((((uintptr_t)dp) | (uintptr_t)dpitch) & 63u) == 0;                                                 // wave-uniform
  auto whole_sector = [&](int e) { // e: the lane's first element; its group of 4 never straddles a sector (4 EB divides 64)
    const int b = e * EB, lo = b & ~63;
    return rows64 && lo >= e0 * EB && lo + 64 <= (e_last + 1) * EB;
  };
  const bool nt1 = whole_sector(eb), nt2 = NS > 4 ? whole_sector(eb + 256) : false;
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dp, (short)0, (int)0xffffffffu, 0x00020000);
  const int nfull = (last_rr + 1) >> 1;                                    // pairs of two rows
  const bool single_last = (last_rr & 1) == 0;                             // ... and one more of ONE row (its second)
  auto pairs = [&](auto plain_tag, auto pad_tag) {
    constexpr bool PLAIN = decltype(plain_tag)::value, PADS = decltype(pad_tag)::value;
    // soff: byte offset of the row in the plane (scalar)
    auto store_row = [&](int soff, int eb, int n_out, bool nt, float v0, float v1, float v2, float v3) {
      if constexpr (PLAIN) {
        if constexpr (EB == 1) {
          u32 q = __builtin_amdgcn_cvt_pk_u8_f32(v0, 0u, 0u);
          q = __builtin_amdgcn_cvt_pk_u8_f32(v1, 1u, q);
          q = __builtin_amdgcn_cvt_pk_u8_f32(v2, 2u, q);
          q = __builtin_amdgcn_cvt_pk_u8_f32(v3, 3u, q);
          if (nt) __builtin_amdgcn_raw_buffer_store_b32(q, drsrc, eb, soff, 2); // (2: nt -- finished output, whole sectors)
          else __builtin_amdgcn_raw_buffer_store_b32(q, drsrc, eb, soff, 0);
        } else if constexpr (EB == 2) {
          const v2u32 q = {finish_bits<T>(v0) | (finish_bits<T>(v1) << 16), finish_bits<T>(v2) | (finish_bits<T>(v3) << 16)};
          if (nt) __builtin_amdgcn_raw_buffer_store_b64(q, drsrc, eb * 2, soff, 2);
          else __builtin_amdgcn_raw_buffer_store_b64(q, drsrc, eb * 2, soff, 0);
        } else {
          const v4u32 q = {__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)};
          if (nt) __builtin_amdgcn_raw_buffer_store_b128(q, drsrc, eb * 4, soff, 2);
          else __builtin_amdgcn_raw_buffer_store_b128(q, drsrc, eb * 4, soff, 0);
        }
      } else {
        uint8_t* const out = dp + (u32)soff + (size_t)eb * EB;
        const float res[4][1] = {{v0}, {v1}, {v2}, {v3}};
        store_px4<T, 1>(out, res, (1u << n_out) - 1u);
      }
    };
    // one pair from strip B; ONLYB: the pair is one row -- its second (the first is not a row of the image)
    auto pass = [&](auto buf_tag, auto onlyb_tag, int soff) {
      constexpr int B = decltype(buf_tag)::value;
      constexpr bool ONLYB = decltype(onlyb_tag)::value;
      ws_barrier();
      if constexpr (PADS) { // image edges: replicas of the first / last pixel of every channel segment
        v2f32* const st = strip + B * kColStrip;
        const int ch = lane >> 3, i = lane & 7;
        if (pad_left && ch < ES && i < kColPadL)
          st[ch * SEG + col_slot<ES>(kColPadL - 1 - i)] = st[ch * SEG + col_slot<ES>(kColPadL)];
        if (pad_right && ch < ES && i < kColPadR)
          st[ch * SEG + col_slot<ES>(edge + 1 + i)] = st[ch * SEG + col_slot<ES>(edge)];
        wave_lds_sync();
      }
      v2f32 res[NS];
#pragma unroll
      for (int half = 0; half + 1 < NS; half += 2)
        ws_row_taps<TAPS, B * kWsStripBytes>(res[half], res[half + 1], ha[half][0], ha[half][1], ha[half + 1][0], ha[half + 1][1], wq[half], wq[half + 1]);
      if constexpr ((NS & 1) != 0)
        ws_row_taps1<TAPS, B * kWsStripBytes>(res[NS - 1], ha[NS - 1][0], ha[NS - 1][1], wq[NS - 1]);
      // (every tap read of the pair has returned -- each block above ends with lgkmcnt(0): the head of the strip is free)
      v2f32* const obuf = strip + B * kColStrip;
#pragma unroll
      for (int p = 0; p < NS; ++p)
        obuf[p * kWave + lane] = res[p];
      wave_lds_sync();
      if (n_out > 0) {
        float4 v0, v1; // (a0, b0, a1, b1), (a2, b2, a3, b3)
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1) : "v"(obuf_rd), "i"(B * kWsStripBytes), "i"(B * kWsStripBytes + 16) : "memory");
        if constexpr (!ONLYB) {
          store_row(soff, eb, n_out, nt1, v0.x, v0.z, v1.x, v1.z);
          store_row(soff + dpitch, eb, n_out, nt1, v0.y, v0.w, v1.y, v1.w);
        } else {
          store_row(soff, eb, n_out, nt1, v0.y, v0.w, v1.y, v1.w);
        }
      }
      if constexpr (NS > 4) {
        if (n_out2 > 0) { // elements 256 .. of the tile: lanes 0 .. 15
          float4 v0, v1;
          asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(v0), "=&v"(v1) : "v"(obuf_rd), "i"(B * kWsStripBytes + 2048), "i"(B * kWsStripBytes + 2064) : "memory");
          if constexpr (!ONLYB) {
            store_row(soff, eb + 256, n_out2, nt2, v0.x, v0.z, v1.x, v1.z);
            store_row(soff + dpitch, eb + 256, n_out2, nt2, v0.y, v0.w, v1.y, v1.w);
          } else {
            store_row(soff, eb + 256, n_out2, nt2, v0.y, v0.w, v1.y, v1.w);
          }
        }
      }
      wave_lds_sync();
    };
    ws_barrier(); // (the one in front of the producer's first write)
    int soff = y_first * dpitch;
#pragma unroll 1
    for (int n = 0; n < nfull; n += 2) {
      pass(std::integral_constant<int, 0>{}, std::false_type{}, soff);
      soff += 2 * dpitch;
      if (n + 1 < nfull) {
        pass(std::integral_constant<int, 1>{}, std::false_type{}, soff);
        soff += 2 * dpitch;
      }
    }
    if (single_last) {
      if (nfull & 1) pass(std::integral_constant<int, 1>{}, std::true_type{}, soff);
      else pass(std::integral_constant<int, 0>{}, std::true_type{}, soff);
    }
  };
  // (four copies of the loop: which stores and whether there are edges to pad is decided once per tile, not once per pair)
  if (pad_left || pad_right) {
    if (plain_store) pairs(std::true_type{}, std::true_type{});
    else pairs(std::false_type{}, std::true_type{});
  } else {
    if (plain_store) pairs(std::true_type{}, std::false_type{});
    else pairs(std::false_type{}, std::false_type{});
  }
}

// Exactly 2:1 along x (src_w == 2 dst_w: x * scale_x is an integer, every column weight is 0 or 1 and the pass along the
// row is the point sample c[2 x] -- bit for bit: fma(0, c, e) == e, and 1 * c + 0 == c).  The columns that are never
// sampled are never filtered: a lane loads 16 source elements per row and filters the 8 that survive, which are the 8
// dst elements it stores -- no strip, no gather, no transposition; half the conversions and FMAs per source byte.  Planes
// of 1- and 2-element pixels of 8 / 16-bit types (NV12, P10, YUV4xx, Y, RGB_PLANAR); floats keep the general form (a zero
// weight on a non-finite neighbour is not a no-op).
template <typename T, int ES, int TAPS, int P>
__device__ __forceinline__ void cols_tile_x2(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                             int dw, int dh, u32 tx, u32 ty, int rps, float* strip) {
  constexpr int EB = (int)sizeof(T);
  static_assert(EB <= 2 && ES <= 2, "cols_tile_x2: 8 / 16-bit planes of 1 or 2 channels");
  constexpr int ND = 4 * EB;                                    // dwords of a lane's 16 source elements
  constexpr int D = EB == 1 ? 8 : 4;                            // rows in flight: this form is bound by its memory stream (8 vs 4: -5 %)
  const int lane = threadIdx.x & 63;
  ColProg r;
  if (!cols_rows<TAPS, P, false, D>(sh, dh, ty * kWavesPerBlock + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), rps, strip, r))
    return;
  const int dwe = dw * ES;                                      // >= 8 (host)
  const int e0 = (int)tx * (kWave * 8);
  const bool has = e0 + 8 * lane < dwe;
  const int eo = min(e0 + 8 * lane, dwe - 8);                   // a row's last lane slides left to end with the row
  const bool slid = dwe - e0 < kWave * 8 && ((dwe - e0) & 7) != 0;                                          // wave-uniform
  const bool plain_store = !slid && ((((uintptr_t)dp) | (uintptr_t)dpitch) & (8u * EB - 1u)) == 0;
  auto conv = [](const u32 (&d)[ND], v2f32 (&f)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (EB == 1 && ES == 1)      // bytes 4 i, 4 i + 2
        f[i] = (v2f32){(float)(d[i] & 0xffu), (float)((d[i] >> 16) & 0xffu)};
      else if constexpr (EB == 1)            // the (U, V) pair of every other pixel: bytes 4 i, 4 i + 1
        f[i] = (v2f32){(float)(d[i] & 0xffu), (float)((d[i] >> 8) & 0xffu)};
      else if constexpr (ES == 1)            // elements 4 i, 4 i + 2: the low halves of dwords 2 i, 2 i + 1
        f[i] = (v2f32){(float)(d[2 * i] & 0xffffu), (float)(d[2 * i + 1] & 0xffffu)};
      else                                   // elements 4 i, 4 i + 1: dword 2 i
        f[i] = (v2f32){(float)(d[2 * i] & 0xffffu), (float)(d[2 * i] >> 16)};
    }
  };
  auto emit = [&](bool, int rr, v2f32 (&c)[4]) {
    if (!has)
      return;
    uint8_t* const out = dp + (u32)((r.y_first + rr) * dpitch) + (size_t)eo * EB;
    const float res[4][2] = {{c[0].x, c[0].y}, {c[1].x, c[1].y}, {c[2].x, c[2].y}, {c[3].x, c[3].y}};
    if (plain_store && EB == 1) {
      u32 q0 = __builtin_amdgcn_cvt_pk_u8_f32(c[0].x, 0u, 0u), q1 = __builtin_amdgcn_cvt_pk_u8_f32(c[2].x, 0u, 0u);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(c[0].y, 1u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(c[2].y, 1u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(c[1].x, 2u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(c[3].x, 2u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(c[1].y, 3u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(c[3].y, 3u, q1);
      const v2u32 q = {q0, q1};
      gstore_nt<v2u32>(out, q);
    } else {
      store_px4<T, 2>(out, res, 0xfu);
    }
  };
  cols_walk<T, TAPS, P, ND, D>(r, sp, spitch, sh, (u32)(2 * eo * EB), conv, emit);
}

// Exactly 3:2 along x (2 src_w == 3 dst_w: x * scale_x is an integer for even x -- the column weights are {0,0,1,0,0,0}: the
// point sample c[1.5 x] -- and an integer + 1/2 for odd x: ONE set of six weights, the same for every odd pixel of the
// plane).  The pass along the row then needs no per-lane weights and no gather: a lane owns 12 source elements (one 12-byte
// load per row for 8-bit planes) = 8 dst elements; the even ones are copies, the odd ones six FMAs with WAVE-UNIFORM weights
// on its own column results plus the one (two) before and the two (four) after, which come from the neighbouring lanes by
// DPP -- lanes 0 and 63 of a wave only supply those halos (62 x 8 dst elements per wave and row).  Same specification
// arithmetic (e over the even taps, o over the odd ones, e + o), same bits.  1080p -> 720p, 2160p -> 1440p, their chroma
// planes; 8 / 16-bit planes of 1- and 2-element pixels whose dst rows are whole groups of 8 elements.
__device__ __forceinline__ float wave_shr1_f(float v) { // lane l gets lane l - 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_shl1_f(float v) { // lane l gets lane l + 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

constexpr int kX32Out = 62 * 8; // dst elements of a wave's row

// SROWS: the plane is 3:2 down the rows as well (1080p -> 720p, 2160p -> 1440p: the whole surface).  Then the walk needs no
// per-row weights either: dst row 2 m sits on source row 3 m (weights {0,0,1,0,0,0}: the column pass is that row, bit for
// bit), dst row 2 m + 1 at 3 m + 1 1/2: the one weight set w(1/2) on rows 3 m - 1 .. 3 m + 4.  A source row s is tap
// k = (s + 1) mod 3 of one odd row and tap k + 3 of the one before it: TWO accumulator slots, and with the walk unrolled
// six times which slot takes which weight on which trip is static -- 12 packed FMAs per row instead of the general
// walk's 24 (four slots at this ratio, a third of their weights zero) + 5 v_readlane.
template <typename T, int ES, int TAPS, int P, bool SROWS>
__device__ __forceinline__ void cols_tile_x32(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                              int dw, int dh, u32 tx, u32 ty, int rps, float* strip) {
  constexpr int EB = (int)sizeof(T);
  static_assert(EB <= 2 && ES <= 2, "cols_tile_x32: 8 / 16-bit planes of 1 or 2 channels");
  static_assert(!SROWS || TAPS == 6, "cols_tile_x32: the static rows are Lanczos-3's");
  constexpr int ND = 3 * EB;                                    // dwords of a lane's 12 source elements
  constexpr int D = EB == 1 ? 4 : 2;
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  const int lane = threadIdx.x & 63;
  ColProg r;
  int y_first = 0;                                              // dst row of emit()'s row 0
  if constexpr (!SROWS) {
    if (!cols_rows<TAPS, P, false, D>(sh, dh, ty * kWavesPerBlock + (u32)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), rps, strip, r))
      return;
    y_first = r.y_first;
  }
  const int dwe = dw * ES, row_el = sw * ES;                    // dwe % 8 == 0 (host), so row_el % 12 == 0
  const int e0 = (int)tx * kX32Out;                             // first dst element of the wave
  const int eo = e0 + 8 * (lane - 1);                           // this lane's 8 dst elements (lane 0 / 63: halo only)
  const int j0 = (eo >> 1) * 3;                                 // = 1.5 eo: its 12 source elements
  const bool outs = lane >= 1 && lane <= 62 && eo < dwe;
  const bool first = j0 <= 0, beyond = j0 >= row_el;            // lane in front of the row / behind it
  const bool left_edge = j0 == 0, right_edge = j0 + 12 == row_el; // its neighbour is outside the image: replicate
  const u32 lane_off = (u32)(min(max(j0, 0), row_el - 12) * EB);
  // the one weight set of the odd pixels: a = 1/2 exactly (1.5 x is exact in FP32)
  LzTap<TAPS> odd;
  if constexpr (TAPS == 6)
    lanczos3_weights(0.5f, odd.w);
  else
    cubic_weights(0.5f, odd.w);
  float wx[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k)
    wx[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, odd.w[k])));
  const bool plain_store = ((((uintptr_t)dp) | (uintptr_t)dpitch) & (8u * EB - 1u)) == 0;        // wave-uniform
  auto conv = [](const u32 (&d)[ND], v2f32 (&f)[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      if constexpr (EB == 1)
        f[i] = (v2f32){(float)((d[i / 2] >> (16 * (i % 2))) & 0xffu), (float)((d[i / 2] >> (16 * (i % 2) + 8)) & 0xffu)};
      else
        f[i] = (v2f32){(float)(d[i] & 0xffffu), (float)(d[i] >> 16)};
    }
  };
  auto emit = [&](bool, int rr, v2f32 (&cc)[6]) {
    // c[0..11]: this lane's column results; halo: HB elements before, HA after (one pixel before, two after)
    constexpr int HB = ES, HA = 2 * ES;
    float c[HB + 12 + HA];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      c[HB + 2 * i] = cc[i].x;
      c[HB + 2 * i + 1] = cc[i].y;
    }
#pragma unroll
    for (int h = 0; h < HB; ++h) {                      // the previous lane's last pixel, or this lane's first one at the edge
      const float pv = wave_shr1_f(c[HB + 12 - HB + h]);
      c[h] = left_edge ? c[HB + h] : pv;
    }
#pragma unroll
    for (int h = 0; h < HA; ++h) {                      // the next lane's first two pixels, or this lane's last one
      const float nx = wave_shl1_f(c[HB + h]);
      c[HB + 12 + h] = right_edge ? c[HB + 12 - ES + (h % ES)] : nx;
    }
    if (!outs)
      return;
    // dst pixel 2 t = source pixel 3 t (a copy); dst pixel 2 t + 1: taps on source pixels 3 t + 1 - kBefore .. , e over the
    // even taps, o over the odd ones -- two results per packed instruction (ES = 1: two odd pixels; ES = 2: the two channels)
    float out[8];
    constexpr int NPX = 8 / ES;                         // dst pixels of the lane
#pragma unroll
    for (int px = 0; px < NPX; px += 2)
#pragma unroll
      for (int ch = 0; ch < ES; ++ch)
        out[px * ES + ch] = c[HB + (3 * (px >> 1)) * ES + ch];
    if constexpr (ES == 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ta = 2 * q, tb = 2 * q + 1;
        v2f32 e = (v2f32){0.0f, 0.0f}, o = (v2f32){0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < TAPS; k += 2) {
          e = __builtin_elementwise_fma((v2f32){wx[k], wx[k]}, (v2f32){c[HB + 3 * ta + 1 - kBefore + k], c[HB + 3 * tb + 1 - kBefore + k]}, e);
          o = __builtin_elementwise_fma((v2f32){wx[k + 1], wx[k + 1]}, (v2f32){c[HB + 3 * ta + 2 - kBefore + k], c[HB + 3 * tb + 2 - kBefore + k]}, o);
        }
        const v2f32 v = e + o;
        out[2 * ta + 1] = v.x;
        out[2 * tb + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        v2f32 e = (v2f32){0.0f, 0.0f}, o = (v2f32){0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < TAPS; k += 2) {
          const int ia = HB + (3 * t + 1 - kBefore + k) * 2, ib = HB + (3 * t + 2 - kBefore + k) * 2;
          e = __builtin_elementwise_fma((v2f32){wx[k], wx[k]}, (v2f32){c[ia], c[ia + 1]}, e);
          o = __builtin_elementwise_fma((v2f32){wx[k + 1], wx[k + 1]}, (v2f32){c[ib], c[ib + 1]}, o);
        }
        const v2f32 v = e + o;
        out[(2 * t + 1) * 2] = v.x;
        out[(2 * t + 1) * 2 + 1] = v.y;
      }
    }
    uint8_t* const o8 = dp + (u32)((y_first + rr) * dpitch) + (size_t)eo * EB;
    if (plain_store && EB == 1) {
      u32 q0 = __builtin_amdgcn_cvt_pk_u8_f32(out[0], 0u, 0u), q1 = __builtin_amdgcn_cvt_pk_u8_f32(out[4], 0u, 0u);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(out[1], 1u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(out[5], 1u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(out[2], 2u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(out[6], 2u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(out[3], 3u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(out[7], 3u, q1);
      const v2u32 q = {q0, q1};
      gstore_nt<v2u32>(o8, q);
    } else {
      const float res[4][2] = {{out[0], out[1]}, {out[2], out[3]}, {out[4], out[5]}, {out[6], out[7]}};
      store_px4<T, 2>(o8, res, 0xfu);
    }
  };
  (void)first; (void)beyond;
  if constexpr (!SROWS) {
    cols_walk<T, TAPS, P, ND, D, 6>(r, sp, spitch, sh, lane_off, conv, emit);
  } else {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m_first = (int)(ty * kWavesPerBlock + wave) * rps;  // rps: dst row PAIRS per wave here
    const int pairs = dh >> 1;                                    // dh is even (host)
    if (m_first >= pairs)
      return;
    const int m_last = min(m_first + rps, pairs) - 1;
    const int s_begin = 3 * m_first - 1;                          // the walk visits rows s_begin .. 3 m_last + 4, clamped
    const int steps = 3 * (m_last - m_first + 1) + 3;
    constexpr int DS = EB == 1 ? 6 : 3;                           // rows in flight (a divisor of the unroll)
    auto issue = [&](int q, u32 (&d)[ND]) {
      const int sr = min(max(s_begin + min(q, steps - 1), 0), sh - 1); // (rows past the wave's last repeat it: the rows of the wave below would be HBM reads)
      const uint8_t* p = sp + (u32)(sr * spitch) + lane_off;
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w0 = gload_u<v3u32>(p);
      d[0] = w0.x; d[1] = w0.y; d[2] = w0.z;
      if constexpr (ND == 6) {
        const v3u32 w1 = gload_u<v3u32>(p + 12);
        d[3] = w1.x; d[4] = w1.y; d[5] = w1.z;
      }
    };
    v2f32 acc[2][6];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 6; ++i)
        acc[j][i] = (v2f32){0.0f, 0.0f};
    u32 pf[DS][ND];
#pragma unroll
    for (int d = 0; d < DS; ++d) {
      issue(d, pf[d]);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int q0 = 0; q0 < steps; q0 += 6) {
#pragma unroll
      for (int d = 0; d < 6; ++d) {
        const int q = q0 + d;
        v2f32 f[6];
        conv(pf[d % DS], f);
#pragma unroll
        for (int i = 0; i < 6; ++i)
          asm volatile("" : "+v"(f[i])); // (pins the conversions in front of the load that takes their registers)
        __builtin_amdgcn_sched_barrier(0);
        issue(q + DS, pf[d % DS]);
        __builtin_amdgcn_sched_barrier(0);
        if (q >= steps) // the last trip only
          continue;
        const int ph = d % 3, ja = (d / 3) & 1, jb = ja ^ 1;      // static
        const v2f32 wa = (v2f32){wx[ph], wx[ph]}, wb = (v2f32){wx[ph + 3], wx[ph + 3]};
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          acc[ja][i] = __builtin_elementwise_fma(wa, f[i], ph == 0 ? (v2f32){0.0f, 0.0f} : acc[ja][i]); // tap 0 starts the slot at +0
          acc[jb][i] = __builtin_elementwise_fma(wb, f[i], acc[jb][i]);
        }
        const int m = m_first + q / 3;                            // the pair whose even row sits on this source row (ph == 1)
        if (ph == 1 && m <= m_last)
          emit(true, 2 * m, f);
        if (ph == 2 && m - 1 >= m_first && m - 1 <= m_last)       // tap 5 of the odd row of the pair before
          emit(true, 2 * (m - 1) + 1, acc[jb]);
      }
    }
  }
}

// ESSET as in resize_taps.hip: 1 = one-channel planes, 12 = NV12 / P10 (Y + UV), 3 = packed RGB
template <typename T, int ESSET, int TAPS, int P>
__global__ void __launch_bounds__(kBlock) k_resize_cols(const ResizeArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kColWave];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* const strip = lds[wave];
  if constexpr (ESSET == 3) {
    cols_tile<T, 3, TAPS, P>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, strip);
  } else {
    if (ESSET == 12 && job.channels == 2)
      cols_tile<T, 2, TAPS, P>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, strip);
    else
      cols_tile<T, 1, TAPS, P>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, strip);
  }
}

// waves per SIMD the register allocation aims at: the producer's accumulators are 8 P registers
template <int EB, int P> constexpr int kWsWaves = EB == 4 ? (P <= 3 ? 4 : 3) : P <= 3 ? (EB == 2 ? 5 : 6) : P <= 4 ? 5 : 4;
template <typename T, int ESSET, int TAPS, int P, int NS>
__global__ void __launch_bounds__(kWsBlock, (kWsWaves<(int)sizeof(T), NS == 4 ? P : P < 4 ? 4 : P>)) k_resize_cols_ws(const ResizeArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kWsLds<P, sizeof(T) == 4>];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  if constexpr (ESSET == 3) {
    cols_tile_ws<T, 3, TAPS, P, NS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, lds, job.xtab, job.ytab);
  } else {
    if (ESSET == 12 && job.channels == 2)
      cols_tile_ws<T, 2, TAPS, P, NS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, lds, job.xtab, job.ytab);
    else
      cols_tile_ws<T, 1, TAPS, P, NS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_n, a.cols_rps, lds, job.xtab, job.ytab);
  }
}

template <typename T, int ESSET, int TAPS, int P>
__global__ void __launch_bounds__(kBlock) k_resize_cols_x2(const ResizeArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kColProg]; // the waves' programs (cols_rows)
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (ESSET == 12 && job.channels == 2)
    cols_tile_x2<T, 2, TAPS, P>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_rps, lds[wave]);
  else
    cols_tile_x2<T, 1, TAPS, P>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_rps, lds[wave]);
}

template <typename T, int ESSET, int TAPS, int P, bool SROWS = false>
__global__ void __launch_bounds__(kBlock) k_resize_cols_x32(const ResizeArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[kWavesPerBlock][kColProg]; // the waves' programs (cols_rows)
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (ESSET == 12 && job.channels == 2)
    cols_tile_x32<T, 2, TAPS, P, SROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_rps, lds[wave]);
  else
    cols_tile_x32<T, 1, TAPS, P, SROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.cols_rps, lds[wave]);
}

// The same arithmetic one output element per thread, TAPS x TAPS global loads each: planes narrower than one lane's 8
// elements, scale factors whose tile would be narrower than 8 elements, and -- under VALI_TUNE_RESIZE_FORCE_GATHER --
// an independent second implementation the parity suites are replayed through (tests/test_gpu_gather_paths.py).
template <typename T, int TAPS>
__global__ void __launch_bounds__(kBlock) k_resize_cols_direct(const ResizeArgs a) {
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int EB = (int)sizeof(T);
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int es = job.channels;
  const int e = (int)tx * kBlock + (int)threadIdx.x, y = (int)ty;
  if (e >= v.dw * es)
    return;
  const float scale_x = (float)v.sw / (float)v.dw, scale_y = (float)v.sh / (float)v.dh;
  const int px = e / es, ch = e - px * es;
  const LzTap<TAPS> cx = make_lz_tap<TAPS>(px, scale_x), cy = make_lz_tap<TAPS>(y, scale_y);
  float eo[2] = {0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < TAPS; ++k) {
    const size_t col = (size_t)(clampi(cx.i - kBefore + k, v.sw - 1) * es + ch) * EB;
    float c = 0.0f;
#pragma unroll
    for (int r = 0; r < TAPS; ++r) {
      const uint8_t* row = v.sp + (size_t)clampi(cy.i - kBefore + r, v.sh - 1) * v.spitch;
      c = __builtin_fmaf(cy.w[r], (float)gload<T>(row + col), c);
    }
    eo[k & 1] = __builtin_fmaf(cx.w[k], c, eo[k & 1]);
  }
  const float res = eo[0] + eo[1];
  T* out = reinterpret_cast<T*>(v.dp + (size_t)y * v.dpitch) + e;
  if constexpr (EB == 4)
    *(VALI_GLOBAL float*)out = res;
  else
    *(VALI_GLOBAL T*)out = (T)finish_bits<T>(res);
}

template <typename T, int ESSET, int TAPS>
static void launch_slots(const ResizeArgs& a, int slots, int xform, dim3 grid, hipStream_t stream) { // xform: 0 general, 2 / 3: the 2:1 / 3:2-along-x forms, 4: 3:2 both ways
  constexpr int P0 = TAPS == 6 ? 3 : 2, P1 = TAPS == 6 ? 4 : 3, P2 = TAPS == 6 ? 6 : 4;
  if constexpr (sizeof(T) <= 2 && ESSET != 3) {
    if (xform == 2) {
      if (slots <= P0)
        hipLaunchKernelGGL((k_resize_cols_x2<T, ESSET, TAPS, P0>), grid, dim3(kBlock), 0, stream, a);
      else if (slots <= P1)
        hipLaunchKernelGGL((k_resize_cols_x2<T, ESSET, TAPS, P1>), grid, dim3(kBlock), 0, stream, a);
      else
        hipLaunchKernelGGL((k_resize_cols_x2<T, ESSET, TAPS, P2>), grid, dim3(kBlock), 0, stream, a);
      return;
    }
    if (xform == 4) { // 3:2 along x AND y (Lanczos-3): the static walk, no slots
      if constexpr (TAPS == 6)
        hipLaunchKernelGGL((k_resize_cols_x32<T, ESSET, TAPS, P0, true>), grid, dim3(kBlock), 0, stream, a);
      return;
    }
    if (xform == 3) {
      if (slots <= P0)
        hipLaunchKernelGGL((k_resize_cols_x32<T, ESSET, TAPS, P0>), grid, dim3(kBlock), 0, stream, a);
      else if (slots <= P1)
        hipLaunchKernelGGL((k_resize_cols_x32<T, ESSET, TAPS, P1>), grid, dim3(kBlock), 0, stream, a);
      else
        hipLaunchKernelGGL((k_resize_cols_x32<T, ESSET, TAPS, P2>), grid, dim3(kBlock), 0, stream, a);
      return;
    }
  }
  if (xform == 5) { // the general form on specialised waves: a workgroup = producer + consumer of ONE tile
    const int cap = tuning(VALI_TUNE_WAVES_PER_CU); // (measurements: workgroups per CU = cap / 2, by unused dynamic LDS)
    constexpr int kLds = kWsLds<P0, sizeof(T) == 4> * 4;
    const unsigned dyn = cap >= 2 && 160 * 1024 / (cap / 2) > kLds ? (unsigned)(160 * 1024 / (cap / 2) - kLds) & ~15u : 0u;
#define VALI_WS_LAUNCH(P, NS) hipLaunchKernelGGL((k_resize_cols_ws<T, ESSET, TAPS, P, NS>), grid, dim3(kWsBlock), dyn, stream, a)
    if (a.cols_n > 4 * kWave) { // wide tiles: five sets per consumer lane
      if (slots <= P0) VALI_WS_LAUNCH(P0, 5);
      else if (slots <= P1) VALI_WS_LAUNCH(P1, 5);
      else VALI_WS_LAUNCH(P2, 5);
    } else {
      if (slots <= P0) VALI_WS_LAUNCH(P0, 4);
      else if (slots <= P1) VALI_WS_LAUNCH(P1, 4);
      else VALI_WS_LAUNCH(P2, 4);
    }
#undef VALI_WS_LAUNCH
    return;
  }
  if (slots <= P0)
    hipLaunchKernelGGL((k_resize_cols<T, ESSET, TAPS, P0>), grid, dim3(kBlock), 0, stream, a);
  else if (slots <= P1)
    hipLaunchKernelGGL((k_resize_cols<T, ESSET, TAPS, P1>), grid, dim3(kBlock), 0, stream, a);
  else
    hipLaunchKernelGGL((k_resize_cols<T, ESSET, TAPS, P2>), grid, dim3(kBlock), 0, stream, a);
}

// Every job of `base` shrinks (or keeps) its plane height.  Returns VALI_OK after launching.
int launch_resize_cols(const ResizeArgs& base, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream) {
  const bool direct_only = tuning(VALI_TUNE_RESIZE_FORCE_GATHER) == 1;
  const int force = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE); // rows per wave: 1 / 2 / 3 force 2 / 1 / 8 rows per slot
  ResizeArgs a = base;
  int esset = 0, tile_n = 256, tile_w = 320, slots = 1;
  bool narrow = false, x2 = elem <= 2 && tuning(VALI_TUNE_RESIZE_POINT) != 0; // exactly 2:1 along x on every plane
  bool x32 = x2;                                                              // exactly 3:2 along x on every plane
  bool y32 = taps == 6;                                                       // ... and down the rows (dst height even)
  for (int k = 0; k < a.njobs; ++k) {
    const ResizeJob& j = a.job[k];
    const int c = j.channels;
    esset = c == 3 ? 3 : (c == 2 || esset == 12) ? 12 : (esset ? esset : 1);
    const int sw = src_w >> j.ssub_x, dw = dst_w >> j.sub_x, sh = src_h >> j.ssub_y, dh = dst_h >> j.sub_y;
    narrow = narrow || sw * c < kColEl;
    x2 = x2 && c <= 2 && sw == 2 * dw && dw * c >= 8;
    x32 = x32 && c <= 2 && 2 * sw == 3 * dw && (dw * c) % 8 == 0;
    y32 = y32 && 2 * sh == 3 * dh && dh % 2 == 0;
    // dst elements per tile: the source span of its pixels (+ taps, + the two extra elements, + alignment slop) must fit
    // the 512 elements a wave loads per row.  A tile starts on a pixel unless pixels are 3 elements (N is a multiple of 4).
    const double sx = (double)sw / (double)dw * (1.0 + 1e-6);
    int nn = 320; // (the specialised waves take up to 5 x 64 elements per tile, the other forms 4 x 64)
    while (nn >= 8 && (((c == 3 ? nn / 3 + 2 : nn / c) - 1) * sx + taps + 4) * c + kColEl - 1 > (double)kColSpan)
      nn -= 4;
    tile_w = nn < tile_w ? nn : tile_w;
    nn = nn < 256 ? nn : 256;
    tile_n = nn < tile_n ? nn : tile_n;
    // slots: the smallest P with P * scale_y >= taps - 1 (a slot's consecutive rows may share ONE source row: cols_rows;
    // and a margin for the rounding of y * scale_y in FP32)
    // (a ratio FP32 represents exactly -- 3:2, 2:1, 5:4 ... -- makes y * scale_y exact: no margin, 3:2 runs on 4 slots, not 6)
    const double sy = (double)sh / (double)dh;
    const float syf = (float)sh / (float)dh;
    const bool exact = (double)syf * (double)dh == (double)sh && sh < (1 << 20);
    int p = 1;
    while (p < 6 && p * sy < (taps - 1) + (exact ? 0.0 : sh * 2.5e-7 + 1e-6))
      ++p;
    slots = p > slots ? p : slots;
  }
  auto launch_direct = [&]() {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dwe = (dst_w >> a.job[k].sub_x) * a.job[k].channels, dh = dst_h >> a.job[k].sub_y;
      a.job[k].first_tile = total;
      a.job[k].tiles_x = (u32)(dwe + kBlock - 1) / kBlock;
      total += a.job[k].tiles_x * (u32)dh;
    }
    a.map = make_tile_map_linear(total, (u32)n);
    const dim3 grid = tile_grid(a.map);
#define VALI_COLS_DIRECT(T)                                                                          \
  do {                                                                                               \
    if (taps == 6) hipLaunchKernelGGL((k_resize_cols_direct<T, 6>), grid, dim3(kBlock), 0, stream, a); \
    else hipLaunchKernelGGL((k_resize_cols_direct<T, 4>), grid, dim3(kBlock), 0, stream, a);           \
  } while (0)
    if (elem == 1) VALI_COLS_DIRECT(uint8_t);
    else if (elem == 2) VALI_COLS_DIRECT(uint16_t);
    else VALI_COLS_DIRECT(float);
#undef VALI_COLS_DIRECT
    VALI_LAUNCH_CHECK();
    return (int)VALI_OK;
  };
  if (narrow || tile_n < 8 || direct_only)
    return launch_direct();
  const int pmin = taps == 6 ? 3 : 2, pmid = taps == 6 ? 4 : 3, pmax = taps == 6 ? 6 : 4;
  const int P = slots <= pmin ? pmin : slots <= pmid ? pmid : pmax;
  // rows per wave = P x rps <= 64 (a lane evaluates a row's taps) and few enough for the wave's program in LDS:
  // (rows - 1) scale_y + taps + 1 source rows, + the walk's rows in flight (<= 8), <= kProgRows<P>
  const bool ws = !x2 && !x32 && tuning(VALI_TUNE_RESIZE_COLS) != 1;           // the general form: specialised waves
  if (ws && tuning(VALI_TUNE_RESIZE_COLS) != 2) {                             // ... with their taps from tables (2: computed in the kernel)
    const int dev = stream_device(stream);
    for (int k = 0; k < a.njobs; ++k) {
      ResizeJob& j = a.job[k];
      j.xtab = tap_table(dev, stream, src_w >> j.ssub_x, dst_w >> j.sub_x, taps);
      j.ytab = tap_table(dev, stream, src_h >> j.ssub_y, dst_h >> j.sub_y, taps);
    }
  }
  if (ws)
    tile_n = tile_w;
  // (specialised waves: programs of 224 / 160 entries, and two rounds of 64 lanes evaluate the rows' taps)
  const int prog_rows = ws ? ((P <= 3 && elem != 4) ? kWsProgRows<3, false> : kWsProgRows<6, true>) : (P <= 3 && elem != 4) ? kProgRows<3, false> : kProgRows<6, true>;
  int rps_max = (ws ? 128 : 64) / P;
  for (int k = 0; k < a.njobs; ++k) {
    const double sy = (double)(src_h >> a.job[k].ssub_y) / (double)(dst_h >> a.job[k].sub_y) * (1.0 + 1e-6);
    while (rps_max > 0 && (P * rps_max - 1) * sy + taps + 1 + 8 > (double)prog_rows)
      --rps_max;
  }
  if (rps_max < 1)
    return launch_direct();
  // even tiles: the same number of tiles along x, none of them nearly empty
  {
    int widest = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dwe = (dst_w >> a.job[k].sub_x) * a.job[k].channels;
      widest = dwe > widest ? dwe : widest;
    }
    const int tiles = (widest + tile_n - 1) / tile_n;
    const int even = (((widest + tiles - 1) / tiles) + 3) & ~3;
    tile_n = even < tile_n ? even : tile_n;
    if (x2)
      tile_n = kWave * 8; // 8 dst elements per lane, nothing shared between tiles
    else if (x32)
      tile_n = kX32Out;   // 62 lanes x 8 dst elements, lanes 0 and 63 supply the halos
  }
  auto count = [&](int rps, bool assign) {
    u32 total = 0;
    const int rows = (ws ? 1 : kWavesPerBlock) * P * rps;
    for (int k = 0; k < a.njobs; ++k) {
      const int dwe = (dst_w >> a.job[k].sub_x) * a.job[k].channels, dh = dst_h >> a.job[k].sub_y;
      const u32 tiles_x = (u32)(dwe + tile_n - 1) / (u32)tile_n;
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (u32)((dh + rows - 1) / rows);
    }
    return total;
  };
  // rows per slot: as many as the program allows (the tile's TAPS - 1 shared source rows weigh least) while the launch still
  // fills the chip, fewer for small launches -- a lone wave walks its rows one memory round trip after the other
  // (the forms without a row pass are bound by their memory stream, and short waves serve it best: 2:1 along x, 64
  // frames of 2160p -> 1920x1088, rows per slot 16 / 8 / 4 / 2 / 1: 3.2 / 3.05 / 2.95 / 2.78 / 3.2 us -- the rows two
  // neighbouring waves both walk come from the L2 while both run; the general form wants long waves: 4.05 / 4.2 / 4.55)
  int rps = x2 ? (rps_max < 2 ? rps_max : 2) : x32 ? (rps_max < 4 ? rps_max : 4) : rps_max;
  if (force == 1) rps = rps_max < 2 ? rps_max : 2;
  else if (force == 2) rps = 1;
  else if (force == 3) rps = rps_max;
  else if (force >= 11 && force <= 42) rps = force - 10 < rps_max ? force - 10 : rps_max; // (measurements: rows per slot = value - 10)
  else
    // (2048 workgroups = 8 waves per SIMD: with fewer the launch's last round leaves SIMDs idle -- 64 frames of 1080p -> 720p
    // were 6912 waves of 30 rows, 1.4 rounds)
    while (rps > 1 && (unsigned long long)count(rps, false) * (unsigned)n < (ws ? 4096ull : 2048ull))
      rps = rps > 2 ? rps / 2 : 1;
  const bool srows = x32 && !x2 && y32 && force != 2;                         // (RESIZE_NO_SEPARABLE = 2: the slot walk, for A/B and tests)
  if (srows) {
    // dst row PAIRS per wave: 3 of a wave's 3 n + 3 source rows are its neighbours' (they come from the L2 while both run);
    // 4 pairs unless that leaves the chip short of workgroups
    auto count_pairs = [&](int pw, bool assign) {
      u32 total = 0;
      for (int k = 0; k < a.njobs; ++k) {
        const int dwe = (dst_w >> a.job[k].sub_x) * a.job[k].channels, pairs = (dst_h >> a.job[k].sub_y) / 2;
        const u32 tiles_x = (u32)(dwe + tile_n - 1) / (u32)tile_n;
        if (assign) {
          a.job[k].first_tile = total;
          a.job[k].tiles_x = tiles_x;
        }
        total += tiles_x * (u32)((pairs + kWavesPerBlock * pw - 1) / (kWavesPerBlock * pw));
      }
      return total;
    };
    // (1080p -> 720p, 64 frames, pairs per wave 1 / 2 / 3 / 4 / 6 / 8 / 12 / 18 / 24: 1.00 / 0.86 / 0.75 / 0.76 / 0.78 / 0.79 / 0.80 /
    // 0.77 / 0.79 us: short waves, like every form without a row pass)
    int pw = force == 1 ? 1 : (force >= 11 && force <= 42) ? force - 10 : 4; // (11 .. 42: measurements, pairs per wave = value - 10)
    while (pw > 1 && (unsigned long long)count_pairs(pw, false) * (unsigned)n < 2048ull)
      pw = pw > 2 ? pw / 2 : 1;
    a.map = make_tile_map_linear(count_pairs(pw, true), (u32)n);
    a.cols_n = tile_n;
    a.cols_rps = pw;
  } else {
    a.map = make_tile_map_linear(count(rps, true), (u32)n);
    a.cols_n = tile_n;
    a.cols_rps = rps;
  }
  const dim3 grid = tile_grid(a.map);
#define VALI_COLS_T(T)                                                               \
  do {                                                                               \
    if (taps == 6) {                                                                 \
      if (esset == 1) launch_slots<T, 1, 6>(a, P, x2 ? 2 : srows ? 4 : x32 ? 3 : ws ? 5 : 0, grid, stream);                      \
      else if (esset == 12) launch_slots<T, 12, 6>(a, P, x2 ? 2 : srows ? 4 : x32 ? 3 : ws ? 5 : 0, grid, stream);               \
      else launch_slots<T, 3, 6>(a, P, x2 ? 2 : srows ? 4 : x32 ? 3 : ws ? 5 : 0, grid, stream);                                 \
    } else {                                                                         \
      if (esset == 1) launch_slots<T, 1, 4>(a, P, x2 ? 2 : x32 ? 3 : ws ? 5 : 0, grid, stream);                      \
      else if (esset == 12) launch_slots<T, 12, 4>(a, P, x2 ? 2 : x32 ? 3 : ws ? 5 : 0, grid, stream);               \
      else launch_slots<T, 3, 4>(a, P, x2 ? 2 : x32 ? 3 : ws ? 5 : 0, grid, stream);                                 \
    }                                                                                \
  } while (0)
  if (elem == 1) VALI_COLS_T(uint8_t);
  else if (elem == 2) VALI_COLS_T(uint16_t);
  else VALI_COLS_T(float);
#undef VALI_COLS_T
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali
