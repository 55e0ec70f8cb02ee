// Lanczos-3 / bicubic of one- and two-channel planes that are exactly DOUBLED in both directions (dst_w = 2 src_w, dst_h = 2 src_h).
//
// This is what the reference's UDPlanar does to the chroma planes of every 4:2:0 -> 4:4:4 surface of unchanged size
// (src/TC/src/UDSurface.cpp:33-93: nppiResize with NPPI_INTER_LANCZOS per plane; YUV420 -> YUV444 and the 10-bit pair), and
// what PySurfaceResizer does to planar surfaces at 2x.  The general rows-first kernel (resize_taps.hip) ran it at 0.2 of the
// HBM roofline: per-lane tap weights, an LDS gather per source row, an LDS ring for the vertical pass.  At exactly 1:2 none of
// that is needed (specification: oracle/vali_oracle.c resize_plane_taps, the src_h < dst_h branch -- rows first):
//
//   x * 0.5 is exact: an even dst column 2 j sits ON source pixel j -- its weights are {0,0,1,0,0,0} ({-0,1,0,-0} for the
//   cubic) and the specification's chains reduce to that pixel bit for bit (0 * t = +0 for t >= 0, fma(1, t, 0) = t,
//   fma(0, u, t) = t) -- and an odd column 2 j + 1 sits at j + 1/2: ONE weight set W = w(1/2) for every odd column of the
//   plane, on pixels j - 2 .. j + 3.  Rows likewise: dst row 2 r is the row-filtered source row r, dst row 2 r + 1 the chain
//   v = W0 h(r-2) ; v = fma(Wk, h(r-2+k), v).
//
// So: a lane owns 4 adjacent source pixels (one 4- / 8-byte load per row, 6 rows in flight), gets the 2 + 3 pixels around them
// from its neighbours by DPP (lanes 0 and 63 of a wave only supply those), filters its 4 odd columns with WAVE-UNIFORM weights
// in registers (e over the even taps, o over the odd ones, e + o: the specification's order), and has the row-filtered row h:
// 8 dst columns.  Down the columns every source row is scattered into TAPS accumulator slots (dst row 2 r' + 1 lives in slot
// r' mod TAPS); with the walk unrolled TAPS times, which weight a slot takes on which trip is STATIC: no tap fetch, no
// control flow, `acc = W0 * h` starts a slot (nothing to clear) and the slot that took W(TAPS-1) is complete.  Per source
// row a wave stores two dst rows of 496 elements; rows outside the plane are their clamped neighbours (the walk simply visits
// the clamped row again).  8 / 16-bit planes whose width is a multiple of 4 (of 2 pixels for the UV plane of NV12 / P10, where a
// lane owns 2 pixels); everything else keeps the general kernel.
#include "resize_common.hpp"
#include "resize_weights.hpp"

namespace vali {

constexpr int kUp2El = 4;                       // source elements per lane: 4 pixels of one channel, 2 of two
constexpr int up2_span(int channels) { return channels == 1 ? 62 * 4 : 61 * 2; } // source pixels of a wave's row

// (bound_ctrl: the one lane without a source reads 0 -- lanes 0 / 63 only supply halos here -- and, more to the point, the
// instruction needs no `old` value: with old = v the compiler copies v into the destination first, one v_mov per shift)
__device__ __forceinline__ float up2_shr1(float v) { // lane l gets lane l - 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float up2_shl1(float v) { // lane l gets lane l + 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// ES = interleaved channels of the plane: 1 (Y, U, V, the planes of RGB_PLANAR) or 2 (the UV plane of NV12 / P10: a lane then
// owns 2 pixels = 4 elements, the three pixels after them are the next lane's two and the first of the lane after that)
template <typename T, int TAPS, int ES>
__device__ __forceinline__ void up2_tile(const PlaneView& v, u32 tx, u32 ty, int rpw) {
  constexpr int EB = (int)sizeof(T);
  constexpr int kBefore = LzTap<TAPS>::kBefore, kAfter = TAPS - 1 - kBefore;
  constexpr int ND = EB;                                        // dwords of a lane's 4 elements
  constexpr int PXL = kUp2El / ES;                              // pixels per lane
  constexpr int LAST = ES == 1 ? 62 : 61;                       // lanes 1 .. LAST produce output, the others supply halos
  constexpr int NC = (PXL + 5) * ES;                            // elements of pixels j0 - 2 .. j0 + PXL + 2
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int r_first = (int)(ty * kWavesPerBlock + wave) * rpw;  // wave-uniform
  if (r_first >= v.sh)
    return;
  const int r_last = min(r_first + rpw, v.sh) - 1;
  const int sw = v.sw;                                          // sw % PXL == 0, sw >= PXL (host)
  const int j0 = (int)tx * (LAST * PXL) + PXL * (lane - 1);     // this lane's source pixels j0 .. j0 + PXL - 1
  const bool outs = lane >= 1 && lane <= LAST && j0 < sw;
  const bool left_edge = j0 == 0, right_edge = j0 + PXL == sw;  // its neighbour is outside the plane: replicate
  const bool next_last = j0 + 2 * PXL == sw;                    // (ES = 2) the next lane's pixels are the row's last
  const int tile_edge = __builtin_amdgcn_readfirstlane((tx == 0 || ((int)tx + 1) * (LAST * PXL) >= sw) ? 1 : 0); // wave-uniform
  const u32 lane_off = (u32)(min(max(j0, 0), sw - PXL) * ES * EB);

  // the one weight set of the odd columns and rows: a = 1/2 exactly
  LzTap<TAPS> odd;
  if constexpr (TAPS == 6)
    lanczos3_weights(0.5f, odd.w);
  else
    cubic_weights(0.5f, odd.w);
  float W[TAPS];
#pragma unroll
  for (int k = 0; k < TAPS; ++k)
    W[k] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, odd.w[k])));

  // the weight pairs of the packed row pass, aligned and moved one slot (ES = 1); pinned in scalar register pairs: left to
  // itself the compiler re-assembles the moved pairs with s_mov in front of every use
  v2f32 wp[TAPS / 2], ws[TAPS / 2 + 1];
#pragma unroll
  for (int k = 0; k < TAPS / 2; ++k)
    wp[k] = (v2f32){W[2 * k], W[2 * k + 1]};
  ws[0] = (v2f32){0.0f, W[0]};
#pragma unroll
  for (int k = 1; k < TAPS / 2; ++k)
    ws[k] = (v2f32){W[2 * k - 1], W[2 * k]};
  ws[TAPS / 2] = (v2f32){W[TAPS - 1], 0.0f};
  if constexpr (ES == 1) {
#pragma unroll
    for (int k = 0; k < TAPS / 2; ++k)
      asm volatile("" : "+s"(wp[k]));
#pragma unroll
    for (int k = 0; k <= TAPS / 2; ++k)
      asm volatile("" : "+s"(ws[k]));
  }
  const int q_begin = r_first - kBefore;                        // the walk visits rows q_begin .. r_last + kAfter, clamped
  const int steps = r_last + kAfter - q_begin + 1;
  uint8_t* const out0 = v.dp + (size_t)(2 * max(j0, 0)) * ES * EB;
  const bool plain_store = EB == 1 && ((((uintptr_t)v.dp) | (uintptr_t)v.dpitch) & 7u) == 0; // wave-uniform

  auto row_ptr = [&](int q) { // q may run past the walk (the prefetch): clamped to the plane
    const int rv = min(max(q_begin + min(q, steps - 1), 0), v.sh - 1); // (past the walk: its last row again, not the next wave's rows)
    return v.sp + (u32)(rv * v.spitch) + lane_off;
  };
  auto issue = [&](int q, u32 (&d)[ND]) {
    if constexpr (ND == 1) {
      d[0] = gload_u<u32>(row_ptr(q));
    } else {
      const v2u32 w = gload_u<v2u32>(row_ptr(q));
      d[0] = w.x; d[1] = w.y;
    }
  };
  auto store8 = [&](int dst_row, const v2f32 (&h)[4]) {
    if (!outs)
      return;
    uint8_t* const o = out0 + (u32)(dst_row * v.dpitch);
    if (plain_store) {
      u32 q0 = __builtin_amdgcn_cvt_pk_u8_f32(h[0].x, 0u, 0u), q1 = __builtin_amdgcn_cvt_pk_u8_f32(h[2].x, 0u, 0u);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(h[0].y, 1u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(h[2].y, 1u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(h[1].x, 2u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(h[3].x, 2u, q1);
      q0 = __builtin_amdgcn_cvt_pk_u8_f32(h[1].y, 3u, q0); q1 = __builtin_amdgcn_cvt_pk_u8_f32(h[3].y, 3u, q1);
      const v2u32 q = {q0, q1};
      gstore_nt<v2u32>(o, q);
    } else {
      const float res[4][2] = {{h[0].x, h[0].y}, {h[1].x, h[1].y}, {h[2].x, h[2].y}, {h[3].x, h[3].y}};
      store_px4<T, 2>(o, res, 0xfu);
    }
  };

  v2f32 acc[TAPS][4];
#pragma unroll
  for (int s = 0; s < TAPS; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[s][i] = (v2f32){0.0f, 0.0f};
  u32 pf[TAPS][ND];
#pragma unroll
  for (int d = 0; d < TAPS; ++d) {
    issue(d, pf[d]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int q0 = 0; q0 < steps; q0 += TAPS) {
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      const int q = q0 + d;
      // ---- the source row: the elements of pixels j0 - 2 .. j0 + PXL + 2 as floats; the lane's own start at c[2 ES] ----
      float c[NC];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (EB == 1)
          c[2 * ES + i] = (float)((pf[d][0] >> (8 * i)) & 0xffu);
        else
          c[2 * ES + i] = (float)((pf[d][i / 2] >> (16 * (i % 2))) & 0xffffu);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
        asm volatile("" : "+v"(c[2 * ES + i])); // (pins the conversions in front of the load that takes their registers)
      __builtin_amdgcn_sched_barrier(0);
      issue(q + TAPS, pf[d]);
      __builtin_amdgcn_sched_barrier(0);
      if (q >= steps) // the last trip only
        continue;
      v2f32 h[4];
      if constexpr (ES == 1) {
        // the two pixels before (the previous lane's last two) and the three after (the next lane's first three); at the
        // plane's own edges the edge pixel -- only tiles that touch an edge pay the selects
        float L0 = up2_shr1(c[4]), L1 = up2_shr1(c[5]);
        float R0 = up2_shl1(c[2]), R1 = up2_shl1(c[3]), R2 = up2_shl1(c[4]);
        if (tile_edge) {
          L0 = left_edge ? c[2] : L0; L1 = left_edge ? c[2] : L1;
          R0 = right_edge ? c[5] : R0; R1 = right_edge ? c[5] : R1; R2 = right_edge ? c[5] : R2;
        }
        // ---- along the row: even dst pixels are the source pixels; odd pixel i takes the taps on pixels i + 2 - kBefore ..
        // of (L0 L1 c2 c3 c4 c5 R0 R1 R2) = aligned register pairs p0 .. p4; a window that starts at an odd position runs
        // over the same pairs with its weights moved one slot, (0,W0) (W1,W2) .. (W_last,0): the low halves then hold the odd
        // taps' chain, the high halves the even taps' -- the specification's two chains, for elements >= 0 bit for bit
        const v2f32 p0 = {L0, L1}, p1 = {c[2], c[3]}, p2 = {c[4], c[5]}, p3 = {R0, R1}, p4 = {R2, 0.0f};
        float o0, o1, o2, o3;
        if constexpr (TAPS == 6) {
          const v2f32 w01 = wp[0], w23 = wp[1], w45 = wp[2];
          const v2f32 s0 = ws[0], s12 = ws[1], s34 = ws[2], s5 = ws[3];
          v2f32 a0 = w01 * p0, a2 = w01 * p1, a1 = s0 * p0, a3 = s0 * p1;
          a0 = __builtin_elementwise_fma(w23, p1, a0); a2 = __builtin_elementwise_fma(w23, p2, a2);
          a1 = __builtin_elementwise_fma(s12, p1, a1); a3 = __builtin_elementwise_fma(s12, p2, a3);
          a0 = __builtin_elementwise_fma(w45, p2, a0); a2 = __builtin_elementwise_fma(w45, p3, a2);
          a1 = __builtin_elementwise_fma(s34, p2, a1); a3 = __builtin_elementwise_fma(s34, p3, a3);
          a1 = __builtin_elementwise_fma(s5, p3, a1);  a3 = __builtin_elementwise_fma(s5, p4, a3);
          o0 = a0.x + a0.y; o2 = a2.x + a2.y;
          o1 = a1.y + a1.x; o3 = a3.y + a3.x;
        } else {
          const v2f32 w01 = wp[0], w23 = wp[1];
          const v2f32 s0 = ws[0], s12 = ws[1], s3 = ws[2];
          v2f32 a0 = s0 * p0, a2 = s0 * p1, a1 = w01 * p1, a3 = w01 * p2;
          a0 = __builtin_elementwise_fma(s12, p1, a0); a2 = __builtin_elementwise_fma(s12, p2, a2);
          a1 = __builtin_elementwise_fma(w23, p2, a1); a3 = __builtin_elementwise_fma(w23, p3, a3);
          a0 = __builtin_elementwise_fma(s3, p2, a0);  a2 = __builtin_elementwise_fma(s3, p3, a2);
          o0 = a0.y + a0.x; o2 = a2.y + a2.x;
          o1 = a1.x + a1.y; o3 = a3.x + a3.y;
          (void)p4;
        }
        asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3)); // (keeps the four sums scalar)
        h[0] = (v2f32){c[2], o0}; h[1] = (v2f32){c[3], o1}; h[2] = (v2f32){c[4], o2}; h[3] = (v2f32){c[5], o3};
      } else {
#pragma unroll
        for (int t = 0; t < 2 * ES; ++t) {                      // the two pixels before: the previous lane's last two
          const float p = up2_shr1(c[PXL * ES + t]);
          c[t] = left_edge ? c[2 * ES + t % ES] : p;
        }
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const float n0 = up2_shl1(c[4 + ch]), n1 = up2_shl1(c[6 + ch]); // the next lane's two pixels ...
          const float n2 = up2_shl1(n0);                                   // ... and the first of the lane after it
          const float own_last = c[6 + ch];
          c[8 + ch] = right_edge ? own_last : n0;
          c[10 + ch] = right_edge ? own_last : n1;
          c[12 + ch] = right_edge ? own_last : next_last ? n1 : n2;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int b = (i + 2 - kBefore) * 2;
          v2f32 he = (v2f32){W[0], W[0]} * (v2f32){c[b], c[b + 1]}, ho = (v2f32){W[1], W[1]} * (v2f32){c[b + 2], c[b + 3]};
#pragma unroll
          for (int k = 2; k < TAPS; k += 2) {
            he = __builtin_elementwise_fma((v2f32){W[k], W[k]}, (v2f32){c[b + 2 * k], c[b + 2 * k + 1]}, he);
            ho = __builtin_elementwise_fma((v2f32){W[k + 1], W[k + 1]}, (v2f32){c[b + 2 * k + 2], c[b + 2 * k + 3]}, ho);
          }
          h[2 * i] = (v2f32){c[(2 + i) * 2], c[(2 + i) * 2 + 1]};
          h[2 * i + 1] = he + ho;
        }
      }
      // ---- down the columns: slot s takes tap (d - s) mod TAPS of this row ----
#pragma unroll
      for (int s = 0; s < TAPS; ++s) {
        const int k = (d - s + TAPS) % TAPS;
        const v2f32 wv = (v2f32){W[k], W[k]};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[s][i] = k == 0 ? wv * h[i] : __builtin_elementwise_fma(wv, h[i], acc[s][i]);
      }
      const int rv = q_begin + q;                               // this (virtual) source row
      if (rv >= r_first && rv <= r_last)                        // dst row 2 rv: the row-filtered row itself
        store8(2 * rv, h);
      const int ro = rv - kAfter;                               // the odd row whose last tap this was
      if (ro >= r_first && ro <= r_last)
        store8(2 * ro + 1, acc[(d + 1) % TAPS]);
    }
  }
}

// UV: the launch has two-channel planes (NV12 / P10); planar surfaces take the one-channel kernel (one register fewer: 7 waves)
template <typename T, int TAPS, bool UV>
__global__ void __launch_bounds__(kBlock) k_resize_up2(const ResizeArgs a) {
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  if (job.kind == 1) { // a plane of unchanged size riding along (UDPlanar's luma)
    plane_copy_tile(v, v.dw * job.channels * (int)sizeof(T), tx, ty);
    return;
  }
  if (UV && job.channels == 2)
    up2_tile<T, TAPS, 2>(v, tx, ty, a.cols_rps);
  else
    up2_tile<T, TAPS, 1>(v, tx, ty, a.cols_rps);
}

// Every job of `base`: one or two channels, dst = 2 x src in both directions, source width a multiple of 4 / channels.
int launch_resize_up2(const ResizeArgs& base, int elem, int taps, int src_w, int src_h, int n, hipStream_t stream) {
  ResizeArgs a = base;
  auto count = [&](int rpw, bool assign) {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int sw = src_w >> a.job[k].ssub_x, sh = src_h >> a.job[k].ssub_y;
      const bool copy = a.job[k].kind == 1; // (unchanged size: the source geometry is the destination's)
      const u32 tiles_x = copy ? (u32)(sw * a.job[k].channels * elem + kCopyW - 1) / kCopyW
                               : (u32)(sw + up2_span(a.job[k].channels) - 1) / (u32)up2_span(a.job[k].channels);
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (copy ? (u32)((sh + kCopyH - 1) / kCopyH) : (u32)((sh + kWavesPerBlock * rpw - 1) / (kWavesPerBlock * rpw)));
    }
    return total;
  };
  // Source rows per wave.  TAPS - 1 rows of every wave are its neighbours' (walked, not stored), which argues for tall waves;
  // measured, 16 rows are as good as it gets and taller ones lose on some boxes by a third (64 frames of 1080p luma in 92-row
  // waves: 3.03 us on one box, 2.08 on the next; 16-row waves 2.9 / 2.2, 8-row waves 2.55 / 2.45) -- thousands of waves each
  // streaming down its own column strip.  Below 16 the launch's ROUNDS decide (256 CUs x the kernel's workgroups per CU; 1.2
  // rounds cost 2): the cheapest (rounds x rows walked per wave) wins, a single frame ends up in the shortest waves that
  // still fit one round.
  bool uv = false;
  for (int k = 0; k < a.njobs; ++k)
    uv = uv || (a.job[k].channels == 2 && a.job[k].kind == 0);
  const int occ = taps == 6 ? 6 : 8; // workgroups per CU (registers: tests/test_kernel_resources.py)
  int rpw = 64;
  const int forced = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE); // 1 / 2 / 3: 8 / 2 / 64 rows per wave
  if (forced == 1) rpw = 8;
  else if (forced == 2) rpw = 2;
  else if (forced >= 11 && forced <= 40) rpw = 2 * (forced - 10); // (measurements)
  else if (forced != 3) {
    unsigned long long best = ~0ull;
    for (int r = 16; r >= 2; r -= 2) {
      const unsigned long long wgs = (unsigned long long)count(r, false) * (unsigned)n;
      const unsigned long long cost = ((wgs + 256ull * occ - 1) / (256ull * occ)) * (unsigned)(r + taps - 1);
      if (cost < best) {
        best = cost;
        rpw = r;
      }
    }
  }
  a.map = make_tile_map_linear(count(rpw, true), (u32)n);
  a.cols_rps = rpw;
  const dim3 grid = tile_grid(a.map);
#define VALI_UP2(T, TAPS)                                                                                   \
  do {                                                                                                      \
    if (uv) hipLaunchKernelGGL((k_resize_up2<T, TAPS, true>), grid, dim3(kBlock), 0, stream, a);            \
    else hipLaunchKernelGGL((k_resize_up2<T, TAPS, false>), grid, dim3(kBlock), 0, stream, a);              \
  } while (0)
  if (elem == 1) {
    if (taps == 6) VALI_UP2(uint8_t, 6);
    else VALI_UP2(uint8_t, 4);
  } else {
    if (taps == 6) VALI_UP2(uint16_t, 6);
    else VALI_UP2(uint16_t, 4);
  }
#undef VALI_UP2
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali
