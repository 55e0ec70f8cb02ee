// Lanczos-3 and bicubic resize (VALI_INTERP_LANCZOS / VALI_INTERP_CUBIC) at non-integer ratios.
//
// Lanczos-3 is the reference's ONLY resize filter: every nppiResize call site passes NPPI_INTER_LANCZOS
// (reference: src/TC/src/TaskResizeSurface.cpp:67,116,224,273; UDPlanar src/TC/src/UDSurface.cpp:45,72), so this
// is what PySurfaceResizer and the planar PySurfaceUD run by default.  Restated as a separable TAPS x TAPS
// interpolating kernel on NPP's sampling grid src = dst * (src_size / dst_size): 3 lobes, normalised taps --
// pinned against NPP output at a non-integer ratio (45.6 dB through JPEG noise, every alternative filter / grid
// lower: tests/test_oracle_reference_pins.py).  Bicubic = Keys / Catmull-Rom (a = -1/2), TAPS = 4.
// Specification (oracle/vali_oracle.c: resize_plane_taps), per plane:
//   f = x * scale ; i = floor(f) ; a = f - i ; taps i - (TAPS/2 - 1) .. i + TAPS/2, indices clamped to the plane
//   rows:    e = w0 t0 ; e = fma(w_k, t_k, e), k even ; o likewise over the odd taps ; h = e + o
//   columns: v = wy0 h0 ; v = fma(wy_r, h_r, v)
//   u8 / u16: round-half-even + saturate ; f32: v
// (integer ratios never come here: every weight is 0 or 1 and the bilinear kernel's point form gives the same
// bytes, resize.hip.)
//
// Work decomposition: every plane is treated as a 1-channel plane of ELEMENTS (a pixel = ES interleaved elements:
// 1 for Y and the planes of planar formats, 2 for the UV plane of NV12 / P10, 3 for packed RGB; taps are ES elements
// apart), so all planes of a surface share one launch and one register budget.  One wave = 256 elements x ROWS dst
// rows, walked top to bottom over the SOURCE rows it needs: each source row goes through the wave's LDS stage once
// (16-byte coalesced loads, 4 rows in flight), is filtered horizontally by element-interleaved lanes (lane l:
// elements l, l+64, l+128, l+192 -- neighbouring lanes read neighbouring LDS dwords; a lane reads the aligned dwords
// covering its taps and funnel-shifts them with v_alignbyte_b32) and joins a ring of TAPS filtered rows in LDS; a
// dst row is emitted when its window is complete, by lanes that own 4 ADJACENT elements (the ring transposes the
// mapping: one float4 per ring row, one wide store).  ROWS = 32 when the launch has tiles to spare (batches): the
// TAPS - 1 rows two neighbouring waves both filter and the tap-weight set-up are spread over 4x the rows; 8 or 2
// for small launches, where a lone wave is bound by its own instruction latency.  What bounds it: FP32 VALU
// (per filtered sample 2 funnel shifts + 6 converts + 3 packed FMA + 1 add; 44 lane-instructions per output sample
// at a 2:1 ratio, VALU busy 62 %, LDS pipe 44 %) -- profiles/r02_lanczos.md has the counters of this and of the four
// designs it replaced.
#include "resize_common.hpp"
#include "resize_weights.hpp"

#include <type_traits>

namespace vali {

// The horizontal filter of one pixel-channel: even and odd taps accumulate in the two halves of packed FP32
// registers (v_pk_mul_f32 / v_pk_fma_f32), h = e + o.  wp[j] = (w[2j], w[2j+1]).
template <int TAPS>
__device__ __forceinline__ float taps_dot(const v2f32 (&wp)[TAPS / 2], const float (&t)[TAPS]) {
  v2f32 acc = wp[0] * (v2f32){t[0], t[1]};
#pragma unroll
  for (int j = 1; j < TAPS / 2; ++j)
    acc = __builtin_elementwise_fma(wp[j], (v2f32){t[2 * j], t[2 * j + 1]}, acc);
  return acc.x + acc.y;
}

// Stage layout of one wave (bytes): [kLzPadL: replicas of pixel 0][the 16-byte chunks of the source span][kLzPadR:
// replicas of the last pixel + slack for whole-dword reads].  With the pads the taps of EVERY element -- clamped at
// an image edge or not -- are TAPS elements ES apart in the LDS row.
//
// How a lane fetches its taps was settled by counters (profiles/r02_lanczos.md), in this order:
//   * TAPS x ES single-element ds_reads (round 1): 24 LDS instructions per lane and source row, LDS-bound;
//   * one MISALIGNED ds_read_b64 at the element's byte address: SQ_LDS_UNALIGNED_STALL = 84 % of the LDS pipe's
//     busy cycles, waves waiting on LDS 41 % of their life -- no faster;
//   * the span staged as FLOATS, taps as dword pairs: with 4 adjacent elements per lane the lanes sit 4 x ratio
//     dwords apart (8-way bank conflicts at 2:1, 2x slower); with the element-interleaved mapping below no
//     conflicts between lanes, but a 32-bit LDS read has 32 banks (2-way at 2:1) and the four ds_write_b128 that
//     stage a chunk of floats cost 52 LDS cycles: LDS pipe 83 % busy;
//   * now: the span stays BYTES (one ds_write_b128 per chunk), a lane reads the ALIGNED dwords that cover its taps
//     (neighbouring lanes read the same or the next dword: broadcast, no conflicts) and funnel-shifts them by its
//     own byte phase (v_alignbyte_b32 takes the shift from a register): the taps then sit at compile-time byte
//     positions and convert with v_cvt_f32_ubyteN.
constexpr int kLzPadL = 32, kLzPadR = 64;
constexpr int kLzCpr = 4;                                 // at most 4 x 64 chunks of 16 source bytes per row
constexpr int kLzStageCap = kLzCpr * kWave * 16 + 96;     // LDS bytes of a wave's stage, pads included

// The TAPS taps of one element (ES elements apart) from the staged row at byte offset `o` -> floats.
template <typename T, int ES, int TAPS>
__device__ __forceinline__ void read_taps(const uint8_t* stage, int o, float (&t)[TAPS]) {
  constexpr int EB = (int)sizeof(T);
  if constexpr (EB == 4) {
    const float* f = reinterpret_cast<const float*>(stage + o);
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
      t[k] = f[k * ES];
  } else {
    constexpr int BYTES = (TAPS - 1) * ES * EB + EB, ND = (BYTES + 3) / 4; // packed dwords after the shift
    const u32* wp = reinterpret_cast<const u32*>(stage + (o & ~3));
    const u32 sh = (u32)o & 3u;
    u32 w[ND + 1], d[ND];
#pragma unroll
    for (int i = 0; i <= ND; ++i)
      w[i] = wp[i];
#pragma unroll
    for (int i = 0; i < ND; ++i)
      d[i] = __builtin_amdgcn_alignbyte(w[i + 1], w[i], sh);
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
      constexpr int dummy = 0; (void)dummy;
      const int b = k * ES * EB;
      if constexpr (EB == 1) t[k] = (float)((d[b / 4] >> (8 * (b % 4))) & 0xffu);
      else t[k] = (float)((d[b / 4] >> (8 * (b % 4))) & 0xffffu);
    }
  }
}

// One wave: 256 ELEMENTS x ROWS rows of a plane whose pixels are ES interleaved elements (ES = 1: Y and the planes
// of planar formats, 2: the UV plane of NV12 / P10, 3: packed RGB).  Working on elements makes every format the
// 1-channel problem: element e = (pixel e / ES, channel e % ES), its taps are ES elements apart.
//   horizontal pass: lane l filters elements l, l + 64, l + 128, l + 192 of the tile -- neighbouring lanes read
//     neighbouring LDS dwords (bank-conflict free at every ratio; with 4 ADJACENT elements per lane the lanes sit
//     4 x ratio dwords apart: 8-way conflicts at 2:1, measured 2x slower than not staging floats at all);
//   the filtered row goes into the wave's LDS ring at its element index, which also TRANSPOSES the mapping:
//   vertical pass: lane l reads elements 4l .. 4l+3 of the TAPS ring rows as one float4 each and stores 4 adjacent
//     elements.
template <typename T, int ES, int TAPS, int ROWS>
__device__ __forceinline__ void taps_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                          int dw, int dh, u32 tx, u32 ty, int stage_bytes, uint8_t* stage,
                                          float* ring) {
  static_assert(ROWS <= kWave, "row taps are evaluated one row per lane");
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int EB = (int)sizeof(T);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int e0 = tx * 256;                                  // first element of the tile
  const int dwe = dw * ES;                                  // elements per dst row
  const int y_first = (ty * kWavesPerBlock + wave) * ROWS;  // wave-uniform
  if (y_first >= dh)
    return;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  const int eb = e0 + 4 * lane;                             // vertical pass / store: 4 adjacent elements
  const int n = min(4, dwe - eb);

  // row taps: lane r evaluates row y_first + r, read back as scalars
  const LzTap<TAPS> vy = make_lz_tap<TAPS>(y_first + (lane & (ROWS - 1)), scale_y);
  auto row_tap = [&](int rr) {
    LzTap<TAPS> t;
    t.i = __builtin_amdgcn_readlane(vy.i, rr);
#pragma unroll
    for (int k = 0; k < TAPS; ++k)
      t.w[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vy.w[k]), rr));
    return t;
  };

  // wave-uniform source span of the tile, in pixels (unclamped, then clamped)
  const int px_first = e0 / ES, px_last = min(e0 + 255, dwe - 1) / ES;
  const int ux0 = (int)__builtin_floorf((float)px_first * scale_x) - kBefore;
  const int ux1 = (int)__builtin_floorf((float)px_last * scale_x) + TAPS - 1 - kBefore;
  const int sx0 = clampi(ux0, sw - 1), sx1 = clampi(ux1, sw - 1);
  const int byte_begin = (sx0 * ES * EB) & ~15;
  const int nbytes = (((sx1 + 1) * ES * EB + 15) & ~15) - byte_begin;
  const bool staged = nbytes + kLzPadL + kLzPadR <= stage_bytes && ((((uintptr_t)sp) | (uintptr_t)spitch) & 15u) == 0;

  if (staged) {
    const int nchunks = nbytes >> 4;
    const int cpr = (nchunks + kWave - 1) >> 6;                // 16-byte chunks per lane per row: 1..4, uniform
    // horizontal pass: this lane's 4 elements, their (even, odd) weight pairs and the LDS byte offset of their first tap
    // (source byte B of the row sits at kLzPadL + B - byte_begin, also the replicas of pixels j < 0 and j >= sw)
    v2f32 wq[4][TAPS / 2];
    int lo[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int e = min(e0 + p * kWave + lane, dwe - 1);
      const int px = e / ES, ch = e - px * ES;
      const LzTap<TAPS> c = make_lz_tap<TAPS>(px, scale_x);
#pragma unroll
      for (int j = 0; j < TAPS / 2; ++j)
        wq[p][j] = (v2f32){c.w[2 * j], c.w[2 * j + 1]};
      lo[p] = kLzPadL + ((min(c.i, sw) - kBefore) * ES + ch) * EB - byte_begin;
    }
    const bool pad_left = ux0 < 0, pad_right = ux1 > sw - 1;   // wave-uniform
    const int edge = kLzPadL + (sw - 1) * ES * EB - byte_begin; // the last pixel of the row
    // The wave walks the source rows it needs in order -- s_begin .. s_end, each staged and filtered ONCE -- and
    // emits every dst row whose window is complete.  Loads run D rows ahead of the row being filtered, in a ring
    // of register sets indexed statically (the walk is unrolled by D): with one row of prefetch a lone wave
    // stalled a full memory latency per source row -- 32 us for a single 2160p frame; batches hide it behind
    // other waves.  D x CPR <= 8 uint4 of registers: narrow spans (the common 8-bit case) look 4 rows ahead.
    const LzTap<TAPS> first = row_tap(0);
    const int last_rr = min(ROWS, dh - y_first) - 1;
    const int s_begin = first.i - kBefore;
    const int s_end = __builtin_amdgcn_readlane(vy.i, last_rr) + TAPS - 1 - kBefore;
    bool mine[kLzCpr];
    int chunk[kLzCpr]; // byte offset of the lane's chunk c in the source span
#pragma unroll
    for (int c = 0; c < kLzCpr; ++c) {
      mine[c] = lane + c * kWave < nchunks;
      chunk[c] = min(lane + c * kWave, nchunks - 1) * 16;
    }
    auto walk = [&](auto cpr_tag, auto depth_tag) {
      constexpr int CPR = decltype(cpr_tag)::value, D = decltype(depth_tag)::value;
      uint4 pf[D][CPR];
      auto issue = [&](int logical, uint4 (&q)[CPR]) {
        const uint8_t* row = sp + (u32)(clampi(logical, sh - 1) * spitch + byte_begin);
#pragma unroll
        for (int c = 0; c < CPR; ++c)
          q[c] = gload16(row + chunk[c]);
      };
      // (the scheduling barriers keep the rows in ISSUE order: vmcnt retires in order, and the compiler's scheduler
      // had put row 0 last -- the first wait of every trip through the unrolled walk then drained the whole ring)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        issue(s_begin + j, pf[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
      int rr = 0;          // next dst row to emit
      int slot = 0;        // ring slot the next source row goes to = (row - s_begin) mod TAPS
      int want_end = s_begin + TAPS - 1; // last source row of dst row rr's window
#pragma unroll 1
      for (int s0 = s_begin; s0 <= s_end; s0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const int cur = s0 + j;
          // Rows past s_end (the last trip only) skip the work but NOT the load: every path through the body issues
          // the same loads in the same order, so the compiler can count them (vmcnt retires in order).  With an
          // early exit here the walk was re-entered, on paper, right behind a fresh issue and every wait drained
          // the whole ring -- the 4-row prefetch ran 1 row deep.
          const bool live = cur <= s_end; // wave-uniform
          if (live) {
#pragma unroll
            for (int c = 0; c < CPR; ++c)
              if (mine[c])
                *reinterpret_cast<uint4*>(stage + kLzPadL + chunk[c]) = pf[j][c];
            wave_lds_sync();
          }
          issue(cur + D, pf[j]); // D rows ahead, in flight while this and the next D - 1 rows are filtered
          if (!live)
            continue;
          if (pad_left || pad_right) { // image edges: replicate the first / last pixel into the pads
            constexpr int PB = ES * EB; // bytes per pixel
            if (pad_left && lane < kBefore * PB)
              stage[kLzPadL - kBefore * PB + lane] = stage[kLzPadL + lane % PB];
            if (pad_right && lane < (TAPS - kBefore) * PB)
              stage[edge + PB + lane] = stage[edge + lane % PB];
            wave_lds_sync();
          }
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            float t[TAPS];
            read_taps<T, ES, TAPS>(stage, lo[p], t);
            ring[slot * 256 + p * kWave + lane] = taps_dot<TAPS>(wq[p], t); // replaces the oldest row
          }
          slot = slot == TAPS - 1 ? 0 : slot + 1;
          wave_lds_sync(); // the strip is re-filled by the next row; the ring row is read by other lanes
          // Every dst row whose window ends with this source row: one at most when shrinking, several when enlarging.
          // The first is straight-line code and only the others loop: the compiler drains vmcnt in front of a loop
          // that stores without loading (its heuristic for gfx9's shared load/store counter), which would throw the
          // prefetched rows away at every emitted row.
          auto emit = [&]() {
            const LzTap<TAPS> cy = row_tap(rr);
            if (n > 0) {
              // the window's first row sits TAPS slots behind the next free one
              v2f32 v01, v23;
#pragma unroll
              for (int r = 0; r < TAPS; ++r) {
                const int sl = slot + r >= TAPS ? slot + r - TAPS : slot + r; // logical row r of the window
                const float4 f = *reinterpret_cast<const float4*>(ring + sl * 256 + 4 * lane);
                const v2f32 wr = (v2f32){cy.w[r], cy.w[r]};
                v01 = r == 0 ? wr * (v2f32){f.x, f.y} : __builtin_elementwise_fma(wr, (v2f32){f.x, f.y}, v01);
                v23 = r == 0 ? wr * (v2f32){f.z, f.w} : __builtin_elementwise_fma(wr, (v2f32){f.z, f.w}, v23);
              }
              const float res[4][1] = {{v01.x}, {v01.y}, {v23.x}, {v23.y}};
              store_px4<T, 1>(dp + (u32)((y_first + rr) * dpitch) + (size_t)eb * EB, res, (1u << n) - 1u);
            }
            ++rr;
            want_end = rr <= last_rr ? __builtin_amdgcn_readlane(vy.i, rr) + TAPS - 1 - kBefore : 0x7fffffff;
          };
          if (rr <= last_rr && want_end == cur) {
            emit();
            while (rr <= last_rr && want_end == cur)
              emit();
          }
        }
      }
    };
    if (cpr == 1)
      walk(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
    else if (cpr == 2)
      walk(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
    else
      walk(std::integral_constant<int, kLzCpr>{}, std::integral_constant<int, 2>{});
    return;
  }

  // Source span wider than the stage (very large downscales) or foreign unaligned memory: direct gather, TAPS^2
  // taps per element, 4 adjacent elements per lane.  (No early exit for lanes without elements: row_tap() reads
  // other lanes with v_readlane, see DESIGN.md "v_readlane after a per-lane exit".)
  v2f32 wq[4][TAPS / 2];
  int ti[4], tch[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int e = min(eb + p, dwe - 1);
    const int px = e / ES;
    tch[p] = e - px * ES;
    const LzTap<TAPS> c = make_lz_tap<TAPS>(px, scale_x);
    ti[p] = c.i - kBefore;
#pragma unroll
    for (int j = 0; j < TAPS / 2; ++j)
      wq[p][j] = (v2f32){c.w[2 * j], c.w[2 * j + 1]};
  }
#pragma unroll 1
  for (int rr = 0; rr < ROWS; ++rr) {
    const int y = y_first + rr;
    if (y >= dh)
      break;
    const LzTap<TAPS> cy = row_tap(rr);
    if (n <= 0)
      continue;
    float res[4][1];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float v = 0.0f;
#pragma unroll
      for (int r = 0; r < TAPS; ++r) {
        const uint8_t* row = sp + (size_t)clampi(cy.i - kBefore + r, sh - 1) * spitch;
        float t[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
          t[k] = (float)gload<T>(row + (size_t)(clampi(ti[p] + k, sw - 1) * ES + tch[p]) * EB);
        const float h = taps_dot<TAPS>(wq[p], t);
        v = r == 0 ? cy.w[0] * h : __builtin_fmaf(cy.w[r], h, v);
      }
      res[p][0] = v;
    }
    store_px4<T, 1>(dp + (u32)(y * dpitch) + (size_t)eb * EB, res, (1u << n) - 1u);
  }
}

// ESSET: which element strides the surface's planes have -- 1: one-channel planes only (Y, YUV4xx, RGB_PLANAR),
// 12: a one-channel and a two-channel plane (NV12 / P10: Y + UV), 3: packed RGB.  All planes go in ONE launch.
template <typename T, int ESSET, int TAPS, int ROWS>
__global__ void __launch_bounds__(kBlock) k_resize_taps(const ResizeArgs a) {
  extern __shared__ uint4 taps_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* const stage = reinterpret_cast<uint8_t*>(taps_lds) + wave * a.lds_per_wave;
  float* const ring = reinterpret_cast<float*>(stage + a.stage_bytes);
  const int cap = a.force_gather ? 0 : a.stage_bytes;
  if constexpr (ESSET == 3) {
    taps_tile<T, 3, TAPS, ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, cap, stage, ring);
  } else {
    if (ESSET == 12 && job.channels == 2)
      taps_tile<T, 2, TAPS, ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, cap, stage, ring);
    else
      taps_tile<T, 1, TAPS, ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, cap, stage, ring);
  }
}

template <typename T, int ESSET, int TAPS>
static void launch_rows(const ResizeArgs& a, int rows, dim3 grid, unsigned lds, hipStream_t stream) {
  if (rows == 32)
    hipLaunchKernelGGL((k_resize_taps<T, ESSET, TAPS, 32>), grid, dim3(kBlock), lds, stream, a);
  else if (rows == 8)
    hipLaunchKernelGGL((k_resize_taps<T, ESSET, TAPS, 8>), grid, dim3(kBlock), lds, stream, a);
  else
    hipLaunchKernelGGL((k_resize_taps<T, ESSET, TAPS, 2>), grid, dim3(kBlock), lds, stream, a);
}

int launch_resize_taps(const ResizeArgs& base, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream) {
  const bool gather_only = tuning(VALI_TUNE_RESIZE_FORCE_GATHER) == 1;
  // A/B and path coverage: 1 / 2 / 3 force 8- / 2- / 32-row waves whatever the size of the launch
  const int force = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE);
  ResizeArgs a = base;
  int span = 0, esset = 0;
  for (int k = 0; k < a.njobs; ++k) {
    const ResizeJob& j = a.job[k];
    const int c = j.channels;
    esset = c == 3 ? 3 : (c == 2 || esset == 12) ? 12 : (esset ? esset : 1);
    const int sw = src_w >> j.ssub_x, dw = dst_w >> j.sub_x;
    // bytes of source a 256-ELEMENT tile row spans (taps, 16-byte alignment slop on both sides)
    const long long px = ((long long)(255 / c + 1) * sw + dw - 1) / dw + taps + 2;
    const long long bytes = px * c * elem + 32;
    span = (int)(bytes > span ? (bytes > (1 << 20) ? (1 << 20) : bytes) : span);
  }
  // tiles: 256 elements x (4 waves x ROWS) rows
  auto count = [&](int rows, bool assign) {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dw = dst_w >> a.job[k].sub_x, dh = dst_h >> a.job[k].sub_y;
      const u32 tiles_x = (u32)(dw * a.job[k].channels + 255) / 256;
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (u32)((dh + kWavesPerBlock * rows - 1) / (kWavesPerBlock * rows));
    }
    return total;
  };
  // (8-row waves from 320 workgroups on, re-measured with the prefetch really 4 rows deep: ONE NV12 2160p -> 1088p frame
  // = 352 such workgroups 15.0 us against 17.4 through 2-row waves, two Y frames 17.6 against 21.3; one Y frame = 280
  // workgroups stays on 2-row waves, 13.6 against 14.7)
  // rows per wave: 32 when the launch still fills the chip 4 times over (batches), 2 when 8-row waves would
  // leave SIMDs idle -- a lone wave is bound by its own instruction latency, so a single frame is cut into many
  // short waves, at the price of filtering more rows twice
  const unsigned long long t32 = (unsigned long long)count(32, false) * (unsigned)n, t8 = (unsigned long long)count(8, false) * (unsigned)n;
  // (float planes: 8-row waves measured 4 % faster than 32-row ones on batches -- RGB_32F 2160p -> 1080p 22.0 vs 23.0 us --
  // their rows are 4x the bytes, so the 5 rows two neighbouring tiles both filter weigh less than the longer tail)
  const int rows = force == 1 ? 8 : force == 2 ? 2 : force == 3 ? 32 : (t32 >= 1024ull && elem < 4) ? 32 : t8 >= 320ull ? 8 : 2;
  a.map = make_tile_map_linear(count(rows, true), (u32)n);
  a.force_gather = gather_only ? 1 : 0;
  a.stage_bytes = (span + kLzPadL + kLzPadR <= kLzStageCap && !gather_only) ? ((span + 15) & ~15) + kLzPadL + kLzPadR : 0;
  a.lds_per_wave = a.stage_bytes + taps * 256 * (int)sizeof(float);
  const unsigned lds = (unsigned)a.lds_per_wave * kWavesPerBlock;
  const dim3 grid = tile_grid(a.map);
#define VALI_TAPS_T(T)                                                                        \
  do {                                                                                        \
    if (taps == 6) {                                                                          \
      if (esset == 1) launch_rows<T, 1, 6>(a, rows, grid, lds, stream);                        \
      else if (esset == 12) launch_rows<T, 12, 6>(a, rows, grid, lds, stream);                 \
      else launch_rows<T, 3, 6>(a, rows, grid, lds, stream);                                   \
    } else {                                                                                  \
      if (esset == 1) launch_rows<T, 1, 4>(a, rows, grid, lds, stream);                        \
      else if (esset == 12) launch_rows<T, 12, 4>(a, rows, grid, lds, stream);                 \
      else launch_rows<T, 3, 4>(a, rows, grid, lds, stream);                                   \
    }                                                                                         \
  } while (0)
  if (elem == 1) VALI_TAPS_T(uint8_t);
  else if (elem == 2) VALI_TAPS_T(uint16_t);
  else VALI_TAPS_T(float);
#undef VALI_TAPS_T
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali
