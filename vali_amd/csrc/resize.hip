// Surface resize, all planes of a surface in ONE launch.
//
// Replaces the reference's ResizeSurface implementations, which call nppiResize_8u_C1R /
// nppiResize_8u_C3R / nppiResize_32f_C{1,3}R per plane and round-trip NV12 through two
// temporary YUV420 surfaces: NV12 -> YUV420 -> 3x nppiResize -> YUV420 -> NV12, five kernels
// and ~41 MB of traffic for a 2160p -> 720p frame
// (reference: src/TC/src/TaskResizeSurface.cpp:34-286; NV12 path :132-188).  Here the Y
// plane and the interleaved UV plane (as a 2-channel image) are resized by one kernel.
//
// Interpolation: BILINEAR (BASELINE config 3, the default) or LANCZOS-3 on NPP's sampling
// grid.  The reference hard-codes NPPI_INTER_LANCZOS (TaskResizeSurface.cpp:67,116,224,273)
// and has no switch; the Lanczos kernel is at the end of this file.  GEOMETRY is pinned by the reference's own
// fixture: tests/data/test_small.nv12 (the expected 848x464 -> 424x232 output of
// tests/test_PySurfaceResizer.py) equals src[2y][2x] of the frame (41.7 dB through JPEG noise)
// and NOT the centre-aligned (x+0.5)*s-0.5 sample (26.4 dB) -- tests/test_oracle_resize.py.
// So nppiResize maps dst -> src as  src = dst * (src_size / dst_size), no half-pixel shift.
// Specification (oracle: vali_oracle_resize_plane), per plane (sw x sh) -> (dw x dh):
//   scale = (float)sw / (float)dw
//   fx = x * scale ; i = floor(fx) ; a = fx - i
//   i = min(i, sw-1) ; i1 = min(i+1, sw-1)            (same for y -> j, b)
//   t0 = fma(a, T[j][i1]-T[j][i], T[j][i]) ; t1 on row j1 ; v = fma(b, t1-t0, t0)
//   u8/u16: round-half-even + saturate ; f32: v
// One lane = 4 adjacent dst pixels of one plane row; a workgroup = 256 x 16 dst pixels of one
// plane; plane jobs are concatenated in the XCD-contiguous TileMap.  HBM traffic = touched
// source rows + dst.
#include "resize_common.hpp"
#include <type_traits>

namespace vali {

template <typename T> __device__ __forceinline__ float rs_load(const uint8_t* row, int idx) {
  return (float)((const T*)row)[idx];
}
template <typename T> __device__ __forceinline__ T rs_finish(float v);
template <> __device__ __forceinline__ uint8_t rs_finish<uint8_t>(float v) { return (uint8_t)quantize_u8(v); }
template <> __device__ __forceinline__ uint16_t rs_finish<uint16_t>(float v) {
  float r = __builtin_rintf(v);
  return (uint16_t)__builtin_fminf(__builtin_fmaxf(r, 0.0f), 65535.0f);
}
template <> __device__ __forceinline__ float rs_finish<float>(float v) { return v; }

// Per-wave LDS staging of the two source rows a dst row needs.  A wave walks kRsRowsPerWave
// dst rows x 256 px with the SAME column taps (coordinates, float divisions and LDS offsets are
// paid once per tile, not per row).  For each dst row it reads the source span
// [byte_begin, byte_begin + nbytes) of rows i0 and i1 with 16-byte coalesced loads -- lane l
// owns chunks l, l+64, ... of both rows -- then gathers its texels from LDS: every 128-byte
// line is fetched ONCE instead of once per byte-gather instruction (the direct gather
// re-requested each line ~8x from L2; profiles/r01_secondary.md).  The loads of row r+1 are
// issued before row r is sampled.  A row whose vertical weight is exactly 0 (integer scale
// factors: BASELINE config 3) never fetches its second source row: fma(0, t1 - t0, t0) == t0.
constexpr int kRsRowsPerWave = 8;
constexpr int kRsTileH = kWavesPerBlock * kRsRowsPerWave;
constexpr int kRsChunks = 3;                             // 16-byte chunks per lane per row
constexpr int kStageRowBytes = kRsChunks * kWave * 16;   // 3072 B = 12x downscale for u8 C1
constexpr int kStageRowAlloc = stage_alloc(kStageRowBytes);
struct alignas(16) StageRows {
  uint8_t row[2][kStageRowAlloc];
};
// 4 pixels of raw elements (no arithmetic: the point-sample path) -> memory, like store_px4
template <typename T, int C>
__device__ __forceinline__ void store_px4_raw(uint8_t* dst, const u32 (&e)[4][C], u32 mask) {
  constexpr int E = (int)sizeof(T), N = 4 * C, NB = N * E, PER = 4 / E;
  u32 w[NB / 4];
#pragma unroll
  for (int k = 0; k < NB / 4; ++k)
    w[k] = 0;
#pragma unroll
  for (int k = 0; k < N; ++k)
    w[k / PER] |= e[k / C][k % C] << (8 * E * (k % PER));
  constexpr u32 kAlign = NB % 16 == 0 ? 15u : (NB % 8 == 0 ? 7u : 3u);
  if (mask == 0xfu && (((uintptr_t)dst) & kAlign) == 0) {
    if constexpr (NB % 16 == 0) {
#pragma unroll
      for (int k = 0; k < NB / 16; ++k) {
        const v4u32 q = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
        ((VALI_GLOBAL v4u32*)dst)[k] = q;
      }
    } else if constexpr (NB % 8 == 0) {
#pragma unroll
      for (int k = 0; k < NB / 8; ++k) {
        const v2u32 q = {w[2 * k], w[2 * k + 1]};
        ((VALI_GLOBAL v2u32*)dst)[k] = q;
      }
    } else if constexpr (NB == 12) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 q = {w[0], w[1], w[2]};
      *(VALI_GLOBAL v3u32*)dst = q;
    } else {
#pragma unroll
      for (int k = 0; k < NB / 4; ++k)
        ((VALI_GLOBAL u32*)dst)[k] = w[k];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (mask & (1u << (k / C)))
      ((VALI_GLOBAL T*)dst)[k] = (T)e[k / C][k % C];
}

// POINT (integer element types, integer scale factors on both axes -- BASELINE config 3's exact 3x):
// f = x * k is an exact integer, every weight is 0 and the result is src[ky y][kx x] whatever the
// filter; the tile then skips the coordinate divisions, the second source row, the float
// conversions and the lerps -- the same bytes as the arithmetic path (tests/test_gpu_resize.py).
template <typename T, int C, bool POINT = false>
__device__ __forceinline__ void resize_tile(const uint8_t* sp, int spitch, int sw, int sh,
                                            uint8_t* dp, int dpitch, int dw, int dh, u32 tx,
                                            u32 ty, StageRows* stage_all, int rows) {
  static_assert(!POINT || sizeof(T) < 4, "float planes keep the arithmetic (0 * inf must stay NaN)");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int x0 = (tx * 64 + lane) * 4;
  // `rows` dst rows per wave: kRsRowsPerWave for batches, 4 or 2 when the launch is small (launch_resize): a wave walks
  // its rows one after the other, so a lone frame is done sooner by more, shorter waves
  const int y_first = ((int)ty * kWavesPerBlock + wave) * rows; // wave-uniform
  if (y_first >= dh)
    return;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  constexpr int PB = C * (int)sizeof(T);
  constexpr bool PAD = kStagePadded<PB>;

  // column taps of the lane's 4 pixels (tail lanes clamp to the last column)
  Lerp lx[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    if constexpr (POINT) {
      lx[p].i0 = lx[p].i1 = min(x0 + p, dw - 1) * (sw / dw);
      lx[p].a = 0.0f;
    } else {
      lx[p] = make_lerp(min(x0 + p, dw - 1), scale_x, sw);
    }
  }
  const int n = min(4, dw - x0); // valid pixels (<= 0: tail lane, staging only)

  // row taps: lane r evaluates row y_first + r, read back as scalars
  Lerp vly;
  if constexpr (POINT) {
    vly.i0 = vly.i1 = min(y_first + (lane & (kRsRowsPerWave - 1)), dh - 1) * (sh / dh);
    vly.a = 0.0f;
  } else {
    vly = make_lerp(y_first + (lane & (kRsRowsPerWave - 1)), scale_y, sh);
  }
  const int last_rr = min(rows, dh - y_first) - 1; // wave-uniform, >= 0
  auto row_lerp = [&](int rr) {
    Lerp l;
    l.i0 = __builtin_amdgcn_readlane(vly.i0, rr);
    l.i1 = __builtin_amdgcn_readlane(vly.i1, rr);
    l.a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, vly.a), rr));
    return l;
  };

  // wave-uniform source span of this tile: first tap of lane 0 .. last tap of lane 63
  const int sx0 = __builtin_amdgcn_readlane(lx[0].i0, 0), sx1 = __builtin_amdgcn_readlane(lx[3].i1, 63);
  const int byte_begin = (sx0 * PB) & ~15;
  const int nbytes = (((sx1 + 1) * PB + 15) & ~15) - byte_begin;
  const bool staged = stage_all != nullptr && nbytes <= kStageRowBytes && ((((uintptr_t)sp) | (uintptr_t)spitch) & 15u) == 0;

  // The sampling code is instantiated twice (LDS rows / global rows) so each copy gets typed
  // ds_read / global_load instructions; all texels of the lane are fetched before any
  // arithmetic so the loads overlap.
  auto sample_and_store = [&](const Lerp& ly, int y, auto fetch) {
    float t[4][4][C]; // [pixel][00,10,01,11][channel]
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      fetch(0, p, 0, t[p][0]); fetch(0, p, 1, t[p][1]);
      fetch(1, p, 0, t[p][2]); fetch(1, p, 1, t[p][3]);
    }
    float res[4][C];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const float t0 = __builtin_fmaf(lx[p].a, t[p][1][ch] - t[p][0][ch], t[p][0][ch]);
        const float t1 = __builtin_fmaf(lx[p].a, t[p][3][ch] - t[p][2][ch], t[p][2][ch]);
        res[p][ch] = __builtin_fmaf(ly.a, t1 - t0, t0);
      }
    store_px4<T, C>(dp + (u32)(y * dpitch) + (size_t)x0 * PB, res, (1u << n) - 1u);
  };

  if (staged) {
    StageRows& st = stage_all[wave];
    const int nchunks = nbytes / 16;
    int lo[4][2]; // LDS byte offsets of the column taps (row-invariant)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      lo[p][0] = stage_off<PAD>(lx[p].i0 * PB - byte_begin);
      lo[p][1] = stage_off<PAD>(lx[p].i1 * PB - byte_begin);
    }
    // Register prefetch pipeline, DEPTH dst rows ahead, CPR 16-byte chunks per lane per row.
    // HBM latency x per-CU bandwidth share needs ~46 KB in flight per CU; one row of a
    // narrow span (768 B at 3x) is far too little, so narrow spans run 4 rows ahead
    // (<1, 4>), wide ones 1 row ahead with 3 chunks (<3, 1>).  The row loop is fully
    // unrolled so the ring of prefetch registers is indexed statically.
    auto pipeline = [&](auto cpr_tag, auto depth_tag) {
      constexpr int CPR = decltype(cpr_tag)::value, DEPTH = decltype(depth_tag)::value;
      uint4 pf[DEPTH][2][CPR];
      // Loads are unconditional and straight-line (the compiler then counts vmcnt instead of
      // draining at every branch): lanes past the span re-read its last chunk, and a row with
      // vertical weight 0 re-reads row i0 (same lines: no extra HBM traffic).
      auto issue = [&](int rr, uint4 (&q)[2][CPR]) {
        const Lerp ly = row_lerp(rr);
        // (float planes always fetch both rows: 0 * inf must stay NaN)
        const bool two = sizeof(T) == 4 || ly.a != 0.0f;
        // scalar row addresses; a plane is < 4 GiB
        const uint8_t* r0 = sp + (u32)(ly.i0 * spitch + byte_begin);
        const uint8_t* r1 = sp + (u32)((two ? ly.i1 : ly.i0) * spitch + byte_begin);
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
          const int k = min(lane + i * kWave, nchunks - 1);
          // non-temporal: 2x and steeper shrinks read every row once (+6..8 %); gentler ratios and enlargements
          // re-read rows out of L2 all the same (measured equal, 4x enlargement -2.5 %)
          q[0][i] = gload16_nt(r0 + k * 16);
          if constexpr (!POINT)
            q[1][i] = gload16_nt(r1 + k * 16);
        }
      };
      auto commit = [&](const uint4 (&q)[2][CPR]) {
#pragma unroll
        for (int i = 0; i < CPR; ++i) {
          const int k = lane + i * kWave;
          if (k < nchunks) {
            stage_put<PAD>(st.row[0], k, q[0][i]);
            if constexpr (!POINT)
              stage_put<PAD>(st.row[1], k, q[1][i]);
          }
        }
      };
      // Every issue is UNCONDITIONAL (rows past the tile's last one re-read it): behind a branch the compiler
      // cannot count the loads in flight and drains vmcnt to 0 at every commit, which exposes the whole memory
      // latency once per row (measured: the prefetch depth had no effect at all until this was straight-line).
#pragma unroll
      for (int k = 0; k < DEPTH; ++k)
        if (k < kRsRowsPerWave) {
          issue(min(k, last_rr), pf[k]);
          __builtin_amdgcn_sched_barrier(0); // rows stay in issue order (vmcnt retires in order)
        }
#pragma unroll
      for (int rr = 0; rr < kRsRowsPerWave; ++rr) {
        const int y = y_first + rr;
        if (rr > last_rr)
          break;
        commit(pf[rr % DEPTH]);
        wave_lds_sync();
        if (rr + DEPTH < kRsRowsPerWave)
          issue(min(rr + DEPTH, last_rr), pf[rr % DEPTH]); // in flight while rows rr .. rr+DEPTH-1 are sampled
        if constexpr (POINT) {
          if (n > 0) {
            u32 e[4][C];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              if constexpr (C == 2 && sizeof(T) == 1) { // the interleaved pair in one LDS read
                const u32 uv = *(const uint16_t*)(st.row[0] + lo[p][0]);
                e[p][0] = uv & 0xffu; e[p][1] = uv >> 8;
              } else {
#pragma unroll
                for (int ch = 0; ch < C; ++ch)
                  e[p][ch] = ((const T*)(st.row[0] + lo[p][0]))[ch];
              }
            }
            store_px4_raw<T, C>(dp + (u32)(y * dpitch) + (size_t)x0 * PB, e, (1u << n) - 1u);
          }
        } else if (n > 0) {
          sample_and_store(row_lerp(rr), y, [&](int r, int p, int t, float (&out)[C]) {
            if constexpr (C == 2 && sizeof(T) == 1) { // the interleaved pair in one LDS read
              const u32 uv = *(const uint16_t*)(st.row[r] + lo[p][t]);
              out[0] = (float)(uv & 0xffu); out[1] = (float)(uv >> 8);
            } else {
#pragma unroll
              for (int ch = 0; ch < C; ++ch)
                out[ch] = (float)((const T*)(st.row[r] + lo[p][t]))[ch];
            }
          });
        }
        wave_lds_sync(); // the strip is re-filled by the next row
      }
    };
    if (nchunks <= kWave)
      pipeline(std::integral_constant<int, 1>{}, std::integral_constant<int, POINT ? kRsRowsPerWave : 4>{});
    else if (nchunks <= 2 * kWave) // 2x of interleaved UV / 4x of a byte plane: 1024 + 16 bytes = 65 chunks
      pipeline(std::integral_constant<int, 2>{}, std::integral_constant<int, POINT ? 4 : 2>{});
    else
      pipeline(std::integral_constant<int, kRsChunks>{}, std::integral_constant<int, 1>{});
  } else {
    // (No early exit for lanes without pixels: row_lerp() reads lanes 0..7 with v_readlane, and a
    // lane that has left the kernel holds whatever the compiler computed for it AFTER the exit --
    // a ragged last tile of < 32 pixels would take its row taps from such lanes.)
#pragma unroll 1
    for (int rr = 0; rr <= last_rr; ++rr) {
      const int y = y_first + rr;
      const Lerp ly = row_lerp(rr);
      if (n > 0) {
        const uint8_t* srows[2] = {sp + (size_t)ly.i0 * spitch, sp + (size_t)ly.i1 * spitch};
        if (PairLoad<T, C>::kMerged && sw >= 3) { // both horizontal taps of a row from one unaligned load
          if constexpr (PairLoad<T, C>::kMerged) {
            float pre[2][4][2][C];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int p = 0; p < 4; ++p)
                load_tap_pair<T, C>(srows[r], lx[p].i0, sw, pre[r][p][0], pre[r][p][1]);
            sample_and_store(ly, y, [&](int r, int p, int t, float (&out)[C]) {
#pragma unroll
              for (int ch = 0; ch < C; ++ch)
                out[ch] = pre[r][p][t][ch];
            });
          }
        } else {
          sample_and_store(ly, y, [&](int r, int p, int t, float (&out)[C]) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
              out[ch] = (float)gload<T>(srows[r] + (size_t)(t ? lx[p].i1 : lx[p].i0) * PB + ch * sizeof(T));
          });
        }
      }
    }
  }
}

// MAXC = the largest channel count among the surface's plane jobs: the kernel is only as
// register-heavy as the format needs (planar formats never carry the packed-RGB code).
template <typename T, int MAXC, bool POINT = false>
__global__ void __launch_bounds__(kBlock) k_resize(const ResizeArgs a) {
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  __shared__ StageRows stage[kWavesPerBlock];
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  if (MAXC >= 3 && job.channels == 3)
    resize_tile<T, 3, POINT>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.force_gather ? nullptr : stage, a.rows);
  else if (MAXC >= 2 && job.channels == 2)
    resize_tile<T, 2, POINT>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.force_gather ? nullptr : stage, a.rows);
  else
    resize_tile<T, 1, POINT>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, a.force_gather ? nullptr : stage, a.rows);
}

// ---------------------------------------------------------------------------------------------------------
// Point sample at a small integer HORIZONTAL factor K = 2, 3 (8-bit planes of 1 or 2 channels: Y, the planes of
// YUV4xx / RGB_PLANAR, both planes of NV12 -- BASELINE config 3 is K = 3): no LDS and a quarter of the vector
// memory instructions of the staged point form (NV12 2160p -> 1080p 2.02 -> 1.75 us; config 3 1.11 -> 1.08).  A thread owns 16 dst BYTES of one row = the 16 K source bytes
// [K xb, K xb + 16 K): K dwordx4 loads, a compile-time byte selection (dst byte b <- source byte
// (b / PB) K PB + b % PB), one dwordx4 store.  A workgroup = 256 dst bytes x 32 rows (16 threads across, two rows
// per thread); any integer vertical factor.  Same bytes as every other form of the resizer at integer factors
// (src[ky y][K x]).  Rows that are not 16-byte aligned take the byte loop at the end.
constexpr int kPkRows = 32;
template <int K, int PB>
__device__ __forceinline__ uint4 pk_select(const u32 (&w)[4 * K]) {
  u32 o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const int s = (b / PB) * K * PB + b % PB;
    o[b / 4] |= ((w[s / 4] >> (8 * (s % 4))) & 0xffu) << (8 * (b % 4));
  }
  return make_uint4(o[0], o[1], o[2], o[3]);
}

template <int K>
__global__ void __launch_bounds__(kBlock) k_resize_pointk(const ResizeArgs a) {
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int pb = job.channels;                       // bytes per pixel (u8 planes): 1 or 2
  const int dbytes = v.dw * pb, sbytes = v.sw * pb;  // row lengths
  const int xb = (tx * 16 + (threadIdx.x & 15)) * 16;
  if (xb >= dbytes)
    return;
  const int ky = v.sh / v.dh;
  const bool aligned = ((((uintptr_t)v.sp) | (uintptr_t)v.spitch | ((uintptr_t)v.dp) | (uintptr_t)v.dpitch) & 15u) == 0;
  const int y0 = ty * kPkRows + (threadIdx.x >> 4);
  // The K source chunks of this lane all inside the ROW (not the pitch: a borrowed view may end where its last row ends):
  // everything but a row's last lane.  That lane takes the byte path below -- ONE branch around the vector path; a guard
  // per chunk put 238 predicated regions into this kernel, paid by every lane whether taken or not.
  const bool inside = xb * K + 16 * K <= sbytes;
  if (aligned && inside) {
    u32 w[2][4 * K];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = min(y0 + 16 * r, v.dh - 1);
      const uint8_t* srow = v.sp + (u32)(y * ky * v.spitch) + (u32)(xb * K);
#pragma unroll
      for (int i = 0; i < K; ++i) {
        const uint4 q = gload16(srow + 16 * i);
        w[r][4 * i] = q.x; w[r][4 * i + 1] = q.y; w[r][4 * i + 2] = q.z; w[r][4 * i + 3] = q.w;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = y0 + 16 * r;
      if (y >= v.dh)
        break;
      const uint4 o = pb == 1 ? pk_select<K, 1>(w[r]) : pk_select<K, 2>(w[r]);
      uint8_t* drow = v.dp + (u32)(y * v.dpitch) + xb;
      if (xb + 16 <= dbytes) {
        const v4u32 q = {o.x, o.y, o.z, o.w};
        *(VALI_GLOBAL v4u32*)drow = q;
      } else {
        store_bytes16(drow, o, dbytes - xb);
      }
    }
    return;
  }
  for (int r = 0; r < 2; ++r) { // foreign rows and a row's last lane: byte by byte (16 loads in flight, then the stores)
    const int y = y0 + 16 * r;
    if (y >= v.dh)
      break;
    const uint8_t* srow = v.sp + (size_t)(y * ky) * v.spitch;
    uint8_t* drow = v.dp + (size_t)y * v.dpitch;
    const int nb = min(16, dbytes - xb);
    u32 t[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) {
      const int bb = xb + min(b, nb - 1);
      t[b] = gload<uint8_t>(srow + (size_t)(bb / pb) * K * pb + bb % pb);
    }
#pragma unroll
    for (int b = 0; b < 16; ++b)
      if (b < nb)
        gstore<uint8_t>(drow + xb + b, (uint8_t)t[b]);
  }
}

// Planes whose size is unchanged (UDPlanar's luma at unchanged size; any filter at 1:1 is the identity for integer element
// types, see launch_resize): a straight copy -- the point form above moved them at 3.5 TB/s.  A workgroup = 2048 bytes x 16
// rows: every thread has its eight 16-byte loads in flight before the first store.
__global__ void __launch_bounds__(kBlock) k_plane_copy(const ResizeArgs a) {
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  plane_copy_tile(v, v.dw * job.channels * a.rows, tx, ty);     // a.rows: bytes per element here
}

template <typename T, int MAXC> constexpr auto k_resize_point = k_resize<T, MAXC, true>;

// plane jobs per pixel format: which components, their subsampling and channel count
static int resize_jobs(int fmt, ResizeJob* j, int* elem) {
  auto set = [&](int k, int comp, int sx, int sy, int ch) {
    j[k].comp = comp; j[k].sub_x = j[k].ssub_x = sx; j[k].sub_y = j[k].ssub_y = sy; j[k].channels = ch;
  };
  *elem = 1;
  switch (fmt) {
  case VALI_FMT_Y: set(0, 0, 0, 0, 1); return 1;
  case VALI_FMT_GRAY12: *elem = 2; set(0, 0, 0, 0, 1); return 1; // any single u16 plane
  case VALI_FMT_NV12: set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 2); return 2;
  case VALI_FMT_P10: case VALI_FMT_P12: *elem = 2; set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 2); return 2;
  case VALI_FMT_YUV420: set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 1); set(2, 2, 1, 1, 1); return 3;
  case VALI_FMT_YUV420_10BIT: *elem = 2; set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 1); set(2, 2, 1, 1, 1); return 3;
  case VALI_FMT_YUV422: set(0, 0, 0, 0, 1); set(1, 1, 1, 0, 1); set(2, 2, 1, 0, 1); return 3;
  case VALI_FMT_YUV444: case VALI_FMT_RGB_PLANAR: set(0, 0, 0, 0, 1); set(1, 1, 0, 0, 1); set(2, 2, 0, 0, 1); return 3;
  case VALI_FMT_YUV444_10BIT: *elem = 2; set(0, 0, 0, 0, 1); set(1, 1, 0, 0, 1); set(2, 2, 0, 0, 1); return 3;
  case VALI_FMT_RGB_32F_PLANAR: *elem = 4; set(0, 0, 0, 0, 1); set(1, 1, 0, 0, 1); set(2, 2, 0, 0, 1); return 3;
  case VALI_FMT_RGB: case VALI_FMT_BGR: set(0, 0, 0, 0, 3); return 1;
  case VALI_FMT_RGB_32F: *elem = 4; set(0, 0, 0, 0, 3); return 1;
  default: return 0;
  }
}

// UDPlanar (reference: src/TC/src/UDSurface.cpp:33-93, the planar rows of UDSurface::SupportedConversions(),
// :117-133): the planes of a 4:2:0 surface, each resized to the size of the matching plane of a 4:4:4 surface.
// The jobs are the destination format's, with the source planes' own subsampling.
static bool ud_planar_pair(int src_fmt, int dst_fmt) {
  return (src_fmt == VALI_FMT_YUV420 && dst_fmt == VALI_FMT_YUV444) ||
         (src_fmt == VALI_FMT_YUV420_10BIT && dst_fmt == VALI_FMT_YUV444_10BIT);
}
static int resize_jobs_pair(int src_fmt, int dst_fmt, ResizeJob* j, int* elem) {
  const int n = resize_jobs(dst_fmt, j, elem);
  if (src_fmt != dst_fmt) {
    ResizeJob sj[3];
    int selem = 1;
    if (resize_jobs(src_fmt, sj, &selem) != n || selem != *elem)
      return 0;
    for (int k = 0; k < n; ++k) {
      j[k].ssub_x = sj[k].ssub_x;
      j[k].ssub_y = sj[k].ssub_y;
    }
  }
  return n;
}

// `subset`: a.job[0 .. a.njobs) is a subset of the surface's planes chosen by the caller (below), not to be rebuilt
static int launch_resize(ResizeArgs& a, int fmt, int src_w, int src_h, int dst_w, int dst_h, int n, int interp,
                         hipStream_t stream, int src_fmt = -1, bool subset = false) {
  int elem = 1;
  if (subset) {
    ResizeJob all[3];
    resize_jobs_pair(src_fmt < 0 ? fmt : src_fmt, fmt, all, &elem);
  } else {
    a.njobs = resize_jobs_pair(src_fmt < 0 ? fmt : src_fmt, fmt, a.job, &elem);
  }
  if (!a.njobs)
    return fail(VALI_ERR_UNSUPPORTED, "resize: unsupported pixel format %d", fmt);
  u32 total = 0;
  // Integer scale factors on every plane: f = x * scale is an exact integer, every fractional
  // weight is 0 and ALL filters reduce to the same point sample src[k y][k x] (bit for bit:
  // fma(0, d, t) == t, {0,0,1,0,0,0} . taps and {-0,1,0,-0} . taps == the centre tap).  Lanczos
  // and cubic then run on the bilinear kernel, which does not even fetch the zero-weight rows
  // (5.7 -> 1.3 us at the reference's own 2160p -> 720p case), and integer element types take its
  // POINT form: no coordinate arithmetic, one source row, no float math.
  bool integer_scale = true;
  for (int k = 0; k < a.njobs; ++k) {
    const int dw = dst_w >> a.job[k].sub_x, dh = dst_h >> a.job[k].sub_y;
    if (dw <= 0 || dh <= 0)
      return fail(VALI_ERR_INVALID_ARG, "resize: destination too small for its chroma planes");
    const int sw = src_w >> a.job[k].ssub_x, sh = src_h >> a.job[k].ssub_y;
    if (sw <= 0 || sh <= 0)   // a 1-pixel-wide 4:2:0 source has no chroma column to sample
      return fail(VALI_ERR_INVALID_ARG, "resize: source too small for its chroma planes");
    integer_scale = integer_scale && sw % dw == 0 && sh % dh == 0 && sw < (1 << 23) &&
                    sh < (1 << 23);
    a.job[k].first_tile = total;
    a.job[k].tiles_x = (u32)(dw + 255) / 256;
    total += a.job[k].tiles_x * (u32)((dh + kRsTileH - 1) / kRsTileH);
  }
  // Rows per wave of k_resize: 8, or 4 / 2 while 8-row waves would leave SIMDs without a wave (a wave walks its rows one
  // memory round trip after the other: ONE NV12 1080p -> 960x548 frame took 5.9 us through 8-row waves).  The taps and
  // point-by-factor kernels have their own tiling and ignore it.
  a.rows = kRsRowsPerWave;
  {
    const int forced = tuning(VALI_TUNE_ROWS_PER_WAVE);
    auto tiles = [&](int rows) {
      u32 t = 0;
      for (int k = 0; k < a.njobs; ++k) {
        const int dh = dst_h >> a.job[k].sub_y;
        t += a.job[k].tiles_x * (u32)((dh + kWavesPerBlock * rows - 1) / (kWavesPerBlock * rows));
      }
      return t;
    };
    if (forced == 2 || forced == 4 || forced == 8)
      a.rows = forced;
    else
      while (a.rows > 2 && (unsigned long long)tiles(a.rows) * (unsigned)n * kWavesPerBlock < 2048ull)
        a.rows /= 2;
    if (a.rows != kRsRowsPerWave) {
      total = 0;
      for (int k = 0; k < a.njobs; ++k) {
        const int dh = dst_h >> a.job[k].sub_y;
        a.job[k].first_tile = total;
        total += a.job[k].tiles_x * (u32)((dh + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows));
      }
    }
  }
  a.map = make_tile_map_linear(total, (u32)n);
  const dim3 grid = tile_grid(a.map), block(kBlock);
  int maxc = 1;
  for (int k = 0; k < a.njobs; ++k)
    maxc = a.job[k].channels > maxc ? a.job[k].channels : maxc;
#define VALI_RS_LAUNCH(KERNEL, T)                                                          \
  do {                                                                                     \
    if (maxc == 1) hipLaunchKernelGGL((KERNEL<T, 1>), grid, block, 0, stream, a);           \
    else if (maxc == 2) hipLaunchKernelGGL((KERNEL<T, 2>), grid, block, 0, stream, a);      \
    else hipLaunchKernelGGL((KERNEL<T, 3>), grid, block, 0, stream, a);                     \
  } while (0)
  const bool filtered = interp != VALI_INTERP_LINEAR && !(integer_scale && elem != 4);
  const bool gather_only = tuning(VALI_TUNE_RESIZE_FORCE_GATHER) == 1;
  a.force_gather = gather_only ? 1 : 0;
  const bool point_on = tuning(VALI_TUNE_RESIZE_POINT) != 0;
  // small integer horizontal factor on 8-bit planes of 1 / 2 channels: the LDS-free strided-window form
  int kx = 0;
  if (integer_scale && elem == 1 && maxc <= 2 && point_on && tuning(VALI_TUNE_RESIZE_POINT) != 2) {
    kx = (src_w >> a.job[0].ssub_x) / (dst_w >> a.job[0].sub_x);
    for (int k = 1; k < a.njobs; ++k)
      if ((src_w >> a.job[k].ssub_x) / (dst_w >> a.job[k].sub_x) != kx)
        kx = 0;
    if (kx < 2 || kx > 3) // K = 4 measured slower than the staged point form (0.80 vs 0.68 us at 2160p -> 960x540)
      kx = 0;
  }
  auto launch_copy = [&](ResizeArgs& c) { // planes of unchanged size
    u32 t = 0;
    for (int k = 0; k < c.njobs; ++k) {
      const int row_bytes = (dst_w >> c.job[k].sub_x) * c.job[k].channels * elem, dh = dst_h >> c.job[k].sub_y;
      c.job[k].first_tile = t;
      c.job[k].tiles_x = (u32)(row_bytes + kCopyW - 1) / kCopyW;
      t += c.job[k].tiles_x * (u32)((dh + kCopyH - 1) / kCopyH);
    }
    c.map = make_tile_map_linear(t, (u32)n);
    c.rows = elem;
    hipLaunchKernelGGL(k_plane_copy, tile_grid(c.map), block, 0, stream, c);
    VALI_LAUNCH_CHECK();
    return (int)VALI_OK;
  };
  bool same_size = integer_scale && elem != 4 && point_on && !gather_only;
  for (int k = 0; k < a.njobs; ++k)
    same_size = same_size && (src_w >> a.job[k].ssub_x) == (dst_w >> a.job[k].sub_x) && (src_h >> a.job[k].ssub_y) == (dst_h >> a.job[k].sub_y);
  if (same_size)
    return launch_copy(a);
  if (kx) {
    u32 t = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dwb = (dst_w >> a.job[k].sub_x) * a.job[k].channels, dh = dst_h >> a.job[k].sub_y;
      a.job[k].first_tile = t;
      a.job[k].tiles_x = (u32)(dwb + 255) / 256;
      t += a.job[k].tiles_x * (u32)((dh + kPkRows - 1) / kPkRows);
    }
    a.map = make_tile_map_linear(t, (u32)n);
    const dim3 g = tile_grid(a.map);
    if (kx == 2) hipLaunchKernelGGL(k_resize_pointk<2>, g, block, 0, stream, a);
    else hipLaunchKernelGGL(k_resize_pointk<3>, g, block, 0, stream, a);
  } else if (integer_scale && elem != 4 && point_on) { // every filter is the point sample (see resize_tile)
    if (elem == 1) VALI_RS_LAUNCH(k_resize_point, uint8_t);
    else VALI_RS_LAUNCH(k_resize_point, uint16_t);
  } else if (filtered) { // Lanczos-3 / bicubic at a non-integer ratio: resize_taps.hip
    // planes that shrink vertically are filtered columns first, the others rows first (the specification's rule:
    // oracle/vali_oracle.c resize_plane_taps); a surface whose planes differ (YUV420 -> YUV444 through UDPlanar: luma
    // shrinks, chroma grows) takes one launch per order
    const int taps = interp == VALI_INTERP_LANCZOS ? 6 : 4;
    // ... and, plane by plane (UDPlanar at unchanged size: luma 1:1, chroma 1:2): planes at an integer ratio are the point
    // sample whatever the others need; one-channel planes exactly doubled both ways have their own kernel (resize_up2.hip)
    ResizeArgs cpy = a, pts = a, up2 = a, cols = a, rows = a, grow = a, x23 = a, rgbg = a;
    cpy.njobs = pts.njobs = up2.njobs = cols.njobs = rows.njobs = grow.njobs = x23.njobs = rgbg.njobs = 0;
    const bool rows_reg = tuning(VALI_TUNE_RESIZE_ROWS) != 0; // 0: every growing plane through k_resize_taps (round 2's kernel)
    const bool special = point_on && !gather_only && elem != 4;
    for (int k = 0; k < a.njobs; ++k) {
      const int sw = src_w >> a.job[k].ssub_x, sh = src_h >> a.job[k].ssub_y;
      const int dw = dst_w >> a.job[k].sub_x, dh = dst_h >> a.job[k].sub_y;
      const bool integer = special && sw % dw == 0 && sh % dh == 0 && sw < (1 << 23) && sh < (1 << 23);
      const bool doubled = special && a.job[k].channels <= 2 && dw == 2 * sw && dh == 2 * sh && sw % (4 / a.job[k].channels) == 0;
      const bool fits = rows_reg && sh < dh && resize_rows_fits(a.job[k], elem, src_w, dst_w, taps);
      const bool is23 = rows_reg && tuning(VALI_TUNE_RESIZE_ROWS) != 2 && special && resize_x23_fits(a.job[k], elem, src_w, src_h, dst_w, dst_h);
      const bool isrgb = rows_reg && special && sh < dh && resize_rows_rgb_fits(a.job[k], elem, taps, sw, dw);
      ResizeArgs& t = integer && sw == dw && sh == dh ? cpy : integer ? pts : doubled ? up2 : sh >= dh ? cols : is23 ? x23 : isrgb ? rgbg
                      : fits ? grow : rows;
      t.job[t.njobs++] = a.job[k];
    }
    // UDPlanar at unchanged size (the reference's everyday planar UD, UDSurface.cpp:84-93): the luma copy rides in the launch
    // that doubles the chroma planes -- one launch instead of two, and the copy's memory time hides behind the filter's
    // instruction time on the same CUs
    if (cpy.njobs && up2.njobs && cpy.njobs + up2.njobs <= 3 && tuning(VALI_TUNE_RESIZE_POINT) == 1) {
      for (int k = 0; k < cpy.njobs; ++k) {
        up2.job[up2.njobs] = cpy.job[k];
        up2.job[up2.njobs++].kind = 1;
      }
      cpy.njobs = 0;
    }
    int rc = VALI_OK;
    if (cpy.njobs)
      rc = launch_copy(cpy);
    if (rc == VALI_OK && pts.njobs) // (never all of them: integer_scale would have been true)
      rc = launch_resize(pts, fmt, src_w, src_h, dst_w, dst_h, n, interp, stream, src_fmt, true);
    if (rc == VALI_OK && up2.njobs)
      rc = launch_resize_up2(up2, elem, taps, src_w, src_h, n, stream);
    if (rc == VALI_OK && cols.njobs)
      rc = launch_resize_cols(cols, elem, taps, src_w, src_h, dst_w, dst_h, n, stream);
    if (rc == VALI_OK && x23.njobs)
      rc = launch_resize_x23(x23, elem, taps, src_w, src_h, dst_w, dst_h, n, stream);
    if (rc == VALI_OK && rgbg.njobs)
      rc = launch_resize_rows_rgb(rgbg, src_h, dst_w, dst_h, n, stream);
    if (rc == VALI_OK && grow.njobs)
      rc = launch_resize_rows(grow, elem, taps, src_w, src_h, dst_w, dst_h, n, stream);
    if (rc == VALI_OK && rows.njobs)
      rc = launch_resize_taps(rows, elem, taps, src_w, src_h, dst_w, dst_h, n, stream);
    return rc;
  } else {
    if (elem == 1) VALI_RS_LAUNCH(k_resize, uint8_t);
    else if (elem == 2) VALI_RS_LAUNCH(k_resize, uint16_t);
    else VALI_RS_LAUNCH(k_resize, float);
  }
#undef VALI_RS_LAUNCH
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali

using namespace vali;

static int resize_one(const vali_surface* src, const vali_surface* dst, int interpolation, vali_stream_t stream,
                      bool ud_planar, const char* entry) {
  VALI_REQUIRE(src && dst, "null argument");
  if (ud_planar) {
    if (!ud_planar_pair(src->format, dst->format))
      return fail(VALI_ERR_UNSUPPORTED, "ud_planar: unsupported pair %d -> %d", src->format, dst->format);
    VALI_REQUIRE(((src->width | src->height) & 1) == 0, "4:2:0 surfaces need even width and height");
  } else {
    VALI_REQUIRE(src->format == dst->format, "src/dst format mismatch");
  }
  VALI_REQUIRE(src->width > 0 && src->height > 0 && dst->width > 0 && dst->height > 0, "empty surface");
  VALI_REQUIRE(subsampled_sizes_ok(src->format, src->width, src->height) && subsampled_sizes_ok(dst->format, dst->width, dst->height),
               "surfaces with subsampled chroma need a width / height that is a multiple of the subsampling");
  VALI_REQUIRE(src->plane[0] && dst->plane[0], "null plane");
  VALI_REQUIRE(planes_fit_32bit(*src) && planes_fit_32bit(*dst), "plane of 4 GiB or more");
  if (interpolation != VALI_INTERP_LINEAR && interpolation != VALI_INTERP_LANCZOS &&
      interpolation != VALI_INTERP_CUBIC)
    return fail(VALI_ERR_UNSUPPORTED, "resize: interpolation %d not implemented", interpolation);
  ResizeArgs a = {};
  a.sw = src->width; a.sh = src->height; a.dw = dst->width; a.dh = dst->height;
  int elem = 1;
  const int nj = resize_jobs_pair(src->format, dst->format, a.job, &elem);
  for (int k = 0; k < nj; ++k) { // resolve the planes on the host (see PlaneJob)
    const int c = a.job[k].comp;
    VALI_REQUIRE(src->plane[c] && dst->plane[c], "null plane");
    a.job[k].sp = (const uint8_t*)src->plane[c];
    a.job[k].dp = (uint8_t*)dst->plane[c];
    a.job[k].spitch = src->pitch[c];
    a.job[k].dpitch = dst->pitch[c];
  }
  hipStream_t s = as_stream(stream);
  VALI_ENTRY_NAMED(s, entry);
  return launch_resize(a, dst->format, src->width, src->height, dst->width, dst->height, 1, interpolation, s, src->format);
}

static int resize_many(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_format, int format,
                       int src_width, int src_height, int dst_width, int dst_height, int interpolation,
                       vali_stream_t stream, const char* entry) {
  VALI_REQUIRE(d_src && d_dst, "null argument");
  VALI_REQUIRE(src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0, "empty geometry");
  VALI_REQUIRE(subsampled_sizes_ok(src_format, src_width, src_height) && subsampled_sizes_ok(format, dst_width, dst_height),
               "surfaces with subsampled chroma need a width / height that is a multiple of the subsampling");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (interpolation != VALI_INTERP_LINEAR && interpolation != VALI_INTERP_LANCZOS &&
      interpolation != VALI_INTERP_CUBIC)
    return fail(VALI_ERR_UNSUPPORTED, "resize: interpolation %d not implemented", interpolation);
  if (n == 0)
    return VALI_OK;
  ResizeArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY_NAMED(s, entry);
  return launch_resize(a, format, src_width, src_height, dst_width, dst_height, n, interpolation, s, src_format);
}

extern "C" {

int vali_resize(const vali_surface* src, const vali_surface* dst, int interpolation, vali_stream_t stream) {
  return resize_one(src, dst, interpolation, stream, false, __func__);
}

int vali_resize_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int format,
                      int src_width, int src_height, int dst_width, int dst_height, int interpolation,
                      vali_stream_t stream) {
  return resize_many(d_src, d_dst, n, format, format, src_width, src_height, dst_width, dst_height, interpolation, stream,
                     __func__);
}

int vali_ud_planar(const vali_surface* src, const vali_surface* dst, int interpolation, vali_stream_t stream) {
  return resize_one(src, dst, interpolation, stream, true, __func__);
}

int vali_ud_planar_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_format, int dst_format,
                         int src_width, int src_height, int dst_width, int dst_height, int interpolation,
                         vali_stream_t stream) {
  if (!ud_planar_pair(src_format, dst_format))
    return fail(VALI_ERR_UNSUPPORTED, "ud_planar: unsupported pair %d -> %d", src_format, dst_format);
  VALI_REQUIRE(((src_width | src_height) & 1) == 0, "4:2:0 surfaces need even width and height");
  return resize_many(d_src, d_dst, n, src_format, dst_format, src_width, src_height, dst_width, dst_height, interpolation,
                     stream, __func__);
}

} // extern "C"
