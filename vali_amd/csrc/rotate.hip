// Surface rotation: replaces nppiRotate_{8u,16u,32f}_{C1,C3}R_Ctx with NPPI_INTER_LINEAR as the
// reference calls it from Rot_8U_C1 ... Rot_32F_C3, once per plane
// (reference: src/TC/src/RotateSurface.cpp:22-125; per-plane / packed drivers :132-159).
// Here all planes of a surface -- and all surfaces of a batch -- are rotated by ONE launch.
//
// NPP's model (NPP documentation of nppiRotate): the source is rotated by `angle` degrees
// about its origin and then shifted,
//     x' =  x cos(a) + y sin(a) + shift_x        y' = -x sin(a) + y cos(a) + shift_y
// every destination pixel is produced by inverse mapping + bilinear interpolation and
// destination pixels whose source point lies outside the source ROI are left untouched.
// With the shifts PySurfaceRotator derives for 90/180/270 (PySurfaceRotator.cpp:47-73) the
// mapping is an exact pixel permutation, which is what the reference's
// rotation etalons pin.
//
// Kernels:
//  * k_rotate_affine_lds<T, TW, PASSES> (round 6): any angle.  cos/sin arrive from the host as floats (snapped to
//    exactly 0/+-1 for multiples of 90 degrees), so the device does only fma/mul/add and is
//    bit-exact with the oracle.  A workgroup owns TW x TH dst pixels (64 x 64 for full launches), stages the bounding box of
//    their source coordinates in LDS with coalesced row loads and interpolates from there; one lane = 4 adjacent dst pixels
//    per pass (wide stores).  RGB 1080p by 30 degrees: 6.3 -> 2.2 us (profiles/r06_rotate.md).
//  * k_rotate_affine<T> (round 2): the same arithmetic with per-pixel gathers from global memory, 32 x 32 dst pixels per
//    workgroup: planes narrower than a staged row, and VALI_TUNE_ROTATE_AFFINE = 1.
//  * k_rotate_tile<P,Q>: the canonical 90 / 270 degree permutations as an LDS-tiled transpose:
//    a 64x64-pixel tile is read with coalesced row segments, written with coalesced row
//    segments of the transposed tile; LDS row stride 64*P+4 bytes keeps the column walks
//    at most 2-way bank conflicted.  P = bytes per pixel (1,2,4,6,12); 3-byte pixels
//    (rotate_tile_rgb8) keep ONE DWORD PER PIXEL in LDS: aligned ds_read_b32 column walks.
//  * k_rotate_half<P>: the canonical 180 degree turn is a reversal: streaming, 4 pixels x 4 rows
//    per lane, no LDS.
// Arithmetic of the affine path (the specification, oracle: vali_oracle_rotate_plane):
//   dx = x' - shift_x ; dy = y' - shift_y                      (float)
//   xs = fma(-s, dy, c*dx) ; ys = fma(c, dy, s*dx)
//   skip unless 0 <= xs <= W-1 and 0 <= ys <= H-1
//   i = floor(xs), a = xs - i ; j = floor(ys), b = ys - j ; i1 = min(i+1, W-1), j1 likewise
//   t0 = fma(a, T[j][i1]-T[j][i], T[j][i]) ; t1 likewise on row j1 ; v = fma(b, t1-t0, t0)
//   u8/u16: round-half-even + saturate ; f32: v.
#include "common.hpp"
#include "dev_util.hpp"

#include <math.h>
#include <stdlib.h>

namespace vali {

typedef PlaneJob RotJob; // dev_util.hpp

struct RotArgs {
  const vali_surface* d_src; // batch: device descriptor arrays
  const vali_surface* d_dst;
  int sw, sh, dw, dh;        // single frame: surface sizes (planes are resolved into the jobs)
  RotJob job[3];
  int njobs;
  float c, s;
  TileMap map;
  int tile_order; // quarter-turn tiles: 0 = walk the source tiles row by row (default), 1 = column by column (A/B)
};

__device__ __forceinline__ bool rot_tile(const RotArgs& a, RotJob& job, u32& tx, u32& ty, u32& frame) {
  return plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame);
}

constexpr int kAffineTile = 32;
constexpr long long kAffBigLaunchTiles = 2048; // 64 x 64 tiles of a launch from which the LDS-staged rotation uses them (8 per CU)
template <typename T, int C>
__device__ __forceinline__ void affine_tile(const RotArgs& a, const RotJob& job, const uint8_t* sp,
                                            int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                            int dw, int dh, u32 tile_x, u32 tile_y) {
  // 32 x 32 destination pixels per workgroup (8 lanes x 4 pixels per row): whatever the angle,
  // the source footprint of the workgroup is a compact <= 49 x 49 pixel square that stays in the
  // CU's L1, where a 256 x 4 strip sweeps up to 130 source rows (1080p RGB at 30 deg: 8.7 -> 6.0 us)
  const int x0 = (tile_x * 8 + (threadIdx.x & 7)) * 4;
  const int y = tile_y * kAffineTile + (threadIdx.x >> 3);
  if (x0 >= dw || y >= dh)
    return;
  constexpr int PB = C * (int)sizeof(T);
  const float dy = (float)y - job.shift_y;
  const float wmax = (float)(sw - 1), hmax = (float)(sh - 1);
  // phase 1: coordinates + all texel loads (clamped to a valid address for missed pixels)
  float fa[4], fb[4], t[4][4][C];
  u32 mask = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float dx = (float)(x0 + p) - job.shift_x;
    const float xs = __builtin_fmaf(-a.s, dy, a.c * dx);
    const float ys = __builtin_fmaf(a.c, dy, a.s * dx);
    const bool hit = (x0 + p < dw) && xs >= 0.0f && xs <= wmax && ys >= 0.0f && ys <= hmax;
    mask |= hit ? (1u << p) : 0u;
    const float xc = hit ? xs : 0.0f, yc = hit ? ys : 0.0f;
    const float fi = __builtin_floorf(xc), fj = __builtin_floorf(yc);
    fa[p] = xc - fi;
    fb[p] = yc - fj;
    const int i = (int)fi, j = (int)fj;
    const int i1 = min(i + 1, sw - 1), j1 = min(j + 1, sh - 1);
    const uint8_t* r0 = sp + (u32)(j * spitch); // a plane is < 4 GiB (vali_hip.h): 32-bit row offsets
    const uint8_t* r1 = sp + (u32)(j1 * spitch);
    if (PairLoad<T, C>::kMerged && sw >= 3) { // (wave-uniform)
      if constexpr (PairLoad<T, C>::kMerged) {
        load_tap_pair<T, C>(r0, i, sw, t[p][0], t[p][1]);
        load_tap_pair<T, C>(r1, i, sw, t[p][2], t[p][3]);
      }
    } else {
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        t[p][0][ch] = (float)gload<T>(r0 + (size_t)i * PB + ch * sizeof(T));
        t[p][1][ch] = (float)gload<T>(r0 + (size_t)i1 * PB + ch * sizeof(T));
        t[p][2][ch] = (float)gload<T>(r1 + (size_t)i * PB + ch * sizeof(T));
        t[p][3][ch] = (float)gload<T>(r1 + (size_t)i1 * PB + ch * sizeof(T));
      }
    }
  }
  if (!mask)
    return;
  // phase 2: interpolate, finish, one wide store when all four pixels exist
  float res[4][C];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const float t0 = __builtin_fmaf(fa[p], t[p][1][ch] - t[p][0][ch], t[p][0][ch]);
      const float t1 = __builtin_fmaf(fa[p], t[p][3][ch] - t[p][2][ch], t[p][2][ch]);
      res[p][ch] = __builtin_fmaf(fb[p], t1 - t0, t0);
    }
  store_px4<T, C>(dp + (u32)(y * dpitch) + (size_t)x0 * PB, res, mask);
}

template <typename T>
__global__ void __launch_bounds__(kBlock) k_rotate_affine(const RotArgs a) {
  RotJob job;
  u32 tx, ty, frame;
  if (!rot_tile(a, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  if (job.channels == 1)
    affine_tile<T, 1>(a, job, v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty);
  else
    affine_tile<T, 3>(a, job, v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty);
}

// ---- any angle, the tile's source footprint staged in LDS (round 6) ---------------------------
// A workgroup owns TW x TH destination pixels.  The inverse map is affine and each of its two float forms (above) is monotone
// in x' and in y' separately (a product and an fma round monotonically), so the source coordinates of the whole tile lie between
// those of its four corners EVALUATED WITH THE SAME EXPRESSIONS: the box [floor(min xs), floor(max xs) + 1] x [floor(min ys),
// floor(max ys) + 1], clipped to the plane, holds every texel the tile reads.  A pure rotation keeps that box within
// (TW - 1)|c| + (TH - 1)|s| + 3 columns (rows likewise): the host sizes the dynamic LDS for the angle.
//   phase 1  the box is loaded with coalesced row pieces (16 bytes of a row per lane, any alignment; packed RGB: 12 bytes = 4
//            pixels) -- every load of a thread is in flight before the first LDS write -- and stored with an odd dword row stride;
//            packed RGB is unpacked to ONE DWORD PER PIXEL so that the two horizontal taps of a row are one ds_read_b64;
//            a box that would cross the right edge of the plane slides left to end with the row; rows past the plane's last
//            repeat it (the row below the last one is only ever weighted with b = 0)
//   phase 2  a lane produces 4 adjacent dst pixels per pass from LDS with the arithmetic of the specification (the same fmaf
//            forms as k_rotate_affine); tiles whose box lies wholly inside the plane skip the per-pixel range tests.
// The column right of the plane's last one (read when xs == W - 1 exactly, with a = 0) is whatever the LDS holds: for integer
// pixels any finite value times 0 leaves t exactly as the specification's clamp does; float pixels select the clamped texel.
template <typename T, int C> struct AffFmt {
  static constexpr int PB = C * (int)sizeof(T);
  static constexpr bool kDwordPx = sizeof(T) == 1 && C == 3; // packed RGB: one dword per pixel in LDS
  static constexpr int LP = kDwordPx ? 4 : PB;               // LDS bytes per pixel
  static constexpr int GCH = kDwordPx ? 12 : 16;             // global bytes of one staging piece (its LDS image is 16 bytes)
};

struct AffGeom { // host-computed, the same for every tile of the launch
  int box_w, box_h; // most columns / rows a tile's box can have
  int nch_max;      // staging pieces per box row at most
  int stride;       // LDS row stride in bytes (odd number of dwords)
};

typedef __attribute__((address_space(3))) uint8_t lds_u8;
__device__ __forceinline__ int cvt_i32_sat(float v) { // v_cvt_i32_f32: saturates, NaN -> 0 (a C cast of an out-of-range float is undefined)
  int r;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ int med3i(int v, int lo, int hi) { return min(max(v, lo), hi); }

// the two horizontal taps of one box row from LDS.  Every access is NATURALLY ALIGNED: a DS read whose address is not a multiple
// of its size returns the right bytes but takes 256 cycles per wave instruction instead of 16 (one lane at a time;
// tools/exp/lds_unaligned.hip, profiles/r06_rotate.md) -- the first form of this kernel read a packed-RGB pair with one
// ds_read_b64 at 4-byte alignment and spent its whole time there.
template <typename T, int C>
__device__ __forceinline__ void lds_tap_pair(const lds_u8* p, bool last_col, float (&t0)[C], float (&t1)[C]) {
  if constexpr (sizeof(T) == 1 && C == 3) {
    const __attribute__((address_space(3))) u32* q = (const __attribute__((address_space(3))) u32*)p; // ds_read2_b32 offset1:1
    const u32 lo = q[0], hi = q[1];
    t0[0] = ubyte_f32<0>(lo); t0[1] = ubyte_f32<1>(lo); t0[2] = ubyte_f32<2>(lo);
    t1[0] = ubyte_f32<0>(hi); t1[1] = ubyte_f32<1>(hi); t1[2] = ubyte_f32<2>(hi);
  } else { // element reads (float pixels select the clamped texel: 0 x inf would not be 0)
    const __attribute__((address_space(3))) T* f = (const __attribute__((address_space(3))) T*)p;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      t0[ch] = (float)f[ch];
      t1[ch] = sizeof(T) == 4 && last_col ? t0[ch] : (float)f[C + ch];
    }
  }
}

template <typename T, int C, int TW, int PASSES, bool INTERIOR>
__device__ __forceinline__ void affine_lds_rows(const RotArgs& a, const RotJob& job, const PlaneView& v, const lds_u8* lds, int stride,
                                                int base, int X0, int Y0, int x_lo, int x_hi, int y_lo, int y_hi) {
  constexpr int PB = AffFmt<T, C>::PB, LP = AffFmt<T, C>::LP, kLanesPerRow = TW / 4, kRowsPerPass = kBlock / kLanesPerRow;
  const int x0 = X0 + (int)(threadIdx.x % kLanesPerRow) * 4;
  const float wmax = (float)(v.sw - 1), hmax = (float)(v.sh - 1);
  float dxs[4], cdx[4], sdx[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    dxs[p] = (float)(x0 + p) - job.shift_x;
    cdx[p] = a.c * dxs[p];
    sdx[p] = a.s * dxs[p];
  }
#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    const int y = Y0 + pass * kRowsPerPass + (int)(threadIdx.x / kLanesPerRow);
    if (!INTERIOR && (x0 >= v.dw || y >= v.dh))
      continue;
    const float dy = (float)y - job.shift_y;
    float fa[4], fb[4], t[4][4][C];
    u32 mask = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float xs = __builtin_fmaf(-a.s, dy, cdx[p]);
      const float ys = __builtin_fmaf(a.c, dy, sdx[p]);
      fa[p] = __builtin_amdgcn_fractf(xs); // = xs - floor(xs) exactly for xs >= 0 (the only values that are used)
      fb[p] = __builtin_amdgcn_fractf(ys);
      int i = cvt_i32_sat(xs), j = cvt_i32_sat(ys); // truncation = floor for xs >= 0
      if constexpr (!INTERIOR) {
        const bool hit = (x0 + p < v.dw) && xs >= 0.0f && xs <= wmax && ys >= 0.0f && ys <= hmax;
        mask |= hit ? (1u << p) : 0u;
        i = med3i(i, x_lo, x_hi); // missed pixels read some texel of the box and are not stored
        j = med3i(j, y_lo, y_hi);
      } else {
        mask = 0xfu;
      }
      const lds_u8* r0 = lds + (__mul24(j, stride) + i * LP + base); // (24-bit multiply: full rate; v_mul_lo_u32 is quarter rate)
      lds_tap_pair<T, C>(r0, i >= v.sw - 1, t[p][0], t[p][1]);
      lds_tap_pair<T, C>(r0 + stride, i >= v.sw - 1, t[p][2], t[p][3]);
    }
    if (!INTERIOR && !mask)
      continue;
    float res[4][C];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const float t0 = __builtin_fmaf(fa[p], t[p][1][ch] - t[p][0][ch], t[p][0][ch]);
        const float t1 = __builtin_fmaf(fa[p], t[p][3][ch] - t[p][2][ch], t[p][2][ch]);
        res[p][ch] = __builtin_fmaf(fb[p], t1 - t0, t0);
      }
    store_px4<T, C>(v.dp + (u32)(y * v.dpitch) + (size_t)x0 * PB, res, mask);
  }
}

// most staging pieces of a tile (any angle) over kBlock threads: box area at 45 degrees
constexpr int aff_isqrt_up(int v) { int r = 0; while (r * r < v) ++r; return r; }
template <typename T, int C, int TW, int TH> constexpr int aff_max_pieces() {
  const int diag = aff_isqrt_up(2 * (TW + TH) * (TW + TH) / 4) + 5;           // (TW + TH) / sqrt(2), rounded up, + slack
  const int per_row = (diag * AffFmt<T, C>::PB + AffFmt<T, C>::GCH - 1) / AffFmt<T, C>::GCH + 1;
  return (per_row * (diag + 1) + kBlock - 1) / kBlock;
}

template <typename T, int C, int TW, int PASSES>
__device__ __forceinline__ void affine_lds_tile(const RotArgs& a, const AffGeom& g, const RotJob& job, const PlaneView& v, u32 tile_x,
                                                u32 tile_y, lds_u8* lds) {
  typedef AffFmt<T, C> F;
  constexpr int kLanesPerRow = TW / 4, TH = kBlock / kLanesPerRow * PASSES;
  const int X0 = tile_x * TW, Y0 = tile_y * TH;
  const int X1 = min(X0 + TW, v.dw) - 1, Y1 = min(Y0 + TH, v.dh) - 1;
  // the box: lane k (k = 0..3 of every wave) evaluates corner k; broadcast before anything diverges
  int min_i, max_i, min_j, max_j;
  {
    const int lane = threadIdx.x & 63;
    const float dx = (float)((lane & 1) ? X1 : X0) - job.shift_x, dy = (float)((lane & 2) ? Y1 : Y0) - job.shift_y;
    const float xs = __builtin_fmaf(-a.s, dy, a.c * dx), ys = __builtin_fmaf(a.c, dy, a.s * dx);
    const int ic = med3i(cvt_i32_sat(__builtin_floorf(xs)), -4, 1 << 24), jc = med3i(cvt_i32_sat(__builtin_floorf(ys)), -4, 1 << 24);
    const int i0 = __builtin_amdgcn_readlane(ic, 0), i1 = __builtin_amdgcn_readlane(ic, 1), i2 = __builtin_amdgcn_readlane(ic, 2),
              i3 = __builtin_amdgcn_readlane(ic, 3);
    const int j0 = __builtin_amdgcn_readlane(jc, 0), j1 = __builtin_amdgcn_readlane(jc, 1), j2 = __builtin_amdgcn_readlane(jc, 2),
              j3 = __builtin_amdgcn_readlane(jc, 3);
    min_i = min(min(i0, i1), min(i2, i3)); max_i = max(max(i0, i1), max(i2, i3));
    min_j = min(min(j0, j1), min(j2, j3)); max_j = max(max(j0, j1), max(j2, j3));
  }
  if (max_i < 0 || min_i > v.sw - 1 || max_j < 0 || min_j > v.sh - 1)
    return; // no pixel of the tile samples inside the plane: the destination stays untouched
  const int x_lo = max(min_i, 0), x_hi = min(min(max_i + 1, v.sw - 1), x_lo + g.box_w - 1);
  const int y_lo = max(min_j, 0), y_hi = min(min(max_j + 1, v.sh - 1), y_lo + g.box_h - 1);
  const int bw = x_hi - x_lo + 1, bh = y_hi - y_lo + 1;
  // phase 1
  const int nch = min((bw * F::PB + F::GCH - 1) / F::GCH, g.nch_max);
  int org; // origin of the LDS rows: a pixel (packed RGB) or a byte offset of the row (everything else)
  if constexpr (F::kDwordPx)
    org = min(x_lo, v.sw - 4 * nch);
  else
    org = min(x_lo * F::PB, v.sw * F::PB - 16 * nch);
  {
    const int nrows = bh + 1, total = nrows * nch;
    const u32 inv = (1u << 20) / (u32)nch + 1u; // k / nch = (k * inv) >> 20: exact for k < 2^20 / nch, no overflow for k < 4096
    constexpr int kMax = aff_max_pieces<T, C, TW, TH>();
    // N pieces per thread, straight line: every load is issued (threads past the last piece repeat it: the same line, and their
    // LDS write is skipped) before the first LDS write -- no branch between the loads, so that the compiler keeps them all in flight
    // (behind per-pass uniform branches it merged the whole register array at every join: a page of moves and vmcnt(0) per load).
    // Two sizes, picked per tile: the worst case of the shape (45 degrees) and half of it.
    auto stage = [&](auto n_tag) {
      constexpr int N = decltype(n_tag)::value;
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      typedef typename std::conditional<F::kDwordPx, v3u32, v4u32>::type Piece;
      Piece w[N];
      int at[N];
#pragma unroll
      for (int q = 0; q < N; ++q) {
        const int k = min((int)threadIdx.x + q * kBlock, total - 1);
        const int r = (int)(((u32)k * inv) >> 20), cidx = k - r * nch;
        const uint8_t* row = v.sp + (u32)min(y_lo + r, v.sh - 1) * (u32)v.spitch; // (a plane is < 4 GiB: 32-bit row offsets; a borrowed view's pitch may exceed 24 bits, so no 24-bit multiply here)
        at[q] = r * g.stride + 16 * cidx;
        if constexpr (F::kDwordPx)
          w[q] = gload_u<v3u32>(row + (org + 4 * cidx) * 3);
        else
          w[q] = gload_u<v4u32>(row + org + 16 * cidx);
      }
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if ((int)threadIdx.x + q * kBlock < total) {
          __attribute__((address_space(3))) u32* l = (__attribute__((address_space(3))) u32*)(lds + at[q]);
          if constexpr (F::kDwordPx) {
            l[0] = w[q].x; // (byte 3 of a pixel's dword is the next pixel's first: never read)
            l[1] = __builtin_amdgcn_alignbyte(w[q].y, w[q].x, 3);
            l[2] = __builtin_amdgcn_alignbyte(w[q].z, w[q].y, 2);
            l[3] = w[q].z >> 8;
          } else {
            l[0] = w[q].x; l[1] = w[q].y; l[2] = w[q].z; l[3] = w[q].w;
          }
        }
      }
    };
    // Three-channel planes keep round 6's first form -- a uniform branch per pass around the load and around the write: there the
    // compiler leaves the loads in flight (checked in the ISA) and the absent passes cost nothing; the straight-line form measured
    // 11 - 14 % slower for packed RGB (2.21 -> 2.47 us at 1080p / 30 degrees), 16 - 33 % faster for one-channel planes (Y 1.15 -> 0.97,
    // YUV420 1.70 -> 1.44, 10-bit 5.3 -> 3.6 with the 64 x 64 tile its registers now allow) -- profiles/r06_rotate.md.
    if constexpr (C == 1) {
      constexpr int kHalf = (kMax + 1) / 2;
      if (total <= kHalf * kBlock)
        stage(std::integral_constant<int, kHalf>{});
      else
        stage(std::integral_constant<int, kMax>{});
    } else {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      typedef typename std::conditional<F::kDwordPx, v3u32, v4u32>::type Piece;
      Piece w[kMax];
      int at[kMax];
#pragma unroll
      for (int q = 0; q < kMax; ++q) {
        if (q * kBlock < total) { // (uniform)
          const int k = min((int)threadIdx.x + q * kBlock, total - 1); // lanes past the end repeat the last piece
          const int r = (int)(((u32)k * inv) >> 20), cidx = k - r * nch;
          const uint8_t* row = v.sp + (u32)min(y_lo + r, v.sh - 1) * (u32)v.spitch;
          at[q] = r * g.stride + 16 * cidx;
          if constexpr (F::kDwordPx)
            w[q] = gload_u<v3u32>(row + (org + 4 * cidx) * 3);
          else
            w[q] = gload_u<v4u32>(row + org + 16 * cidx);
        }
      }
#pragma unroll
      for (int q = 0; q < kMax; ++q) {
        if (q * kBlock < total) {
          __attribute__((address_space(3))) u32* l = (__attribute__((address_space(3))) u32*)(lds + at[q]);
          if constexpr (F::kDwordPx) {
            l[0] = w[q].x; // (byte 3 of a pixel's dword is the next pixel's first: never read)
            l[1] = __builtin_amdgcn_alignbyte(w[q].y, w[q].x, 3);
            l[2] = __builtin_amdgcn_alignbyte(w[q].z, w[q].y, 2);
            l[3] = w[q].z >> 8;
          } else {
            l[0] = w[q].x; l[1] = w[q].y; l[2] = w[q].z; l[3] = w[q].w;
          }
        }
      }
    }
  }
  __syncthreads();
  // phase 2
  const int base = -(F::kDwordPx ? org * 4 : org) - y_lo * g.stride;
  const bool interior = min_i >= 0 && max_i + 1 <= v.sw - 1 && min_j >= 0 && max_j + 1 <= v.sh - 1 && X0 + TW <= v.dw && Y0 + TH <= v.dh &&
                        max_i + 1 - min_i < g.box_w && max_j + 1 - min_j < g.box_h;
  if (interior)
    affine_lds_rows<T, C, TW, PASSES, true>(a, job, v, lds, g.stride, base, X0, Y0, x_lo, x_hi, y_lo, y_hi);
  else
    affine_lds_rows<T, C, TW, PASSES, false>(a, job, v, lds, g.stride, base, X0, Y0, x_lo, x_hi, y_lo, y_hi);
}

// CSEL: 0 = the launch may hold one- and three-channel planes (8-bit); 1 = one-channel planes only -- the tall tiles of Y / planar
// YUV launches, whose three-channel instantiation (never reached) would only cost registers
template <typename T, int TW, int PASSES, int CSEL = 0>
__global__ void __launch_bounds__(kBlock) k_rotate_affine_lds(const RotArgs a, const AffGeom g1, const AffGeom g3) {
  extern __shared__ __attribute__((aligned(16))) uint8_t aff_lds_raw[];
  lds_u8* lds = (lds_u8*)aff_lds_raw;
  RotJob job;
  u32 tx, ty, frame;
  if (!rot_tile(a, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  // the pixel formats have u8 x 1, u8 x 3, u16 x 1 and f32 x 3 planes (rotate_jobs)
  if constexpr (sizeof(T) == 1 && CSEL == 1) {
    affine_lds_tile<T, 1, TW, PASSES>(a, g1, job, v, tx, ty, lds);
  } else if constexpr (sizeof(T) == 1) {
    if (job.channels == 1)
      affine_lds_tile<T, 1, TW, PASSES>(a, g1, job, v, tx, ty, lds);
    else
      affine_lds_tile<T, 3, TW, PASSES>(a, g3, job, v, tx, ty, lds);
  } else if constexpr (sizeof(T) == 2) {
    affine_lds_tile<T, 1, TW, PASSES>(a, g1, job, v, tx, ty, lds);
  } else {
    affine_lds_tile<T, 3, TW, PASSES>(a, g3, job, v, tx, ty, lds);
  }
}

// ---- canonical 90 / 270 degree permutation: LDS-tiled transpose -----------------------------
// QUARTER = 1: dst(x', y') = src(W-1-y', x')   [angle 90,  shift_y = W-1]
// QUARTER = 3: dst(x', y') = src(y', H-1-x')   [angle 270, shift_x = H-1]
constexpr int kRotTile = 64;

// 3-byte pixels are held in LDS as ONE DWORD PER PIXEL: the column walk of the transposed
// store then costs one aligned ds_read_b32 per pixel instead of three ds_read_u8 plus the
// shifts to reassemble them (2.4 -> see profiles/r01_secondary.md), and a row stride of 65
// dwords makes the (16 row groups) x (4 columns) of a wave hit 64 different banks.
constexpr int kRotTileHRgb = 64;     // tile rows: 64, or kRotTileHRgbTall for large frames (below)
constexpr int kRotTileHRgbTall = 128; // 384-byte destination segments = 3 whole lines.  Round 2, one box, batch 64: 2160p
                                      // 8.8 -> 8.3 us (10.8 -> 8.4-9.2 on a slower box), 1080p equal, 720p 0.91 -> 1.0,
                                      // 640x360 0.25 -> 0.30 (33 KB of LDS: too few workgroups for small frames) -> used
                                      // from kRotTallPixels source pixels per plane on; 64x256, 128x128, 128x64, 32x128
                                      // and 32x256 tiles all measured slower than one of these two
constexpr int kRotTallPixels = 4 << 20;
constexpr int kRotTileWRgb = 64; // tile columns for 3-byte pixels; 128 (source segments = 3 whole 128-byte lines, 33 KB of
                                 // LDS per workgroup) measured slower in round 2: 1080p 3.13 vs 2.71 us, 2160p 11.9 vs 10.2
template <int QUARTER, int TH>
__device__ __forceinline__ void rotate_tile_rgb8(const PlaneView& v, u32 tile_x, u32 tile_y) {
  constexpr int P = 3, TW = kRotTileWRgb, SD = TW + 1; // LDS row stride in dwords; TH = tile rows
  __shared__ u32 lds[TH * SD];
  const int src_w = v.sw, src_h = v.sh, dst_w = v.dw, dst_h = v.dh;
  const uint8_t* src = v.sp;
  uint8_t* dst = v.dp;
  const int src_pitch = v.spitch, dst_pitch = v.dpitch;
  // src tile origin (col, row).  At 270 degrees source row r lands in dst column H - 1 - r: the tiles are anchored at the LAST
  // source row (the ragged tile is the first one: its rows before the plane are loaded from row 0 and never stored), so that a
  // tile's dst segments start at multiples of TH pixels as they do at 90 -- from H - 64 k they straddled the 64-byte sectors
  // (1080p: 2.9 us against 2.1-2.25 at 90 degrees, and non-temporal stores cost 20 % instead of saving 7; round 5)
  const int cx = tile_x * TW, ry = (int)tile_y * TH - (QUARTER == 3 ? (TH - src_h % TH) % TH : 0);
  const int tw = min(TW, src_w - cx), th = min(TH, src_h - ry);
  const int t = threadIdx.x;
  auto src_row = [&](int r) { return src + (size_t)min(max(ry + r, 0), src_h - 1) * src_pitch + (size_t)cx * P; };

  // phase 1: TW/4 lanes x 4 pixels (one 12-byte load) per source row
  // (vector accesses at ANY byte alignment: a 270 degree turn of a frame whose height is not a multiple of 4 starts
  // its destination segments at odd offsets -- through the byte path that measured 10.5 instead of 4.3 us, 2720x1530)
  constexpr bool vec = true;
  {
  constexpr int kLanesPerSrcRow = TW / 4, kSrcRowsPerPass = kBlock / kLanesPerSrcRow;
  const int chunk = t % kLanesPerSrcRow;
  if (vec && tw >= 4) {
    // Every load of the thread is issued before the first is unpacked.  Behind per-pass edge tests the compiler waits for
    // each load before it issues the next -- four memory round trips in a row, which is what bounded the kernel (a workgroup
    // lived ~10 us).  Round 4: the tiles at the frame's bottom / right edge take this path too -- rows past the tile re-read
    // its last row, a lane whose 4 pixels would cross the tile's right edge slides left to end with it (the LDS columns it
    // shares with its neighbour are written twice with the same values); phase 2 never reads rows / columns past the tile.
    // 1080 rows are 16 whole tiles and one of 56 rows: with those 30 of 510 workgroups on the serial path the 1080p frame
    // took 2.68 us against 2.16 for 1920x1024 (profiles/r04_rotate.md).
    typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
    constexpr int kPasses = TH / kSrcRowsPerPass;
    v3u32 w[kPasses];
    const int col = min(chunk * 4, tw - 4);       // first pixel of this lane's group inside the tile
    const int r0 = t / kLanesPerSrcRow;
#pragma unroll
    for (int pass = 0; pass < kPasses; ++pass)
      w[pass] = gload_u<v3u32>(src_row(min(pass * kSrcRowsPerPass + r0, th - 1)) + col * 3);
#pragma unroll
    for (int pass = 0; pass < kPasses; ++pass) {
      u32* l = lds + (pass * kSrcRowsPerPass + r0) * SD + col;
      l[0] = w[pass].x & 0xffffffu;
      l[1] = (w[pass].x >> 24) | ((w[pass].y & 0xffffu) << 8);
      l[2] = (w[pass].y >> 16) | ((w[pass].z & 0xffu) << 16);
      l[3] = w[pass].z >> 8;
    }
  } else
#pragma unroll
  for (int pass = 0; pass < TH / kSrcRowsPerPass; ++pass) {
    const int r = pass * kSrcRowsPerPass + t / kLanesPerSrcRow;
    if (r >= th || chunk * 4 >= tw || ry + r < 0)
      continue;
    const uint8_t* q = src_row(r) + chunk * 12;
    u32 px[4];
    if (vec && chunk * 4 + 4 <= tw) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = gload_u<v3u32>(q);
      px[0] = w.x & 0xffffffu;
      px[1] = (w.x >> 24) | ((w.y & 0xffffu) << 8);
      px[2] = (w.y >> 16) | ((w.z & 0xffu) << 16);
      px[3] = w.z >> 8;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        px[k] = chunk * 4 + k < tw ? ((u32)gload<uint8_t>(q + 3 * k) | ((u32)gload<uint8_t>(q + 3 * k + 1) << 8) |
                                      ((u32)gload<uint8_t>(q + 3 * k + 2) << 16))
                                   : 0u;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      lds[r * SD + chunk * 4 + k] = px[k];
  }
  }
  __syncthreads();

  // phase 2: dst rows.  TH/4 lanes x 4 pixels cover one dst row of the tile.
  constexpr int kLanesPerRow = TH / 4, kRowsPerPass = kBlock / kLanesPerRow;
  const int chunk = t % kLanesPerRow;
#pragma unroll
  for (int pass = 0; pass < TW / kRowsPerPass; ++pass) {
    const int lc = pass * kRowsPerPass + t / kLanesPerRow; // column of the src tile feeding this dst row
    if (lc >= tw)
      continue;
    int dy_, dx0;
    if constexpr (QUARTER == 1) {
      dy_ = src_w - 1 - (cx + lc);
      dx0 = ry + chunk * 4;
    } else {
      dy_ = cx + lc;
      dx0 = src_h - 1 - (ry + chunk * 4 + 3);
    }
    if (dy_ < 0 || dy_ >= dst_h)
      continue;
    u32 px[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int jj = QUARTER == 1 ? chunk * 4 + q : chunk * 4 + 3 - q;
      ok[q] = jj < th && dx0 + q >= 0 && dx0 + q < dst_w;
      px[q] = ok[q] ? lds[jj * SD + lc] : 0u;
    }
    uint8_t* o = dst + (size_t)dy_ * dst_pitch + (ptrdiff_t)dx0 * P;
    if (ok[0] && ok[1] && ok[2] && ok[3]) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = {px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8)};
      // (non-temporal: a tile's 192-byte segments start on 64-byte boundaries at either turn; 1080p at 90 degrees 2.25 -> 2.10 us)
      gstore_u_nt<v3u32>(o, w);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q]) {
          gstore<uint8_t>(o + 3 * q, (uint8_t)px[q]);
          gstore<uint8_t>(o + 3 * q + 1, (uint8_t)(px[q] >> 8));
          gstore<uint8_t>(o + 3 * q + 2, (uint8_t)(px[q] >> 16));
        }
    }
  }
}

__device__ __forceinline__ uint4 rot_load16(const void* p) {
  const v4u32 w = gload_u<v4u32>(p);
  return make_uint4(w.x, w.y, w.z, w.w);
}

template <int P, int QUARTER, int THR = kRotTileHRgb> // THR: tile rows of the 3-byte form
__global__ void __launch_bounds__(kBlock) k_rotate_tile(const RotArgs a) {
  constexpr int S = kRotTile * P + 4; // LDS row stride in bytes (dword aligned, odd dwords)
  __shared__ __attribute__((aligned(16))) uint8_t lds[P == 3 ? 16 : kRotTile * S];
  RotJob job;
  u32 tile_x, tile_y, frame;
  if (!rot_tile(a, job, tile_x, tile_y, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  if (a.tile_order == 1) { // walk the source tiles column by column: consecutive workgroups write neighbouring dst
                           // segments.  Measured both ways round: RGB 2160p 90 deg 10.8 -> 10.15 us and 1080p 270 deg
                           // 2.80 -> 2.72 in one harness, 1080p 90 deg 2.40 -> 2.62 in BASELINE config 4's: not the default
    const u32 th = P == 3 ? THR : kRotTile;
    const u32 tiles_y = ((u32)v.sh + th - 1) / th, local = tile_y * job.tiles_x + tile_x;
    tile_x = local / tiles_y;
    tile_y = local - tile_x * tiles_y;
  }
  if constexpr (P == 3) {
    rotate_tile_rgb8<QUARTER, THR>(v, tile_x, tile_y);
  } else {
  const int src_w = v.sw, src_h = v.sh, dst_w = v.dw, dst_h = v.dh;
  const uint8_t* src = v.sp;
  uint8_t* dst = v.dp;
  const int src_pitch = v.spitch, dst_pitch = v.dpitch;

  // src tile origin (col, row); at 270 degrees anchored at the last source row (rotate_tile_rgb8)
  const int cx = tile_x * kRotTile, ry = (int)tile_y * kRotTile - (QUARTER == 3 ? (kRotTile - src_h % kRotTile) % kRotTile : 0);
  const int tw = min(kRotTile, src_w - cx), th = min(kRotTile, src_h - ry);
  const int t = threadIdx.x;
  auto src_row = [&](int r) { return src + (size_t)min(max(ry + r, 0), src_h - 1) * src_pitch + (size_t)cx * P; };

  // phase 1: coalesced row segments -> LDS (16-byte / 4-byte vectors when aligned)
  const uint8_t* sbase = src_row(0);
  const int row_bytes = tw * P;
  constexpr bool aligned16 = true; // (16-byte loads at any alignment, see rotate_tile_rgb8)
  if (aligned16 && tw == kRotTile) {
    // full-width tiles: the thread's P loads are all in flight before the first LDS write (the run-time loop below waits
    // for each load before it issues the next); rows past a bottom-edge tile re-read its last row (never read back)
    constexpr int V = kRotTile * P / 16; // 16-byte vectors per tile row; kRotTile * V / kBlock = P per thread
    uint4 w[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = t + i * kBlock, r = k / V, c = k - r * V;
      w[i] = rot_load16(src_row(min(r, th - 1)) + c * 16);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const int k = t + i * kBlock, r = k / V, c = k - r * V;
      u32* l = (u32*)(lds + r * S + c * 16);
      l[0] = w[i].x; l[1] = w[i].y; l[2] = w[i].z; l[3] = w[i].w;
    }
  } else if (aligned16 && (row_bytes & 15) == 0) {
    const int v_per_row = row_bytes / 16;
    for (int k = t; k < th * v_per_row; k += kBlock) {
      const int r = k / v_per_row, v = k - r * v_per_row;
      const uint4 w = rot_load16(src_row(r) + v * 16);
      u32* l = (u32*)(lds + r * S + v * 16);
      l[0] = w.x; l[1] = w.y; l[2] = w.z; l[3] = w.w;
    }
  } else if ((((uintptr_t)sbase | (uintptr_t)src_pitch) & 3u) == 0 && (row_bytes & 3) == 0) {
    const int dw_per_row = row_bytes / 4;
    for (int k = t; k < th * dw_per_row; k += kBlock) {
      const int r = k / dw_per_row, v = k - r * dw_per_row;
      *(u32*)(lds + r * S + v * 4) = gload<u32>(src_row(r) + v * 4);
    }
  } else {
    for (int k = t; k < th * row_bytes; k += kBlock) {
      const int r = k / row_bytes, v = k - r * row_bytes;
      lds[r * S + v] = gload<uint8_t>(src_row(r) + v);
    }
  }
  __syncthreads();

  // phase 2: dst rows.  16 lanes x 4 pixels cover one dst row of the tile; 16 rows per pass.
  const int chunk = t & 15;
  for (int pass = 0; pass < kRotTile / 16; ++pass) {
    const int lc = pass * 16 + (t >> 4); // column of the src tile feeding this dst row
    if (lc >= tw)
      continue;
    int dy_, dx0;
    if constexpr (QUARTER == 1) {
      dy_ = src_w - 1 - (cx + lc);
      dx0 = ry + chunk * 4;
    } else {
      dy_ = cx + lc;
      dx0 = src_h - 1 - (ry + chunk * 4 + 3);
    }
    if (dy_ < 0 || dy_ >= dst_h)
      continue;
    bool ok[4];
    uint8_t* o = dst + (size_t)dy_ * dst_pitch + (size_t)dx0 * P;
    if constexpr (P % 4 == 0) {
      // whole-dword pixels (f32 / f32x3): dword LDS reads, no byte reassembly
      constexpr int D = P / 4;
      u32 w[4 * D];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = QUARTER == 1 ? chunk * 4 + q : chunk * 4 + 3 - q;
        ok[q] = jj < th && dx0 + q >= 0 && dx0 + q < dst_w;
#pragma unroll
        for (int k = 0; k < D; ++k)
          w[q * D + k] = ok[q] ? *reinterpret_cast<const u32*>(lds + jj * S + lc * P + 4 * k) : 0u;
      }
      if (ok[0] && ok[1] && ok[2] && ok[3]) {
#pragma unroll
        for (int k = 0; k < 4 * D; ++k)
          gstore_u_nt<u32>(o + 4 * k, w[k]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ok[q])
#pragma unroll
            for (int k = 0; k < D; ++k)
              for (int b = 0; b < 4; ++b) gstore<uint8_t>(o + q * P + 4 * k + b, (uint8_t)(w[q * D + k] >> (8 * b)));
      }
    } else {
      uint8_t px[4 * P];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int jj = QUARTER == 1 ? chunk * 4 + q : chunk * 4 + 3 - q;
        ok[q] = jj < th && dx0 + q >= 0 && dx0 + q < dst_w;
#pragma unroll
        for (int b = 0; b < P; ++b)
          px[q * P + b] = ok[q] ? lds[jj * S + lc * P + b] : (uint8_t)0;
      }
      if (ok[0] && ok[1] && ok[2] && ok[3]) {
        u32 w[P];
        __builtin_memcpy(w, px, 4 * P);
#pragma unroll
        for (int k = 0; k < P; ++k)
          gstore_u_nt<u32>(o + 4 * k, w[k]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (ok[q])
#pragma unroll
            for (int b = 0; b < P; ++b)
              gstore<uint8_t>(o + q * P + b, px[q * P + b]);
      }
    }
  }
  } // P != 3
}

// Canonical half turn: dst(x', y') = src(W-1-x', H-1-y') -- a reversal, no arithmetic, so it is
// a streaming kernel: lane = 4 pixels x 4 rows, the pixel order reversed in registers
// (byte j of the 4-pixel group comes from byte (3 - j / P) P + j % P).  The bilinear kernel it
// replaces for this angle gathers four texels per pixel to reproduce the same bytes
// (9.6 us per 1080p RGB frame).
constexpr int kHalfRowsPerWave = 4, kHalfTileH = kWavesPerBlock * kHalfRowsPerWave;
template <int P>
__global__ void __launch_bounds__(kBlock) k_rotate_half(const RotArgs a) {
  RotJob job;
  u32 tile_x, tile_y, frame;
  if (!rot_tile(a, job, tile_x, tile_y, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  const int W = v.dw, H = v.dh;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x0 = (tile_x * 64 + lane) * 4;
  if (x0 >= W)
    return;
  const int n = min(4, W - x0);
  constexpr int ND = P; // dwords of a 4-pixel group
  // dword accesses at any byte alignment (a width that is not a multiple of 4 mirrors the groups onto odd offsets: the
  // whole frame used to take the byte loop); the byte loop is left to the last, partial group of a row
  const bool vec = n == 4;
  u32 in[kHalfRowsPerWave][ND];
  const int y_first = tile_y * kHalfTileH + wave * kHalfRowsPerWave;
  if (vec) {
#pragma unroll
    for (int r = 0; r < kHalfRowsPerWave; ++r) {
      const int y = min(y_first + r, H - 1);
      const uint8_t* q = v.sp + (size_t)(H - 1 - y) * v.spitch + (size_t)(W - 4 - x0) * P;
#pragma unroll
      for (int k = 0; k < ND; ++k) in[r][k] = gload_u<u32>(q + 4 * k);
    }
#pragma unroll
    for (int r = 0; r < kHalfRowsPerWave; ++r) {
      const int y = y_first + r;
      if (y >= H)
        break;
      u32 out[ND];
#pragma unroll
      for (int j = 0; j < ND; ++j) {
        u32 w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int ob = 4 * j + b, ib = (3 - ob / P) * P + ob % P;
          w |= ((in[r][ib >> 2] >> (8 * (ib & 3))) & 0xffu) << (8 * b);
        }
        out[j] = w;
      }
      uint8_t* o = v.dp + (size_t)y * v.dpitch + (size_t)x0 * P;
#pragma unroll
      for (int k = 0; k < ND; ++k) gstore_u_nt<u32>(o + 4 * k, out[k]);
    }
    return;
  }
  // ragged right edge / unaligned planes: byte copies
  for (int r = 0; r < kHalfRowsPerWave; ++r) {
    const int y = y_first + r;
    if (y >= H)
      break;
    const uint8_t* srow = v.sp + (size_t)(H - 1 - y) * v.spitch;
    uint8_t* drow = v.dp + (size_t)y * v.dpitch;
    for (int k = 0; k < n; ++k)
      for (int b = 0; b < P; ++b)
        gstore<uint8_t>(drow + (size_t)(x0 + k) * P + b, gload<uint8_t>(srow + (size_t)(W - 1 - x0 - k) * P + b));
  }
}

template <int DUMMY = 0> static int launch_half(const RotArgs& a, int pixel_bytes, dim3 grid, hipStream_t s) {
  const dim3 block(kBlock);
  switch (pixel_bytes) {
#define VALI_ROT_CASE(P)                                                                    \
  case P:                                                                                   \
    hipLaunchKernelGGL((k_rotate_half<P>), grid, block, 0, s, a);                            \
    break;
    VALI_ROT_CASE(1)
    VALI_ROT_CASE(2)
    VALI_ROT_CASE(3)
    VALI_ROT_CASE(4)
    VALI_ROT_CASE(6)
    VALI_ROT_CASE(12)
#undef VALI_ROT_CASE
  default:
    return fail(VALI_ERR_UNSUPPORTED, "rotate: unsupported pixel size %d", pixel_bytes);
  }
  return VALI_OK;
}

// plane jobs per pixel format (RotateSurface::Run switch, RotateSurface.cpp:168-208)
static int rotate_jobs(int fmt, RotJob* j, int* elem) {
  auto set = [&](int k, int comp, int sx, int sy, int ch) {
    j[k].comp = comp; j[k].sub_x = j[k].ssub_x = sx; j[k].sub_y = j[k].ssub_y = sy; j[k].channels = ch;
  };
  *elem = 1;
  switch (fmt) {
  case VALI_FMT_Y: set(0, 0, 0, 0, 1); return 1;
  case VALI_FMT_RGB: case VALI_FMT_BGR: set(0, 0, 0, 0, 3); return 1;
  case VALI_FMT_RGB_32F: *elem = 4; set(0, 0, 0, 0, 3); return 1;
  case VALI_FMT_YUV420: set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 1); set(2, 2, 1, 1, 1); return 3;
  case VALI_FMT_YUV420_10BIT: *elem = 2; set(0, 0, 0, 0, 1); set(1, 1, 1, 1, 1); set(2, 2, 1, 1, 1); return 3;
  case VALI_FMT_YUV422: set(0, 0, 0, 0, 1); set(1, 1, 1, 0, 1); set(2, 2, 1, 0, 1); return 3;
  case VALI_FMT_YUV444: set(0, 0, 0, 0, 1); set(1, 1, 0, 0, 1); set(2, 2, 0, 0, 1); return 3;
  case VALI_FMT_YUV444_10BIT: *elem = 2; set(0, 0, 0, 0, 1); set(1, 1, 0, 0, 1); set(2, 2, 0, 0, 1); return 3;
  default: return 0;
  }
}

static void rotate_coeffs(double angle_deg, float* c, float* s) {
  // multiples of 90 degrees: exact 0 / +-1 (cos(pi/2) in floating point is 6e-17, which
  // would push border pixels outside the source)
  const double q = fmod(angle_deg, 360.0);
  const double n = q < 0 ? q + 360.0 : q;
  if (n == 0.0) { *c = 1.f; *s = 0.f; }
  else if (n == 90.0) { *c = 0.f; *s = 1.f; }
  else if (n == 180.0) { *c = -1.f; *s = 0.f; }
  else if (n == 270.0) { *c = 0.f; *s = -1.f; }
  else {
    const double r = angle_deg * 3.14159265358979323846 / 180.0;
    *c = (float)cos(r);
    *s = (float)sin(r);
  }
}

template <int QUARTER> static int launch_tile(const RotArgs& a, int pixel_bytes, bool tall, dim3 grid, hipStream_t s) {
  const dim3 block(kBlock);
  if (pixel_bytes == 3 && tall) {
    hipLaunchKernelGGL((k_rotate_tile<3, QUARTER, kRotTileHRgbTall>), grid, block, 0, s, a);
    return VALI_OK;
  }
  switch (pixel_bytes) {
#define VALI_ROT_CASE(P)                                                                    \
  case P:                                                                                   \
    hipLaunchKernelGGL((k_rotate_tile<P, QUARTER>), grid, block, 0, s, a);                   \
    break;
    VALI_ROT_CASE(1)
    VALI_ROT_CASE(2)
    VALI_ROT_CASE(3)
    VALI_ROT_CASE(4)
    VALI_ROT_CASE(6)
    VALI_ROT_CASE(12)
#undef VALI_ROT_CASE
  default:
    return fail(VALI_ERR_UNSUPPORTED, "rotate: unsupported pixel size %d", pixel_bytes);
  }
  return VALI_OK;
}

// ---- host side of the LDS-staged form ----
template <typename T, int C> static AffGeom aff_geom(float c, float s, int tw, int th) {
  const double ac = fabs((double)c), as = fabs((double)s);
  AffGeom g;
  // floor(max) + 1 - floor(min) + 1 <= floor(span) + 3 columns; + 1 of slack for the rounding of the float forms
  g.box_w = (int)floor((tw - 1) * ac + (th - 1) * as) + 4;
  g.box_h = (int)floor((tw - 1) * as + (th - 1) * ac) + 4;
  g.nch_max = (g.box_w * AffFmt<T, C>::PB + AffFmt<T, C>::GCH - 1) / AffFmt<T, C>::GCH;
  g.stride = 16 * g.nch_max + 4; // an odd number of dwords
  return g;
}
static unsigned aff_lds_bytes(const AffGeom& g) { return (unsigned)((g.box_h + 1) * g.stride + 32); }

// every plane of the launch must be at least as wide as the widest staged row (the box slides left to end with the row)
template <typename T, int C> static bool aff_plane_ok(const AffGeom& g, int psw) { return psw * AffFmt<T, C>::PB >= g.nch_max * AffFmt<T, C>::GCH; }

template <typename T, int TW, int PASSES, int CSEL = 0>
static bool launch_affine_lds(RotArgs& a, int n, int sw, int sh, int dw, int dh, hipStream_t stream) {
  constexpr int TH = kBlock / (TW / 4) * PASSES;
  const AffGeom g1 = aff_geom<T, 1>(a.c, a.s, TW, TH), g3 = aff_geom<T, 3>(a.c, a.s, TW, TH);
  bool one = false, three = false;
  u32 total = 0;
  for (int k = 0; k < a.njobs; ++k) {
    RotJob& j = a.job[k];
    const int psw = sw >> j.sub_x, pdw = dw >> j.sub_x, pdh = dh >> j.sub_y;
    if (j.channels == 1) {
      one = true;
      if (!aff_plane_ok<T, 1>(g1, psw) || (g1.box_h + 1) * g1.nch_max > kBlock * aff_max_pieces<T, 1, TW, TH>())
        return false;
    } else {
      three = true;
      if (CSEL == 1)
        return false;
      if (!aff_plane_ok<T, 3>(g3, psw) || (g3.box_h + 1) * g3.nch_max > kBlock * aff_max_pieces<T, 3, TW, TH>())
        return false;
    }
    j.first_tile = total;
    j.tiles_x = (u32)(pdw + TW - 1) / TW;
    total += j.tiles_x * (u32)((pdh + TH - 1) / TH);
  }
  const unsigned lds = max(one ? aff_lds_bytes(g1) : 0u, three ? aff_lds_bytes(g3) : 0u);
  if (lds > 64u * 1024u)
    return false;
  a.map = make_tile_map_linear(total, (u32)n);
  hipLaunchKernelGGL((k_rotate_affine_lds<T, TW, PASSES, CSEL>), tile_grid(a.map), dim3(kBlock), lds, stream, a, g1, g3);
  return true;
}

template <typename T> static bool launch_affine_lds_form(RotArgs& a, int form, int n, int sw, int sh, int dw, int dh, hipStream_t stream) {
  // tile shape: 64 x 64 destination pixels where the launch still has enough workgroups to fill the chip a few times over,
  // 32 x 32 for small launches and for float pixels (whose 64 x 64 box does not fit the LDS budget) -- profiles/r06_rotate.md
  if (form == 0) {
    long long tiles64 = 0;
    for (int k = 0; k < a.njobs; ++k)
      tiles64 += (long long)(((dw >> a.job[k].sub_x) + 63) / 64) * (((dh >> a.job[k].sub_y) + 63) / 64);
    bool one_channel = true;
    for (int k = 0; k < a.njobs; ++k)
      one_channel = one_channel && a.job[k].channels == 1;
    form = sizeof(T) == 4 ? 6
           : (sizeof(T) == 1 && one_channel && tiles64 * n >= 2 * kAffBigLaunchTiles) ? 5   // Y / planar YUV: the per-tile work weighs more, 64 x 128
           : tiles64 * n >= kAffBigLaunchTiles ? 4 : tiles64 * n >= kAffBigLaunchTiles / 4 ? 3 : 6;
  }
  if constexpr (sizeof(T) != 4) {
    switch (form) {
    case 2: return launch_affine_lds<T, 32, 2>(a, n, sw, sh, dw, dh, stream);
    case 5: // 64 x 128: one-channel 8-bit planes only (a packed-RGB box of that tile does not fit the LDS budget)
      if (sizeof(T) == 1 && launch_affine_lds<uint8_t, 64, 8, 1>(a, n, sw, sh, dw, dh, stream)) return true;
      if (launch_affine_lds<T, 64, 4>(a, n, sw, sh, dw, dh, stream)) return true;
      break;
    case 3: if (launch_affine_lds<T, 64, 2>(a, n, sw, sh, dw, dh, stream)) return true; break;
    case 4:
      if (launch_affine_lds<T, 64, 4>(a, n, sw, sh, dw, dh, stream))
        return true;
      break; // (a plane narrower than its staged row: 32 x 32)
    default: break;
    }
  }
  return launch_affine_lds<T, 32, 1>(a, n, sw, sh, dw, dh, stream);
}

// per_plane_shifts != 0: quarter turns whose shifts are derived from each plane's own size
// (what PySurfaceRotator's normalisation means for every plane); otherwise the given
// shifts apply to every plane unchanged (NPP semantics of RotPlanar).
static int launch_rotate(RotArgs& a, int fmt, int sw, int sh, int dw, int dh, double angle,
                         double shift_x, double shift_y, int per_plane_shifts, int n,
                         hipStream_t stream) {
  int elem = 1;
  a.njobs = rotate_jobs(fmt, a.job, &elem);
  if (!a.njobs)
    return fail(VALI_ERR_UNSUPPORTED, "rotate: unsupported pixel format %d", fmt);
  rotate_coeffs(angle, &a.c, &a.s);
  const bool q90 = a.c == 0.f && a.s == 1.f, q180 = a.c == -1.f, q270 = a.c == 0.f && a.s == -1.f;
  if (per_plane_shifts && !(q90 || q180 || q270 || a.c == 1.f))
    return fail(VALI_ERR_INVALID_ARG, "rotate: per-plane shifts need a multiple of 90 degrees");
  bool canonical = per_plane_shifts != 0 ||
                   (q90 && shift_x == 0.0 && shift_y == (double)(sw - 1) && a.njobs == 1) ||
                   (q270 && shift_y == 0.0 && shift_x == (double)(sh - 1) && a.njobs == 1);
  const bool no_tile = tuning(VALI_TUNE_ROTATE_NO_TILE) == 1;
  a.tile_order = tuning(VALI_TUNE_ROTATE_NO_TILE) == 2 ? 1 : 0;
  const bool tiled = canonical && (q90 || q270) && !no_tile;
  // half turn with each plane's own (W-1, H-1) shifts: a reversal
  const bool half = q180 && !no_tile && sw == dw && sh == dh &&
                    (per_plane_shifts != 0 ||
                     (a.njobs == 1 && shift_x == (double)(sw - 1) && shift_y == (double)(sh - 1)));

  const bool tall = (long long)sw * sh >= kRotTallPixels; // 3-byte pixels only (one plane: the surface size is the plane's)
  u32 total = 0;
  for (int k = 0; k < a.njobs; ++k) {
    RotJob& j = a.job[k];
    const int psw = sw >> j.sub_x, psh = sh >> j.sub_y, pdw = dw >> j.sub_x, pdh = dh >> j.sub_y;
    if (psw <= 0 || psh <= 0 || pdw <= 0 || pdh <= 0)
      return fail(VALI_ERR_INVALID_ARG, "rotate: surface too small for its chroma planes");
    if (per_plane_shifts) {
      j.shift_x = q180 ? (float)(psw - 1) : q270 ? (float)(psh - 1) : 0.f;
      j.shift_y = q90 ? (float)(psw - 1) : q180 ? (float)(psh - 1) : 0.f;
    } else {
      j.shift_x = (float)shift_x;
      j.shift_y = (float)shift_y;
    }
    j.first_tile = total;
    if (tiled) {
      const int th = elem * j.channels == 3 ? (tall ? kRotTileHRgbTall : kRotTileHRgb) : kRotTile;
      const int tw = elem * j.channels == 3 ? kRotTileWRgb : kRotTile;
      j.tiles_x = (u32)(psw + tw - 1) / tw;
      total += j.tiles_x * (u32)((psh + th - 1) / th);
    } else if (half) {
      j.tiles_x = (u32)(pdw + 255) / 256;
      total += j.tiles_x * (u32)((pdh + kHalfTileH - 1) / kHalfTileH);
    } else {
      j.tiles_x = (u32)(pdw + kAffineTile - 1) / kAffineTile;
      total += j.tiles_x * (u32)((pdh + kAffineTile - 1) / kAffineTile);
    }
  }
  if (!tiled && !half) {
    // any other angle: the LDS-staged form unless a plane is narrower than a staged row (or VALI_TUNE_ROTATE_AFFINE = 1)
    const int form = tuning(VALI_TUNE_ROTATE_AFFINE);
    const bool done = form == 1 ? false
                      : elem == 1 ? launch_affine_lds_form<uint8_t>(a, form, n, sw, sh, dw, dh, stream)
                      : elem == 2 ? launch_affine_lds_form<uint16_t>(a, form, n, sw, sh, dw, dh, stream)
                                  : launch_affine_lds_form<float>(a, form, n, sw, sh, dw, dh, stream);
    if (done) {
      VALI_LAUNCH_CHECK();
      return VALI_OK;
    }
    total = 0; // the gather form: its own tiling
    for (int k = 0; k < a.njobs; ++k) {
      RotJob& j = a.job[k];
      const int pdw = dw >> j.sub_x, pdh = dh >> j.sub_y;
      j.first_tile = total;
      j.tiles_x = (u32)(pdw + kAffineTile - 1) / kAffineTile;
      total += j.tiles_x * (u32)((pdh + kAffineTile - 1) / kAffineTile);
    }
  }
  a.map = make_tile_map_linear(total, (u32)n);
  const dim3 grid = tile_grid(a.map), block(kBlock);
  if (tiled) {
    const int pixel_bytes = elem * a.job[0].channels; // all jobs of a format share it
    const int rc = q90 ? launch_tile<1>(a, pixel_bytes, tall, grid, stream) : launch_tile<3>(a, pixel_bytes, tall, grid, stream);
    if (rc != VALI_OK)
      return rc;
  } else if (half) {
    const int rc = launch_half<>(a, elem * a.job[0].channels, grid, stream);
    if (rc != VALI_OK)
      return rc;
  } else if (elem == 1) {
    hipLaunchKernelGGL(k_rotate_affine<uint8_t>, grid, block, 0, stream, a);
  } else if (elem == 2) {
    hipLaunchKernelGGL(k_rotate_affine<uint16_t>, grid, block, 0, stream, a);
  } else {
    hipLaunchKernelGGL(k_rotate_affine<float>, grid, block, 0, stream, a);
  }
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali

using namespace vali;

extern "C" {

int vali_rotate_coeffs(double angle_deg, float* c, float* s) {
  VALI_REQUIRE(c && s, "null argument");
  rotate_coeffs(angle_deg, c, s);
  return VALI_OK;
}

int vali_rotate(const vali_surface* src, const vali_surface* dst, double angle, double shift_x,
                double shift_y, int per_plane_shifts, vali_stream_t stream) {
  VALI_REQUIRE(src && dst, "null argument");
  VALI_REQUIRE(src->format == dst->format, "src/dst format mismatch");
  VALI_REQUIRE(src->width > 0 && src->height > 0 && dst->width > 0 && dst->height > 0, "empty surface");
  VALI_REQUIRE(src->plane[0] && dst->plane[0], "null plane");
  VALI_REQUIRE(planes_fit_32bit(*src) && planes_fit_32bit(*dst), "plane of 4 GiB or more");
  RotArgs a = {};
  a.sw = src->width; a.sh = src->height; a.dw = dst->width; a.dh = dst->height;
  int elem = 1;
  const int nj = rotate_jobs(src->format, a.job, &elem);
  for (int k = 0; k < nj; ++k) { // resolve the planes on the host (see PlaneJob)
    const int c = a.job[k].comp;
    VALI_REQUIRE(src->plane[c] && dst->plane[c], "null plane");
    a.job[k].sp = (const uint8_t*)src->plane[c];
    a.job[k].dp = (uint8_t*)dst->plane[c];
    a.job[k].spitch = src->pitch[c];
    a.job[k].dpitch = dst->pitch[c];
  }
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_rotate(a, src->format, src->width, src->height, dst->width, dst->height, angle,
                       shift_x, shift_y, per_plane_shifts, 1, s);
}

int vali_rotate_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int format,
                      int src_width, int src_height, int dst_width, int dst_height, double angle,
                      double shift_x, double shift_y, int per_plane_shifts, vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst, "null argument");
  VALI_REQUIRE(src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0, "empty geometry");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  RotArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_rotate(a, format, src_width, src_height, dst_width, dst_height, angle, shift_x,
                       shift_y, per_plane_shifts, n, s);
}

} // extern "C"
