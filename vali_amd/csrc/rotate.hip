// Plane rotation: replaces nppiRotate_{8u,16u,32f}_{C1,C3}R_Ctx with NPPI_INTER_LINEAR as the
// reference calls it from Rot_8U_C1 ... Rot_32F_C3
// (reference: src/TC/src/RotateSurface.cpp:22-125; per-plane / packed drivers :132-159).
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
// Two kernels:
//  * k_rotate_affine<T,C>: any angle.  cos/sin arrive from the host as floats (snapped to
//    exactly 0/+-1 for multiples of 90 degrees), so the device does only fma/mul/add and is
//    bit-exact with the oracle.  One lane = 4 adjacent dst pixels (wide stores).
//  * k_rotate_tile<P>: the canonical 90 / 270 degree permutations as an LDS-tiled transpose:
//    a 64x64-pixel tile is read with coalesced row segments, written with coalesced row
//    segments of the transposed tile; LDS row stride 64*P+4 bytes keeps the column walks
//    at most 2-way bank conflicted.  P = bytes per pixel (1,2,3,4,6,12).
// Arithmetic of the affine path (the specification, oracle: vali_oracle_rotate):
//   dx = x' - shift_x ; dy = y' - shift_y                      (float)
//   xs = fma(-s, dy, c*dx) ; ys = fma(c, dy, s*dx)
//   skip unless 0 <= xs <= W-1 and 0 <= ys <= H-1
//   i = floor(xs), a = xs - i ; j = floor(ys), b = ys - j ; i1 = min(i+1, W-1), j1 likewise
//   t0 = fma(a, T[j][i1]-T[j][i], T[j][i]) ; t1 likewise on row j1 ; v = fma(b, t1-t0, t0)
//   u8/u16: round-half-even + saturate ; f32: v.
#include "common.hpp"
#include "dev_util.hpp"

#include <math.h>

namespace vali {

struct RotArgs {
  const uint8_t* src;
  uint8_t* dst;
  int src_pitch, dst_pitch;
  int src_w, src_h, dst_w, dst_h;
  float c, s, shift_x, shift_y;
  TileMap map;
};

template <typename T> __device__ __forceinline__ float texel_f(const uint8_t* row, int idx) {
  return (float)((const T*)row)[idx];
}

template <typename T> __device__ __forceinline__ T finish(float v);
template <> __device__ __forceinline__ uint8_t finish<uint8_t>(float v) {
  return (uint8_t)quantize_u8(v);
}
template <> __device__ __forceinline__ uint16_t finish<uint16_t>(float v) {
  float r = __builtin_rintf(v);
  r = __builtin_fminf(__builtin_fmaxf(r, 0.0f), 65535.0f);
  return (uint16_t)r;
}
template <> __device__ __forceinline__ float finish<float>(float v) { return v; }

template <typename T, int C>
__global__ void __launch_bounds__(kBlock) k_rotate_affine(const RotArgs a) {
  u32 tile_x, tile_y;
  if (!tile_of_block(a.map, tile_x, tile_y))
    return;
  const int x0 = (tile_x * 64 + (threadIdx.x & 63)) * 4;
  const int y = tile_y * 4 + (threadIdx.x >> 6);
  if (x0 >= a.dst_w || y >= a.dst_h)
    return;
  const float dy = (float)y - a.shift_y;
  const float wmax = (float)(a.src_w - 1), hmax = (float)(a.src_h - 1);
  T out[4][C];
  bool hit[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float dx = (float)(x0 + p) - a.shift_x;
    const float xs = __builtin_fmaf(-a.s, dy, a.c * dx);
    const float ys = __builtin_fmaf(a.c, dy, a.s * dx);
    hit[p] = (x0 + p < a.dst_w) && xs >= 0.0f && xs <= wmax && ys >= 0.0f && ys <= hmax;
    if (hit[p]) {
      const float fi = __builtin_floorf(xs), fj = __builtin_floorf(ys);
      const float fa = xs - fi, fb = ys - fj;
      const int i = (int)fi, j = (int)fj;
      const int i1 = min(i + 1, a.src_w - 1), j1 = min(j + 1, a.src_h - 1);
      const uint8_t* r0 = a.src + (size_t)j * a.src_pitch;
      const uint8_t* r1 = a.src + (size_t)j1 * a.src_pitch;
#pragma unroll
      for (int ch = 0; ch < C; ++ch) {
        const float t00 = texel_f<T>(r0, i * C + ch), t10 = texel_f<T>(r0, i1 * C + ch);
        const float t01 = texel_f<T>(r1, i * C + ch), t11 = texel_f<T>(r1, i1 * C + ch);
        const float t0 = __builtin_fmaf(fa, t10 - t00, t00);
        const float t1 = __builtin_fmaf(fa, t11 - t01, t01);
        out[p][ch] = finish<T>(__builtin_fmaf(fb, t1 - t0, t0));
      }
    }
  }
  T* drow = (T*)(a.dst + (size_t)y * a.dst_pitch) + (size_t)x0 * C;
  constexpr int kBytes = 4 * C * (int)sizeof(T);
  if (hit[0] && hit[1] && hit[2] && hit[3] && (((uintptr_t)drow) & (kBytes % 16 == 0 ? 15u : 3u)) == 0 &&
      kBytes % 4 == 0) {
    // all four pixels present: one wide store (4, 12, 16, 24 or 48 bytes per lane)
    u32 w[kBytes / 4];
    __builtin_memcpy(w, out, kBytes);
    u32* o = (u32*)drow;
    if constexpr (kBytes % 16 == 0) {
#pragma unroll
      for (int k = 0; k < kBytes / 16; ++k)
        ((uint4*)o)[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    } else {
#pragma unroll
      for (int k = 0; k < kBytes / 4; ++k)
        o[k] = w[k];
    }
  } else {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      if (hit[p])
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
          drow[p * C + ch] = out[p][ch];
  }
}

// ---- canonical 90 / 270 degree permutation: LDS-tiled transpose -----------------------------
// QUARTER = 1: dst(x', y') = src(W-1-y', x')   [angle 90,  shift_y = W-1]
// QUARTER = 3: dst(x', y') = src(y', H-1-x')   [angle 270, shift_x = H-1]
constexpr int kRotTile = 64;

template <int P, int QUARTER>
__global__ void __launch_bounds__(kBlock) k_rotate_tile(const RotArgs a) {
  constexpr int S = kRotTile * P + 4; // LDS row stride in bytes (dword aligned, odd dwords)
  __shared__ __attribute__((aligned(16))) uint8_t lds[kRotTile * S];
  u32 tile_x, tile_y;
  if (!tile_of_block(a.map, tile_x, tile_y))
    return;
  const int cx = tile_x * kRotTile, ry = tile_y * kRotTile; // src tile origin (col, row)
  const int tw = min(kRotTile, a.src_w - cx), th = min(kRotTile, a.src_h - ry);
  const int t = threadIdx.x;

  // phase 1: coalesced row segments -> LDS (dwords when the segment is dword aligned)
  const uint8_t* sbase = a.src + (size_t)ry * a.src_pitch + (size_t)cx * P;
  const int row_bytes = tw * P;
  if ((((uintptr_t)sbase | (uintptr_t)a.src_pitch) & 3u) == 0 && (row_bytes & 3) == 0) {
    const int dw_per_row = row_bytes / 4;
    for (int k = t; k < th * dw_per_row; k += kBlock) {
      const int r = k / dw_per_row, d = k - r * dw_per_row;
      *(u32*)(lds + r * S + d * 4) = *(const u32*)(sbase + (size_t)r * a.src_pitch + d * 4);
    }
  } else {
    for (int k = t; k < th * row_bytes; k += kBlock) {
      const int r = k / row_bytes, d = k - r * row_bytes;
      lds[r * S + d] = sbase[(size_t)r * a.src_pitch + d];
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
      dy_ = a.src_w - 1 - (cx + lc);
      dx0 = ry + chunk * 4;
    } else {
      dy_ = cx + lc;
      dx0 = a.src_h - 1 - (ry + chunk * 4 + 3);
    }
    if (dy_ < 0 || dy_ >= a.dst_h)
      continue;
    // the 4 pixels of this lane: dst x = dx0 + q  <-  tile row j(q)
    uint8_t px[4 * P];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = QUARTER == 1 ? chunk * 4 + q : chunk * 4 + 3 - q;
      ok[q] = j < th && dx0 + q >= 0 && dx0 + q < a.dst_w;
#pragma unroll
      for (int b = 0; b < P; ++b)
        px[q * P + b] = ok[q] ? lds[j * S + lc * P + b] : (uint8_t)0;
    }
    uint8_t* o = a.dst + (size_t)dy_ * a.dst_pitch + (size_t)dx0 * P;
    if (ok[0] && ok[1] && ok[2] && ok[3] && (((uintptr_t)o) & 3u) == 0 && (4 * P) % 4 == 0) {
      u32 w[P];
      __builtin_memcpy(w, px, 4 * P);
#pragma unroll
      for (int k = 0; k < P; ++k)
        ((u32*)o)[k] = w[k];
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q])
#pragma unroll
          for (int b = 0; b < P; ++b)
            o[q * P + b] = px[q * P + b];
    }
  }
}

template <int QUARTER> static int launch_tile(const RotArgs& a, int pixel_bytes, hipStream_t s) {
  const dim3 grid(a.map.per_xcd * 8u), block(kBlock);
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

} // namespace vali

using namespace vali;

extern "C" {

int vali_rotate_coeffs(double angle_deg, float* c, float* s) {
  VALI_REQUIRE(c && s, "null argument");
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
  return VALI_OK;
}

int vali_rotate_plane(const void* src, int src_pitch, int src_width, int src_height, void* dst,
                      int dst_pitch, int dst_width, int dst_height, int elem_size, int channels,
                      double angle, double shift_x, double shift_y, vali_stream_t stream) {
  VALI_REQUIRE(src && dst, "null plane");
  VALI_REQUIRE(src_width > 0 && src_height > 0 && dst_width > 0 && dst_height > 0, "empty plane");
  VALI_REQUIRE(elem_size == 1 || elem_size == 2 || elem_size == 4, "elem_size must be 1, 2 or 4");
  VALI_REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
  RotArgs a = {};
  a.src = (const uint8_t*)src;
  a.dst = (uint8_t*)dst;
  a.src_pitch = src_pitch; a.dst_pitch = dst_pitch;
  a.src_w = src_width; a.src_h = src_height; a.dst_w = dst_width; a.dst_h = dst_height;
  vali_rotate_coeffs(angle, &a.c, &a.s);
  a.shift_x = (float)shift_x;
  a.shift_y = (float)shift_y;
  hipStream_t s = as_stream(stream);
  DeviceScope scope(stream_device(s));

  // canonical quarter turns -> tiled transpose
  const bool q90 = a.c == 0.f && a.s == 1.f && shift_x == 0.0 && shift_y == (double)(src_width - 1);
  const bool q270 = a.c == 0.f && a.s == -1.f && shift_y == 0.0 && shift_x == (double)(src_height - 1);
  static const bool no_tile = [] { const char* e = getenv("VALI_ROTATE_NO_TILE"); return e && e[0] == '1'; }();
  if ((q90 || q270) && !no_tile) {
    a.map = make_tile_map((src_width + kRotTile - 1) / kRotTile, (src_height + kRotTile - 1) / kRotTile);
    const int rc = q90 ? launch_tile<1>(a, elem_size * channels, s) : launch_tile<3>(a, elem_size * channels, s);
    if (rc != VALI_OK)
      return rc;
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }

  a.map = make_tile_map((dst_width + 255) / 256, (dst_height + 3) / 4);
  const dim3 grid(a.map.per_xcd * 8u), block(kBlock);
#define VALI_ROT_AFFINE(T, C) hipLaunchKernelGGL((k_rotate_affine<T, C>), grid, block, 0, s, a)
  if (elem_size == 1) { if (channels == 1) VALI_ROT_AFFINE(uint8_t, 1); else VALI_ROT_AFFINE(uint8_t, 3); }
  else if (elem_size == 2) { if (channels == 1) VALI_ROT_AFFINE(uint16_t, 1); else VALI_ROT_AFFINE(uint16_t, 3); }
  else { if (channels == 1) VALI_ROT_AFFINE(float, 1); else VALI_ROT_AFFINE(float, 3); }
#undef VALI_ROT_AFFINE
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // extern "C"
