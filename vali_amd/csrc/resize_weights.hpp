// Tap weights of the Lanczos-3 / bicubic resizers, shared by the rows-first kernel (resize_taps.hip: planes that grow
// vertically) and the columns-first kernel (resize_cols.hip: planes that shrink vertically).  Specification:
// oracle/vali_oracle.c (vali_oracle_lanczos3_weights, vali_oracle_cubic_weights, lz_make_tap) -- the same IEEE
// operations in the same order.
#pragma once
#include "dev_util.hpp"

namespace vali {

__device__ __forceinline__ float lz_sin_poly(float z) { // sin z, 0 <= z <= pi/2
  const float z2 = z * z;
  float p = __builtin_fmaf(z2, -2.5052108e-8f, 2.7557319e-6f);
  p = __builtin_fmaf(z2, p, -1.9841270e-4f);
  p = __builtin_fmaf(z2, p, 8.3333333e-3f);
  p = __builtin_fmaf(z2, p, -1.6666667e-1f);
  return __builtin_fmaf(z * z2, p, z);
}
__device__ __forceinline__ float lz_cos_poly(float z) { // cos z, 0 <= z <= pi/3
  const float z2 = z * z;
  float p = __builtin_fmaf(z2, 2.0876757e-9f, -2.7557319e-7f);
  p = __builtin_fmaf(z2, p, 2.4801587e-5f);
  p = __builtin_fmaf(z2, p, -1.3888889e-3f);
  p = __builtin_fmaf(z2, p, 4.1666667e-2f);
  p = __builtin_fmaf(z2, p, -0.5f);
  return __builtin_fmaf(z2, p, 1.0f);
}
// oracle: vali_oracle_lanczos3_weights -- two IEEE divisions (1 / product of the six t^2, 1 / sum), not twelve
__device__ __forceinline__ void lanczos3_weights(float a, float (&w)[6]) {
  const float y = a <= 0.5f ? a : 1.0f - a;
  const float s1 = lz_sin_poly(y * 3.14159265f);
  const float z = a * 1.04719755f;
  const float s3 = lz_sin_poly(z), c3 = lz_cos_poly(z);
  const float h = 0.866025404f;
  const float q[6] = {__builtin_fmaf(c3, h, -0.5f * s3), __builtin_fmaf(c3, h, 0.5f * s3), s3,
                      __builtin_fmaf(c3, -h, 0.5f * s3), __builtin_fmaf(c3, -h, -0.5f * s3), -s3};
  float d[6], inv[6], raw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float t = a + (float)(2 - k);
    d[k] = t * t;
  }
  const float p01 = d[0] * d[1], p23 = d[2] * d[3], p45 = d[4] * d[5];
  const float r = 1.0f / ((p01 * p23) * p45);
  inv[0] = r * ((d[1] * p23) * p45); inv[1] = r * ((d[0] * p23) * p45);
  inv[2] = r * ((p01 * d[3]) * p45); inv[3] = r * ((p01 * d[2]) * p45);
  inv[4] = r * ((p01 * p23) * d[5]); inv[5] = r * ((p01 * p23) * d[4]);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float s = (k & 1) ? -s1 : s1;
    raw[k] = (s * q[k]) * inv[k];
  }
  const float sum = ((((raw[0] + raw[1]) + raw[2]) + raw[3]) + raw[4]) + raw[5];
  const float rs = 1.0f / sum;
  const bool on_grid = a == 0.0f;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    w[k] = on_grid ? (k == 2 ? 1.0f : 0.0f) : raw[k] * rs;
}

// Keys / Catmull-Rom cubic convolution (a = -1/2), taps i-1 .. i+2; oracle: vali_oracle_cubic_weights
__device__ __forceinline__ void cubic_weights(float a, float (&w)[4]) {
  const float a2 = a * a;
  w[0] = a * __builtin_fmaf(a, __builtin_fmaf(a, -0.5f, 1.0f), -0.5f);
  w[1] = __builtin_fmaf(a2, __builtin_fmaf(a, 1.5f, -2.5f), 1.0f);
  w[2] = a * __builtin_fmaf(a, __builtin_fmaf(a, -1.5f, 2.0f), 0.5f);
  w[3] = a2 * __builtin_fmaf(a, 0.5f, -0.5f);
}

// TAPS = 6: Lanczos-3, TAPS = 4: cubic; the taps are i - kBefore .. i + TAPS - 1 - kBefore
template <int TAPS> struct LzTap {
  static constexpr int kBefore = TAPS / 2 - 1;
  int i;      // floor of the source coordinate
  float w[TAPS];
};
template <int TAPS> __device__ __forceinline__ LzTap<TAPS> make_lz_tap(int x, float scale) {
  const float f = (float)x * scale;
  const float fl = __builtin_floorf(f);
  LzTap<TAPS> t;
  if constexpr (TAPS == 6)
    lanczos3_weights(f - fl, t.w);
  else
    cubic_weights(f - fl, t.w);
  t.i = (int)fl;
  return t;
}
__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

typedef float v2f32 __attribute__((ext_vector_type(2)));

} // namespace vali
