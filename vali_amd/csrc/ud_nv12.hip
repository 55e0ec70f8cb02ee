// UD ("upsample chroma + downscale/resize + optional YUV->RGB") of NV12 / P10 surfaces.
//
// Replaces the reference's only first-party CUDA kernels, RescaleConvertYUV<T> and
// RescaleConvertRGB<T> with their launchers UD_NV12 / UD_NV12_HBD
// (reference: src/TC/src/ResizeUtils.cu:21-176, src/TC/inc/ResizeUtils.hpp:30-50; callers
// UDSemiPlanar / UDSemiPlanarHBD, src/TC/src/UDSurface.cpp:84-115).
//
// The reference samples two CUDA texture objects (Y: WxH 1-ch, UV: W/2 x H/2 2-ch) with
// cudaFilterModeLinear + cudaReadModeNormalizedFloat at unnormalised coordinates
// X = x / scale_x (no half-pixel centring), address mode clamp.  gfx950 has no texture
// sampler path worth using here, so the filter is restated arithmetically
// (CUDA programming guide, "Linear Filtering"):
//   xB = X - 0.5 ; i = floor(xB) ; alpha = frac(xB) rounded to 8 fractional bits
//   tex = sum_{ij} w_ij * T[clamp(i+di), clamp(j+dj)],  w from alpha/beta, sum(w) = 65536
// evaluated EXACTLY in integers, then normalised with one float multiply
//   val = float(S) * float(1 / (65536 * max))        max = 255 (u8) or 65535 (u16)
// Output stage exactly as the reference: YUV: (T)(val * 2^bits); RGB: u -= .5, v -= .5,
//   r = y + 1.140 v ; g = y - 0.394 u - 0.581 v ; b = y + 2.032 u   (nvcc contracts these
//   into FMAs; restated with explicit fmaf), u8: x256, truncate toward zero, saturate
//   (what cvt.rzi.u8.f32 does; pinned by the reference goldens: RGB == trunc(RGB_32F*256)).
// Oracle: vali_oracle_ud_nv12 (oracle/vali_oracle.c), bit-exact.
//
// Work decomposition: one lane = 4 adjacent dst pixels of a row, so u8 planar output is a
// dword, u8 packed 12 B (dwordx3), f32 16/48 B per lane; a workgroup = 256 x 32 dst pixels,
// each wave walking 8 rows with the same column taps.  The source rows a dst row samples are
// staged per wave in LDS with 16-byte coalesced loads (every 128-byte line fetched once);
// tiles walk the frame through the XCD-contiguous TileMap.  HBM traffic = touched source
// rows + the dst surface.
//
// Kernels in this file:
//   k_ud_nv12<T,OUT,STAGED,ROT>  any scale factor, NV12 / P10, every output; VALU-bound (4.8 us per
//                                2160p -> 1080p frame)
//   k_ud_lean<OUT,RATIO>         source exactly twice as wide as the output (RATIO 2) or as wide (1), NV12, any height,
//                                8-bit outputs, un-rotated, width a multiple of 8: wide loads, no LDS staging
//                                (round 4; round 1's k_ud_down2 also took ragged widths and the half turn and spilled)
//   k_ud_half / k_ud_32 / k_ud_half_t   exactly 2:1 and 3:2 both ways, and 2:1 turned by 90 / 270 degrees
//   k_ud_down2_t<ROT>            2:1 along x at any height for the 90 / 270 degree outputs
// all produce the same bits (tests/test_gpu_ud_down2.py); profiles/r01_ud_down2.md has the
// counters that led from the first to the other two.
#include "common.hpp"
#include "dev_util.hpp"

#include <stdlib.h>
#include <type_traits>

namespace vali {

enum : int {
  UD_YUV444 = 0,      // 3 planes, same element type as the source
  UD_RGB_U8 = 1,      // packed
  UD_RGB_U8_PLANAR = 2,
  UD_RGB_F32 = 3,     // packed
  UD_RGB_F32_PLANAR = 4
};

struct UdArgs {
  const vali_surface* d_src;
  const vali_surface* d_dst;
  vali_surface src, dst;
  TileMap map;
  int rows; // dst rows a wave walks: kUdRowsPerWave, or fewer when the launch is small (ud_rows_for); rotated outputs: always 8
};

template <typename T> struct TexelTraits;
template <> struct TexelTraits<uint8_t> {
  static constexpr float kInvDen = 1.0f / 16711680.0f;    // 1 / (65536 * 255)
  static constexpr float kMax = 256.0f;
};
template <> struct TexelTraits<uint16_t> {
  static constexpr float kInvDen = 1.0f / 4294901760.0f;  // 1 / (65536 * 65535)
  static constexpr float kMax = 65536.0f;
};

struct Tap {
  int i0, i1;   // clamped texel indices
  u32 w0, w1;   // 8.8 weights, w0 + w1 = 256
};

// coordinate -> two taps; `coord` is the unnormalised texture coordinate.
__host__ __device__ inline Tap make_tap(float coord, int size) {
  const float b = coord - 0.5f;
  const float fl = __builtin_floorf(b);
  const float frac = b - fl;
  const u32 q = (u32)(frac * 256.0f + 0.5f); // 0..256
  const int i = (int)fl;
  Tap t;
  t.i0 = i < 0 ? 0 : (i > size - 1 ? size - 1 : i);
  t.i1 = i + 1 < 0 ? 0 : (i + 1 > size - 1 ? size - 1 : i + 1);
  // the masks change no value (q is 0..256); they tell the compiler both weights fit 9 bits
  // so the bilinear products select the full-rate v_mul_u32_u24 / v_mad_u32_u24
  t.w1 = q & 0x1ffu;
  t.w0 = (256u - q) & 0x1ffu;
  return t;
}

// float -> integer conversion of the reference (cvt.rzi + saturation)
template <typename T> __device__ __forceinline__ u32 trunc_sat(float v) {
  constexpr float hi = sizeof(T) == 1 ? 255.0f : 65535.0f;
  const float c = __builtin_fminf(__builtin_fmaxf(v, 0.0f), hi); // NaN -> 0 like cvt.rzi
  return (u32)c;                                                  // v_cvt_u32_f32 truncates
}

// per-wave staging: the two luma rows and the two chroma rows a dst row samples.  Lane l owns
// bytes [16 l, 16 l + 16) of each of the four rows: one prefetch register per row, source
// address = wave-uniform row base (SGPRs) + a per-lane constant, no per-row address arithmetic.
constexpr int kUdRowBytes = kWave * 16;         // 1 KiB per staged row
constexpr int kUdRowsPerWave = 8; // dst rows a wave walks with the same column taps
constexpr int kUdTileH = kWavesPerBlock * kUdRowsPerWave;
#ifndef VALI_UD_DEPTH
#define VALI_UD_DEPTH 2
#endif
constexpr int kUdDepth = VALI_UD_DEPTH; // dst rows of prefetch in the staged general kernel
// CH = 16-byte chunks per lane per row: 1 for 8-bit sources, 2 for 16-bit ones (P10: the same
// 2x downscale spans twice the bytes; with 1 KiB rows it fell to the gather path, 17.7 us)
template <int CH> struct alignas(16) UdStage {
  uint8_t luma[2][CH * kUdRowBytes];
  uint8_t chroma[2][CH * kUdRowBytes];
};

// Scale folded into the normalisation constant so the output stage is a bare truncation:
// 2^k * (float(S) * kInvDen) == float(S) * (2^k * kInvDen) and 2^k * fma(a, v, y) ==
// fma(a, 2^k v, 2^k y) bit for bit (power-of-two scaling commutes with IEEE rounding; nothing
// here is near the subnormal range), so the reference's "val * 256" costs no instruction.
template <typename T, int OUT> struct UdScale {
  static constexpr float value = OUT == UD_YUV444 ? TexelTraits<T>::kMax
                                 : (OUT == UD_RGB_U8 || OUT == UD_RGB_U8_PLANAR) ? 256.0f : 1.0f;
};

// Twelve pre-scaled values -> 12 TRUNCATED, saturated bytes == the reference's cvt.rzi.u8.f32 + packing.
// v_cvt_pk_u8_f32 converts in the wave's FP32 rounding mode (MODE[1:0]; measured on gfx950 over 65536 values incl.
// ties, negatives, > 255 and NaN: mode 3 = toward zero gives exactly trunc + saturate, tools/exp/rtz.hip), so the
// separate v_trunc_f32 per byte (12 of the ~86 VALU instructions of four pixels in the exact-2x kernel) becomes two
// scalar s_setreg around a block of twelve converts.  ONE asm statement: its inputs are finished before it starts
// and nothing else can be scheduled inside, so no other float instruction ever runs in the switched mode.
// Layout kQuads: w[d] = bytes v[4d..4d+3]; !kQuads: w[p] = bytes v[3p], v[3p+1], v[3p+2], 0 (one dword per pixel).
template <bool kQuads>
__device__ __forceinline__ void trunc_pack12(const float (&v)[12], u32* w) {
  if constexpr (kQuads) {
    u32 w0, w1, w2;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 3, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %10, 3, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %13, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %14, 3, %2\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]));
    w[0] = w0; w[1] = w1; w[2] = w2;
  } else {
    u32 w0, w1, w2, w3;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                 "v_cvt_pk_u8_f32 %0, %4, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %0, %5, 1, %0\n\t"
                 "v_cvt_pk_u8_f32 %0, %6, 2, %0\n\t"
                 "v_cvt_pk_u8_f32 %1, %7, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %1, %8, 1, %1\n\t"
                 "v_cvt_pk_u8_f32 %1, %9, 2, %1\n\t"
                 "v_cvt_pk_u8_f32 %2, %10, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %2, %11, 1, %2\n\t"
                 "v_cvt_pk_u8_f32 %2, %12, 2, %2\n\t"
                 "v_cvt_pk_u8_f32 %3, %13, 0, 0\n\t"
                 "v_cvt_pk_u8_f32 %3, %14, 1, %3\n\t"
                 "v_cvt_pk_u8_f32 %3, %15, 2, %3\n\t"
                 "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0"
                 : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                 : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(v[8]),
                   "v"(v[9]), "v"(v[10]), "v"(v[11]));
    w[0] = w0; w[1] = w1; w[2] = w2; w[3] = w3;
  }
}
// three dwords of four bytes each: (a0..a3), (b0..b3), (c0..c3)
__device__ __forceinline__ void trunc_pack3x4(float a0, float a1, float a2, float a3, float b0, float b1, float b2, float b3,
                                              float c0, float c1, float c2, float c3, u32& wa, u32& wb, u32& wc) {
  const float v[12] = {a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3};
  u32 w[3];
  trunc_pack12<true>(v, w);
  wa = w[0]; wb = w[1]; wc = w[2];
}
// one dword per pixel (bytes c0, c1, c2, 0) for 4 pixels
__device__ __forceinline__ void trunc_pack_px4(const float* c0, const float* c1, const float* c2, u32* px) {
  const float v[12] = {c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2], c0[3], c1[3], c2[3]};
  trunc_pack12<false>(v, px);
}

// store 4 pixels of one dst row; c0/c1/c2 = Y,U,V or R,G,B already multiplied by
// UdScale<T,OUT>; n = valid pixels; y is wave-uniform
// The vector stores of ud_store: neighbouring lanes write neighbouring pixels, so a wave instruction covers whole
// 128-byte lines -> non-temporal (VALI_UD_NT_STORES=0 builds the plain forms for an A/B).  Packed RGB_32F stays
// plain: its three 16-byte pieces per lane are 48 bytes apart.
#ifndef VALI_UD_NT_STORES
#define VALI_UD_NT_STORES 1
#endif
#if VALI_UD_NT_STORES
#define UD_ST(T) gstore_nt<T>
#define UD_ST8 store8_nt
#define UD_ST16F store16f_nt
#define UD_ST16 gstore16_nt
#else
#define UD_ST(T) gstore<T>
#define UD_ST8 store8
#define UD_ST16F store16f
#define UD_ST16 gstore16
#endif
template <typename T, int OUT>
__device__ __forceinline__ void ud_store(const SurfRef& d, int x0, int y, int n, const float (&c0)[4],
                                         const float (&c1)[4], const float (&c2)[4]) {
  uint8_t* pd0 = d.p[0];
  uint8_t* pd1 = d.p[1];
  uint8_t* pd2 = d.p[2];
  const int dp0 = d.pitch[0], dp1 = d.pitch[1], dp2 = d.pitch[2];
  if constexpr (OUT == UD_YUV444) {
    uint8_t* o0 = pd0 + (u32)(y * dp0) + (size_t)x0 * sizeof(T);
    uint8_t* o1 = pd1 + (u32)(y * dp1) + (size_t)x0 * sizeof(T);
    uint8_t* o2 = pd2 + (u32)(y * dp2) + (size_t)x0 * sizeof(T);
    constexpr u32 kA = sizeof(T) == 1 ? 3u : 7u;
    const bool fast = n == 4 && ((((uintptr_t)o0) | ((uintptr_t)o1) | ((uintptr_t)o2)) & kA) == 0;
    if constexpr (sizeof(T) == 1) {
      u32 w0, w1, w2;
      trunc_pack3x4(c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], w0, w1, w2);
      if (fast) {
        UD_ST(u32)(o0, w0); UD_ST(u32)(o1, w1); UD_ST(u32)(o2, w2);
      } else {
        for (int p = 0; p < n; ++p) {
          gstore<uint8_t>(o0 + p, (uint8_t)(w0 >> (8 * p))); gstore<uint8_t>(o1 + p, (uint8_t)(w1 >> (8 * p)));
          gstore<uint8_t>(o2 + p, (uint8_t)(w2 >> (8 * p)));
        }
      }
    } else {
      u32 q0[4], q1[4], q2[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        q0[p] = trunc_sat<T>(c0[p]); q1[p] = trunc_sat<T>(c1[p]); q2[p] = trunc_sat<T>(c2[p]);
      }
      if (fast) {
        UD_ST8(o0, make_uint2(q0[0] | (q0[1] << 16), q0[2] | (q0[3] << 16)));
        UD_ST8(o1, make_uint2(q1[0] | (q1[1] << 16), q1[2] | (q1[3] << 16)));
        UD_ST8(o2, make_uint2(q2[0] | (q2[1] << 16), q2[2] | (q2[3] << 16)));
      } else {
        for (int p = 0; p < n; ++p) {
          gstore<T>(o0 + p * sizeof(T), (T)q0[p]); gstore<T>(o1 + p * sizeof(T), (T)q1[p]); gstore<T>(o2 + p * sizeof(T), (T)q2[p]);
        }
      }
    }
  } else if constexpr (OUT == UD_RGB_U8_PLANAR) {
    u32 wr, wg, wb;
    trunc_pack3x4(c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3], c2[0], c2[1], c2[2], c2[3], wr, wg, wb);
    uint8_t* o0 = pd0 + (u32)(y * dp0) + x0;
    uint8_t* o1 = pd1 + (u32)(y * dp0) + x0;
    uint8_t* o2 = pd2 + (u32)(y * dp0) + x0;
    if (n == 4 && ((((uintptr_t)o0) | ((uintptr_t)o1) | ((uintptr_t)o2)) & 3u) == 0) {
      UD_ST(u32)(o0, wr); UD_ST(u32)(o1, wg); UD_ST(u32)(o2, wb);
    } else {
      for (int p = 0; p < n; ++p) {
        gstore<uint8_t>(o0 + p, (uint8_t)(wr >> (8 * p))); gstore<uint8_t>(o1 + p, (uint8_t)(wg >> (8 * p)));
        gstore<uint8_t>(o2 + p, (uint8_t)(wb >> (8 * p)));
      }
    }
  } else if constexpr (OUT == UD_RGB_U8) {
    // bytes in memory order: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    u32 w0, w1, w2;
    trunc_pack3x4(c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2], c0[3], c1[3], c2[3], w0, w1, w2);
    uint8_t* o = pd0 + (u32)(y * dp0) + (size_t)x0 * 3;
    if (n == 4 && (((uintptr_t)o) & 3u) == 0) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = {w0, w1, w2};
      UD_ST(v3u32)(o, w); // global_store_dwordx3
    } else {
      const u32 ww[3] = {w0, w1, w2};
      for (int k = 0; k < 3 * n; ++k)
        gstore<uint8_t>(o + k, (uint8_t)(ww[k >> 2] >> (8 * (k & 3))));
    }
  } else if constexpr (OUT == UD_RGB_F32_PLANAR) {
    uint8_t* o0 = pd0 + (u32)(y * dp0) + (size_t)x0 * 4;
    uint8_t* o1 = pd1 + (u32)(y * dp0) + (size_t)x0 * 4;
    uint8_t* o2 = pd2 + (u32)(y * dp0) + (size_t)x0 * 4;
    if (n == 4 && ((((uintptr_t)o0) | ((uintptr_t)o1) | ((uintptr_t)o2)) & 15u) == 0) {
      UD_ST16F(o0, make_float4(c0[0], c0[1], c0[2], c0[3]));
      UD_ST16F(o1, make_float4(c1[0], c1[1], c1[2], c1[3]));
      UD_ST16F(o2, make_float4(c2[0], c2[1], c2[2], c2[3]));
    } else {
      for (int p = 0; p < n; ++p) { gstore<float>(o0 + 4 * p, c0[p]); gstore<float>(o1 + 4 * p, c1[p]); gstore<float>(o2 + 4 * p, c2[p]); }
    }
  } else { // UD_RGB_F32 packed
    uint8_t* o = pd0 + (u32)(y * dp0) + (size_t)x0 * 12;
    if (n == 4 && (((uintptr_t)o) & 15u) == 0) {
      store16f(o + 0, make_float4(c0[0], c1[0], c2[0], c0[1]));
      store16f(o + 16, make_float4(c1[1], c2[1], c0[2], c1[2]));
      store16f(o + 32, make_float4(c2[2], c0[3], c1[3], c2[3]));
    } else {
      for (int p = 0; p < n; ++p) { gstore<float>(o + 12 * p, c0[p]); gstore<float>(o + 12 * p + 4, c1[p]); gstore<float>(o + 12 * p + 8, c2[p]); }
    }
  }
}

// Source spans (bytes, 16-byte granular) the 256 columns of tile `tile_x` read from a luma row
// and from a chroma row; the host uses it to choose the staged kernel.  The device derives the
// same numbers from the taps of lane 0's first and lane 63's last pixel (same float
// expressions, so the two always agree).
struct UdSpan {
  int yb, yn, cb, cn;
};
__host__ __device__ inline UdSpan ud_span_of(int ly0, int ly1, int lc0, int lc1, int E) {
  UdSpan r;
  r.yb = (ly0 * E) & ~15;
  r.yn = (((ly1 + 1) * E + 15) & ~15) - r.yb;
  r.cb = (lc0 * 2 * E) & ~15;
  r.cn = (((lc1 + 1) * 2 * E + 15) & ~15) - r.cb;
  return r;
}
template <typename T>
__host__ inline UdSpan ud_span(int tile_x, int dw, int sw, float scale_x) {
  const int xt0 = tile_x * 256, xt1 = (xt0 + 255 < dw - 1) ? xt0 + 255 : dw - 1;
  const float c0 = (float)xt0 / scale_x, c1 = (float)xt1 / scale_x;
  return ud_span_of(make_tap(c0, sw).i0, make_tap(c1, sw).i1, make_tap(c0 * 0.5f, sw / 2).i0,
                    make_tap(c1 * 0.5f, sw / 2).i1, (int)sizeof(T));
}

// ---- output of one dst row: plain, or rotated by quarter turns (NV12 -> packed RGB only) ----
// odd ROT: the workgroup's 256 x 32 output tile is collected in LDS, one dword per pixel
// (r | g << 8 | b << 16), and written transposed after a barrier (ud_rot_store).  258 dwords per
// tile row: the store phase's (8 columns) x (8 row groups) of a wave then hit 64 different banks
// (8 g + 2 j + column).
constexpr int kRotStride = 256 * 4 + 8;
constexpr int kRotTileBytes = kUdTileH * kRotStride;

// c0/c1/c2 as for ud_store; (wave, rr) = the row's place in the workgroup tile; dw x dh = size of
// the (virtual) un-rotated UD output.  STRIDED: the lane's pixel p is column lane + 64 p of the
// wave's 256 (else tile column tcol + p).
template <typename T, int OUT, int ROT, bool STRIDED>
__device__ __forceinline__ void ud_emit(const SurfRef& d, uint8_t* rot_tile, int wave, int lane, int rr,
                                        int x0, int y, int n, int dw, int dh, const float (&c0)[4],
                                        const float (&c1)[4], const float (&c2)[4], int tcol = 0) {
  if constexpr (ROT == 0) {
    ud_store<T, OUT>(d, x0, y, n, c0, c1, c2);
  } else if constexpr (ROT == 2) {
    // dst(uw-1-x, uh-1-y) = ud(x, y): the lane's 4 pixels in reverse order
    u32 w0, w1, w2;
    trunc_pack3x4(c0[3], c1[3], c2[3], c0[2], c1[2], c2[2], c0[1], c1[1], c2[1], c0[0], c1[0], c2[0], w0, w1, w2);
    uint8_t* row = d.p[0] + (u32)((dh - 1 - y) * d.pitch[0]);
    uint8_t* o = row + (ptrdiff_t)(dw - 4 - x0) * 3;
    if (n == 4) { // (any byte alignment: a width that is not a multiple of 4 mirrors the groups onto odd offsets)
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = {w0, w1, w2};
      gstore_u<v3u32>(o, w);
    } else {
      const u32 ww[3] = {w0, w1, w2};
      for (int p = 0; p < n; ++p) // pixel p sits at bytes 3 (3 - p) .. of the reversed group
        for (int b = 0; b < 3; ++b) {
          const int k = 3 * (3 - p) + b;
          gstore<uint8_t>(row + (ptrdiff_t)(dw - 1 - x0 - p) * 3 + b, (uint8_t)(ww[k >> 2] >> (8 * (k & 3))));
        }
    }
  } else {
    // row (wave, rr) of the workgroup tile, one dword per pixel; written transposed after the barrier
    u32* t = reinterpret_cast<u32*>(rot_tile + (wave * kUdRowsPerWave + rr) * kRotStride);
    u32 px[4];
    trunc_pack_px4(c0, c1, c2, px);
    if constexpr (STRIDED) {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        t[lane + kWave * p] = px[p];
    } else { // tcol = the tile column of pixel 0
#pragma unroll
      for (int p = 0; p < 4; ++p)
        t[tcol + p] = px[p];
    }
  }
}

// Transposed store of the 256 (x) x 32 (y) tile: destination row <-> tile column.  8 lanes x 4
// pixels (12 bytes each, one global_store_dwordx3) cover the 32 pixels of a destination row
// segment, 32 destination rows per pass.   ROT 1: dst(y, uw-1-x) = ud(x, y), pixels in rising y;
// ROT 3: dst(uh-1-y, x) = ud(x, y), pixels in falling y.  Call after __syncthreads().
template <int ROT>
__device__ __forceinline__ void ud_rot_store(const SurfRef& d, const uint8_t* rot_tile, u32 tile_x,
                                             u32 tile_y, int dw, int dh) {
  const int t = threadIdx.x, g = t & 7;
  const int yb = tile_y * kUdTileH;            // first UD row of the tile
  const int rows = min(kUdTileH, dh - yb);     // valid UD rows in the tile
#pragma unroll 1
  for (int pass = 0; pass < 256 / 32; ++pass) {
    const int cx = pass * 32 + (t >> 3);       // tile column
    const int x = tile_x * 256 + cx;           // UD column
    if (x >= dw)
      continue;
    // tile rows of this lane's 4 pixels, in destination order
    int tr[4];
    bool ok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      tr[j] = ROT == 1 ? 4 * g + j : kUdTileH - 1 - (4 * g + j);
      ok[j] = tr[j] < rows;
    }
    u32 px[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      px[j] = ok[j] ? *reinterpret_cast<const u32*>(rot_tile + tr[j] * kRotStride + cx * 4) : 0u;
    }
    // destination: row, and the x' of pixel j = 0
    const int drow = ROT == 1 ? dw - 1 - x : x;
    const int dx0 = ROT == 1 ? yb + 4 * g : dh - 1 - yb - (kUdTileH - 1 - 4 * g);
    uint8_t* o = d.p[0] + (u32)(drow * d.pitch[0]) + (ptrdiff_t)dx0 * 3;
    if (ok[0] && ok[1] && ok[2] && ok[3]) { // (any byte alignment: ROT 3 with a height that is not a multiple of 4)
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 w = {px[0] | (px[1] << 24), (px[1] >> 8) | (px[2] << 16), (px[2] >> 16) | (px[3] << 8)};
      gstore_u<v3u32>(o, w);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (ok[j]) {
          gstore<uint8_t>(o + 3 * j, (uint8_t)px[j]);
          gstore<uint8_t>(o + 3 * j + 1, (uint8_t)(px[j] >> 8));
          gstore<uint8_t>(o + 3 * j + 2, (uint8_t)(px[j] >> 16));
        }
    }
  }
}

// One workgroup = 256 x 32 dst pixels: wave w walks dst rows 8w..8w+7 of the tile with the
// SAME four column taps per lane, so the float divisions of the coordinates (IEEE, ~12
// instructions each) are paid once per 8 rows.  Instruction count is what bounds this kernel
// (profiles/r01_secondary.md), hence:
//  * the chroma coordinate x / (2 scale) is taken as 0.5f * (x / scale): identical bits
//    (power-of-two scaling commutes with rounding), half the divisions;
//  * the 8 row taps of a wave are evaluated lane-parallel (lane r computes row r) and read
//    back with v_readlane, so per row they are SGPRs and every row base address is scalar;
//  * STAGED: each dst row's two luma + two chroma source rows go through the wave's LDS strip,
//    lane l owning the same 16 bytes of every row; the loads of row r+1 are issued before row
//    r is sampled (register prefetch), so HBM latency overlaps the arithmetic.
//    !STAGED (source span wider than the strip, i.e. downscale beyond ~4x): direct byte gather.
//
// ROT != 0 (NV12 -> packed RGB only): the UD result is written rotated by ROT quarter turns,
// exactly what PySurfaceRotator's canonical 90 / 180 / 270 degree turn of the UD output gives
// (rotate.hip: ROT 1: dst(x', y') = ud(W-1-y', x'); 2: ud(W-1-x', H-1-y'); 3: ud(y', H-1-x')),
// without the intermediate surface (BASELINE config 4 as one pass: 18.7 instead of 31.1 MB).
// For odd ROT the workgroup collects its 256 x 32 output tile in LDS, one dword per pixel, and
// writes it transposed after a barrier (destination row <-> tile column): see the end of the kernel.
template <typename T, int OUT, bool STAGED, int ROT = 0, int MINW = 1>
// (MINW: round 1 ran the packed-RGB instantiation with __launch_bounds__(kBlock, 5) -- 96 VGPRs, 5 waves per SIMD, +5 %.
// With the truncating pack and the prefetch ring it needed 37 spilled registers, ran 1.5x slower than the plain one AND
// mis-rendered 4-pixel-wide outputs (tools/stress_ud.py): the instantiation is gone, VALI_TUNE_UD_OCC5 is a no-op.)
__global__ void __launch_bounds__(kBlock, MINW) k_ud_nv12(const UdArgs a) {
  static_assert(ROT == 0 || (sizeof(T) == 1 && OUT == UD_RGB_U8), "rotated output: NV12 -> RGB only");
  constexpr int CH = (int)sizeof(T);
  __shared__ UdStage<CH> stage[STAGED ? kWavesPerBlock : 1];
  u32 tile_x, tile_y, frame;
  if constexpr ((ROT & 1) != 0) {
    // transposed output: consecutive workgroups walk DOWN the UD image, i.e. along the
    // destination rows, so the 96-byte pieces of a destination line meet in L2
    u32 t;
    if (!frame_tile_of_block(a.map, frame, t))
      return;
    const u32 tiles_y = a.map.per_frame / a.map.tiles_x;
    tile_x = t / tiles_y;
    tile_y = t - tile_x * tiles_y;
  } else if (!tile_of_block(a.map, tile_x, tile_y, frame)) {
    return;
  }
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const uint8_t* py = s.p[0];
  const uint8_t* puv = s.p[1];
  const int sp_y = s.pitch[0], sp_uv = s.pitch[1], sw = s.width, sh = s.height;
  // size of the (virtual) un-rotated UD output
  const int dw = (ROT & 1) ? d.height : d.width, dh = (ROT & 1) ? d.width : d.height;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int x0 = (tile_x * 64 + lane) * 4;
  const int rpw = ROT == 0 ? a.rows : kUdRowsPerWave;             // rows of this wave (the row taps are still evaluated for 8)
  const int y_first = (tile_y * kWavesPerBlock + wave) * rpw;     // wave-uniform
  // odd ROT: the workgroup's output tile is collected in LDS and written transposed (ud_emit)
  __shared__ __attribute__((aligned(16))) uint8_t rot_tile[(ROT & 1) ? kRotTileBytes : 16];
  auto body = [&]() { // (a lambda so that its early exits still reach the transposed store below)
  if (y_first >= dh)
    return;

  // ResizeUtils.cu:135-136: scale = 1.0f * dst / src ; :36-37: coord = x / scale
  const float scale_x = 1.0f * (float)dw / (float)sw;
  const float scale_y = 1.0f * (float)dh / (float)sh;
  constexpr int E = (int)sizeof(T);

  // Transposed output (odd ROT): the lane SAMPLES pixels lane, lane+64, lane+128, lane+192 of the
  // wave's 256 -- at a 2x downscale neighbouring lanes then read LDS bytes 2 apart (two lanes per
  // dword, all banks distinct) instead of 8 apart (a 2-way bank conflict on every tap) -- which is
  // free here because the pixels go to the LDS tile one dword each anyway (4.83 -> 4.69 us).
  // For the plain output the same mapping plus an LDS exchange back to 4 neighbouring pixels per
  // lane measured no better (4.7-4.9 vs 4.65 us), so it keeps the adjacent mapping.
  constexpr bool kStrided = (ROT & 1) != 0;
  const int xw = tile_x * 256; // first pixel of the wave
  auto slot_x = [&](int p) { return kStrided ? xw + lane + kWave * p : x0 + p; };

  // column taps of this lane's 4 pixels (clamped to the last column for tail lanes)
  Tap tx[4], tcx[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = min(slot_x(p), dw - 1);
    const float cx = (float)x / scale_x;
    tx[p] = make_tap(cx, sw);
    tcx[p] = make_tap(cx * 0.5f, sw / 2); // == x / (scale_x * 2.0f)
  }
  // valid pixels of this lane, a prefix of its 4 slots (<= 0: tail lane, staging only)
  const int n = kStrided ? min(4, (dw - xw - lane + kWave - 1) / kWave) : min(4, dw - x0);
  // column weights replicated into both 16-bit lanes (8-bit sources, see sample())
  u32 pw[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    pw[p][0] = tx[p].w0 * 0x10001u; pw[p][1] = tx[p].w1 * 0x10001u;
    pw[p][2] = tcx[p].w0 * 0x10001u; pw[p][3] = tcx[p].w1 * 0x10001u;
  }

  // row taps: lane r evaluates row y_first + r; rows are read back as scalars
  struct RowTaps {
    Tap ty, tcy;
  };
  const float cyl = (float)(y_first + (lane & (kUdRowsPerWave - 1))) / scale_y;
  const Tap vty = make_tap(cyl, sh), vtcy = make_tap(cyl * 0.5f, sh / 2);
  auto row_taps = [&](int rr) {
    RowTaps r;
    r.ty.i0 = __builtin_amdgcn_readlane(vty.i0, rr);
    r.ty.i1 = __builtin_amdgcn_readlane(vty.i1, rr);
    r.ty.w0 = (u32)__builtin_amdgcn_readlane((int)vty.w0, rr);
    r.ty.w1 = (u32)__builtin_amdgcn_readlane((int)vty.w1, rr);
    r.tcy.i0 = __builtin_amdgcn_readlane(vtcy.i0, rr);
    r.tcy.i1 = __builtin_amdgcn_readlane(vtcy.i1, rr);
    r.tcy.w0 = (u32)__builtin_amdgcn_readlane((int)vtcy.w0, rr);
    r.tcy.w1 = (u32)__builtin_amdgcn_readlane((int)vtcy.w1, rr);
    return r;
  };

  // texels -> c0/c1/c2 (scaled by UdScale) of the lane's 4 pixels
  constexpr float kScale = UdScale<T, OUT>::value;
  constexpr float kNorm = TexelTraits<T>::kInvDen * kScale;
  // luma(r, p, t): texel of source row r (0/1), pixel p, horizontal tap t (0/1);
  // chroma(r, p, t): the interleaved pair, U in the low T-sized field, V in the high one.
  auto sample = [&](const RowTaps& rt, auto luma, auto chroma, float (&c0)[4], float (&c1)[4], float (&c2)[4]) {
    u32 sy[4], su[4], sv[4]; // the integer filter sums S (see the header of this file)
    if constexpr (E == 1) {
      // 8-bit texels: the horizontal stage fits 16 bits exactly (w0 q0 + w1 q1 <= 256 * 255),
      // so it runs two rows at a time in packed 16-bit lanes, and the vertical stage is one
      // 2-element dot product:
      //   A = {T[row0][i0], T[row1][i0]}  B = {T[row0][i1], T[row1][i1]}        (v_perm_b32)
      //   H = A * {w0,w0} + B * {w1,w1}           (v_pk_mul_lo_u16, v_pk_mad_u16) = {top, bot}
      //   S = H.x * wy0 + H.y * wy1                                         (v_dot2_u32_u16)
      // 5 VALU per component instead of 6 multiplies + byte extraction; same integers.
      typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
      auto as_v2 = [](u32 w) { return __builtin_bit_cast(v2u16, w); };
      const v2u16 wyl = as_v2(rt.ty.w0 | (rt.ty.w1 << 16)), wyc = as_v2(rt.tcy.w0 | (rt.tcy.w1 << 16));
      // two pixels at a time (reads first, then arithmetic): half the live texel registers
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32 l[2][2][2], c[2][2][2]; // [pixel][tap][row]
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            l[pp][t][0] = luma(0, 2 * h + pp, t); l[pp][t][1] = luma(1, 2 * h + pp, t);
            c[pp][t][0] = chroma(0, 2 * h + pp, t); c[pp][t][1] = chroma(1, 2 * h + pp, t);
          }
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
          const int p = 2 * h + pp;
          const v2u16 wx0 = as_v2(pw[p][0]), wx1 = as_v2(pw[p][1]), wc0 = as_v2(pw[p][2]), wc1 = as_v2(pw[p][3]);
          const v2u16 ya = as_v2(__builtin_amdgcn_perm(l[pp][0][1], l[pp][0][0], 0x0c040c00u));
          const v2u16 yb = as_v2(__builtin_amdgcn_perm(l[pp][1][1], l[pp][1][0], 0x0c040c00u));
          const v2u16 ua = as_v2(__builtin_amdgcn_perm(c[pp][0][1], c[pp][0][0], 0x0c040c00u));
          const v2u16 ub = as_v2(__builtin_amdgcn_perm(c[pp][1][1], c[pp][1][0], 0x0c040c00u));
          const v2u16 va = as_v2(__builtin_amdgcn_perm(c[pp][0][1], c[pp][0][0], 0x0c050c01u));
          const v2u16 vb = as_v2(__builtin_amdgcn_perm(c[pp][1][1], c[pp][1][0], 0x0c050c01u));
          sy[p] = __builtin_amdgcn_udot2(ya * wx0 + yb * wx1, wyl, 0u, false);
          su[p] = __builtin_amdgcn_udot2(ua * wc0 + ub * wc1, wyc, 0u, false);
          sv[p] = __builtin_amdgcn_udot2(va * wc0 + vb * wc1, wyc, 0u, false);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      u32 l[4][4], cu[4][4], cv[4][4]; // [pixel][00,10,01,11]
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        l[p][0] = luma(0, p, 0); l[p][1] = luma(0, p, 1);
        l[p][2] = luma(1, p, 0); l[p][3] = luma(1, p, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const u32 w = chroma(q >> 1, p, q & 1);
          cu[p][q] = w & 0xffffu; cv[p][q] = w >> 16;
        }
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        // every factor fits in 24 bits (weights <= 256, texels <= 65535, row sums < 2^24):
        // v_mul_u32_u24 / v_mad_u32_u24 are full rate, v_mul_lo_u32 is quarter rate
        auto bil = [](u32 wy0, u32 wy1, u32 wx0, u32 wx1, const u32 (&q)[4]) {
          const u32 top = __umul24(wx0, q[0]) + __umul24(wx1, q[1]);
          const u32 bot = __umul24(wx0, q[2]) + __umul24(wx1, q[3]);
          return __umul24(wy0, top) + __umul24(wy1, bot);
        };
        sy[p] = bil(rt.ty.w0, rt.ty.w1, tx[p].w0, tx[p].w1, l[p]);
        su[p] = bil(rt.tcy.w0, rt.tcy.w1, tcx[p].w0, tcx[p].w1, cu[p]);
        sv[p] = bil(rt.tcy.w0, rt.tcy.w1, tcx[p].w0, tcx[p].w1, cv[p]);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float ny = (float)sy[p] * kNorm;
      const float nu = (float)su[p] * kNorm;
      const float nv = (float)sv[p] * kNorm;
      if constexpr (OUT == UD_YUV444) {
        c0[p] = ny; c1[p] = nu; c2[p] = nv;
      } else {
        const float u = nu - 0.5f * kScale, v = nv - 0.5f * kScale;
        c0[p] = __builtin_fmaf(1.140f, v, ny);
        c1[p] = __builtin_fmaf(-0.581f, v, __builtin_fmaf(-0.394f, u, ny));
        c2[p] = __builtin_fmaf(2.032f, u, ny);
      }
    }
  };

  // ---- output: plain, or rotated by quarter turns ----
  auto emit = [&](int rr, int y, const float (&c0)[4], const float (&c1)[4], const float (&c2)[4]) {
    ud_emit<T, OUT, ROT, kStrided>(d, rot_tile, wave, lane, rr, x0, y, n, dw, dh, c0, c1, c2);
  };

  // direct byte gather: source span wider than the strip, or foreign memory that is not
  // 16-byte aligned
  auto gather_rows = [&]() {
    // (no early exit for lanes without pixels: row_taps() reads lanes 0..7 with v_readlane, and a
    // lane that has left holds whatever the compiler computed for it after the exit)
#pragma unroll 1
    for (int rr = 0; rr < rpw; ++rr) {
      const int y = y_first + rr;
      if (y >= dh)
        break;
      const RowTaps rt = row_taps(rr);
      if (n <= 0)
        continue;
      const uint8_t* yrow[2] = {py + (size_t)rt.ty.i0 * sp_y, py + (size_t)rt.ty.i1 * sp_y};
      const uint8_t* crow[2] = {puv + (size_t)rt.tcy.i0 * sp_uv, puv + (size_t)rt.tcy.i1 * sp_uv};
      float c0[4], c1[4], c2[4];
      // The two horizontal taps are the same texel (clamped edge) or neighbours: one unaligned load of
      // two luma texels / two chroma pairs per (row, pixel) -- 16 vector-memory instructions per lane
      // and row instead of 32 or 48 (this form is bound by the texture addresser: 2160p -> 960x540
      // 4.1 -> 2.x us).  A source narrower than 4 pixels has a single chroma pair per row: element loads.
      typedef uint16_t u16_unaligned __attribute__((aligned(1)));
      typedef u32 u32_unaligned __attribute__((aligned(1)));
      typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
      if (sw >= 4) {
        u32 lt[2][4][2], ct[2][4][2]; // [row][pixel][tap]
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int lb = min(tx[p].i0, sw - 2), cb = min(tcx[p].i0, sw / 2 - 2);
            u32 llo, lhi, clo, chi;
            if constexpr (E == 1) {
              const u32 w = *(const VALI_GLOBAL u16_unaligned*)(yrow[r] + lb);
              llo = w & 0xffu; lhi = w >> 8;
              const u32 c = *(const VALI_GLOBAL u32_unaligned*)(crow[r] + 2 * cb);
              clo = c & 0xffffu; chi = c >> 16;
            } else {
              const u32 w = *(const VALI_GLOBAL u32_unaligned*)(yrow[r] + 2 * lb);
              llo = w & 0xffffu; lhi = w >> 16;
              const unsigned long long c = *(const VALI_GLOBAL u64_unaligned*)(crow[r] + 4 * cb);
              clo = (u32)c; chi = (u32)(c >> 32);
            }
            lt[r][p][0] = tx[p].i0 == lb ? llo : lhi; lt[r][p][1] = tx[p].i1 == lb ? llo : lhi;
            ct[r][p][0] = tcx[p].i0 == cb ? clo : chi; ct[r][p][1] = tcx[p].i1 == cb ? clo : chi;
          }
        sample(rt, [&](int r, int p, int t) { return lt[r][p][t]; }, [&](int r, int p, int t) { return ct[r][p][t]; },
               c0, c1, c2);
      } else {
        sample(rt,
               [&](int r, int p, int t) { return (u32)gload<T>(yrow[r] + (size_t)(t ? tx[p].i1 : tx[p].i0) * E); },
               [&](int r, int p, int t) {
                 const uint8_t* q = crow[r] + (size_t)(t ? tcx[p].i1 : tcx[p].i0) * 2 * E;
                 return (u32)gload<T>(q) | ((u32)gload<T>(q + E) << (8 * E));
               },
               c0, c1, c2);
      }
      emit(rr, y, c0, c1, c2);
    }
  };

  if constexpr (STAGED) {
    const bool aligned = ((((uintptr_t)py) | ((uintptr_t)puv) | (uintptr_t)sp_y | (uintptr_t)sp_uv) & 15u) == 0;
    if (!aligned) {
      gather_rows();
      return;
    }
    // spans of this tile: first tap of lane 0 .. last tap of lane 63 (columns are monotonic)
    const UdSpan sp = ud_span_of(__builtin_amdgcn_readlane(tx[0].i0, 0), __builtin_amdgcn_readlane(tx[3].i1, 63),
                                 __builtin_amdgcn_readlane(tcx[0].i0, 0), __builtin_amdgcn_readlane(tcx[3].i1, 63), E);
    UdStage<CH>& st = stage[wave];
    // Loads are unconditional and straight-line (lanes past a span re-read its last 16 bytes),
    // so the compiler counts vmcnt instead of draining at branches.
    int off[CH], off_y[CH], off_c[CH];
    bool in_y[CH], in_c[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      off[c] = lane * 16 + c * kUdRowBytes;
      in_y[c] = off[c] < sp.yn; in_c[c] = off[c] < sp.cn;
      off_y[c] = min(off[c], sp.yn - 16); off_c[c] = min(off[c], sp.cn - 16);
    }
    // Prefetch ring, kUdDepth dst rows deep: per row this lane's chunks of luma0, luma1, chroma0, chroma1.  The walk
    // is unrolled by the depth so the ring is indexed statically, and EVERY trip issues the same loads in the same
    // order (rows past the tile's last one re-read it and skip the work): the compiler can then count the loads in
    // flight -- vmcnt retires in order -- instead of draining them at a branch.  With one row of prefetch the
    // kernel sat at ~50 % VALU and ~40 % LDS utilisation, waiting a memory latency per row.  (Plain loads: chroma rows
    // and, below 2x, luma rows are read by several dst rows out of L2 -- the non-temporal form measured 10-35 % slower;
    // the padded LDS rows of the bilinear resizer measured no different here.)
    constexpr int DEPTH = kUdDepth;
    uint4 pf[DEPTH][4][CH];
    const int last_rr = min(rpw, dh - y_first) - 1; // wave-uniform, >= 0
    auto issue = [&](int rr, uint4 (&q)[4][CH]) {
      const RowTaps rt = row_taps(min(rr, last_rr));
      // scalar address arithmetic; a plane is < 4 GiB, so 32-bit row offsets (s_mul_i32)
      const uint8_t* y0 = py + (u32)(rt.ty.i0 * sp_y + sp.yb);
      const uint8_t* y1 = py + (u32)(rt.ty.i1 * sp_y + sp.yb);
      const uint8_t* q0 = puv + (u32)(rt.tcy.i0 * sp_uv + sp.cb);
      const uint8_t* q1 = puv + (u32)(rt.tcy.i1 * sp_uv + sp.cb);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        q[0][c] = gload16(y0 + off_y[c]); q[1][c] = gload16(y1 + off_y[c]);
        q[2][c] = gload16(q0 + off_c[c]); q[3][c] = gload16(q1 + off_c[c]);
      }
    };
    auto commit = [&](const uint4 (&q)[4][CH]) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (in_y[c]) {
          *reinterpret_cast<uint4*>(&st.luma[0][off[c]]) = q[0][c];
          *reinterpret_cast<uint4*>(&st.luma[1][off[c]]) = q[1][c];
        }
        if (in_c[c]) {
          *reinterpret_cast<uint4*>(&st.chroma[0][off[c]]) = q[2][c];
          *reinterpret_cast<uint4*>(&st.chroma[1][off[c]]) = q[3][c];
        }
      }
    };
    // LDS byte offsets of the column taps (row-invariant)
    int ly[4][2], lc[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ly[p][0] = tx[p].i0 * E - sp.yb; ly[p][1] = tx[p].i1 * E - sp.yb;
      lc[p][0] = tcx[p].i0 * 2 * E - sp.cb; lc[p][1] = tcx[p].i1 * 2 * E - sp.cb;
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      issue(k, pf[k]);
      __builtin_amdgcn_sched_barrier(0); // rows stay in issue order
    }
#pragma unroll 1
    for (int r0 = 0; r0 <= last_rr; r0 += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; ++j) {
        const int rr = r0 + j;
        const bool live = rr <= last_rr; // wave-uniform
        if (live) {
          commit(pf[j]);
          wave_lds_sync();
        }
        issue(rr + DEPTH, pf[j]); // in flight while rows rr .. rr + DEPTH - 1 are sampled
        if (live) {
          if (n > 0) {
            const RowTaps cur = row_taps(rr);
            float c0[4], c1[4], c2[4];
            sample(cur,
                   [&](int r, int p, int t) { return (u32) * (const T*)(st.luma[r] + ly[p][t]); },
                   [&](int r, int p, int t) { // U and V are neighbours: one LDS read for the pair
                     if constexpr (E == 1)
                       return (u32) * (const uint16_t*)(st.chroma[r] + lc[p][t]);
                     else
                       return *(const u32*)(st.chroma[r] + lc[p][t]);
                   },
                   c0, c1, c2);
            emit(rr, y_first + rr, c0, c1, c2);
          }
          wave_lds_sync(); // the strip is re-filled by the next row
        }
      }
    }
  } else {
    gather_rows();
  }
  }; // body
  body();

  if constexpr ((ROT & 1) != 0) {
    __syncthreads();
    ud_rot_store<ROT>(d, rot_tile, tile_x, tile_y, dw, dh);
  }
}

// ---- exact 2x horizontal downscale of NV12 (src width == 2 x UD width): BASELINE config 4 ----
// scale_x = 0.5 exactly, so every column coordinate is X = 2x: xB = 2x - 0.5, i = 2x - 1,
// frac = 0.5 -> both horizontal weights are 128/256, for luma (columns 2x-1, 2x) and for chroma
// (coordinate x, pairs x-1, x); index -1 clamps to 0.  The filter sum of the general kernel,
//   S = sum_r wy_r (128 T[r][c0] + 128 T[r][c1]) = 128 (wy_0 h_0 + wy_1 h_1),  h_r = T[r][c0] + T[r][c1],
// is the same integer, and float(S) * kNorm == float(S / 128) * (128 kNorm) bit for bit, so the
// output is identical to k_ud_nv12's (and the oracle's).  The vertical taps stay general (any
// height).  What changes is the cost.  The general kernel is bound by the vector-memory front end
// (profiles/r01_ud_down2.md: the texture addresser takes ~16 cycles per wave instruction whatever
// its width, and 256 pixels cost it 4 loads + LDS traffic), so this one is built around the
// fewest, widest memory instructions: a lane owns 8 pixels = 16 consecutive luma bytes and 8
// chroma pairs per source row -- ONE dwordx4 per row -- the byte before them comes from the
// neighbouring lane (DPP wave shift), and a wave's only extra load is the dword before its first
// column (one instruction for all four rows).  No LDS, no divisions per column; the pair sums are
// v_dot4_u32_u8 against 0/1 byte masks.  A wave = 512 x 8 output pixels.
constexpr int kD2LanePx = 8;
constexpr int kD2WaveW = kWave * kD2LanePx;

__device__ __forceinline__ u32 wave_shr1(u32 v) { // lane l gets lane l-1's value (lane 0 keeps its own)
  return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// Row taps of one output row (wave-uniform in k_ud_lean, per half-wave in k_ud_down2_t)
struct D2Taps {
  Tap ty, tcy;
};
// Four pixels from three dwords per source row: P = the 4 bytes before A, then A, B
// (rows 0,1 = luma i0,i1 ; rows 2,3 = chroma i0,i1).
struct D2Quad {
  u32 p[4], a[4], b[4];
};
struct D2Src {
  const uint8_t* py;
  const uint8_t* puv;
  int sp_y, sp_uv, sw;
};
// byte gather with the texture clamp: tail lanes (their dwordx4 could pass the end of the
// row) and foreign memory that is not 16-byte aligned
__device__ __forceinline__ D2Quad d2_gather(const D2Src& s, const D2Taps& rt, int xh) {
  const uint8_t* py = s.py;
  const uint8_t* puv = s.puv;
  const int sp_y = s.sp_y, sp_uv = s.sp_uv, sw = s.sw;
  D2Quad r;
  const int boff = 2 * xh;
  const uint8_t* yr[2] = {py + (u32)(rt.ty.i0 * sp_y), py + (u32)(rt.ty.i1 * sp_y)};
  const uint8_t* cr[2] = {puv + (u32)(rt.tcy.i0 * sp_uv), puv + (u32)(rt.tcy.i1 * sp_uv)};
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    auto lb = [&](int c) { return (u32)gload<uint8_t>(yr[k] + min(max(c, 0), sw - 1)); };
    auto cb = [&](int q, int e) { return (u32)gload<uint8_t>(cr[k] + 2 * min(max(q, 0), sw / 2 - 1) + e); };
    r.p[k] = lb(boff - 1) << 24;
    r.a[k] = lb(boff) | (lb(boff + 1) << 8) | (lb(boff + 2) << 16) | (lb(boff + 3) << 24);
    r.b[k] = lb(boff + 4) | (lb(boff + 5) << 8) | (lb(boff + 6) << 16) | (lb(boff + 7) << 24);
    r.p[2 + k] = (cb(xh - 1, 0) << 16) | (cb(xh - 1, 1) << 24);
    r.a[2 + k] = cb(xh, 0) | (cb(xh, 1) << 8) | (cb(xh + 1, 0) << 16) | (cb(xh + 1, 1) << 24);
    r.b[2 + k] = cb(xh + 2, 0) | (cb(xh + 2, 1) << 8) | (cb(xh + 3, 0) << 16) | (cb(xh + 3, 1) << 24);
  }
  return r;
}

// P A B of four rows -> the three components (scaled by UdScale) of 4 pixels
// kEven: both vertical weights are 128 (luma and chroma; every row of an exact 2x vertical
// downscale): wy_0 h_0 + wy_1 h_1 = 128 (h_0 + h_1), the second row's dot product accumulates onto
// the first and the 128 moves into the normalisation constant (a power of two: same bits)
template <int OUT, bool kEven>
__device__ __forceinline__ void d2_compute(const D2Taps& rt, const D2Quad& r, float* c0, float* c1, float* c2) {
  using T = uint8_t;
  constexpr float kScale = UdScale<T, OUT>::value;
  constexpr float kNorm = TexelTraits<T>::kInvDen * kScale * 128.0f;
  u32 sy[4], su[4], sv[4];
  if constexpr (kEven) {
    u32 q[4], w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q[k] = __builtin_amdgcn_alignbyte(r.a[k], r.p[k], k < 2 ? 3 : 2);
      w[k] = __builtin_amdgcn_alignbyte(r.b[k], r.a[k], k < 2 ? 3 : 2);
    }
    auto two = [](u32 v0, u32 v1, u32 m) { return __builtin_amdgcn_udot4(v1, m, __builtin_amdgcn_udot4(v0, m, 0u, false), false); };
    sy[0] = two(q[0], q[1], 0x00000101u); sy[1] = two(q[0], q[1], 0x01010000u);
    sy[2] = two(w[0], w[1], 0x00000101u); sy[3] = two(w[0], w[1], 0x01010000u);
    const u32 v0[4] = {q[2], r.a[2], w[2], r.b[2]}, v1[4] = {q[3], r.a[3], w[3], r.b[3]};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      su[p] = two(v0[p], v1[p], 0x00010001u);
      sv[p] = two(v0[p], v1[p], 0x01000100u);
    }
  } else {
  {
    u32 h[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32 q = __builtin_amdgcn_alignbyte(r.a[k], r.p[k], 3); // cols 2xh-1 .. 2xh+2
      const u32 w = __builtin_amdgcn_alignbyte(r.b[k], r.a[k], 3); // cols 2xh+3 .. 2xh+6
      h[k][0] = __builtin_amdgcn_udot4(q, 0x00000101u, 0u, false);
      h[k][1] = __builtin_amdgcn_udot4(q, 0x01010000u, 0u, false);
      h[k][2] = __builtin_amdgcn_udot4(w, 0x00000101u, 0u, false);
      h[k][3] = __builtin_amdgcn_udot4(w, 0x01010000u, 0u, false);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
      sy[p] = __umul24(rt.ty.w0, h[0][p]) + __umul24(rt.ty.w1, h[1][p]);
  }
  {
    u32 hu[2][4], hv[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const u32 q = __builtin_amdgcn_alignbyte(r.a[2 + k], r.p[2 + k], 2); // pairs xh-1, xh
      const u32 w = __builtin_amdgcn_alignbyte(r.b[2 + k], r.a[2 + k], 2); // pairs xh+1, xh+2
      const u32 v[4] = {q, r.a[2 + k], w, r.b[2 + k]};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        hu[k][p] = __builtin_amdgcn_udot4(v[p], 0x00010001u, 0u, false);
        hv[k][p] = __builtin_amdgcn_udot4(v[p], 0x01000100u, 0u, false);
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      su[p] = __umul24(rt.tcy.w0, hu[0][p]) + __umul24(rt.tcy.w1, hu[1][p]);
      sv[p] = __umul24(rt.tcy.w0, hv[0][p]) + __umul24(rt.tcy.w1, hv[1][p]);
    }
  }
  }
  constexpr float kN = kEven ? kNorm * 128.0f : kNorm;
  // two pixels per instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: IEEE, same bits as
  // the scalar forms of k_ud_nv12)
  typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int p = 0; p < 4; p += 2) {
    const v2f kn = {kN, kN};
    const v2f ny = (v2f){(float)sy[p], (float)sy[p + 1]} * kn;
    const v2f nu = (v2f){(float)su[p], (float)su[p + 1]} * kn;
    const v2f nv = (v2f){(float)sv[p], (float)sv[p + 1]} * kn;
    if constexpr (OUT == UD_YUV444) {
      c0[p] = ny.x; c0[p + 1] = ny.y; c1[p] = nu.x; c1[p + 1] = nu.y; c2[p] = nv.x; c2[p + 1] = nv.y;
    } else {
      const v2f half = {0.5f * kScale, 0.5f * kScale};
      const v2f u = nu - half, v = nv - half;
      const v2f r = __builtin_elementwise_fma((v2f){1.140f, 1.140f}, v, ny);
      const v2f g = __builtin_elementwise_fma((v2f){-0.581f, -0.581f}, v,
                                              __builtin_elementwise_fma((v2f){-0.394f, -0.394f}, u, ny));
      const v2f bl = __builtin_elementwise_fma((v2f){2.032f, 2.032f}, u, ny);
      c0[p] = r.x; c0[p + 1] = r.y; c1[p] = g.x; c1[p + 1] = g.y; c2[p] = bl.x; c2[p + 1] = bl.y;
    }
  }
}

// ---- the same scheme at a 1:1 width ratio (src width == UD width: colour conversion with chroma
// interpolation, any height).  X = x: luma taps (x-1, x) with weights 128/128; chroma coordinate
// x/2: even x -> pairs (x/2-1, x/2) 128/128, odd x -> the single pair (x-1)/2 with weight 256
// (fraction 0) = 128 (T + T).  A lane's 8 pixels need 8 + 1 luma bytes and 4 + 1 chroma pairs per
// source row: one dwordx2 each plus the dword before it (p/a/b as in D2Quad, but a, b are the
// lane's own 8 bytes and the result is 8 pixels).
struct D1Oct {
  u32 p[4], a[4], b[4];
};
// P A B of four rows -> the three components (scaled by UdScale) of the lane's 8 pixels
template <int OUT, bool kEven>
__device__ __forceinline__ void d1_compute(const D2Taps& rt, const D1Oct& r, float* c0, float* c1, float* c2) {
  using T = uint8_t;
  constexpr float kScale = UdScale<T, OUT>::value;
  constexpr float kNorm = TexelTraits<T>::kInvDen * kScale * 128.0f;
  // operand (0: q = P|A shifted, 1: A, 2: w = A|B shifted, 3: B) and byte mask of pixel j's pair sum
  constexpr int kLumaOp[8] = {0, 0, 0, 1, 2, 2, 2, 3};
  constexpr u32 kLumaMask[8] = {0x00000101u, 0x00010100u, 0x01010000u, 0x01010000u,
                                0x00000101u, 0x00010100u, 0x01010000u, 0x01010000u};
  constexpr int kChromaOp[8] = {0, 1, 1, 1, 2, 3, 3, 3};
  constexpr u32 kChromaMaskU[8] = {0x00010001u, 0x00000002u, 0x00010001u, 0x00020000u,
                                   0x00010001u, 0x00000002u, 0x00010001u, 0x00020000u};
  u32 op[4][4]; // [row][operand]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int sh = k < 2 ? 3 : 2;
    op[k][0] = __builtin_amdgcn_alignbyte(r.a[k], r.p[k], sh);
    op[k][1] = r.a[k];
    op[k][2] = __builtin_amdgcn_alignbyte(r.b[k], r.a[k], sh);
    op[k][3] = r.b[k];
  }
  u32 sy[8], su[8], sv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if constexpr (kEven) {
      auto two = [](u32 v0, u32 v1, u32 m) { return __builtin_amdgcn_udot4(v1, m, __builtin_amdgcn_udot4(v0, m, 0u, false), false); };
      sy[j] = two(op[0][kLumaOp[j]], op[1][kLumaOp[j]], kLumaMask[j]);
      su[j] = two(op[2][kChromaOp[j]], op[3][kChromaOp[j]], kChromaMaskU[j]);
      sv[j] = two(op[2][kChromaOp[j]], op[3][kChromaOp[j]], kChromaMaskU[j] << 8);
    } else {
      auto one = [](u32 v, u32 m) { return __builtin_amdgcn_udot4(v, m, 0u, false); };
      sy[j] = __umul24(rt.ty.w0, one(op[0][kLumaOp[j]], kLumaMask[j])) + __umul24(rt.ty.w1, one(op[1][kLumaOp[j]], kLumaMask[j]));
      su[j] = __umul24(rt.tcy.w0, one(op[2][kChromaOp[j]], kChromaMaskU[j])) +
              __umul24(rt.tcy.w1, one(op[3][kChromaOp[j]], kChromaMaskU[j]));
      sv[j] = __umul24(rt.tcy.w0, one(op[2][kChromaOp[j]], kChromaMaskU[j] << 8)) +
              __umul24(rt.tcy.w1, one(op[3][kChromaOp[j]], kChromaMaskU[j] << 8));
    }
  }
  constexpr float kN = kEven ? kNorm * 128.0f : kNorm;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float ny = (float)sy[j] * kN, nu = (float)su[j] * kN, nv = (float)sv[j] * kN;
    if constexpr (OUT == UD_YUV444) {
      c0[j] = ny; c1[j] = nu; c2[j] = nv;
    } else {
      const float u = nu - 0.5f * kScale, v = nv - 0.5f * kScale;
      c0[j] = __builtin_fmaf(1.140f, v, ny);
      c1[j] = __builtin_fmaf(-0.581f, v, __builtin_fmaf(-0.394f, u, ny));
      c2[j] = __builtin_fmaf(2.032f, u, ny);
    }
  }
}

// ---- exactly 2:1 in BOTH directions (src = 2 dw x 2 dh), output width a multiple of 8: BASELINE config 4's UD ----
// Round 1's k_ud_down2 served every height, ragged widths, foreign alignment and the half turn from one body, and paid for
// it on the geometry that matters most: 667 instructions per wave and row in the hot loop, 77 of them v_readlane (16 needed:
// 75 spilled SGPRs come back through lanes), 39 skipped-over exec-mask branches of the byte-wise strip flush -- and the
// kernel is bound by the instructions it issues (73 M VALU + 26 M scalar per 64-frame launch = 75 % of its 221 us at
// one instruction per 4 cycles and SIMD; profiles/r03_ud_half.md).  Here every row's vertical weights are 128 / 128 by
// construction (rows 2y-1, 2y of luma, y-1, y of chroma: make_tap(2y) and make_tap(y)), so there are no row taps to
// broadcast and no second arithmetic form; every lane has 8 pixels or none; accesses are the unaligned-tolerant forms of
// the same instructions, so alignment needs no second path either.  Same bytes as the general kernel (d2_compute<OUT, true>).
template <int OUT>
__global__ void __launch_bounds__(kBlock) k_ud_half(const UdArgs a) {
  using T = uint8_t;
  constexpr bool kPacked = OUT == UD_RGB_U8;
  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int dw = d.width, dh = d.height;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xw = tile_x * kD2WaveW;
  const int y_first = (int)(tile_y * kWavesPerBlock + wave) * a.rows; // wave-uniform
  if (y_first >= dh)
    return;
  const int last = min(a.rows, dh - y_first) - 1;
  const int x0 = xw + lane * kD2LanePx;
  const bool has = x0 < dw;                                            // dw % 8 == 0: 8 pixels or none
  const u32 off16 = (u32)(2 * min(x0, dw - kD2LanePx));                // idle lanes re-read the row's last group
  const u32 offw = (u32)max(2 * xw - 4, 0);
  const u32 edge_shift = xw == 0 ? ((lane & 3) < 2 ? 24u : 16u) : 0u;  // lanes 0..3 fetch the dword before the wave's first byte in rows 0..3
  struct Rows {
    uint4 v[4]; // the lane's 16 bytes of luma 2y-1, luma 2y, chroma y-1, chroma y
    u32 before; // lane k < 4: the dword before the wave's first byte in row k
  };
  auto issue = [&](int rr) {
    const int y = y_first + min(rr, last);
    const uint8_t* row[4] = {s.p[0] + (u32)(max(2 * y - 1, 0) * s.pitch[0]), s.p[0] + (u32)(2 * y * s.pitch[0]),
                             s.p[1] + (u32)(max(y - 1, 0) * s.pitch[1]), s.p[1] + (u32)(y * s.pitch[1])};
    Rows r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k == 2)     // chroma y - 1 is the row before's chroma y: carried in registers (below), not read again -- the second
        continue;     // read came from HBM more often than not (FETCH_SIZE: 1.117 x the source, a third of the chroma plane)
      const v4u32 w = gload_u<v4u32>(row[k] + off16);
      r.v[k] = make_uint4(w.x, w.y, w.z, w.w);
    }
    r.v[2] = make_uint4(0u, 0u, 0u, 0u);
    const int lk = lane & 3;
    const uint8_t* rb = lk == 0 ? row[0] : lk == 1 ? row[1] : lk == 2 ? row[2] : row[3];
    r.before = gload_u<u32>(rb + offw) << edge_shift; // (the image's left edge: column -1 is column 0 -- the row's first byte / pair moves to the top)
    return r;
  };
  __shared__ __attribute__((aligned(16))) uint8_t strip[kPacked ? kWavesPerBlock : 1][kPacked ? kD2WaveW * 3 : 16];
  const int nbytes = min(kD2WaveW, dw - xw) * 3;                        // packed RGB bytes of the wave's row, a multiple of 24
  const bool dst16 = ((((uintptr_t)d.p[0]) | (uintptr_t)d.pitch[0]) & 15u) == 0 && (xw * 3 & 15) == 0; // wave-uniform
  auto step = [&](int rr, Rows rows, const uint4& chroma_above) {
    rows.v[2] = chroma_above;
    const int y = y_first + rr;
    u32 prev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // the 4 bytes before the lane's own: the previous lane's last dword; lane 0 (no lane to shift
      // from: it keeps `old`): the wave's extra load -- issue() has already turned it into the clamp at the image's left edge
      const u32 before = (u32)__builtin_amdgcn_readlane((int)rows.before, k);
      prev[k] = (u32)__builtin_amdgcn_update_dpp((int)before, (int)rows.v[k].w, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    }
    D2Quad q0, q1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      q0.p[k] = prev[k]; q0.a[k] = rows.v[k].x; q0.b[k] = rows.v[k].y;
      q1.p[k] = rows.v[k].y; q1.a[k] = rows.v[k].z; q1.b[k] = rows.v[k].w;
    }
    float c0[8], c1[8], c2[8];
    const D2Taps none = {};
    d2_compute<OUT, true>(none, q0, c0, c1, c2);
    d2_compute<OUT, true>(none, q1, c0 + 4, c1 + 4, c2 + 4);
    if constexpr (kPacked) {
      u32 w[6];
      trunc_pack3x4(c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2], c0[3], c1[3], c2[3], w[0], w[1], w[2]);
      trunc_pack3x4(c0[4], c1[4], c2[4], c0[5], c1[5], c2[5], c0[6], c1[6], c2[6], c0[7], c1[7], c2[7], w[3], w[4], w[5]);
      uint8_t* st = strip[kPacked ? wave : 0];
      if (has) {
        *reinterpret_cast<uint2*>(st + 24 * lane) = make_uint2(w[0], w[1]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 8) = make_uint2(w[2], w[3]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 16) = make_uint2(w[4], w[5]);
      }
      wave_lds_sync();
      uint8_t* orow = d.p[0] + (u32)(y * d.pitch[0]) + (u32)(xw * 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) { // strip byte b -> destination byte orow + b, 16 bytes per lane (nbytes: a multiple of 24, so
        const int b = (lane + kWave * h) * 16; // the last piece may be 8 bytes)
        if (b + 16 <= nbytes) {
          const uint4 v = *reinterpret_cast<const uint4*>(st + b);
          if (dst16)
            UD_ST16(orow + b, v);
          else
            gstore_u_nt<v4u32>(orow + b, (v4u32){v.x, v.y, v.z, v.w});
        } else if (b < nbytes) {
          const uint2 v = *reinterpret_cast<const uint2*>(st + b);
          gstore_u_nt<v2u32>(orow + b, (v2u32){v.x, v.y});
        }
      }
      wave_lds_sync(); // the strip is re-used by the next row
    } else if (has) {
      const int pp[3] = {d.pitch[0], OUT == UD_YUV444 ? d.pitch[1] : d.pitch[0], OUT == UD_YUV444 ? d.pitch[2] : d.pitch[0]};
      u32 pw[6]; // lo / hi dwords of the lane's 8 pixels in the three planes
      trunc_pack3x4(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7], c1[0], c1[1], c1[2], c1[3], pw[0], pw[1], pw[2]);
      trunc_pack3x4(c1[4], c1[5], c1[6], c1[7], c2[0], c2[1], c2[2], c2[3], c2[4], c2[5], c2[6], c2[7], pw[3], pw[4], pw[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        uint8_t* o = d.p[k] + (u32)(y * pp[k]) + (u32)x0;
        if ((((uintptr_t)d.p[k] | (uintptr_t)pp[k]) & 7u) == 0) // wave-uniform
          UD_ST8(o, make_uint2(pw[2 * k], pw[2 * k + 1]));
        else
          gstore_u_nt<v2u32>(o, (v2u32){pw[2 * k], pw[2 * k + 1]});
      }
    }
  };
  // two rows in flight, the walk unrolled by two so that both register sets are named statically
  uint4 above; // the chroma row above the next dst row's: loaded once per wave, then the row before's own chroma row
  {
    const v4u32 w = gload_u<v4u32>(s.p[1] + (u32)(max(y_first - 1, 0) * s.pitch[1]) + off16);
    above = make_uint4(w.x, w.y, w.z, w.w);
  }
  __builtin_amdgcn_sched_barrier(0);
  Rows ra = issue(0);
  __builtin_amdgcn_sched_barrier(0);
  Rows rb = issue(1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int rr = 0; rr <= last; rr += 2) {
    step(rr, ra, above);
    above = ra.v[3];
    ra = issue(rr + 2);
    if (rr + 1 <= last)
      step(rr + 1, rb, above);
    above = rb.v[3];
    rb = issue(rr + 3);
  }
}

// ---- the lean form at ANY height: source width = RATIO x UD width (RATIO 2: every other exact 2:1-wide geometry; RATIO 1:
// colour conversion with chroma interpolation at unchanged width), output width a multiple of 8, un-rotated ----
// Round 1's k_ud_down2 served these together with ragged widths, foreign alignment, the half turn and the byte-gather
// tails from one body and spilled 13-75 SGPRs in every instantiation.  This is k_ud_half's body
// with the row taps put back: lane rr of the wave evaluates the taps of its rr-th dst row (row offsets and the two
// weight pairs), a row's eight scalars arrive by v_readlane; every lane has 8 pixels or none; the unaligned-tolerant
// forms of the loads and stores make alignment a non-issue.  Ragged widths: the RAGGED instantiation below; the half turn
// at these ratios runs on the general kernel: correct and bit-identical, a little slower, and rare.
// EVEN: every row's vertical weights are (128, 128), (256, 0) or (0, 256) -- unchanged height (luma rows y - 1, y at
// 128 / 128; chroma 128 / 128 on even rows, the single row (y - 1) / 2 on odd ones) and exactly halved height.  A
// weight of 256 on one row is 128 on that row twice: the lane-parallel taps point both row slots at it, and what is
// left is k_ud_half's arithmetic -- the second row's dot product accumulates onto the first, no multiplications, no
// weights to broadcast (float(128 S') * c == float(S') * (128 c), a power of two: the same bits).
// RAGGED: output widths that are not multiples of 8 (>= 8) -- the row's last lane has fewer than 8 pixels; its window SLIDES left
// to end with the row (pixels dw - 8 .. dw - 1: a full group again; the pixels it shares with its neighbour are computed and
// stored twice with the same bytes), and since the bytes before that window are not its neighbour's, lanes 4..7 fetch them the
// way lanes 0..3 fetch the bytes before the wave's.  Its own instantiation: the aligned geometries pay nothing for it.
template <int OUT, int RATIO, bool EVEN, bool RAGGED = false>
__global__ void __launch_bounds__(kBlock) k_ud_lean(const UdArgs a) {
  using T = uint8_t;
  static_assert(RATIO == 1 || RATIO == 2, "source width = RATIO x UD width");
  constexpr bool kPacked = OUT == UD_RGB_U8;
  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int dw = d.width, dh = d.height, sh = s.height;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xw = tile_x * kD2WaveW;
  const int y_first = (int)(tile_y * kWavesPerBlock + wave) * a.rows; // wave-uniform
  if (y_first >= dh)
    return;
  const int last = min(a.rows, dh - y_first) - 1;
  const int x0 = xw + lane * kD2LanePx;
  const bool has = x0 < dw;                                            // 8 pixels or none (RAGGED: the row's last lane may have fewer)
  const int xs_w = dw - kD2LanePx;                                     // RAGGED: first pixel of the slid window (wave-uniform)
  const bool slid = RAGGED && has && x0 > xs_w;
  const int xl = slid ? xs_w : x0;                                     // first pixel of the lane's window
  const u32 offl = (u32)(RATIO * min(xl, dw - kD2LanePx));             // the lane's 8 RATIO bytes of a row; idle lanes re-read the last group
  // lanes 0..3 fetch the dword before the wave's first byte in rows 0..3 (RAGGED: lanes 4..7 the one before the slid window)
  const int xb = RAGGED && (lane & 4) ? xs_w : xw;
  const u32 offw = (u32)max(RATIO * xb - 4, 0);
  const u32 edge_shift = xb == 0 ? ((lane & 3) < 2 ? 24u : 16u) : 0u;
  // row taps, lane-parallel (lane rr: the wave's rr-th row; rows past the last repeat it)
  const float scale_y = 1.0f * (float)dh / (float)sh;                  // ResizeUtils.cu:136
  const float cyl = (float)(y_first + min(lane & (kUdRowsPerWave - 1), last)) / scale_y;
  Tap vty = make_tap(cyl, sh), vtcy = make_tap(cyl * 0.5f, sh / 2);
  if constexpr (EVEN) { // (host: the height is unchanged or halved) a row with the whole weight takes both slots
    if (vty.w1 == 0u) vty.i1 = vty.i0;
    if (vty.w0 == 0u) vty.i0 = vty.i1;
    if (vtcy.w1 == 0u) vtcy.i1 = vtcy.i0;
    if (vtcy.w0 == 0u) vtcy.i0 = vtcy.i1;
  }
  const u32 vro[4] = {(u32)(vty.i0 * s.pitch[0]), (u32)(vty.i1 * s.pitch[0]), (u32)(vtcy.i0 * s.pitch[1]), (u32)(vtcy.i1 * s.pitch[1])};
  struct Rows {
    uint4 v[4]; // the lane's bytes of luma i0, luma i1, chroma i0, chroma i1 (RATIO 1: x, y only)
    u32 before; // lane k < 4: the dword before the wave's first byte in row k
  };
  auto issue = [&](int rr) {
    const int q = min(rr, last);
    const uint8_t* row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      row[k] = (k < 2 ? s.p[0] : s.p[1]) + (u32)__builtin_amdgcn_readlane((int)vro[k], q);
    Rows r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (RATIO == 2) {
        const v4u32 w = gload_u<v4u32>(row[k] + offl);
        r.v[k] = make_uint4(w.x, w.y, w.z, w.w);
      } else {
        const v2u32 w = gload_u<v2u32>(row[k] + offl);
        r.v[k] = make_uint4(w.x, w.y, 0u, 0u);
      }
    }
    const int lk = lane & 3;
    const uint8_t* rb = lk == 0 ? row[0] : lk == 1 ? row[1] : lk == 2 ? row[2] : row[3];
    r.before = gload_u<u32>(rb + offw) << edge_shift; // (the image's left edge: column -1 is column 0)
    return r;
  };
  __shared__ __attribute__((aligned(16))) uint8_t strip[kPacked ? kWavesPerBlock : 1][kPacked ? kD2WaveW * 3 : 16];
  const int nbytes = (min(kD2WaveW, dw - xw) / kD2LanePx) * (kD2LanePx * 3); // packed RGB bytes of the wave's FULL lanes: a multiple of 24
  const bool dst16 = ((((uintptr_t)d.p[0]) | (uintptr_t)d.pitch[0]) & 15u) == 0 && (xw * 3 & 15) == 0; // wave-uniform
  auto step = [&](int rr, const Rows& rows) {
    const int y = y_first + rr;
    D2Taps rt = {};                                   // (the rows are loaded: only the weights are used from here on)
    if constexpr (!EVEN) {
      rt.ty.w0 = (u32)__builtin_amdgcn_readlane((int)vty.w0, rr);
      rt.ty.w1 = (u32)__builtin_amdgcn_readlane((int)vty.w1, rr);
      rt.tcy.w0 = (u32)__builtin_amdgcn_readlane((int)vtcy.w0, rr);
      rt.tcy.w1 = (u32)__builtin_amdgcn_readlane((int)vtcy.w1, rr);
    }
    u32 prev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { // the 4 bytes before the lane's own: the previous lane's last dword; lane 0 (no lane to shift
      // from: it keeps `old`): the wave's extra load -- issue() has already turned it into the clamp at the image's left edge
      const u32 before = (u32)__builtin_amdgcn_readlane((int)rows.before, k);
      prev[k] = (u32)__builtin_amdgcn_update_dpp((int)before, (int)(RATIO == 2 ? rows.v[k].w : rows.v[k].y), 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
      if constexpr (RAGGED) {
        const u32 before_s = (u32)__builtin_amdgcn_readlane((int)rows.before, 4 + k);
        prev[k] = slid ? before_s : prev[k];
      }
    }
    float c0[8], c1[8], c2[8];
    if constexpr (RATIO == 2) {
      D2Quad q0, q1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        q0.p[k] = prev[k]; q0.a[k] = rows.v[k].x; q0.b[k] = rows.v[k].y;
        q1.p[k] = rows.v[k].y; q1.a[k] = rows.v[k].z; q1.b[k] = rows.v[k].w;
      }
      d2_compute<OUT, EVEN>(rt, q0, c0, c1, c2);
      d2_compute<OUT, EVEN>(rt, q1, c0 + 4, c1 + 4, c2 + 4);
    } else {
      D1Oct o;
#pragma unroll
      for (int k = 0; k < 4; ++k) { o.p[k] = prev[k]; o.a[k] = rows.v[k].x; o.b[k] = rows.v[k].y; }
      d1_compute<OUT, EVEN>(rt, o, c0, c1, c2);
    }
    if constexpr (kPacked) {
      u32 w[6];
      trunc_pack3x4(c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2], c0[3], c1[3], c2[3], w[0], w[1], w[2]);
      trunc_pack3x4(c0[4], c1[4], c2[4], c0[5], c1[5], c2[5], c0[6], c1[6], c2[6], c0[7], c1[7], c2[7], w[3], w[4], w[5]);
      uint8_t* st = strip[kPacked ? wave : 0];
      if (slid) { // its 24 bytes are not on the strip's lane grid: straight to memory
        uint8_t* o = d.p[0] + (u32)(y * d.pitch[0]) + (u32)(xl * 3);
        gstore_u_nt<v2u32>(o, (v2u32){w[0], w[1]});
        gstore_u_nt<v2u32>(o + 8, (v2u32){w[2], w[3]});
        gstore_u_nt<v2u32>(o + 16, (v2u32){w[4], w[5]});
      } else if (has) {
        *reinterpret_cast<uint2*>(st + 24 * lane) = make_uint2(w[0], w[1]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 8) = make_uint2(w[2], w[3]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 16) = make_uint2(w[4], w[5]);
      }
      wave_lds_sync();
      uint8_t* orow = d.p[0] + (u32)(y * d.pitch[0]) + (u32)(xw * 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) { // strip byte b -> destination byte orow + b, 16 bytes per lane (nbytes: a multiple of 24, so
        const int b = (lane + kWave * h) * 16; // the last piece may be 8 bytes)
        if (b + 16 <= nbytes) {
          const uint4 v = *reinterpret_cast<const uint4*>(st + b);
          if (dst16)
            UD_ST16(orow + b, v);
          else
            gstore_u_nt<v4u32>(orow + b, (v4u32){v.x, v.y, v.z, v.w});
        } else if (b < nbytes) {
          const uint2 v = *reinterpret_cast<const uint2*>(st + b);
          gstore_u_nt<v2u32>(orow + b, (v2u32){v.x, v.y});
        }
      }
      wave_lds_sync(); // the strip is re-used by the next row
    } else if (has) {
      const int pp[3] = {d.pitch[0], OUT == UD_YUV444 ? d.pitch[1] : d.pitch[0], OUT == UD_YUV444 ? d.pitch[2] : d.pitch[0]};
      u32 pw[6]; // lo / hi dwords of the lane's 8 pixels in the three planes
      trunc_pack3x4(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7], c1[0], c1[1], c1[2], c1[3], pw[0], pw[1], pw[2]);
      trunc_pack3x4(c1[4], c1[5], c1[6], c1[7], c2[0], c2[1], c2[2], c2[3], c2[4], c2[5], c2[6], c2[7], pw[3], pw[4], pw[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        uint8_t* o = d.p[k] + (u32)(y * pp[k]) + (u32)xl;
        if (!slid && (((uintptr_t)d.p[k] | (uintptr_t)pp[k]) & 7u) == 0) // (wave-uniform but for the slid lane)
          UD_ST8(o, make_uint2(pw[2 * k], pw[2 * k + 1]));
        else
          gstore_u_nt<v2u32>(o, (v2u32){pw[2 * k], pw[2 * k + 1]});
      }
    }
  };
  // two rows in flight, the walk unrolled by two so that both register sets are named statically
  Rows ra = issue(0);
  __builtin_amdgcn_sched_barrier(0);
  Rows rb = issue(1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int rr = 0; rr <= last; rr += 2) {
    step(rr, ra);
    ra = issue(rr + 2);
    if (rr + 1 <= last)
      step(rr + 1, rb);
    rb = issue(rr + 3);
  }
}

// ---- exactly 3:2 in both directions (1080p -> 720p, 2160p -> 1440p: THE pre-processing / transcode ratio) ----
// X = 1.5 x: even x sample between two texels (weights 128 / 128), odd x ON a texel (256 / 0); the chroma coordinate
// 0.75 x walks through the fractions .5 / .25 / 0 / .75, i.e. weights (128,128) (192,64) (256,0) (64,192); the same vertically
// with the period of four rows.  The integer filter sum of the general kernel, S = sum_r wy_r (wx_0 T[r][i] + wx_1 T[r][i+1]),
// is therefore a sum of texels with small constant weights -- 128 * 128 * (a_r b_t) for luma with a, b in {1,1} / {2},
// 64 * 64 * (a_r b_t) for chroma with a, b in {2,2} {3,1} {4,0} {1,3} -- which v_dot4_u32_u8 evaluates against constant byte
// masks: one or two dot products per component and pixel (the general kernel: 15 vector instructions), no LDS, no column
// taps, no row taps.  A lane owns 8 output pixels = 12 luma bytes and 6 chroma pairs per source row (one dwordx3 each), the
// bytes before them come from the neighbouring lane (DPP).  Same bits as k_ud_nv12 / the oracle: float(2^k S') * c ==
// float(S') * (2^k c); where the reference's float coordinate lands a hair below an integer the tap pair (i-1, i) carries
// the weights (0, 256), which is the same sample.  Source = 1.5 x the output both ways, width % 8 == 0, height % 4 == 0.
template <int OUT, int M> struct Ud32Masks {                       // M = y & 3
  static constexpr u32 ay0 = (M & 1) ? 2u : 1u, ay1 = (M & 1) ? 0u : 1u;                    // luma row weights / 128
  static constexpr u32 ac0 = M == 0 ? 2u : M == 1 ? 3u : M == 2 ? 4u : 1u, ac1 = 4u - ac0;  // chroma row weights / 64
};
template <int OUT, int M>
__device__ __forceinline__ void ud32_compute(const u32 (&P)[4], const uint4 (&R)[4], float (&c0)[8], float (&c1)[8], float (&c2)[8]) {
  using T = uint8_t;
  using W = Ud32Masks<OUT, M>;
  // rows 0, 1 = luma i0, i1 ; rows 2, 3 = chroma i0, i1 ; R[k].x/.y/.z = the lane's 12 bytes, P[k] = the 4 bytes before them
  u32 lq[2], cq0[2], cq2[2], cq6[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    lq[r] = __builtin_amdgcn_alignbyte(R[r].x, P[r], 3);              // luma bytes -1 .. 2
    cq0[r] = __builtin_amdgcn_alignbyte(R[2 + r].x, P[2 + r], 2);      // chroma bytes -2 .. 1 (pairs 6m-1, 6m)
    cq2[r] = __builtin_amdgcn_alignbyte(R[2 + r].y, R[2 + r].x, 2);    // bytes 2 .. 5
    cq6[r] = __builtin_amdgcn_alignbyte(R[2 + r].z, R[2 + r].y, 2);    // bytes 6 .. 9
  }
  constexpr u32 kLuma[8] = {0x00000101u, 0x00000200u, 0x01010000u, 0x00000002u, 0x00010100u, 0x02000000u, 0x00000101u, 0x00020000u};
  constexpr u32 kChroma[4] = {0x00020002u, 0x00010003u, 0x00000004u, 0x00030001u};   // (b0, b1) of x & 3 on the U bytes 0, 2
  u32 sy[8], su[8], sv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const u32 lo[2] = {j == 0 ? lq[0] : j < 3 ? R[0].x : j < 6 ? R[0].y : R[0].z, j == 0 ? lq[1] : j < 3 ? R[1].x : j < 6 ? R[1].y : R[1].z};
    sy[j] = __builtin_amdgcn_udot4(lo[0], kLuma[j] * W::ay0, 0u, false);
    if constexpr (W::ay1 != 0)
      sy[j] = __builtin_amdgcn_udot4(lo[1], kLuma[j] * W::ay1, sy[j], false);
    const u32 co[2] = {j == 0 ? cq0[0] : j == 1 ? R[2].x : j < 4 ? cq2[0] : j == 4 ? R[2].y : j == 5 ? cq6[0] : R[2].z,
                       j == 0 ? cq0[1] : j == 1 ? R[3].x : j < 4 ? cq2[1] : j == 4 ? R[3].y : j == 5 ? cq6[1] : R[3].z};
    su[j] = __builtin_amdgcn_udot4(co[0], kChroma[j & 3] * W::ac0, 0u, false);
    sv[j] = __builtin_amdgcn_udot4(co[0], (kChroma[j & 3] * W::ac0) << 8, 0u, false);
    if constexpr (W::ac1 != 0) {
      su[j] = __builtin_amdgcn_udot4(co[1], kChroma[j & 3] * W::ac1, su[j], false);
      sv[j] = __builtin_amdgcn_udot4(co[1], (kChroma[j & 3] * W::ac1) << 8, sv[j], false);
    }
  }
  constexpr float kScale = UdScale<T, OUT>::value;
  constexpr float kNl = TexelTraits<T>::kInvDen * kScale * 16384.0f, kNc = TexelTraits<T>::kInvDen * kScale * 4096.0f;
  typedef float v2f __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int p = 0; p < 8; p += 2) {
    const v2f ny = (v2f){(float)sy[p], (float)sy[p + 1]} * (v2f){kNl, kNl};
    const v2f nu = (v2f){(float)su[p], (float)su[p + 1]} * (v2f){kNc, kNc};
    const v2f nv = (v2f){(float)sv[p], (float)sv[p + 1]} * (v2f){kNc, kNc};
    if constexpr (OUT == UD_YUV444) {
      c0[p] = ny.x; c0[p + 1] = ny.y; c1[p] = nu.x; c1[p + 1] = nu.y; c2[p] = nv.x; c2[p + 1] = nv.y;
    } else {
      const v2f half = {0.5f * kScale, 0.5f * kScale};
      const v2f u = nu - half, v = nv - half;
      const v2f r = __builtin_elementwise_fma((v2f){1.140f, 1.140f}, v, ny);
      const v2f g = __builtin_elementwise_fma((v2f){-0.581f, -0.581f}, v, __builtin_elementwise_fma((v2f){-0.394f, -0.394f}, u, ny));
      const v2f bl = __builtin_elementwise_fma((v2f){2.032f, 2.032f}, u, ny);
      c0[p] = r.x; c0[p + 1] = r.y; c1[p] = g.x; c1[p + 1] = g.y; c2[p] = bl.x; c2[p + 1] = bl.y;
    }
  }
}

template <int OUT>
__global__ void __launch_bounds__(kBlock) k_ud_32(const UdArgs a) {
  constexpr bool kPacked = OUT == UD_RGB_U8;
  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int dw = d.width, dh = d.height, sh = s.height;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int xw = tile_x * kD2WaveW;
  const int y_first = (int)(tile_y * kWavesPerBlock + wave) * a.rows; // wave-uniform
  if (y_first >= dh)
    return;
  const int last = min(a.rows, dh - y_first) - 1;
  const int x0 = xw + lane * kD2LanePx;
  const bool has = x0 < dw;                                            // dw % 8 == 0: 8 pixels or none
  const u32 off12 = (u32)(3 * (min(x0, dw - kD2LanePx) >> 1));         // 1.5 x0: idle lanes re-read the row's last group
  const u32 offw = (u32)max(3 * (xw >> 1) - 4, 0);
  typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
  struct Rows {
    uint4 v[4]; // .x .y .z = the lane's 12 bytes of luma i0, luma i1, chroma i0, chroma i1
    u32 before; // lane k < 4: the dword before the wave's first byte in row k
  };
  // byte offsets of the four source rows of every dst row of the wave, lane-parallel (lane r = row y_first + r) and read
  // back as scalars: the clamps, shifts and multiplies cost a wave ~25 scalar instructions per row otherwise -- and this
  // kernel, like its siblings, is bound by the instructions it issues.  make_tap clamps BOTH taps: for y = 0 the pair
  // (-1, 0) becomes (0, 0).
  u32 vro[4];
  {
    const int y = y_first + min(lane & 31, last);
    const int l0 = (3 * y - 1) >> 1, c0r = (3 * y - 2) >> 2;       // floor(1.5 y - 0.5), floor(0.75 y - 0.5): -1 for y = 0
    vro[0] = (u32)(max(l0, 0) * s.pitch[0]);
    vro[1] = (u32)(min(max(l0 + 1, 0), sh - 1) * s.pitch[0]);
    vro[2] = (u32)(max(c0r, 0) * s.pitch[1]);
    vro[3] = (u32)(min(max(c0r + 1, 0), sh / 2 - 1) * s.pitch[1]);
  }
  auto issue = [&](int rr) {
    const int r = min(rr, last);
    const uint8_t* row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      row[k] = (k < 2 ? s.p[0] : s.p[1]) + (u32)__builtin_amdgcn_readlane((int)vro[k], r);
    Rows rws;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const v3u32 w = gload_u<v3u32>(row[k] + off12);
      rws.v[k] = make_uint4(w.x, w.y, w.z, 0u);
    }
    const int lk = lane & 3;
    const uint8_t* rb = lk == 0 ? row[0] : lk == 1 ? row[1] : lk == 2 ? row[2] : row[3];
    rws.before = gload_u<u32>(rb + offw);
    return rws;
  };
  __shared__ __attribute__((aligned(16))) uint8_t strip[kPacked ? kWavesPerBlock : 1][kPacked ? kD2WaveW * 3 : 16];
  const int nbytes = min(kD2WaveW, dw - xw) * 3;
  const bool dst16 = ((((uintptr_t)d.p[0]) | (uintptr_t)d.pitch[0]) & 15u) == 0 && (xw * 3 & 15) == 0; // wave-uniform
  auto step = [&](int rr, const Rows& rows) {
    const int y = y_first + rr;
    u32 prev[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 before = (u32)__builtin_amdgcn_readlane((int)rows.before, k);
      const u32 edge = k < 2 ? rows.v[k].x << 24 : rows.v[k].x << 16;
      const u32 sh1 = wave_shr1(rows.v[k].z);
      prev[k] = lane == 0 ? (xw == 0 ? edge : before) : sh1;
    }
    float c0[8], c1[8], c2[8];
    switch (y & 3) { // wave-uniform
    case 0: ud32_compute<OUT, 0>(prev, rows.v, c0, c1, c2); break;
    case 1: ud32_compute<OUT, 1>(prev, rows.v, c0, c1, c2); break;
    case 2: ud32_compute<OUT, 2>(prev, rows.v, c0, c1, c2); break;
    default: ud32_compute<OUT, 3>(prev, rows.v, c0, c1, c2); break;
    }
    if constexpr (kPacked) {
      u32 w[6];
      trunc_pack3x4(c0[0], c1[0], c2[0], c0[1], c1[1], c2[1], c0[2], c1[2], c2[2], c0[3], c1[3], c2[3], w[0], w[1], w[2]);
      trunc_pack3x4(c0[4], c1[4], c2[4], c0[5], c1[5], c2[5], c0[6], c1[6], c2[6], c0[7], c1[7], c2[7], w[3], w[4], w[5]);
      uint8_t* st = strip[kPacked ? wave : 0];
      if (has) {
        *reinterpret_cast<uint2*>(st + 24 * lane) = make_uint2(w[0], w[1]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 8) = make_uint2(w[2], w[3]);
        *reinterpret_cast<uint2*>(st + 24 * lane + 16) = make_uint2(w[4], w[5]);
      }
      wave_lds_sync();
      uint8_t* orow = d.p[0] + (u32)(y * d.pitch[0]) + (u32)(xw * 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int b = (lane + kWave * h) * 16;
        if (b + 16 <= nbytes) {
          const uint4 v = *reinterpret_cast<const uint4*>(st + b);
          if (dst16)
            UD_ST16(orow + b, v);
          else
            gstore_u_nt<v4u32>(orow + b, (v4u32){v.x, v.y, v.z, v.w});
        } else if (b < nbytes) {
          const uint2 v = *reinterpret_cast<const uint2*>(st + b);
          gstore_u_nt<v2u32>(orow + b, (v2u32){v.x, v.y});
        }
      }
      wave_lds_sync();
    } else if (has) {
      const int pp[3] = {d.pitch[0], OUT == UD_YUV444 ? d.pitch[1] : d.pitch[0], OUT == UD_YUV444 ? d.pitch[2] : d.pitch[0]};
      u32 pw[6];
      trunc_pack3x4(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5], c0[6], c0[7], c1[0], c1[1], c1[2], c1[3], pw[0], pw[1], pw[2]);
      trunc_pack3x4(c1[4], c1[5], c1[6], c1[7], c2[0], c2[1], c2[2], c2[3], c2[4], c2[5], c2[6], c2[7], pw[3], pw[4], pw[5]);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        uint8_t* o = d.p[k] + (u32)(y * pp[k]) + (u32)x0;
        if ((((uintptr_t)d.p[k] | (uintptr_t)pp[k]) & 7u) == 0) // wave-uniform
          UD_ST8(o, make_uint2(pw[2 * k], pw[2 * k + 1]));
        else
          gstore_u_nt<v2u32>(o, (v2u32){pw[2 * k], pw[2 * k + 1]});
      }
    }
  };
  Rows ra = issue(0);
  __builtin_amdgcn_sched_barrier(0);
  Rows rb = issue(1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
  for (int rr = 0; rr <= last; rr += 2) {
    step(rr, ra);
    ra = issue(rr + 2);
    if (rr + 1 <= last)
      step(rr + 1, rb);
    rb = issue(rr + 3);
  }
}

// ---- the same lean form with the output written turned by 90 / 270 degrees: BASELINE config 4 as ONE pass ----
// k_ud_down2_t collects 256 x 32 tiles, so a destination row receives 32 pixels = 96 bytes per tile: three quarters of a
// line, 1.13x the written bytes at the memory (profiles/r02_secondary_traffic.md).  Here a workgroup owns 64 columns x 128
// (or 64: the default, see the launch) rows of the (virtual) un-rotated output: a destination row then gets 128 pixels =
// 384 bytes = three whole lines per tile (64: 192 bytes, completed by the next tile of the same XCD).  Compute phase: 8 lanes x 8 pixels cover a row of the tile (one 128-byte line of luma per source row), 8 rows per
// wave instruction, each wave walks 32 (16) rows in 4 (2) steps; pixels go to LDS as one dword each (row stride 65 dwords: the
// 8 x 8 lanes of a step and the 16 x 4 lanes of a store hit all banks twice at most).  Store phase: 16 lanes x 4 pixels
// (12 bytes, one dwordx3) walk half a destination segment for 4 neighbouring destination rows.
template <int ROT, int TH>
__global__ void __launch_bounds__(kBlock) k_ud_half_t(const UdArgs a) {
  static_assert(ROT == 1 || ROT == 3, "quarter turns only");
  constexpr int TW = 64, SD = TW + 1, RW = TH / kWavesPerBlock, STEPS = RW / 8; // rows per wave, 8-row steps per wave
  static_assert(TH == 64 || TH == 128, "tile rows");
  __shared__ u32 tile[TH * SD];
  u32 tile_x, tile_y, frame;
  {
    u32 t;
    if (!frame_tile_of_block(a.map, frame, t))
      return;
    const u32 tiles_y = a.map.per_frame / a.map.tiles_x; // consecutive workgroups walk DOWN the UD image: neighbouring
    tile_x = t / tiles_y;                                // segments of the same destination rows
    tile_y = t - tile_x * tiles_y;
  }
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int dw = d.height, dh = d.width; // size of the (virtual) un-rotated UD output; dw % 8 == 0, dh % 4 == 0 (host)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = lane >> 3, c = lane & 7;
  // (270 degrees: UD row y lands in dst column uh - 1 - y -- the tiles are anchored at the LAST row, the ragged tile is the first
  // one, so that a tile's dst segments start at multiples of TH pixels = whole 64-byte sectors as at 90 degrees: rotate.hip)
  const int xw = tile_x * TW, yb = (int)tile_y * TH - (ROT == 3 ? (TH - dh % TH) % TH : 0);
  const int x0 = xw + 8 * c;
  const bool has = x0 < dw;
  const u32 off16 = (u32)(2 * min(x0, dw - kD2LanePx));
  const u32 offb = (u32)max(2 * min(x0, dw - kD2LanePx) - 4, 0);   // the dword before the lane's own bytes
  struct Rows {
    uint4 v[4];     // the lane's 16 bytes of luma 2y-1, luma 2y, chroma y-1, chroma y
    u32 before[4];  // the dword in front of them
  };
  auto issue = [&](int step) {
    const int y = max(min(yb + wave * RW + min(step, STEPS - 1) * 8 + g, dh - 1), 0);
    const u32 ro[4] = {(u32)(max(2 * y - 1, 0) * s.pitch[0]), (u32)(2 * y * s.pitch[0]),
                       (u32)(max(y - 1, 0) * s.pitch[1]), (u32)(y * s.pitch[1])};
    Rows r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint8_t* row = (k < 2 ? s.p[0] : s.p[1]) + ro[k];
      const v4u32 w = gload_u<v4u32>(row + off16);
      r.v[k] = make_uint4(w.x, w.y, w.z, w.w);
      r.before[k] = gload_u<u32>(row + offb);
    }
    return r;
  };
  auto compute = [&](int step, const Rows& r) {
    const int rr = wave * RW + step * 8 + g;
    D2Quad q0, q1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 edge = k < 2 ? r.v[k].x << 24 : r.v[k].x << 16;       // column -1 = column 0 at the image's left edge
      q0.p[k] = x0 == 0 ? edge : r.before[k]; q0.a[k] = r.v[k].x; q0.b[k] = r.v[k].y;
      q1.p[k] = r.v[k].y; q1.a[k] = r.v[k].z; q1.b[k] = r.v[k].w;
    }
    float c0[8], c1[8], c2[8];
    const D2Taps none = {};
    d2_compute<UD_RGB_U8, true>(none, q0, c0, c1, c2);
    d2_compute<UD_RGB_U8, true>(none, q1, c0 + 4, c1 + 4, c2 + 4);
    u32 px[8];
    trunc_pack_px4(c0, c1, c2, px);
    trunc_pack_px4(c0 + 4, c1 + 4, c2 + 4, px + 4);
    if (has && yb + rr < dh && yb + rr >= 0) {
      u32* t = tile + rr * SD + 8 * c;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        t[j] = px[j];
    }
  };
  // two steps in flight, both register sets named statically
  Rows ra = issue(0);
  __builtin_amdgcn_sched_barrier(0);
  Rows rb = issue(1);
  __builtin_amdgcn_sched_barrier(0);
  compute(0, ra);
  if constexpr (STEPS == 4) {
    ra = issue(2);
    compute(1, rb);
    rb = issue(3);
    compute(2, ra);
    compute(3, rb);
  } else {
    compute(1, rb);
  }
  __syncthreads();
  // transposed store: destination row <-> tile column.  ROT 1: dst(y, uw-1-x) = ud(x, y), pixels in rising y;
  // ROT 3: dst(uh-1-y, x) = ud(x, y), pixels in falling y (ud_rot_store)
  const int t = threadIdx.x, qd = t & 15, cc = (t >> 4) & 3, grp = t >> 6; // row quad, column in the group of 4, wave
  constexpr int HALVES = TH / 64;                                           // 16 row quads per store instruction
#pragma unroll 1
  for (int pass = 0; pass < 4 * HALVES; ++pass) {
    const int half = pass % HALVES, lc = ((pass / HALVES) * 4 + grp) * 4 + cc; // 16 groups of 4 columns x the halves of the rows
    const int x = xw + lc;
    const int quad = half * 16 + qd;                                       // rows 4 quad .. 4 quad + 3 of the tile
    const int r0 = ROT == 1 ? 4 * quad : TH - 4 - 4 * quad;                // lowest tile row of the quad
    if (x >= dw || yb + r0 >= dh || yb + r0 < 0)                           // (dh % 4 == 0: a quad is whole or absent)
      continue;
    u32 pxl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pxl[j] = tile[(ROT == 1 ? r0 + j : r0 + 3 - j) * SD + lc];
    const int drow = ROT == 1 ? dw - 1 - x : x;
    const int dx0 = ROT == 1 ? yb + r0 : dh - 1 - (yb + r0 + 3);
    uint8_t* o = d.p[0] + (u32)(drow * d.pitch[0]) + (u32)(dx0 * 3);
    typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
    const v3u32 w = {pxl[0] | (pxl[1] << 24), (pxl[1] >> 8) | (pxl[2] << 16), (pxl[2] >> 16) | (pxl[3] << 8)};
    gstore_u_nt<v3u32>(o, w);
  }
}

// The same arithmetic with the output written turned by 90 / 270 degrees (ROT 1 / 3; NV12 -> packed
// RGB): BASELINE config 4 as one pass.  The transposed store needs the 256 x 32 workgroup tile of
// k_ud_nv12 (one dword per pixel in LDS, ud_rot_store), so a wave covers 256 columns and fetches TWO
// output rows per instruction: lanes 0-31 row 2i, lanes 32-63 row 2i+1 of its 8 (8 pixels = one
// dwordx4 per lane per source row, as above).  Row taps and row addresses are therefore per
// half-wave (vector registers); lanes 0 and 32 take the dword before their row from the wave's
// extra load.
template <int ROT>
__global__ void __launch_bounds__(kBlock) k_ud_down2_t(const UdArgs a) {
  using T = uint8_t;
  constexpr int OUT = UD_RGB_U8;
  static_assert(ROT == 1 || ROT == 3, "quarter turns only");
  u32 tile_x, tile_y, frame;
  {
    u32 t;
    if (!frame_tile_of_block(a.map, frame, t))
      return;
    const u32 tiles_y = a.map.per_frame / a.map.tiles_x; // consecutive workgroups walk DOWN the UD image
    tile_x = t / tiles_y;
    tile_y = t - tile_x * tiles_y;
  }
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const uint8_t* py = s.p[0];
  const uint8_t* puv = s.p[1];
  const int sp_y = s.pitch[0], sp_uv = s.pitch[1], sw = s.width, sh = s.height;
  const int dw = d.height, dh = d.width; // size of the (virtual) un-rotated UD output
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int half = lane >> 5, li = lane & 31;
  const int xw = tile_x * 256;
  const int x0 = xw + li * kD2LanePx;
  const int y_first = tile_y * kUdTileH + wave * kUdRowsPerWave; // wave-uniform
  __shared__ __attribute__((aligned(16))) uint8_t rot_tile[kRotTileBytes];
  auto body = [&]() { // (a lambda so that its early exits still reach the transposed store below)
    if (y_first >= dh)
      return;
    const int n = min(kD2LanePx, dw - x0);
    const float scale_y = 1.0f * (float)dh / (float)sh;
    const float cyl = (float)(y_first + (lane & (kUdRowsPerWave - 1))) / scale_y;
    const Tap vty = make_tap(cyl, sh), vtcy = make_tap(cyl * 0.5f, sh / 2);
    const int last = min(kUdRowsPerWave - 1, dh - 1 - y_first); // last valid row of the wave
    // taps of row pair `it`: lanes 0-31 get row 2 it, lanes 32-63 row 2 it + 1 (clamped to `last`);
    // `even` = all four vertical weights of both rows are 128 (wave-uniform)
    auto pair_taps = [&](int it, bool& even) {
      const int ra = min(2 * it, last), rb = min(2 * it + 1, last);
      auto pick = [&](int v) {
        const int sa = __builtin_amdgcn_readlane(v, ra), sb = __builtin_amdgcn_readlane(v, rb);
        return half ? sb : sa;
      };
      D2Taps r;
      r.ty.i0 = pick(vty.i0); r.ty.i1 = pick(vty.i1);
      r.ty.w0 = (u32)pick((int)vty.w0); r.ty.w1 = (u32)pick((int)vty.w1);
      r.tcy.i0 = pick(vtcy.i0); r.tcy.i1 = pick(vtcy.i1);
      r.tcy.w0 = (u32)pick((int)vtcy.w0); r.tcy.w1 = (u32)pick((int)vtcy.w1);
      even = __builtin_amdgcn_readlane((int)vty.w0, ra) == 128 && __builtin_amdgcn_readlane((int)vty.w0, rb) == 128 &&
             __builtin_amdgcn_readlane((int)vtcy.w0, ra) == 128 && __builtin_amdgcn_readlane((int)vtcy.w0, rb) == 128;
      return r;
    };
    const D2Src srcv = {py, puv, sp_y, sp_uv, sw};
    // LDS tile row of output row rr of the wave
    auto tile_row = [&](int rr) { return reinterpret_cast<u32*>(rot_tile + (wave * kUdRowsPerWave + rr) * kRotStride); };
    auto slow = [&](const D2Taps& rt, int rr) { // 4 pixels at a time through the byte gather
#pragma unroll 1
      for (int h = 0; h < 2; ++h)
        if (n > 4 * h) {
          float c0[4], c1[4], c2[4];
          d2_compute<OUT, false>(rt, d2_gather(srcv, rt, x0 + 4 * h), c0, c1, c2);
          ud_emit<T, OUT, ROT, false>(d, rot_tile, wave, lane, rr, x0 + 4 * h, y_first + rr, min(4, n - 4 * h), dw, dh,
                                      c0, c1, c2, li * kD2LanePx + 4 * h);
        }
    };
    const bool aligned = ((((uintptr_t)py) | ((uintptr_t)puv) | (uintptr_t)sp_y | (uintptr_t)sp_uv) & 15u) == 0 && sw >= 16;
    if (!aligned) {
#pragma unroll 1
      for (int it = 0; 2 * it <= last; ++it) {
        bool even;
        const D2Taps rt = pair_taps(it, even);
        const int rr = 2 * it + half;
        if (rr <= last && n > 0)
          slow(rt, rr);
      }
      return;
    }
    struct Rows {
      uint4 v[4]; // the lane's 16 bytes of luma i0, luma i1, chroma i0, chroma i1 of ITS row
      u32 before; // lanes 0-3 / 32-35: the dword before the wave's first byte in row k of their half
    };
    const int off16 = min(2 * x0, (sw - 16) & ~15);
    const int offw = max(2 * xw - 4, 0);
    auto issue = [&](const D2Taps& rt) {
      Rows r;
      const uint8_t* row[4] = {py + (u32)(rt.ty.i0 * sp_y), py + (u32)(rt.ty.i1 * sp_y),
                               puv + (u32)(rt.tcy.i0 * sp_uv), puv + (u32)(rt.tcy.i1 * sp_uv)};
#pragma unroll
      for (int k = 0; k < 4; ++k)
        r.v[k] = gload16(row[k] + (u32)off16);
      const uint8_t* rb = li == 0 ? row[0] : li == 1 ? row[1] : li == 2 ? row[2] : row[3];
      r.before = gload<u32>(rb + (u32)offw);
      return r;
    };
    auto step = [&](int it, const D2Taps& cur, bool even, const Rows& rows) {
      const int rr = 2 * it + half;
      u32 before[4]; // broadcast before any divergence
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 f0 = (u32)__builtin_amdgcn_readlane((int)rows.before, k);
        const u32 f1 = (u32)__builtin_amdgcn_readlane((int)rows.before, 32 + k);
        before[k] = half ? f1 : f0;
      }
      if (rr > last)
        return;
      if (n == kD2LanePx) {
        u32 prev[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 edge = k < 2 ? rows.v[k].x << 24 : rows.v[k].x << 16;
          const u32 first = xw == 0 ? edge : before[k];
          const u32 sh1 = wave_shr1(rows.v[k].w);
          prev[k] = li == 0 ? first : sh1;
        }
        D2Quad q0, q1;
        float c0[8], c1[8], c2[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          q0.p[k] = prev[k]; q0.a[k] = rows.v[k].x; q0.b[k] = rows.v[k].y;
          q1.p[k] = rows.v[k].y; q1.a[k] = rows.v[k].z; q1.b[k] = rows.v[k].w;
        }
        if (even) {
          d2_compute<OUT, true>(cur, q0, c0, c1, c2);
          d2_compute<OUT, true>(cur, q1, c0 + 4, c1 + 4, c2 + 4);
        } else {
          d2_compute<OUT, false>(cur, q0, c0, c1, c2);
          d2_compute<OUT, false>(cur, q1, c0 + 4, c1 + 4, c2 + 4);
        }
        u32 px[8];
        trunc_pack_px4(c0, c1, c2, px);
        trunc_pack_px4(c0 + 4, c1 + 4, c2 + 4, px + 4);
        u32* t = tile_row(rr) + li * kD2LanePx; // 8-byte aligned (tile rows are 1032 bytes apart)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<uint2*>(t + 2 * g) = make_uint2(px[2 * g], px[2 * g + 1]);
      } else if (n > 0) {
        slow(cur, rr);
      }
    };
    bool ea;
    D2Taps ta = pair_taps(0, ea);
    Rows ra = issue(ta);
#pragma unroll 1
    for (int it = 0; 2 * it <= last; ++it) {
      bool eb;
      const D2Taps tb = pair_taps(it + 1, eb); // (clamped to the last row: a harmless re-read at the end)
      const Rows rb = issue(tb);
      step(it, ta, ea, ra);
      ta = tb;
      ea = eb;
      ra = rb;
    }
  };
  body();
  __syncthreads();
  ud_rot_store<ROT>(d, rot_tile, tile_x, tile_y, dw, dh);
}

static int ud_out_kind(int src_fmt, int dst_fmt) {
  // SupportedConversions(), src/TC/src/UDSurface.cpp:117-133 (semi-planar sources)
  if (src_fmt == VALI_FMT_NV12) {
    switch (dst_fmt) {
    case VALI_FMT_YUV444: return UD_YUV444;
    case VALI_FMT_RGB: return UD_RGB_U8;
    case VALI_FMT_RGB_PLANAR: return UD_RGB_U8_PLANAR;
    case VALI_FMT_RGB_32F: return UD_RGB_F32;
    case VALI_FMT_RGB_32F_PLANAR: return UD_RGB_F32_PLANAR;
    default: return -1;
    }
  }
  if (src_fmt == VALI_FMT_P10) {
    switch (dst_fmt) {
    case VALI_FMT_YUV444_10BIT: return UD_YUV444;
    case VALI_FMT_RGB_32F: return UD_RGB_F32;
    case VALI_FMT_RGB_32F_PLANAR: return UD_RGB_F32_PLANAR;
    default: return -1;
    }
  }
  return -1;
}

// dst_w / dst_h: size of the DESTINATION surface; for odd `rot` the UD output itself is
// dst_h x dst_w and is written turned.
static int launch_ud(UdArgs& a, int src_fmt, int src_w, int src_h, int dst_w, int dst_h, int dst_fmt, int n,
                     hipStream_t stream, int rot = 0) {
  const int kind = ud_out_kind(src_fmt, dst_fmt);
  if (kind < 0)
    return fail(VALI_ERR_UNSUPPORTED, "ud_nv12: unsupported format pair %d -> %d", src_fmt, dst_fmt);
  if (rot < 0 || rot > 3)
    return fail(VALI_ERR_INVALID_ARG, "ud_nv12: quarter_turns must be 0..3 (got %d)", rot);
  if (rot != 0 && !(src_fmt == VALI_FMT_NV12 && kind == UD_RGB_U8))
    return fail(VALI_ERR_UNSUPPORTED, "ud_nv12: rotated output is implemented for NV12 -> RGB only");
  if (rot & 1) {
    const int t = dst_w;
    dst_w = dst_h;
    dst_h = t;
  }
  // Rows per wave: 8 -- the column taps are paid once per 8 rows -- unless that leaves SIMDs without a wave.  A wave
  // walks its rows one memory round trip after the other, so ONE 1080p frame through 8-row waves (136 waves on 1024
  // SIMDs) took 9.2 us however little work it is; 2- or 4-row waves give the same frame 4x / 2x the waves, each a
  // quarter / half as long.  Batches keep 8.  (Un-rotated outputs only: the transposed forms collect 256 x 32 tiles.)
  auto rows_for = [&](int tiles_x) {
    const int forced = tuning(VALI_TUNE_ROWS_PER_WAVE);
    if (rot == 0 && (forced == 2 || forced == 4 || forced == 8))
      return forced;
    int rows = kUdRowsPerWave;
    while (rot == 0 && rows > 2 &&
           (long long)tiles_x * ((dst_h + kWavesPerBlock * rows - 1) / (kWavesPerBlock * rows)) * n * kWavesPerBlock < 2048)
      rows /= 2;
    return rows;
  };
  a.rows = rows_for((dst_w + 255) / 256);
  a.map = make_tile_map((dst_w + 255) / 256, (dst_h + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows), (u32)n);
  const dim3 grid = tile_grid(a.map), block(kBlock);
  // staged kernel iff every tile's source spans fit the strip (same float math as the device)
  bool staged = true;
  {
    const float scale_x = 1.0f * (float)dst_w / (float)src_w;
    for (int t = 0; t < (dst_w + 255) / 256 && staged; ++t) {
      const UdSpan sp = src_fmt == VALI_FMT_NV12 ? ud_span<uint8_t>(t, dst_w, src_w, scale_x)
                                                 : ud_span<uint16_t>(t, dst_w, src_w, scale_x);
      const int cap = kUdRowBytes * (src_fmt == VALI_FMT_NV12 ? 1 : 2);
      staged = sp.yn <= cap && sp.cn <= cap;
    }
  }
  // VALI_UD_FORCE_GATHER=1: the direct-gather form for every geometry, and no exact-2x kernels (tests
  // reach the gather code with ordinary sizes; it normally serves > 4x downscales only)
  const bool force_gather = tuning(VALI_TUNE_UD_FORCE_GATHER) == 1;
  if (force_gather)
    staged = false;
  // exact 2x horizontal downscale of NV12: the division-free, LDS-free kernel (VALI_UD_DOWN2=0
  // keeps the general one, for A/B measurements)
  // The exact-ratio kernels own 8 output pixels per lane.  Un-rotated, a width that is not a multiple of 8 slides its
  // last lane left (round 1's k_ud_down2: `slid`); the turned forms have no such lane and would leave it on their byte-gather path
  // with the whole wave waiting (1916x1076 -> 958x538 half-turned: 2.4 us against 1.1 through the general kernel), so
  // those geometries go to the general kernel.  VALI_TUNE_UD_DOWN2 = 2 keeps them here (A/B, tests of that path).
  const int down2_mode = tuning(VALI_TUNE_UD_DOWN2);
  // the lean exact-ratio kernels own 8 output pixels per lane: widths that are multiples of 8, un-rotated; ragged widths and
  // the half turn at these ratios run on the general kernel (bit-identical)
  const bool lean_on = down2_mode != 0 && !force_gather && src_fmt == VALI_FMT_NV12 && rot == 0 && dst_w >= kD2LanePx &&
                       kind != UD_RGB_F32 && kind != UD_RGB_F32_PLANAR;
  const bool lean_ragged = dst_w % kD2LanePx != 0;      // the row's last lane slides left (k_ud_lean<..., RAGGED>) // (float outputs are store-bound: 4 pixels per lane fill their stores better)
  const bool down2_on = down2_mode != 0 && (down2_mode == 2 || dst_w % kD2LanePx == 0 || (rot == 0 && dst_w >= kD2LanePx));
  if (lean_on && src_w == dst_w) { // 1:1 width: colour conversion with chroma interpolation
    a.rows = rows_for((dst_w + kD2WaveW - 1) / kD2WaveW);
    a.map = make_tile_map((dst_w + kD2WaveW - 1) / kD2WaveW, (dst_h + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows), (u32)n);
    const dim3 g1 = tile_grid(a.map);
    const bool even = (src_h == dst_h || src_h == 2 * dst_h) && down2_mode != 2; // (UD_DOWN2 = 2: the general rows, for A/B and tests)
#define VALI_UD_LEAN(K) do { if (lean_ragged) { if (even) hipLaunchKernelGGL((k_ud_lean<K, 1, true, true>), g1, block, 0, stream, a); \
                                               else hipLaunchKernelGGL((k_ud_lean<K, 1, false, true>), g1, block, 0, stream, a); }   \
                             else if (even) hipLaunchKernelGGL((k_ud_lean<K, 1, true>), g1, block, 0, stream, a); \
                             else hipLaunchKernelGGL((k_ud_lean<K, 1, false>), g1, block, 0, stream, a); } while (0)
    if (kind == UD_YUV444) VALI_UD_LEAN(UD_YUV444);
    else if (kind == UD_RGB_U8) VALI_UD_LEAN(UD_RGB_U8);
    else VALI_UD_LEAN(UD_RGB_U8_PLANAR);
#undef VALI_UD_LEAN
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (down2_mode == 1 && !force_gather && src_fmt == VALI_FMT_NV12 && 2 * src_w == 3 * dst_w && 2 * src_h == 3 * dst_h && rot == 0 &&
      dst_w % kD2LanePx == 0 && dst_h % 4 == 0 && kind != UD_RGB_F32 && kind != UD_RGB_F32_PLANAR) { // exactly 3:2 both ways
    a.rows = rows_for((dst_w + kD2WaveW - 1) / kD2WaveW);
    a.map = make_tile_map((dst_w + kD2WaveW - 1) / kD2WaveW, (dst_h + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows), (u32)n);
    const dim3 g32 = tile_grid(a.map);
    if (kind == UD_YUV444) hipLaunchKernelGGL((k_ud_32<UD_YUV444>), g32, block, 0, stream, a);
    else if (kind == UD_RGB_U8) hipLaunchKernelGGL((k_ud_32<UD_RGB_U8>), g32, block, 0, stream, a);
    else hipLaunchKernelGGL((k_ud_32<UD_RGB_U8_PLANAR>), g32, block, 0, stream, a);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (down2_mode == 1 && !force_gather && src_fmt == VALI_FMT_NV12 && src_w == 2 * dst_w && src_h == 2 * dst_h && rot == 0 &&
      dst_w % kD2LanePx == 0 && kind != UD_RGB_F32 && kind != UD_RGB_F32_PLANAR) { // exactly 2:1 both ways: the lean kernel
    a.rows = rows_for((dst_w + kD2WaveW - 1) / kD2WaveW);
    a.map = make_tile_map((dst_w + kD2WaveW - 1) / kD2WaveW, (dst_h + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows), (u32)n);
    const dim3 gh = tile_grid(a.map);
    if (kind == UD_YUV444) hipLaunchKernelGGL((k_ud_half<UD_YUV444>), gh, block, 0, stream, a);
    else if (kind == UD_RGB_U8) hipLaunchKernelGGL((k_ud_half<UD_RGB_U8>), gh, block, 0, stream, a);
    else hipLaunchKernelGGL((k_ud_half<UD_RGB_U8_PLANAR>), gh, block, 0, stream, a);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (lean_on && src_w == 2 * dst_w) { // 2:1 along x, any height
    a.rows = rows_for((dst_w + kD2WaveW - 1) / kD2WaveW);
    a.map = make_tile_map((dst_w + kD2WaveW - 1) / kD2WaveW, (dst_h + kWavesPerBlock * a.rows - 1) / (kWavesPerBlock * a.rows), (u32)n);
    const dim3 g2 = tile_grid(a.map);
    const bool even = (src_h == dst_h || src_h == 2 * dst_h) && down2_mode != 2; // (exactly halved heights at widths that are multiples of 8: k_ud_half above)
#define VALI_UD_LEAN(K) do { if (lean_ragged) { if (even) hipLaunchKernelGGL((k_ud_lean<K, 2, true, true>), g2, block, 0, stream, a); \
                                               else hipLaunchKernelGGL((k_ud_lean<K, 2, false, true>), g2, block, 0, stream, a); }   \
                             else if (even) hipLaunchKernelGGL((k_ud_lean<K, 2, true>), g2, block, 0, stream, a); \
                             else hipLaunchKernelGGL((k_ud_lean<K, 2, false>), g2, block, 0, stream, a); } while (0)
    if (kind == UD_YUV444) VALI_UD_LEAN(UD_YUV444);
    else if (kind == UD_RGB_U8) VALI_UD_LEAN(UD_RGB_U8);
    else VALI_UD_LEAN(UD_RGB_U8_PLANAR);
#undef VALI_UD_LEAN
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (down2_mode == 1 && !force_gather && src_fmt == VALI_FMT_NV12 && src_w == 2 * dst_w && src_h == 2 * dst_h && (rot & 1) &&
      dst_w % kD2LanePx == 0 && dst_h % 4 == 0) { // exactly 2:1 both ways: 64 x 128 tiles, whole-line destination segments
    // 64-row tiles (17 KB of LDS, 9 workgroups per CU) beat 128-row ones (33 KB, 4 per CU) although their destination segments
    // are 192 bytes instead of three whole lines -- the neighbouring tile follows on the same XCD and the L2 merges the halves:
    // 2160p -> 1080p 90 deg 3.67 vs 3.75 us, 270 deg 3.71 vs 4.15, 1080p -> 540p 0.87 vs 0.93 (VALI_TUNE_ROTATE_NO_TILE = 3: the tall form)
    const bool tall = tuning(VALI_TUNE_ROTATE_NO_TILE) == 3;
    a.map = make_tile_map((dst_w + 63) / 64, tall ? (dst_h + 127) / 128 : (dst_h + 63) / 64, (u32)n);
    const dim3 gt = tile_grid(a.map);
    if (tall) {
      if (rot == 1) hipLaunchKernelGGL((k_ud_half_t<1, 128>), gt, block, 0, stream, a);
      else hipLaunchKernelGGL((k_ud_half_t<3, 128>), gt, block, 0, stream, a);
    } else {
      if (rot == 1) hipLaunchKernelGGL((k_ud_half_t<1, 64>), gt, block, 0, stream, a);
      else hipLaunchKernelGGL((k_ud_half_t<3, 64>), gt, block, 0, stream, a);
    }
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (down2_on && !force_gather && src_fmt == VALI_FMT_NV12 && src_w == 2 * dst_w && (rot & 1)) {
    if (rot == 1) hipLaunchKernelGGL((k_ud_down2_t<1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_ud_down2_t<3>), grid, block, 0, stream, a);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
#define VALI_UD_CASE(T, K)                                                                  \
  case K:                                                                                   \
    if (staged)                                                                             \
      hipLaunchKernelGGL((k_ud_nv12<T, K, true>), grid, block, 0, stream, a);                \
    else                                                                                    \
      hipLaunchKernelGGL((k_ud_nv12<T, K, false>), grid, block, 0, stream, a);               \
    break;
  if (rot != 0) {
#define VALI_UD_ROT(R)                                                                      \
  case R:                                                                                   \
    if (staged)                                                                             \
      hipLaunchKernelGGL((k_ud_nv12<uint8_t, UD_RGB_U8, true, R>), grid, block, 0, stream, a);  \
    else                                                                                    \
      hipLaunchKernelGGL((k_ud_nv12<uint8_t, UD_RGB_U8, false, R>), grid, block, 0, stream, a); \
    break;
    switch (rot) {
      VALI_UD_ROT(1)
      VALI_UD_ROT(2)
      VALI_UD_ROT(3)
    }
#undef VALI_UD_ROT
  } else if (src_fmt == VALI_FMT_NV12) {
    switch (kind) {
      VALI_UD_CASE(uint8_t, UD_YUV444)
      VALI_UD_CASE(uint8_t, UD_RGB_U8)
      VALI_UD_CASE(uint8_t, UD_RGB_U8_PLANAR)
      VALI_UD_CASE(uint8_t, UD_RGB_F32)
      VALI_UD_CASE(uint8_t, UD_RGB_F32_PLANAR)
    }
  } else {
    switch (kind) {
      VALI_UD_CASE(uint16_t, UD_YUV444)
      VALI_UD_CASE(uint16_t, UD_RGB_F32)
      VALI_UD_CASE(uint16_t, UD_RGB_F32_PLANAR)
    }
  }
#undef VALI_UD_CASE
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali

using namespace vali;

extern "C" {

int vali_ud_nv12(const vali_surface* src, const vali_surface* dst, vali_stream_t stream) {
  VALI_REQUIRE(src && dst, "null argument");
  VALI_REQUIRE(src->width >= 2 && src->height >= 2 && dst->width > 0 && dst->height > 0,
               "empty surface");
  VALI_REQUIRE(subsampled_sizes_ok(src->format, src->width, src->height), "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(src->plane[0] && src->plane[1] && dst->plane[0], "null plane");
  VALI_REQUIRE(planes_fit_32bit(*src) && planes_fit_32bit(*dst), "plane of 4 GiB or more");
  if (dst->format != VALI_FMT_RGB && dst->format != VALI_FMT_RGB_32F)
    VALI_REQUIRE(dst->plane[1] && dst->plane[2], "null dst plane");
  UdArgs a = {};
  a.src = *src;
  a.dst = *dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_ud(a, src->format, src->width, src->height, dst->width, dst->height, dst->format, 1, s);
}

int vali_ud_nv12_rot(const vali_surface* src, const vali_surface* dst, int quarter_turns,
                     vali_stream_t stream) {
  VALI_REQUIRE(src && dst, "null argument");
  VALI_REQUIRE(src->width >= 2 && src->height >= 2 && dst->width > 0 && dst->height > 0,
               "empty surface");
  VALI_REQUIRE(subsampled_sizes_ok(src->format, src->width, src->height), "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(src->plane[0] && src->plane[1] && dst->plane[0], "null plane");
  VALI_REQUIRE(planes_fit_32bit(*src) && planes_fit_32bit(*dst), "plane of 4 GiB or more");
  UdArgs a = {};
  a.src = *src;
  a.dst = *dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_ud(a, src->format, src->width, src->height, dst->width, dst->height, dst->format, 1, s, quarter_turns);
}

int vali_ud_nv12_rot_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_format,
                           int src_width, int src_height, int dst_width, int dst_height, int dst_format,
                           int quarter_turns, vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst, "null argument");
  VALI_REQUIRE(src_width >= 2 && src_height >= 2 && dst_width > 0 && dst_height > 0, "empty geometry");
  VALI_REQUIRE(((src_width | src_height) & 1) == 0, "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  UdArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_ud(a, src_format, src_width, src_height, dst_width, dst_height, dst_format, n, s, quarter_turns);
}

int vali_ud_nv12_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_format,
                       int src_width, int src_height, int dst_width, int dst_height, int dst_format,
                       vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst, "null argument");
  VALI_REQUIRE(src_width >= 2 && src_height >= 2 && dst_width > 0 && dst_height > 0, "empty geometry");
  VALI_REQUIRE(((src_width | src_height) & 1) == 0, "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  UdArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_ud(a, src_format, src_width, src_height, dst_width, dst_height, dst_format, n, s);
}

} // extern "C"
