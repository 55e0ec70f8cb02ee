/*
 * vali_oracle.c -- CPU restatement (the specification) of the hot path.
 * TEST INFRASTRUCTURE ONLY: see vali_oracle.h for who may use it and for the
 * parity status of each family.
 *
 * Build: gcc -O3 -std=c11 -mavx2 -mfma -ffp-contract=off (oracle/Makefile or
 * vali_amd/build.py).  -ffp-contract=off matters: every rounding step below is
 * part of the specification, the HIP kernels are compiled with the same flag
 * and use fmaf() in exactly the places this file does.
 */
#include "vali_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* --------------------------------------------------------------------------
 * quantiser: round to nearest, ties to even, saturate to [0, 255].
 * (NPP's rounding mode is not documented for these conversions -- SURVEY.md
 * A.3 -- nearest-even is this build's definition.)
 * -------------------------------------------------------------------------- */
static inline uint8_t q_u8(float v) {
  float r = __builtin_rintf(v); /* default rounding mode = nearest even */
  if (!(r > 0.0f))
    r = 0.0f; /* also maps NaN to 0 like fmaxf(NaN,0) */
  if (r > 255.0f)
    r = 255.0f;
  return (uint8_t)r;
}
uint8_t vali_oracle_q_u8(float v) { return q_u8(v); }

int vali_oracle_csc(int variant, vali_csc* out) {
  if (!out)
    return VALI_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  switch (variant) {
  case 0: /* NPP "YUV" */
    out->y0 = 0.0f; out->cy = 1.0f;
    out->crv = 1.140f; out->cgu = -0.394f; out->cgv = -0.581f; out->cbu = 2.032f;
    return VALI_OK;
  case 1: /* NPP 709 CSC (limited range) */
    out->y0 = 16.0f; out->cy = 1.164f;
    out->crv = 1.793f; out->cgu = -0.213f; out->cgv = -0.533f; out->cbu = 2.112f;
    return VALI_OK;
  case 2: /* NPP 709 HDTV (full range) */
    out->y0 = 0.0f; out->cy = 1.0f;
    out->crv = 1.5748f; out->cgu = -0.1873f; out->cgv = -0.4681f; out->cbu = 1.8556f;
    return VALI_OK;
  case 3: /* NPP YCbCr (601 limited range) */
    out->y0 = 16.0f; out->cy = 1.164f;
    out->crv = 1.596f; out->cgu = -0.392f; out->cgv = -0.813f; out->cbu = 2.017f;
    return VALI_OK;
  default:
    return VALI_ERR_UNSUPPORTED;
  }
}

/* --------------------------------------------------------------------------
 * YUV -> RGB for one pixel.  Operation order IS the specification:
 *   Uc = U - 128 ; Vc = V - 128                 (exact)
 *   rv  = crv * Vc                               (1 rounding)
 *   guv = fma(cgu, Uc, cgv * Vc)                 (2 roundings)
 *   bu  = cbu * Uc                               (1 rounding)
 *   Yf  = cy * (Y - y0)                          (exact sub, 1 rounding)
 *   R = Yf + rv ; G = Yf + guv ; B = Yf + bu     (1 rounding each)
 * -------------------------------------------------------------------------- */
typedef struct { float rv, guv, bu; } chroma_term;

static inline chroma_term chroma(float u, float v, const vali_csc* k) {
  const float uc = u - 128.0f, vc = v - 128.0f;
  chroma_term t;
  t.rv = k->crv * vc;
  t.guv = fmaf(k->cgu, uc, k->cgv * vc);
  t.bu = k->cbu * uc;
  return t;
}

static inline float luma(float y, const vali_csc* k) { return k->cy * (y - k->y0); }

static int rgb_layout_ok(int fmt) {
  return fmt == VALI_FMT_RGB || fmt == VALI_FMT_BGR || fmt == VALI_FMT_RGB_PLANAR;
}

int vali_oracle_nv12_to_rgb(const vali_surface* src, const vali_surface* dst,
                            const vali_csc* csc) {
  if (!src || !dst || !csc)
    return VALI_ERR_INVALID_ARG;
  if (src->format != VALI_FMT_NV12 || !rgb_layout_ok(dst->format))
    return VALI_ERR_UNSUPPORTED;
  if (src->width != dst->width || src->height != dst->height || src->width <= 0 ||
      src->height <= 0)
    return VALI_ERR_INVALID_ARG;
  const int W = src->width, H = src->height;
  const uint8_t* py = (const uint8_t*)src->plane[0];
  const uint8_t* puv = (const uint8_t*)src->plane[1];
  /* the same per-pixel arithmetic for every layout; the layout switch sits outside the pixel
   * loop so that the CPU baseline bench.py times is not dominated by branches and calls */
  const int planar = dst->format == VALI_FMT_RGB_PLANAR, rgb = dst->format == VALI_FMT_RGB;
  for (int y = 0; y < H; ++y) {
    const uint8_t* yrow = py + (size_t)y * src->pitch[0];
    const uint8_t* crow = puv + (size_t)(y / 2) * src->pitch[1];
    uint8_t* o0 = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0];
    uint8_t* o1 = planar ? (uint8_t*)dst->plane[1] + (size_t)y * dst->pitch[0] : 0;
    uint8_t* o2 = planar ? (uint8_t*)dst->plane[2] + (size_t)y * dst->pitch[0] : 0;
    for (int x = 0; x < W; ++x) {
      const uint8_t* c = crow + (x & ~1);
      const chroma_term t = chroma((float)c[0], (float)c[1], csc);
      const float yf = luma((float)yrow[x], csc);
      const uint8_t R = q_u8(yf + t.rv), G = q_u8(yf + t.guv), B = q_u8(yf + t.bu);
      if (planar) {
        o0[x] = R; o1[x] = G; o2[x] = B;
      } else if (rgb) {
        o0[3 * x] = R; o0[3 * x + 1] = G; o0[3 * x + 2] = B;
      } else {
        o0[3 * x] = B; o0[3 * x + 1] = G; o0[3 * x + 2] = R;
      }
    }
  }
  return VALI_OK;
}

int vali_oracle_nv12_to_rgb_mt(const vali_surface* src, const vali_surface* dst, int n,
                               const vali_csc* csc, int threads) {
  if (!src || !dst || !csc || n < 0)
    return VALI_ERR_INVALID_ARG;
  int rc = VALI_OK;
#ifdef _OPENMP
  if (threads < 1)
    threads = 1;
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1)
#endif
  for (int i = 0; i < n; ++i) {
    const int r = vali_oracle_nv12_to_rgb(&src[i], &dst[i], csc);
    if (r != VALI_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc = r;
    }
  }
  return rc;
}

/* ==========================================================================
 * UD (reference: src/TC/src/ResizeUtils.cu).
 *
 * CUDA linear filtering at unnormalised coordinate X of a texture with N texels
 * (CUDA C programming guide, appendix "Texture Fetching / Linear Filtering"):
 *   xB = X - 0.5 ; i = floor(xB) ; alpha = frac(xB), stored with 8 fractional bits
 *   tex(X) = (1-alpha) T[i] + alpha T[i+1], indices clamped to [0, N-1]
 * The 2-D filter is the tensor product.  With alpha = qa/256, beta = qb/256 the sum
 *   S = sum w_ij T_ij ,  w00=(256-qa)(256-qb) ... , sum w = 65536
 * is an exact integer; the normalised-float value is S / (65536 * max).  This build
 * defines that last step as ONE float multiply by the float constant 1/(65536*max).
 * (How the hardware rounds alpha and its internal precision are not observable
 * offline -- the goldens' input frame is missing -- so this is the restatement's
 * definition; the OUTPUT stage below is pinned by the goldens.)
 * ========================================================================== */
typedef struct { int i0, i1; uint32_t w0, w1; } tap;

static inline tap make_tap(float coord, int size) {
  const float b = coord - 0.5f;
  const float fl = floorf(b);
  const float frac = b - fl;
  const uint32_t q = (uint32_t)(frac * 256.0f + 0.5f);
  const int i = (int)fl;
  tap t;
  t.i0 = i < 0 ? 0 : (i > size - 1 ? size - 1 : i);
  t.i1 = i + 1 < 0 ? 0 : (i + 1 > size - 1 ? size - 1 : i + 1);
  t.w1 = q;
  t.w0 = 256u - q;
  return t;
}

/* cvt.rzi.<int>.f32 with saturation: what (uint8_t)(float) / (uint16_t)(float) compile
 * to on the reference's GPU; proven for u8 by RGB == trunc(RGB_32F * 256) on the goldens. */
static inline uint32_t trunc_sat(float v, float hi) {
  if (!(v > 0.0f))
    return 0;
  if (v > hi)
    v = hi;
  return (uint32_t)v;
}

static inline uint32_t texel(const uint8_t* row, int idx, int elem) {
  return elem == 1 ? row[idx] : ((const uint16_t*)row)[idx];
}

int vali_oracle_ud_nv12(const vali_surface* src, const vali_surface* dst) {
  if (!src || !dst)
    return VALI_ERR_INVALID_ARG;
  const int elem = src->format == VALI_FMT_NV12 ? 1 : (src->format == VALI_FMT_P10 ? 2 : 0);
  if (!elem)
    return VALI_ERR_UNSUPPORTED;
  const int f = dst->format;
  const int yuv = (elem == 1 && f == VALI_FMT_YUV444) || (elem == 2 && f == VALI_FMT_YUV444_10BIT);
  const int rgb8 = elem == 1 && (f == VALI_FMT_RGB || f == VALI_FMT_RGB_PLANAR);
  const int rgbf = f == VALI_FMT_RGB_32F || f == VALI_FMT_RGB_32F_PLANAR;
  if (!yuv && !rgb8 && !rgbf)
    return VALI_ERR_UNSUPPORTED;
  const int sw = src->width, sh = src->height, dw = dst->width, dh = dst->height;
  if (sw < 2 || sh < 2 || dw <= 0 || dh <= 0)
    return VALI_ERR_INVALID_ARG;
  const float inv_den = elem == 1 ? 1.0f / 16711680.0f : 1.0f / 4294901760.0f;
  const float maxv = elem == 1 ? 256.0f : 65536.0f;
  const float sat = elem == 1 ? 255.0f : 65535.0f;
  /* ResizeUtils.cu:135-136 */
  const float scale_x = 1.0f * (float)dw / (float)sw;
  const float scale_y = 1.0f * (float)dh / (float)sh;
  const uint8_t* py = (const uint8_t*)src->plane[0];
  const uint8_t* puv = (const uint8_t*)src->plane[1];
  for (int y = 0; y < dh; ++y) {
    const tap ty = make_tap((float)y / scale_y, sh);                 /* :36, :68 */
    const tap tcy = make_tap((float)y / (scale_y * 2.0f), sh / 2);    /* :37, :69 */
    const uint8_t* yr0 = py + (size_t)ty.i0 * src->pitch[0];
    const uint8_t* yr1 = py + (size_t)ty.i1 * src->pitch[0];
    const uint8_t* cr0 = puv + (size_t)tcy.i0 * src->pitch[1];
    const uint8_t* cr1 = puv + (size_t)tcy.i1 * src->pitch[1];
    for (int x = 0; x < dw; ++x) {
      const tap tx = make_tap((float)x / scale_x, sw);
      const tap tcx = make_tap((float)x / (scale_x * 2.0f), sw / 2);
      const uint32_t sy = ty.w0 * (tx.w0 * texel(yr0, tx.i0, elem) + tx.w1 * texel(yr0, tx.i1, elem)) +
                          ty.w1 * (tx.w0 * texel(yr1, tx.i0, elem) + tx.w1 * texel(yr1, tx.i1, elem));
      const uint32_t su =
          tcy.w0 * (tcx.w0 * texel(cr0, 2 * tcx.i0, elem) + tcx.w1 * texel(cr0, 2 * tcx.i1, elem)) +
          tcy.w1 * (tcx.w0 * texel(cr1, 2 * tcx.i0, elem) + tcx.w1 * texel(cr1, 2 * tcx.i1, elem));
      const uint32_t sv =
          tcy.w0 * (tcx.w0 * texel(cr0, 2 * tcx.i0 + 1, elem) + tcx.w1 * texel(cr0, 2 * tcx.i1 + 1, elem)) +
          tcy.w1 * (tcx.w0 * texel(cr1, 2 * tcx.i0 + 1, elem) + tcx.w1 * texel(cr1, 2 * tcx.i1 + 1, elem));
      const float ny = (float)sy * inv_den, nu = (float)su * inv_den, nv = (float)sv * inv_den;
      float c0, c1, c2;
      if (yuv) {
        c0 = ny; c1 = nu; c2 = nv;
      } else {
        /* ResizeUtils.cu:71-77; nvcc's default -fmad=true contracts each a + b*c */
        const float u = nu - 0.5f, v = nv - 0.5f;
        c0 = fmaf(1.140f, v, ny);
        c1 = fmaf(-0.581f, v, fmaf(-0.394f, u, ny));
        c2 = fmaf(2.032f, u, ny);
      }
      if (yuv) {                                                     /* :40-42 */
        for (int c = 0; c < 3; ++c) {
          const float val = (c == 0 ? c0 : c == 1 ? c1 : c2) * maxv;
          uint8_t* row = (uint8_t*)dst->plane[c] + (size_t)y * dst->pitch[c];
          if (elem == 1)
            row[x] = (uint8_t)trunc_sat(val, sat);
          else
            ((uint16_t*)row)[x] = (uint16_t)trunc_sat(val, sat);
        }
      } else if (rgb8) {                                             /* :45-54, :79-95 */
        const uint8_t R = (uint8_t)trunc_sat(c0 * 256.0f, 255.0f), G = (uint8_t)trunc_sat(c1 * 256.0f, 255.0f),
                      B = (uint8_t)trunc_sat(c2 * 256.0f, 255.0f);
        if (f == VALI_FMT_RGB_PLANAR) {
          ((uint8_t*)dst->plane[0])[(size_t)y * dst->pitch[0] + x] = R;
          ((uint8_t*)dst->plane[1])[(size_t)y * dst->pitch[0] + x] = G;
          ((uint8_t*)dst->plane[2])[(size_t)y * dst->pitch[0] + x] = B;
        } else {
          uint8_t* q = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0] + (size_t)x * 3;
          q[0] = R; q[1] = G; q[2] = B;
        }
      } else {
        if (f == VALI_FMT_RGB_32F_PLANAR) {
          ((float*)((uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0]))[x] = c0;
          ((float*)((uint8_t*)dst->plane[1] + (size_t)y * dst->pitch[0]))[x] = c1;
          ((float*)((uint8_t*)dst->plane[2] + (size_t)y * dst->pitch[0]))[x] = c2;
        } else {
          float* q = (float*)((uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0]) + (size_t)x * 3;
          q[0] = c0; q[1] = c1; q[2] = c2;
        }
      }
    }
  }
  return VALI_OK;
}

/* Output stages of the UD kernels exposed on their own so tests can pin them against the
 * reference's golden files without the (missing) input frame. */
void vali_oracle_ud_rgb_from_yuv(float ny, float nu, float nv, float* rgb) {
  const float u = nu - 0.5f, v = nv - 0.5f;        /* ResizeUtils.cu:71-77 */
  rgb[0] = fmaf(1.140f, v, ny);
  rgb[1] = fmaf(-0.581f, v, fmaf(-0.394f, u, ny));
  rgb[2] = fmaf(2.032f, u, ny);
}

uint8_t vali_oracle_ud_store_u8(float normalised) {  /* Denormalize<uint8_t> + (uint8_t) cast */
  return (uint8_t)trunc_sat(normalised * 256.0f, 255.0f);
}

/* ==========================================================================
 * Rotation (reference: src/TC/src/RotateSurface.cpp:22-125 -> nppiRotate_*R_Ctx,
 * NPPI_INTER_LINEAR; shift normalisation src/python_vali/src/PySurfaceRotator.cpp:40-77).
 * NPP model: x' = x cos a + y sin a + sx ; y' = -x sin a + y cos a + sy ; destination
 * pixels are inverse-mapped and bilinearly interpolated; those whose source point lies
 * outside the source plane are left untouched.  NPP's interpolation arithmetic is not
 * published: the operation order below is this build's definition.  For multiples of
 * 90 degrees cos/sin are exact, the weights are 0 and the result is the exact pixel
 * permutation the reference's etalons (frame_0_{90,180,270}_deg.jpg) show.
 * ========================================================================== */
static void rotate_coeffs(double angle_deg, float* c, float* s) {
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

static inline float rot_texel(const uint8_t* row, int idx, int elem) {
  return elem == 1 ? (float)row[idx] : elem == 2 ? (float)((const uint16_t*)row)[idx]
                                                 : ((const float*)row)[idx];
}

int vali_oracle_rotate_plane(const void* src, int src_pitch, int src_w, int src_h, void* dst,
                             int dst_pitch, int dst_w, int dst_h, int elem, int channels,
                             double angle, double shift_x, double shift_y) {
  if (!src || !dst || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0)
    return VALI_ERR_INVALID_ARG;
  if ((elem != 1 && elem != 2 && elem != 4) || (channels != 1 && channels != 3))
    return VALI_ERR_INVALID_ARG;
  float c, s;
  rotate_coeffs(angle, &c, &s);
  const float sx = (float)shift_x, sy = (float)shift_y;
  const float wmax = (float)(src_w - 1), hmax = (float)(src_h - 1);
  for (int y = 0; y < dst_h; ++y) {
    const float dy = (float)y - sy;
    for (int x = 0; x < dst_w; ++x) {
      const float dx = (float)x - sx;
      const float xs = fmaf(-s, dy, c * dx);
      const float ys = fmaf(c, dy, s * dx);
      if (!(xs >= 0.0f && xs <= wmax && ys >= 0.0f && ys <= hmax))
        continue;
      const float fi = floorf(xs), fj = floorf(ys);
      const float fa = xs - fi, fb = ys - fj;
      const int i = (int)fi, j = (int)fj;
      const int i1 = i + 1 < src_w ? i + 1 : src_w - 1, j1 = j + 1 < src_h ? j + 1 : src_h - 1;
      const uint8_t* r0 = (const uint8_t*)src + (size_t)j * src_pitch;
      const uint8_t* r1 = (const uint8_t*)src + (size_t)j1 * src_pitch;
      uint8_t* drow = (uint8_t*)dst + (size_t)y * dst_pitch;
      for (int ch = 0; ch < channels; ++ch) {
        const float t00 = rot_texel(r0, i * channels + ch, elem), t10 = rot_texel(r0, i1 * channels + ch, elem);
        const float t01 = rot_texel(r1, i * channels + ch, elem), t11 = rot_texel(r1, i1 * channels + ch, elem);
        const float t0 = fmaf(fa, t10 - t00, t00);
        const float t1 = fmaf(fa, t11 - t01, t01);
        const float v = fmaf(fb, t1 - t0, t0);
        const int o = x * channels + ch;
        if (elem == 1) {
          drow[o] = vali_oracle_q_u8(v);
        } else if (elem == 2) {
          float r = rintf(v);
          if (!(r > 0.0f)) r = 0.0f;
          if (r > 65535.0f) r = 65535.0f;
          ((uint16_t*)drow)[o] = (uint16_t)r;
        } else {
          ((float*)drow)[o] = v;
        }
      }
    }
  }
  return VALI_OK;
}

/* ==========================================================================
 * Bilinear resize (reference: src/TC/src/TaskResizeSurface.cpp -> nppiResize_*R_Ctx).
 * Sampling geometry src = dst * (src_size / dst_size) (no half-pixel shift): pinned by the
 * reference fixture tests/data/test_small.nv12, which equals src[2y][2x] of its source frame
 * (tests/test_oracle_resize.py).  The interpolation kernel differs from the reference
 * (bilinear per BASELINE.json vs NPP Lanczos, input golden missing): operation order below
 * is this build's definition; on integer scale factors both reduce to the same point sample.
 * ========================================================================== */
typedef struct { int i0, i1; float a; } lerp_t;

static inline lerp_t make_lerp(int x, float scale, int size) {
  const float f = (float)x * scale;
  const float fl = floorf(f);
  lerp_t l;
  l.a = f - fl;
  const int i = (int)fl;
  l.i0 = i < size - 1 ? i : size - 1;
  l.i1 = i + 1 < size - 1 ? i + 1 : size - 1;
  return l;
}

int vali_oracle_resize_plane(const void* src, int src_pitch, int src_w, int src_h, void* dst,
                             int dst_pitch, int dst_w, int dst_h, int elem, int channels) {
  if (!src || !dst || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0)
    return VALI_ERR_INVALID_ARG;
  if ((elem != 1 && elem != 2 && elem != 4) || channels < 1 || channels > 3)
    return VALI_ERR_INVALID_ARG;
  const float scale_x = (float)src_w / (float)dst_w, scale_y = (float)src_h / (float)dst_h;
  for (int y = 0; y < dst_h; ++y) {
    const lerp_t ly = make_lerp(y, scale_y, src_h);
    const uint8_t* r0 = (const uint8_t*)src + (size_t)ly.i0 * src_pitch;
    const uint8_t* r1 = (const uint8_t*)src + (size_t)ly.i1 * src_pitch;
    uint8_t* drow = (uint8_t*)dst + (size_t)y * dst_pitch;
    for (int x = 0; x < dst_w; ++x) {
      const lerp_t lx = make_lerp(x, scale_x, src_w);
      for (int ch = 0; ch < channels; ++ch) {
        const float t00 = rot_texel(r0, lx.i0 * channels + ch, elem), t10 = rot_texel(r0, lx.i1 * channels + ch, elem);
        const float t01 = rot_texel(r1, lx.i0 * channels + ch, elem), t11 = rot_texel(r1, lx.i1 * channels + ch, elem);
        const float t0 = fmaf(lx.a, t10 - t00, t00);
        const float t1 = fmaf(lx.a, t11 - t01, t01);
        const float v = fmaf(ly.a, t1 - t0, t0);
        const int o = x * channels + ch;
        if (elem == 1) {
          drow[o] = vali_oracle_q_u8(v);
        } else if (elem == 2) {
          float r = rintf(v);
          if (!(r > 0.0f)) r = 0.0f;
          if (r > 65535.0f) r = 65535.0f;
          ((uint16_t*)drow)[o] = (uint16_t)r;
        } else {
          ((float*)drow)[o] = v;
        }
      }
    }
  }
  return VALI_OK;
}

/* ==========================================================================
 * Lanczos-3 resize (reference: nppiResize_*R_Ctx with NPPI_INTER_LANCZOS,
 * src/TC/src/TaskResizeSurface.cpp:67,116,224,273).  NPP is closed source and the input of
 * the reference's resize golden is missing, so the tap arithmetic is this build's definition
 * (PARITY UNPINNED beyond the geometry): a 6x6 interpolating (not anti-aliasing) Lanczos-3
 * kernel on the pinned sampling grid src = dst * (src_size / dst_size) -- the fixture
 * test_small.nv12 shows the reference's 2x downscale is the point sample src[2y][2x], which an
 * interpolating kernel evaluated at integer offsets gives and an area-scaled one would not.
 *
 *   f = x * scale ; i = floor(f) ; a = f - i ; taps i-2 .. i+3, indices clamped to the plane
 *   raw_k = sin(pi t_k) sin(pi t_k / 3) / t_k^2 ,  t_k = a + (2 - k)     (3/pi^2 cancels)
 *   w_k = raw_k / (((((raw_0 + raw_1) + raw_2) + raw_3) + raw_4) + raw_5)
 *   a == 0: w = {0,0,1,0,0,0}
 * The twelve divisions of that definition are TWO: with d_k = t_k^2 and P = ((d0 d1)(d2 d3))(d4 d5),
 *   1/d_k = (1/P) * (product of the other five d, grouped as in the code) ;  w_k = raw_k * (1/sum)
 * (an IEEE division costs a GPU lane a dozen instructions, and every lane of every tile evaluates five
 * tap sets: the weights were a third of the kernel's arithmetic).  Same weights to ~1e-7.
 * sin/cos are FIXED polynomials (below) so CPU and GPU agree bit for bit; sin(pi(a+m)) =
 * (-1)^m sin(pi a) and sin(pi(a+m)/3) by the angle-sum identity from sin/cos(pi a / 3).
 *   rows:    even and odd taps accumulate separately -- e = w_0 t_0 ; e = fma(w_k, t_k, e), k = 2, 4 ;
 *            o = w_1 t_1 ; o = fma(w_k, t_k, o), k = 3, 5 ; h_r = e + o -- the form a packed-FP32 pipe
 *            (two lanes of one instruction) evaluates
 *   columns: v = wy_0 h_0 ; v = fma(wy_r, h_r, v), r = 1..5
 *   u8/u16: round-half-even + saturate ; f32: v
 * ========================================================================== */
static inline float lz_sin_poly(float z) { /* sin z, 0 <= z <= pi/2, Taylor to z^11 */
  const float z2 = z * z;
  float p = fmaf(z2, -2.5052108e-8f, 2.7557319e-6f);
  p = fmaf(z2, p, -1.9841270e-4f);
  p = fmaf(z2, p, 8.3333333e-3f);
  p = fmaf(z2, p, -1.6666667e-1f);
  return fmaf(z * z2, p, z);
}
static inline float lz_cos_poly(float z) { /* cos z, 0 <= z <= pi/3, Taylor to z^12 */
  const float z2 = z * z;
  float p = fmaf(z2, 2.0876757e-9f, -2.7557319e-7f);
  p = fmaf(z2, p, 2.4801587e-5f);
  p = fmaf(z2, p, -1.3888889e-3f);
  p = fmaf(z2, p, 4.1666667e-2f);
  p = fmaf(z2, p, -0.5f);
  return fmaf(z2, p, 1.0f);
}

void vali_oracle_lanczos3_weights(float a, float w[6]) {
  if (a == 0.0f) {
    w[0] = w[1] = w[3] = w[4] = w[5] = 0.0f;
    w[2] = 1.0f;
    return;
  }
  const float y = a <= 0.5f ? a : 1.0f - a;
  const float s1 = lz_sin_poly(y * 3.14159265f);          /* sin(pi a) */
  const float z = a * 1.04719755f;                         /* pi a / 3 */
  const float s3 = lz_sin_poly(z), c3 = lz_cos_poly(z);
  const float h = 0.866025404f;                            /* sin(pi/3) */
  /* m = 2, 1, 0, -1, -2, -3 : sin(z + m pi/3) and the sign (-1)^m of sin(pi (a + m)) */
  const float q[6] = {fmaf(c3, h, -0.5f * s3), fmaf(c3, h, 0.5f * s3), s3,
                      fmaf(c3, -h, 0.5f * s3), fmaf(c3, -h, -0.5f * s3), -s3};
  const float sg[6] = {1.0f, -1.0f, 1.0f, -1.0f, 1.0f, -1.0f};
  float d[6], inv[6], raw[6];
  for (int k = 0; k < 6; ++k) {
    const float t = a + (float)(2 - k);
    d[k] = t * t;
  }
  const float p01 = d[0] * d[1], p23 = d[2] * d[3], p45 = d[4] * d[5];
  const float r = 1.0f / ((p01 * p23) * p45);
  inv[0] = r * ((d[1] * p23) * p45); inv[1] = r * ((d[0] * p23) * p45);
  inv[2] = r * ((p01 * d[3]) * p45); inv[3] = r * ((p01 * d[2]) * p45);
  inv[4] = r * ((p01 * p23) * d[5]); inv[5] = r * ((p01 * p23) * d[4]);
  for (int k = 0; k < 6; ++k)
    raw[k] = ((sg[k] * s1) * q[k]) * inv[k];
  const float sum = ((((raw[0] + raw[1]) + raw[2]) + raw[3]) + raw[4]) + raw[5];
  const float rs = 1.0f / sum;
  for (int k = 0; k < 6; ++k)
    w[k] = raw[k] * rs;
}

/*
 * Bicubic (NPPI_INTER_CUBIC = 4; BASELINE.json's north_star names a "bilinear/bicubic resizer",
 * the reference itself never passes it): the Keys / Catmull-Rom cubic convolution kernel
 * (a = -1/2) on the same sampling grid, taps i-1 .. i+2:
 *   w0 = ((-a/2 + 1) a - 1/2) a      w1 = (3a/2 - 5/2) a^2 + 1
 *   w2 = ((-3a/2 + 2) a + 1/2) a     w3 = (a/2 - 1/2) a^2
 * in Horner form with explicit fma, not renormalised (the four sum to 1 analytically); a == 0
 * gives {0,1,0,0} by the formulas.  PARITY UNPINNED: which cubic NPP implements is not published.
 * Rows/columns accumulate exactly like the Lanczos filter above (4 taps instead of 6).
 */
void vali_oracle_cubic_weights(float a, float w[4]) {
  const float a2 = a * a;
  w[0] = a * fmaf(a, fmaf(a, -0.5f, 1.0f), -0.5f);
  w[1] = fmaf(a2, fmaf(a, 1.5f, -2.5f), 1.0f);
  w[2] = a * fmaf(a, fmaf(a, -1.5f, 2.0f), 0.5f);
  w[3] = a2 * fmaf(a, 0.5f, -0.5f);
}

typedef struct { int idx[6]; float w[6]; } lz_tap_t;

/* taps = 6: Lanczos-3 (i-2 .. i+3) ; taps = 4: cubic (i-1 .. i+2) */
static inline lz_tap_t lz_make_tap(int x, float scale, int size, int taps) {
  const float f = (float)x * scale;
  const float fl = floorf(f);
  lz_tap_t t;
  if (taps == 6)
    vali_oracle_lanczos3_weights(f - fl, t.w);
  else
    vali_oracle_cubic_weights(f - fl, t.w);
  const int i = (int)fl;
  for (int k = 0; k < taps; ++k) {
    int j = i - (taps / 2 - 1) + k;
    j = j < 0 ? 0 : (j > size - 1 ? size - 1 : j);
    t.idx[k] = j;
  }
  return t;
}

static inline void taps_store(uint8_t* drow, int o, float v, int elem) {
  if (elem == 1) {
    drow[o] = vali_oracle_q_u8(v);
  } else if (elem == 2) {
    float rr = rintf(v);
    if (!(rr > 0.0f)) rr = 0.0f;
    if (rr > 65535.0f) rr = 65535.0f;
    ((uint16_t*)drow)[o] = (uint16_t)rr;
  } else {
    ((float*)drow)[o] = v;
  }
}

/*
 * Order of the two passes ("shrink before you stretch", a rule of the geometry alone):
 *   src_h >= dst_h  columns first:  c_j = 0 ; c_j = fma(wy_r, src[row_r][j], c_j), r = 0..taps-1, for every source
 *                   element j of the row; then along the row e = 0 ; e = fma(wx_k, c[idx_k], e) over the even taps,
 *                   o likewise over the odd taps, v = e + o
 *   src_h <  dst_h  rows first (the definition of rounds 1-2, unchanged): e = wx_0 t_0 ; e = fma(wx_k, t_k, e), k even ;
 *                   o likewise ; h_r = e + o ; v = wy_0 h_0 ; v = fma(wy_r, h_r, v)
 * A vertical shrink filtered columns-first touches every source sample ONCE (one int->float conversion per sample,
 * row weights uniform along the row) and runs the gather along the row on dst_h rows instead of src_h: on the GPU
 * 2.16 row-filtered samples per output sample at 2:1 became 1.  The accumulators start at +0 in the columns-first form
 * (taps with weight zero are then exact no-ops), the float results differ from the rows-first order in the last bit;
 * the pin against NPP's output (tests/test_oracle_reference_pins.py, 45.6 dB through JPEG noise) does not see it.
 */
static int resize_plane_taps(const void* src, int src_pitch, int src_w, int src_h, void* dst,
                             int dst_pitch, int dst_w, int dst_h, int elem, int channels, int taps) {
  if (!src || !dst || src_w <= 0 || src_h <= 0 || dst_w <= 0 || dst_h <= 0)
    return VALI_ERR_INVALID_ARG;
  if ((elem != 1 && elem != 2 && elem != 4) || channels < 1 || channels > 3)
    return VALI_ERR_INVALID_ARG;
  const float scale_x = (float)src_w / (float)dst_w, scale_y = (float)src_h / (float)dst_h;
  if (src_h >= dst_h) { /* columns first */
    const int ne = src_w * channels;
    float* col = (float*)malloc((size_t)ne * sizeof(float));
    lz_tap_t* txs = (lz_tap_t*)malloc((size_t)dst_w * sizeof(lz_tap_t));
    if (!col || !txs) {
      free(col);
      free(txs);
      return VALI_ERR_INVALID_ARG;
    }
    for (int x = 0; x < dst_w; ++x)
      txs[x] = lz_make_tap(x, scale_x, src_w, taps);
    for (int y = 0; y < dst_h; ++y) {
      const lz_tap_t ty = lz_make_tap(y, scale_y, src_h, taps);
      uint8_t* drow = (uint8_t*)dst + (size_t)y * dst_pitch;
      for (int j = 0; j < ne; ++j)
        col[j] = 0.0f;
      for (int r = 0; r < taps; ++r) {
        const uint8_t* row = (const uint8_t*)src + (size_t)ty.idx[r] * src_pitch;
        for (int j = 0; j < ne; ++j)
          col[j] = fmaf(ty.w[r], rot_texel(row, j, elem), col[j]);
      }
      for (int x = 0; x < dst_w; ++x) {
        const lz_tap_t* tx = &txs[x];
        for (int ch = 0; ch < channels; ++ch) {
          float e = 0.0f, o = 0.0f;
          for (int k = 0; k < taps; k += 2) {
            e = fmaf(tx->w[k], col[tx->idx[k] * channels + ch], e);
            o = fmaf(tx->w[k + 1], col[tx->idx[k + 1] * channels + ch], o);
          }
          taps_store(drow, x * channels + ch, e + o, elem);
        }
      }
    }
    free(col);
    free(txs);
    return VALI_OK;
  }
  for (int y = 0; y < dst_h; ++y) {
    const lz_tap_t ty = lz_make_tap(y, scale_y, src_h, taps);
    uint8_t* drow = (uint8_t*)dst + (size_t)y * dst_pitch;
    for (int x = 0; x < dst_w; ++x) {
      const lz_tap_t tx = lz_make_tap(x, scale_x, src_w, taps);
      for (int ch = 0; ch < channels; ++ch) {
        float v = 0.0f;
        for (int r = 0; r < taps; ++r) {
          const uint8_t* row = (const uint8_t*)src + (size_t)ty.idx[r] * src_pitch;
          float he = tx.w[0] * rot_texel(row, tx.idx[0] * channels + ch, elem);
          float ho = tx.w[1] * rot_texel(row, tx.idx[1] * channels + ch, elem);
          for (int k = 2; k < taps; k += 2) {
            he = fmaf(tx.w[k], rot_texel(row, tx.idx[k] * channels + ch, elem), he);
            ho = fmaf(tx.w[k + 1], rot_texel(row, tx.idx[k + 1] * channels + ch, elem), ho);
          }
          const float hsum = he + ho;
          v = r == 0 ? ty.w[0] * hsum : fmaf(ty.w[r], hsum, v);
        }
        taps_store(drow, x * channels + ch, v, elem);
      }
    }
  }
  return VALI_OK;
}

int vali_oracle_resize_plane_lanczos(const void* src, int src_pitch, int src_w, int src_h,
                                     void* dst, int dst_pitch, int dst_w, int dst_h, int elem,
                                     int channels) {
  return resize_plane_taps(src, src_pitch, src_w, src_h, dst, dst_pitch, dst_w, dst_h, elem, channels, 6);
}

int vali_oracle_resize_plane_cubic(const void* src, int src_pitch, int src_w, int src_h, void* dst,
                                   int dst_pitch, int dst_w, int dst_h, int elem, int channels) {
  return resize_plane_taps(src, src_pitch, src_w, src_h, dst, dst_pitch, dst_w, dst_h, elem, channels, 4);
}
