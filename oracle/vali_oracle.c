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
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* --------------------------------------------------------------------------
 * quantiser: round to nearest, ties to even, saturate to [0, 255].
 * (NPP's rounding mode is not documented for these conversions -- SURVEY.md
 * A.3 -- nearest-even is this build's definition.)
 * -------------------------------------------------------------------------- */
uint8_t vali_oracle_q_u8(float v) {
  float r = rintf(v); /* default rounding mode = nearest even */
  if (!(r > 0.0f))
    r = 0.0f; /* also maps NaN to 0 like fmaxf(NaN,0) */
  if (r > 255.0f)
    r = 255.0f;
  return (uint8_t)r;
}

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
  for (int y = 0; y < H; ++y) {
    const uint8_t* yrow = py + (size_t)y * src->pitch[0];
    const uint8_t* crow = puv + (size_t)(y / 2) * src->pitch[1];
    for (int x = 0; x < W; ++x) {
      const uint8_t* c = crow + (x & ~1);
      const chroma_term t = chroma((float)c[0], (float)c[1], csc);
      const float yf = luma((float)yrow[x], csc);
      const uint8_t R = vali_oracle_q_u8(yf + t.rv), G = vali_oracle_q_u8(yf + t.guv),
                    B = vali_oracle_q_u8(yf + t.bu);
      if (dst->format == VALI_FMT_RGB_PLANAR) {
        ((uint8_t*)dst->plane[0])[(size_t)y * dst->pitch[0] + x] = R;
        ((uint8_t*)dst->plane[1])[(size_t)y * dst->pitch[0] + x] = G;
        ((uint8_t*)dst->plane[2])[(size_t)y * dst->pitch[0] + x] = B;
      } else {
        uint8_t* q = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0] + (size_t)x * 3;
        q[0] = dst->format == VALI_FMT_RGB ? R : B;
        q[1] = G;
        q[2] = dst->format == VALI_FMT_RGB ? B : R;
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
