/*
 * vali_oracle_cvt.c -- CPU restatement of the generic format-pair converter
 * (vali_convert, vali_amd/csrc/cvt_generic.hip).  TEST INFRASTRUCTURE ONLY (see
 * vali_oracle.h).  One scalar routine covers every pair; per pair it restates the NPP call
 * the reference makes (reference: src/TC/src/TaskConvertSurface.cpp:158-962):
 *
 *   YUV -> RGB  : NPP colour model with the published constants (vali_oracle_csc), chroma of
 *                 4:2:0 sources taken from the pixel's 2x2 block (nearest), as in
 *                 vali_oracle_nv12_to_rgb.
 *   RGB -> YUV  : out_c = fma(kB, B, fma(kG, G, fma(kR, R, offset))), round-half-even,
 *                 saturate.  Constants (vali_oracle_rgb2yuv) are the ones NVIDIA documents
 *                 for nppiRGBToYUV / nppiRGBToYCbCr / nppiRGBToGray (recalled).
 *                 4:2:0 chroma = mean of the four un-rounded values of the 2x2 block,
 *                 ((c00 + c01) + (c10 + c11)) * 0.25 -- NPP's subsampling filter is not
 *                 published; this is the build's definition ("parity unpinned").
 *   copies / (de)interleaves / channel swaps: exact.
 *   P10/P12 -> NV12: min(255, (v + 128) >> 8): nppiDivC_16u_C1RSfs(256, scale 0) rounds to
 *                 nearest, nppiConvert_16u8u saturates (rounding mode unpinned: golden missing).
 *   RGB -> RGB_32F: v / 255.0f (nppiScale_8u32f_C3R with [0,1]).
 */
#include "vali_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

enum { K_NV12, K_YUV420, K_YUV444, K_RGB, K_BGR, K_RGBP, K_Y, K_NONE };

static int kind_of(int fmt) {
  switch (fmt) {
  case VALI_FMT_NV12: return K_NV12;
  case VALI_FMT_YUV420: return K_YUV420;
  case VALI_FMT_YUV444: return K_YUV444;
  case VALI_FMT_RGB: return K_RGB;
  case VALI_FMT_BGR: return K_BGR;
  case VALI_FMT_RGB_PLANAR: return K_RGBP;
  case VALI_FMT_Y: return K_Y;
  default: return K_NONE;
  }
}
static int is420(int k) { return k == K_NV12 || k == K_YUV420; }
static int isyuv(int k) { return k == K_NV12 || k == K_YUV420 || k == K_YUV444; }
static int isrgb(int k) { return k == K_RGB || k == K_BGR || k == K_RGBP; }

int vali_oracle_rgb2yuv(int variant, float m[3][4]) {
  /* 0: nppiRGBToYUV (JPEG range) ; 1: nppiRGBToYCbCr (MPEG range).  Row 0 of variant 0 is
   * also nppiRGBToGray. */
  static const float yuv[3][4] = {{0.299f, 0.587f, 0.114f, 0.0f},
                                  {-0.147f, -0.289f, 0.436f, 128.0f},
                                  {0.615f, -0.515f, -0.100f, 128.0f}};
  static const float ycc[3][4] = {{0.257f, 0.504f, 0.098f, 16.0f},
                                  {-0.148f, -0.291f, 0.439f, 128.0f},
                                  {0.439f, -0.368f, -0.071f, 128.0f}};
  if (variant == 0) memcpy(m, yuv, sizeof(yuv));
  else if (variant == 1) memcpy(m, ycc, sizeof(ycc));
  else return VALI_ERR_UNSUPPORTED;
  return VALI_OK;
}

static void px_read(const vali_surface* s, int k, int x, int y, float c[3]) {
  const uint8_t* p0 = (const uint8_t*)s->plane[0];
  if (k == K_RGB || k == K_BGR) {
    const uint8_t* q = p0 + (size_t)y * s->pitch[0] + (size_t)x * 3;
    c[0] = q[k == K_RGB ? 0 : 2]; c[1] = q[1]; c[2] = q[k == K_RGB ? 2 : 0];
  } else if (k == K_RGBP || k == K_YUV444) {
    for (int i = 0; i < 3; ++i) c[i] = ((const uint8_t*)s->plane[i])[(size_t)y * s->pitch[i] + x];
  } else if (k == K_NV12) {
    c[0] = p0[(size_t)y * s->pitch[0] + x];
    const uint8_t* q = (const uint8_t*)s->plane[1] + (size_t)(y / 2) * s->pitch[1] + (x & ~1);
    c[1] = q[0]; c[2] = q[1];
  } else if (k == K_YUV420) {
    c[0] = p0[(size_t)y * s->pitch[0] + x];
    c[1] = ((const uint8_t*)s->plane[1])[(size_t)(y / 2) * s->pitch[1] + x / 2];
    c[2] = ((const uint8_t*)s->plane[2])[(size_t)(y / 2) * s->pitch[2] + x / 2];
  } else {
    c[0] = p0[(size_t)y * s->pitch[0] + x]; c[1] = c[2] = 128.0f;
  }
}

static float dot_rgb(const float m[4], float r, float g, float b) {
  return fmaf(m[2], b, fmaf(m[1], g, fmaf(m[0], r, m[3])));
}

static void xform(int sk, int dk, const vali_cvt_params* p, const float c[3], float o[3]) {
  if (isyuv(sk) && isrgb(dk)) {
    const vali_csc* k = &p->yuv2rgb;
    const float uc = c[1] - 128.0f, vc = c[2] - 128.0f;
    const float rv = k->crv * vc, guv = fmaf(k->cgu, uc, k->cgv * vc), bu = k->cbu * uc;
    const float yf = k->cy * (c[0] - k->y0);
    o[0] = yf + rv; o[1] = yf + guv; o[2] = yf + bu;
  } else if (isrgb(sk) && (isyuv(dk) || dk == K_Y)) {
    for (int i = 0; i < 3; ++i) o[i] = dot_rgb(p->rgb2yuv[i], c[0], c[1], c[2]);
  } else {
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
  }
}

static int convert_elem(const vali_surface* src, const vali_surface* dst) {
  const int W = src->width, H = src->height;
  if ((src->format == VALI_FMT_P10 || src->format == VALI_FMT_P12) && dst->format == VALI_FMT_NV12) {
    const int rows = H + (H + 1) / 2;
    for (int y = 0; y < rows; ++y) {
      const uint16_t* s = (const uint16_t*)((const uint8_t*)src->plane[0] + (size_t)y * src->pitch[0]);
      uint8_t* d = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0];
      for (int x = 0; x < W; ++x) {
        const uint32_t v = ((uint32_t)s[x] + 128u) >> 8;
        d[x] = (uint8_t)(v > 255u ? 255u : v);
      }
    }
    return 1;
  }
  if (src->format == VALI_FMT_RGB && dst->format == VALI_FMT_RGB_32F) {
    for (int y = 0; y < H; ++y) {
      const uint8_t* s = (const uint8_t*)src->plane[0] + (size_t)y * src->pitch[0];
      float* d = (float*)((uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0]);
      for (int x = 0; x < 3 * W; ++x) d[x] = (float)s[x] / 255.0f;
    }
    return 1;
  }
  if (src->format == VALI_FMT_RGB_32F && dst->format == VALI_FMT_RGB_32F_PLANAR) {
    for (int y = 0; y < H; ++y) {
      const float* s = (const float*)((const uint8_t*)src->plane[0] + (size_t)y * src->pitch[0]);
      for (int c = 0; c < 3; ++c) {
        float* d = (float*)((uint8_t*)dst->plane[c] + (size_t)y * dst->pitch[c]);
        for (int x = 0; x < W; ++x) d[x] = s[3 * x + c];
      }
    }
    return 1;
  }
  return 0;
}

int vali_oracle_convert(const vali_surface* src, const vali_surface* dst,
                        const vali_cvt_params* params) {
  if (!src || !dst || !params)
    return VALI_ERR_INVALID_ARG;
  if (src->width != dst->width || src->height != dst->height || src->width <= 0 || src->height <= 0)
    return VALI_ERR_INVALID_ARG;
  if (convert_elem(src, dst))
    return VALI_OK;
  const int sk = kind_of(src->format), dk = kind_of(dst->format);
  if (sk == K_NONE || dk == K_NONE)
    return VALI_ERR_UNSUPPORTED;
  const int W = src->width, H = src->height;
  for (int qy = 0; qy < (H + 1) / 2; ++qy)
    for (int qx = 0; qx < (W + 1) / 2; ++qx) {
      float cu[4], cv[4];
      for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
          const int x = qx * 2 + dx, y = qy * 2 + dy;
          const int in = x < W && y < H;
          float c[3] = {0.0f, 128.0f, 128.0f}, o[3];
          if (in) px_read(src, sk, x, y, c);
          xform(sk, dk, params, c, o);
          cu[dy * 2 + dx] = o[1]; cv[dy * 2 + dx] = o[2];
          if (!in) continue;
          const uint8_t q0 = vali_oracle_q_u8(o[0]), q1 = vali_oracle_q_u8(o[1]), q2 = vali_oracle_q_u8(o[2]);
          if (dk == K_RGB || dk == K_BGR) {
            uint8_t* w = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0] + (size_t)x * 3;
            w[dk == K_RGB ? 0 : 2] = q0; w[1] = q1; w[dk == K_RGB ? 2 : 0] = q2;
          } else if (dk == K_RGBP || dk == K_YUV444) {
            ((uint8_t*)dst->plane[0])[(size_t)y * dst->pitch[0] + x] = q0;
            ((uint8_t*)dst->plane[1])[(size_t)y * dst->pitch[1] + x] = q1;
            ((uint8_t*)dst->plane[2])[(size_t)y * dst->pitch[2] + x] = q2;
          } else {
            ((uint8_t*)dst->plane[0])[(size_t)y * dst->pitch[0] + x] = q0;
          }
        }
      if (is420(dk)) {
        float fu, fv;
        if (is420(sk)) { fu = cu[0]; fv = cv[0]; }
        else {
          fu = ((cu[0] + cu[1]) + (cu[2] + cu[3])) * 0.25f;
          fv = ((cv[0] + cv[1]) + (cv[2] + cv[3])) * 0.25f;
        }
        const uint8_t qu = vali_oracle_q_u8(fu), qv = vali_oracle_q_u8(fv);
        if (dk == K_NV12) {
          uint8_t* w = (uint8_t*)dst->plane[1] + (size_t)qy * dst->pitch[1] + qx * 2;
          w[0] = qu; w[1] = qv;
        } else {
          ((uint8_t*)dst->plane[1])[(size_t)qy * dst->pitch[1] + qx] = qu;
          ((uint8_t*)dst->plane[2])[(size_t)qy * dst->pitch[2] + qx] = qv;
        }
      }
    }
  return VALI_OK;
}
