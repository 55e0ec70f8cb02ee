/* AVX2 form of the NV12 -> packed RGB restatement: TEST INFRASTRUCTURE, used only as the CPU
 * baseline of bench.py (`cpu_baseline`) -- a scalar loop undersells what the host cores can do
 * (the reference's own CPU path, FFmpeg libswscale, is SIMD code).
 *
 * Eight pixels per iteration with exactly the operations of vali_oracle_nv12_to_rgb
 * (vali_oracle.c: chroma(), luma(), q_u8()), in the same order, in IEEE single precision:
 *   uc = U - 128 ; vc = V - 128 ; rv = crv * vc ; guv = fma(cgu, uc, cgv * vc) ; bu = cbu * uc
 *   yf = cy * (Y - y0) ; R = yf + rv ; G = yf + guv ; B = yf + bu
 *   q = round-half-even, clamp to [0, 255]
 * so its output is bit-identical to the scalar restatement (tests/test_oracle_color.py checks it
 * on random and gradient frames); widths that are not a multiple of 8 finish in the scalar loop.
 */
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include "vali_oracle.h"

static inline __m256 q_ps(__m256 v) {
  __m256 r = _mm256_round_ps(v, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
  r = _mm256_max_ps(r, _mm256_setzero_ps()); /* max(NaN, 0) = 0 like the scalar !(r > 0) */
  return _mm256_min_ps(r, _mm256_set1_ps(255.0f));
}

/* eight float lanes -> eight bytes */
static inline void store8_u8(uint8_t* dst, __m256 v) {
  const __m256i i32 = _mm256_cvttps_epi32(v);
  const __m128i lo = _mm256_castsi256_si128(i32), hi = _mm256_extracti128_si256(i32, 1);
  const __m128i i16 = _mm_packs_epi32(lo, hi);
  const __m128i i8 = _mm_packus_epi16(i16, i16);
  _mm_storel_epi64((__m128i*)dst, i8);
}

int vali_oracle_nv12_to_rgb_simd(const vali_surface* src, const vali_surface* dst, const vali_csc* csc) {
  if (!src || !dst || !csc)
    return VALI_ERR_INVALID_ARG;
  if (src->format != VALI_FMT_NV12 || (dst->format != VALI_FMT_RGB && dst->format != VALI_FMT_BGR))
    return VALI_ERR_UNSUPPORTED;
  if (src->width != dst->width || src->height != dst->height || src->width <= 0 || src->height <= 0)
    return VALI_ERR_INVALID_ARG;
  const int W = src->width, H = src->height, W8 = W & ~7;
  const int rgb = dst->format == VALI_FMT_RGB;
  const __m256 y0 = _mm256_set1_ps(csc->y0), cy = _mm256_set1_ps(csc->cy), crv = _mm256_set1_ps(csc->crv),
               cgu = _mm256_set1_ps(csc->cgu), cgv = _mm256_set1_ps(csc->cgv), cbu = _mm256_set1_ps(csc->cbu),
               c128 = _mm256_set1_ps(128.0f);
  /* U0 V0 U1 V1 ... (8 bytes = 4 chroma pairs) -> U = u0 u0 u1 u1 u2 u2 u3 u3, V likewise */
  const __m256i sel_u = _mm256_setr_epi32(0, 0, 2, 2, 4, 4, 6, 6), sel_v = _mm256_setr_epi32(1, 1, 3, 3, 5, 5, 7, 7);
  for (int y = 0; y < H; ++y) {
    const uint8_t* yrow = (const uint8_t*)src->plane[0] + (size_t)y * src->pitch[0];
    const uint8_t* crow = (const uint8_t*)src->plane[1] + (size_t)(y / 2) * src->pitch[1];
    uint8_t* out = (uint8_t*)dst->plane[0] + (size_t)y * dst->pitch[0];
    for (int x = 0; x < W8; x += 8) {
      const __m256 Y = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)(yrow + x))));
      const __m256 UV = _mm256_cvtepi32_ps(_mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i*)(crow + x))));
      const __m256 uc = _mm256_sub_ps(_mm256_permutevar8x32_ps(UV, sel_u), c128);
      const __m256 vc = _mm256_sub_ps(_mm256_permutevar8x32_ps(UV, sel_v), c128);
      const __m256 rv = _mm256_mul_ps(crv, vc);
      const __m256 guv = _mm256_fmadd_ps(cgu, uc, _mm256_mul_ps(cgv, vc));
      const __m256 bu = _mm256_mul_ps(cbu, uc);
      const __m256 yf = _mm256_mul_ps(cy, _mm256_sub_ps(Y, y0));
      uint8_t r[8], g[8], b[8];
      store8_u8(r, q_ps(_mm256_add_ps(yf, rv)));
      store8_u8(g, q_ps(_mm256_add_ps(yf, guv)));
      store8_u8(b, q_ps(_mm256_add_ps(yf, bu)));
      uint8_t* o = out + 3 * x;
      const uint8_t* first = rgb ? r : b;
      const uint8_t* last = rgb ? b : r;
      for (int k = 0; k < 8; ++k) {
        o[3 * k] = first[k];
        o[3 * k + 1] = g[k];
        o[3 * k + 2] = last[k];
      }
    }
  }
  if (W8 != W) { /* ragged right edge: the scalar restatement on a view of the last columns */
    vali_surface s = *src, d = *dst;
    /* columns W8 .. W-1: W8 is even, so the chroma pairing is unchanged */
    s.plane[0] = (uint8_t*)src->plane[0] + W8;
    s.plane[1] = (uint8_t*)src->plane[1] + W8;
    d.plane[0] = (uint8_t*)dst->plane[0] + 3 * (size_t)W8;
    s.width = d.width = W - W8;
    return vali_oracle_nv12_to_rgb(&s, &d, csc);
  }
  return VALI_OK;
}

int vali_oracle_nv12_to_rgb_simd_mt(const vali_surface* src, const vali_surface* dst, int n,
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
    const int r = vali_oracle_nv12_to_rgb_simd(&src[i], &dst[i], csc);
    if (r != VALI_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
      rc = r;
    }
  }
  return rc;
}

/*
 * The CPU baseline of bench.py, NUMA-fair: every OpenMP thread converts ITS OWN frame -- source copied from the template
 * and destination first-touched by the thread itself, so both live on the thread's memory node -- over and over for
 * `seconds`, and the function returns the total number of frames converted.  (Round 2 handed all threads the same
 * read-only inputs and outputs first-touched by one thread: 20 % parallel efficiency on a 2-socket host, the second
 * socket's threads pulling every byte across the fabric.)  Thread i pins itself to the i-th CPU of `cpus` (n_cpus
 * entries; NULL: no pinning) before it touches its buffers; the calling thread's own affinity is restored on return.
 * (Not OMP_PROC_BIND: that also binds the process's INITIAL thread -- the one that launches GPU work -- to one CPU.)
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <sched.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

long long vali_oracle_nv12_to_rgb_bench(const uint8_t* nv12, int width, int height, const vali_csc* csc, int threads,
                                         double seconds, int simd, double* elapsed, const int* cpus, int n_cpus) {
  cpu_set_t original;
  const int have_original = sched_getaffinity(0, sizeof original, &original) == 0;
  if (!nv12 || !csc || width <= 0 || height <= 0 || (width & 1) || (height & 1))
    return -1;
  long long total = 0;
  double t_all = 0.0;
  const size_t src_bytes = (size_t)width * height * 3 / 2, dst_bytes = (size_t)width * height * 3;
#ifdef _OPENMP
  if (threads < 1)
    threads = 1;
#pragma omp parallel num_threads(threads) reduction(+ : total) reduction(max : t_all)
#endif
  {
#ifdef _OPENMP
    if (cpus && n_cpus > 0) {
      cpu_set_t one;
      CPU_ZERO(&one);
      CPU_SET(cpus[omp_get_thread_num() % n_cpus], &one);
      (void)sched_setaffinity(0, sizeof one, &one);
    }
#endif
    uint8_t* in = (uint8_t*)malloc(src_bytes);
    uint8_t* out = (uint8_t*)malloc(dst_bytes);
    long long mine = 0;
    if (in && out) {
      memcpy(in, nv12, src_bytes);   /* first touch by the thread that will read it */
      memset(out, 0, dst_bytes);
      vali_surface s, d;
      memset(&s, 0, sizeof s);
      memset(&d, 0, sizeof d);
      s.plane[0] = in; s.plane[1] = in + (size_t)width * height; s.pitch[0] = s.pitch[1] = width;
      s.width = width; s.height = height; s.format = VALI_FMT_NV12;
      d.plane[0] = out; d.pitch[0] = 3 * width; d.width = width; d.height = height; d.format = VALI_FMT_RGB;
      (simd ? vali_oracle_nv12_to_rgb_simd : vali_oracle_nv12_to_rgb)(&s, &d, csc); /* untimed: page tables, caches */
#ifdef _OPENMP
#pragma omp barrier
      const double t0 = omp_get_wtime();
      double t = t0;
      while (t - t0 < seconds) {
        (simd ? vali_oracle_nv12_to_rgb_simd : vali_oracle_nv12_to_rgb)(&s, &d, csc);
        ++mine;
        t = omp_get_wtime();
      }
      t_all = t - t0;
#else
      (void)seconds;
      mine = 1;
#endif
    }
    free(in);
    free(out);
    total += mine;
  }
  if (have_original)
    (void)sched_setaffinity(0, sizeof original, &original);
  if (elapsed)
    *elapsed = t_all;
  return total;
}
