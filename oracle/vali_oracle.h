/*
 * vali_oracle.h -- CPU restatement of the surface-processing hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under vali_amd/ may include, link, import or
 * call this.  Allowed users: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py (as the checker / the timed CPU baseline).
 *
 * Same descriptor types as include/vali_hip.h, but every pointer is a HOST
 * pointer.  Each function cites the reference lines whose behaviour it restates.
 *
 * PARITY STATUS (see DESIGN.md "Oracle pinning"):
 *   - the arithmetic of the reference's colour conversions, resize and rotate
 *     lives in NVIDIA NPP (closed source, CUDA Toolkit >= 11.2, dlopen'd at
 *     src/TC/src/LibNpp.cpp:20-51) and its CPU converter in FFmpeg n7.1
 *     libswscale (src/CMakeLists.txt:25-40); neither is under /root/reference
 *     and neither can be built here -> NPP-backed paths are "parity unpinned"
 *     beyond the published NPP matrices and the reference tests' PSNR >= 42 dB;
 *   - the UD (upsample+downscale) kernels are first-party reference code
 *     (src/TC/src/ResizeUtils.cu:21-158) and are restated exactly, with the
 *     output stage pinned against the reference's own golden files;
 *   - k*90-degree rotation is pinned against the reference's JPEG etalons.
 */
#ifndef VALI_ORACLE_H
#define VALI_ORACLE_H

#include <stdint.h>

#include "../include/vali_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* round-half-even + saturate: the quantiser of every colour kernel */
VALI_API uint8_t vali_oracle_q_u8(float v);

/*
 * Colour-matrix table.  variant:
 *   0 = NPP "YUV"      (nppiNV12ToRGB_8u_P2C3R, nppiYUV420ToRGB, nppiYUVToRGB;
 *                       BT.601 + JPEG in the reference)
 *   1 = NPP "709CSC"   (nppiNV12ToRGB_709CSC_8u_P2C3R; BT.709 + MPEG)
 *   2 = NPP "709HDTV"  (nppiNV12ToRGB_709HDTV_8u_P2C3R; BT.709 + JPEG, the default)
 *   3 = NPP "YCbCr"    (nppiYCbCr420ToRGB / nppiYCbCrToBGR; BT.601 + MPEG)
 * Values are the constants published in NVIDIA's NPP documentation (recalled,
 * see SURVEY.md Appendix A.3); selection logic restates
 * reference src/TC/src/TaskConvertSurface.cpp:128-149.
 */
VALI_API int vali_oracle_csc(int variant, vali_csc* out);

/* NV12 -> RGB/BGR/RGB_PLANAR u8.  dst->format picks the layout.
 * Restates nv12_rgb / nv12_bgr (TaskConvertSurface.cpp:61-156) with the NPP
 * per-pixel model: nearest chroma (one UV sample per 2x2 luma block). */
VALI_API int vali_oracle_nv12_to_rgb(const vali_surface* src, const vali_surface* dst,
                                     const vali_csc* csc);

/* same conversion over n frames using `threads` OpenMP threads (frames are
 * independent): the multi-core CPU baseline of bench.py */
VALI_API int vali_oracle_nv12_to_rgb_mt(const vali_surface* src, const vali_surface* dst,
                                        int n, const vali_csc* csc, int threads);

/*
 * UD: chroma upsample + resize (+ YUV->RGB) of NV12 / P10.
 * Restates RescaleConvertYUV<T> / RescaleConvertRGB<T> and their launcher Impl<T>
 * (reference: src/TC/src/ResizeUtils.cu:21-158) with CUDA's documented linear texture
 * filter (unnormalised coordinates, clamp addressing, normalised-float reads, 8-bit
 * fractional weights).  Supported pairs: UDSurface.cpp:117-133 (semi-planar rows).
 * Pinned by the reference's 640x360 goldens through their internal identities
 * (tests/test_oracle_ud.py); the goldens' INPUT frame is not available offline.
 */
VALI_API int vali_oracle_ud_nv12(const vali_surface* src, const vali_surface* dst);
/* the two output stages on their own (ResizeUtils.cu:45-54, 71-77) */
VALI_API void vali_oracle_ud_rgb_from_yuv(float ny, float nu, float nv, float* rgb);
VALI_API uint8_t vali_oracle_ud_store_u8(float normalised);

/* Generic format-pair converter (every pair vali_convert implements); see vali_oracle_cvt.c
 * for the per-pair restatement and its parity status. */
VALI_API int vali_oracle_convert(const vali_surface* src, const vali_surface* dst,
                                 const vali_cvt_params* params);
/* RGB -> YUV matrices: 0 = nppiRGBToYUV (+ row 0 = nppiRGBToGray), 1 = nppiRGBToYCbCr */
VALI_API int vali_oracle_rgb2yuv(int variant, float m[3][4]);

/* Plane rotation, NPP nppiRotate model with bilinear interpolation
 * (reference: src/TC/src/RotateSurface.cpp:22-125).  Pinned for 90/180/270 degrees by the
 * reference's rotation etalons (tests/test_oracle_rotate.py); other angles: parity unpinned. */
VALI_API int vali_oracle_rotate_plane(const void* src, int src_pitch, int src_w, int src_h,
                                      void* dst, int dst_pitch, int dst_w, int dst_h, int elem,
                                      int channels, double angle, double shift_x, double shift_y);

/* Bilinear plane resize on NPP's sampling grid src = dst * src_size / dst_size.  `channels`
 * interleaved channels (2 = the UV plane of NV12).  The reference's resizer is NPP Lanczos
 * (src/TC/src/TaskResizeSurface.cpp:67) whose input golden is missing: interpolation weights
 * unpinned; the sampling GEOMETRY is pinned by test_small.nv12 (tests/test_oracle_resize.py). */
VALI_API int vali_oracle_resize_plane(const void* src, int src_pitch, int src_w, int src_h,
                                      void* dst, int dst_pitch, int dst_w, int dst_h, int elem,
                                      int channels);

/* AVX2 form of vali_oracle_nv12_to_rgb for packed RGB / BGR (vali_oracle_simd.c): the same
 * operations eight pixels at a time, bit-identical output; exists only so that bench.py's
 * cpu_baseline measures SIMD host code, as the reference's libswscale path is. */
VALI_API int vali_oracle_nv12_to_rgb_simd(const vali_surface* src, const vali_surface* dst,
                                          const vali_csc* csc);
VALI_API int vali_oracle_nv12_to_rgb_simd_mt(const vali_surface* src, const vali_surface* dst, int n,
                                             const vali_csc* csc, int threads);
/* bench.py's CPU baseline: `threads` OpenMP threads, each converting its own first-touched copy of `nv12` for `seconds`;
 * returns the frames converted by all threads together (< 0: bad arguments), *elapsed = the longest thread's time;
 * thread i pins itself to cpus[i % n_cpus] (cpus == NULL: no pinning) */
VALI_API long long vali_oracle_nv12_to_rgb_bench(const uint8_t* nv12, int width, int height, const vali_csc* csc,
                                                 int threads, double seconds, int simd, double* elapsed,
                                                 const int* cpus, int n_cpus);

/* Lanczos-3 (6x6 taps, interpolating, same sampling grid) variant: the reference's
 * NPPI_INTER_LANCZOS restated with this build's own tap arithmetic (weights from fixed
 * polynomials, fma accumulation order documented in vali_oracle.c) -- parity unpinned against
 * NPP beyond the geometry; on integer scale factors it is the same point sample. */
VALI_API void vali_oracle_lanczos3_weights(float a, float w[6]);
VALI_API int vali_oracle_resize_plane_lanczos(const void* src, int src_pitch, int src_w, int src_h,
                                              void* dst, int dst_pitch, int dst_w, int dst_h,
                                              int elem, int channels);

/* Bicubic (Keys / Catmull-Rom, a = -1/2; taps i-1 .. i+2) on the same grid: the mode
 * BASELINE.json's north_star names next to bilinear (NPPI_INTER_CUBIC); parity unpinned. */
VALI_API void vali_oracle_cubic_weights(float a, float w[4]);
VALI_API int vali_oracle_resize_plane_cubic(const void* src, int src_pitch, int src_w, int src_h,
                                            void* dst, int dst_pitch, int dst_w, int dst_h,
                                            int elem, int channels);

#ifdef __cplusplus
}
#endif
#endif
