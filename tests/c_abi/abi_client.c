/* A plain C99 client of include/vali_hip.h: what the reference's C++ side (or any FFI) sees.
 * No Python, no torch, no C++: allocate pitched planes, upload an NV12 frame, convert it to
 * RGB with the BT.709 limited-range coefficients, resize it, download both, write them out.
 *   usage: abi_client <in.nv12> <width> <height> <out.rgb> <out_half.nv12>
 * Exit code 0 on success; prints vali_last_error() otherwise.  tests/test_gpu_c_abi.py
 * compares the two outputs with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vali_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != VALI_OK) {                                                        \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, vali_last_error());         \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 6) {
    fprintf(stderr, "usage: %s in.nv12 width height out.rgb out_half.nv12\n", argv[0]);
    return 2;
  }
  const int w = atoi(argv[2]), h = atoi(argv[3]), dev = 0;
  const size_t in_bytes = (size_t)w * h * 3 / 2, rgb_bytes = (size_t)w * h * 3;
  const int hw = w / 2, hh = h / 2;
  const size_t half_bytes = (size_t)hw * hh * 3 / 2;
  unsigned char* host = (unsigned char*)malloc(in_bytes);
  unsigned char* out = (unsigned char*)malloc(rgb_bytes);
  unsigned char* half = (unsigned char*)malloc(half_bytes);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(host, 1, in_bytes, f) != in_bytes) {
    fprintf(stderr, "cannot read %s\n", argv[1]);
    return 2;
  }
  fclose(f);

  int count = 0;
  CHECK(vali_device_count(&count));
  if (count < 1) {
    fprintf(stderr, "no device\n");
    return 3;
  }
  vali_stream_t stream = NULL;
  CHECK(vali_stream_create(dev, &stream));

  /* NV12 = one W x 1.5H plane (SurfaceNV12, Surfaces.cpp); plane[1] = the UV rows */
  void *d_nv12 = NULL, *d_rgb = NULL, *d_half = NULL;
  size_t p_nv12 = 0, p_rgb = 0, p_half = 0;
  CHECK(vali_mem_alloc_pitch(dev, (size_t)w, (size_t)h * 3 / 2, &d_nv12, &p_nv12));
  CHECK(vali_mem_alloc_pitch(dev, (size_t)w * 3, (size_t)h, &d_rgb, &p_rgb));
  CHECK(vali_mem_alloc_pitch(dev, (size_t)hw, (size_t)hh * 3 / 2, &d_half, &p_half));
  CHECK(vali_memcpy2d_async(dev, d_nv12, p_nv12, host, (size_t)w, (size_t)w, (size_t)h * 3 / 2, 0, stream));

  vali_surface src, dst, small;
  memset(&src, 0, sizeof src); memset(&dst, 0, sizeof dst); memset(&small, 0, sizeof small);
  src.plane[0] = d_nv12; src.plane[1] = (char*)d_nv12 + (size_t)h * p_nv12;
  src.pitch[0] = src.pitch[1] = (int)p_nv12; src.width = w; src.height = h; src.format = VALI_FMT_NV12;
  dst.plane[0] = d_rgb; dst.pitch[0] = (int)p_rgb; dst.width = w; dst.height = h; dst.format = VALI_FMT_RGB;
  small.plane[0] = d_half; small.plane[1] = (char*)d_half + (size_t)hh * p_half;
  small.pitch[0] = small.pitch[1] = (int)p_half; small.width = hw; small.height = hh; small.format = VALI_FMT_NV12;

  /* nppiNV12ToRGB_709CSC_8u_P2C3R coefficients (TaskConvertSurface.cpp:128-136) */
  vali_csc k;
  memset(&k, 0, sizeof k);
  k.y0 = 16.0f; k.cy = 1.164f; k.crv = 1.793f; k.cgu = -0.213f; k.cgv = -0.533f; k.cbu = 2.112f;
  CHECK(vali_nv12_to_rgb(&src, &dst, &k, stream));
  CHECK(vali_resize(&src, &small, VALI_INTERP_LANCZOS, stream));
  CHECK(vali_memcpy2d_async(dev, out, (size_t)w * 3, d_rgb, p_rgb, (size_t)w * 3, (size_t)h, 1, stream));
  CHECK(vali_memcpy2d_async(dev, half, (size_t)hw, d_half, p_half, (size_t)hw, (size_t)hh * 3 / 2, 1, stream));
  CHECK(vali_stream_sync(dev, stream));

  f = fopen(argv[4], "wb");
  if (!f || fwrite(out, 1, rgb_bytes, f) != rgb_bytes) return 2;
  fclose(f);
  f = fopen(argv[5], "wb");
  if (!f || fwrite(half, 1, half_bytes, f) != half_bytes) return 2;
  fclose(f);

  /* error behaviour: unsupported interpolation / null arguments come back as codes, not crashes */
  if (vali_resize(&src, &small, 12345, stream) != VALI_ERR_UNSUPPORTED) return 4;
  if (vali_nv12_to_rgb(NULL, &dst, &k, stream) != VALI_ERR_INVALID_ARG) return 4;

  vali_mem_free(dev, d_nv12); vali_mem_free(dev, d_rgb); vali_mem_free(dev, d_half);
  vali_stream_destroy(dev, stream);
  free(host); free(out); free(half);
  printf("ok %s\n", vali_version());
  return 0;
}
