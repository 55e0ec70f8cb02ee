// Does gfx950 convert in the buffer path?  buffer_load_format_xyzw through a descriptor of format 8_8_8_8 USCALED: 4 source
// bytes -> 4 floats per lane, no v_cvt.  Checks values (aligned and byte-shifted addresses) and measures a streaming pass
// (sum of all bytes of a 64 MiB buffer) against plain loads + v_cvt_f32_ubyteN.
// build: hipcc --offload-arch=gfx950 -O3 tools/exp/typed_load.hip -o tools/exp/typed_load
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v4f typed4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  v4f v;
  asm volatile("buffer_load_format_xyzw %0, %1, %2, %3 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(voff), "s"(r), "s"(soff) : "memory");
  return v;
}
__global__ void k_check(const unsigned char* p, float* out, int shift, unsigned fmt) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), (short)0, (int)0xffffffffu, (int)fmt);
  const v4f v = typed4(r, (int)threadIdx.x * 4 + shift, 0);
  out[threadIdx.x * 4 + 0] = v.x; out[threadIdx.x * 4 + 1] = v.y; out[threadIdx.x * 4 + 2] = v.z; out[threadIdx.x * 4 + 3] = v.w;
}
// streaming: every lane 8 bytes per row, `rows` rows, sums
template <int MODE> __global__ void __launch_bounds__(256) k_stream(const unsigned char* p, float* out, int pitch, int rows, unsigned fmt) {
  const int lane_off = (int)(blockIdx.x * 256 + threadIdx.x) * 8;
  float acc = 0.f;
  if constexpr (MODE == 0) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), (short)0, (int)0xffffffffu, 0x00020000);
#pragma unroll 4
    for (int y = 0; y < rows; ++y) {
      const v2u w = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, y * pitch, 0);
      acc += (float)(w.x & 255u) + (float)((w.x >> 8) & 255u) + (float)((w.x >> 16) & 255u) + (float)(w.x >> 24) +
             (float)(w.y & 255u) + (float)((w.y >> 8) & 255u) + (float)((w.y >> 16) & 255u) + (float)(w.y >> 24);
    }
  } else {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), (short)0, (int)0xffffffffu, (int)fmt);
#pragma unroll 1
    for (int y = 0; y < rows; y += 4) {
      v4f a[4], b[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        asm volatile("buffer_load_format_xyzw %0, %1, %2, %3 offen" : "=v"(a[d]) : "v"(lane_off), "s"(r), "s"((y + d) * pitch));
        asm volatile("buffer_load_format_xyzw %0, %1, %2, %3 offen offset:4" : "=v"(b[d]) : "v"(lane_off), "s"(r), "s"((y + d) * pitch));
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
#pragma unroll
      for (int d = 0; d < 4; ++d)
        acc += a[d].x + a[d].y + a[d].z + a[d].w + b[d].x + b[d].y + b[d].z + b[d].w;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  const unsigned fmt = 0x50000u | 0x2000u | 0xFACu; // DATA_FORMAT 8_8_8_8 (10), NUM_FORMAT USCALED (2), DST_SEL x y z w
  unsigned char* d; float* o;
  const int pitch = 8192 * 8, rows = 1024; // 64 MiB: 8192 lanes x 8 bytes x 1024 rows
  CK(hipMalloc(&d, (size_t)pitch * rows + 64)); CK(hipMalloc(&o, 1 << 20));
  std::vector<unsigned char> h((size_t)pitch * rows + 64);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 2654435761u) >> 13);
  CK(hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice));
  for (int shift : {0, 1, 2, 3, 5}) {
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d, o, shift, fmt);
    float r[256]; CK(hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != (float)h[i + shift];
    printf("shift %d: %d of 256 wrong (first values %g %g %g %g, bytes %d %d %d %d)\n", shift, bad, r[0], r[1], r[2], r[3], h[shift], h[shift + 1], h[shift + 2], h[shift + 3]);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int mode = 0; mode < 2; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a));
      if (mode == 0) hipLaunchKernelGGL(k_stream<0>, dim3(32), dim3(256), 0, 0, d, o, pitch, rows, fmt);
      else hipLaunchKernelGGL(k_stream<1>, dim3(32), dim3(256), 0, 0, d, o, pitch, rows, fmt);
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      std::vector<float> r(8192); CK(hipMemcpy(r.data(), o, 8192 * 4, hipMemcpyDeviceToHost));
      double s = 0; for (float v : r) s += v;
      printf("%s: %.3f ms, %.2f TB/s, sum %.0f\n", mode ? "typed (2 x format_xyzw per row)" : "plain b64 + 8 cvt", ms, (double)pitch * rows / ms / 1e9, s);
    }
  return 0;
}
