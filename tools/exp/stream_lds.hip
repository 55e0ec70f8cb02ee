// Does LDS traffic hide under a wave's global-memory stream on gfx950?  (profiles/r04_lanczos.md: in the columns-first
// Lanczos kernel it ADDS.)  A wave walks ROWS rows of a 2160p-sized plane, 8 bytes per lane and row (the walk of
// resize_cols.hip: 4 rows in flight, 8 conversions + 12 packed FMAs per row), and per row issues K LDS instructions of one
// kind whose results feed nothing but a final checksum.  MODE: 0 none, 1 ds_read_b64 waited for per row, 2 ds_read_b64
// waited for once per 4 rows, 3 ds_write_b64, 4 ds_write_b128 (K / 2 of them), 5 VALU filler of the same instruction
// count instead (v_pk_fma_f32).   hipcc --offload-arch=gfx950 -O3 -o stream_lds stream_lds.hip ; ./stream_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));

constexpr int W = 3840, H = 2160, ROWS = 96, D = 4;

template <int MODE, int K>
__global__ void __launch_bounds__(256) k(const unsigned char* __restrict__ src, float* __restrict__ out, int frames) {
  __shared__ __attribute__((aligned(16))) float lds[4][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_x = W / 512, tiles_y = (H + 4 * ROWS - 1) / (4 * ROWS);
  const int per = tiles_x * tiles_y;
  const int t = blockIdx.x % per, f = blockIdx.x / per;
  if (f >= frames) return;
  const int tx = t % tiles_x, ty = t / tiles_x;
  const int y0 = (ty * 4 + wave) * ROWS;
  if (y0 >= H) return;
  const unsigned char* p = src + (size_t)f * W * H + (size_t)tx * 512 + 8 * lane;
  const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)&lds[wave][0] + 8u * lane;
  v2f acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (v2f){0.f, 0.f};
  v2f sink = (v2f){0.f, 0.f};
  v2u pf[D];
  for (int d = 0; d < D; ++d) { pf[d] = *(const v2u*)(p + (size_t)min(y0 + d, H - 1) * W); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll 1
  for (int r0 = 0; r0 < ROWS; r0 += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const v2u q = pf[d];
      v2f x[4] = {(v2f){(float)(q.x & 255u), (float)((q.x >> 8) & 255u)}, (v2f){(float)((q.x >> 16) & 255u), (float)(q.x >> 24)},
                  (v2f){(float)(q.y & 255u), (float)((q.y >> 8) & 255u)}, (v2f){(float)((q.y >> 16) & 255u), (float)(q.y >> 24)}};
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i]));
      __builtin_amdgcn_sched_barrier(0);
      pf[d] = *(const v2u*)(p + (size_t)min(y0 + r0 + d + D, H - 1) * W);
      __builtin_amdgcn_sched_barrier(0);
      for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 4; ++i)
          acc[4 * j + i] = __builtin_elementwise_fma((v2f){0.25f + j, 0.25f + j}, x[i], acc[4 * j + i]);
      if constexpr (MODE == 1 || MODE == 2) {
        v2f t[K];
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2)
          asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t[k2]) : "v"(la), "n"(512 * (k2 % 8)) : "memory");
        if (MODE == 1 || d == D - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2) asm volatile("" : "+v"(t[k2]));
        if (MODE == 1) for (int k2 = 0; k2 < K; ++k2) sink += t[k2];
      } else if constexpr (MODE == 3) {
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2)
          asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(la), "v"(acc[k2 % 12]), "n"(512 * (k2 % 8)) : "memory");
      } else if constexpr (MODE == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int k2 = 0; k2 < K / 2; ++k2) {
          v4f w = {acc[k2].x, acc[k2].y, acc[k2 + 1].x, acc[k2 + 1].y};
          asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(la + 8u * lane), "v"(w), "n"(2048 * (k2 % 2)) : "memory");
        }
      } else if constexpr (MODE == 5) {
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2)
          sink = __builtin_elementwise_fma(acc[k2 % 12], (v2f){1.0001f, 0.9999f}, sink);
      }
    }
  }
  float s = sink.x + sink.y;
  for (int i = 0; i < 12; ++i) s += acc[i].x + acc[i].y;
  if (s == 1.2345e-30f) out[0] = s;
}

template <int MODE, int K> void run(const char* name, const unsigned char* src, float* out, int frames) {
  const int per = (W / 512) * ((H + 4 * ROWS - 1) / (4 * ROWS));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE, K>), dim3(per * frames), dim3(256), 0, 0, src, out, frames);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
  }
  printf("%-44s K = %2d per row: %7.3f us per frame  (%5.2f TB/s of source)\n", name, K, best * 1e3 / frames, (double)W * H * frames / (best * 1e-3) / 1e12);
}

int main() {
  const int frames = 96;
  unsigned char* src; float* out;
  CK(hipMalloc(&src, (size_t)W * H * frames)); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 7, (size_t)W * H * frames));
  run<0, 0>("stream + conversions + 12 packed FMAs", src, out, frames);
  run<5, 6>("+ VALU filler", src, out, frames);
  run<5, 12>("+ VALU filler", src, out, frames);
  run<1, 6>("+ ds_read_b64, waited for every row", src, out, frames);
  run<1, 12>("+ ds_read_b64, waited for every row", src, out, frames);
  run<2, 6>("+ ds_read_b64, waited for every 4th row", src, out, frames);
  run<2, 12>("+ ds_read_b64, waited for every 4th row", src, out, frames);
  run<3, 2>("+ ds_write_b64", src, out, frames);
  run<3, 6>("+ ds_write_b64", src, out, frames);
  run<4, 2>("+ ds_write_b128 (K / 2)", src, out, frames);
  run<4, 4>("+ ds_write_b128 (K / 2)", src, out, frames);
  return 0;
}
