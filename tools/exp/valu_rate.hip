// VALU issue cost on gfx950: cycles per wave instruction and SIMD, for the instructions the resize kernels are made of,
// at 1 / 2 / 4 / 8 waves per SIMD.  build: hipcc --offload-arch=gfx950 -O3 tools/exp/valu_rate.hip -o tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kIter = 2000, kUnroll = 16;
template <int OP> __global__ void __launch_bounds__(1024) k(float* out, long long* cyc) {
  v2f a[kUnroll];
  unsigned u = threadIdx.x * 2654435761u;
  for (int i = 0; i < kUnroll; ++i) a[i] = (v2f){(float)i + threadIdx.x, 1.0f};
  v2f w = {1.0001f, 0.9999f};
  float s = 0.5f;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < kIter; ++it) {
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
      if constexpr (OP == 0) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(w));
      if constexpr (OP == 1) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(a[i].x) : "v"(s));
      if constexpr (OP == 2) asm volatile("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(a[i].x) : "v"(u));
      if constexpr (OP == 3) asm volatile("v_mov_b64 %0, %1" : "=v"(a[i]) : "v"(w));
      if constexpr (OP == 4) asm volatile("v_pk_mov_b32 %0, %1, %0 op_sel:[1,0]" : "+v"(a[i]) : "v"(w));
      if constexpr (OP == 5) asm volatile("v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(w));
      if constexpr (OP == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i].x) : "v"(s));
      if constexpr (OP == 7) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(u) : "v"(a[i].x));
      if constexpr (OP == 8) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(w));
      if constexpr (OP == 9) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i].x) : "v"(s));
      if constexpr (OP == 10) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(w));
    }
  }
  long long t1 = clock64();
  float r = 0;
  for (int i = 0; i < kUnroll; ++i) r += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r + (float)u;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, float* out, long long* cyc) {
  printf("%-28s", name);
  for (int waves_per_simd : {1, 2, 4, 8}) {  // one block per CU-ish: 256 blocks of (waves_per_simd * 4) waves -- but max block 1024 = 16 waves = 4 / SIMD
    const int threads = waves_per_simd <= 4 ? waves_per_simd * 256 : 1024, blocks = waves_per_simd <= 4 ? 256 : 512;
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc);
    CK(hipDeviceSynchronize());
    long long h[512];
    CK(hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost));
    double avg = 0;
    for (int i = 0; i < blocks; ++i) avg += (double)h[i];
    avg /= blocks;
    // clock64 = s_memtime: 100 MHz constant clock?  report per-instruction time in ns via events instead
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<OP>), dim3(blocks), dim3(threads), 0, 0, out, cyc);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double insts_per_simd = (double)kIter * kUnroll * waves_per_simd;   // wave instructions each SIMD issues
    printf("  w%d: %6.2f ns/inst/SIMD (%5.2f cyc @2.4GHz)", waves_per_simd, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
  }
  printf("\n");
}
int main() {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 4 * 1024 * 512)); CK(hipMalloc(&cyc, 8 * 512));
  run<0>("v_pk_fma_f32", out, cyc);
  run<5>("v_pk_fma_f32 op_sel", out, cyc);
  run<1>("v_fma_f32", out, cyc);
  run<2>("v_cvt_f32_ubyte1", out, cyc);
  run<3>("v_mov_b64", out, cyc);
  run<6>("v_mov_b32", out, cyc);
  run<4>("v_pk_mov_b32", out, cyc);
  run<7>("v_cvt_pk_u8_f32", out, cyc);
  run<8>("v_pk_add_f32", out, cyc);
  run<9>("v_add_f32", out, cyc);
  run<10>("v_pk_mul_f32", out, cyc);
  return 0;
}
