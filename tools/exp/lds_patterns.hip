// LDS access patterns on gfx950, cycles per wave instruction with 4 waves (one per SIMD) of ONE workgroup hammering the
// CU's LDS: which lane strides / instruction forms conflict.  hipcc --offload-arch=gfx950 -O2 -o lds_patterns lds_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 8
template <int PAT> __device__ __forceinline__ void access(unsigned a) {
  // a = this lane's LDS byte address
  if constexpr (PAT == 0) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 2) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:64" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 3) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:2" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 4) { __uint128_t v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 5) { __uint128_t v = 0; asm volatile("ds_write_b128 %0, %1" : : "v"(a), "v"(v) : "memory"); }
  if constexpr (PAT == 6) { unsigned long long v = 0; asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(v) : "memory"); }
  if constexpr (PAT == 7) { unsigned v = 0; asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory"); }
  if constexpr (PAT == 8) { __uint128_t v; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 9) { unsigned long long v = 0; asm volatile("ds_write2st64_b64 %0, %1, %1 offset1:1" : : "v"(a), "v"(v) : "memory"); }
  if constexpr (PAT == 10) { unsigned long long v; asm volatile("ds_read2st64_b32 %0, %1 offset1:9" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 11) { unsigned v = 0; asm volatile("ds_write2_b32 %0, %1, %1 offset1:2" : : "v"(a), "v"(v) : "memory"); }
}

template <int PAT> __global__ void __launch_bounds__(256) k(int stride, int iters, long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (unsigned)(wave * 16384 + (lane * stride) % 8192);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
      access<PAT>(a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0)
    out[wave] = t1 - t0;
}

template <int PAT> void run(const char* name, std::vector<int> strides) {
  long long* d; hipMalloc(&d, 64);
  for (int s : strides) {
    const int iters = 2000;
    hipLaunchKernelGGL(k<PAT>, dim3(1), dim3(256), 0, 0, s, iters, d);
    hipLaunchKernelGGL(k<PAT>, dim3(1), dim3(256), 0, 0, s, iters, d);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    // clock64 = s_memtime at 100 MHz on this chip: report relative units per CU-wide instruction (4 waves x REP x iters)
    double per = 0; for (int w = 0; w < 4; ++w) per += (double)h[w]; per /= 4.0;
    printf("%-22s lane stride %3d B: %8.3f ticks per 1000 wave-instr (4 waves concurrently)\n", name, s, per * 1000.0 / (iters * REP));
  }
  hipFree(d);
}

int main() {
  run<1>("ds_read_b32", {4, 8, 16, 32});
  run<0>("ds_read_b64", {0, 8, 16, 32, 64});
  run<4>("ds_read_b128", {0, 8, 16, 32, 64}); // 8: every other lane 8 bytes off a 16-byte boundary; 0: broadcast
  run<2>("ds_read2_b32 +64dw", {4, 8, 16});
  run<3>("ds_read2_b32 +2dw", {8, 16, 32});
  run<10>("ds_read2st64_b32 +9", {4, 8, 16});
  run<8>("ds_read2_b64 +1", {16, 32});
  run<7>("ds_write_b32", {4, 8, 32});
  run<6>("ds_write_b64", {8, 16, 32});
  run<5>("ds_write_b128", {16, 32, 64});
  run<9>("ds_write2st64_b64", {8, 16});
  run<11>("ds_write2_b32 +2dw", {8, 16, 32});
  return 0;
}
