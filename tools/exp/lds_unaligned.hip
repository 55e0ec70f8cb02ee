// Misaligned LDS reads on gfx950: are they correct, and what do they cost?  (round 6: the LDS-staged rotation reads the two
// horizontal taps of a packed-RGB row -- 6 bytes at ANY byte address -- with one ds_read_b64.)
// build: hipcc --offload-arch=gfx950 -O2 -o tools/exp/lds_unaligned tools/exp/lds_unaligned.hip
// Reports ticks (s_memtime) per 1000 wave instructions with 4 waves of one workgroup issuing concurrently (as lds_patterns.hip)
// and checks every loaded value against the bytes written.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 8
// PAT: 0 ds_read_b64, 1 ds_read_b32, 2 ds_read_u16, 3 ds_read_b96, 4 ds_read_b128, 5 ds_read2_b32 offset1:1, 6 ds_read_u8
template <int PAT> __device__ __forceinline__ unsigned long long access(unsigned a) {
  unsigned long long r = 0;
  if constexpr (PAT == 0) { unsigned long long v; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v; }
  if constexpr (PAT == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v; }
  if constexpr (PAT == 2) { unsigned v; asm volatile("ds_read_u16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v; }
  if constexpr (PAT == 3) { typedef unsigned v3 __attribute__((ext_vector_type(3))); v3 v; asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v.x ^ ((unsigned long long)v.y << 32) ^ v.z; }
  if constexpr (PAT == 4) { typedef unsigned v4 __attribute__((ext_vector_type(4))); v4 v; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v.x ^ ((unsigned long long)v.y << 32) ^ v.z ^ ((unsigned long long)v.w << 32); }
  if constexpr (PAT == 5) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v; }
  if constexpr (PAT == 6) { unsigned v; asm volatile("ds_read_u8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory"); r = v; }
  return r;
}
template <int PAT> __device__ __forceinline__ void issue(unsigned a) {
  if constexpr (PAT == 0) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 1) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 2) { unsigned v; asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 3) { typedef unsigned v3 __attribute__((ext_vector_type(3))); v3 v; asm volatile("ds_read_b96 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 4) { typedef unsigned v4 __attribute__((ext_vector_type(4))); v4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 5) { unsigned long long v; asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(a) : "memory"); }
  if constexpr (PAT == 6) { unsigned v; asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(a) : "memory"); }
}
__host__ __device__ inline unsigned char byte_at(unsigned off) { return (unsigned char)((off * 2654435761u) >> 13); }
__host__ __device__ inline unsigned long long expect(int pat, unsigned off) {
  auto le = [&](unsigned o, int n) { unsigned long long v = 0; for (int k = 0; k < n; ++k) v |= (unsigned long long)byte_at(o + k) << (8 * k); return v; };
  switch (pat) {
  case 0: return le(off, 8);
  case 1: return le(off, 4);
  case 2: return le(off, 2);
  case 3: return le(off, 4) ^ (le(off + 4, 4) << 32) ^ le(off + 8, 4);
  case 4: return le(off, 4) ^ (le(off + 4, 4) << 32) ^ le(off + 8, 4) ^ (le(off + 12, 4) << 32);
  case 5: return le(off, 8);
  default: return le(off, 1);
  }
}

// lane address: base + wave * 16384 + f(lane): MODE 0: lane * stride + misalign ; MODE 1: the rotation's pattern -- 8 lanes x 8 rows of a
// 32 x 8 dst block whose pixels step (c, s) in the source: byte address = ((j * rowpx) + i) * px + misalign with i = round(4 l8 c - r s), j = round(4 l8 s + r c)
template <int PAT> __global__ void __launch_bounds__(256) k(int mode, int stride, int misalign, int px, int rowpx, float c, float s, int iters,
                                                             long long* out, unsigned* bad) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  for (int i = threadIdx.x; i < 65536; i += 256) lds[i] = byte_at(i);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned off;
  if (mode == 0) off = wave * 16384 + (lane * stride) % 8192 + misalign;
  else {
    const int l8 = lane & 7, r = lane >> 3;
    const int i = (int)__builtin_rintf(4.f * l8 * c - r * s) + 12, j = (int)__builtin_rintf(4.f * l8 * s + r * c) + 20;
    off = wave * 16384 + (j * rowpx + i) * px + misalign;
  }
  unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + off;
  __syncthreads();
  if (access<PAT>(a) != expect(PAT, off)) atomicAdd(bad, 1u);
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r)
      issue<PAT>(a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  if (lane == 0)
    out[wave] = t1 - t0;
}

struct Case { int mode, stride, misalign, px, rowpx; float c, s; };
template <int PAT> void run(const char* name, std::vector<Case> cases) {
  long long* d; unsigned* bad; hipMalloc(&d, 64); hipMalloc(&bad, 4);
  for (const Case& q : cases) {
    const int iters = 2000;
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k<PAT>, dim3(1), dim3(256), 0, 0, q.mode, q.stride, q.misalign, q.px, q.rowpx, q.c, q.s, iters, d, bad);
    hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k<PAT>, dim3(1), dim3(256), 0, 0, q.mode, q.stride, q.misalign, q.px, q.rowpx, q.c, q.s, iters, d, bad);
    hipDeviceSynchronize();
    long long h[4]; unsigned hb = 0; hipMemcpy(h, d, 32, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    double per = 0; for (int w = 0; w < 4; ++w) per += (double)h[w]; per /= 4.0;
    if (q.mode == 0)
      printf("%-16s lane stride %3d B + %d: %8.3f ticks / 1000 wave-instr   wrong lanes %u\n", name, q.stride, q.misalign, per * 1000.0 / (iters * REP), hb);
    else
      printf("%-16s rotation c=%.3f s=%.3f px %d B row %d px + %d: %8.3f ticks / 1000 wave-instr   wrong lanes %u\n", name, q.c, q.s, q.px, q.rowpx,
             q.misalign, per * 1000.0 / (iters * REP), hb);
  }
  hipFree(d); hipFree(bad);
}

int main() {
  const float c30 = 0.8660254f, s30 = 0.5f, c45 = 0.70710678f, c10 = 0.98480775f, s10 = 0.17364818f;
  run<0>("ds_read_b64", {{0, 8, 0}, {0, 8, 4}, {0, 8, 1}, {0, 8, 3}, {0, 12, 0}, {0, 12, 1}, {0, 3, 0}, {0, 6, 0}, {0, 16, 4}, {0, 16, 5},
                         {1, 0, 0, 4, 49, c30, s30}, {1, 0, 0, 4, 49, c45, c45}, {1, 0, 0, 4, 49, c10, s10}, {1, 0, 0, 4, 48, c30, s30},
                         {1, 0, 0, 3, 64, c30, s30}, {1, 0, 0, 3, 64, c45, c45}, {1, 0, 0, 3, 64, c10, s10}, {1, 0, 1, 3, 64, c30, s30},
                         {1, 0, 0, 8, 49, c30, s30}, {1, 0, 0, 8, 49, c45, c45}});
  run<5>("ds_read2_b32+1", {{0, 8, 0}, {0, 8, 4}, {0, 12, 0}, {1, 0, 0, 4, 49, c30, s30}, {1, 0, 0, 4, 49, c45, c45}, {1, 0, 0, 4, 49, c10, s10}});
  run<1>("ds_read_b32", {{0, 4, 0}, {0, 4, 1}, {0, 4, 2}, {0, 3, 0}, {1, 0, 0, 4, 49, c30, s30}, {1, 0, 0, 3, 64, c30, s30}, {1, 0, 0, 1, 64, c30, s30},
                         {1, 0, 0, 2, 64, c30, s30}});
  run<2>("ds_read_u16", {{0, 2, 0}, {0, 2, 1}, {0, 1, 0}, {1, 0, 0, 1, 64, c30, s30}, {1, 0, 0, 1, 64, c45, c45}, {1, 0, 0, 2, 64, c30, s30}});
  run<6>("ds_read_u8", {{0, 1, 0}, {1, 0, 0, 1, 64, c30, s30}});
  run<3>("ds_read_b96", {{0, 16, 0}, {0, 12, 0}, {0, 12, 4}, {0, 12, 1}, {1, 0, 0, 12, 49, c30, s30}, {1, 0, 0, 12, 49, c45, c45}});
  run<4>("ds_read_b128", {{0, 16, 0}, {0, 16, 4}, {0, 16, 8}, {0, 16, 1}, {0, 12, 0}, {1, 0, 0, 8, 49, c30, s30}, {1, 0, 0, 8, 49, c45, c45},
                          {1, 0, 0, 12, 49, c30, s30}});
  return 0;
}
