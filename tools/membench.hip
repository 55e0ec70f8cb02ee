// Dev tool (not part of the library): HBM streaming ceilings on MI355X for the access
// mixes the surface kernels use.  hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld(const uint4* p) {
  if constexpr (NT) { v4u v = __builtin_nontemporal_load((const v4u*)p); return make_uint4(v.x, v.y, v.z, v.w); }
  else return *p;
}
template <bool NT> __device__ __forceinline__ void st(uint4* p, uint4 v) {
  if constexpr (NT) { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (v4u*)p); }
  else *p = v;
}

// read R streams... simple kernels; n = number of uint4 per stream
template <bool NTL, bool NTS>
__global__ void k_copy(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st<NTS>(out + i, ld<NTL>(in + i));
}
template <bool NTL, bool NTS>
__global__ void k_expand2(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = ld<NTL>(in + i);
    st<NTS>(out + i, v);
    v = make_uint4(~v.x, ~v.y, ~v.z, ~v.w);
    st<NTS>(out + n + i, v);
  }
}
template <bool NTL>
__global__ void k_read(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 v = ld<NTL>(in + i);
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345 && acc.y == 77) out[0] = acc;
}
template <bool NTS>
__global__ void k_write(uint4* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    st<NTS>(out + i, make_uint4((unsigned)i, 1, 2, 3));
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t n = (size_t)4 << 30 >> 4;  // 4 GiB of uint4 per stream
  uint4 *in, *out;
  CK(hipMalloc(&in, n * 16)); CK(hipMalloc(&out, 2 * n * 16));
  CK(hipMemset(in, 0x5a, n * 16)); CK(hipMemset(out, 0, 2 * n * 16));
  const double GB = 1e9;
  int grids[] = {0, 2048, 4096, 8192, 16384};
  for (int g : grids) {
    unsigned blocks = g ? g : (unsigned)((n + 255) / 256);
    printf("grid=%u\n", blocks);
#define RUN(name, bytes, ...) { float ms = timeit([&] { __VA_ARGS__; }); printf("  %-28s %8.3f ms  %8.1f GB/s\n", name, ms, (bytes) / (ms * 1e-3) / GB); }
    RUN("read", n * 16.0, (k_read<false><<<blocks, 256>>>(in, out, n)));
    RUN("read nt", n * 16.0, (k_read<true><<<blocks, 256>>>(in, out, n)));
    RUN("write", n * 16.0, (k_write<false><<<blocks, 256>>>(out, n)));
    RUN("write nt", n * 16.0, (k_write<true><<<blocks, 256>>>(out, n)));
    RUN("copy 1:1", n * 32.0, (k_copy<false, false><<<blocks, 256>>>(in, out, n)));
    RUN("copy 1:1 nt-st", n * 32.0, (k_copy<false, true><<<blocks, 256>>>(in, out, n)));
    RUN("copy 1:1 nt-ld nt-st", n * 32.0, (k_copy<true, true><<<blocks, 256>>>(in, out, n)));
    RUN("expand 1:2", n * 48.0, (k_expand2<false, false><<<blocks, 256>>>(in, out, n)));
    RUN("expand 1:2 nt-st", n * 48.0, (k_expand2<false, true><<<blocks, 256>>>(in, out, n)));
    RUN("expand 1:2 nt-ld nt-st", n * 48.0, (k_expand2<true, true><<<blocks, 256>>>(in, out, n)));
  }
  return 0;
}
