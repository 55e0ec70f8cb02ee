// Dev tool (not part of the library): design-space sweep for the NV12->RGB packed kernel.
// Every variant converts F frames of WxH NV12 (separate allocations) into packed RGB and
// is checked against variant 0 by a checksum; prints GB/s of algorithmic traffic.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/nv12_variants.hip -o tools/nv12_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef uint32_t u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Frame { const uint8_t* y; const uint8_t* uv; uint8_t* rgb; };
struct Csc { float y0, cy, crv, cgu, cgv, cbu; };

#define GLB __attribute__((address_space(1)))
template <bool NT, bool G = false> __device__ __forceinline__ uint4 ld(const uint8_t* p) {
  if constexpr (G) { v4u v = NT ? __builtin_nontemporal_load((const GLB v4u*)p) : *(const GLB v4u*)p; return make_uint4(v.x, v.y, v.z, v.w); }
  else if constexpr (NT) { v4u v = __builtin_nontemporal_load((const v4u*)p); return make_uint4(v.x, v.y, v.z, v.w); }
  else return *(const uint4*)p;
}
template <bool NT, bool G = false> __device__ __forceinline__ void st(uint8_t* p, uint4 v) {
  v4u w = {v.x, v.y, v.z, v.w};
  if constexpr (G) { if constexpr (NT) __builtin_nontemporal_store(w, (GLB v4u*)p); else *(GLB v4u*)p = w; }
  else if constexpr (NT) __builtin_nontemporal_store(w, (v4u*)p);
  else *(uint4*)p = v;
}
template <int I> __device__ __forceinline__ float ub(u32 w) { return (float)((w >> (8 * I)) & 0xffu); }
template <int S> __device__ __forceinline__ u32 pk(float v, u32 o) { return __builtin_amdgcn_cvt_pk_u8_f32(v, S, o); }
struct CT { float rv, guv, bu; };
__device__ __forceinline__ CT chroma(float u, float v, const Csc& k) {
  const float uc = u - 128.f, vc = v - 128.f; CT t; t.rv = k.crv * vc; t.guv = __builtin_fmaf(k.cgu, uc, k.cgv * vc); t.bu = k.cbu * uc; return t; }
__device__ __forceinline__ void emit4(u32 y4, const CT& a, const CT& b, const Csc& k, u32* o) {
  const float y0 = k.cy * (ub<0>(y4) - k.y0), y1 = k.cy * (ub<1>(y4) - k.y0), y2 = k.cy * (ub<2>(y4) - k.y0), y3 = k.cy * (ub<3>(y4) - k.y0);
  u32 d0 = 0, d1 = 0, d2 = 0;
  d0 = pk<0>(y0 + a.rv, d0); d0 = pk<1>(y0 + a.guv, d0); d0 = pk<2>(y0 + a.bu, d0); d0 = pk<3>(y1 + a.rv, d0);
  d1 = pk<0>(y1 + a.guv, d1); d1 = pk<1>(y1 + a.bu, d1); d1 = pk<2>(y2 + b.rv, d1); d1 = pk<3>(y2 + b.guv, d1);
  d2 = pk<0>(y2 + b.bu, d2); d2 = pk<1>(y3 + b.rv, d2); d2 = pk<2>(y3 + b.guv, d2); d2 = pk<3>(y3 + b.bu, d2);
  o[0] = d0; o[1] = d1; o[2] = d2;
}
__device__ __forceinline__ void convert16x2(uint4 ya, uint4 yb, uint4 uv, const Csc& k, u32* o0, u32* o1) {
  const u32 a[4] = {ya.x, ya.y, ya.z, ya.w}, b[4] = {yb.x, yb.y, yb.z, yb.w}, c[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const CT c01 = chroma(ub<0>(c[j]), ub<1>(c[j]), k), c23 = chroma(ub<2>(c[j]), ub<3>(c[j]), k);
    emit4(a[j], c01, c23, k, o0 + 3 * j); emit4(b[j], c01, c23, k, o1 + 3 * j);
  }
}
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

template <bool NTS, bool G = false>
__device__ __forceinline__ void strip_store(uint4* L, int lane, const u32* o, bool valid, uint8_t* rb, int valid_bytes) {
  if (valid) { L[lane * 3] = make_uint4(o[0], o[1], o[2], o[3]); L[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]); L[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]); }
  wsync();
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int off = (k * 64 + lane) * 16; if (off < valid_bytes) st<NTS, G>(rb + off, L[k * 64 + lane]); }
  wsync();
}

// FLAGS: 1 = nt stores, 2 = nt loads, 4 = waves stacked in y (block = 4 row pairs x 1024 px),
//        8 = XCD-contiguous row-pair remap, 16 = skeleton (no math: copy bytes), 32 = direct stores
template <int FLAGS, int RPT /*row pairs per thread*/>
__global__ void __launch_bounds__(256) k_conv(const Frame* fr, int W, int H, int sp, int dp, Csc k, int rp_per_xcd) {
  constexpr bool NTS = FLAGS & 1, NTL = FLAGS & 2, STACK = FLAGS & 4, XCD = FLAGS & 8, SKEL = FLAGS & 16, DIRECT = FLAGS & 32, GL = FLAGS & 64;
  __shared__ uint4 lds[4][192];
  extern __shared__ uint4 dyn_pad[];
  const Frame f = fr[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = W / 16, row_pairs = H / 2;
  int by = blockIdx.y;
  if constexpr (XCD) { const int idx = by >> 3, c = idx / rp_per_xcd, w = idx - c * rp_per_xcd; by = (c * 8 + (by & 7)) * rp_per_xcd + w; }
  int wave_g0, rp0;
  if constexpr (STACK) { wave_g0 = blockIdx.x * 64; rp0 = (by * 4 + wave) * RPT; }
  else { wave_g0 = blockIdx.x * 256 + wave * 64; rp0 = by * RPT; }
  if (wave_g0 >= groups || rp0 >= row_pairs) return;
  const int g = wave_g0 + lane, x0 = g * 16;
  const bool valid = g < groups;
  const int valid_bytes = min(64, groups - wave_g0) * 48;
  uint4 ya[RPT], yb[RPT], uv[RPT];
  if (valid) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int rp = min(rp0 + r, row_pairs - 1);
      ya[r] = ld<NTL, GL>(f.y + (size_t)(2 * rp) * sp + x0);
      yb[r] = ld<NTL, GL>(f.y + (size_t)(2 * rp + 1) * sp + x0);
      uv[r] = ld<NTL, GL>(f.uv + (size_t)rp * sp + x0);
    }
  }
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const int rp = rp0 + r;
    if (rp >= row_pairs) break;
    u32 o0[12], o1[12];
    if (valid) {
      if constexpr (SKEL) {
        const u32 a[4] = {ya[r].x, ya[r].y, ya[r].z, ya[r].w}, b[4] = {yb[r].x, yb[r].y, yb[r].z, yb[r].w}, c[4] = {uv[r].x, uv[r].y, uv[r].z, uv[r].w};
#pragma unroll
        for (int j = 0; j < 12; ++j) { o0[j] = a[j & 3] ^ c[(j >> 2) & 3]; o1[j] = b[j & 3] + c[(j >> 2) & 3]; }
      } else convert16x2(ya[r], yb[r], uv[r], k, o0, o1);
    }
    uint8_t* rb = f.rgb + (size_t)(2 * rp) * dp + (size_t)wave_g0 * 48;
    if constexpr (DIRECT) {
      if (valid) {
        uint8_t* p = f.rgb + (size_t)(2 * rp) * dp + (size_t)g * 48;
        st<NTS>(p, make_uint4(o0[0], o0[1], o0[2], o0[3])); st<NTS>(p + 16, make_uint4(o0[4], o0[5], o0[6], o0[7])); st<NTS>(p + 32, make_uint4(o0[8], o0[9], o0[10], o0[11]));
        p += dp;
        st<NTS>(p, make_uint4(o1[0], o1[1], o1[2], o1[3])); st<NTS>(p + 16, make_uint4(o1[4], o1[5], o1[6], o1[7])); st<NTS>(p + 32, make_uint4(o1[8], o1[9], o1[10], o1[11]));
      }
    } else {
      strip_store<NTS, GL>(lds[wave], lane, o0, valid, rb, valid_bytes);
      strip_store<NTS, GL>(lds[wave], lane, o1, valid, rb + dp, valid_bytes);
    }
  }
}

// one ROW per wave-row: lane = 16 px x 1 row, UV re-read by the second row (L2 hit)
template <int FLAGS>
__global__ void __launch_bounds__(256) k_conv_row(const Frame* fr, int W, int H, int sp, int dp, Csc k, int C) {
  constexpr bool NTS = FLAGS & 1; constexpr bool XCD = FLAGS & 8;
  __shared__ uint4 lds[4][192];
  extern __shared__ uint4 dyn_pad2[];
  const Frame f = fr[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = W / 16;
  int row = blockIdx.y;
  if constexpr (XCD) { const int idx = row >> 3, c = idx / C, w = idx - c * C; row = (c * 8 + (row & 7)) * C + w; }
  const int wave_g0 = blockIdx.x * 256 + wave * 64;
  if (wave_g0 >= groups || row >= H) return;
  const int g = wave_g0 + lane, x0 = g * 16;
  const bool valid = g < groups;
  u32 o0[12], o1[12];
  if (valid) {
    const uint4 ya = *(const uint4*)(f.y + (size_t)row * sp + x0);
    const uint4 uv = *(const uint4*)(f.uv + (size_t)(row >> 1) * sp + x0);
    convert16x2(ya, ya, uv, k, o0, o1);
  }
  strip_store<NTS>(lds[wave], lane, o0, valid, f.rgb + (size_t)row * dp + (size_t)wave_g0 * 48, min(64, groups - wave_g0) * 48);
}

__global__ void k_sum(const uint8_t* p, size_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * blockDim.x) {
    const u32 w = ((const u32*)p)[i]; s += (w & 0xff) * 1ull + ((w >> 8) & 0xff) * 3ull + ((w >> 16) & 0xff) * 7ull + (w >> 24) * 11ull + (i & 1023) * (w & 0xff);
  }
  atomicAdd(out, s);
}

int main(int argc, char** argv) {
  const int W = 3840, H = 2160, F = argc > 1 ? atoi(argv[1]) : 256;
  const int sp = W, dp = W * 3;
  std::vector<Frame> fr(F);
  std::vector<uint8_t> host((size_t)sp * H * 3 / 2);
  srand(1);
  for (auto& b : host) b = (uint8_t)(16 + rand() % 220);
  for (int i = 0; i < F; ++i) {
    uint8_t *y, *rgb;
    CK(hipMalloc(&y, (size_t)sp * H * 3 / 2)); CK(hipMalloc(&rgb, (size_t)dp * H));
    CK(hipMemcpy(y, host.data(), host.size(), hipMemcpyHostToDevice));
    fr[i] = {y, y + (size_t)sp * H, rgb};
  }
  Frame* dfr; CK(hipMalloc(&dfr, F * sizeof(Frame))); CK(hipMemcpy(dfr, fr.data(), F * sizeof(Frame), hipMemcpyHostToDevice));
  unsigned long long* dsum; CK(hipMalloc(&dsum, 8));
  const Csc k = {16.f, 1.164f, 1.793f, -0.213f, -0.533f, 2.112f};
  const double bytes = (double)F * (W * H * 1.5 + W * H * 3.0);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  unsigned long long ref = 0;
  auto run = [&](const char* name, auto launch, bool check) {
    for (int i = 0; i < F; ++i) CK(hipMemsetAsync(fr[i].rgb, 0, 64, 0));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0;
    const int reps = 8;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
    CK(hipMemset(dsum, 0, 8));
    k_sum<<<1024, 256>>>(fr[F - 1].rgb, (size_t)dp * H, dsum);
    unsigned long long s; CK(hipMemcpy(&s, dsum, 8, hipMemcpyDeviceToHost));
    if (!ref) ref = s;
    printf("%-44s best %7.3f ms %7.1f GB/s | avg %7.3f ms %7.1f GB/s %s\n", name, best, bytes / best / 1e6, tot / reps, bytes / (tot / reps) / 1e6,
           check ? (s == ref ? "ok" : "MISMATCH") : "");
  };
  const int rows = H / 2;
#define GRID_X dim3((W / 16 + 255) / 256, rows, F)
#define L(FL, RPT, grid, rpx) [&] { k_conv<FL, RPT><<<grid, 256>>>(dfr, W, H, sp, dp, k, rpx); }
  auto pad8 = [](int n, int C) { int per = 8 * C; return (n + per - 1) / per * per; };
#define LD(FL, RPT, grid, rpx, dyn) [&] { k_conv<FL, RPT><<<grid, 256, dyn>>>(dfr, W, H, sp, dp, k, rpx); }
  auto dyn_for = [](int blocks) { return (160 * 1024 / blocks) / 1024 * 1024 - 12 * 1024 - 512; };
  for (int rep = 0; rep < 3; ++rep) {
    run("flat   : xcd135 nts <=4 blocks/CU", LD(9, 1, dim3(1, rows, F), rows / 8, dyn_for(4)), true);
    run("global : xcd135 nts <=4 blocks/CU", LD(73, 1, dim3(1, rows, F), rows / 8, dyn_for(4)), true);
  }
  run("global : xcd135 nts <=5 blocks/CU", LD(73, 1, dim3(1, rows, F), rows / 8, dyn_for(5)), true);
  run("global : xcd135 nts <=3 blocks/CU", LD(73, 1, dim3(1, rows, F), rows / 8, dyn_for(3)), true);
  run("global : xcd135 nt ld+st <=4", LD(75, 1, dim3(1, rows, F), rows / 8, dyn_for(4)), true);
  return 0;
}
