// Dev tool: second design-space sweep for NV12->RGB (block->work mappings).  Same arithmetic
// and LDS-staged nt stores as the library kernel; only the mapping of workgroups to
// (frame, row pair, x segment) and the workgroup size change.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/nv12_variants2.hip -o tools/nv12_variants2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef uint32_t u32;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Frame { const uint8_t* y; const uint8_t* uv; uint8_t* rgb; };
struct Csc { float y0, cy, crv, cgu, cgv, cbu; };

__device__ __forceinline__ void st_nt(uint8_t* p, uint4 v) { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (v4u*)p); }
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
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
__device__ __forceinline__ void strip_store(uint4* L, int lane, const u32* o, bool valid, uint8_t* rb, int valid_bytes) {
  if (valid) { L[lane * 3] = make_uint4(o[0], o[1], o[2], o[3]); L[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]); L[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]); }
  wsync();
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int off = (k * 64 + lane) * 16; if (off < valid_bytes) st_nt(rb + off, L[k * 64 + lane]); }
  wsync();
}

// MODE 0: library mapping (XCD gets a contiguous eighth of every frame; grid.y = frame)
// MODE 1: frame-per-XCD (XCD k walks whole frames k, k+8, ...; 1-D grid)
// MODE 2: as 0 but XCD-contiguous over the WHOLE batch (XCD k owns frames [k*F/8, (k+1)*F/8))
// BLOCK = threads per workgroup (64..512); a workgroup covers BLOCK/64 waves of one row pair
// (BLOCK <= 256) or two row pairs (BLOCK = 512).
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_conv(const Frame* fr, int F, int W, int H, int sp, int dp, Csc k) {
  extern __shared__ uint4 lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int groups = W / 16, rows = H / 2;
  constexpr int WPB = BLOCK / 64;
  const int segs = (groups + 64 * (WPB > 4 ? 4 : WPB) - 1) / (64 * (WPB > 4 ? 4 : WPB)); // x segments per row pair
  const int rp_per_block = WPB > 4 ? 2 : 1;
  const int tiles_per_frame = segs * ((rows + rp_per_block - 1) / rp_per_block);
  u32 b = blockIdx.x;
  int frame, tile;
  if constexpr (MODE == 0) {
    frame = blockIdx.y;
    const int per = (tiles_per_frame + 7) / 8;
    tile = (b & 7) * per + (b >> 3);
    if (tile >= tiles_per_frame) return;
  } else if constexpr (MODE == 1) {
    const u32 xcd = b & 7, idx = b >> 3;
    frame = (idx / tiles_per_frame) * 8 + xcd;
    tile = idx % tiles_per_frame;
    if (frame >= F) return;
  } else {
    const u32 xcd = b & 7, idx = b >> 3;
    const int fpx = (F + 7) / 8;
    frame = xcd * fpx + idx / tiles_per_frame;
    tile = idx % tiles_per_frame;
    if (frame >= F || (int)(idx / tiles_per_frame) >= fpx) return;
  }
  const Frame f = fr[frame];
  const int seg = tile % segs;
  int rp = (tile / segs) * rp_per_block;
  int w_in_row = wave;
  if (WPB > 4) { rp += wave >> 2; w_in_row = wave & 3; }
  const int wave_g0 = (seg * (WPB > 4 ? 4 : WPB) + w_in_row) * 64;
  if (wave_g0 >= groups || rp >= rows) return;
  const int g = wave_g0 + lane, x0 = g * 16;
  const bool valid = g < groups;
  const int valid_bytes = min(64, groups - wave_g0) * 48;
  u32 o0[12], o1[12];
  if (valid) {
    const uint4 ya = *(const uint4*)(f.y + (size_t)(2 * rp) * sp + x0);
    const uint4 yb = *(const uint4*)(f.y + (size_t)(2 * rp + 1) * sp + x0);
    const uint4 uv = *(const uint4*)(f.uv + (size_t)rp * sp + x0);
    const u32 a[4] = {ya.x, ya.y, ya.z, ya.w}, bb[4] = {yb.x, yb.y, yb.z, yb.w}, c[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const CT c01 = chroma(ub<0>(c[j]), ub<1>(c[j]), k), c23 = chroma(ub<2>(c[j]), ub<3>(c[j]), k);
      emit4(a[j], c01, c23, k, o0 + 3 * j); emit4(bb[j], c01, c23, k, o1 + 3 * j);
    }
  }
  uint4* L = lds + wave * 192;
  uint8_t* rb = f.rgb + (size_t)(2 * rp) * dp + (size_t)wave_g0 * 48;
  strip_store(L, lane, o0, valid, rb, valid_bytes);
  strip_store(L, lane, o1, valid, rb + dp, valid_bytes);
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
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < F; ++i) CK(hipMemsetAsync(fr[i].rgb, 0, 64, 0));
    launch(); CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0; const int reps = 8;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; tot += ms; }
    CK(hipMemset(dsum, 0, 8));
    k_sum<<<1024, 256>>>(fr[F - 1].rgb, (size_t)dp * H, dsum);
    unsigned long long s; CK(hipMemcpy(&s, dsum, 8, hipMemcpyDeviceToHost));
    if (!ref) ref = s;
    printf("%-52s best %7.3f ms %7.1f GB/s | avg %7.3f ms %7.1f GB/s %s\n", name, best, bytes / best / 1e6, tot / reps, bytes / (tot / reps) / 1e6, s == ref ? "ok" : "MISMATCH");
  };
  const int rows = H / 2, groups = W / 16;
  auto lds_for = [](int block, int waves_per_cu) { const int bpc = waves_per_cu / (block / 64); unsigned b = ((160u * 1024u / bpc) & ~1023u) - 512u; return b > 65536u ? 65536u : b; };
  auto tiles = [&](int block) { const int wpb = block / 64; const int per_row = wpb > 4 ? 4 : wpb; const int segs = (groups + 64 * per_row - 1) / (64 * per_row); return segs * ((rows + (wpb > 4 ? 2 : 1) - 1) / (wpb > 4 ? 2 : 1)); };
#define GRID0(B) dim3(((tiles(B) + 7) / 8) * 8, F)
#define GRID1(B) dim3((unsigned)(((F + 7) / 8) * tiles(B) * 8))
  for (int rep = 0; rep < 2; ++rep) {
    run("mode0 (library) block256 16w/CU", [&] { k_conv<0, 256><<<GRID0(256), 256, lds_for(256, 16)>>>(dfr, F, W, H, sp, dp, k); });
    run("mode1 frame-per-XCD block256 16w/CU", [&] { k_conv<1, 256><<<GRID1(256), 256, lds_for(256, 16)>>>(dfr, F, W, H, sp, dp, k); });
    run("mode2 batch-contiguous-per-XCD block256 16w/CU", [&] { k_conv<2, 256><<<GRID1(256), 256, lds_for(256, 16)>>>(dfr, F, W, H, sp, dp, k); });
  }
  run("mode0 block128 16w/CU", [&] { k_conv<0, 128><<<GRID0(128), 128, lds_for(128, 16)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode0 block64  16w/CU", [&] { k_conv<0, 64><<<GRID0(64), 64, lds_for(64, 16)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode0 block512 16w/CU", [&] { k_conv<0, 512><<<GRID0(512), 512, lds_for(512, 16)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode0 block128 20w/CU", [&] { k_conv<0, 128><<<GRID0(128), 128, lds_for(128, 20)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode0 block128 12w/CU", [&] { k_conv<0, 128><<<GRID0(128), 128, lds_for(128, 12)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode1 block128 16w/CU", [&] { k_conv<1, 128><<<GRID1(128), 128, lds_for(128, 16)>>>(dfr, F, W, H, sp, dp, k); });
  run("mode0 (library) block256 16w/CU again", [&] { k_conv<0, 256><<<GRID0(256), 256, lds_for(256, 16)>>>(dfr, F, W, H, sp, dp, k); });
  return 0;
}
