// Dev tool (not part of the library): what the memory system gives the ACCESS PATTERN of the quarter-turn kernels, with no
// transposition and no arithmetic.  hipcc --offload-arch=gfx950 -O3 tools/rot_pattern.hip -o tools/rot_pattern
//
// k_rotate_tile<3,*,64> (vali_amd/csrc/rotate.hip) moves packed RGB in 64x64-pixel tiles: a workgroup reads 64 source rows x
// 192 bytes (16 lanes x 12 B per row, 16 rows per wave instruction... 4 passes) and writes 64 destination rows x 192 bytes
// of the tile at the transposed position.  This tool does exactly those loads and stores -- same lanes, same widths, same
// addresses, same workgroup order (XCD-contiguous tile map over the batch) -- and stores what it loaded.  Variants:
//   seg  S     bytes per tile row (192 = the kernel; 384 = its 128-row tiles; 256 / 512 = whole 128-byte lines)
//   rows R     rows per tile
//   mode 0     tile (tx,ty) -> tile (ty,tx)    (the quarter turn's placement)
//        1     tile (tx,ty) -> the same tile   (a tiled straight copy: the pattern without the transposed placement)
//        2     row-major streaming copy of the same bytes, 16 B per lane (the plain ceiling)
// Output: TB/s of (bytes read + bytes written) per variant, frames x (W x H x 3) bytes each way.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

struct Args {
  const unsigned char* src; unsigned char* dst;
  int w_bytes, h, spitch, dpitch_t, dh_t;   // source plane: h rows of w_bytes; transposed plane: dh_t rows, pitch dpitch_t
  int seg, rows, tiles_x, tiles_y, frames, mode;
  size_t sframe, dframe;
  unsigned total, per_xcd;
};

template <int SEG, int ROWS, int VB, int MODE> // bytes per tile row, rows per tile, vector bytes per lane, placement
__global__ void __launch_bounds__(256) k_tiles(Args a) {
  const unsigned b = blockIdx.x, t = (b & 7u) * a.per_xcd + (b >> 3);
  if (t >= a.total) return;
  const unsigned per_frame = a.tiles_x * a.tiles_y, frame = t / per_frame, local = t - frame * per_frame;
  const unsigned ty = local / a.tiles_x, tx = local - ty * a.tiles_x;
  constexpr int LPR = SEG / VB, RPP = 256 / LPR, PASSES = ROWS / RPP;          // loads: lanes per row, rows per pass
  constexpr int OB = MODE == 0 ? ROWS * 3 : SEG, OR = MODE == 0 ? SEG / 3 : ROWS; // the written tile: row bytes, rows
  constexpr int OLPR = OB / VB, ORPP = 256 / OLPR, OPASSES = OR / ORPP;
  static_assert(PASSES == OPASSES && PASSES * RPP == ROWS && OPASSES * ORPP == OR, "same vectors in as out");
  const int chunk = threadIdx.x % LPR, r0 = threadIdx.x / LPR, o_chunk = threadIdx.x % OLPR, o_r0 = threadIdx.x / OLPR;
  const unsigned char* s = a.src + frame * a.sframe + (size_t)(ty * ROWS) * a.spitch + (size_t)tx * SEG;
  unsigned char* d = MODE == 0 ? a.dst + frame * a.dframe + (size_t)(tx * (SEG / 3)) * a.dpitch_t + (size_t)ty * (ROWS * 3)
                               : a.dst + frame * a.sframe + (size_t)(ty * ROWS) * a.spitch + (size_t)tx * SEG;
  const int dp = MODE == 0 ? a.dpitch_t : a.spitch;
  typedef typename std::conditional<VB == 12, v3u, v4u>::type V;
  typedef V VU __attribute__((aligned(4)));
  V q[PASSES];
#pragma unroll
  for (int p = 0; p < PASSES; ++p)
    q[p] = *(const VU __attribute__((address_space(1)))*)(s + (size_t)(p * RPP + r0) * a.spitch + chunk * VB);
#pragma unroll
  for (int p = 0; p < PASSES; ++p)
    *(VU __attribute__((address_space(1)))*)(d + (size_t)(p * ORPP + o_r0) * dp + o_chunk * VB) = q[p];
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = in[i];
}

template <typename F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int W = argc > 1 ? atoi(argv[1]) : 1920, H = argc > 2 ? atoi(argv[2]) : 1080, frames = argc > 3 ? atoi(argv[3]) : 64;
  const int sets = 3; // rotate through 3 surface sets: > 1.5 GiB touched between two visits of a frame
  const int spitch = ((W * 3 + 255) / 256) * 256, dpitch = ((H * 3 + 255) / 256) * 256;
  const size_t sframe = (size_t)spitch * H, dframe = (size_t)dpitch * W;
  const size_t fmax = sframe > dframe ? sframe : dframe;
  unsigned char *src, *dst;
  CK(hipMalloc(&src, fmax * frames * sets)); CK(hipMalloc(&dst, fmax * frames * sets));
  CK(hipMemset(src, 0x5a, fmax * frames * sets)); CK(hipMemset(dst, 0, fmax * frames * sets));
  const double bytes = 2.0 * W * H * 3 * frames;
  printf("RGB %dx%d x %d frames (%.1f MB each way per launch), pitches %d / %d\n", W, H, frames, bytes / 2e6, spitch, dpitch);
  Args a;
  a.w_bytes = W * 3; a.h = H; a.spitch = spitch; a.dpitch_t = dpitch; a.dh_t = W; a.frames = frames;
  a.sframe = sframe; a.dframe = dframe;
  int set = 0;
  auto variant = [&](auto seg_, auto rows_, auto vb_, auto mode_, const char* what) {
    constexpr int SEG = decltype(seg_)::value, ROWS = decltype(rows_)::value, VB = decltype(vb_)::value, MODE = decltype(mode_)::value;
    if ((W * 3) % SEG || H % ROWS) {
      printf("  (skipped, size: %s)\n", what);
      return;
    }
    a.seg = SEG; a.rows = ROWS; a.mode = MODE; a.tiles_x = (W * 3) / SEG; a.tiles_y = H / ROWS;
    a.total = a.tiles_x * a.tiles_y * frames; a.per_xcd = (a.total + 7) / 8;
    auto run = [&]() {
      a.src = src + (size_t)set * fmax * frames; a.dst = dst + (size_t)set * fmax * frames;
      set = (set + 1) % sets;
      hipLaunchKernelGGL((k_tiles<SEG, ROWS, VB, MODE>), dim3(a.per_xcd * 8), dim3(256), 0, 0, a);
    };
    const float ms = timeit(run);
    printf("  %6.2f TB/s  %7.3f us/frame  %s\n", bytes / (ms * 1e-3) / 1e12, ms * 1e3 / frames, what);
  };
#define IC(n) std::integral_constant<int, n>{}
  variant(IC(192), IC(64), IC(12), IC(0), "the kernel's pattern: 64x64 px tiles, 192-byte segments, transposed placement");
  variant(IC(192), IC(64), IC(12), IC(1), "same tiles, straight placement");
  variant(IC(384), IC(128), IC(12), IC(0), "128x128 px tiles: 384-byte segments both ways (12-byte lanes)");
  variant(IC(384), IC(128), IC(12), IC(1), "128x128 px tiles, straight placement");
  variant(IC(768), IC(256), IC(12), IC(0), "256x256 px tiles: 768-byte segments (6 whole lines) both ways");
  variant(IC(256), IC(64), IC(16), IC(1), "64-row tiles of 256 bytes (2 whole lines), 16-byte lanes, straight placement");
  {
    const size_t n = (size_t)W * H * 3 * frames / 16;
    int set = 0;
    auto run = [&]() {
      hipLaunchKernelGGL(k_stream, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const uint4*)(src + (size_t)set * fmax * frames),
                         (uint4*)(dst + (size_t)set * fmax * frames), n);
      set = (set + 1) % sets;
    };
    const float ms = timeit(run);
    printf("  %6.2f TB/s  %7.3f us/frame  row-major streaming copy of the same bytes (16 B per lane)\n", bytes / (ms * 1e-3) / 1e12, ms * 1e3 / frames);
  }
  return 0;
}
