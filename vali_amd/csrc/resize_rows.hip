// Lanczos-3 / bicubic resize of planes that GROW vertically (dst_h > src_h: the upscale to display size), rows first.
//
// Specification (oracle/vali_oracle.c resize_plane_taps, the src_h < dst_h branch; reference call sites
// src/TC/src/TaskResizeSurface.cpp:67,116,224,273 and UDSurface.cpp:45,72 -- NPPI_INTER_LANCZOS everywhere):
//   h_s[x] = e + o,  e = wx0 t0 ; e = fma(wx_k, t_k, e) over the even taps, o likewise over the odd ones   (source row s)
//   v      = wy0 h_0 ; v = fma(wy_r, h_r, v)                                                                (window rows r)
// h depends on (source row, dst x) only, so a source row is filtered along x ONCE and every dst row is a vertical
// combination of TAPS filtered rows.  resize_taps.hip (round 2) does that through two LDS structures -- the staged BYTES of
// the source row, from which every lane funnel-shifts and converts its own 6 taps, and a ring of filtered rows that the
// vertical pass reads back -- 44 lane-instructions per output sample, 122 VGPRs, 4 waves per SIMD: 0.19 of the HBM
// roofline at 720p -> 1080p.  This kernel keeps the arithmetic and changes where the data sits:
//
//   * an enlargement re-uses every source sample ~TAPS / scale times along the row, so the row is converted to FLOAT once,
//     at staging time (4 pixels per lane and row: one 4 / 8 / 16-byte load, TAPS rows in flight), channels de-interleaved
//     (every plane is the one-channel problem);
//   * the staged row exists twice, A[j] = s[j] and B[j] = s[j + 1] (the neighbour's first pixel arrives by DPP
//     `wave_shl:1`): the six taps of a sample are three ALIGNED 8-byte reads of (t0,t1) (t2,t3) (t4,t5) from A or from B by
//     the parity of its first tap -- exactly the (even, odd) operand pairs of three `v_pk_fma_f32`, no shifts, no converts;
//   * a lane owns 2 ADJACENT dst elements in each of 2 groups of 128 (elements 2l, 2l+1 and 128+2l, 128+2l+1 of the
//     tile): neighbouring lanes read 2 x scale floats apart, so the 32 lanes of an LDS pass stay inside one 64-bank window at
//     every enlargement >= 1.07 (no bank conflicts), and the pair is the natural operand of the vertical `v_pk_fma_f32`;
//   * the TAPS filtered rows of the vertical window live in REGISTERS (TAPS x 2 pairs per lane; the walk over the source
//     rows is unrolled TAPS times, so the ring position is a compile-time constant): the vertical pass reads no LDS;
//   * the row weights of a wave's dst rows sit in LDS as pre-splatted pairs (w,w) and arrive as broadcast 16-byte reads:
//     3 LDS instructions per dst row instead of 7 `v_readlane`, and no register copies in front of the packed FMAs;
//   * 8-bit output: the 2 x 2 bytes of a lane and those of its neighbour (DPP quad_perm) make one dword per lane -- even
//     lanes store the pair's bytes of group 0, odd lanes those of group 1 (two contiguous 128-byte runs per wave and row).
//
// Geometries that do not fit (packed RGB and float planes, source spans wider than 64 groups of 4 pixels, i.e. planes
// that grow vertically but shrink along x) stay on k_resize_taps; rows whose last 4-pixel group would read past the
// pitch (foreign tight-pitch memory with a width that is not a multiple of 4) take the direct-gather form below.
#include "resize_common.hpp"
#include "resize_weights.hpp"

#include <type_traits>

namespace vali {

// whole-tile stores of these kernels are non-temporal (aux bit 1): a wave writes its rows once, 4 - 5 KB apart, and nobody reads
// them back -- as plain stores they pass through the L2 like data to keep and the LOADS queue behind their write-backs:
// RGB 720p -> 1600x900 3.77 us plain, 2.65 nt, 2.60 with every store aimed at one row, 2.40 without stores (round 5)
constexpr int kStoreNt = 2;
constexpr int kRwPadL = 4;       // floats in front of the first staged pixel (left edge replicas; keeps 16-byte alignment)
constexpr int kRwTile = 256;     // dst ELEMENTS per wave and row: 2 groups x 64 lanes x 2 adjacent elements

// Rows a wave of a 64-row instantiation really takes (round 6): the host counts ceil(dh / 256) workgroups down a plane whatever
// this returns, and the rows are spread EVENLY over their 4 waves -- 900 rows are 4 workgroups of 4 x 57 instead of three full ones
// and one whose third wave has 4 rows and whose fourth has none while the workgroup holds its slot for a full wave's time
// (720p -> 1600x900: -10 %, profiles/r06_growing.md).  Shorter instantiations (small launches) keep their fixed count.
template <int ROWS> __device__ __forceinline__ int rows_of_wave(int dh) {
  if constexpr (ROWS < 64)
    return ROWS;
  const int groups = (dh + 4 * ROWS - 1) / (4 * ROWS);
  return (dh + 4 * groups - 1) / (4 * groups);
}

// (bound_ctrl = 1: no `old` operand to copy into the destination first -- lane 63's value is never used, every lane of a
// quad_perm has a source)
__device__ __forceinline__ float rw_wave_shl1(float v) { // lane l gets lane l + 1's value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ u32 rw_swap_pairs(u32 v) { // lane l gets lane l ^ 1's value (quad_perm:[1,0,3,2])
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, true);
}

// the 4 pixels x ES channels of one lane's group, as loaded
template <typename T, int ES> struct RwGroup {
  static constexpr int kWords = 4 * ES * (int)sizeof(T) / 4; // 1, 2, 2, 4 dwords
  u32 w[kWords];
};
template <typename T, int ES> __device__ __forceinline__ RwGroup<T, ES> rw_load(const uint8_t* p) {
  RwGroup<T, ES> g;
  if constexpr (RwGroup<T, ES>::kWords == 1) {
    g.w[0] = gload_u<u32>(p);
  } else if constexpr (RwGroup<T, ES>::kWords == 2) {
    const v2u32 q = gload_u<v2u32>(p);
    g.w[0] = q.x; g.w[1] = q.y;
  } else {
    const v4u32 q = gload_u<v4u32>(p);
    g.w[0] = q.x; g.w[1] = q.y; g.w[2] = q.z; g.w[3] = q.w;
  }
  return g;
}
// element n (= pixel * ES + channel) of the group as a float
template <typename T, int ES, int N> __device__ __forceinline__ float rw_elem(const RwGroup<T, ES>& g) {
  if constexpr (sizeof(T) == 1)
    return ubyte_f32<N % 4>(g.w[N / 4]);
  else
    return (float)((N % 2) ? (g.w[N / 2] >> 16) : (g.w[N / 2] & 0xffffu));
}

template <int TAPS> __device__ __forceinline__ float rw_dot(const v2f32 (&wp)[TAPS / 2], const v2f32 (&t)[TAPS / 2]) {
  v2f32 acc = wp[0] * t[0];
#pragma unroll
  for (int j = 1; j < TAPS / 2; ++j)
    acc = __builtin_elementwise_fma(wp[j], t[j], acc);
  return acc.x + acc.y;
}

// LDS of ONE wave (floats): [channel 0: A (kRwSf) B (kRwSf)] [channel 1: A B] [wtab: ROWS x (4 + 2 TAPS)] [cnt: 64].
// kRwSf = 288: room for all 64 groups of 4 pixels plus both pads (no lane is ever predicated off in the staging writes), and
// 288 = 32 (mod 64): a lane reading copy A and one reading copy B at the same row position sit 32 banks apart.
constexpr int kRwSf = 288;

__host__ __device__ constexpr int rw_wave_floats(int es, int rows, int taps) { return es * 2 * kRwSf + rows * (4 + 2 * taps) + 64; }

// One wave: 256 ELEMENTS x ROWS dst rows of a plane whose pixels are ES interleaved elements (1: Y and the planes of
// planar formats, 2: the UV plane of NV12 / P10).  The four waves of a workgroup are stacked down the rows and share their
// columns: each evaluates ONE of the four column tap sets a lane needs (a Lanczos weight set is ~110 instructions) and
// publishes it through LDS.
template <typename T, int ES, int TAPS, int ROWS>
__device__ __forceinline__ void rows_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                          int dw, int dh, u32 tx, u32 ty, bool allow_staged, float* wg_lds, int wave_floats) {
  static_assert(ROWS <= kWave, "row taps are evaluated one row per lane");
  constexpr int kBefore = LzTap<TAPS>::kBefore;
  constexpr int EB = (int)sizeof(T);
  constexpr int WT = 4 + 2 * TAPS;    // floats per row of the weight table: [i, -, -, -, (w0,w0), (w1,w1) ...]
  constexpr int NS = ES == 1 ? 2 : 1; // tap sets per group: two pixels (ES = 1) or one pixel's (U, V)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int e0 = tx * kRwTile;
  const int dwe = dw * ES;
  const int rows_w = __builtin_amdgcn_readfirstlane(rows_of_wave<ROWS>(dh));
  const int y_first = (ty * kWavesPerBlock + wave) * rows_w;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  float* const lds = wg_lds + wave * wave_floats;
  float* const stage = lds;
  float* const wtab = lds + ES * 2 * kRwSf;
  int* const cnt = reinterpret_cast<int*>(wtab + ROWS * WT);

  // wave-uniform source span of the tile, in pixels (the same for the four waves of the workgroup)
  const int px_first = e0 / ES, px_last = min(e0 + kRwTile - 1, dwe - 1) / ES;
  const int ux0 = (int)__builtin_floorf((float)px_first * scale_x) - kBefore;            // unclamped
  const int ux1 = min((int)__builtin_floorf((float)px_last * scale_x), sw) + TAPS - 1 - kBefore;
  const int pix0 = max(ux0, 0) & ~3;                           // first staged pixel: group-aligned in the plane
  const int ngroups = (ux1 - pix0) / 4 + 1;                    // groups of 4 pixels up to the last tap
  const bool pad_left = ux0 < 0, pad_right = ux1 > sw - 1;
  const bool staged = allow_staged && ngroups <= kWave && (long long)((sw + 3) & ~3) * (ES * EB) <= (long long)spitch;

  // this lane's output elements: group g -> elements e0 + 128 g + 2 lane, + 1
  const int eg[2] = {e0 + 2 * lane, e0 + 128 + 2 * lane};

  if (staged) {
    // ---- column tap sets: wave w evaluates set w (ES = 1: (group, pixel) = (w >> 1, w & 1); ES = 2: group w, waves 0 / 1)
    v2f32 wq[2][NS][TAPS / 2];
    int addr[2][NS];
    {
      const int g = ES == 1 ? wave >> 1 : wave & 1, q = ES == 1 ? wave & 1 : 0;
      const int e = min(e0 + 128 * g + 2 * lane + q, dwe - 1);
      const LzTap<TAPS> c = make_lz_tap<TAPS>(e / ES, scale_x);
      float* mine = lds + 8 * lane;
      *reinterpret_cast<float4*>(mine) = make_float4(__builtin_bit_cast(float, c.i), c.w[0], c.w[1], c.w[2]);
      if constexpr (TAPS == 6)
        *reinterpret_cast<float4*>(mine + 4) = make_float4(c.w[3], c.w[4], c.w[5], 0.0f);
      else
        mine[4] = c.w[3];
      __syncthreads();
#pragma unroll
      for (int gg = 0; gg < 2; ++gg)
#pragma unroll
        for (int qq = 0; qq < NS; ++qq) {
          const float* from = wg_lds + (ES == 1 ? gg * 2 + qq : gg) * wave_floats + 8 * lane;
          const float4 a = *reinterpret_cast<const float4*>(from);
          float w[6] = {a.y, a.z, a.w, 0.0f, 0.0f, 0.0f};
          if constexpr (TAPS == 6) {
            const float4 b = *reinterpret_cast<const float4*>(from + 4);
            w[3] = b.x; w[4] = b.y; w[5] = b.z;
          } else {
            w[3] = from[4];
          }
#pragma unroll
          for (int j = 0; j < TAPS / 2; ++j)
            wq[gg][qq][j] = (v2f32){w[2 * j], w[2 * j + 1]};
          const int j0 = kRwPadL + (min(__builtin_bit_cast(int, a.x), sw) - kBefore) - pix0; // stage index of tap 0
          addr[gg][qq] = ((j0 & 1) ? kRwSf + j0 - 1 : j0) * 4;                                // odd: copy B, one float earlier
        }
      __syncthreads(); // everybody has read: the regions become the waves' own stages
    }
    if (y_first >= dh)
      return;

    // ---- row taps: lane r evaluates row y_first + r and publishes it (weights pre-splatted for the packed FMAs); cnt[t] =
    // how many dst rows complete their window with the wave's t-th source row
    const LzTap<TAPS> vy = make_lz_tap<TAPS>(y_first + (lane & (ROWS - 1)), scale_y);
    const int last_rr = min(rows_w, dh - y_first) - 1;
    const int s_begin = __builtin_amdgcn_readlane(vy.i, 0) - kBefore;
    const int s_end = __builtin_amdgcn_readlane(vy.i, last_rr) + TAPS - 1 - kBefore;
    cnt[lane] = 0;
    if (lane < ROWS) {
      float* row = wtab + lane * WT;
#pragma unroll
      for (int k = 0; k < TAPS / 2; ++k)
        *reinterpret_cast<float4*>(row + 4 + 4 * k) = make_float4(vy.w[2 * k], vy.w[2 * k], vy.w[2 * k + 1], vy.w[2 * k + 1]);
    }
    wave_lds_sync();
    if (lane <= last_rr)
      __hip_atomic_fetch_add(cnt + (vy.i + TAPS - 1 - kBefore - s_begin), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    wave_lds_sync();
    const int cntv = cnt[lane]; // (ROWS * scale_y + TAPS <= 64 source rows: the plane grows)

    const uint8_t* const stage_b = reinterpret_cast<const uint8_t*>(stage);
    // staging: lane m owns pixels pix0 + 4m .. + 3 (lanes past the span re-read its last group: their floats are never used)
    const int last_group = min((sw - 1) & ~3, pix0 + 4 * (ngroups - 1));
    const u32 goff = (u32)(min(pix0 + 4 * lane, last_group) * (ES * EB));
    // right edge: pixels past sw - 1 take the last pixel's value.  keep[k]: pixel k of this lane's group is its own
    const bool edge = pad_left || pad_right;                                   // wave-uniform
    const int edge_lane = (sw - 1 - pix0) >> 2, edge_k = (sw - 1 - pix0) & 3;  // wave-uniform
    bool keep[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      keep[k] = pix0 + 4 * lane + k <= sw - 1;

    RwGroup<T, ES> pf[TAPS];
    auto issue = [&](int logical, RwGroup<T, ES>& q) {
      q = rw_load<T, ES>(sp + (u32)(clampi(logical, sh - 1) * spitch) + goff);
    };
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      issue(s_begin + j, pf[j]);
      __builtin_amdgcn_sched_barrier(0); // rows in ISSUE order: vmcnt retires in order
    }

    v2f32 ring[TAPS][2]; // filtered rows of the vertical window: [slot][group] = (element 2l, element 2l + 1)
#pragma unroll
    for (int j = 0; j < TAPS; ++j)
      ring[j][0] = ring[j][1] = (v2f32){0.0f, 0.0f};

    // output: 8-bit planes exchange their byte pairs so that every lane stores one dword; the lane's pointer walks down the rows
    const bool odd = lane & 1;
    const int est = EB == 1 ? (odd ? e0 + 128 + 2 * (lane - 1) : e0 + 2 * lane) : eg[0]; // first element behind `optr`
    const u32 sel = odd ? 0x03020706u : 0x05040100u;  // v_perm_b32 {neighbour: bytes 4-7, own: bytes 0-3}, see emit
    const bool full = e0 + kRwTile <= dwe;            // wave-uniform: every lane of the tile has all its elements
    uint8_t* optr = dp + (size_t)y_first * dpitch + (size_t)est * EB;
    const float* wt = wtab + 4;                       // weights of the next dst row

#pragma unroll 1
    for (int s0 = s_begin; s0 <= s_end; s0 += TAPS) {
#pragma unroll
      for (int j = 0; j < TAPS; ++j) {
        const int cur = s0 + j;
        const bool live = cur <= s_end; // wave-uniform; rows past the end skip the work, never the load
        if (live) {
          // ---- stage source row `cur`: convert once, de-interleave, write A and B
          auto stage_channel = [&](auto ctag) {
            constexpr int C = decltype(ctag)::value;
            float f[4] = {rw_elem<T, ES, 0 * ES + C>(pf[j]), rw_elem<T, ES, 1 * ES + C>(pf[j]),
                          rw_elem<T, ES, 2 * ES + C>(pf[j]), rw_elem<T, ES, 3 * ES + C>(pf[j])};
            float* const a = stage + C * 2 * kRwSf;
            if (edge) { // the tiles at the image's left / right edge only
              if (pad_right) {
                const float sel_k = edge_k == 0 ? f[0] : edge_k == 1 ? f[1] : edge_k == 2 ? f[2] : f[3];
                const float ev = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sel_k), edge_lane));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                  f[k] = keep[k] ? f[k] : ev;
              }
              if (pad_left && lane == 0) { // pixels -1, -2 replicate pixel 0 (pix0 == 0 here)
                *reinterpret_cast<float4*>(a) = make_float4(f[0], f[0], f[0], f[0]);
                *reinterpret_cast<float4*>(a + kRwSf) = make_float4(f[0], f[0], f[0], f[0]);
              }
            }
            const float fn = rw_wave_shl1(f[0]);
            *reinterpret_cast<float4*>(a + kRwPadL + 4 * lane) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(a + kRwSf + kRwPadL + 4 * lane) = make_float4(f[1], f[2], f[3], fn);
          };
          stage_channel(std::integral_constant<int, 0>{});
          if constexpr (ES == 2)
            stage_channel(std::integral_constant<int, 1>{});
          wave_lds_sync();
        }
        issue(cur + TAPS, pf[j]); // TAPS rows ahead
        if (!live)
          continue;
        // ---- horizontal pass: 2 groups x 2 elements, three aligned pair reads per element; the four chains side by side
        {
          // (one statement: the compiler pairs adjacent 8-byte reads into ds_read2_b64, which costs the LDS pipe twice the
          // two plain reads, MI355X_MICROARCH.md LDS table; it does not track LDS reads issued from assembly, hence the wait)
          v2f32 t[4][TAPS / 2];
          u32 la[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            la[u] = (u32)(size_t)stage_b + (u32)addr[u >> 1][ES == 1 ? (u & 1) : 0];
          constexpr int CH = ES == 2 ? 2 * kRwSf * 4 : 0; // byte offset of the second channel's copies
          if constexpr (TAPS == 6)
            asm volatile("ds_read_b64 %0, %12\n\tds_read_b64 %3, %13 offset:%16\n\tds_read_b64 %6, %14\n\tds_read_b64 %9, %15 offset:%16\n\t"
                         "ds_read_b64 %1, %12 offset:8\n\tds_read_b64 %4, %13 offset:%17\n\tds_read_b64 %7, %14 offset:8\n\tds_read_b64 %10, %15 offset:%17\n\t"
                         "ds_read_b64 %2, %12 offset:16\n\tds_read_b64 %5, %13 offset:%18\n\tds_read_b64 %8, %14 offset:16\n\tds_read_b64 %11, %15 offset:%18\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[0][2]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[1][2]),
                           "=&v"(t[2][0]), "=&v"(t[2][1]), "=&v"(t[2][2]), "=&v"(t[3][0]), "=&v"(t[3][1]), "=&v"(t[3][2])
                         : "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "n"(CH), "n"(CH + 8), "n"(CH + 16)
                         : "memory");
          else
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %2, %9 offset:%12\n\tds_read_b64 %4, %10\n\tds_read_b64 %6, %11 offset:%12\n\t"
                         "ds_read_b64 %1, %8 offset:8\n\tds_read_b64 %3, %9 offset:%13\n\tds_read_b64 %5, %10 offset:8\n\tds_read_b64 %7, %11 offset:%13\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(t[0][0]), "=&v"(t[0][1]), "=&v"(t[1][0]), "=&v"(t[1][1]), "=&v"(t[2][0]), "=&v"(t[2][1]),
                           "=&v"(t[3][0]), "=&v"(t[3][1])
                         : "v"(la[0]), "v"(la[1]), "v"(la[2]), "v"(la[3]), "n"(CH), "n"(CH + 8)
                         : "memory");
          v2f32 acc[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            acc[u] = wq[u >> 1][ES == 1 ? (u & 1) : 0][0] * t[u][0];
#pragma unroll
          for (int k = 1; k < TAPS / 2; ++k)
#pragma unroll
            for (int u = 0; u < 4; ++u)
              acc[u] = __builtin_elementwise_fma(wq[u >> 1][ES == 1 ? (u & 1) : 0][k], t[u][k], acc[u]);
          float h[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            h[u] = acc[u].x + acc[u].y;
          asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])); // (keeps the sums scalar: packed, they cost register copies)
          ring[j][0] = (v2f32){h[0], h[1]};
          ring[j][1] = (v2f32){h[2], h[3]};
        }
        wave_lds_sync(); // the stage is refilled by the next row
        // ---- every dst row whose window ends with this source row (cnt: several when enlarging); slot j is the newest row,
        // logical row r of the window sits in slot (j + 1 + r) mod TAPS
        auto emit = [&]() {
          v2f32 wy[TAPS];
#pragma unroll
          for (int k = 0; k < TAPS / 2; ++k) {
            const float4 q4 = *reinterpret_cast<const float4*>(wt + 4 * k);
            wy[2 * k] = (v2f32){q4.x, q4.y};
            wy[2 * k + 1] = (v2f32){q4.z, q4.w};
          }
          wt += WT;
          v2f32 v[2];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            v[g] = wy[0] * ring[(j + 1) % TAPS][g];
#pragma unroll
            for (int r = 1; r < TAPS; ++r)
              v[g] = __builtin_elementwise_fma(wy[r], ring[(j + 1 + r) % TAPS][g], v[g]);
          }
          if constexpr (EB == 1) {
            u32 p = 0;
            p = __builtin_amdgcn_cvt_pk_u8_f32(v[0].x, 0u, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(v[0].y, 1u, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(v[1].x, 2u, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(v[1].y, 3u, p);
            // even lane: (own group 0, neighbour's group 0) = elements 2l .. 2l + 3; odd lane: (neighbour's group 1, own
            // group 1) = elements 128 + 2(l - 1) .. + 3
            const u32 word = __builtin_amdgcn_perm(rw_swap_pairs(p), p, sel);
            if (full) {
              gstore_u<u32>(optr, word);
            } else {
              const int n = dwe - est;
#pragma unroll
              for (int k = 0; k < 4; ++k)
                if (k < n)
                  ((VALI_GLOBAL uint8_t*)optr)[k] = (uint8_t)(word >> (8 * k));
            }
          } else {
            const u32 w0 = finish_bits<T>(v[0].x) | (finish_bits<T>(v[0].y) << 16);
            const u32 w1 = finish_bits<T>(v[1].x) | (finish_bits<T>(v[1].y) << 16);
            if (full) {
              gstore_u<u32>(optr, w0);
              gstore_u<u32>(optr + 128 * EB, w1);
            } else {
              const int n = dwe - est;
              if (n > 0) ((VALI_GLOBAL uint16_t*)optr)[0] = (uint16_t)w0;
              if (n > 1) ((VALI_GLOBAL uint16_t*)optr)[1] = (uint16_t)(w0 >> 16);
              if (n > 128) ((VALI_GLOBAL uint16_t*)optr)[128] = (uint16_t)w1;
              if (n > 129) ((VALI_GLOBAL uint16_t*)optr)[129] = (uint16_t)(w1 >> 16);
            }
          }
          optr += dpitch;
        };
        // (straight-line first: the compiler drains vmcnt in front of a loop that stores without loading, DESIGN.md 5d #4,
        // which would throw the prefetched rows away; enlargements below 3x never reach the loop)
        const int c = __builtin_amdgcn_readlane(cntv, cur - s_begin);
        if (c > 0) {
          emit();
          if (c > 1) {
            emit();
            for (int k = 2; k < c; ++k)
              emit();
          }
        }
      }
    }
    return;
  }
  if (y_first >= dh)
    return;

  // Direct gather (tight-pitch foreign memory whose last group would read past the row, forced by the tuning switch):
  // TAPS x TAPS taps per element, the specification's loops as they stand.  Rolled: this path is about correctness, and
  // must not cost the staged path registers.
  {
    const LzTap<TAPS> vy = make_lz_tap<TAPS>(y_first + (lane & (ROWS - 1)), scale_y);
    if (lane < ROWS) {
      float* row = wtab + lane * WT;
      row[0] = __builtin_bit_cast(float, vy.i);
#pragma unroll
      for (int k = 0; k < TAPS; ++k)
        row[4 + 2 * k] = vy.w[k];
    }
  }
  const int last_rr = min(rows_w, dh - y_first) - 1;
  wave_lds_sync();
#pragma unroll 1
  for (int rr = 0; rr <= last_rr; ++rr) {
    const float* wt = wtab + rr * WT;
    const int iy = __builtin_bit_cast(int, wt[0]);
#pragma unroll 1
    for (int k4 = 0; k4 < 4; ++k4) {
      const int e = eg[k4 >> 1] + (k4 & 1);
      if (e >= dwe)
        continue;
      const int px = e / ES, ch = e - px * ES;
      const LzTap<TAPS> c = make_lz_tap<TAPS>(px, scale_x);
      float v = 0.0f;
#pragma unroll 1
      for (int r = 0; r < TAPS; ++r) {
        const uint8_t* row = sp + (size_t)clampi(iy - kBefore + r, sh - 1) * spitch;
        float t[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS; ++k)
          t[k] = (float)gload<T>(row + (size_t)(clampi(c.i - kBefore + k, sw - 1) * ES + ch) * EB);
        float he = c.w[0] * t[0], ho = c.w[1] * t[1];
#pragma unroll
        for (int k = 2; k < TAPS; k += 2) {
          he = __builtin_fmaf(c.w[k], t[k], he);
          ho = __builtin_fmaf(c.w[k + 1], t[k + 1], ho);
        }
        const float hs = he + ho;
        const float wr = wt[4 + 2 * r];
        v = r == 0 ? wr * hs : __builtin_fmaf(wr, hs, v);
      }
      ((VALI_GLOBAL T*)(dp + (size_t)(y_first + rr) * dpitch))[e] = (T)finish_bits<T>(v);
    }
  }
}

// ESSET: 1 = one-channel planes only, 12 = a one-channel and a two-channel plane (NV12 / P10), 2 = two-channel only
template <typename T, int ESSET, int TAPS, int ROWS>
__global__ void __launch_bounds__(kBlock) k_resize_rows(const ResizeArgs a) {
  extern __shared__ uint4 rows_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  float* const lds = reinterpret_cast<float*>(rows_lds);
  const bool allow = !a.force_gather;
  if (ESSET != 1 && job.channels == 2)
    rows_tile<T, 2, TAPS, ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, allow, lds, a.lds_per_wave / 4);
  else if constexpr (ESSET != 2)
    rows_tile<T, 1, TAPS, ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, allow, lds, a.lds_per_wave / 4);
}

// =====================================================================================================================
// The same rows-first arithmetic WITHOUT the LDS stage (round 4): 8-bit planes of 1 / 2 channels that grow along x as well
// (1 / 3 < scale_x < 1).  profiles/r04_lanczos.md: what the staged form above pays for is not its arithmetic but two LDS
// round trips per source row (stage write -> 12 tap reads) in waves that have three neighbours per SIMD.  Here a lane owns 4
// ADJACENT dst elements and loads, per source row, the 12 (16) source bytes their windows span from ITS OWN byte address
// floor(x0 s) - 2 -- whatever its alignment; neighbouring lanes overlap by two thirds, in the L1 -- converts them once and
// filters from registers: dst pixel x0 + j starts its window D_j or D_j + 1 registers in (D_j = floor(j s): a property of
// the scale, a template parameter; the + 1 varies from lane to lane), so its chains run over SEVEN registers D_j .. D_j + 6
// with the weights (w0 .. w5, 0) or (0, w0 .. w5): the specification's e over the even taps and o over the odd ones swap
// places in the second case; their members, their order and their sum do not (0 * t = +-0, and w * t + (+-0) = w * t).
// Two-channel planes: the lane's 4 elements are 2 pixels, 8 source pixels = 16 bytes, a register pair is one pixel's (U, V)
// and a tap one packed FMA whose weight is one half of a VGPR pair (op_sel).  Image edges: a lane whose window passes the
// first / last pixel loads from the clamped address and moves its bytes with the replica shifted in (v_alignbyte_b32 under
// a per-lane condition; edge tiles only).  The vertical pass is the staged form's: TAPS filtered rows in registers, row
// weights as broadcast LDS reads, dst rows emitted as the source rows complete them.
template <int D2, int D3> __device__ __forceinline__ constexpr int rwr_d(int q) { return q <= 1 ? 0 : q == 2 ? D2 : D3; }
template <int H> __device__ __forceinline__ void rwr_pk_fma(v2f32& acc, v2f32 w, v2f32 f) { // (w[H], w[H]) * f + acc
  if constexpr (H == 0)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(f));
  else
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(f));
}
// pixel K of packed RGB bytes held in dwords: (R, G) and B as floats
template <int K> __device__ __forceinline__ void rgb_window_px(const u32 (&w)[7], v2f32& rg, float& b) {
  rg = (v2f32){ubyte_f32<(3 * K) % 4>(w[(3 * K) / 4]), ubyte_f32<(3 * K + 1) % 4>(w[(3 * K + 1) / 4])};
  b = ubyte_f32<(3 * K + 2) % 4>(w[(3 * K + 2) / 4]);
}
constexpr int kRrShare = 512; // floats of a wave's LDS in front of its weight table: one column tap set (64 lanes x 8)

template <int ES, int ROWS, int D2, int D3>
__device__ __forceinline__ void rows_reg_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                              int dw, int dh, u32 tx, u32 ty, float* wg_lds, int wave_floats) {
  using T = uint8_t;
  constexpr int TAPS = 6, kBefore = LzTap<TAPS>::kBefore;
  constexpr int WT = 4 + 2 * TAPS;
  constexpr int ND = ES == 1 ? 3 : 4;         // dwords a lane loads per source row: 12 / 16 bytes
  constexpr int NPX = 4 / ES;                 // dst pixels of the lane
  constexpr int NCOL = ES == 1 ? 12 : 8;      // source pixels in its window
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int dwe = dw * ES;
  const int e0 = (int)tx * kRwTile;
  const int nl = (min(e0 + kRwTile, dwe) - e0 + 3) >> 2;         // lanes with elements
  const int eb = e0 + 4 * min(lane, nl - 1);                    // (lanes past the row repeat its last lane, and store nothing)
  const int n_out = lane < nl ? min(4, dwe - eb) : 0;
  const int rows_w = __builtin_amdgcn_readfirstlane(rows_of_wave<ROWS>(dh));
  const int y_first = (ty * kWavesPerBlock + wave) * rows_w;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  float* const lds = wg_lds + wave * wave_floats;
  float* const wtab = lds + kRrShare;
  int* const cnt = reinterpret_cast<int*>(wtab + ROWS * WT);
  const int x0 = eb / ES;

  // ---- column taps: the four waves of the workgroup share their columns; wave w evaluates pixel w % NPX of every lane ----
  float W[NPX][7];
  int first;                                                    // source pixel in the lane's register 0
  {
    const LzTap<TAPS> c = make_lz_tap<TAPS>(min(x0 + wave % NPX, dw - 1), scale_x);
    float* mine = lds + 8 * lane;
    *reinterpret_cast<float4*>(mine) = make_float4(__builtin_bit_cast(float, c.i), c.w[0], c.w[1], c.w[2]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(c.w[3], c.w[4], c.w[5], 0.0f);
    __syncthreads();
    int i0 = 0;
#pragma unroll
    for (int q = 0; q < NPX; ++q) {
      const float* from = wg_lds + q * wave_floats + 8 * lane;
      const float4 a = *reinterpret_cast<const float4*>(from), b = *reinterpret_cast<const float4*>(from + 4);
      const int iq = __builtin_bit_cast(int, a.x);
      if (q == 0)
        i0 = iq;
      const int dq = ES == 1 ? rwr_d<D2, D3>(q) : 0;              // (a constant once the loop is unrolled)
      const bool late = iq - i0 > dq;                           // its window starts one register further on (q = 0: never)
      const float w[6] = {a.y, a.z, a.w, b.x, b.y, b.z};
      W[q][0] = late ? 0.0f : w[0];
#pragma unroll
      for (int k = 1; k < 6; ++k)
        W[q][k] = late ? w[k - 1] : w[k];
      W[q][6] = late ? w[5] : 0.0f;
    }
    first = i0 - kBefore;
    __syncthreads();
  }
  if (y_first >= dh)
    return;

  // ---- row taps (as in rows_tile): lane r evaluates row y_first + r; cnt[t]: dst rows the wave's t-th source row completes
  const LzTap<TAPS> vy = make_lz_tap<TAPS>(y_first + (lane & (ROWS - 1)), scale_y);
  const int last_rr = min(rows_w, dh - y_first) - 1;
  const int s_begin = __builtin_amdgcn_readlane(vy.i, 0) - kBefore;
  const int s_end = __builtin_amdgcn_readlane(vy.i, last_rr) + TAPS - 1 - kBefore;
  cnt[lane] = 0;
  if (lane < ROWS) {
    float* row = wtab + lane * WT;
#pragma unroll
    for (int k = 0; k < TAPS / 2; ++k)
      *reinterpret_cast<float4*>(row + 4 + 4 * k) = make_float4(vy.w[2 * k], vy.w[2 * k], vy.w[2 * k + 1], vy.w[2 * k + 1]);
  }
  wave_lds_sync();
  if (lane <= last_rr)
    __hip_atomic_fetch_add(cnt + (vy.i + TAPS - 1 - kBefore - s_begin), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  wave_lds_sync();
  const int cntv = cnt[lane];

  // ---- the lane's window: source pixels first .. first + NCOL - 1, loaded from a0, bytes moved |mv| pixels at the edges ----
  const int a0 = clampi(first, sw - NCOL);                      // (the host refuses planes narrower than the window)
  const int mv = first - a0;                                    // < 0: the left edge, > 0: the right edge
  const bool any_left = __builtin_amdgcn_readfirstlane((int)(__ballot(mv < 0) != 0ull)) != 0;
  const bool any_right = __builtin_amdgcn_readfirstlane((int)(__ballot(mv > 0) != 0ull)) != 0;
  const u32 goff = (u32)(a0 * ES);
  struct Row { u32 w[ND]; };
  // (through a raw buffer descriptor: the row's offset is the scalar operand, the lane's the one VGPR of the address)
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sp), (short)0, (int)0xffffffffu, 0x00020000);
  auto issue = [&](int logical, Row& q) {
    const int row = clampi(logical, sh - 1) * spitch;
    if constexpr (ND == 3) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 v = __builtin_amdgcn_raw_buffer_load_b96(srsrc, (int)goff, row, 0);
      q.w[0] = v.x; q.w[1] = v.y; q.w[2] = v.z;
    } else {
      const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)goff, row, 0);
      q.w[0] = v.x; q.w[1] = v.y; q.w[2] = v.z; q.w[3] = v.w;
    }
  };
  Row pf[TAPS];
#pragma unroll
  for (int j = 0; j < TAPS; ++j) {
    issue(s_begin + j, pf[j]);
    __builtin_amdgcn_sched_barrier(0); // rows in ISSUE order: vmcnt retires in order
  }
  v2f32 ring[TAPS][2]; // filtered rows of the vertical window: [slot] = (elements 0, 1), (elements 2, 3) of the lane
#pragma unroll
  for (int j = 0; j < TAPS; ++j)
    ring[j][0] = ring[j][1] = (v2f32){0.0f, 0.0f};
  uint8_t* optr = dp + (size_t)y_first * dpitch + (size_t)eb;
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dp, (short)0, (int)0xffffffffu, 0x00020000);
  int orow = y_first * dpitch;                      // the dst row's offset (scalar): whole groups store through the descriptor
  const float* wt = wtab + 4;                       // weights of the next dst row

#pragma unroll 1
  for (int s0 = s_begin; s0 <= s_end; s0 += TAPS) {
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      const int cur = s0 + j;
      const bool live = cur <= s_end; // wave-uniform; rows past the end skip the work, never the load
      u32 w[ND];
#pragma unroll
      for (int k = 0; k < ND; ++k)
        w[k] = pf[j].w[k];
#pragma unroll
      for (int k = 0; k < ND; ++k)
        asm volatile("" : "+v"(w[k])); // (the row's registers are read before the load below takes them)
      __builtin_amdgcn_sched_barrier(0);
      issue(cur + TAPS, pf[j]); // TAPS rows ahead
      __builtin_amdgcn_sched_barrier(0);
      if (!live)
        continue;
      // (the distance in bytes, one binary digit at a time: whole dwords, then v_alignbyte_b32; what enters is the replica)
      if (any_left) {      // replicas of the first pixel enter in front: 1 or 2 pixels
        const u32 rep = ES == 1 ? (w[0] & 0xffu) * 0x01010101u : (w[0] & 0xffffu) * 0x00010001u;
        const int nb = max(-mv, 0) * ES;                          // bytes, <= 4 (0: not this lane)
        if constexpr (ES == 2) {
          const bool go = (nb & 4) != 0;
#pragma unroll
          for (int k = ND - 1; k >= 0; --k)
            w[k] = go ? (k ? w[k - 1] : rep) : w[k];
        }
        {
          const bool go = (nb & 2) != 0;
#pragma unroll
          for (int k = ND - 1; k >= 0; --k) {
            const u32 moved = __builtin_amdgcn_alignbyte(w[k], k ? w[k - 1] : rep, 2);
            w[k] = go ? moved : w[k];
          }
        }
        if constexpr (ES == 1) {
          const bool go = (nb & 1) != 0;
#pragma unroll
          for (int k = ND - 1; k >= 0; --k) {
            const u32 moved = __builtin_amdgcn_alignbyte(w[k], k ? w[k - 1] : rep, 3);
            w[k] = go ? moved : w[k];
          }
        }
      }
      if (any_right) {     // replicas of the last pixel enter behind: up to the whole window but one pixel
        const u32 rep = ES == 1 ? (w[ND - 1] >> 24) * 0x01010101u : (w[ND - 1] >> 16) * 0x00010001u;
        const int nb = min(max(mv, 0) * ES, 4 * ND - ES);         // bytes (0: not this lane)
        {
          const bool go = (nb & 8) != 0;
#pragma unroll
          for (int k = 0; k < ND; ++k)
            w[k] = go ? (k + 2 < ND ? w[k + 2] : rep) : w[k];
        }
        {
          const bool go = (nb & 4) != 0;
#pragma unroll
          for (int k = 0; k < ND; ++k)
            w[k] = go ? (k + 1 < ND ? w[k + 1] : rep) : w[k];
        }
        {
          const bool go = (nb & 2) != 0;
#pragma unroll
          for (int k = 0; k < ND; ++k) {
            const u32 moved = __builtin_amdgcn_alignbyte(k + 1 < ND ? w[k + 1] : rep, w[k], 2);
            w[k] = go ? moved : w[k];
          }
        }
        if constexpr (ES == 1) {
          const bool go = (nb & 1) != 0;
#pragma unroll
          for (int k = 0; k < ND; ++k) {
            const u32 moved = __builtin_amdgcn_alignbyte(k + 1 < ND ? w[k + 1] : rep, w[k], 1);
            w[k] = go ? moved : w[k];
          }
        }
      }
      // ---- the pass along the row, from registers
      float h[4];
      if constexpr (ES == 1) {
        float c[12];
#pragma unroll
        for (int i = 0; i < 12; ++i)
          c[i] = (float)((w[i / 4] >> (8 * (i % 4))) & 0xffu);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dq = rwr_d<D2, D3>(q);                      // (a constant once the loop is unrolled: registers, not memory)
          float e = W[q][0] * c[dq], o = W[q][1] * c[dq + 1];
          e = __builtin_fmaf(W[q][2], c[dq + 2], e);
          o = __builtin_fmaf(W[q][3], c[dq + 3], o);
          e = __builtin_fmaf(W[q][4], c[dq + 4], e);
          o = __builtin_fmaf(W[q][5], c[dq + 5], o);
          if (q > 0)
            e = __builtin_fmaf(W[q][6], c[dq + 6], e);
          h[q] = e + o;
        }
      } else {
        v2f32 cc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
          cc[i] = (v2f32){(float)((w[i / 2] >> (16 * (i % 2))) & 0xffu), (float)((w[i / 2] >> (16 * (i % 2) + 8)) & 0xffu)};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v2f32 e = (v2f32){0.0f, 0.0f}, o = (v2f32){0.0f, 0.0f};
          const v2f32 w01 = (v2f32){W[q][0], W[q][1]}, w23 = (v2f32){W[q][2], W[q][3]}, w45 = (v2f32){W[q][4], W[q][5]},
                      w6 = (v2f32){W[q][6], 0.0f};
          rwr_pk_fma<0>(e, w01, cc[0]);
          rwr_pk_fma<1>(o, w01, cc[1]);
          rwr_pk_fma<0>(e, w23, cc[2]);
          rwr_pk_fma<1>(o, w23, cc[3]);
          rwr_pk_fma<0>(e, w45, cc[4]);
          rwr_pk_fma<1>(o, w45, cc[5]);
          if (q > 0)
            rwr_pk_fma<0>(e, w6, cc[6]);
          const v2f32 v = e + o;
          h[2 * q] = v.x;
          h[2 * q + 1] = v.y;
        }
      }
      asm volatile("" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
      ring[j][0] = (v2f32){h[0], h[1]};
      ring[j][1] = (v2f32){h[2], h[3]};
      // ---- every dst row whose window ends with this source row; slot j is the newest row, logical row r of the window
      // sits in slot (j + 1 + r) mod TAPS
      auto emit = [&]() {
        v2f32 wy[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k) {
          const float4 q4 = *reinterpret_cast<const float4*>(wt + 4 * k);
          wy[2 * k] = (v2f32){q4.x, q4.y};
          wy[2 * k + 1] = (v2f32){q4.z, q4.w};
        }
        wt += WT;
        v2f32 v[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          v[g] = wy[0] * ring[(j + 1) % TAPS][g];
#pragma unroll
          for (int r = 1; r < TAPS; ++r)
            v[g] = __builtin_elementwise_fma(wy[r], ring[(j + 1 + r) % TAPS][g], v[g]);
        }
        u32 p = 0;
        p = __builtin_amdgcn_cvt_pk_u8_f32(v[0].x, 0u, p);
        p = __builtin_amdgcn_cvt_pk_u8_f32(v[0].y, 1u, p);
        p = __builtin_amdgcn_cvt_pk_u8_f32(v[1].x, 2u, p);
        p = __builtin_amdgcn_cvt_pk_u8_f32(v[1].y, 3u, p);
        if (n_out == 4) {
          __builtin_amdgcn_raw_buffer_store_b32(p, drsrc, eb, orow, kStoreNt);
        } else {
          uint8_t* const o = optr + (u32)(orow - y_first * dpitch);
#pragma unroll
          for (int k = 0; k < 3; ++k)
            if (k < n_out)
              ((VALI_GLOBAL uint8_t*)o)[k] = (uint8_t)(p >> (8 * k));
        }
        orow += dpitch;
      };
      const int c = __builtin_amdgcn_readlane(cntv, cur - s_begin);
      if (c > 0) {
        emit();
        if (c > 1) {
          emit();
          for (int k = 2; k < c; ++k)
            emit();
        }
      }
    }
  }
}

template <int ROWS, int D2, int D3>
__global__ void __launch_bounds__(kBlock) k_resize_rows_reg(const ResizeArgs a) {
  extern __shared__ uint4 rows_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  float* const lds = reinterpret_cast<float*>(rows_lds);
  if (job.channels == 2)
    rows_reg_tile<2, ROWS, D2, D3>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, lds, a.lds_per_wave / 4);
  else
    rows_reg_tile<1, ROWS, D2, D3>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, lds, a.lds_per_wave / 4);
}

// Host: may the register form take a plane of c channels?  Its lanes assume that dst pixel x0 + j starts its window D_j or
// D_j + 1 source pixels after pixel x0's, D = (0, 0, d2, d3) -- true of the real numbers, checked here against the FP32
// products the device forms.  Cached per (sw, dw, c): the check walks the row once.
struct RrFit { int sw, dw, c, ok, d2, d3; };
static bool rows_reg_fits(int sw, int dw, int c, int& d2, int& d3) {
  static thread_local RrFit cache[8];
  static thread_local int next = 0;
  for (const RrFit& f : cache)
    if (f.sw == sw && f.dw == dw && f.c == c && f.sw != 0) {
      d2 = f.d2; d3 = f.d3;
      return f.ok != 0;
    }
  const float s = (float)sw / (float)dw;
  bool ok = sw < dw && 3 * sw > dw && sw >= 16;
  int e2 = 0, e3 = 0;
  if (ok) {
    auto fl = [&](int x) { return (int)__builtin_floorf((float)x * s); };
    if (c == 1) {
      e2 = (int)__builtin_floorf(2.0f * s);
      e3 = (int)__builtin_floorf(3.0f * s);
      for (int x0 = 0; x0 < dw && ok; x0 += 4)
        for (int j = 1; j < 4 && ok; ++j) {
          if (x0 + j >= dw)
            break;
          const int b = fl(x0 + j) - fl(x0), dj = j == 1 ? 0 : j == 2 ? e2 : e3;
          ok = b == dj || b == dj + 1;
        }
      ok = ok && ((e2 == 0 && e3 == 1) || (e2 == 1 && (e3 == 1 || e3 == 2)));
    } else {
      for (int x0 = 0; x0 + 1 < dw && ok; x0 += 2) {
        const int b = fl(x0 + 1) - fl(x0);
        ok = b == 0 || b == 1;
      }
    }
  }
  cache[next] = RrFit{sw, dw, c, ok ? 1 : 0, e2, e3};
  next = (next + 1) & 7;
  d2 = e2; d3 = e3;
  return ok;
}

// =====================================================================================================================
// Packed 8-bit RGB that grows on both axes at any ratio in (1/3, 1) (round 5): the register form for three channels.  A lane
// owns 2 adjacent dst pixels (6 bytes).  Their windows span 7 source pixels -- pixel 1's starts with pixel 0's or one pixel
// later, the weights absorb the difference as above (the host checks the 0 / 1 against the FP32 products the device forms) --,
// so the lane loads the 24 bytes at ITS address 3 (floor(x0 s) - 2), whatever its alignment, converts the 21 it needs once and
// filters from registers: a tap is one packed FMA on a pixel's (R, G) -- the weight a half of a register pair -- and one FMA on
// its B.  The filtered rows of the vertical window stay in registers as (R, G) of pixel 0, (R, G) of pixel 1, (B, B): the
// column pass is three packed chains per dst row whose row weights are picked by op_sel.  Whole tiles store a quad's 24 bytes
// as three aligned 8-byte stores (as the 3:2 form below).
// Tiles at the image's left / right edge (some lane's window leaves the row): every lane loads its 7 pixels one by one from
// their clamped positions -- 4 bytes that END with the pixel, or start with it at pixel 0: nothing is read outside the row --
// and packs them into the 24 bytes the other tiles load.
constexpr int kRgbTilePx = 128;   // dst pixels per wave and row: 64 lanes x 2
template <int ROWS>
__device__ __forceinline__ void rows_rgb_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch, int dw, int dh,
                                              u32 tx, u32 ty, float* wg_lds, int wave_floats) {
  constexpr int TAPS = 6, kBefore = LzTap<TAPS>::kBefore, WT = 4 + 2 * TAPS, AHEAD = 3, NW = 7;
  static_assert(ROWS <= kWave && (ROWS & (ROWS - 1)) == 0 && TAPS % AHEAD == 0, "one row tap set per lane; slot j % AHEAD is row t % AHEAD");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int p0 = (int)tx * kRgbTilePx;
  const int nl = (min(kRgbTilePx, dw - p0) + 1) >> 1;              // lanes with pixels
  const int xb = p0 + 2 * min(lane, nl - 1);                       // (lanes past the row repeat its last lane, and store nothing)
  const int n_px = lane < nl ? min(2, dw - xb) : 0;
  const int full = __builtin_amdgcn_readfirstlane(p0 + kRgbTilePx <= dw ? 1 : 0);
  const int rows_w = __builtin_amdgcn_readfirstlane(rows_of_wave<ROWS>(dh));
  const int y_first = (int)(ty * kWavesPerBlock + wave) * rows_w;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  float* const lds = wg_lds + wave * wave_floats;
  float* const wtab = lds + kRrShare;
  int* const cnt = reinterpret_cast<int*>(wtab + ROWS * WT);

  // ---- column taps: the four waves of the workgroup share their columns; wave w evaluates pixel w & 1 of every lane ----
  v2f32 W[2][4];                                                   // (W0 W1) (W2 W3) (W4 W5) (W6 0) over the 7 registers of the window
  int first;                                                       // source pixel in the lane's register 0
  {
    const LzTap<TAPS> c = make_lz_tap<TAPS>(min(xb + (wave & 1), dw - 1), scale_x);
    float* mine = lds + 8 * lane;
    *reinterpret_cast<float4*>(mine) = make_float4(__builtin_bit_cast(float, c.i), c.w[0], c.w[1], c.w[2]);
    *reinterpret_cast<float4*>(mine + 4) = make_float4(c.w[3], c.w[4], c.w[5], 0.0f);
    __syncthreads();
    int i0 = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float* from = wg_lds + q * wave_floats + 8 * lane;
      const float4 a = *reinterpret_cast<const float4*>(from), b = *reinterpret_cast<const float4*>(from + 4);
      const int iq = __builtin_bit_cast(int, a.x);
      if (q == 0)
        i0 = iq;
      const bool late = iq > i0;                                   // its window starts one register further on (q = 0: never)
      const float w[6] = {a.y, a.z, a.w, b.x, b.y, b.z};
      float s7[8];
      s7[0] = late ? 0.0f : w[0];
#pragma unroll
      for (int k = 1; k < 6; ++k)
        s7[k] = late ? w[k - 1] : w[k];
      s7[6] = late ? w[5] : 0.0f;
      s7[7] = 0.0f;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        W[q][k] = (v2f32){s7[2 * k], s7[2 * k + 1]};
    }
    first = i0 - kBefore;
    __syncthreads();
  }
  if (y_first >= dh)
    return;

  // ---- row taps: lane r evaluates row y_first + r; cnt[t]: dst rows the wave's t-th source row completes ----
  const LzTap<TAPS> vy = make_lz_tap<TAPS>(y_first + (lane & (ROWS - 1)), scale_y);
  const int last_rr = min(rows_w, dh - y_first) - 1;
  const int s_begin = __builtin_amdgcn_readlane(vy.i, 0) - kBefore;
  const int s_end = __builtin_amdgcn_readlane(vy.i, last_rr) + TAPS - 1 - kBefore;
  cnt[lane] = 0;
  if (lane < ROWS) {
    float* row = wtab + lane * WT;
    *reinterpret_cast<float4*>(row + 4) = make_float4(vy.w[0], vy.w[1], vy.w[2], vy.w[3]);
    *reinterpret_cast<float2*>(row + 8) = make_float2(vy.w[4], vy.w[5]);
  }
  wave_lds_sync();
  if (lane <= last_rr)
    __hip_atomic_fetch_add(cnt + (vy.i + TAPS - 1 - kBefore - s_begin), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  wave_lds_sync();
  const int cntv = cnt[lane];

  // ---- the lane's window: source pixels first .. first + 6 ----
  const int a0 = clampi(first, sw - 8);                            // (the host refuses planes narrower than 16 pixels)
  const int edge_tile = __builtin_amdgcn_readfirstlane((int)(__ballot(first != a0) != 0ull));
  const u32 goff = (u32)(a0 * 3);
  u32 eoff[NW], eshift = 0;                                        // edge tiles: where pixel k's 4 bytes start; 2 bits each: by how many bytes it sits in them
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    const int idx = clampi(first + k, sw - 1);
    eoff[k] = idx > 0 ? (u32)(3 * idx - 1) : 0u;
    eshift |= (idx > 0 ? 1u : 0u) << (2 * k);
  }
  struct Row { u32 w[NW]; };
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sp), (short)0, (int)0xffffffffu, 0x00020000);
  // (the walk exists twice, for edge tiles and for the others: with both load forms behind a branch inside ONE loop the compiler
  // cannot count the loads in flight and waits for all of them -- vmcnt(0) at every row, the prefetch gone)
  auto walk = [&](auto edge_tag) {
  constexpr bool EDGE = decltype(edge_tag)::value;
  auto issue = [&](int logical, Row& q) {
    const int row = clampi(logical, sh - 1) * spitch;              // scalar
    if constexpr (EDGE) {
#pragma unroll
      for (int k = 0; k < NW; ++k)
        q.w[k] = __builtin_amdgcn_raw_buffer_load_b32(srsrc, (int)eoff[k], row, 0);
    } else {
      const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(srsrc, (int)goff, row, 0);
      const v2u32 u = __builtin_amdgcn_raw_buffer_load_b64(srsrc, (int)goff + 16, row, 0);
      q.w[0] = v.x; q.w[1] = v.y; q.w[2] = v.z; q.w[3] = v.w; q.w[4] = u.x; q.w[5] = u.y;
      asm volatile("" : "=v"(q.w[6]));                             // (not loaded, not read)
    }
  };
  Row pf[AHEAD];
#pragma unroll
  for (int j = 0; j < AHEAD; ++j) {
    issue(s_begin + j, pf[j]);
    __builtin_amdgcn_sched_barrier(0); // rows in ISSUE order: vmcnt retires in order
  }
  v2f32 ring[TAPS][3]; // filtered rows of the vertical window: [slot] = (R, G) of pixel 0, (R, G) of pixel 1, (B of 0, B of 1)
#pragma unroll
  for (int j = 0; j < TAPS; ++j)
    ring[j][0] = ring[j][1] = ring[j][2] = (v2f32){0.0f, 0.0f};
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dp, (short)0, (int)0xffffffffu, 0x00020000);
  const int ooff = xb * 3;                          // the lane's byte offset in a dst row
  int orow = y_first * dpitch;                      // the dst row's (scalar)
  const int quad_lane = lane & 3;
  const u32 quad_sel6 = 0x03020100u + 0x02020202u * (u32)quad_lane;
  const float* wt = wtab + 4;                       // weights of the next dst row

#pragma unroll 1
  for (int s0 = s_begin; s0 <= s_end; s0 += TAPS) {
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      const int cur = s0 + j;
      const bool live = cur <= s_end; // wave-uniform; rows past the end skip the work, never the load
      u32 w[NW];
#pragma unroll
      for (int k = 0; k < NW; ++k)
        w[k] = pf[j % AHEAD].w[k];
      if constexpr (EDGE) { // one pixel per register -> the 24 packed bytes (v_perm_b32 picks bytes by name: what lies beside a pixel is ignored)
        u32 px[NW];
#pragma unroll
        for (int k = 0; k < NW; ++k)
          px[k] = __builtin_amdgcn_alignbyte(0u, w[k], (eshift >> (2 * k)) & 3u); // (the selector masked: only gfx9 ignores the bits above 1)
        w[0] = __builtin_amdgcn_perm(px[1], px[0], 0x04020100u);
        w[1] = __builtin_amdgcn_perm(px[2], px[1], 0x05040201u);
        w[2] = __builtin_amdgcn_perm(px[3], px[2], 0x06050402u);
        w[3] = __builtin_amdgcn_perm(px[5], px[4], 0x04020100u);
        w[4] = __builtin_amdgcn_perm(px[6], px[5], 0x05040201u);
        w[5] = px[6] >> 16;
      }
#pragma unroll
      for (int k = 0; k < 6; ++k)
        asm volatile("" : "+v"(w[k])); // (the row's registers are read before the load below takes them)
      __builtin_amdgcn_sched_barrier(0);
      issue(cur + AHEAD, pf[j % AHEAD]);
      __builtin_amdgcn_sched_barrier(0);
      if (!live)
        continue;
      // ---- the pass along the row, from registers: byte 3 k + c of the 24 is channel c of window pixel k
      v2f32 hrg[2];
      float hb[2];
      {
        v2f32 crg[NW];
        float cb[NW];
        rgb_window_px<0>(w, crg[0], cb[0]); rgb_window_px<1>(w, crg[1], cb[1]); rgb_window_px<2>(w, crg[2], cb[2]);
        rgb_window_px<3>(w, crg[3], cb[3]); rgb_window_px<4>(w, crg[4], cb[4]); rgb_window_px<5>(w, crg[5], cb[5]);
        rgb_window_px<6>(w, crg[6], cb[6]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v2f32 e = (v2f32){0.0f, 0.0f}, o = (v2f32){0.0f, 0.0f};
          rwr_pk_fma<0>(e, W[q][0], crg[0]);
          rwr_pk_fma<1>(o, W[q][0], crg[1]);
          rwr_pk_fma<0>(e, W[q][1], crg[2]);
          rwr_pk_fma<1>(o, W[q][1], crg[3]);
          rwr_pk_fma<0>(e, W[q][2], crg[4]);
          rwr_pk_fma<1>(o, W[q][2], crg[5]);
          float eb = W[q][0].x * cb[0], ob = W[q][0].y * cb[1];
          eb = __builtin_fmaf(W[q][1].x, cb[2], eb);
          ob = __builtin_fmaf(W[q][1].y, cb[3], ob);
          eb = __builtin_fmaf(W[q][2].x, cb[4], eb);
          ob = __builtin_fmaf(W[q][2].y, cb[5], ob);
          if (q > 0) {
            rwr_pk_fma<0>(e, W[q][3], crg[6]);
            eb = __builtin_fmaf(W[q][3].x, cb[6], eb);
          }
          hrg[q] = e + o;
          hb[q] = eb + ob;
        }
      }
      asm volatile("" : "+v"(hrg[0]), "+v"(hrg[1]), "+v"(hb[0]), "+v"(hb[1]));
      ring[j][0] = hrg[0];
      ring[j][1] = hrg[1];
      ring[j][2] = (v2f32){hb[0], hb[1]};
      // ---- every dst row whose window ends with this source row; slot j is the newest row, logical row r of the window
      // sits in slot (j + 1 + r) mod TAPS
      auto emit = [&]() {
        v2f32 wp[3];
        const float4 qa = *reinterpret_cast<const float4*>(wt);
        const float2 qb = *reinterpret_cast<const float2*>(wt + 4);
        wp[0] = (v2f32){qa.x, qa.y};
        wp[1] = (v2f32){qa.z, qa.w};
        wp[2] = (v2f32){qb.x, qb.y};
        wt += WT;
        v2f32 v[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(v[g]) : "v"(wp[0]), "v"(ring[(j + 1) % TAPS][g]));
          rwr_pk_fma<1>(v[g], wp[0], ring[(j + 2) % TAPS][g]);
          rwr_pk_fma<0>(v[g], wp[1], ring[(j + 3) % TAPS][g]);
          rwr_pk_fma<1>(v[g], wp[1], ring[(j + 4) % TAPS][g]);
          rwr_pk_fma<0>(v[g], wp[2], ring[(j + 5) % TAPS][g]);
          rwr_pk_fma<1>(v[g], wp[2], ring[(j + 6) % TAPS][g]);
        }
        u32 w0 = 0, w1 = 0; // r0 g0 b0 r1 | g1 b1
        w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].x, 0u, w0);
        w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].y, 1u, w0);
        w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].x, 2u, w0);
        w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].x, 3u, w0);
        w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].y, 0u, w1);
        w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].y, 1u, w1);
        if (full) { // a quad's 24 bytes as three aligned 8-byte stores (rows23_tile's store_row)
          const u32 n0 = (u32)__builtin_amdgcn_update_dpp(0, (int)w0, 0xf9, 0xf, 0xf, true);   // quad_perm:[1,2,3,3]: the next lane's
          const u32 n1 = (u32)__builtin_amdgcn_update_dpp(0, (int)w1, 0xf9, 0xf, 0xf, true);
          const u32 z1 = __builtin_amdgcn_perm(n0, w1, 0x05040100u);
          const u32 z2 = __builtin_amdgcn_alignbyte(n1, n0, 2);
          const u32 o0 = __builtin_amdgcn_perm(z1, w0, quad_sel6), o1 = __builtin_amdgcn_perm(z2, z1, quad_sel6);
          if (quad_lane != 3)
            __builtin_amdgcn_raw_buffer_store_b64((v2u32){o0, o1}, drsrc, ooff + 2 * quad_lane, orow, kStoreNt);
        } else if (n_px == 2) {
          __builtin_amdgcn_raw_buffer_store_b32(w0, drsrc, ooff, orow, 0);
          __builtin_amdgcn_raw_buffer_store_b16((short)w1, drsrc, ooff + 4, orow, 0);
        } else if (n_px == 1) {
          __builtin_amdgcn_raw_buffer_store_b16((short)w0, drsrc, ooff, orow, 0);
          __builtin_amdgcn_raw_buffer_store_b8((char)(w0 >> 16), drsrc, ooff + 2, orow, 0);
        }
        orow += dpitch;
      };
      const int c = __builtin_amdgcn_readlane(cntv, cur - s_begin);
      if (c > 0) {
        emit();
        if (c > 1) {
          emit();
          for (int k = 2; k < c; ++k)
            emit();
        }
      }
    }
  }
  };
  if (edge_tile)
    walk(std::true_type{});
  else
    walk(std::false_type{});
}

template <int ROWS>
__global__ void __launch_bounds__(kBlock, 4) k_resize_rows_rgb(const ResizeArgs a) {
  extern __shared__ uint4 rows_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  rows_rgb_tile<ROWS>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, reinterpret_cast<float*>(rows_lds), a.lds_per_wave / 4);
}

bool resize_rows_rgb_fits(const ResizeJob& j, int elem, int taps, int src_w, int dst_w) {
  int d2, d3;
  return elem == 1 && taps == 6 && j.channels == 3 && j.ssub_x == 0 && j.sub_x == 0 && tuning(VALI_TUNE_RESIZE_ROWS) != 0 &&
         tuning(VALI_TUNE_RESIZE_ROWS) != 3 && tuning(VALI_TUNE_RESIZE_FORCE_GATHER) != 1 && rows_reg_fits(src_w, dst_w, 3, d2, d3);
}

int launch_resize_rows_rgb(const ResizeArgs& base, int src_h, int dst_w, int dst_h, int n, hipStream_t stream) {
  const int force = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE); // 1 / 2 / 3: 8- / 2- / 32-row waves whatever the launch size
  ResizeArgs a = base;
  auto count = [&](int rows, bool assign) {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const u32 tiles_x = (u32)(dst_w + kRgbTilePx - 1) / kRgbTilePx;
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (u32)((dst_h + kWavesPerBlock * rows - 1) / (kWavesPerBlock * rows));
    }
    return total;
  };
  // rows per wave as launch_resize_rows chooses them (long waves when the launch stays full and the wave's source rows fit
  // its 64-entry completion table)
  const bool fits64 = 64.0 * (double)src_h / (double)dst_h * (1.0 + 1e-6) + 6 + 2 <= 64.0;
  const unsigned long long t64 = (unsigned long long)count(64, false) * (unsigned)n, t32 = (unsigned long long)count(32, false) * (unsigned)n,
                           t8 = (unsigned long long)count(8, false) * (unsigned)n;
  const int rows = force == 1 ? 8 : force == 2 ? 2 : force == 3 ? 32 : (fits64 && (force == 4 || t64 >= 1024ull)) ? 64 : t32 >= 1024ull ? 32 : t8 >= 320ull ? 8 : 2;
  a.map = make_tile_map_linear(count(rows, true), (u32)n);
  a.lds_per_wave = (kRrShare + rows * (4 + 2 * 6) + 64) * 4;
  const unsigned lds = (unsigned)a.lds_per_wave * kWavesPerBlock;
  const dim3 grid = tile_grid(a.map);
  if (rows == 64) hipLaunchKernelGGL((k_resize_rows_rgb<64>), grid, dim3(kBlock), lds, stream, a);
  else if (rows == 32) hipLaunchKernelGGL((k_resize_rows_rgb<32>), grid, dim3(kBlock), lds, stream, a);
  else if (rows == 8) hipLaunchKernelGGL((k_resize_rows_rgb<8>), grid, dim3(kBlock), lds, stream, a);
  else hipLaunchKernelGGL((k_resize_rows_rgb<2>), grid, dim3(kBlock), lds, stream, a);
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

// =====================================================================================================================
// Planes enlarged by exactly 3:2 on BOTH axes (720p -> 1080p, 480p -> 720p, 1440p -> 2160p and their chroma planes): the
// everyday upscale.  scale = fl(2/3), and x * scale lands where the pattern says for every x (3m * fl(2/3) rounds to
// exactly 2m: the relative error 2^-25 of fl(2/3) is below half an ulp of any integer; the other two residues sit a third
// away from an integer), so per dst triple (3m, 3m+1, 3m+2):
//   3m   : i = 2m,     a = 0    -> weights {0,0,1,0,0,0}: the sample IS source pixel 2m, bit for bit (0 * t = +0 for
//                                  the unsigned element types this kernel takes, and t + 0 = t)
//   3m+1 : i = 2m,     a ~ 2/3
//   3m+2 : i = 2m + 1, a ~ 1/3     (the weights of these two differ from x to x in their last bits: per-lane registers)
// The tap POSITIONS are static, so nothing is gathered: a lane owns 4 source pixels (2 of a two-channel plane) = 6 dst
// pixels (3), converts them once, gets the 2 pixels before and the 3 after from its neighbours by DPP (`wave_shr:1` /
// `wave_shl:1`; lane 0 and lane 63 keep the DPP's `old` operand, pre-loaded from one halo load per row) and filters from
// registers -- no LDS stage, no tap reads.  Samples whose first tap sits at an ODD position run over the same aligned
// register pairs with their weights moved one slot: (0,w0) (w1,w2) (w3,w4) (w5,0) -- the low halves then accumulate the
// odd taps and the high halves the even ones, each in the specification's order (a chain that starts with
// fma(w, t, +0) equals the one that starts with w * t for t >= 0; the trailing fma(0, t, e) is e).  Down the rows the same
// pattern: dst row 3M is the quantised filtered row 2M itself, rows 3M+1 / 3M+2 are 6-tap combinations; the walk over the
// source rows is static (odd source rows complete two dst rows, even ones one).
// Per source row and lane: 34 vector instructions along the row + 37 down the rows for 9 output samples -- against 113
// wave instructions per 6 in the general kernel above.
template <typename T, int ES> struct R23 {
  static constexpr int kPx = ES == 1 ? 4 : 2;                    // source pixels per lane
  static constexpr int kWords = (kPx * ES * (int)sizeof(T) + 3) / 4; // dwords of the lane's group: 1 (u8), 2 (u16; packed RGB: 6 bytes)
  static constexpr int kHaloWords = 4 * ES * (int)sizeof(T) / 4; // the halo group is always 4 pixels
};
template <int WORDS> struct R23Load {
  u32 w[WORDS];
};
template <int WORDS> __device__ __forceinline__ R23Load<WORDS> r23_load(const uint8_t* p) {
  R23Load<WORDS> g;
  if constexpr (WORDS == 1) {
    g.w[0] = gload_u<u32>(p);
  } else if constexpr (WORDS == 2) {
    const v2u32 q = gload_u<v2u32>(p);
    g.w[0] = q.x; g.w[1] = q.y;
  } else {
    const v4u32 q = gload_u<v4u32>(p);
    g.w[0] = q.x; g.w[1] = q.y; g.w[2] = q.z; g.w[3] = q.w;
  }
  return g;
}
// the same through a raw buffer descriptor over the plane: the row's offset is the SCALAR operand, the lane's offset the one
// VGPR of the address (resize_cols.hip cols_walk: no 64-bit vector add per row and group)
template <int WORDS> __device__ __forceinline__ R23Load<WORDS> r23_load_b(__amdgpu_buffer_rsrc_t rsrc, u32 lane_off, u32 row_off) {
  R23Load<WORDS> g;
  if constexpr (WORDS == 1) {
    g.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off, (int)row_off, 0);
  } else if constexpr (WORDS == 2) {
    const v2u32 q = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_off, (int)row_off, 0);
    g.w[0] = q.x; g.w[1] = q.y;
  } else if constexpr (WORDS == 3) {
    typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
    const v3u32 q = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)lane_off, (int)row_off, 0);
    g.w[0] = q.x; g.w[1] = q.y; g.w[2] = q.z;
  } else {
    const v4u32 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)row_off, 0);
    g.w[0] = q.x; g.w[1] = q.y; g.w[2] = q.z; g.w[3] = q.w;
  }
  return g;
}
// two packed 8-bit RGB pixels: 6 bytes that start at any even address and end where the row may end -- 4 + 2
__device__ __forceinline__ R23Load<2> r23_load_b6(__amdgpu_buffer_rsrc_t rsrc, u32 lane_off, u32 row_off) {
  R23Load<2> g;
  g.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off, (int)row_off, 0);
  g.w[1] = (u32)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, (int)lane_off + 4, (int)row_off, 0);
  return g;
}
template <typename T, int N, int WORDS> __device__ __forceinline__ float r23_elem(const R23Load<WORDS>& g) {
  if constexpr (sizeof(T) == 1)
    return ubyte_f32<N % 4>(g.w[N / 4]);
  else
    return (float)((N % 2) ? (g.w[N / 2] >> 16) : (g.w[N / 2] & 0xffffu));
}
__device__ __forceinline__ float r23_shr1(float old, float v) { // lane l gets lane l - 1's v; lane 0 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float r23_shl1(float old, float v) { // lane l gets lane l + 1's v; lane 63 keeps `old`
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

__device__ __forceinline__ u32 r23_shr1(u32 old, u32 v) { // the same on packed bytes
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ u32 r23_shl1(u32 old, u32 v) {
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xf, 0xf, false);
}
// byte N of the 6 bytes (lo: 4, hi: 2) of two packed RGB pixels
template <int N> __device__ __forceinline__ float r23_byte6(u32 lo, u32 hi) {
  if constexpr (N < 4)
    return ubyte_f32<N>(lo);
  else
    return ubyte_f32<N - 4>(hi);
}
// the two filtered samples of a dst triple from the pixel pairs around them (two-channel and packed RGB planes): d1 starts at
// an even tap position, d2 at an odd one (6 taps; the other way round with 4)
template <int TAPS>
__device__ __forceinline__ void r23_pair_chains(const v2f32 (&wa)[TAPS / 2], const v2f32 (&wb)[TAPS / 2 + 1], v2f32 p0, v2f32 p1, v2f32 p2,
                                                v2f32 p3, float& d1, float& d2) {
  if constexpr (TAPS == 6) {
    v2f32 a1 = wa[0] * p0, a2 = wb[0] * p0;
    a1 = __builtin_elementwise_fma(wa[1], p1, a1); a2 = __builtin_elementwise_fma(wb[1], p1, a2);
    a1 = __builtin_elementwise_fma(wa[2], p2, a1); a2 = __builtin_elementwise_fma(wb[2], p2, a2);
    a2 = __builtin_elementwise_fma(wb[3], p3, a2);
    d1 = a1.x + a1.y;
    d2 = a2.y + a2.x;
  } else {
    v2f32 a1 = wb[0] * p0, a2 = wa[0] * p1;
    a1 = __builtin_elementwise_fma(wb[1], p1, a1); a2 = __builtin_elementwise_fma(wa[1], p2, a2);
    a1 = __builtin_elementwise_fma(wb[2], p2, a1);
    d1 = a1.y + a1.x;
    d2 = a2.x + a2.y;
    (void)p3;
  }
  v2f32 d = {d1, d2}; // (the sums land in the register pair the ring keeps them in)
  asm volatile("" : "+v"(d));
  d1 = d.x; d2 = d.y;
}

template <typename T, int ES, int TAPS, int RW>
__device__ __forceinline__ void rows23_tile(const uint8_t* sp, int spitch, int sw, int sh, uint8_t* dp, int dpitch,
                                            int dw, int dh, u32 tx, u32 ty, float* wg_lds, int wave_floats) {
  static_assert(RW % 3 == 0 && RW <= kWave, "whole dst triples per wave, one row tap set per lane");
  constexpr int EB = (int)sizeof(T);
  constexpr int WT = 4 + 2 * TAPS;
  constexpr int PX = R23<T, ES>::kPx, NW = R23<T, ES>::kWords, HW = R23<T, ES>::kHaloWords;
  constexpr int NF = ES == 1 ? 4 : 2;   // filtered (non-copy) samples per lane whose weights are kept: d1 d2 (d4 d5)
  constexpr int DE = ES == 1 ? 6 : 3 * ES; // dst ELEMENTS per lane (6 pixels, or 3 pixels x 2 / 3 channels)
  constexpr int NP = DE / 2;            // ... as register pairs
  constexpr bool TAIL = (DE & 1) != 0;  // ... and a single register (packed RGB: 9 elements)
  static_assert(ES < 3 || EB == 1, "three channels: 8-bit packed RGB only");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int X0 = (int)tx * (kWave * PX);               // first source pixel of the tile
  const int ps = X0 + PX * lane;                       // this lane's first source pixel
  const int pd = ps / 2 * 3;                           // ... and first dst pixel
  const int y_first = (int)(ty * kWavesPerBlock + wave) * RW;
  const float scale_x = (float)sw / (float)dw, scale_y = (float)sh / (float)dh;
  float* const lds = wg_lds + wave * wave_floats;

  // ---- column tap sets of the filtered samples: wave w evaluates sample w of every lane (ES = 2: waves 0 / 1), shared
  // through LDS.  Sample f of the lane is dst pixel pd + {1, 2, 4, 5}[f].
  v2f32 wa[ES == 1 ? 2 : 1][TAPS / 2];       // first tap at an even position: (w0,w1) (w2,w3) (w4,w5)
  v2f32 wb[ES == 1 ? 2 : 1][TAPS / 2 + 1];   // ... at an odd one: (0,w0) (w1,w2) (w3,w4) (w5,0)
  {
    const int f = ES == 1 ? wave : (wave & 1);
    const int x = min(pd + f + 1 + (f >> 1), dw - 1);
    const LzTap<TAPS> c = make_lz_tap<TAPS>(x, scale_x);
    float* mine = lds + 8 * lane;
    *reinterpret_cast<float4*>(mine) = make_float4(c.w[0], c.w[1], c.w[2], c.w[3]);
    if constexpr (TAPS == 6)
      *reinterpret_cast<float2*>(mine + 4) = make_float2(c.w[4], c.w[5]);
    __syncthreads();
#pragma unroll
    for (int f2 = 0; f2 < NF; ++f2) {
      const float* from = wg_lds + f2 * wave_floats + 8 * lane;
      const float4 a = *reinterpret_cast<const float4*>(from);
      float w[6] = {a.x, a.y, a.z, a.w, 0.0f, 0.0f};
      if constexpr (TAPS == 6) {
        const float2 b = *reinterpret_cast<const float2*>(from + 4);
        w[4] = b.x; w[5] = b.y;
      }
      // (the 4-tap window starts one pixel before i: there the samples with an EVEN i start at an odd position)
      if (((f2 & 1) == 0) == (TAPS == 6)) { // first tap at an even position: d1, d4 (6 taps) / d2, d5 (4 taps)
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k)
          wa[f2 >> 1][k] = (v2f32){w[2 * k], w[2 * k + 1]};
      } else {
        wb[f2 >> 1][0] = (v2f32){0.0f, w[0]};
#pragma unroll
        for (int k = 1; k < TAPS / 2; ++k)
          wb[f2 >> 1][k] = (v2f32){w[2 * k - 1], w[2 * k]};
        wb[f2 >> 1][TAPS / 2] = (v2f32){w[TAPS - 1], 0.0f};
      }
    }
    __syncthreads();
  }
  if (y_first >= dh)
    return;

  // ---- row taps of the wave's dst rows (pre-splatted pairs, broadcast reads in the vertical pass)
  float* const wtab = lds;
  {
    const LzTap<TAPS> vy = make_lz_tap<TAPS>(min(y_first + lane, dh - 1), scale_y);
    if (lane < RW) {
      float* row = wtab + lane * WT;
      if constexpr (ES == 3) { // as they are: the column pass picks the half it needs (op_sel), 3 registers instead of 6 pairs
        *reinterpret_cast<float4*>(row + 4) = make_float4(vy.w[0], vy.w[1], vy.w[2], vy.w[3]);
        if constexpr (TAPS == 6)
          *reinterpret_cast<float2*>(row + 8) = make_float2(vy.w[4], vy.w[5]);
      } else {
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k)
          *reinterpret_cast<float4*>(row + 4 + 4 * k) = make_float4(vy.w[2 * k], vy.w[2 * k], vy.w[2 * k + 1], vy.w[2 * k + 1]);
      }
    }
  }
  wave_lds_sync();
  const int last_rr = min(RW, dh - y_first) - 1;
  const int s_begin = y_first / 3 * 2 - 2;             // even; source row t of the walk is s_begin + t

  // ---- addresses.  A lane's group must lie inside its row: groups that would start past sw - PX start there instead and
  // are put right in registers (tiles at the right edge only).  Halo: lane 0 loads the 4 pixels before the tile, lane 63 the
  // 4 after it, every other lane its own group again (the same lines: one more request, no more traffic).
  // (wave-uniform flags as scalar integers: branches on them are s_cmp + s_cbranch, not mask arithmetic)
  const int first_tile = __builtin_amdgcn_readfirstlane(tx == 0 ? 1 : 0);
  const int edge = __builtin_amdgcn_readfirstlane(X0 + kWave * PX + 4 > sw ? 1 : 0); // some lane (or lane 63's halo) reaches past the row
  const int gs = min(ps, sw - PX);                     // where the lane's group really starts
  const bool in_row = ps + PX <= sw, straddle = !in_row && ps < sw; // (straddle: ES = 1 and sw % 4 == 2 only)
  const int hp = lane == 0 ? max(X0 - 4, 0) : lane == 63 ? ps + PX : ps;
  const int hs = min(hp, sw - 4);
  const bool h_in = hp + 4 <= sw, h_straddle = !h_in && hp < sw;     // (the halo group is 4 pixels: may straddle by 2)
  const u32 goff = (u32)(gs * ES * EB), hoff = (u32)(hs * ES * EB);

  // rows in flight: TAPS, or 3 for packed RGB under Lanczos (5 registers per row; its ring of filtered rows takes 54)
  constexpr int AHEAD = ES == 3 && TAPS == 6 ? 3 : TAPS;
  static_assert(TAPS % AHEAD == 0, "slot j % AHEAD of the unrolled walk must be row t % AHEAD");
  R23Load<NW> pf[AHEAD];
  R23Load<HW> hf[AHEAD];
  const bool halo_lane = lane == 0 || lane == 63;
  const __amdgpu_buffer_rsrc_t srsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sp), (short)0, (int)0xffffffffu, 0x00020000);
  auto issue = [&](int t, R23Load<NW>& q, R23Load<HW>& h) {
    const u32 row = (u32)(clampi(s_begin + t, sh - 1) * spitch);   // scalar
    if constexpr (ES == 3) {
      q = r23_load_b6(srsrc, goff, row);
      // 12 bytes for 2 lanes of 64: the other lanes' registers hold whatever they held (nobody reads them; the empty asm
      // "defines" them so that nothing is kept alive or copied for their sake)
      asm volatile("" : "=v"(h.w[0]), "=v"(h.w[1]), "=v"(h.w[2]));
      if (halo_lane)
        h = r23_load_b<HW>(srsrc, hoff, row);
    } else {
      q = r23_load_b<NW>(srsrc, goff, row);
      h = r23_load_b<HW>(srsrc, hoff, row);
    }
  };
#pragma unroll
  for (int j = 0; j < AHEAD; ++j) {
    issue(j, pf[j], hf[j]);
    __builtin_amdgcn_sched_barrier(0);
  }

  // filtered rows: [slot][pair]; ES = 1: (d0,d1) (d2,d3) (d4,d5), ES = 2: (u,v) of the lane's 3 pixels, ES = 3: the 9 elements
  // r0 g0 b0 r1 ... b2 of its 3 pixels two by two
  v2f32 ring[TAPS][NP];
  float rtail[TAPS];
#pragma unroll
  for (int j = 0; j < TAPS; ++j) {
#pragma unroll
    for (int g = 0; g < NP; ++g)
      ring[j][g] = (v2f32){0.0f, 0.0f};
    rtail[j] = 0.0f;
  }

  // ---- output: 6 elements per lane, lanes contiguous
  const int nel = min(DE, max(0, (dw - pd) * ES)); // DE, 3 (ES = 1) or 0
  const int full = __builtin_amdgcn_readfirstlane((X0 + kWave * PX) <= sw ? 1 : 0); // wave-uniform: every lane has its 6
  uint8_t* optr = dp + (size_t)y_first * dpitch + (size_t)pd * ES * EB;
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(dp, (short)0, (int)0xffffffffu, 0x00020000);
  const int ooff = pd * ES * EB;                   // the lane's byte offset in a dst row
  int orow = y_first * dpitch;                     // the row's (scalar) -- the full tiles store through the descriptor
  const int quad_lane = lane & 3;                  // packed RGB: where the lane's dwords of its quad's 36 bytes start, and the
  const int quad_off = ooff - quad_lane;           // byte selector that assembles them (v_perm_b32 over {own dword, dword before})
  const u32 quad_sel = 0x07060504u - 0x01010101u * (u32)quad_lane;
  const u32 quad_sel6 = 0x03020100u + 0x02020202u * (u32)quad_lane; // 6 bytes per lane (8-bit planes of 1 / 2 channels): bytes 2q .. 2q + 3
  (void)quad_off; (void)quad_sel; (void)quad_sel6;
  const float* wt = wtab + 4;
  int rr = 0;
  constexpr int T0 = TAPS == 6 ? 5 : 4;            // first source row of the walk that completes a dst row

  auto store_row = [&](const v2f32 (&v)[NP], float vt) {
    (void)vt;
    if constexpr (ES == 3) {
      // 9 bytes per lane at 9 * lane: one 8-byte store from an address of any alignment + one byte
      u32 w0 = 0, w1 = 0, w2 = 0;
      // ring order: (d1, d2) of R, G, B, then the copied sample's R, G | B -- bytes: r0 g0 b0 r1 | g1 b1 r2 g2 | b2
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[3].x, 0u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[3].y, 1u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(vt, 2u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].x, 3u, w0);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].x, 0u, w1);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].x, 1u, w1);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].y, 2u, w1);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].y, 3u, w1);
      w2 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].y, 0u, w2);
      if (full) {
        // whole tiles: the 4 lanes of a quad hold 36 bytes = 9 dwords; lane q stores dwords 2q, 2q + 1 (and lane 3 the ninth):
        // they start q bytes before its own -- the tail of the lane before it, fetched within the quad
        const u32 tail = __builtin_amdgcn_alignbyte(w2, w1, 1);                                     // the lane's bytes 5 .. 8
        const u32 prev = (u32)__builtin_amdgcn_update_dpp(0, (int)tail, 0x90, 0xf, 0xf, true);      // quad_perm:[0,0,1,2]
        const u32 o0 = __builtin_amdgcn_perm(w0, prev, quad_sel), o1 = __builtin_amdgcn_perm(w1, w0, quad_sel);
        // ONE 12-byte store per lane: the third dword is the next lane's first (written twice with the same bits), lane 3's the
        // quad's ninth -- as 8 bytes + a masked dword every 36-byte piece left the chip in two partial passes (non-temporal
        // stores do not wait in the L2 to be merged: 8.7 % more bytes written than the destination holds)
        const u32 nxt = (u32)__builtin_amdgcn_update_dpp(0, (int)o0, 0xf9, 0xf, 0xf, true);         // quad_perm:[1,2,3,3]
        typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
        __builtin_amdgcn_raw_buffer_store_b96((v3u32){o0, o1, quad_lane == 3 ? tail : nxt}, drsrc, quad_off, orow, kStoreNt);
      } else if (nel == DE) {
        __builtin_amdgcn_raw_buffer_store_b64((v2u32){w0, w1}, drsrc, ooff, orow, 0);
        __builtin_amdgcn_raw_buffer_store_b8((char)w2, drsrc, ooff + 8, orow, 0);
      }
    } else if constexpr (EB == 1) {
      u32 w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].x, 0u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[0].y, 1u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].x, 2u, w0);
      w0 = __builtin_amdgcn_cvt_pk_u8_f32(v[1].y, 3u, w0);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].x, 0u, w1);
      w1 = __builtin_amdgcn_cvt_pk_u8_f32(v[2].y, 1u, w1);
      if (full) {
        // whole tiles: a quad's 24 bytes leave as 3 aligned 8-byte stores (lanes 0 .. 2: dwords 2q, 2q + 1 = the lane's bytes
        // from 2q on, then the next lane's) -- as 4 + 2 bytes per lane at stride 6 every 32-byte sector was written in two
        // partial passes (HBM traffic 1.15 x, a fifth of the kernel's time)
        const u32 n0 = (u32)__builtin_amdgcn_update_dpp(0, (int)w0, 0xf9, 0xf, 0xf, true);   // quad_perm:[1,2,3,3]: the next lane's
        const u32 n1 = (u32)__builtin_amdgcn_update_dpp(0, (int)w1, 0xf9, 0xf, 0xf, true);
        const u32 z1 = __builtin_amdgcn_perm(n0, w1, 0x05040100u);                            // own 4 5, next 0 1
        const u32 z2 = __builtin_amdgcn_alignbyte(n1, n0, 2);                                 // next 2 .. 5
        const u32 o0 = __builtin_amdgcn_perm(z1, w0, quad_sel6), o1 = __builtin_amdgcn_perm(z2, z1, quad_sel6);
        if (quad_lane != 3)
          __builtin_amdgcn_raw_buffer_store_b64((v2u32){o0, o1}, drsrc, ooff + 2 * quad_lane, orow, kStoreNt);
      } else if (nel == DE) {
        gstore_u<u32>(optr, w0);
        gstore_u<uint16_t>(optr + 4, (uint16_t)w1);
      } else if (nel == 3) {
        gstore_u<uint16_t>(optr, (uint16_t)w0);
        gstore<uint8_t>(optr + 2, (uint8_t)(w0 >> 16));
      }
    } else {
      const u32 w0 = finish_bits<T>(v[0].x) | (finish_bits<T>(v[0].y) << 16);
      const u32 w1 = finish_bits<T>(v[1].x) | (finish_bits<T>(v[1].y) << 16);
      const u32 w2 = finish_bits<T>(v[2].x) | (finish_bits<T>(v[2].y) << 16);
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 q = {w0, w1, w2};
      if (full) {
        __builtin_amdgcn_raw_buffer_store_b96(q, drsrc, ooff, orow, kStoreNt);
      } else if (nel == DE) {
        gstore_u<v3u32>(optr, q);
      } else if (nel == 3) {
        gstore_u<u32>(optr, w0);
        gstore_u<uint16_t>(optr + 4, (uint16_t)w1);
      }
    }
    if (!full)
      optr += dpitch;
    orow += dpitch;
    wt += WT;
    ++rr;
  };

  int t0 = 0;
#pragma unroll 1
  do {
#pragma unroll
    for (int j = 0; j < TAPS; ++j) {
      const int t = t0 + j;
      const bool live = rr <= last_rr; // wave-uniform
      if (live) {
        // ---- the lane's pixels and their neighbours, per channel: P[0..9] = L2 L3 f0 f1 f2 f3 R0 R1 R2 0 (ES = 1),
        // P[0..7] = Lp0 Lp1 f0 f1 Rp0 Rp1 RRp0 0 (ES = 2): aligned register pairs
        float el[ES == 1 ? 2 : DE]; // ES >= 2: the 3 pixels' elements, channel-interleaved as they are stored
        auto channel = [&](auto ctag, v2f32& o0, v2f32& o1, v2f32& o2) {
          constexpr int C = decltype(ctag)::value;
          float f[PX], g[4];
#pragma unroll
          for (int k = 0; k < PX; ++k)
            f[k] = 0.0f;
          f[0] = r23_elem<T, 0 * ES + C, NW>(pf[j % AHEAD]);
          f[1] = r23_elem<T, 1 * ES + C, NW>(pf[j % AHEAD]);
          if constexpr (PX == 4) {
            f[2] = r23_elem<T, 2 * ES + C, NW>(pf[j % AHEAD]);
            f[3] = r23_elem<T, 3 * ES + C, NW>(pf[j % AHEAD]);
          }
          g[0] = r23_elem<T, 0 * ES + C, HW>(hf[j % AHEAD]);
          g[1] = r23_elem<T, 1 * ES + C, HW>(hf[j % AHEAD]);
          g[2] = r23_elem<T, 2 * ES + C, HW>(hf[j % AHEAD]);
          g[3] = r23_elem<T, 3 * ES + C, HW>(hf[j % AHEAD]);
          if (edge) { // right-most tiles: groups that were moved left to stay inside the row, pixels past the row's end
            if constexpr (PX == 4) {
              f[0] = in_row ? f[0] : straddle ? f[2] : f[3];
              f[1] = in_row ? f[1] : f[3];
              f[2] = in_row ? f[2] : f[3];
            } else {
              f[0] = in_row ? f[0] : f[1];
            }
            g[0] = h_in ? g[0] : h_straddle ? g[2] : g[3];
            g[1] = h_in ? g[1] : g[3];
            g[2] = h_in ? g[2] : g[3];
          }
          if (first_tile) { // pixels -1, -2 replicate pixel 0
            g[2] = lane == 0 ? f[0] : g[2];
            g[3] = lane == 0 ? f[0] : g[3];
          }
          if constexpr (ES == 1) {
            const float L2 = r23_shr1(g[2], f[2]), L3 = r23_shr1(g[3], f[3]);
            const float R0 = r23_shl1(g[0], f[0]), R1 = r23_shl1(g[1], f[1]), R2 = r23_shl1(g[2], f[2]);
            const v2f32 p0 = {L2, L3}, p1 = {f[0], f[1]}, p2 = {f[2], f[3]}, p3 = {R0, R1}, p4 = {R2, 0.0f};
            float d1, d2, d4, d5;
            if constexpr (TAPS == 6) {
              // d1: i = 4l      taps L2 L3 f0 f1 f2 f3        d4: i = 4l + 2  taps f0 f1 f2 f3 R0 R1
              v2f32 a1 = wa[0][0] * p0, a4 = wa[1][0] * p1;
              a1 = __builtin_elementwise_fma(wa[0][1], p1, a1); a4 = __builtin_elementwise_fma(wa[1][1], p2, a4);
              a1 = __builtin_elementwise_fma(wa[0][2], p2, a1); a4 = __builtin_elementwise_fma(wa[1][2], p3, a4);
              // d2: i = 4l + 1  taps L3 f0 f1 f2 f3 R0        d5: i = 4l + 3  taps f1 f2 f3 R0 R1 R2
              v2f32 a2 = wb[0][0] * p0, a5 = wb[1][0] * p1;
              a2 = __builtin_elementwise_fma(wb[0][1], p1, a2); a5 = __builtin_elementwise_fma(wb[1][1], p2, a5);
              a2 = __builtin_elementwise_fma(wb[0][2], p2, a2); a5 = __builtin_elementwise_fma(wb[1][2], p3, a5);
              a2 = __builtin_elementwise_fma(wb[0][3], p3, a2); a5 = __builtin_elementwise_fma(wb[1][3], p4, a5);
              d1 = a1.x + a1.y; d4 = a4.x + a4.y;
              d2 = a2.y + a2.x; d5 = a5.y + a5.x; // (high halves hold the even taps: e + o)
            } else {
              // 4 taps i-1 .. i+2.  d1: L3 f0 f1 f2 (first tap odd)  d2: f0 f1 f2 f3 (even)  d4: f1 f2 f3 R0 (odd)  d5: f2 f3 R0 R1 (even)
              v2f32 a1 = wb[0][0] * p0, a4 = wb[1][0] * p1;
              a1 = __builtin_elementwise_fma(wb[0][1], p1, a1); a4 = __builtin_elementwise_fma(wb[1][1], p2, a4);
              a1 = __builtin_elementwise_fma(wb[0][2], p2, a1); a4 = __builtin_elementwise_fma(wb[1][2], p3, a4);
              v2f32 a2 = wa[0][0] * p1, a5 = wa[1][0] * p2;
              a2 = __builtin_elementwise_fma(wa[0][1], p2, a2); a5 = __builtin_elementwise_fma(wa[1][1], p3, a5);
              d1 = a1.y + a1.x; d4 = a4.y + a4.x;
              d2 = a2.x + a2.y; d5 = a5.x + a5.y;
              (void)p4;
            }
            asm volatile("" : "+v"(d1), "+v"(d2), "+v"(d4), "+v"(d5));
            o0 = (v2f32){f[0], d1};
            o1 = (v2f32){d2, f[2]};
            o2 = (v2f32){d4, d5};
          } else {
            const float L0 = r23_shr1(g[2], f[0]), L1 = r23_shr1(g[3], f[1]);
            const float R0 = r23_shl1(g[0], f[0]), R1 = r23_shl1(g[1], f[1]);
            const float RR0 = r23_shl1(g[2], R0);
            const v2f32 p0 = {L0, L1}, p1 = {f[0], f[1]}, p2 = {R0, R1}, p3 = {RR0, 0.0f};
            float d1, d2;
            if constexpr (TAPS == 6) {
              // d1: i = 2l taps L0 L1 f0 f1 R0 R1 ; d2: i = 2l + 1 taps L1 f0 f1 R0 R1 RR0
              v2f32 a1 = wa[0][0] * p0, a2 = wb[0][0] * p0;
              a1 = __builtin_elementwise_fma(wa[0][1], p1, a1); a2 = __builtin_elementwise_fma(wb[0][1], p1, a2);
              a1 = __builtin_elementwise_fma(wa[0][2], p2, a1); a2 = __builtin_elementwise_fma(wb[0][2], p2, a2);
              a2 = __builtin_elementwise_fma(wb[0][3], p3, a2);
              d1 = a1.x + a1.y;
              d2 = a2.y + a2.x;
            } else {
              // d1: i = 2l taps L1 f0 f1 R0 (odd) ; d2: i = 2l + 1 taps f0 f1 R0 R1 (even)
              v2f32 a1 = wb[0][0] * p0, a2 = wa[0][0] * p1;
              a1 = __builtin_elementwise_fma(wb[0][1], p1, a1); a2 = __builtin_elementwise_fma(wa[0][1], p2, a2);
              a1 = __builtin_elementwise_fma(wb[0][2], p2, a1);
              d1 = a1.y + a1.x;
              d2 = a2.x + a2.y;
              (void)p3;
            }
            asm volatile("" : "+v"(d1), "+v"(d2));
            // pixels 3l, 3l+1, 3l+2 of this channel
            el[0 * ES + C] = f[0]; el[1 * ES + C] = d1; el[2 * ES + C] = d2;
            (void)o0; (void)o1; (void)o2;
          }
        };
        if constexpr (ES == 3) {
          // packed RGB: the neighbours travel as PACKED bytes (5 lane shifts per row, not 5 per channel) and every float is
          // converted where it is used.  q: the lane's 2 pixels (6 bytes), h: the halo group's 4 (12 bytes; lane 0: the 4
          // before the tile, lane 63: the 4 after its own)
          u32 q0 = pf[j % AHEAD].w[0], q1 = pf[j % AHEAD].w[1];
          u32 h0 = hf[j % AHEAD].w[0], h1 = hf[j % AHEAD].w[1], h2 = hf[j % AHEAD].w[2];
          if (edge) { // right-most tiles: groups moved left to stay inside the row hold its last pixel(s) where theirs would be
            const u32 last = (q0 >> 24) | (q1 << 8);                              // the group's second pixel
            q0 = in_row ? q0 : (last | (last << 24));
            q1 = in_row ? q1 : (last >> 8);
            const u32 g2 = __builtin_amdgcn_alignbyte(h2, h1, 2) & 0xffffffu, g3 = h2 >> 8;
            const u32 pa = h_straddle ? g2 : g3;                                  // halo pixels (pa, g3, g3, g3)
            h0 = h_in ? h0 : (pa | (g3 << 24));
            h1 = h_in ? h1 : ((g3 >> 8) | (g3 << 16));
            h2 = h_in ? h2 : ((g3 >> 16) | (g3 << 8));
          }
          u32 la = __builtin_amdgcn_alignbyte(h2, h1, 2), lb = h2 >> 16;         // halo pixels 2 and 3 laid out like (q0, q1)
          if (first_tile) { // pixels -1, -2 replicate pixel 0
            asm volatile(""); // (a branch on the scalar flag: as selects this costs every tile 5 instructions per row)
            const u32 px0 = q0 & 0xffffffu;
            la = lane == 0 ? (px0 | (px0 << 24)) : la;
            lb = lane == 0 ? (px0 >> 8) : lb;
          }
          const u32 l0 = r23_shr1(la, q0), l1 = r23_shr1(lb, q1);                // the 2 pixels before the lane's
          const u32 r0 = r23_shl1(h0, q0), r1 = r23_shl1(h1, q1);                // the 2 after them
          const u32 rr = r23_shl1(la, r0);                                        // the third (lane 63: halo pixel 2)
          auto rgb = [&](auto ctag) {
            constexpr int C = decltype(ctag)::value;
            const v2f32 p0 = {r23_byte6<C>(l0, l1), r23_byte6<C + 3>(l0, l1)};
            const v2f32 p1 = {r23_byte6<C>(q0, q1), r23_byte6<C + 3>(q0, q1)};
            const v2f32 p2 = {r23_byte6<C>(r0, r1), r23_byte6<C + 3>(r0, r1)};
            const v2f32 p3 = {ubyte_f32<C>(rr), 0.0f};
            float d1, d2;
            r23_pair_chains<TAPS>(wa[0], wb[0], p0, p1, p2, p3, d1, d2);
            // (the ring keeps a channel's two filtered samples as one pair -- the chains' sums land there -- and the three
            // copied samples behind them; store_row picks its bytes by name)
            el[2 * C] = d1; el[2 * C + 1] = d2; el[6 + C] = p1.x;
          };
          rgb(std::integral_constant<int, 0>{});
          rgb(std::integral_constant<int, 1>{});
          rgb(std::integral_constant<int, 2>{});
#pragma unroll
          for (int g = 0; g < NP; ++g)
            ring[j][g] = (v2f32){el[2 * g], el[2 * g + 1]};
          rtail[j] = el[DE - 1];
          (void)channel;
        } else if constexpr (ES == 1) {
          channel(std::integral_constant<int, 0>{}, ring[j][0], ring[j][1], ring[j][2]);
        } else {
          v2f32 u0, u1, u2;
          channel(std::integral_constant<int, 0>{}, u0, u1, u2);
          channel(std::integral_constant<int, 1>{}, u0, u1, u2);
#pragma unroll
          for (int g = 0; g < NP; ++g)
            ring[j][g] = (v2f32){el[2 * g], el[2 * g + 1]};
        }
      }
      if constexpr (ES == 3)
        __builtin_amdgcn_sched_barrier(0); // (the row pass is over before the column pass starts: their registers do not add up)
      issue(t + AHEAD, pf[j % AHEAD], hf[j % AHEAD]);
      if constexpr (ES == 3)
        __builtin_amdgcn_sched_barrier(0);
      if (!live || t < T0)
        continue;
      // ---- dst rows this source row completes.  Newest filtered row: slot j; logical row r of a window: slot (j + 1 + r) % TAPS
      auto full_row = [&]() {
        if constexpr (ES == 3) {
          v2f32 wp[TAPS / 2];
          const float4 qa = *reinterpret_cast<const float4*>(wt);
          wp[0] = (v2f32){qa.x, qa.y};
          wp[1] = (v2f32){qa.z, qa.w};
          if constexpr (TAPS == 6) {
            const float2 qb = *reinterpret_cast<const float2*>(wt + 4);
            wp[2] = (v2f32){qb.x, qb.y};
          }
          v2f32 v[NP];
#pragma unroll
          for (int g = 0; g < NP; ++g) {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(v[g]) : "v"(wp[0]), "v"(ring[(j + 1) % TAPS][g]));
            rwr_pk_fma<1>(v[g], wp[0], ring[(j + 2) % TAPS][g]);
#pragma unroll
            for (int k = 1; k < TAPS / 2; ++k) {
              rwr_pk_fma<0>(v[g], wp[k], ring[(j + 1 + 2 * k) % TAPS][g]);
              rwr_pk_fma<1>(v[g], wp[k], ring[(j + 2 + 2 * k) % TAPS][g]);
            }
          }
          float vt = wp[0].x * rtail[(j + 1) % TAPS];
          vt = __builtin_fmaf(wp[0].y, rtail[(j + 2) % TAPS], vt);
#pragma unroll
          for (int k = 1; k < TAPS / 2; ++k) {
            vt = __builtin_fmaf(wp[k].x, rtail[(j + 1 + 2 * k) % TAPS], vt);
            vt = __builtin_fmaf(wp[k].y, rtail[(j + 2 + 2 * k) % TAPS], vt);
          }
          store_row(v, vt);
          return;
        }
        v2f32 wy[TAPS];
#pragma unroll
        for (int k = 0; k < TAPS / 2; ++k) {
          const float4 q4 = *reinterpret_cast<const float4*>(wt + 4 * k);
          wy[2 * k] = (v2f32){q4.x, q4.y};
          wy[2 * k + 1] = (v2f32){q4.z, q4.w};
        }
        v2f32 v[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
          v[g] = wy[0] * ring[(j + 1) % TAPS][g];
#pragma unroll
          for (int r = 1; r < TAPS; ++r)
            v[g] = __builtin_elementwise_fma(wy[r], ring[(j + 1 + r) % TAPS][g], v[g]);
        }
        float vt = 0.0f;
        if constexpr (TAIL) {
          vt = wy[0].x * rtail[(j + 1) % TAPS];
#pragma unroll
          for (int r = 1; r < TAPS; ++r)
            vt = __builtin_fmaf(wy[r].x, rtail[(j + 1 + r) % TAPS], vt);
        }
        store_row(v, vt);
      };
      // parity of t == parity of j (TAPS is even).  TAPS = 6: odd rows 2M + 3 complete dst rows 3M (the filtered row 2M
      // itself: 3 slots back) and 3M + 1, even rows 2M + 4 complete 3M + 2.  TAPS = 4: even rows 2M + 2 complete 3M (2 slots
      // back) and 3M + 1, odd rows 2M + 3 complete 3M + 2.
      const bool two = TAPS == 6 ? (j & 1) == 1 : (j & 1) == 0; // (a constant once the walk is unrolled)
      if (two) {
        constexpr int back = TAPS == 6 ? 3 : 2;
        v2f32 c[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g)
          c[g] = ring[(j + TAPS - back) % TAPS][g];
        store_row(c, rtail[(j + TAPS - back) % TAPS]);
        if (rr <= last_rr)
          full_row();
      } else {
        full_row();
      }
    }
    t0 += TAPS;
  } while (rr <= last_rr);
}

template <typename T, int ESSET, int TAPS, int RW>
__global__ void __launch_bounds__(kBlock) k_resize_rows_x23(const ResizeArgs a) {
  extern __shared__ uint4 rows_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  float* const lds = reinterpret_cast<float*>(rows_lds);
  if (ESSET != 1 && job.channels == 2)
    rows23_tile<T, 2, TAPS, RW>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, lds, a.lds_per_wave / 4);
  else if constexpr (ESSET != 2)
    rows23_tile<T, 1, TAPS, RW>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, lds, a.lds_per_wave / 4);
}

// packed 8-bit RGB: 9 register pairs more than the planes of one / two channels -- asked to fit 4 waves per SIMD
template <int TAPS, int RW>
__global__ void __launch_bounds__(kBlock, 4) k_resize_rows_x23_rgb(const ResizeArgs a) {
  extern __shared__ uint4 rows_lds[];
  ResizeJob job;
  u32 tx, ty, frame;
  if (!plane_tile(a.job, a.njobs, a.map, job, tx, ty, frame))
    return;
  const PlaneView v = plane_view(a.d_src, a.d_dst, frame, job, a.sw, a.sh, a.dw, a.dh);
  rows23_tile<uint8_t, 3, TAPS, RW>(v.sp, v.spitch, v.sw, v.sh, v.dp, v.dpitch, v.dw, v.dh, tx, ty, reinterpret_cast<float*>(rows_lds),
                                    a.lds_per_wave / 4);
}

bool resize_x23_fits(const ResizeJob& j, int elem, int src_w, int src_h, int dst_w, int dst_h) {
  const long long sw = src_w >> j.ssub_x, sh = src_h >> j.ssub_y, dw = dst_w >> j.sub_x, dh = dst_h >> j.sub_y;
  return (elem == 1 || (elem == 2 && j.channels <= 2)) && j.channels <= 3 && 3 * sw == 2 * dw && 3 * sh == 2 * dh && sw >= 4 && sh >= 2 &&
         sw < (1 << 22) && sh < (1 << 22);
}

template <typename T, int ESSET, int TAPS>
static void launch_x23_k(const ResizeArgs& a, int rw, dim3 grid, unsigned lds, hipStream_t stream) {
  if constexpr (ESSET == 3) {
    if (rw == 48)
      hipLaunchKernelGGL((k_resize_rows_x23_rgb<TAPS, 48>), grid, dim3(kBlock), lds, stream, a);
    else
      hipLaunchKernelGGL((k_resize_rows_x23_rgb<TAPS, 12>), grid, dim3(kBlock), lds, stream, a);
  } else if (rw == 48)
    hipLaunchKernelGGL((k_resize_rows_x23<T, ESSET, TAPS, 48>), grid, dim3(kBlock), lds, stream, a);
  else
    hipLaunchKernelGGL((k_resize_rows_x23<T, ESSET, TAPS, 12>), grid, dim3(kBlock), lds, stream, a);
}

int launch_resize_x23(const ResizeArgs& base, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                      hipStream_t stream) {
  const int force = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE); // 1 / 2: 12-row waves, 3: 48-row waves whatever the launch size
  ResizeArgs a = base;
  int esset = 0;
  for (int k = 0; k < a.njobs; ++k) {
    const int c = a.job[k].channels;
    esset = c == 3 ? 3 : esset == 0 ? (c == 2 ? 2 : 1) : ((esset == 1 && c == 2) || (esset == 2 && c == 1)) ? 12 : esset; // (packed RGB: one plane)
  }
  auto count = [&](int rw, bool assign) {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dw = dst_w >> a.job[k].sub_x, dh = dst_h >> a.job[k].sub_y;
      const u32 tile_el = a.job[k].channels == 3 ? 576u : 384u; // 64 lanes x 9 / 6 dst elements
      const u32 tiles_x = ((u32)(dw * a.job[k].channels) + tile_el - 1) / tile_el;
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (u32)((dh + kWavesPerBlock * rw - 1) / (kWavesPerBlock * rw));
    }
    return total;
  };
  const unsigned long long t48 = (unsigned long long)count(48, false) * (unsigned)n;
  const int rw = force == 1 || force == 2 ? 12 : force == 3 ? 48 : t48 >= 1024ull ? 48 : 12;
  a.map = make_tile_map_linear(count(rw, true), (u32)n);
  const int wt_floats = rw * (4 + 2 * taps);
  a.lds_per_wave = ((wt_floats > 512 ? wt_floats : 512) * 4 + 15) & ~15; // the weight table, or the 64 x 8 floats of the tap exchange
  const unsigned lds = (unsigned)a.lds_per_wave * kWavesPerBlock;
  const dim3 grid = tile_grid(a.map);
#define VALI_X23_T(T)                                                                          \
  do {                                                                                         \
    if (taps == 6) {                                                                           \
      if (esset == 3) launch_x23_k<uint8_t, 3, 6>(a, rw, grid, lds, stream);                    \
      else if (esset == 1) launch_x23_k<T, 1, 6>(a, rw, grid, lds, stream);                     \
      else if (esset == 12) launch_x23_k<T, 12, 6>(a, rw, grid, lds, stream);                   \
      else launch_x23_k<T, 2, 6>(a, rw, grid, lds, stream);                                     \
    } else {                                                                                   \
      if (esset == 3) launch_x23_k<uint8_t, 3, 4>(a, rw, grid, lds, stream);                    \
      else if (esset == 1) launch_x23_k<T, 1, 4>(a, rw, grid, lds, stream);                     \
      else if (esset == 12) launch_x23_k<T, 12, 4>(a, rw, grid, lds, stream);                   \
      else launch_x23_k<T, 2, 4>(a, rw, grid, lds, stream);                                     \
    }                                                                                          \
  } while (0)
  if (elem == 1) VALI_X23_T(uint8_t);
  else VALI_X23_T(uint16_t);
#undef VALI_X23_T
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

template <typename T, int ESSET, int TAPS>
static void launch_rows_k(const ResizeArgs& a, int rows, dim3 grid, unsigned lds, hipStream_t stream) {
  if (rows == 64)
    hipLaunchKernelGGL((k_resize_rows<T, ESSET, TAPS, 64>), grid, dim3(kBlock), lds, stream, a);
  else if (rows == 32)
    hipLaunchKernelGGL((k_resize_rows<T, ESSET, TAPS, 32>), grid, dim3(kBlock), lds, stream, a);
  else if (rows == 8)
    hipLaunchKernelGGL((k_resize_rows<T, ESSET, TAPS, 8>), grid, dim3(kBlock), lds, stream, a);
  else
    hipLaunchKernelGGL((k_resize_rows<T, ESSET, TAPS, 2>), grid, dim3(kBlock), lds, stream, a);
}

// groups of 4 source pixels a 256-element tile of job j can span (taps included, group alignment slop on the left)
static int rows_groups(const ResizeJob& j, int src_w, int dst_w, int taps) {
  const int c = j.channels;
  const long long sw = src_w >> j.ssub_x, dw = dst_w >> j.sub_x;
  const long long px = ((long long)(kRwTile / c) * sw + dw - 1) / dw + taps + 1 + 3;
  return (int)((px + 3) / 4 + 2);
}

bool resize_rows_fits(const ResizeJob& j, int elem, int src_w, int dst_w, int taps) {
  return (elem == 1 || elem == 2) && j.channels <= 2 && rows_groups(j, src_w, dst_w, taps) <= kWave;
}

int launch_resize_rows(const ResizeArgs& base, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream) {
  const bool gather_only = tuning(VALI_TUNE_RESIZE_FORCE_GATHER) == 1;
  const int force = tuning(VALI_TUNE_RESIZE_NO_SEPARABLE); // 1 / 2 / 3: 8- / 2- / 32-row waves whatever the launch size
  ResizeArgs a = base;
  int groups = 0, esset = 0, maxc = 1;
  for (int k = 0; k < a.njobs; ++k) {
    const int c = a.job[k].channels;
    esset = esset == 0 ? (c == 2 ? 2 : 1) : ((esset == 1 && c == 2) || (esset == 2 && c == 1)) ? 12 : esset;
    maxc = c > maxc ? c : maxc;
    const int g = rows_groups(a.job[k], src_w, dst_w, taps);
    groups = g > groups ? g : groups;
  }
  auto count = [&](int rows, bool assign) {
    u32 total = 0;
    for (int k = 0; k < a.njobs; ++k) {
      const int dw = dst_w >> a.job[k].sub_x, dh = dst_h >> a.job[k].sub_y;
      const u32 tiles_x = (u32)(dw * a.job[k].channels + kRwTile - 1) / kRwTile;
      if (assign) {
        a.job[k].first_tile = total;
        a.job[k].tiles_x = tiles_x;
      }
      total += tiles_x * (u32)((dh + kWavesPerBlock * rows - 1) / (kWavesPerBlock * rows));
    }
    return total;
  };
  const unsigned long long t32 = (unsigned long long)count(32, false) * (unsigned)n, t8 = (unsigned long long)count(8, false) * (unsigned)n;
  // 64-row waves (round 4: a wave's TAPS - 1 extra source rows weigh half as much as with 32; 8 / 2-row waves measured 1.5 x /
  // 4 x SLOWER on a full launch: this kernel wants long waves) when the launch stays full and the wave's source rows still
  // fit its 64-entry completion table
  bool fits64 = true;
  for (int k = 0; k < a.njobs; ++k)
    fits64 = fits64 && 64.0 * (double)(src_h >> a.job[k].ssub_y) / (double)(dst_h >> a.job[k].sub_y) * (1.0 + 1e-6) + taps + 2 <= 64.0;
  const unsigned long long t64 = (unsigned long long)count(64, false) * (unsigned)n;
  const int rows = force == 1 ? 8 : force == 2 ? 2 : force == 3 ? 32 : (fits64 && (force == 4 || t64 >= 1024ull)) ? 64 : t32 >= 1024ull ? 32 : t8 >= 320ull ? 8 : 2;
  a.map = make_tile_map_linear(count(rows, true), (u32)n);
  a.force_gather = gather_only ? 1 : 0;
  (void)groups;
  // the register form (no LDS stage): 8-bit planes that grow along x as well, Lanczos-3 (RESIZE_ROWS = 3: the staged form, A/B and tests)
  {
    bool reg = elem == 1 && taps == 6 && !gather_only && tuning(VALI_TUNE_RESIZE_ROWS) != 3;
    int rd2 = 1, rd3 = 2;
    bool have = false;
    for (int k = 0; k < a.njobs && reg; ++k) {
      const int c = a.job[k].channels;
      int d2 = 0, d3 = 0;
      reg = c <= 2 && rows_reg_fits(src_w >> a.job[k].ssub_x, dst_w >> a.job[k].sub_x, c, d2, d3);
      if (reg && c == 1) {
        reg = !have || (d2 == rd2 && d3 == rd3);   // one launch, one set of window offsets
        rd2 = d2; rd3 = d3;
        have = true;
      }
    }
    if (reg) {
      a.lds_per_wave = (kRrShare + rows * (4 + 2 * 6) + 64) * 4;
      const unsigned lds_reg = (unsigned)a.lds_per_wave * kWavesPerBlock;
      const dim3 grid_reg = tile_grid(a.map);
#define VALI_RR(R, A, B) hipLaunchKernelGGL((k_resize_rows_reg<R, A, B>), grid_reg, dim3(kBlock), lds_reg, stream, a)
#define VALI_RR_D(R)                                  \
  do {                                                \
    if (rd2 == 0) VALI_RR(R, 0, 1);                   \
    else if (rd3 == 1) VALI_RR(R, 1, 1);              \
    else VALI_RR(R, 1, 2);                            \
  } while (0)
      if (rows == 64) VALI_RR_D(64);
      else if (rows == 32) VALI_RR_D(32);
      else if (rows == 8) VALI_RR_D(8);
      else VALI_RR_D(2);
#undef VALI_RR_D
#undef VALI_RR
      VALI_LAUNCH_CHECK();
      return VALI_OK;
    }
  }
  a.stage_bytes = kRwSf * 4;
  a.lds_per_wave = ((rw_wave_floats(maxc, rows, taps) * 4) + 15) & ~15;
  const unsigned lds = (unsigned)a.lds_per_wave * kWavesPerBlock;
  const dim3 grid = tile_grid(a.map);
#define VALI_ROWS_T(T)                                                                         \
  do {                                                                                         \
    if (taps == 6) {                                                                           \
      if (esset == 1) launch_rows_k<T, 1, 6>(a, rows, grid, lds, stream);                       \
      else if (esset == 12) launch_rows_k<T, 12, 6>(a, rows, grid, lds, stream);                \
      else launch_rows_k<T, 2, 6>(a, rows, grid, lds, stream);                                  \
    } else {                                                                                   \
      if (esset == 1) launch_rows_k<T, 1, 4>(a, rows, grid, lds, stream);                       \
      else if (esset == 12) launch_rows_k<T, 12, 4>(a, rows, grid, lds, stream);                \
      else launch_rows_k<T, 2, 4>(a, rows, grid, lds, stream);                                  \
    }                                                                                          \
  } while (0)
  if (elem == 1) VALI_ROWS_T(uint8_t);
  else VALI_ROWS_T(uint16_t);
#undef VALI_ROWS_T
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali
