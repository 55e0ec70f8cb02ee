// NV12 -> RGB / BGR (packed u8) and RGB_PLANAR (u8): the headline kernel.
//
// Replaces nppiNV12ToRGB_709HDTV/709CSC/(601)_8u_P2C3R_Ctx and the BGR twins the
// reference calls from nv12_rgb / nv12_bgr
// (reference: src/TC/src/TaskConvertSurface.cpp:61-156), plus the fused
// NV12 -> RGB_PLANAR pair BASELINE config 2 names (the reference needs the
// two-step chain NV12->RGB->RGB_PLANAR, TaskConvertSurface.cpp:737-766).
//
// Algorithmic traffic per pixel: 1.5 B read (Y + half a UV pair), 3 B written.
// Work decomposition: one lane = 16 px x 2 rows (the 2x2 chroma footprint of
// 8 UV pairs): loads 16 B Y(row0) + 16 B Y(row1) + 16 B UV, stores 2 x 48 B.
// One 256-thread workgroup = one row pair x 4096 px.  grid.x walks the tiles of a
// frame through the XCD-contiguous TileMap (dev_util.hpp), grid.y = frame.
// Packed stores go through a per-wave LDS strip so each store instruction writes
// 1 KiB contiguous, non-temporal.  Residency is capped at 16 waves per CU by the
// dynamic-LDS size: fewer concurrent row streams per L2 measured +4% HBM
// throughput at 2160p (profiles/r01_variants.md: 5.56 -> 6.16 TB/s in total).
// Arithmetic (bit-exact with oracle/vali_oracle.c: vali_oracle_nv12_to_rgb):
//   Yf = cy*(Y-y0); Uc=U-128; Vc=V-128
//   R = Yf + crv*Vc ; G = Yf + fma(cgu,Uc,cgv*Vc) ; B = Yf + cbu*Uc
//   u8 = saturate(round-half-even(.)).   Chroma siting: nearest (2x2 block).
#include "common.hpp"
#include "dev_util.hpp"

namespace vali {

enum : int { LAYOUT_RGB = 0, LAYOUT_BGR = 1, LAYOUT_PLANAR = 2 };

struct Nv12RgbArgs {
  const vali_surface* d_src; // device descriptor arrays (batch) or nullptr
  const vali_surface* d_dst;
  vali_surface src;          // by-value descriptors (single frame)
  vali_surface dst;
  vali_csc csc;
  TileMap map;               // tiles of one frame: x segments x row pairs
  int rp;                    // row pairs stacked in one workgroup (narrow frames)
};

// 4 pixels of one row: y4 = 4 luma bytes, chroma terms c01 (px 0,1) / c23 (px 2,3).
// Packed: writes dwords o[0..2] of the 12-byte group.  Planar: r/g/b dwords.
template <int LAYOUT>
__device__ __forceinline__ void emit4(u32 y4, const ChromaTerm& c01, const ChromaTerm& c23,
                                      const vali_csc& k, u32* o3, u32& pr, u32& pg,
                                      u32& pb) {
  const float y0 = luma_term(ubyte_f32<0>(y4), k), y1 = luma_term(ubyte_f32<1>(y4), k),
              y2 = luma_term(ubyte_f32<2>(y4), k), y3 = luma_term(ubyte_f32<3>(y4), k);
  const float r0 = y0 + c01.rv, g0 = y0 + c01.guv, b0 = y0 + c01.bu;
  const float r1 = y1 + c01.rv, g1 = y1 + c01.guv, b1 = y1 + c01.bu;
  const float r2 = y2 + c23.rv, g2 = y2 + c23.guv, b2 = y2 + c23.bu;
  const float r3 = y3 + c23.rv, g3 = y3 + c23.guv, b3 = y3 + c23.bu;
  if constexpr (LAYOUT == LAYOUT_PLANAR) {
    u32 r = 0, g = 0, b = 0;
    r = pack_u8<0>(r0, r); r = pack_u8<1>(r1, r); r = pack_u8<2>(r2, r); r = pack_u8<3>(r3, r);
    g = pack_u8<0>(g0, g); g = pack_u8<1>(g1, g); g = pack_u8<2>(g2, g); g = pack_u8<3>(g3, g);
    b = pack_u8<0>(b0, b); b = pack_u8<1>(b1, b); b = pack_u8<2>(b2, b); b = pack_u8<3>(b3, b);
    pr = r; pg = g; pb = b;
  } else {
    // first/last channel in memory order
    const float f0 = LAYOUT == LAYOUT_RGB ? r0 : b0, l0 = LAYOUT == LAYOUT_RGB ? b0 : r0;
    const float f1 = LAYOUT == LAYOUT_RGB ? r1 : b1, l1 = LAYOUT == LAYOUT_RGB ? b1 : r1;
    const float f2 = LAYOUT == LAYOUT_RGB ? r2 : b2, l2 = LAYOUT == LAYOUT_RGB ? b2 : r2;
    const float f3 = LAYOUT == LAYOUT_RGB ? r3 : b3, l3 = LAYOUT == LAYOUT_RGB ? b3 : r3;
    u32 d0 = 0, d1 = 0, d2 = 0;
    d0 = pack_u8<0>(f0, d0); d0 = pack_u8<1>(g0, d0); d0 = pack_u8<2>(l0, d0); d0 = pack_u8<3>(f1, d0);
    d1 = pack_u8<0>(g1, d1); d1 = pack_u8<1>(l1, d1); d1 = pack_u8<2>(f2, d1); d1 = pack_u8<3>(g2, d1);
    d2 = pack_u8<0>(l2, d2); d2 = pack_u8<1>(f3, d2); d2 = pack_u8<2>(g3, d2); d2 = pack_u8<3>(l3, d2);
    o3[0] = d0; o3[1] = d1; o3[2] = d2;
  }
}

// The lane's 16 x 2 pixels: two luma vectors + the shared chroma vector -> packed rows o0 / o1
// (or r[4] g[4] b[4] per row when planar).
template <int LAYOUT>
__device__ __forceinline__ void convert_group(const uint4& ya, const uint4& yb, const uint4& uv, const vali_csc& k,
                                              u32 (&o0)[12], u32 (&o1)[12]) {
  const u32 yw0[4] = {ya.x, ya.y, ya.z, ya.w};
  const u32 yw1[4] = {yb.x, yb.y, yb.z, yb.w};
  const u32 uvw[4] = {uv.x, uv.y, uv.z, uv.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const ChromaTerm c01 = chroma_term(ubyte_f32<0>(uvw[j]), ubyte_f32<1>(uvw[j]), k);
    const ChromaTerm c23 = chroma_term(ubyte_f32<2>(uvw[j]), ubyte_f32<3>(uvw[j]), k);
    if constexpr (LAYOUT == LAYOUT_PLANAR) {
      emit4<LAYOUT>(yw0[j], c01, c23, k, nullptr, o0[j], o0[4 + j], o0[8 + j]);
      emit4<LAYOUT>(yw1[j], c01, c23, k, nullptr, o1[j], o1[4 + j], o1[8 + j]);
    } else {
      u32 dummy;
      emit4<LAYOUT>(yw0[j], c01, c23, k, &o0[3 * j], dummy, dummy, dummy);
      emit4<LAYOUT>(yw1[j], c01, c23, k, &o1[3 * j], dummy, dummy, dummy);
    }
  }
}

// STAGED: packed rows leave through the per-wave LDS strip (3 x 1 KiB contiguous per
// wave and row, non-temporal); !STAGED: each lane stores its own 3 x 16 B at a 48 B lane
// stride (kept as the measured alternative, see DESIGN.md "store path A/B").
template <int LAYOUT, bool STAGED>
__global__ void __launch_bounds__(kBlock) k_nv12_rgb8(const Nv12RgbArgs a) {
  // dynamic LDS: [0, 12 KiB) = the 4 per-wave strips; the rest only caps residency
  extern __shared__ uint4 dyn_lds[];
  PackedStrip* const strips = reinterpret_cast<PackedStrip*>(dyn_lds);

  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;

  const uint8_t* py;
  const uint8_t* puv;
  uint8_t* pd0;
  uint8_t* pd1;
  uint8_t* pd2;
  int sp_y, sp_uv, dp, W, H;
  if (a.d_src) {
    const vali_surface* s = a.d_src + frame;
    const vali_surface* d = a.d_dst + frame;
    py = (const uint8_t*)s->plane[0]; puv = (const uint8_t*)s->plane[1];
    sp_y = s->pitch[0]; sp_uv = s->pitch[1]; W = s->width; H = s->height;
    pd0 = (uint8_t*)d->plane[0]; pd1 = (uint8_t*)d->plane[1]; pd2 = (uint8_t*)d->plane[2];
    dp = d->pitch[0];
  } else {
    py = (const uint8_t*)a.src.plane[0]; puv = (const uint8_t*)a.src.plane[1];
    sp_y = a.src.pitch[0]; sp_uv = a.src.pitch[1]; W = a.src.width; H = a.src.height;
    pd0 = (uint8_t*)a.dst.plane[0]; pd1 = (uint8_t*)a.dst.plane[1]; pd2 = (uint8_t*)a.dst.plane[2];
    dp = a.dst.pitch[0];
  }
  const vali_csc k = a.csc;

  const int lane = threadIdx.x & (kWave - 1);
  const int wave_all = threadIdx.x / kWave;
  const int wpr = (int)(blockDim.x / kWave) / a.rp;        // waves side by side on one row pair
  const int wave = wave_all;                                // (LDS strip index)
  const int wave_g0 = (tile_x * wpr + wave_all % wpr) * kWave; // first group of this wave
  const int groups = (W + kLanePx - 1) / kLanePx;
  if (wave_g0 >= groups)
    return; // whole wave out of the row: nothing to cooperate on
  const int g = wave_g0 + lane;
  const int x0 = g * kLanePx;
  const int crow = tile_y * a.rp + wave_all / wpr;          // chroma row = row pair index
  if (crow * 2 >= H)
    return;
  const int row0 = crow * 2;          // H is even (checked on the host): row0 + 1 exists

  // Uniform (per frame) fast-path test: full 16-px groups and 16-B aligned rows.
  uintptr_t align_bits = (uintptr_t)py | (uintptr_t)puv | (uintptr_t)sp_y |
                         (uintptr_t)sp_uv | (uintptr_t)pd0 | (uintptr_t)dp;
  if constexpr (LAYOUT == LAYOUT_PLANAR)
    align_bits |= (uintptr_t)pd1 | (uintptr_t)pd2;
  const bool fast = ((W & (kLanePx - 1)) == 0) && ((align_bits & 15u) == 0);

  if (fast) {
    const bool lane_valid = g < groups;
    u32 o0[12], o1[12];          // packed rows (or r[4] g[4] b[4] when planar)
    if (lane_valid) {
      const uint4 ya = load16(py + (size_t)row0 * sp_y + x0);
      const uint4 yb = load16(py + (size_t)(row0 + 1) * sp_y + x0);
      const uint4 uv = load16(puv + (size_t)crow * sp_uv + x0);
      convert_group<LAYOUT>(ya, yb, uv, k, o0, o1);
    }
    if constexpr (LAYOUT == LAYOUT_PLANAR) {
      if (lane_valid) {
        uint8_t* const planes[3] = {pd0, pd1, pd2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          uint8_t* p = planes[c] + (size_t)row0 * dp + x0;
          store16_nt(p, make_uint4(o0[4 * c], o0[4 * c + 1], o0[4 * c + 2], o0[4 * c + 3]));
          store16_nt(p + dp, make_uint4(o1[4 * c], o1[4 * c + 1], o1[4 * c + 2], o1[4 * c + 3]));
        }
      }
    } else if constexpr (!STAGED) {
      if (lane_valid) {
        uint4* p = reinterpret_cast<uint4*>(pd0 + (size_t)row0 * dp + (size_t)g * 48);
        p[0] = make_uint4(o0[0], o0[1], o0[2], o0[3]);
        p[1] = make_uint4(o0[4], o0[5], o0[6], o0[7]);
        p[2] = make_uint4(o0[8], o0[9], o0[10], o0[11]);
        uint4* q = reinterpret_cast<uint4*>(pd0 + (size_t)(row0 + 1) * dp + (size_t)g * 48);
        q[0] = make_uint4(o1[0], o1[1], o1[2], o1[3]);
        q[1] = make_uint4(o1[4], o1[5], o1[6], o1[7]);
        q[2] = make_uint4(o1[8], o1[9], o1[10], o1[11]);
      }
    } else {
      PackedStrip& strip = strips[wave];
      const int valid_lanes = min(kWave, groups - wave_g0);
      const int valid_bytes = valid_lanes * 48;
      uint8_t* rb = pd0 + (size_t)row0 * dp + (size_t)wave_g0 * 48;
      strip_store_row(strip, lane, o0, lane_valid, rb, valid_bytes);
      strip_store_row(strip, lane, o1, lane_valid, rb + dp, valid_bytes);
    }
    return;
  }

  // Ragged path: any (even) width, any base pointer, any pitch -- 854x480, 1366x768, 1918x1078, a torch tensor
  // with an odd row stride.  The SAME decomposition and arithmetic at (nearly) the same speed: every access is
  // the misaligned form of the 16-byte access, and the one group per row that the right edge cuts slides its
  // window left to END with the row -- 16 whole pixels again, the overlap with its neighbour is computed twice
  // and stored twice with identical bytes.  Its packed pixels are off the strip's 48-byte lane grid, so that
  // lane stores its 48 bytes itself.  (Byte-granular accesses cost the memory pipeline as much as 16-byte ones:
  // a first version that moved the cut group's bytes one by one ran at 44-66 % of the aligned speed.)
  {
    const bool lane_valid = g < groups;
    const bool cut = lane_valid && x0 + kLanePx > W;
    const int xs = cut ? max(W - kLanePx, 0) : x0;
    const int n_px = min(kLanePx, W);                 // < 16 only for frames narrower than one group (uniform)
    u32 o0[12], o1[12];
    if (lane_valid) {
      const uint4 ya = load16_n(py + (size_t)row0 * sp_y + xs, n_px);
      const uint4 yb = load16_n(py + (size_t)(row0 + 1) * sp_y + xs, n_px);
      const uint4 uv = load16_n(puv + (size_t)crow * sp_uv + xs, n_px);
      convert_group<LAYOUT>(ya, yb, uv, k, o0, o1);
    }
    if constexpr (LAYOUT == LAYOUT_PLANAR) {
      if (lane_valid) {
        uint8_t* const planes[3] = {pd0, pd1, pd2};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          uint8_t* p = planes[c] + (size_t)row0 * dp + xs;
          store16_n(p, make_uint4(o0[4 * c], o0[4 * c + 1], o0[4 * c + 2], o0[4 * c + 3]), n_px);
          store16_n(p + dp, make_uint4(o1[4 * c], o1[4 * c + 1], o1[4 * c + 2], o1[4 * c + 3]), n_px);
        }
      }
    } else {
      PackedStrip& strip = strips[wave];
      const int full_lanes = min(kWave, W / kLanePx - wave_g0);      // whole groups of this wave
      uint8_t* rb = pd0 + (size_t)row0 * dp + (size_t)wave_g0 * 48;
      strip_store_row_u(strip, lane, o0, lane_valid && !cut, rb, full_lanes * 48);
      strip_store_row_u(strip, lane, o1, lane_valid && !cut, rb + dp, full_lanes * 48);
      if (cut) {
        packed_group_store(pd0 + (size_t)row0 * dp + (size_t)xs * 3, n_px, o0);
        packed_group_store(pd0 + (size_t)(row0 + 1) * dp + (size_t)xs * 3, n_px, o1);
      }
    }
  }
}

static int launch_nv12_rgb(Nv12RgbArgs& a, int width, int height, int n, int dst_format,
                           hipStream_t stream) {
  const int groups = (width + kLanePx - 1) / kLanePx;
  // workgroup width: smallest multiple of a wave covering the row, capped at 256
  int block = ((groups + kWave - 1) / kWave) * kWave;
  if (block > kBlock)
    block = kBlock;
  // Narrow frames (a row pair needs <= 128 lanes: up to 2048 px) stack 2 or 4 row pairs in one
  // 256-thread workgroup -- adjacent rows, the same residency in fewer, fuller workgroups:
  // 1080p 5.72 -> 6.03-6.19 TB/s, 720p 5.02 -> 5.11 (profiles/r01_variants.md sweep 8).
  // VALI_TUNE_NV12_ROWPAIRS = 1 restores one row pair per workgroup (A/B only).
  const int rp_env = tuning(VALI_TUNE_NV12_ROWPAIRS);
  const int row_block = block;
  a.rp = rp_env > 0 ? rp_env : kBlock / row_block;
  if (a.rp < 1 || row_block * a.rp > kBlock)
    a.rp = 1;
  a.map = make_tile_map((groups + row_block - 1) / row_block, ((height + 1) / 2 + a.rp - 1) / a.rp, (u32)n);
  block = row_block * a.rp;
  const dim3 grid = tile_grid(a.map);
  // A/B switches (include/vali_hip.h: vali_tuning_key)
  const bool direct = tuning(VALI_TUNE_NV12_DIRECT_STORE) == 1;
  const int waves_override = tuning(VALI_TUNE_WAVES_PER_CU);
  const int waves_per_cu = waves_override > 0 ? waves_override : streaming_waves_per_cu(groups, row_block, 16);
  const unsigned lds =
      residency_lds_bytes(block, waves_per_cu, (unsigned)sizeof(PackedStrip) * (unsigned)((block + kWave - 1) / kWave));
  switch (dst_format) {
  case VALI_FMT_RGB:
    if (direct)
      hipLaunchKernelGGL((k_nv12_rgb8<LAYOUT_RGB, false>), grid, dim3(block), lds, stream, a);
    else
      hipLaunchKernelGGL((k_nv12_rgb8<LAYOUT_RGB, true>), grid, dim3(block), lds, stream, a);
    break;
  case VALI_FMT_BGR:
    if (direct)
      hipLaunchKernelGGL((k_nv12_rgb8<LAYOUT_BGR, false>), grid, dim3(block), lds, stream, a);
    else
      hipLaunchKernelGGL((k_nv12_rgb8<LAYOUT_BGR, true>), grid, dim3(block), lds, stream, a);
    break;
  case VALI_FMT_RGB_PLANAR:
    hipLaunchKernelGGL((k_nv12_rgb8<LAYOUT_PLANAR, true>), grid, dim3(block), lds, stream, a);
    break;
  default:
    return fail(VALI_ERR_UNSUPPORTED, "nv12_to_rgb: unsupported dst format %d", dst_format);
  }
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali

using namespace vali;

extern "C" {

int vali_nv12_to_rgb(const vali_surface* src, const vali_surface* dst, const vali_csc* csc,
                     vali_stream_t stream) {
  VALI_REQUIRE(src && dst && csc, "null argument");
  VALI_REQUIRE(src->format == VALI_FMT_NV12, "src must be NV12");
  VALI_REQUIRE(src->width > 0 && src->height > 0, "empty src");
  VALI_REQUIRE(src->width == dst->width && src->height == dst->height, "src/dst size mismatch");
  VALI_REQUIRE(((src->width | src->height) & 1) == 0, "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(src->plane[0] && src->plane[1] && dst->plane[0], "null plane");
  if (dst->format == VALI_FMT_RGB_PLANAR)
    VALI_REQUIRE(dst->plane[1] && dst->plane[2], "null planar plane");
  Nv12RgbArgs a = {};
  a.src = *src;
  a.dst = *dst;
  a.csc = *csc;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_nv12_rgb(a, src->width, src->height, 1, dst->format, s);
}

int vali_nv12_to_rgb_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int width,
                           int height, int dst_format, const vali_csc* csc,
                           vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst && csc, "null argument");
  VALI_REQUIRE(width > 0 && height > 0, "empty geometry");
  VALI_REQUIRE(((width | height) & 1) == 0, "4:2:0 surfaces need even width and height");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  Nv12RgbArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  a.csc = *csc;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_nv12_rgb(a, width, height, n, dst_format, s);
}

// ---- diagnostics ------------------------------------------------------------

__global__ void k_debug_quantize(const float* in, uint8_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = (uint8_t)quantize_u8(in[i]);
}

__global__ void k_debug_quantize_portable(const float* in, uint8_t* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    out[i] = (uint8_t)quantize_u8_portable(in[i]);
}

int vali_debug_quantize_u8(const float* d_in, uint8_t* d_out, int n, vali_stream_t stream) {
  VALI_REQUIRE(d_in && d_out && n >= 0, "bad argument");
  if (!n)
    return VALI_OK;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  hipLaunchKernelGGL(k_debug_quantize, dim3((n + 255) / 256), dim3(256), 0, s, d_in, d_out, n);
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

int vali_debug_quantize_u8_portable(const float* d_in, uint8_t* d_out, int n,
                                    vali_stream_t stream) {
  VALI_REQUIRE(d_in && d_out && n >= 0, "bad argument");
  if (!n)
    return VALI_OK;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  hipLaunchKernelGGL(k_debug_quantize_portable, dim3((n + 255) / 256), dim3(256), 0, s, d_in,
                     d_out, n);
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // extern "C"
