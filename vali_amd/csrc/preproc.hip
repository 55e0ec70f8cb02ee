// Fused inference pre-processing: NV12 -> (bilinear resize) -> RGB -> float -> normalised,
// planar or packed, in ONE pass.
//
// The reference has no such kernel; its samples and tests/test_TorchSegmentation.py:176-240
// run the chain
//   PySurfaceConverter NV12 -> RGB            (nppiNV12ToRGB*_8u_P2C3R)
//   PySurfaceConverter RGB -> RGB_32F         (nppiScale_8u32f_C3R: v / 255)
//   PySurfaceConverter RGB_32F -> RGB_32F_PLANAR
//   torch.divide(x, 255.0) ; torchvision Normalize: (x - mean[c]) / std[c]
// (optionally behind a PySurfaceResizer), i.e. three to four surface launches plus two torch
// kernels and ~80 bytes of HBM traffic per pixel.  Here: 1.5 B read + 12 B written per pixel.
//
// DEFINITION = that chain, step by step, so the result is bit-identical to running the chain
// with this library's own kernels (tests/test_gpu_preproc.py checks exactly that):
//   1. sizes differ: NV12' = bilinear resize of the Y plane and of the interleaved UV plane on
//      the grid src = dst * (src_size / dst_size), each rounded half-even to u8 (resize.hip)
//   2. q_c = the u8 RGB of cvt_nv12_rgb.hip (vali_csc coefficients, round-half-even, saturate)
//   3. out_c = ((q_c / 255.0f) / div - mean[c]) / std[c]            three IEEE divisions
// Step 3 depends only on (c, q): each workgroup evaluates it ONCE for all 3 x 256 inputs into
// an LDS table with exactly those operations and every pixel then costs one ds_read_b32 --
// same bits, no per-pixel divisions.
//
// Work decomposition: lane = 4 dst px x 2 rows (one chroma sample pair per row pair), so each
// planar store instruction is one float4 per lane = 1 KiB contiguous per wave; a workgroup
// covers 1024 x 8 dst pixels (the 4 waves side by side, each walking 4 row pairs, so a plane
// row leaves the workgroup as 4 KiB contiguous) and the table is built once per workgroup.  Same-size inputs stream (dword luma loads); resized inputs gather their taps
// through L1/L2 (the destination of a network input is small, the traffic is the source).
// Oracle: composition of vali_oracle_resize_plane, vali_oracle_nv12_to_rgb and float32
// numpy arithmetic (tests/test_gpu_preproc.py, tests/test_oracle_preproc.py).
#include "common.hpp"
#include "dev_util.hpp"

namespace vali {

struct PreprocArgs {
  const vali_surface* d_src;
  const vali_surface* d_dst;
  vali_surface src, dst;
  vali_preproc_params prm;
  TileMap map;
  int row_pairs; // dst row pairs a wave walks: kPpRowPairsPerWave, or 2 / 1 when the launch is small
};

constexpr int kPpRowPairsPerWave = 4;
constexpr int kPpTileH = kPpRowPairsPerWave * 2;                 // 8 dst rows
constexpr int kPpTileW = kWavesPerBlock * kWave * 4;             // 1024 dst px: the 4 waves sit side by side

// SAME = source and destination sizes are equal (decided on the host): that instantiation
// carries none of the resize code and fits twice as many waves per SIMD.
// OUT: destination layout.  The float layouts apply step 3 through the table; the 8-bit ones
// stop after step 2 (resize + colour conversion only: the other two-task chain of the samples).
enum : int { PP_F32_PLANAR = 0, PP_F32_PACKED = 1, PP_U8_RGB = 2, PP_U8_BGR = 3, PP_U8_PLANAR = 4 };

template <bool SAME, int OUT>
__global__ void __launch_bounds__(kBlock) k_nv12_preproc(const PreprocArgs a) {
  constexpr bool kFloat = OUT == PP_F32_PLANAR || OUT == PP_F32_PACKED;
  __shared__ float lut[kFloat ? 3 : 1][256];
  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;
  if constexpr (kFloat) {
    // step 3 for every (channel, u8 value)
    for (int e = threadIdx.x; e < 3 * 256; e += kBlock) {
      const int c = e >> 8, q = e & 255;
      const float f = (float)q / 255.0f;
      const float g = f / a.prm.div;
      lut[c][q] = (g - a.prm.mean[c]) / a.prm.std_[c];
    }
    __syncthreads();
  }

  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const uint8_t* py = s.p[0];
  const uint8_t* puv = s.p[1];
  const int sp_y = s.pitch[0], sp_uv = s.pitch[1], sw = s.width, sh = s.height;
  const int dw = d.width, dh = d.height, dp = d.pitch[0];
  const vali_csc k = a.prm.csc;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int x0 = tile_x * kPpTileW + (wave * kWave + lane) * 4;
  if (x0 >= dw)
    return;
  const int n = min(4, dw - x0); // dw is even: n is 2 or 4
  constexpr bool same = SAME;
  const bool fast_luma = same && ((((uintptr_t)py) | (uintptr_t)sp_y | ((uintptr_t)puv) | (uintptr_t)sp_uv) & 3u) == 0;

  // resize geometry (resize.hip): per plane scale = src_size / dst_size
  const float lsx = (float)sw / (float)dw, lsy = (float)sh / (float)dh;
  const float csx = (float)(sw >> 1) / (float)(dw >> 1), csy = (float)(sh >> 1) / (float)(dh >> 1);
  Lerp lx[4], cxl[2];
  if constexpr (!same) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
      lx[p] = make_lerp(min(x0 + p, dw - 1), lsx, sw);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      cxl[j] = make_lerp(min((x0 >> 1) + j, (dw >> 1) - 1), csx, sw >> 1);
  }

#pragma unroll 1
  for (int it = 0; it < a.row_pairs; ++it) {
    const int y0 = (tile_y * a.row_pairs + it) * 2; // wave-uniform
    if (y0 >= dh)
      break;
    // ---- the resized NV12' texels of this lane: 2 x 4 luma, 2 chroma pairs ----
    float yv[2][4], uu[2], vv[2];
    if constexpr (same) {
      const uint8_t* r0 = py + (u32)(y0 * sp_y) + x0;
      const uint8_t* r1 = py + (u32)((y0 + 1) * sp_y) + x0;
      const uint8_t* rc = puv + (u32)((y0 >> 1) * sp_uv) + x0;
      u32 w0, w1, wc;
      if (fast_luma && n == 4) {
        w0 = gload<u32>(r0); w1 = gload<u32>(r1); wc = gload<u32>(rc);
      } else {
        w0 = w1 = wc = 0;
        for (int p = 0; p < n; ++p) {
          w0 |= (u32)gload<uint8_t>(r0 + p) << (8 * p);
          w1 |= (u32)gload<uint8_t>(r1 + p) << (8 * p);
          wc |= (u32)gload<uint8_t>(rc + p) << (8 * p);
        }
      }
      yv[0][0] = ubyte_f32<0>(w0); yv[0][1] = ubyte_f32<1>(w0); yv[0][2] = ubyte_f32<2>(w0); yv[0][3] = ubyte_f32<3>(w0);
      yv[1][0] = ubyte_f32<0>(w1); yv[1][1] = ubyte_f32<1>(w1); yv[1][2] = ubyte_f32<2>(w1); yv[1][3] = ubyte_f32<3>(w1);
      uu[0] = ubyte_f32<0>(wc); vv[0] = ubyte_f32<1>(wc); uu[1] = ubyte_f32<2>(wc); vv[1] = ubyte_f32<3>(wc);
    } else {
      // bilinear taps, arithmetic of resize_tile (t0, t1, v; then round-half-even to u8)
      // The two horizontal taps are neighbours (i1 = min(i0 + 1, last)), so a row's pair comes from
      // ONE (unaligned) load -- 2 bytes of luma, 4 bytes = two UV pairs of chroma: 20 vector-memory
      // instructions per lane and row pair instead of 48; the texture addresser was this path's
      // bound (profiles/r01_ud_down2.md: ~16 cycles per wave instruction whatever its width).
      // At the right edge (i0 = last) the load starts one texel earlier and both taps take its
      // second half.
      typedef uint16_t u16_unaligned __attribute__((aligned(1)));
      typedef u32 u32_unaligned __attribute__((aligned(1)));
      auto lerp3 = [](float t00, float t10, float t01, float t11, float ax, float ay) {
        const float t0 = __builtin_fmaf(ax, t10 - t00, t00);
        const float t1 = __builtin_fmaf(ax, t11 - t01, t01);
        return (float)quantize_u8(__builtin_fmaf(ay, t1 - t0, t0));
      };
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const Lerp ly = make_lerp(min(y0 + r, dh - 1), lsy, sh);
        const uint8_t* r0 = py + (size_t)ly.i0 * sp_y;
        const uint8_t* r1 = py + (size_t)ly.i1 * sp_y;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int base = min(lx[p].i0, sw - 2);
          const bool edge = lx[p].i0 != base;
          const u32 w0 = *(const VALI_GLOBAL u16_unaligned*)(r0 + base);
          const u32 w1 = *(const VALI_GLOBAL u16_unaligned*)(r1 + base);
          const float t10 = (float)(w0 >> 8), t11 = (float)(w1 >> 8);
          const float t00 = edge ? t10 : (float)(w0 & 0xffu), t01 = edge ? t11 : (float)(w1 & 0xffu);
          yv[r][p] = lerp3(t00, t10, t01, t11, lx[p].a, ly.a);
        }
      }
      const Lerp cy = make_lerp(y0 >> 1, csy, sh >> 1);
      const uint8_t* c0 = puv + (size_t)cy.i0 * sp_uv;
      const uint8_t* c1 = puv + (size_t)cy.i1 * sp_uv;
      if (sw >= 4) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int base = min(cxl[j].i0, (sw >> 1) - 2);
          const bool edge = cxl[j].i0 != base;
          const u32 w0 = *(const VALI_GLOBAL u32_unaligned*)(c0 + 2 * base); // U V U' V'
          const u32 w1 = *(const VALI_GLOBAL u32_unaligned*)(c1 + 2 * base);
          const float u10 = ubyte_f32<2>(w0), v10 = ubyte_f32<3>(w0), u11 = ubyte_f32<2>(w1), v11 = ubyte_f32<3>(w1);
          const float u00 = edge ? u10 : ubyte_f32<0>(w0), v00 = edge ? v10 : ubyte_f32<1>(w0);
          const float u01 = edge ? u11 : ubyte_f32<0>(w1), v01 = edge ? v11 : ubyte_f32<1>(w1);
          uu[j] = lerp3(u00, u10, u01, u11, cxl[j].a, cy.a);
          vv[j] = lerp3(v00, v10, v01, v11, cxl[j].a, cy.a);
        }
      } else { // a 2-pixel-wide source has ONE chroma pair per row: no neighbour to fetch with it
        const float u0 = (float)gload<uint8_t>(c0), v0 = (float)gload<uint8_t>(c0 + 1);
        const float u1 = (float)gload<uint8_t>(c1), v1 = (float)gload<uint8_t>(c1 + 1);
        uu[0] = uu[1] = lerp3(u0, u0, u1, u1, 0.0f, cy.a);
        vv[0] = vv[1] = lerp3(v0, v0, v1, v1, 0.0f, cy.a);
      }
    }
    // ---- step 2 + 3 ----
    const ChromaTerm ct[2] = {chroma_term(uu[0], vv[0], k), chroma_term(uu[1], vv[1], k)};
    if constexpr (kFloat) {
      float o[2][3][4]; // [row][channel][pixel]
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float yf = luma_term(yv[r][p], k);
          const ChromaTerm& c = ct[p >> 1];
          o[r][0][p] = lut[0][quantize_u8(yf + c.rv)];
          o[r][1][p] = lut[1][quantize_u8(yf + c.guv)];
          o[r][2][p] = lut[2][quantize_u8(yf + c.bu)];
        }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int y = y0 + r;
        if (y >= dh)
          break;
        if constexpr (OUT == PP_F32_PLANAR) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            uint8_t* q = d.p[c] + (u32)(y * dp) + (size_t)x0 * 4;
            // a wave writes 1 KiB, the workgroup 4 KiB contiguous per plane row: non-temporal pays
            // in this layout (5.55 -> 5.29 us at 1080p; it cost 5-9 % with 256 x 32 tiles)
            if (n == 4 && (((uintptr_t)q) & 15u) == 0)
              store16f_nt(q, make_float4(o[r][c][0], o[r][c][1], o[r][c][2], o[r][c][3]));
            else
              for (int p = 0; p < n; ++p) gstore<float>(q + 4 * p, o[r][c][p]);
          }
        } else {
          uint8_t* q = d.p[0] + (u32)(y * dp) + (size_t)x0 * 12;
          if (n == 4 && (((uintptr_t)q) & 15u) == 0) {
            store16f(q + 0, make_float4(o[r][0][0], o[r][1][0], o[r][2][0], o[r][0][1]));
            store16f(q + 16, make_float4(o[r][1][1], o[r][2][1], o[r][0][2], o[r][1][2]));
            store16f(q + 32, make_float4(o[r][2][2], o[r][0][3], o[r][1][3], o[r][2][3]));
          } else {
            for (int p = 0; p < n; ++p) {
              gstore<float>(q + 12 * p, o[r][0][p]); gstore<float>(q + 12 * p + 4, o[r][1][p]); gstore<float>(q + 12 * p + 8, o[r][2][p]);
            }
          }
        }
      }
    } else {
      // 8-bit outputs: the quantised bytes themselves
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int y = y0 + r;
        if (y >= dh)
          break;
        float cr[4], cg[4], cb[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float yf = luma_term(yv[r][p], k);
          const ChromaTerm& c = ct[p >> 1];
          cr[p] = yf + c.rv; cg[p] = yf + c.guv; cb[p] = yf + c.bu;
        }
        if constexpr (OUT == PP_U8_PLANAR) {
          u32 w[3] = {0u, 0u, 0u};
          w[0] = pack_u8<0>(cr[0], w[0]); w[0] = pack_u8<1>(cr[1], w[0]); w[0] = pack_u8<2>(cr[2], w[0]); w[0] = pack_u8<3>(cr[3], w[0]);
          w[1] = pack_u8<0>(cg[0], w[1]); w[1] = pack_u8<1>(cg[1], w[1]); w[1] = pack_u8<2>(cg[2], w[1]); w[1] = pack_u8<3>(cg[3], w[1]);
          w[2] = pack_u8<0>(cb[0], w[2]); w[2] = pack_u8<1>(cb[1], w[2]); w[2] = pack_u8<2>(cb[2], w[2]); w[2] = pack_u8<3>(cb[3], w[2]);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            uint8_t* q = d.p[c] + (u32)(y * dp) + x0;
            if (n == 4 && (((uintptr_t)q) & 3u) == 0)
              gstore<u32>(q, w[c]);
            else
              for (int p = 0; p < n; ++p) gstore<uint8_t>(q + p, (uint8_t)(w[c] >> (8 * p)));
          }
        } else {
          // memory order f g l per pixel: f = R (RGB) or B (BGR), l the other one
          const float (&cf)[4] = OUT == PP_U8_RGB ? cr : cb;
          const float (&cl)[4] = OUT == PP_U8_RGB ? cb : cr;
          u32 d0 = 0u, d1 = 0u, d2 = 0u;
          d0 = pack_u8<0>(cf[0], d0); d0 = pack_u8<1>(cg[0], d0); d0 = pack_u8<2>(cl[0], d0); d0 = pack_u8<3>(cf[1], d0);
          d1 = pack_u8<0>(cg[1], d1); d1 = pack_u8<1>(cl[1], d1); d1 = pack_u8<2>(cf[2], d1); d1 = pack_u8<3>(cg[2], d1);
          d2 = pack_u8<0>(cl[2], d2); d2 = pack_u8<1>(cf[3], d2); d2 = pack_u8<2>(cg[3], d2); d2 = pack_u8<3>(cl[3], d2);
          uint8_t* q = d.p[0] + (u32)(y * dp) + (size_t)x0 * 3;
          if (n == 4 && (((uintptr_t)q) & 3u) == 0) {
            typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
            const v3u32 w = {d0, d1, d2};
            *(VALI_GLOBAL v3u32*)q = w;
          } else {
            const u32 ww[3] = {d0, d1, d2};
            for (int b = 0; b < 3 * n; ++b) gstore<uint8_t>(q + b, (uint8_t)(ww[b >> 2] >> (8 * (b & 3))));
          }
        }
      }
    }
  }
}

static int launch_preproc(PreprocArgs& a, int src_w, int src_h, int dst_w, int dst_h, int dst_fmt, int n,
                          hipStream_t stream) {
  int out;
  switch (dst_fmt) {
  case VALI_FMT_RGB_32F_PLANAR: out = PP_F32_PLANAR; break;
  case VALI_FMT_RGB_32F: out = PP_F32_PACKED; break;
  case VALI_FMT_RGB: out = PP_U8_RGB; break;
  case VALI_FMT_BGR: out = PP_U8_BGR; break;
  case VALI_FMT_RGB_PLANAR: out = PP_U8_PLANAR; break;
  default:
    return fail(VALI_ERR_UNSUPPORTED,
                "nv12_preproc: destination must be RGB_32F[_PLANAR], RGB, BGR or RGB_PLANAR (got %d)", dst_fmt);
  }
  // Row pairs per wave: 4, or 2 / 1 while that would leave SIMDs without a wave (the pairs of a wave run one memory round
  // trip after the other: ONE 1080p -> 640x384 frame took 8.3 us through 4-pair waves, 48 workgroups on 256 CUs).
  a.row_pairs = kPpRowPairsPerWave;
  {
    const int forced = tuning(VALI_TUNE_ROWS_PER_WAVE); // 2 / 4 / 8 dst rows
    const long long tiles_x = (dst_w + kPpTileW - 1) / kPpTileW;
    if (forced == 2 || forced == 4 || forced == 8)
      a.row_pairs = forced / 2;
    else
      while (a.row_pairs > 1 && tiles_x * ((dst_h + 2 * a.row_pairs - 1) / (2 * a.row_pairs)) * n * kWavesPerBlock < 2048)
        a.row_pairs /= 2;
  }
  a.map = make_tile_map((dst_w + kPpTileW - 1) / kPpTileW, (dst_h + 2 * a.row_pairs - 1) / (2 * a.row_pairs), (u32)n);
  const dim3 grid = tile_grid(a.map), block(kBlock);
  const bool same = src_w == dst_w && src_h == dst_h;
#define VALI_PP_CASE(O)                                                                      \
  case O:                                                                                   \
    if (same)                                                                               \
      hipLaunchKernelGGL((k_nv12_preproc<true, O>), grid, block, 0, stream, a);              \
    else                                                                                    \
      hipLaunchKernelGGL((k_nv12_preproc<false, O>), grid, block, 0, stream, a);             \
    break;
  switch (out) {
    VALI_PP_CASE(PP_F32_PLANAR)
    VALI_PP_CASE(PP_F32_PACKED)
    VALI_PP_CASE(PP_U8_RGB)
    VALI_PP_CASE(PP_U8_BGR)
    VALI_PP_CASE(PP_U8_PLANAR)
  }
#undef VALI_PP_CASE
  VALI_LAUNCH_CHECK();
  return VALI_OK;
}

} // namespace vali

using namespace vali;

extern "C" {

int vali_nv12_preproc(const vali_surface* src, const vali_surface* dst, const vali_preproc_params* params,
                      vali_stream_t stream) {
  VALI_REQUIRE(src && dst && params, "null argument");
  VALI_REQUIRE(src->format == VALI_FMT_NV12, "source must be NV12");
  VALI_REQUIRE(src->width >= 2 && src->height >= 2 && dst->width >= 2 && dst->height >= 2, "empty surface");
  VALI_REQUIRE(((src->width | src->height | dst->width | dst->height) & 1) == 0, "4:2:0 needs even sizes");
  VALI_REQUIRE(src->plane[0] && src->plane[1] && dst->plane[0], "null plane");
  VALI_REQUIRE(planes_fit_32bit(*src) && planes_fit_32bit(*dst), "plane of 4 GiB or more");
  if (dst->format == VALI_FMT_RGB_32F_PLANAR || dst->format == VALI_FMT_RGB_PLANAR)
    VALI_REQUIRE(dst->plane[1] && dst->plane[2], "null dst plane");
  PreprocArgs a = {};
  a.src = *src;
  a.dst = *dst;
  a.prm = *params;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_preproc(a, src->width, src->height, dst->width, dst->height, dst->format, 1, s);
}

int vali_nv12_preproc_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_width,
                            int src_height, int dst_width, int dst_height, int dst_format,
                            const vali_preproc_params* params, vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst && params, "null argument");
  VALI_REQUIRE(src_width >= 2 && src_height >= 2 && dst_width >= 2 && dst_height >= 2 &&
                   ((src_width | src_height | dst_width | dst_height) & 1) == 0,
               "bad geometry");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  PreprocArgs a = {};
  a.d_src = d_src;
  a.d_dst = d_dst;
  a.prm = *params;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_preproc(a, src_width, src_height, dst_width, dst_height, dst_format, n, s);
}

} // extern "C"
