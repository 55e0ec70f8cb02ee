// All 8-bit format-pair conversions of ConvertSurface except the tuned NV12->RGB kernel:
// one templated kernel k_cvt8<SRC,DST> + three element-type kernels (P10/P12->NV12,
// RGB->RGB_32F, RGB_32F->RGB_32F_PLANAR).
//
// Replaces the NPP entry points behind the reference's converter table
// (reference: src/TC/src/TaskConvertSurface.cpp:158-962; dispatch :1035-1090):
//   nv12_yuv420   nppiNV12ToYUV420_8u_P2P3R / nppiYCbCr420_8u_P2P3R      (:158-200)
//   yuv420_nv12   nppiYCbCr420_8u_P3P2R                                   (:706-735)
//   nv12_y        cuMemcpy2DAsync of the luma plane                       (:202-230)
//   rbg8_y        nppiRGBToGray_8u_C3C1R                                  (:232-252)
//   yuv420_rgb/bgr nppiYUV420To{RGB,BGR}_8u_P3C3R / nppiYCbCr420To...     (:254-344)
//   yuv444_bgr/rgb nppiYCbCrToBGR / nppiYUVTo{BGR,RGB}_8u_P3C3R           (:346-434)
//   bgr/rgb/rgb_planar -> yuv444  nppi{BGR,RGB}To{YUV,YCbCr}_8u_*         (:481-619)
//   y_yuv444      nppiSet_8u_C1R(128) x2 + nppiCopy_8u_C1R                (:621-655)
//   rgb_yuv420    nppiRGBToYUV420_8u_C3P3R / nppiRGBToYCbCr420_8u_C3P3R   (:657-704)
//   rgb8_(de)interleave nppiCopy_8u_C3P3R / P3C3R                         (:737-796)
//   rgb_bgr/bgr_rgb nppiSwapChannels_8u_C3R                               (:798-852)
//   rbg8_rgb32f   nppiScale_8u32f_C3R(0,1)                                (:854-884)
//   rgb32f_deinterleave nppiCopy_32f_C3P3R                                (:886-916)
//   p16_nv12      nppiDivC_16u_C1RSfs(256) + nppiConvert_16u8u_C1R        (:918-962)
//
// Work decomposition of k_cvt8 = the NV12->RGB kernel's: one lane owns 16 px x 2 rows
// (the 2x2 chroma footprint), 16-byte plane accesses, packed rows through the per-wave
// LDS strip, XCD-contiguous TileMap, nt stores, 16 waves/CU residency cap.
// Arithmetic (bit-exact with oracle/vali_oracle.c: vali_oracle_convert):
//   YUV->RGB : the vali_csc model of cvt_nv12_rgb.hip (nearest chroma for 4:2:0)
//   RGB->YUV : c = fma(kB, B, fma(kG, G, fma(kR, R, offset))) per output channel, then
//              round-half-even + saturate; 4:2:0 chroma = mean of the four un-rounded
//              values ((c00 + c01) + (c10 + c11)) * 0.25
//   copies / swaps / (de)interleaves: byte permutations, exact.
#include "common.hpp"
#include "dev_util.hpp"

namespace vali {

enum : int { K_NV12 = 0, K_YUV420 = 1, K_YUV444 = 2, K_RGB = 3, K_BGR = 4, K_RGBP = 5, K_Y = 6, K_NONE = -1 };

__host__ __device__ constexpr bool k_is420(int k) { return k == K_NV12 || k == K_YUV420; }
__host__ __device__ constexpr bool k_isyuv(int k) { return k == K_NV12 || k == K_YUV420 || k == K_YUV444; }
__host__ __device__ constexpr bool k_isrgb(int k) { return k == K_RGB || k == K_BGR || k == K_RGBP; }
__host__ __device__ constexpr bool k_ispacked(int k) { return k == K_RGB || k == K_BGR; }

struct CvtArgs {
  const vali_surface* d_src;
  const vali_surface* d_dst;
  vali_surface src, dst;
  vali_cvt_params p;
  TileMap map;
  int rp; // row pairs stacked in one workgroup (narrow frames, as in cvt_nv12_rgb.hip)
};

// ---- per-pixel arithmetic ---------------------------------------------------------------
struct Chroma {
  float rv, guv, bu;
};
__device__ __forceinline__ Chroma chroma_of(float u, float v, const vali_csc& k) {
  const float uc = u - 128.0f, vc = v - 128.0f;
  Chroma t;
  t.rv = k.crv * vc;
  t.guv = __builtin_fmaf(k.cgu, uc, k.cgv * vc);
  t.bu = k.cbu * uc;
  return t;
}
__device__ __forceinline__ float luma_of(float y, const vali_csc& k) { return k.cy * (y - k.y0); }

__device__ __forceinline__ float dot_rgb(const float (&m)[4], float r, float g, float b) {
  return __builtin_fmaf(m[2], b, __builtin_fmaf(m[1], g, __builtin_fmaf(m[0], r, m[3])));
}

// 4 pixels: (y4,u4,v4) dwords -> planar r,g,b dwords.  Per-pixel chroma (4:4:4 sources).
__device__ __forceinline__ void yuv444_to_rgb4(u32 y4, u32 u4, u32 v4, const vali_csc& k, u32& r,
                                               u32& g, u32& b) {
  r = g = b = 0;
#define VALI_PX(I)                                                                          \
  {                                                                                         \
    const Chroma c = chroma_of(ubyte_f32<I>(u4), ubyte_f32<I>(v4), k);                       \
    const float yf = luma_of(ubyte_f32<I>(y4), k);                                           \
    r = pack_u8<I>(yf + c.rv, r); g = pack_u8<I>(yf + c.guv, g); b = pack_u8<I>(yf + c.bu, b); \
  }
  VALI_PX(0) VALI_PX(1) VALI_PX(2) VALI_PX(3)
#undef VALI_PX
}

// 4 pixels of one row sharing two chroma samples (4:2:0 sources): px 0,1 <- ca ; px 2,3 <- cb
__device__ __forceinline__ void yuv420_to_rgb4(u32 y4, const Chroma& ca, const Chroma& cb,
                                               const vali_csc& k, u32& r, u32& g, u32& b) {
  const float y0 = luma_of(ubyte_f32<0>(y4), k), y1 = luma_of(ubyte_f32<1>(y4), k),
              y2 = luma_of(ubyte_f32<2>(y4), k), y3 = luma_of(ubyte_f32<3>(y4), k);
  r = g = b = 0;
  r = pack_u8<0>(y0 + ca.rv, r); r = pack_u8<1>(y1 + ca.rv, r); r = pack_u8<2>(y2 + cb.rv, r); r = pack_u8<3>(y3 + cb.rv, r);
  g = pack_u8<0>(y0 + ca.guv, g); g = pack_u8<1>(y1 + ca.guv, g); g = pack_u8<2>(y2 + cb.guv, g); g = pack_u8<3>(y3 + cb.guv, g);
  b = pack_u8<0>(y0 + ca.bu, b); b = pack_u8<1>(y1 + ca.bu, b); b = pack_u8<2>(y2 + cb.bu, b); b = pack_u8<3>(y3 + cb.bu, b);
}

// ---- the 16 x 2 pixel block a lane works on ----------------------------------------------
struct Block {
  u32 c0[2][4], c1[2][4], c2[2][4]; // full-resolution channels in canonical order (R,G,B / Y,U,V)
  u32 cu[2], cv[2];                 // 4:2:0 chroma: 8 samples each (shared by both rows)
};

struct Geo {
  int lane, wave, wave_g0, g, x0, row0, groups, valid_lanes;
  bool lane_valid, has_row1;
  u32 tile_y;
  // ragged path (any width / alignment): the group the right edge cuts slides left to END with the row (16 whole
  // pixels again; the overlap with its neighbour is computed and stored twice with identical bytes)
  int xs;          // first pixel of the lane's window: x0, or W - 16 for the cut group
  bool cut;        // this lane holds the cut group: off the strip's 48-byte lane grid, packed pixels go direct
  int n_px;        // 16; fewer only for frames narrower than one group (uniform)
  int full_lanes;  // whole groups of this wave = lanes that use the strip
};

// ANY = the ragged path (any width, base pointer and pitch): the misaligned forms of the same accesses
// (dev_util.hpp load16_n, strip_fetch_u, ...); !ANY = full 16-pixel groups on 16-byte aligned rows.
template <bool ANY> __device__ __forceinline__ uint4 ld16(const uint8_t* p, int n) {
  if constexpr (ANY) return load16_n(p, n);
  else return load16(p);
}
template <bool ANY> __device__ __forceinline__ void st16(uint8_t* p, uint4 v, int n) {
  if constexpr (ANY) store16_n(p, v, n);
  else store16_nt(p, v);
}
typedef v2u32 v2u32_u __attribute__((aligned(1)));
template <bool ANY> __device__ __forceinline__ uint2 ld8(const uint8_t* p, int n) {
  if constexpr (ANY) {
    if (n >= 8) {
      const v2u32 w = *(const v2u32_u*)p;
      return make_uint2(w.x, w.y);
    }
    const uint4 v = load_bytes16(p, n);
    return make_uint2(v.x, v.y);
  } else {
    return *reinterpret_cast<const uint2*>(p);
  }
}
template <bool ANY> __device__ __forceinline__ void st8(uint8_t* p, uint2 v, int n) {
  if constexpr (ANY) {
    if (n >= 8) {
      const v2u32 w = {v.x, v.y};
      *(v2u32_u*)p = w;
    } else {
      store_bytes16(p, make_uint4(v.x, v.y, 0u, 0u), n);
    }
  } else {
    *reinterpret_cast<uint2*>(p) = v;
  }
}

template <int SRC, bool ANY>
__device__ __forceinline__ void load_block(Block& b, const SurfRef& s, const Geo& q, PackedStrip& strip) {
  const int r1 = q.row0 + (q.has_row1 ? 1 : 0);
  if constexpr (k_ispacked(SRC)) {
    const uint8_t* rb = s.p[0] + (size_t)q.row0 * s.pitch[0] + (size_t)q.wave_g0 * 48;
    const int vb = ANY ? q.full_lanes * 48 : q.valid_lanes * 48;
    // both rows' global loads are issued before either row goes through the strip
    StripRegs regs[2];
    u32 oc[2][12];      // ragged path: the cut group's own 48 bytes per row
    if constexpr (ANY) {
      strip_fetch_u(regs[0], q.lane, rb, vb);
      strip_fetch_u(regs[1], q.lane, s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
      if (q.cut) {
        packed_group_load(s.p[0] + (size_t)q.row0 * s.pitch[0] + (size_t)q.xs * 3, q.n_px, oc[0]);
        packed_group_load(s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.xs * 3, q.n_px, oc[1]);
      }
    } else {
      strip_fetch(regs[0], q.lane, rb, vb);
      strip_fetch(regs[1], q.lane, s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
    }
    u32 o[12];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      strip_unpack(strip, q.lane, regs[r], o, q.lane_valid && !(ANY && q.cut));
      if constexpr (ANY) {
        if (q.cut) {
#pragma unroll
          for (int i = 0; i < 12; ++i) o[i] = oc[r][i];
        }
      }
      if (q.lane_valid) {
        if constexpr (SRC == K_RGB) deinterleave3(o, b.c0[r], b.c1[r], b.c2[r]);
        else deinterleave3(o, b.c2[r], b.c1[r], b.c0[r]);
      }
    }
    return;
  }
  if (!q.lane_valid)
    return;
  auto ld = [&](const uint8_t* plane, int pitch, int row, u32 (&dst)[4]) {
    const uint4 v = ld16<ANY>(plane + (size_t)row * pitch + (ANY ? q.xs : q.x0), q.n_px);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  };
  ld(s.p[0], s.pitch[0], q.row0, b.c0[0]);
  ld(s.p[0], s.pitch[0], r1, b.c0[1]);
  if constexpr (SRC == K_YUV444 || SRC == K_RGBP) {
    ld(s.p[1], s.pitch[1], q.row0, b.c1[0]); ld(s.p[1], s.pitch[1], r1, b.c1[1]);
    ld(s.p[2], s.pitch[2], q.row0, b.c2[0]); ld(s.p[2], s.pitch[2], r1, b.c2[1]);
  } else if constexpr (SRC == K_NV12) {
    const uint4 uv = ld16<ANY>(s.p[1] + (size_t)q.tile_y * s.pitch[1] + (ANY ? q.xs : q.x0), q.n_px);
    b.cu[0] = __builtin_amdgcn_perm(uv.y, uv.x, 0x06040200u); b.cv[0] = __builtin_amdgcn_perm(uv.y, uv.x, 0x07050301u);
    b.cu[1] = __builtin_amdgcn_perm(uv.w, uv.z, 0x06040200u); b.cv[1] = __builtin_amdgcn_perm(uv.w, uv.z, 0x07050301u);
  } else if constexpr (SRC == K_YUV420) {
    const uint2 u = ld8<ANY>(s.p[1] + (size_t)q.tile_y * s.pitch[1] + (ANY ? q.xs : q.x0) / 2, q.n_px / 2);
    const uint2 v = ld8<ANY>(s.p[2] + (size_t)q.tile_y * s.pitch[2] + (ANY ? q.xs : q.x0) / 2, q.n_px / 2);
    b.cu[0] = u.x; b.cu[1] = u.y; b.cv[0] = v.x; b.cv[1] = v.y;
  }
}

template <int DST, bool ANY>
__device__ __forceinline__ void store_block(const Block& b, const SurfRef& d, const Geo& q, PackedStrip& strip) {
  if constexpr (k_ispacked(DST)) {
    uint8_t* rb = d.p[0] + (size_t)q.row0 * d.pitch[0] + (size_t)q.wave_g0 * 48;
    const int vb = ANY ? q.full_lanes * 48 : q.valid_lanes * 48;
    u32 o[12];
    // (a compile-time trip count: with `r < (has_row1 ? 2 : 1)` the loop stayed rolled once the ragged forms were
    // added, b.c0[r] became a run-time index and the whole Block moved to scratch memory -- 116 bytes per lane and
    // every packed-destination pair at 2.5-4.3 TB/s instead of 5.9-6.2, found late in round 2 by tools/cliffs.py)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r == 1 && !q.has_row1)
        break;
      if (q.lane_valid) {
        if constexpr (DST == K_RGB) interleave3(b.c0[r], b.c1[r], b.c2[r], o);
        else interleave3(b.c2[r], b.c1[r], b.c0[r], o);
      }
      if constexpr (ANY) {
        strip_store_row_u(strip, q.lane, o, q.lane_valid && !q.cut, rb + (size_t)r * d.pitch[0], vb);
        if (q.cut)
          packed_group_store(d.p[0] + (size_t)(q.row0 + r) * d.pitch[0] + (size_t)q.xs * 3, q.n_px, o);
      } else {
        strip_store_row(strip, q.lane, o, q.lane_valid, rb + (size_t)r * d.pitch[0], vb);
      }
    }
    return;
  }
  if (!q.lane_valid)
    return;
  auto st = [&](uint8_t* plane, int pitch, int row, const u32 (&src)[4]) {
    st16<ANY>(plane + (size_t)row * pitch + (ANY ? q.xs : q.x0), make_uint4(src[0], src[1], src[2], src[3]), q.n_px);
  };
  st(d.p[0], d.pitch[0], q.row0, b.c0[0]);
  if (q.has_row1) st(d.p[0], d.pitch[0], q.row0 + 1, b.c0[1]);
  if constexpr (DST == K_YUV444 || DST == K_RGBP) {
    st(d.p[1], d.pitch[1], q.row0, b.c1[0]); st(d.p[2], d.pitch[2], q.row0, b.c2[0]);
    if (q.has_row1) { st(d.p[1], d.pitch[1], q.row0 + 1, b.c1[1]); st(d.p[2], d.pitch[2], q.row0 + 1, b.c2[1]); }
  } else if constexpr (DST == K_NV12) {
    const uint4 uv = make_uint4(__builtin_amdgcn_perm(b.cv[0], b.cu[0], 0x05010400u), __builtin_amdgcn_perm(b.cv[0], b.cu[0], 0x07030602u),
                                __builtin_amdgcn_perm(b.cv[1], b.cu[1], 0x05010400u), __builtin_amdgcn_perm(b.cv[1], b.cu[1], 0x07030602u));
    st16<ANY>(d.p[1] + (size_t)q.tile_y * d.pitch[1] + (ANY ? q.xs : q.x0), uv, q.n_px);
  } else if constexpr (DST == K_YUV420) {
    st8<ANY>(d.p[1] + (size_t)q.tile_y * d.pitch[1] + (ANY ? q.xs : q.x0) / 2, make_uint2(b.cu[0], b.cu[1]), q.n_px / 2);
    st8<ANY>(d.p[2] + (size_t)q.tile_y * d.pitch[2] + (ANY ? q.xs : q.x0) / 2, make_uint2(b.cv[0], b.cv[1]), q.n_px / 2);
  }
}

// in-register transform of the block: SRC channels -> DST channels
template <int SRC, int DST>
__device__ __forceinline__ void transform_block(Block& b, const vali_cvt_params& p) {
  if constexpr (k_isyuv(SRC) && k_isrgb(DST)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (k_is420(SRC)) {
        const u32 cu = b.cu[j >> 1], cv = b.cv[j >> 1];
        const Chroma ca = (j & 1) ? chroma_of(ubyte_f32<2>(cu), ubyte_f32<2>(cv), p.yuv2rgb) : chroma_of(ubyte_f32<0>(cu), ubyte_f32<0>(cv), p.yuv2rgb);
        const Chroma cb = (j & 1) ? chroma_of(ubyte_f32<3>(cu), ubyte_f32<3>(cv), p.yuv2rgb) : chroma_of(ubyte_f32<1>(cu), ubyte_f32<1>(cv), p.yuv2rgb);
        u32 r, g, bb;
        yuv420_to_rgb4(b.c0[0][j], ca, cb, p.yuv2rgb, r, g, bb);
        u32 r1, g1, b1;
        yuv420_to_rgb4(b.c0[1][j], ca, cb, p.yuv2rgb, r1, g1, b1);
        b.c0[0][j] = r; b.c1[0][j] = g; b.c2[0][j] = bb;
        b.c0[1][j] = r1; b.c1[1][j] = g1; b.c2[1][j] = b1;
      } else {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          u32 rr, gg, bb;
          yuv444_to_rgb4(b.c0[r][j], b.c1[r][j], b.c2[r][j], p.yuv2rgb, rr, gg, bb);
          b.c0[r][j] = rr; b.c1[r][j] = gg; b.c2[r][j] = bb;
        }
      }
    }
  } else if constexpr (k_isrgb(SRC) && (k_isyuv(DST) || DST == K_Y)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float us[2][4], vs[2][4];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const u32 R = b.c0[r][j], G = b.c1[r][j], B = b.c2[r][j];
        u32 y = 0, u = 0, v = 0;
#define VALI_PX(I)                                                                          \
  {                                                                                         \
    const float fr = ubyte_f32<I>(R), fg = ubyte_f32<I>(G), fb = ubyte_f32<I>(B);            \
    y = pack_u8<I>(dot_rgb(p.rgb2yuv[0], fr, fg, fb), y);                                    \
    if constexpr (DST != K_Y) {                                                             \
      const float fu = dot_rgb(p.rgb2yuv[1], fr, fg, fb), fv = dot_rgb(p.rgb2yuv[2], fr, fg, fb); \
      if constexpr (k_is420(DST)) { us[r][I] = fu; vs[r][I] = fv; }                          \
      else { u = pack_u8<I>(fu, u); v = pack_u8<I>(fv, v); }                                 \
    }                                                                                       \
  }
        VALI_PX(0) VALI_PX(1) VALI_PX(2) VALI_PX(3)
#undef VALI_PX
        b.c0[r][j] = y;
        if constexpr (DST == K_YUV444) { b.c1[r][j] = u; b.c2[r][j] = v; }
      }
      if constexpr (k_is420(DST)) {
        // 2x2 mean of the un-rounded chroma: ((c00 + c01) + (c10 + c11)) * 0.25
        const float ua = ((us[0][0] + us[0][1]) + (us[1][0] + us[1][1])) * 0.25f;
        const float ub = ((us[0][2] + us[0][3]) + (us[1][2] + us[1][3])) * 0.25f;
        const float va = ((vs[0][0] + vs[0][1]) + (vs[1][0] + vs[1][1])) * 0.25f;
        const float vb = ((vs[0][2] + vs[0][3]) + (vs[1][2] + vs[1][3])) * 0.25f;
        u32& cu = b.cu[j >> 1];
        u32& cv = b.cv[j >> 1];
        if (j & 1) { cu = pack_u8<2>(ua, cu); cu = pack_u8<3>(ub, cu); cv = pack_u8<2>(va, cv); cv = pack_u8<3>(vb, cv); }
        else { cu = pack_u8<0>(ua, 0u); cu = pack_u8<1>(ub, cu); cv = pack_u8<0>(va, 0u); cv = pack_u8<1>(vb, cv); }
        // keep the four 4-pixel groups from being interleaved: 16 live chroma floats per group
        // times four groups is what pushed this branch to 106 VGPRs
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if constexpr (SRC == K_Y && DST == K_YUV444) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 4; ++j) b.c1[r][j] = b.c2[r][j] = 0x80808080u;
  }
  // everything else (NV12<->YUV420, RGB<->RGB_PLANAR, RGB<->BGR, NV12->Y) is a pure
  // relabelling of the loaded channels: load_block / store_block do the permutation.
}

// Packed RGB / BGR -> 4:2:0 (YUV420 / NV12): load + transform fused and ROW-SEQUENTIAL.  Both
// rows' global loads are issued first; then row 0 is unpacked, turned into luma bytes and
// chroma PAIR SUMS, and only then row 1 is unpacked and finishes the 2x2 means.  The generic
// load_block + transform_block keeps both unpacked rows and 32 chroma floats live (106 VGPRs,
// 4 waves per SIMD); this order needs about half of that.  Same operations in the same
// order: ((c00 + c01) + (c10 + c11)) * 0.25.
template <int SRC, int DST, bool ANY>
__device__ __forceinline__ void load_transform_packed_420(Block& b, const SurfRef& s, const Geo& q,
                                                          PackedStrip& strip, const vali_cvt_params& p) {
  static_assert(k_ispacked(SRC) && k_is420(DST), "packed RGB/BGR -> 4:2:0 only");
  const int r1 = q.row0 + (q.has_row1 ? 1 : 0);
  const int vb = ANY ? q.full_lanes * 48 : q.valid_lanes * 48;
  StripRegs regs[2];
  u32 oc[2][12];
  if constexpr (ANY) {
    strip_fetch_u(regs[0], q.lane, s.p[0] + (size_t)q.row0 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
    strip_fetch_u(regs[1], q.lane, s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
    if (q.cut) {
      packed_group_load(s.p[0] + (size_t)q.row0 * s.pitch[0] + (size_t)q.xs * 3, q.n_px, oc[0]);
      packed_group_load(s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.xs * 3, q.n_px, oc[1]);
    }
  } else {
    strip_fetch(regs[0], q.lane, s.p[0] + (size_t)q.row0 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
    strip_fetch(regs[1], q.lane, s.p[0] + (size_t)r1 * s.pitch[0] + (size_t)q.wave_g0 * 48, vb);
  }
  float su[8], sv[8]; // row-0 pair sums of the 8 chroma samples
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    u32 o[12], R[4], G[4], B[4];
    strip_unpack(strip, q.lane, regs[r], o, q.lane_valid && !(ANY && q.cut));
    if constexpr (ANY) {
      if (q.cut) {
#pragma unroll
        for (int i = 0; i < 12; ++i) o[i] = oc[r][i];
      }
    }
    if (!q.lane_valid)
      continue;
    if constexpr (SRC == K_RGB) deinterleave3(o, R, G, B);
    else deinterleave3(o, B, G, R);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 y = 0;
      float fu[4], fv[4];
#define VALI_PX(I)                                                                          \
  {                                                                                         \
    const float fr = ubyte_f32<I>(R[j]), fg = ubyte_f32<I>(G[j]), fb = ubyte_f32<I>(B[j]);   \
    y = pack_u8<I>(dot_rgb(p.rgb2yuv[0], fr, fg, fb), y);                                    \
    fu[I] = dot_rgb(p.rgb2yuv[1], fr, fg, fb);                                              \
    fv[I] = dot_rgb(p.rgb2yuv[2], fr, fg, fb);                                              \
  }
      VALI_PX(0) VALI_PX(1) VALI_PX(2) VALI_PX(3)
#undef VALI_PX
      b.c0[r][j] = y;
      const float ua = fu[0] + fu[1], ub = fu[2] + fu[3], va = fv[0] + fv[1], vb2 = fv[2] + fv[3];
      if (r == 0) {
        su[2 * j] = ua; su[2 * j + 1] = ub; sv[2 * j] = va; sv[2 * j + 1] = vb2;
      } else {
        const float mu0 = (su[2 * j] + ua) * 0.25f, mu1 = (su[2 * j + 1] + ub) * 0.25f;
        const float mv0 = (sv[2 * j] + va) * 0.25f, mv1 = (sv[2 * j + 1] + vb2) * 0.25f;
        u32& cu = b.cu[j >> 1];
        u32& cv = b.cv[j >> 1];
        if (j & 1) { cu = pack_u8<2>(mu0, cu); cu = pack_u8<3>(mu1, cu); cv = pack_u8<2>(mv0, cv); cv = pack_u8<3>(mv1, cv); }
        else { cu = pack_u8<0>(mu0, 0u); cu = pack_u8<1>(mu1, cu); cv = pack_u8<0>(mv0, 0u); cv = pack_u8<1>(mv1, cv); }
      }
      __builtin_amdgcn_sched_barrier(0); // one 4-pixel group at a time
    }
  }
}

template <int K> __device__ __forceinline__ uintptr_t align_bits_of(const SurfRef& s) {
  uintptr_t a = (uintptr_t)s.p[0] | (uintptr_t)s.pitch[0];
  if constexpr (K == K_NV12) a |= (uintptr_t)s.p[1] | (uintptr_t)s.pitch[1];
  if constexpr (K == K_YUV444 || K == K_RGBP) a |= (uintptr_t)s.p[1] | (uintptr_t)s.p[2] | (uintptr_t)s.pitch[1] | (uintptr_t)s.pitch[2];
  if constexpr (K == K_YUV420) a |= (((uintptr_t)s.p[1] | (uintptr_t)s.p[2] | (uintptr_t)s.pitch[1] | (uintptr_t)s.pitch[2]) & 7u) << 1; // 8-B chroma vectors
  return a;
}

template <int SRC, int DST>
__global__ void __launch_bounds__(kBlock) k_cvt8(const CvtArgs a) {
  extern __shared__ uint4 dyn_lds[];
  PackedStrip* const strips = reinterpret_cast<PackedStrip*>(dyn_lds);
  u32 tile_x, tile_y, frame;
  if (!tile_of_block(a.map, tile_x, tile_y, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int W = s.width, H = s.height;
  Geo q;
  q.lane = threadIdx.x & (kWave - 1);
  q.wave = threadIdx.x / kWave;                               // (also the LDS strip index)
  const int wpr = (int)(blockDim.x / kWave) / a.rp;           // waves side by side on one row pair
  q.wave_g0 = (tile_x * wpr + q.wave % wpr) * kWave;
  q.groups = (W + kLanePx - 1) / kLanePx;
  const int crow = tile_y * a.rp + q.wave / wpr;              // row pair of this wave
  if (q.wave_g0 >= q.groups || crow * 2 >= H)
    return;
  q.g = q.wave_g0 + q.lane;
  q.x0 = q.g * kLanePx;
  q.row0 = crow * 2;
  q.tile_y = crow;
  q.has_row1 = q.row0 + 1 < H;
  q.lane_valid = q.g < q.groups;
  q.valid_lanes = min(kWave, q.groups - q.wave_g0);
  q.cut = q.lane_valid && q.x0 + kLanePx > W;
  q.xs = q.cut ? max(W - kLanePx, 0) : q.x0;
  q.n_px = min(kLanePx, W);
  q.full_lanes = min(kWave, W / kLanePx - q.wave_g0);

  // uniform per frame: full 16-pixel groups on 16-byte aligned rows, or the ragged forms of the same accesses
  // (854x480, 1366x768, 1918x1078, tensors with odd strides): a vector body with a byte-granular tail, never
  // a scalar frame
  const bool fast = ((W & (kLanePx - 1)) == 0) && (((align_bits_of<SRC>(s) | align_bits_of<DST>(d)) & 15u) == 0);
  Block b;
  if (fast) {
    if constexpr (k_ispacked(SRC) && k_is420(DST)) {
      load_transform_packed_420<SRC, DST, false>(b, s, q, strips[q.wave], a.p);
    } else {
      load_block<SRC, false>(b, s, q, strips[q.wave]);
      if (q.lane_valid)
        transform_block<SRC, DST>(b, a.p);
    }
    store_block<DST, false>(b, d, q, strips[q.wave]);
  } else {
    if constexpr (k_ispacked(SRC) && k_is420(DST)) {
      load_transform_packed_420<SRC, DST, true>(b, s, q, strips[q.wave], a.p);
    } else {
      load_block<SRC, true>(b, s, q, strips[q.wave]);
      if (q.lane_valid)
        transform_block<SRC, DST>(b, a.p);
    }
    store_block<DST, true>(b, d, q, strips[q.wave]);
  }
}

// ---- element-type kernels ------------------------------------------------------------------
struct ElemArgs {
  const vali_surface* d_src;
  const vali_surface* d_dst;
  vali_surface src, dst;
  TileMap map;
};

// P10/P12 (MSB-aligned u16) -> NV12: round(v / 256) saturated, on the whole W x 1.5H plane.
// A workgroup converts 1024 elements x 16 rows: wave w owns rows 4w..4w+3, and per row a lane
// loads two 16-byte pieces 1 KiB apart (each load instruction of the wave = 1 KiB contiguous)
// and stores two 8-byte pieces 512 B apart; the 8 loads of a lane are issued before any
// arithmetic.  (The first version -- one row x 2048 elements per workgroup, one load per
// thread -- ran at 5.25 TB/s.)
constexpr int kP16TileW = 1024, kP16RowsPerWave = 4, kP16TileH = kWavesPerBlock * kP16RowsPerWave;
__device__ __forceinline__ u32 p16_pair(u32 w) { // two u16 -> two u8 in the low half
  const u32 lo = min(((w & 0xffffu) + 128u) >> 8, 255u), hi = min(((w >> 16) + 128u) >> 8, 255u);
  return lo | (hi << 8);
}
__global__ void __launch_bounds__(kBlock) k_p16_to_nv12(const ElemArgs a) {
  u32 tx, ty, frame;
  if (!tile_of_block(a.map, tx, ty, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int W = s.width, H = s.height, rows = H + H / 2; // luma rows, then the interleaved chroma rows (same width in elements)
  auto srow_of = [&](int y) { return y < H ? s.p[0] + (size_t)y * s.pitch[0] : s.p[1] + (size_t)(y - H) * s.pitch[1]; };
  auto drow_of = [&](int y) { return y < H ? d.p[0] + (size_t)y * d.pitch[0] : d.p[1] + (size_t)(y - H) * d.pitch[1]; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xa = tx * kP16TileW + lane * 8, xb = xa + 512; // the lane's two 8-element groups
  const int y0 = ty * kP16TileH + wave * kP16RowsPerWave;
  if (y0 >= rows || xa >= W)
    return;
  // whole 8-element groups go through the vector path whatever the row alignment (misaligned forms of the same
  // accesses); only a group cut by the right edge takes the element loop
  const bool ga = xa + 8 <= W, gb = xb + 8 <= W;
  uint4 v[kP16RowsPerWave][2];
#pragma unroll
  for (int r = 0; r < kP16RowsPerWave; ++r) {
    const uint8_t* srow = srow_of(min(y0 + r, rows - 1));
    v[r][0] = ga ? load16_u(srow + (size_t)xa * 2) : make_uint4(0, 0, 0, 0);
    v[r][1] = gb ? load16_u(srow + (size_t)xb * 2) : make_uint4(0, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < kP16RowsPerWave; ++r) {
    if (y0 + r >= rows)
      break;
    uint8_t* drow = drow_of(y0 + r);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int x = h ? xb : xa;
      if (h ? gb : ga) {
        const uint4 w = v[r][h];
        typedef v2u32 v2u32_u __attribute__((aligned(1)));
        const v2u32 o = {p16_pair(w.x) | (p16_pair(w.y) << 16), p16_pair(w.z) | (p16_pair(w.w) << 16)};
        *(v2u32_u*)(drow + x) = o;
      } else if (x < W) {
        const u16_u* srow = (const u16_u*)srow_of(y0 + r);
        for (int k = 0; k < 8 && x + k < W; ++k)
          drow[x + k] = (uint8_t)min(((u32)srow[x + k] + 128u) >> 8, 255u);
      }
    }
  }
}

// RGB u8 -> RGB_32F: f = v / 255 (correctly rounded division), 3W elements per row.
// A workgroup converts 4096 consecutive elements of a row in 4 passes of 1024: per pass a
// lane loads one dword (4 elements) and stores one float4, so every store instruction of a
// wave writes 1 KiB contiguous (non-temporal); the 4 loads are issued before the arithmetic.
constexpr int kU8F32Tile = 4096;
__global__ void __launch_bounds__(kBlock) k_rgb8_to_f32(const ElemArgs a) {
  u32 tx, ty, frame;
  if (!tile_of_block(a.map, tx, ty, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int n = s.width * 3, y = ty;
  if (y >= s.height)
    return;
  const uint8_t* srow = s.p[0] + (size_t)y * s.pitch[0];
  uint8_t* drow = d.p[0] + (size_t)y * d.pitch[0];
  const int base = tx * kU8F32Tile + threadIdx.x * 4;
  typedef u32 u32_u __attribute__((aligned(1)));
  typedef v4f32 v4f32_a4 __attribute__((aligned(4)));   // float rows are 4-byte aligned, not more (foreign tensors)
  u32 w[4];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e0 = base + it * 1024;
    w[it] = e0 + 4 <= n ? *(const VALI_GLOBAL u32_u*)(srow + e0) : 0u;
  }
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e0 = base + it * 1024;
    if (e0 >= n)
      break;
    if (e0 + 4 <= n) {
      const u32 q = w[it];
      const v4f32 f = {ubyte_f32<0>(q) / 255.0f, ubyte_f32<1>(q) / 255.0f, ubyte_f32<2>(q) / 255.0f, ubyte_f32<3>(q) / 255.0f};
      __builtin_nontemporal_store(f, (VALI_GLOBAL v4f32_a4*)(drow + (size_t)e0 * 4));
    } else {
      for (int k = 0; k < 4 && e0 + k < n; ++k)
        gstore<float>(drow + (size_t)(e0 + k) * 4, (float)gload<uint8_t>(srow + e0 + k) / 255.0f);
    }
  }
}

// RGB_32F packed -> RGB_32F_PLANAR: lane = 4 pixels (48 B in, 3 x 16 B out) x 2 rows.
// A wave's 256 pixels of a row are 3 KiB contiguous: they are loaded as 3 x 1 KiB coalesced
// pieces (lane l: bytes 16 l of each KiB), pass through the wave's LDS strip, and come back as
// the lane's own 48 bytes (ds_read_b128 at a 48-byte lane stride is conflict-free) -- instead
// of three loads that each touch 16 bytes in every 48.  Both rows' loads are issued first.
struct alignas(16) F32Strip {
  uint8_t b[3 * 1024];
};
__global__ void __launch_bounds__(kBlock) k_f32_deinterleave(const ElemArgs a) {
  __shared__ F32Strip strips[kWavesPerBlock];
  u32 tx, ty, frame;
  if (!tile_of_block(a.map, tx, ty, frame))
    return;
  const SurfRef s = load_surface(a.d_src, a.src, frame);
  const SurfRef d = load_surface(a.d_dst, a.dst, frame);
  const int W = s.width, H = s.height;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xw = (tx * kWavesPerBlock + wave) * 256; // first pixel of the wave
  const int x0 = xw + lane * 4, y0 = ty * 2;
  if (xw >= W || y0 >= H)
    return;
  typedef v4u32 v4u32_a4 __attribute__((aligned(4)));   // float rows: 4-byte aligned is all a foreign tensor promises
  typedef v4f32 v4f32_a4 __attribute__((aligned(4)));
  if (xw + 256 <= W) {
    F32Strip& st = strips[wave];
    const int rows = min(2, H - y0);
    uint4 v[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const uint8_t* srow = s.p[0] + (size_t)min(y0 + r, H - 1) * s.pitch[0] + (size_t)xw * 12;
#pragma unroll
      for (int i = 0; i < 3; ++i)
      {
        const v4u32 t = *(const v4u32_a4*)(srow + i * 1024 + lane * 16);
        v[r][i] = make_uint4(t.x, t.y, t.z, t.w);
      }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (r >= rows)
        break;
#pragma unroll
      for (int i = 0; i < 3; ++i)
        *reinterpret_cast<uint4*>(&st.b[i * 1024 + lane * 16]) = v[r][i];
      wave_lds_sync();
      const float4 a0 = *reinterpret_cast<const float4*>(&st.b[lane * 48]);
      const float4 a1 = *reinterpret_cast<const float4*>(&st.b[lane * 48 + 16]);
      const float4 a2 = *reinterpret_cast<const float4*>(&st.b[lane * 48 + 32]);
      wave_lds_sync();
      const int y = y0 + r;
      const v4f32 c0 = {a0.x, a0.w, a1.z, a2.y}, c1 = {a0.y, a1.x, a1.w, a2.z}, c2 = {a0.z, a1.y, a2.x, a2.w};
      *(VALI_GLOBAL v4f32_a4*)(d.p[0] + (size_t)y * d.pitch[0] + (size_t)x0 * 4) = c0;
      *(VALI_GLOBAL v4f32_a4*)(d.p[1] + (size_t)y * d.pitch[1] + (size_t)x0 * 4) = c1;
      *(VALI_GLOBAL v4f32_a4*)(d.p[2] + (size_t)y * d.pitch[2] + (size_t)x0 * 4) = c2;
    }
    return;
  }
  // ragged last chunk of a row (fewer than 256 pixels left): per-lane vectors, elements for the cut group
  if (x0 >= W)
    return;
  for (int r = 0; r < 2 && y0 + r < H; ++r) {
    const int y = y0 + r;
    const float* srow = (const float*)(s.p[0] + (size_t)y * s.pitch[0]) + (size_t)x0 * 3;
    float* o0 = (float*)(d.p[0] + (size_t)y * d.pitch[0]) + x0;
    float* o1 = (float*)(d.p[1] + (size_t)y * d.pitch[1]) + x0;
    float* o2 = (float*)(d.p[2] + (size_t)y * d.pitch[2]) + x0;
    if (x0 + 4 <= W) {
      const v4f32 a0 = ((const v4f32_a4*)srow)[0], a1 = ((const v4f32_a4*)srow)[1], a2 = ((const v4f32_a4*)srow)[2];
      const v4f32 c0 = {a0.x, a0.w, a1.z, a2.y}, c1 = {a0.y, a1.x, a1.w, a2.z}, c2 = {a0.z, a1.y, a2.x, a2.w};
      *(v4f32_a4*)o0 = c0;
      *(v4f32_a4*)o1 = c1;
      *(v4f32_a4*)o2 = c2;
    } else {
      for (int k = 0; k < 4 && x0 + k < W; ++k) { o0[k] = srow[3 * k]; o1[k] = srow[3 * k + 1]; o2[k] = srow[3 * k + 2]; }
    }
  }
}

// ---- host side -----------------------------------------------------------------------------
static int kind_of(int fmt) {
  switch (fmt) {
  case VALI_FMT_NV12: return K_NV12;
  case VALI_FMT_YUV420: return K_YUV420;
  case VALI_FMT_YUV444: return K_YUV444;
  case VALI_FMT_RGB: return K_RGB;
  case VALI_FMT_BGR: return K_BGR;
  case VALI_FMT_RGB_PLANAR: return K_RGBP;
  case VALI_FMT_Y: return K_Y;
  default: return K_NONE;
  }
}

template <int SRC, int DST>
static void launch_cvt8(const CvtArgs& a, dim3 grid, int block, unsigned lds, hipStream_t s) {
  hipLaunchKernelGGL((k_cvt8<SRC, DST>), grid, dim3(block), lds, s, a);
}

static int launch_convert(CvtArgs& a, ElemArgs& e, int src_fmt, int dst_fmt, int width, int height, int n,
                          hipStream_t stream) {
  // element-type conversions first
  if ((src_fmt == VALI_FMT_P10 || src_fmt == VALI_FMT_P12) && dst_fmt == VALI_FMT_NV12) {
    if ((width | height) & 1)
      return fail(VALI_ERR_INVALID_ARG, "convert: 4:2:0 surfaces need even width and height (%dx%d)", width, height);
    e.map = make_tile_map((width + kP16TileW - 1) / kP16TileW, (height + (height + 1) / 2 + kP16TileH - 1) / kP16TileH, (u32)n);
    hipLaunchKernelGGL(k_p16_to_nv12, tile_grid(e.map), dim3(kBlock), 0, stream, e);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (src_fmt == VALI_FMT_RGB && dst_fmt == VALI_FMT_RGB_32F) {
    e.map = make_tile_map((width * 3 + kU8F32Tile - 1) / kU8F32Tile, height, (u32)n);
    hipLaunchKernelGGL(k_rgb8_to_f32, tile_grid(e.map), dim3(kBlock), 0, stream, e);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  if (src_fmt == VALI_FMT_RGB_32F && dst_fmt == VALI_FMT_RGB_32F_PLANAR) {
    e.map = make_tile_map((width + kBlock * 4 - 1) / (kBlock * 4), (height + 1) / 2, (u32)n);
    hipLaunchKernelGGL(k_f32_deinterleave, tile_grid(e.map), dim3(kBlock), 0, stream, e);
    VALI_LAUNCH_CHECK();
    return VALI_OK;
  }
  const int sk = kind_of(src_fmt), dk = kind_of(dst_fmt);
  if (sk == K_NONE || dk == K_NONE)
    return fail(VALI_ERR_UNSUPPORTED, "convert: unsupported pair %d -> %d", src_fmt, dst_fmt);
  // 4:2:0 planes hold (W/2) x (H/2) chroma samples (Surfaces.cpp:231-246): an odd size would put the last
  // row pair's chroma outside the plane
  if ((k_is420(sk) || k_is420(dk)) && ((width | height) & 1))
    return fail(VALI_ERR_INVALID_ARG, "convert: 4:2:0 surfaces need even width and height (%dx%d)", width, height);
  const int groups = (width + kLanePx - 1) / kLanePx;
  int block = ((groups + kWave - 1) / kWave) * kWave;
  if (block > kBlock)
    block = kBlock;
  // narrow frames stack row pairs into a 256-thread workgroup (profiles/r01_variants.md sweep 8)
  const int rp_env = tuning(VALI_TUNE_NV12_ROWPAIRS);
  const int row_block = block;
  a.rp = rp_env > 0 ? rp_env : kBlock / row_block;
  if (a.rp < 1 || row_block * a.rp > kBlock)
    a.rp = 1;
  a.map = make_tile_map((groups + row_block - 1) / row_block, ((height + 1) / 2 + a.rp - 1) / a.rp, (u32)n);
  block = row_block * a.rp;
  const dim3 grid = tile_grid(a.map);
  // residency cap: 16 waves/CU measured best for plane -> packed streams (profiles/r01_variants.md);
  // VALI_TUNE_WAVES_PER_CU is an A/B switch
  // Packed SOURCES (global -> LDS strip -> registers before any arithmetic) have a longer
  // dependent chain per wave and want 24 waves/CU (RGB->RGB_PLANAR 5.79 -> 5.98, RGB->YUV444
  // 4.78 -> 5.38, RGB->Y 4.76 -> 5.71 TB/s); everything else keeps 16.
  const int waves_override = tuning(VALI_TUNE_WAVES_PER_CU);
  const int waves_per_cu = waves_override > 0 ? waves_override : streaming_waves_per_cu(groups, row_block, k_ispacked(sk) ? 24 : 16);
  const unsigned lds = residency_lds_bytes(block, waves_per_cu, (unsigned)sizeof(PackedStrip) * kWavesPerBlock);
#define VALI_PAIR(S, D)                                                                     \
  if (sk == S && dk == D) {                                                                 \
    launch_cvt8<S, D>(a, grid, block, lds, stream);                                          \
    VALI_LAUNCH_CHECK();                                                                    \
    return VALI_OK;                                                                         \
  }
  // the pairs of ConvertSurface::GetSupportedConversions() (TaskConvertSurface.cpp:966-994)
  VALI_PAIR(K_NV12, K_YUV420) VALI_PAIR(K_YUV420, K_NV12) VALI_PAIR(K_NV12, K_RGB) VALI_PAIR(K_NV12, K_BGR)
  VALI_PAIR(K_NV12, K_RGBP) VALI_PAIR(K_RGB, K_RGBP) VALI_PAIR(K_RGBP, K_RGB) VALI_PAIR(K_RGBP, K_YUV444)
  VALI_PAIR(K_Y, K_YUV444) VALI_PAIR(K_YUV420, K_RGB) VALI_PAIR(K_RGB, K_YUV420) VALI_PAIR(K_RGB, K_YUV444)
  VALI_PAIR(K_RGB, K_BGR) VALI_PAIR(K_BGR, K_RGB) VALI_PAIR(K_YUV420, K_BGR) VALI_PAIR(K_YUV444, K_BGR)
  VALI_PAIR(K_YUV444, K_RGB) VALI_PAIR(K_BGR, K_YUV444) VALI_PAIR(K_NV12, K_Y) VALI_PAIR(K_RGB, K_Y)
  // natural extras sharing the same code
  VALI_PAIR(K_YUV444, K_RGBP) VALI_PAIR(K_YUV420, K_RGBP) VALI_PAIR(K_BGR, K_YUV420) VALI_PAIR(K_RGB, K_NV12)
  VALI_PAIR(K_BGR, K_NV12) VALI_PAIR(K_BGR, K_Y) VALI_PAIR(K_RGBP, K_Y) VALI_PAIR(K_RGBP, K_YUV420)
#undef VALI_PAIR
  return fail(VALI_ERR_UNSUPPORTED, "convert: unsupported pair %d -> %d", src_fmt, dst_fmt);
}

} // namespace vali

using namespace vali;

extern "C" {

int vali_convert(const vali_surface* src, const vali_surface* dst, const vali_cvt_params* params,
                 vali_stream_t stream) {
  VALI_REQUIRE(src && dst && params, "null argument");
  VALI_REQUIRE(src->width > 0 && src->height > 0, "empty src");
  VALI_REQUIRE(src->width == dst->width && src->height == dst->height, "src/dst size mismatch");
  VALI_REQUIRE(src->plane[0] && dst->plane[0], "null plane");
  CvtArgs a = {};
  a.src = *src; a.dst = *dst; a.p = *params;
  ElemArgs e = {};
  e.src = *src; e.dst = *dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_convert(a, e, src->format, dst->format, src->width, src->height, 1, s);
}

int vali_convert_batch(const vali_surface* d_src, const vali_surface* d_dst, int n, int src_format,
                       int dst_format, int width, int height, const vali_cvt_params* params,
                       vali_stream_t stream) {
  VALI_REQUIRE(d_src && d_dst && params, "null argument");
  VALI_REQUIRE(width > 0 && height > 0, "empty geometry");
  VALI_REQUIRE(n >= 0 && n <= 65535, "batch size out of range (0..65535)");
  if (n == 0)
    return VALI_OK;
  CvtArgs a = {};
  a.d_src = d_src; a.d_dst = d_dst; a.p = *params;
  ElemArgs e = {};
  e.d_src = d_src; e.d_dst = d_dst;
  hipStream_t s = as_stream(stream);
  VALI_ENTRY(s);
  return launch_convert(a, e, src_format, dst_format, width, height, n, s);
}

} // extern "C"
