// Device-side building blocks shared by the colour/geometry kernels (gfx950).
//
// Conventions used by every kernel in this library:
//  * a lane owns 16 horizontally adjacent pixels (one 16-byte luma vector), so
//    every global access of a u8 plane is a dwordx4 and a full wave touches
//    1 KiB contiguous per instruction;
//  * packed 3-byte pixels are staged through a per-wave LDS strip so that the
//    wave's 64 x 48 B = 3 KiB of a row leave (or enter) the CU as three fully
//    contiguous 1 KiB dwordx4 instructions instead of 64 strided 16 B pieces;
//  * no MFMA: these are HBM-bound per-pixel ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/vali_hip.h"

namespace vali {

typedef uint32_t u32;

constexpr int kLanePx = 16;   // pixels per lane per row
constexpr int kWave = 64;     // gfx950 wavefront
constexpr int kBlock = 256;   // default workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

// byte i (0..3) of a dword as float: lowers to v_cvt_f32_ubyte{i}
template <int I> __device__ __forceinline__ float ubyte_f32(u32 w) {
  return (float)((w >> (8 * I)) & 0xffu);
}

template <int I> __device__ __forceinline__ u32 ubyte(u32 w) {
  return (w >> (8 * I)) & 0xffu;
}

// The float -> u8 quantiser of every colour kernel: round to nearest even,
// saturate to [0,255] (oracle: vali_q_u8 in oracle/vali_oracle.c).
// v_cvt_pk_u8_f32 does exactly that in one instruction and also merges the
// byte into `old`; tests/test_gpu_quantizer.py pins the instruction against
// the portable form over a dense sweep of inputs.
#ifndef VALI_USE_CVT_PK_U8
#define VALI_USE_CVT_PK_U8 1
#endif

__device__ __forceinline__ u32 quantize_u8_portable(float v) {
  float r = __builtin_rintf(v);
  r = __builtin_fminf(__builtin_fmaxf(r, 0.0f), 255.0f);
  return (u32)r;
}

template <int SEL> __device__ __forceinline__ u32 pack_u8(float v, u32 old) {
#if VALI_USE_CVT_PK_U8
  return __builtin_amdgcn_cvt_pk_u8_f32(v, SEL, old);
#else
  return old | (quantize_u8_portable(v) << (8 * SEL));
#endif
}

__device__ __forceinline__ u32 quantize_u8(float v) {
#if VALI_USE_CVT_PK_U8
  return __builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0u);
#else
  return quantize_u8_portable(v);
#endif
}

// ---- arithmetic shared by several kernels (each restated in oracle/vali_oracle*.c) -----
// YUV -> RGB in the vali_csc form: R = cy*(Y-y0) + crv*V', G = cy*(Y-y0) + (cgu*U' + cgv*V'),
// B = cy*(Y-y0) + cbu*U'  with U' = U-128, V' = V-128 (oracle: vali_oracle_nv12_to_rgb)
struct ChromaTerm {
  float rv, guv, bu;
};

__device__ __forceinline__ ChromaTerm chroma_term(float u, float v, const vali_csc& k) {
  const float uc = u - 128.0f, vc = v - 128.0f;
  ChromaTerm t;
  t.rv = k.crv * vc;
  t.guv = __builtin_fmaf(k.cgu, uc, k.cgv * vc);
  t.bu = k.cbu * uc;
  return t;
}

__device__ __forceinline__ float luma_term(float y, const vali_csc& k) {
  return k.cy * (y - k.y0);
}

// bilinear tap on the resize grid f = x * scale (oracle: make_lerp in vali_oracle.c)
struct Lerp {
  int i0, i1;
  float a;
};

__device__ __forceinline__ Lerp make_lerp(int x, float scale, int size) {
  const float f = (float)x * scale;
  const float fl = __builtin_floorf(f);
  Lerp l;
  l.a = f - fl;
  const int i = (int)fl;
  l.i0 = min(i, size - 1);
  l.i1 = min(i + 1, size - 1);
  return l;
}

// 16-byte global accesses.  Stores of finished output use the non-temporal form:
// the surface kernels never re-read what they write, and streaming full 128-byte
// lines past the L2 measured +2..3% on the 1:2 read:write mix (profiles/r01_variants.md).
// NEVER use it for partial-line (strided 16 B) stores: that measured 2.4x slower.
//
// Plane pointers reach the kernels through descriptors in memory, so the compiler only knows
// them as generic pointers and would emit flat_load/flat_store (which also tick the LDS
// counter and serialise against ds_* traffic).  Every surface lives in global memory: the
// helpers below cast to address space 1 so the ISA is global_load/global_store.
#define VALI_GLOBAL __attribute__((address_space(1)))
typedef unsigned v4u32 __attribute__((ext_vector_type(4)));
typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T gload(const void* p) {
  return *(const VALI_GLOBAL T*)p;
}
template <typename T> __device__ __forceinline__ void gstore(void* p, T v) {
  *(VALI_GLOBAL T*)p = v;
}
// any byte alignment: the same instruction on an under-aligned type (gfx950 runs HSA queues in unaligned-access mode;
// a misaligned wave access touches at most one more line)
template <typename T> __device__ __forceinline__ T gload_u(const void* p) {
  typedef T TU __attribute__((aligned(1)));
  return *(const VALI_GLOBAL TU*)p;
}
template <typename T> __device__ __forceinline__ void gstore_u(void* p, T v) {
  typedef T TU __attribute__((aligned(1)));
  *(VALI_GLOBAL TU*)p = v;
}
// non-temporal: finished output whose wave instruction covers whole 128-byte lines (neighbouring lanes write
// neighbouring bytes); never for strided pieces
template <typename T> __device__ __forceinline__ void gstore_nt(void* p, T v) {
  __builtin_nontemporal_store(v, (VALI_GLOBAL T*)p);
}
template <typename T> __device__ __forceinline__ void gstore_u_nt(void* p, T v) { // ... from an address of any alignment
  typedef T TU __attribute__((aligned(1)));
  __builtin_nontemporal_store(v, (VALI_GLOBAL TU*)p);
}

// The one-shot streaming converters (cvt_nv12_rgb.hip, cvt_generic.hip: every thread loads, converts, stores, no loop)
// keep GENERIC pointers for their 16-byte accesses: an A/B on NV12->RGB 2160p measured flat_load/flat_store 1.2%
// FASTER than the global_* forms (6.18 vs 6.11 TB/s, 3 interleaved repetitions, profiles/r01_variants.md).
// NEVER in a kernel that keeps loads in flight across a loop: while a flat access is pending the compiler must treat
// vmcnt as out of order and turns every wait into vmcnt(0) -- one flat_store per row of the exact-2x UD kernel
// drained its whole prefetch (round 2).  Those kernels use gload16 / gstore16 and the VALI_GLOBAL forms only.
__device__ __forceinline__ void store16_nt(void* p, uint4 v) {
  const v4u32 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, (v4u32*)p);
}
__device__ __forceinline__ void store16(void* p, uint4 v) { *(uint4*)p = v; }
__device__ __forceinline__ uint4 load16(const void* p) { return *(const uint4*)p; }
// Forms for rows that are NOT 16-byte aligned (foreign pitches, odd base pointers, ragged widths): the same
// dwordx4 instruction on an under-aligned type -- gfx950 runs HSA queues in unaligned-access mode, a
// misaligned wave access only touches one more 128-byte line.  load_bytes16 / store_bytes16 move n < 16
// bytes one by one (frames narrower than one 16-pixel group only -- a byte access costs the memory pipeline
// as much as a 16-byte one, so wider frames never use them: their cut group slides left instead); fully
// unrolled, so the dword array stays in registers.  No form ever touches a byte outside the row.
typedef v4u32 v4u32_u __attribute__((aligned(1)));
typedef unsigned short u16_u __attribute__((aligned(1)));
__device__ __forceinline__ uint4 load16_u(const void* p) {
  const v4u32 w = *(const v4u32_u*)p;
  return make_uint4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ void store16_u(void* p, uint4 v) {
  const v4u32 w = {v.x, v.y, v.z, v.w};
  *(v4u32_u*)p = w;
}
__device__ __forceinline__ uint4 load_bytes16(const uint8_t* p, int n) {
  u32 w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < n)
      w[k / 4] |= (u32)p[k] << (8 * (k % 4));
  return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void store_bytes16(uint8_t* p, uint4 v, int n) {
  const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (k < n)
      p[k] = (uint8_t)(w[k / 4] >> (8 * (k % 4)));
}
// n valid bytes of a 16-byte group: whole group -> one (possibly misaligned) vector access
__device__ __forceinline__ uint4 load16_n(const uint8_t* p, int n) { return n >= 16 ? load16_u(p) : load_bytes16(p, n); } // n <= 0: zeros
__device__ __forceinline__ void store16_n(uint8_t* p, uint4 v, int n) {
  if (n >= 16) store16_u(p, v);
  else store_bytes16(p, v, n);
}
// typed (global address space) forms for the gather kernels
__device__ __forceinline__ uint4 gload16(const void* p) {
  const v4u32 w = *(const VALI_GLOBAL v4u32*)p;
  return make_uint4(w.x, w.y, w.z, w.w);
}
// rows that are read ONCE (downscales of 2x and more): the non-temporal form measured 3 % on the bilinear resizer
__device__ __forceinline__ uint4 gload16_nt(const void* p) {
  const v4u32 w = __builtin_nontemporal_load((const VALI_GLOBAL v4u32*)p);
  return make_uint4(w.x, w.y, w.z, w.w);
}
__device__ __forceinline__ void gstore16(void* p, uint4 v) {
  const v4u32 w = {v.x, v.y, v.z, v.w};
  *(VALI_GLOBAL v4u32*)p = w;
}
__device__ __forceinline__ void gstore16_nt(void* p, uint4 v) { // whole 128-byte lines per wave instruction only
  const v4u32 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, (VALI_GLOBAL v4u32*)p);
}
__device__ __forceinline__ uint2 load8(const void* p) {
  const v2u32 w = *(const VALI_GLOBAL v2u32*)p;
  return make_uint2(w.x, w.y);
}
__device__ __forceinline__ void store8(void* p, uint2 v) {
  const v2u32 w = {v.x, v.y};
  *(VALI_GLOBAL v2u32*)p = w;
}
__device__ __forceinline__ void store8_nt(void* p, uint2 v) {
  const v2u32 w = {v.x, v.y};
  __builtin_nontemporal_store(w, (VALI_GLOBAL v2u32*)p);
}
__device__ __forceinline__ void store16f(void* p, float4 v) {
  const v4f32 w = {v.x, v.y, v.z, v.w};
  *(VALI_GLOBAL v4f32*)p = w;
}
// non-temporal: only where a wave instruction writes whole 128-byte lines
__device__ __forceinline__ void store16f_nt(void* p, float4 v) {
  const v4f32 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, (VALI_GLOBAL v4f32*)p);
}
__device__ __forceinline__ float4 load16f(const void* p) {
  const v4f32 w = *(const VALI_GLOBAL v4f32*)p;
  return make_float4(w.x, w.y, w.z, w.w);
}

// ---------------------------------------------------------------------------
// XCD-aware tile map.
//
// The dispatcher places workgroup b of a launch on XCD b % 8 (observed, never relied
// on for correctness -- any placement gives the same result).  A streaming kernel
// whose consecutive workgroups walk consecutive rows therefore interleaves the rows of
// a frame over the 8 private L2s.  Giving each XCD one CONTIGUOUS eighth of the frame's
// tiles instead measured +3.5% on NV12->RGB 2160p (DRAM page locality of each L2's
// fill/evict stream), +7% together with nt stores (profiles/r01_variants.md).
// Over a BATCH the contiguous share is taken over the whole tile list of the launch
// (frame-major): XCD k owns tiles [k*T/8, (k+1)*T/8), i.e. whole consecutive frames.  That
// measured another +6% (6.20 -> 6.57 TB/s, 82% of the HBM peak; sweep 5 in
// profiles/r01_variants.md); a single frame degenerates to the per-frame split above.
//   launch: grid.x = 8 * per_xcd, per_xcd = ceil(total / 8), total = frames * per_frame
//   t      = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8   (skip if >= total)
//   frame  = t / per_frame ; tile = t % per_frame
// ---------------------------------------------------------------------------
struct TileMap {
  u32 total;      // tiles of the whole launch (= frames * per_frame)
  u32 per_xcd;    // ceil(total / 8)
  u32 tiles_x;    // tiles along x of one frame
  u32 per_frame;  // tiles of one frame
};

__host__ __device__ inline TileMap make_tile_map(u32 tiles_x, u32 tiles_y, u32 frames = 1u) {
  TileMap m;
  m.per_frame = tiles_x * tiles_y;
  m.total = m.per_frame * frames;
  m.per_xcd = (m.total + 7u) / 8u;
  m.tiles_x = tiles_x;
  return m;
}

// same for a tile list whose length per frame is already known (plane-job kernels)
__host__ __device__ inline TileMap make_tile_map_linear(u32 per_frame, u32 frames) {
  TileMap m;
  m.per_frame = per_frame;
  m.total = per_frame * frames;
  m.per_xcd = (m.total + 7u) / 8u;
  m.tiles_x = 1u;
  return m;
}

inline dim3 tile_grid(const TileMap& m) { return dim3(m.per_xcd * 8u); }

// linear tile of this workgroup -> (frame, tile within the frame); false for grid padding
__device__ __forceinline__ bool frame_tile_of_block(const TileMap& m, u32& frame, u32& tile) {
  const u32 b = blockIdx.x;
  const u32 t = (b & 7u) * m.per_xcd + (b >> 3);
  if (t >= m.total)
    return false;
  if (m.total == m.per_frame) {
    frame = 0;
    tile = t;
  } else {
    frame = t / m.per_frame;
    tile = t - frame * m.per_frame;
  }
  return true;
}

// (frame, tile_x, tile_y) of this workgroup; false for grid padding
__device__ __forceinline__ bool tile_of_block(const TileMap& m, u32& tx, u32& ty, u32& frame) {
  u32 t;
  if (!frame_tile_of_block(m, frame, t))
    return false;
  if (m.tiles_x == 1u) {
    tx = 0;
    ty = t;
  } else {
    ty = t / m.tiles_x;
    tx = t - ty * m.tiles_x;
  }
  return true;
}

// Device-side copy of a vali_surface descriptor: from the batch's device array when
// there is one, from the kernel argument otherwise (explicit branches keep the two
// address spaces apart -- a pointer select between them costs scratch + flat loads).
struct SurfRef {
  uint8_t* p[3];
  int pitch[3];
  int width, height;
};

__device__ __forceinline__ SurfRef load_surface(const vali_surface* arr, const vali_surface& one,
                                                u32 index) {
  SurfRef r;
  if (arr) {
    const vali_surface* s = arr + index;
    r.p[0] = (uint8_t*)s->plane[0]; r.p[1] = (uint8_t*)s->plane[1]; r.p[2] = (uint8_t*)s->plane[2];
    r.pitch[0] = s->pitch[0]; r.pitch[1] = s->pitch[1]; r.pitch[2] = s->pitch[2];
    r.width = s->width; r.height = s->height;
  } else {
    r.p[0] = (uint8_t*)one.plane[0]; r.p[1] = (uint8_t*)one.plane[1]; r.p[2] = (uint8_t*)one.plane[2];
    r.pitch[0] = one.pitch[0]; r.pitch[1] = one.pitch[1]; r.pitch[2] = one.pitch[2];
    r.width = one.width; r.height = one.height;
  }
  return r;
}

// ---------------------------------------------------------------------------
// Plane jobs: kernels that treat every plane of a surface as an independent image
// (resize, rotate) concatenate the planes' tiles into one tile list per frame.
// A job names the component and its subsampling; for single-frame launches the host also
// resolves the plane pointers into the job (sp/dp), so the device never indexes a
// register-resident descriptor with a run-time component number (that sends the whole
// descriptor to scratch memory).  Batched launches read arr[frame].plane[comp] from the
// device descriptor array with an ordinary dynamically-indexed load.
// ---------------------------------------------------------------------------
struct PlaneJob {
  int comp;           // component index in vali_surface.plane[] / pitch[]
  int sub_x, sub_y;   // log2 subsampling of the DESTINATION plane relative to the surface size
  int ssub_x, ssub_y; // the same for the SOURCE plane (differs only for UDPlanar: YUV420 -> YUV444)
  int channels;       // interleaved channels in the plane
  int kind;           // 0: the kernel's own operation; 1: a plane of unchanged size that the kernel copies (plane_copy_tile:
                      // UDPlanar's luma rides in the launch that doubles its chroma planes)
  u32 first_tile, tiles_x;
  float shift_x, shift_y; // rotate only
  const uint8_t* sp;  // single-frame launches: resolved by the host
  uint8_t* dp;
  int spitch, dpitch;
  const float4* xtab; // Lanczos / bicubic resize: tap tables of the plane's axes (tap_table.hip), or null
  const float4* ytab;
};

struct PlaneView {
  const uint8_t* sp;
  uint8_t* dp;
  int spitch, dpitch, sw, sh, dw, dh;
};

// sw/sh/dw/dh: the single-frame SURFACE sizes (ignored for batches)
__device__ __forceinline__ PlaneView plane_view(const vali_surface* d_src, const vali_surface* d_dst,
                                                u32 frame, const PlaneJob& job, int sw, int sh,
                                                int dw, int dh) {
  PlaneView v;
  if (d_src) {
    const vali_surface* s = d_src + frame;
    const vali_surface* d = d_dst + frame;
    v.sp = (const uint8_t*)s->plane[job.comp];
    v.dp = (uint8_t*)d->plane[job.comp];
    v.spitch = s->pitch[job.comp];
    v.dpitch = d->pitch[job.comp];
    sw = s->width; sh = s->height; dw = d->width; dh = d->height;
  } else {
    v.sp = job.sp; v.dp = job.dp; v.spitch = job.spitch; v.dpitch = job.dpitch;
  }
  v.sw = sw >> job.ssub_x; v.sh = sh >> job.ssub_y;
  v.dw = dw >> job.sub_x; v.dh = dh >> job.sub_y;
  return v;
}

// tile index of this workgroup -> (job, tile_x, tile_y) through the XCD-contiguous map;
// false for grid padding.  Conditional copies instead of jobs[j] (no dynamic indexing).
__device__ __forceinline__ bool plane_tile(const PlaneJob (&jobs)[3], int njobs, const TileMap& map,
                                           PlaneJob& job, u32& tx, u32& ty, u32& frame) {
  u32 t;
  if (!frame_tile_of_block(map, frame, t))
    return false;
  job = jobs[0];
  if (njobs > 1 && t >= jobs[1].first_tile) job = jobs[1];
  if (njobs > 2 && t >= jobs[2].first_tile) job = jobs[2];
  const u32 local = t - job.first_tile;
  ty = local / job.tiles_x;
  tx = local - ty * job.tiles_x;
  return true;
}

// LDS layout of a staged source row (bilinear resize, general UD): 4 bytes of padding after every 128 bytes of row
// data.  At the common 2x / 4x downscales a lane's taps sit 8 / 16 / 32 bytes from its neighbour's, so 32 lanes of a
// byte read hit only 16 / 8 / 4 of the 32 banks (the unpadded strip of the bilinear resizer spent 52 % of its LDS
// cycles in bank conflicts); the pad moves each following 128-byte group one bank on.  Applied when a pixel never
// straddles a 128-byte group (pixel size 1, 2, 4, 8, 16 bytes); packed RGB keeps the plain layout.
template <int PB> constexpr bool kStagePadded = (128 % PB) == 0;
template <bool PAD> __device__ __forceinline__ int stage_off(int a) { return PAD ? a + ((a >> 7) << 2) : a; }
constexpr int stage_alloc(int row_bytes) { return row_bytes + 4 * (row_bytes / 128); }
// one 16-byte chunk (chunk index k of the row) -> the strip: 4-byte aligned once padded, so dword writes, which the
// pad also keeps conflict-free
template <bool PAD> __device__ __forceinline__ void stage_put(uint8_t* row, int k, const uint4& q) {
  if constexpr (PAD) {
    u32* w = reinterpret_cast<u32*>(row + k * 16 + ((k >> 3) << 2));
    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
  } else {
    *reinterpret_cast<uint4*>(row + k * 16) = q;
  }
}

// Order LDS traffic of ONE wave: DS instructions of a wave execute in issue
// order, so a compiler-level fence is all that is needed between the strided
// writes and the transposed reads of the same wave-private strip.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// Packed-3-channel row strip (u8): 64 lanes x 48 B.
//
// store: lane l owns bytes [48l, 48l+48) of the strip (its 16 RGB pixels) as
// three uint4; the wave writes them to global memory as 3 x 1 KiB contiguous
// pieces.  `valid_bytes` (multiple of 16) trims the last wave of a row.
// LDS banking: ds_write_b128 is serviced in 8-lane groups; at a 48 B lane
// stride the 8 lanes of a group cover 8 distinct 4-bank slots -> conflict free.
// ds_read_b128 of consecutive uint4 is conflict free by construction.
// ---------------------------------------------------------------------------
struct alignas(16) PackedStrip {
  uint4 v[kWave * 3];
};

__device__ __forceinline__ void strip_store_row(PackedStrip& strip, int lane,
                                                const u32 (&o)[12], bool lane_valid,
                                                uint8_t* row_base, int valid_bytes) {
  if (lane_valid) {
    strip.v[lane * 3 + 0] = make_uint4(o[0], o[1], o[2], o[3]);
    strip.v[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    strip.v[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]);
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k * kWave + lane) * 16;
    if (off < valid_bytes)
      store16_nt(row_base + off, strip.v[k * kWave + lane]);
  }
  wave_lds_sync();
}

// The same for rows of ANY alignment (misaligned forms of the 16-byte stores; plain, not non-temporal: a
// misaligned wave store leaves partial lines at both ends).  valid_bytes is still a multiple of 48: the group
// a ragged width cuts does not go through the strip (see the ragged paths of the converters).
__device__ __forceinline__ void strip_store_row_u(PackedStrip& strip, int lane, const u32 (&o)[12], bool lane_valid,
                                                  uint8_t* row_base, int valid_bytes) {
  if (lane_valid) {
    strip.v[lane * 3 + 0] = make_uint4(o[0], o[1], o[2], o[3]);
    strip.v[lane * 3 + 1] = make_uint4(o[4], o[5], o[6], o[7]);
    strip.v[lane * 3 + 2] = make_uint4(o[8], o[9], o[10], o[11]);
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k * kWave + lane) * 16;
    if (off < valid_bytes)
      store16_u(row_base + off, strip.v[k * kWave + lane]);
  }
  wave_lds_sync();
}

// load: the inverse -- 3 x 1 KiB contiguous global reads, then each lane picks
// up its own 48 bytes.
__device__ __forceinline__ void strip_load_row(PackedStrip& strip, int lane,
                                               u32 (&o)[12], bool lane_valid,
                                               const uint8_t* row_base,
                                               int valid_bytes) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k * kWave + lane) * 16;
    if (off < valid_bytes)
      strip.v[k * kWave + lane] = load16(row_base + off);
  }
  wave_lds_sync();
  if (lane_valid) {
    const uint4 a = strip.v[lane * 3 + 0], b = strip.v[lane * 3 + 1],
                c = strip.v[lane * 3 + 2];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    o[8] = c.x; o[9] = c.y; o[10] = c.z; o[11] = c.w;
  }
  wave_lds_sync();
}

// Two-phase form of strip_load_row for kernels that read SEVERAL packed rows per lane:
// strip_fetch issues the 3 global loads of a row into registers (no LDS, no wait), so the
// loads of all rows are in flight together; strip_unpack then runs the row through the
// strip.  Measured on RGB -> RGB_PLANAR / YUV444 2160p (profiles/r01_converters.md).
struct StripRegs {
  uint4 v[3];
};

__device__ __forceinline__ void strip_fetch(StripRegs& r, int lane, const uint8_t* row_base,
                                            int valid_bytes) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k * kWave + lane) * 16;
    r.v[k] = off < valid_bytes ? load16(row_base + off) : make_uint4(0, 0, 0, 0);
  }
}

// any alignment (see strip_store_row_u)
__device__ __forceinline__ void strip_fetch_u(StripRegs& r, int lane, const uint8_t* row_base, int valid_bytes) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int off = (k * kWave + lane) * 16;
    r.v[k] = off < valid_bytes ? load16_u(row_base + off) : make_uint4(0, 0, 0, 0);
  }
}

// The 48 bytes of ONE lane's 16 packed pixels straight from / to global memory (the group a ragged width cuts:
// its window slides left so that it ends with the row, which takes it off the strip's 48-byte lane grid).
// n_px < 16 only for frames narrower than one group.
__device__ __forceinline__ void packed_group_load(const uint8_t* p, int n_px, u32 (&o)[12]) {
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint4 v = load16_n(p + 16 * k, n_px * 3 - 16 * k);
    o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
  }
}
__device__ __forceinline__ void packed_group_store(uint8_t* p, int n_px, const u32 (&o)[12]) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
    store16_n(p + 16 * k, make_uint4(o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]), n_px * 3 - 16 * k);
}

__device__ __forceinline__ void strip_unpack(PackedStrip& strip, int lane, const StripRegs& r,
                                             u32 (&o)[12], bool lane_valid) {
#pragma unroll
  for (int k = 0; k < 3; ++k)
    strip.v[k * kWave + lane] = r.v[k];
  wave_lds_sync();
  if (lane_valid) {
    const uint4 a = strip.v[lane * 3 + 0], b = strip.v[lane * 3 + 1], c = strip.v[lane * 3 + 2];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
    o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    o[8] = c.x; o[9] = c.y; o[10] = c.z; o[11] = c.w;
  }
  wave_lds_sync();
}

// Scatter 16 pixels (3 channels, already quantised into per-channel dwords
// c0/c1/c2, 4 pixels per dword) into the 12 dwords of a packed row segment.
// v_perm_b32 does an arbitrary 4-of-8 byte pick in one instruction.
//   packed bytes of 4 pixels: a0 b0 c0 a1 | b1 c1 a2 b2 | c2 a3 b3 c3
__device__ __forceinline__ void interleave3(const u32 (&a)[4], const u32 (&b)[4],
                                            const u32 (&c)[4], u32 (&o)[12]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // __builtin_amdgcn_perm(hi, lo, sel): byte k of result = bytes{hi:7..4, lo:3..0}[sel.k]
    const u32 ab_lo = __builtin_amdgcn_perm(b[j], a[j], 0x05010400u); // a0 b0 a1 b1
    const u32 ab_hi = __builtin_amdgcn_perm(b[j], a[j], 0x07030602u); // a2 b2 a3 b3
    // d0 = a0 b0 c0 a1
    o[3 * j + 0] = __builtin_amdgcn_perm(c[j], ab_lo, 0x02040100u);
    // d1 = b1 c1 a2 b2 : b1 = ab_lo.3, c1 = c.1, a2 = ab_hi.0, b2 = ab_hi.1
    const u32 t = __builtin_amdgcn_perm(c[j], ab_lo, 0x00000503u);    // b1 c1 . .
    o[3 * j + 1] = __builtin_amdgcn_perm(ab_hi, t, 0x05040100u);
    // d2 = c2 a3 b3 c3
    o[3 * j + 2] = __builtin_amdgcn_perm(c[j], ab_hi, 0x07030206u);
  }
}

// Inverse of interleave3.
__device__ __forceinline__ void deinterleave3(const u32 (&o)[12], u32 (&a)[4],
                                              u32 (&b)[4], u32 (&c)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32 d0 = o[3 * j], d1 = o[3 * j + 1], d2 = o[3 * j + 2];
    // a = d0.0 d0.3 d1.2 d2.1
    const u32 a01 = __builtin_amdgcn_perm(d1, d0, 0x00060300u); // a0 a1 a2 .
    a[j] = __builtin_amdgcn_perm(d2, a01, 0x05020100u);
    // b = d0.1 d1.0 d1.3 d2.2
    const u32 b01 = __builtin_amdgcn_perm(d1, d0, 0x00070401u); // b0 b1 b2 .
    b[j] = __builtin_amdgcn_perm(d2, b01, 0x06020100u);
    // c = d0.2 d1.1 d2.0 d2.3
    const u32 c01 = __builtin_amdgcn_perm(d1, d0, 0x00000502u); // c0 c1 . .
    c[j] = __builtin_amdgcn_perm(d2, c01, 0x07040100u);
  }
}

// ---------------------------------------------------------------------------
// Finish + store 4 adjacent pixels of C interleaved channels held as floats (geometry
// kernels: resize, rotate).  u8/u16: round-half-even + saturate; f32: as is.  The 4*C
// elements are packed into dwords IN REGISTERS (a byte array + memcpy goes through scratch
// memory and serialises every load behind it) and leave as one wide store when the lane
// has all 4 pixels and the address is aligned; `mask` = which of the 4 pixels exist.
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ u32 finish_bits(float v);
template <> __device__ __forceinline__ u32 finish_bits<uint8_t>(float v) { return quantize_u8(v); }
template <> __device__ __forceinline__ u32 finish_bits<uint16_t>(float v) {
  const float r = __builtin_fminf(__builtin_fmaxf(__builtin_rintf(v), 0.0f), 65535.0f);
  return (u32)r;
}
template <> __device__ __forceinline__ u32 finish_bits<float>(float v) { return __float_as_uint(v); }

#ifndef VALI_PX4_NT
#define VALI_PX4_NT 1
#endif
#if VALI_PX4_NT
#define VALI_PX4_ST(T) gstore_nt<T>
#else
#define VALI_PX4_ST(T) gstore<T>
#endif
template <typename T, int C>
__device__ __forceinline__ void store_px4(uint8_t* dst, const float (&res)[4][C], u32 mask) {
  constexpr int E = (int)sizeof(T), N = 4 * C, NB = N * E, PER = 4 / E; // PER elements per dword
  u32 w[NB / 4];
#pragma unroll
  for (int k = 0; k < NB / 4; ++k)
    w[k] = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const float v = res[k / C][k % C];
    if constexpr (E == 1)
      w[k / 4] = __builtin_amdgcn_cvt_pk_u8_f32(v, (u32)(k % 4), w[k / 4]);
    else
      w[k / PER] |= finish_bits<T>(v) << (8 * E * (k % PER));
  }
  constexpr u32 kAlign = NB % 16 == 0 ? 15u : (NB % 8 == 0 ? 7u : 3u);
  if (mask == 0xfu && (((uintptr_t)dst) & kAlign) == 0) {
    // One instruction per lane (NB = 4, 8, 12, 16: neighbouring lanes write neighbouring bytes, the wave instruction
    // covers whole 128-byte lines) goes out non-temporal; the wider groups are 2-3 pieces per lane, NB bytes apart.
    if constexpr (NB == 16) {
      const v4u32 q = {w[0], w[1], w[2], w[3]};
      VALI_PX4_ST(v4u32)(dst, q);
    } else if constexpr (NB % 16 == 0) {
#pragma unroll
      for (int k = 0; k < NB / 16; ++k) {
        const v4u32 q = {w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]};
        ((VALI_GLOBAL v4u32*)dst)[k] = q;
      }
    } else if constexpr (NB == 8) {
      const v2u32 q = {w[0], w[1]};
      VALI_PX4_ST(v2u32)(dst, q);
    } else if constexpr (NB % 8 == 0) {
#pragma unroll
      for (int k = 0; k < NB / 8; ++k) {
        const v2u32 q = {w[2 * k], w[2 * k + 1]};
        ((VALI_GLOBAL v2u32*)dst)[k] = q;
      }
    } else if constexpr (NB == 12) {
      typedef unsigned v3u32 __attribute__((ext_vector_type(3)));
      const v3u32 q = {w[0], w[1], w[2]};
      VALI_PX4_ST(v3u32)(dst, q); // global_store_dwordx3 (needs 4-byte alignment only)
    } else if constexpr (NB == 4) {
      VALI_PX4_ST(u32)(dst, w[0]);
    } else {
#pragma unroll
      for (int k = 0; k < NB / 4; ++k)
        ((VALI_GLOBAL u32*)dst)[k] = w[k];
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < N; ++k)
    if (mask & (1u << (k / C))) {
      const u32 bits = w[k / PER] >> (8 * E * (k % PER));
      ((VALI_GLOBAL T*)dst)[k] = __builtin_bit_cast(T, (typename std::conditional<E == 1, uint8_t, typename std::conditional<E == 2, uint16_t, u32>::type>::type)bits);
    }
}

// The two horizontal taps of a bilinear sample are neighbours (i and min(i + 1, sw - 1)), so a
// row's pair comes from ONE unaligned load instead of 2 C byte loads: u8 x 1 -> 2 bytes, u8 x 2 -> 4
// bytes, u8 x 3 -> 8 bytes (the 6 of two packed pixels), u16 x 1 -> 4 bytes.  The texture addresser takes ~16
// cycles per wave instruction whatever its width (profiles/r01_ud_down2.md), and packed RGB paid
// it 12 times per pixel.  At the last column (i = sw - 1, only when the coordinate is exactly on
// it) the window starts one pixel earlier and both taps take its second pixel; the 8-byte window
// of packed RGB also slides left at the end of the row and is shifted back in registers.
// Needs sw >= 3 (checked by the caller); float and the other layouts keep the element loads.
template <typename T, int C> struct PairLoad {
  static constexpr bool kMerged = (sizeof(T) == 1 && C <= 3) || (sizeof(T) == 2 && C == 1);
};
template <typename T, int C>
__device__ __forceinline__ void load_tap_pair(const uint8_t* row, int i, int sw, float (&t0)[C], float (&t1)[C]) {
  typedef uint16_t u16_unaligned __attribute__((aligned(1)));
  typedef u32 u32_unaligned __attribute__((aligned(1)));
  typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
  const int base = min(i, sw - 2);
  const bool edge = i != base;
  if constexpr (sizeof(T) == 1 && C == 1) {
    const u32 w = *(const VALI_GLOBAL u16_unaligned*)(row + base);
    t1[0] = (float)(w >> 8);
    t0[0] = edge ? t1[0] : (float)(w & 0xffu);
  } else if constexpr (sizeof(T) == 1 && C == 2) {
    const u32 w = *(const VALI_GLOBAL u32_unaligned*)(row + 2 * base);
    t1[0] = ubyte_f32<2>(w); t1[1] = ubyte_f32<3>(w);
    t0[0] = edge ? t1[0] : ubyte_f32<0>(w);
    t0[1] = edge ? t1[1] : ubyte_f32<1>(w);
  } else if constexpr (sizeof(T) == 2 && C == 1) {
    const u32 w = *(const VALI_GLOBAL u32_unaligned*)(row + 2 * base);
    t1[0] = (float)(w >> 16);
    t0[0] = edge ? t1[0] : (float)(w & 0xffffu);
  } else {
    const int want = 3 * base, start = min(want, 3 * sw - 8);
    const unsigned long long q = *(const VALI_GLOBAL u64_unaligned*)(row + start) >> (8 * (want - start));
    const u32 lo = (u32)q, hi = (u32)(q >> 32);
    t1[0] = ubyte_f32<3>(lo); t1[1] = ubyte_f32<0>(hi); t1[2] = ubyte_f32<1>(hi);
    t0[0] = edge ? t1[0] : ubyte_f32<0>(lo);
    t0[1] = edge ? t1[1] : ubyte_f32<1>(lo);
    t0[2] = edge ? t1[2] : ubyte_f32<2>(lo);
  }
}


} // namespace vali
