// Shared between resize.hip (bilinear / point kernels, entry points) and resize_taps.hip (Lanczos-3 / bicubic).
#pragma once
#include "common.hpp"
#include "dev_util.hpp"

namespace vali {

typedef PlaneJob ResizeJob; // dev_util.hpp

struct ResizeArgs {
  const vali_surface* d_src; // batch: device descriptor arrays
  const vali_surface* d_dst;
  int sw, sh, dw, dh;        // single frame: surface sizes (planes are resolved into the jobs)
  ResizeJob job[3];
  int njobs;
  TileMap map;
  int rows;         // k_resize: dst rows per wave (8; 4 or 2 for small launches)
  int force_gather; // VALI_TUNE_RESIZE_FORCE_GATHER: no LDS staging (tests reach the gather forms with ordinary sizes)
  // taps kernels (resize_taps.hip): dynamic LDS layout per wave = [stage_bytes: the staged source row][ring]
  int stage_bytes;  // bytes of a wave's stage (pads included), multiple of 16; 0 = gather only
  int lds_per_wave; // stage_bytes + ring bytes
  // columns-first taps kernel (resize_cols.hip)
  int cols_n;       // dst elements per tile along x (multiple of 4, <= 256)
  int cols_rps;     // dst rows per slot of a wave (1, 2, 4, 8): a wave owns slots x cols_rps rows; resize_up2.hip: source rows per wave
};

// Planes whose size is unchanged (UDPlanar's luma at unchanged size; any filter at 1:1 is the identity for integer element
// types, see launch_resize): a straight copy.  A workgroup = 2048 bytes x 16 rows: every thread has its eight 16-byte loads
// in flight before the first store.  Shared by k_plane_copy (resize.hip) and k_resize_up2 (jobs of kind 1).
constexpr int kCopyW = 2048, kCopyH = 16;
__device__ __forceinline__ void plane_copy_tile(const PlaneView& v, int row_bytes, u32 tx, u32 ty) {
  const int c = (int)tx * kCopyW + (int)(threadIdx.x & 127u) * 16;
  const int r0 = (int)ty * kCopyH + (int)(threadIdx.x >> 7);
  if (row_bytes < 16) { // (uniform) planes narrower than one vector: bytes
    if (c == 0)
      for (int i = 0; i < kCopyH / 2; ++i) {
        const int r = r0 + 2 * i;
        for (int b = 0; r < v.dh && b < row_bytes; ++b)
          gstore<uint8_t>(v.dp + (size_t)r * v.dpitch + b, gload<uint8_t>(v.sp + (size_t)r * v.spitch + b));
      }
    return;
  }
  if (c >= row_bytes)
    return;
  // a row's last vector slides left to END with the row (the bytes it shares with its neighbour are written twice with the
  // same value); rows past the plane are their clamped neighbour, loaded and not stored: no predicated loads -- behind a
  // condition every load of the array gets its own wait and a page of register copies
  const int cc = min(c, row_bytes - 16);
  v4u32 q[kCopyH / 2];
#pragma unroll
  for (int i = 0; i < kCopyH / 2; ++i)
    q[i] = gload_u<v4u32>(v.sp + (size_t)min(r0 + 2 * i, v.dh - 1) * v.spitch + cc);
#pragma unroll
  for (int i = 0; i < kCopyH / 2; ++i) {
    const int r = r0 + 2 * i;
    if (r < v.dh)
      gstore_u<v4u32>(v.dp + (size_t)r * v.dpitch + cc, q[i]);
  }
}

// Lanczos-3 (taps = 6) / bicubic (taps = 4) over the plane jobs of `a` (job[].comp / sub / channels filled in, planes
// resolved for single-frame launches); one launch per channel count present.  elem = bytes per element.
int launch_resize_taps(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream);

// The same rows-first arithmetic with the filtered rows in registers (resize_rows.hip): 8 / 16-bit planes of 1 / 2 channels
// that grow vertically and whose 256-element tile spans at most 64 groups of 4 source pixels (resize_rows_fits).
bool resize_rows_fits(const ResizeJob& j, int elem, int src_w, int dst_w, int taps);
int launch_resize_rows(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream);

// ... and for 8 / 16-bit planes of 1 / 2 channels enlarged by exactly 3:2 on both axes (720p -> 1080p): static tap positions,
// no LDS stage (resize_rows.hip, k_resize_rows_x23).
// ... packed 8-bit RGB that grows on both axes at any ratio in (1/3, 1): the register form for three channels (k_resize_rows_rgb)
bool resize_rows_rgb_fits(const ResizeJob& j, int elem, int taps, int src_w, int dst_w);
int launch_resize_rows_rgb(const ResizeArgs& a, int src_h, int dst_w, int dst_h, int n, hipStream_t stream);
bool resize_x23_fits(const ResizeJob& j, int elem, int src_w, int src_h, int dst_w, int dst_h);
int launch_resize_x23(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                      hipStream_t stream);

// Tap tables (tap_table.hip): entry x of the table of an axis src_n -> dst_n = {w0 w1 w2 w3} {w4 w5 i -} of dst sample x (bicubic:
// {w0 w1 w2 w3} {- - i -}), written once per geometry and device by make_lz_tap itself; null: the caller computes its taps (stream
// being captured, table memory exhausted).
const float4* tap_table(int device, hipStream_t stream, int src_n, int dst_n, int taps);

// The same filters for jobs whose planes all SHRINK (or keep) their height -- columns first (resize_cols.hip).  Which
// plane takes which order is part of the specification (oracle/vali_oracle.c resize_plane_taps): src_h >= dst_h.
int launch_resize_cols(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream);

// The same filters for jobs whose one-channel planes are exactly doubled in both directions (resize_up2.hip): source width a
// multiple of 4, 8 / 16-bit elements.
int launch_resize_up2(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int n, hipStream_t stream);

} // namespace vali
