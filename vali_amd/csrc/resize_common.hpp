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
bool resize_x23_fits(const ResizeJob& j, int elem, int src_w, int src_h, int dst_w, int dst_h);
int launch_resize_x23(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                      hipStream_t stream);

// The same filters for jobs whose planes all SHRINK (or keep) their height -- columns first (resize_cols.hip).  Which
// plane takes which order is part of the specification (oracle/vali_oracle.c resize_plane_taps): src_h >= dst_h.
int launch_resize_cols(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int dst_w, int dst_h, int n,
                       hipStream_t stream);

// The same filters for jobs whose one-channel planes are exactly doubled in both directions (resize_up2.hip): source width a
// multiple of 4, 8 / 16-bit elements.
int launch_resize_up2(const ResizeArgs& a, int elem, int taps, int src_w, int src_h, int n, hipStream_t stream);

} // namespace vali
