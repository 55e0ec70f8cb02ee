// Host-side helpers shared by every translation unit of libvali_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/vali_hip.h"

namespace vali {

// Text of the last failure on this thread (returned by vali_last_error()).
std::string& last_error();

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Make `device` current for the scope; restores the previous one on exit.
// One process per GPU is the intended deployment, so this is normally a no-op
// pair of hipGetDevice calls.
class DeviceScope {
public:
  explicit DeviceScope(int device);
  ~DeviceScope();
  bool ok() const { return m_ok; }

private:
  int m_prev = -1;
  bool m_switched = false;
  bool m_ok = true;
};

// Device a launch on `stream` must be issued from (the stream's device, or the
// current device for the null stream).
int stream_device(hipStream_t stream);

// Dynamic-LDS size that caps residency at `waves_per_cu` (160 KiB LDS per CU), never below
// `min_bytes` (what the kernel really uses).  Fewer concurrent row streams per L2 measured
// +4% HBM throughput for the streaming converters (profiles/r01_variants.md).
unsigned residency_lds_bytes(int block_threads, int waves_per_cu, unsigned min_bytes);

// Residency of the streaming converters, in waves per CU.  16 waves of FULL lanes measured best
// for plane -> packed streams (profiles/r01_variants.md); when the row width leaves lanes idle
// (1280 px = 80 of 128 lanes, 2560 px = 160 of 192) the same bytes in flight need more waves:
// 720p 5.29 -> 5.85 TB/s, 1440p 5.18 -> 5.85 TB/s with 24-28 (sweep in profiles/r01_variants.md).
inline int streaming_waves_per_cu(int groups, int block, int full_lane_waves) {
  const int lanes = ((groups + block - 1) / block) * block; // lanes launched per row pair
  const int util_pct = groups * 100 / lanes;
  return util_pct >= 90 ? full_lane_waves : util_pct >= 70 ? full_lane_waves * 3 / 2 : full_lane_waves * 7 / 4;
}

// Current value of a vali_tuning_key (include/vali_hip.h): one relaxed atomic load.
int tuning(int key);

// roctx range for the lifetime of the object when VALI_TUNE_ROCTX is on (the reference's NvtxMark,
// src/TC/inc/Tasks.hpp:32-59); otherwise one relaxed load.
class TraceRange {
public:
  explicit TraceRange(const char* name);
  ~TraceRange();

private:
  bool m_pushed = false;
};

inline hipStream_t as_stream(vali_stream_t s) { return (hipStream_t)s; }

inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// The gather kernels (resize, UD, rotate, pre-processing) form row offsets in 32 bits
// (s_mul_i32 instead of a 64-bit VALU multiply-add per row): every plane of a surface must
// stay below 4 GiB.  3 * height rows bounds the tallest plane layout (stacked planar).
inline bool planes_fit_32bit(const vali_surface& s) {
  for (int c = 0; c < 3; ++c)
    if (s.plane[c] && (s.pitch[c] < 0 || (uint64_t)s.pitch[c] * (uint64_t)s.height * 3u >= (1ull << 32)))
      return false;
  return true;
}

// ONE rule for sizes of surfaces with subsampled chroma, applied by every entry point: the chroma planes hold
// (W / 2) x (H / 2) samples (4:2:0: NV12, P10, P12, YUV420, YUV420_10bit) or (W / 2) x H (YUV422), so W (and H) must be
// even -- an odd size would address a chroma sample the plane does not have.  VALI_ERR_INVALID_ARG otherwise
// (python_vali: (False, TaskExecInfo.INVALID_INPUT)).
inline bool subsampled_sizes_ok(int format, int width, int height) {
  switch (format) {
  case VALI_FMT_NV12: case VALI_FMT_P10: case VALI_FMT_P12: case VALI_FMT_YUV420: case VALI_FMT_YUV420_10BIT:
    return ((width | height) & 1) == 0;
  case VALI_FMT_YUV422:
    return (width & 1) == 0;
  default:
    return true;
  }
}

} // namespace vali

// First lines of every operator entry point: trace range, then the device of the stream (the current device
// for the null stream) is made current for the call.  No usable device -> VALI_ERR_NO_DEVICE, never a silent
// launch on whatever happens to be current.
#define VALI_ENTRY_NAMED(stream, name)                                                       \
  ::vali::TraceRange _trace(name);                                                            \
  const int _dev = ::vali::stream_device(stream);                                             \
  if (_dev < 0)                                                                               \
    return ::vali::fail(VALI_ERR_NO_DEVICE, "%s: no usable HIP device", name);                \
  ::vali::DeviceScope scope(_dev);                                                            \
  if (!scope.ok())                                                                            \
    return ::vali::fail(VALI_ERR_NO_DEVICE, "%s: cannot select device %d", name, _dev)
#define VALI_ENTRY(stream) VALI_ENTRY_NAMED(stream, __func__)

#define VALI_HIP_CHECK(expr)                                                   \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess)                                                      \
      return ::vali::fail(VALI_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr,    \
                          hipGetErrorString(_e), __FILE__, __LINE__);          \
  } while (0)

#define VALI_REQUIRE(cond, msg)                                                \
  do {                                                                         \
    if (!(cond))                                                               \
      return ::vali::fail(VALI_ERR_INVALID_ARG, "%s: %s", __func__, msg);      \
  } while (0)

// Kernel launches cannot fail asynchronously here in a way hipGetLastError sees
// immediately, but a bad configuration does.
#define VALI_LAUNCH_CHECK()                                                    \
  do {                                                                         \
    hipError_t _e = hipGetLastError();                                         \
    if (_e != hipSuccess)                                                      \
      return ::vali::fail(VALI_ERR_RUNTIME, "%s: kernel launch failed: %s",    \
                          __func__, hipGetErrorString(_e));                    \
  } while (0)
