// Tap tables of the Lanczos-3 / bicubic resizers: for one axis of a plane (src_n -> dst_n samples) entry x holds the six
// (four) weights and the first source index of dst sample x -- what make_lz_tap (resize_weights.hpp) returns, computed by
// make_lz_tap itself, once per geometry and device, by a small kernel on the stream of the first caller.
//
// Why: the columns-first kernels evaluate a tile's taps in every workgroup -- four sets of column taps (~130 instructions
// each) in every consumer wave, one set of row taps in every producer -- and are bound by the instructions they issue
// (profiles/r05_lanczos.md): 2160p -> 1936x1088 spent 14 % of its instructions re-deriving the same 1936 + 1088 + 968 + 544
// tap sets 290 times per frame.  A table entry is two 16-byte loads that hit the L2.
//
// The tables are immutable once written and live until the process ends (a 2160p axis is 120 KB; at most kMaxTables
// geometries are kept, later ones are computed in the kernels as before).  A caller on another stream waits for the
// writer's event until that event has completed once.  While a stream is being captured into a graph nothing is
// allocated, launched or waited for: the kernels compute their taps themselves (tap_table returns null).
#include <mutex>
#include <unordered_map>

#include "resize_common.hpp"
#include "resize_weights.hpp"

namespace vali {

namespace {

template <int TAPS> __global__ void __launch_bounds__(kBlock) k_tap_table(float4* out, int src_n, int dst_n) {
  const int x = (int)(blockIdx.x * kBlock + threadIdx.x);
  if (x >= dst_n)
    return;
  const LzTap<TAPS> t = make_lz_tap<TAPS>(x, (float)src_n / (float)dst_n);
  if constexpr (TAPS == 6) {
    out[2 * x] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
    out[2 * x + 1] = make_float4(t.w[4], t.w[5], __int_as_float(t.i), 0.0f);
  } else {
    out[2 * x] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
    out[2 * x + 1] = make_float4(0.0f, 0.0f, __int_as_float(t.i), 0.0f);
  }
}

struct Entry {
  float4* d = nullptr;
  hipEvent_t ready = nullptr;
  bool done = false; // the writer's event has been seen complete
};

constexpr size_t kMaxTables = 256;
std::mutex g_mutex;
// (leaked on purpose: no HIP call from a static destructor)
std::unordered_map<unsigned long long, Entry>& tables() {
  static auto* m = new std::unordered_map<unsigned long long, Entry>();
  return *m;
}

} // namespace

const float4* tap_table(int device, hipStream_t stream, int src_n, int dst_n, int taps) {
  if (src_n <= 0 || dst_n <= 0 || src_n >= (1 << 24) || dst_n >= (1 << 24) || device < 0 || device > 255)
    return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  const unsigned long long key = ((unsigned long long)device << 56) | ((unsigned long long)(taps == 6) << 48) |
                                 ((unsigned long long)src_n << 24) | (unsigned long long)dst_n;
  std::lock_guard<std::mutex> lock(g_mutex);
  auto& m = tables();
  auto it = m.find(key);
  if (it == m.end()) {
    if (m.size() >= kMaxTables)
      return nullptr;
    Entry e;
    if (hipMalloc((void**)&e.d, (size_t)dst_n * 2 * sizeof(float4)) != hipSuccess ||
        hipEventCreateWithFlags(&e.ready, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      if (e.d)
        (void)hipFree(e.d);
      return nullptr;
    }
    const dim3 grid((unsigned)((dst_n + kBlock - 1) / kBlock));
    if (taps == 6)
      hipLaunchKernelGGL(k_tap_table<6>, grid, dim3(kBlock), 0, stream, e.d, src_n, dst_n);
    else
      hipLaunchKernelGGL(k_tap_table<4>, grid, dim3(kBlock), 0, stream, e.d, src_n, dst_n);
    if (hipGetLastError() != hipSuccess || hipEventRecord(e.ready, stream) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);
      (void)hipFree(e.d);
      (void)hipEventDestroy(e.ready);
      return nullptr;
    }
    m.emplace(key, e);
    return e.d; // (same stream: ordered behind the kernel that writes it)
  }
  Entry& e = it->second;
  if (!e.done) {
    if (hipEventQuery(e.ready) == hipSuccess)
      e.done = true;
    else {
      (void)hipGetLastError();
      if (hipStreamWaitEvent(stream, e.ready, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
      }
    }
  }
  return e.d;
}

} // namespace vali
