// Tap tables of the Lanczos-3 / bicubic resizers: for one axis of a plane (src_n -> dst_n samples) entry x holds the six
// (four) weights and the first source index of dst sample x -- what make_lz_tap (resize_weights.hpp) returns, computed by
// make_lz_tap itself, once per geometry and device, by a small kernel on the stream of the first caller.
//
// Why: the columns-first kernels evaluate a tile's taps in every workgroup -- four sets of column taps (~130 instructions
// each) in every consumer wave, one set of row taps in every producer -- and are bound by the instructions they issue
// (profiles/r05_lanczos.md): 2160p -> 1936x1088 spent 14 % of its instructions re-deriving the same 1936 + 1088 + 968 + 544
// tap sets 290 times per frame.  A table entry is two 16-byte loads that hit the L2.
//
// Memory (round 6): every table of a device lives in ONE arena of kArenaBytes that is reserved by the first Lanczos / bicubic
// call of the process on that device -- the only allocation this file ever makes, so no later call can stall behind a
// hipMalloc (which may synchronise the device).  Tables are immutable while they exist.  When the arena or the table count
// (VALI_TUNE_TAP_MAX_TABLES) is exhausted the least recently used tables are evicted and their space is re-used; before the
// kernel that writes the new table may run, the caller's stream is made to wait for everything the streams that read an
// evicted table have been given so far (an event recorded on each of them at eviction time) -- a launch that still reads
// the old table finishes first.  A table read by more streams than kMaxUsers is simply never evicted.  A caller on another
// stream than the writer's waits for the writer's event until that event has completed once.
// While a stream is being captured into a graph nothing is allocated, launched or waited for: the kernels compute their
// taps themselves (tap_table returns null); the same for axes of more than kMaxAxis samples (a table of 1 MiB).
// VALI_TUNE_TAP_FALLBACKS counts the calls that got no table outside a capture, VALI_TUNE_TAP_EVICTIONS the evictions.
#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "resize_common.hpp"
#include "resize_weights.hpp"

namespace vali {

void tuning_add(int key, int delta); // runtime.hip

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

constexpr size_t kArenaBytes = 32u << 20; // per device: 1 M destination samples, e.g. 270 axes of a 2160p frame
constexpr int kMaxAxis = 32768;           // samples of the longest axis that gets a table (1 MiB)
constexpr size_t kGranule = 256;
constexpr int kMaxUsers = 4;
constexpr int kMinTables = 8;

typedef unsigned long long Key;

struct Entry {
  size_t off = 0, bytes = 0;
  hipEvent_t ready = nullptr;
  bool done = false;         // the writer's event has been seen complete
  hipStream_t writer = nullptr; // the stream whose k_tap_table launch fills it
  hipStream_t user[kMaxUsers]; // streams that were handed this table
  int users = 0;
  bool pinned = false;       // read by more streams than `user` holds: never evicted
  std::list<Key>::iterator lru;
};

struct Arena {
  uint8_t* base = nullptr;
  bool failed = false;
  std::map<size_t, size_t> free_blocks; // offset -> bytes, coalesced
  std::unordered_map<Key, Entry> tables;
  std::list<Key> lru;                   // front = most recently used
  std::vector<hipEvent_t> spare_events;
};

std::mutex g_mutex;
// (leaked on purpose: no HIP call from a static destructor)
Arena& arena_of(int device) {
  static auto* a = new std::unordered_map<int, Arena>();
  return (*a)[device];
}

bool arena_reserve(Arena& a) {
  if (a.base || a.failed)
    return a.base != nullptr;
  // the one allocation; a capture that this THREAD has open on another stream must survive it
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
  void* p = nullptr;
  const hipError_t e = hipMalloc(&p, kArenaBytes);
  if (swapped)
    (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    a.failed = true;
    return false;
  }
  a.base = (uint8_t*)p;
  a.free_blocks[0] = kArenaBytes;
  return true;
}

bool arena_take(Arena& a, size_t bytes, size_t* off) {
  for (auto it = a.free_blocks.begin(); it != a.free_blocks.end(); ++it)
    if (it->second >= bytes) {
      *off = it->first;
      const size_t rest = it->second - bytes, at = it->first + bytes;
      a.free_blocks.erase(it);
      if (rest)
        a.free_blocks[at] = rest;
      return true;
    }
  return false;
}

void arena_give(Arena& a, size_t off, size_t bytes) {
  auto it = a.free_blocks.emplace(off, bytes).first;
  auto next = std::next(it);
  if (next != a.free_blocks.end() && it->first + it->second == next->first) {
    it->second += next->second;
    a.free_blocks.erase(next);
  }
  if (it != a.free_blocks.begin()) {
    auto prev = std::prev(it);
    if (prev->first + prev->second == it->first) {
      prev->second += it->second;
      a.free_blocks.erase(it);
    }
  }
}

hipEvent_t take_event(Arena& a) {
  if (!a.spare_events.empty()) {
    hipEvent_t e = a.spare_events.back();
    a.spare_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return e;
}

// Evict the least recently used table that may be evicted; `stream` (the stream that is about to write a new table) is
// ordered behind everything the evicted table's readers have been given.  false: nothing can be evicted.
bool evict_one(Arena& a, hipStream_t stream) {
  for (auto it = a.lru.rbegin(); it != a.lru.rend(); ++it) {
    Entry& e = a.tables[*it];
    if (e.pinned)
      continue;
    bool ordered = true;
    for (int k = 0; k < e.users && ordered; ++k) {
      if (e.user[k] == stream)
        continue; // in order on the caller's own stream
      hipEvent_t fence = take_event(a);
      ordered = fence && hipEventRecord(fence, e.user[k]) == hipSuccess && hipStreamWaitEvent(stream, fence, 0) == hipSuccess;
      if (fence)
        a.spare_events.push_back(fence); // (a recorded event may be re-recorded: the wait above captured this record)
    }
    if (!ordered) { // e.g. a reader stream that no longer exists: its work is ordered by draining the device -- rare, and only here
      (void)hipGetLastError();
      if (hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        return false;
      }
    }
    if (!e.done) // its writer may not have run yet: the new writer must not overtake it
      (void)hipStreamWaitEvent(stream, e.ready, 0);
    arena_give(a, e.off, e.bytes);
    if (e.ready)
      a.spare_events.push_back(e.ready);
    const Key k = *it;
    a.lru.erase(std::next(it).base());
    a.tables.erase(k);
    tuning_add(VALI_TUNE_TAP_EVICTIONS, 1);
    return true;
  }
  return false;
}

const float4* fall_back() {
  tuning_add(VALI_TUNE_TAP_FALLBACKS, 1);
  return nullptr;
}

} // namespace

// vali_stream_destroy: the stream's work is complete and its handle is about to die -- no table may name it as a reader any more
// (an eviction would record an event on it).
void tap_table_forget_stream(int device, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_mutex);
  Arena& a = arena_of(device);
  for (auto& kv : a.tables) {
    Entry& e = kv.second;
    int n = 0;
    for (int k = 0; k < e.users; ++k)
      if (e.user[k] != stream)
        e.user[n++] = e.user[k];
    e.users = n;
    if (!e.done && e.ready && hipEventQuery(e.ready) == hipSuccess)
      e.done = true;
    (void)hipGetLastError();
  }
}

const float4* tap_table(int device, hipStream_t stream, int src_n, int dst_n, int taps) {
  if (src_n <= 0 || dst_n <= 0 || src_n >= (1 << 24) || device < 0 || device > 255)
    return nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  if (dst_n > kMaxAxis)
    return fall_back();
  const Key key = ((Key)(taps == 6) << 48) | ((Key)src_n << 24) | (Key)dst_n;
  std::lock_guard<std::mutex> lock(g_mutex);
  Arena& a = arena_of(device);
  auto it = a.tables.find(key);
  if (it != a.tables.end()) {
    Entry& e = it->second;
    a.lru.splice(a.lru.begin(), a.lru, e.lru);
    bool known = e.pinned;
    for (int k = 0; k < e.users && !known; ++k)
      known = e.user[k] == stream;
    if (!known) {
      if (e.users < kMaxUsers)
        e.user[e.users++] = stream;
      else
        e.pinned = true;
    }
    if (!e.done) {
      if (hipEventQuery(e.ready) == hipSuccess)
        e.done = true;
      else {
        (void)hipGetLastError();
        if (e.writer != stream && hipStreamWaitEvent(stream, e.ready, 0) != hipSuccess) {
          (void)hipGetLastError();
          return fall_back();
        }
      }
    }
    return (const float4*)(a.base + e.off);
  }
  if (!arena_reserve(a))
    return fall_back();
  const size_t bytes = (((size_t)dst_n * 2 * sizeof(float4)) + kGranule - 1) / kGranule * kGranule;
  // (never fewer than kMinTables: one launch takes up to 6 tables -- 2 axes of 3 planes -- before its kernel is enqueued, and a table
  // handed out to THIS launch must not be evicted and re-written by the next call of the same launch)
  const size_t max_tables = (size_t)std::max(kMinTables, tuning(VALI_TUNE_TAP_MAX_TABLES));
  Entry e;
  while (a.tables.size() >= max_tables)
    if (!evict_one(a, stream))
      return fall_back();
  while (!arena_take(a, bytes, &e.off))
    if (!evict_one(a, stream))
      return fall_back();
  e.bytes = bytes;
  e.ready = take_event(a);
  float4* d = (float4*)(a.base + e.off);
  bool ok = e.ready != nullptr;
  if (ok) {
    const dim3 grid((unsigned)((dst_n + kBlock - 1) / kBlock));
    if (taps == 6)
      hipLaunchKernelGGL(k_tap_table<6>, grid, dim3(kBlock), 0, stream, d, src_n, dst_n);
    else
      hipLaunchKernelGGL(k_tap_table<4>, grid, dim3(kBlock), 0, stream, d, src_n, dst_n);
    ok = hipGetLastError() == hipSuccess && hipEventRecord(e.ready, stream) == hipSuccess;
  }
  if (!ok) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(stream); // nothing may still write the block that goes back to the arena
    arena_give(a, e.off, e.bytes);
    if (e.ready)
      a.spare_events.push_back(e.ready);
    return fall_back();
  }
  e.writer = stream;
  e.user[0] = stream;
  e.users = 1;
  a.lru.push_front(key);
  e.lru = a.lru.begin();
  a.tables.emplace(key, e);
  return d; // (same stream: ordered behind the kernel that writes it)
}

} // namespace vali
