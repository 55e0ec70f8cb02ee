// Runtime half of the C ABI: devices, streams, events, pitched memory, 2-D copies.
// HIP-native replacement for the reference's dlsym'd CUDA driver table
// (reference: src/TC/inc/LibCuda.hpp, src/TC/src/CudaUtils.cpp:20-299,
//  src/TC/src/SurfacePlane.cpp:186-213).  There is no context push/pop on HIP:
// a device is selected per call and a stream carries its device.
#include "common.hpp"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <memory>
#include <chrono>
#include <mutex>
#include <unordered_map>

namespace vali {

void tap_table_forget_stream(int device, hipStream_t stream); // tap_table.hip

// ---- the tuning table (include/vali_hip.h: vali_tuning_key) -- the ONLY place the library reads the environment
namespace {
struct TuneDef {
  const char* env;
  int def;
};
const TuneDef kTuneDefs[VALI_TUNE_COUNT] = {
    {"VALI_NV12_ROWPAIRS", 0}, {"VALI_WAVES_PER_CU", 0},      {"VALI_NV12_DIRECT_STORE", 0}, {"VALI_RESIZE_FORCE_GATHER", 0},
    {"VALI_RESIZE_POINT", 1},  {"VALI_UD_FORCE_GATHER", 0},   {"VALI_UD_DOWN2", 1},          {"VALI_UD_OCC5", 0},
    {"VALI_ROTATE_NO_TILE", 0}, {"VALI_ROCTX", 0},            {"VALI_RESIZE_NO_SEPARABLE", 0},
    {"VALI_ROWS_PER_WAVE", 0}, {"VALI_BLOCKING_WAIT", 0},    {"VALI_RESIZE_ROWS", 1},
    {"VALI_RESIZE_COLS", 0},   {"VALI_ROTATE_AFFINE", 0},   {"VALI_TAP_MAX_TABLES", 1024},
    {"VALI_TAP_FALLBACKS", 0}, {"VALI_TAP_EVICTIONS", 0}};
std::atomic<int> g_tune[VALI_TUNE_COUNT];
std::once_flag g_tune_once;

typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;

bool load_roctx() {
  static const bool ok = [] {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h)
        continue;
      g_roctx_push = (roctx_push_fn)dlsym(h, "roctxRangePushA");
      g_roctx_pop = (roctx_pop_fn)dlsym(h, "roctxRangePop");
      if (g_roctx_push && g_roctx_pop)
        return true;
    }
    g_roctx_push = nullptr;
    g_roctx_pop = nullptr;
    return false;
  }();
  return ok;
}

void tune_init() {
  std::call_once(g_tune_once, [] {
    for (int k = 0; k < VALI_TUNE_COUNT; ++k) {
      const char* e = getenv(kTuneDefs[k].env);
      g_tune[k].store(e && *e ? atoi(e) : kTuneDefs[k].def, std::memory_order_relaxed);
    }
    if (g_tune[VALI_TUNE_ROCTX].load(std::memory_order_relaxed) && !load_roctx())
      g_tune[VALI_TUNE_ROCTX].store(0, std::memory_order_relaxed);
  });
}
} // namespace

void tuning_add(int key, int delta) { // the counters of the table (VALI_TUNE_TAP_FALLBACKS ...)
  tune_init();
  g_tune[key].fetch_add(delta, std::memory_order_relaxed);
}

int tuning(int key) {
  tune_init();
  return g_tune[key].load(std::memory_order_relaxed);
}

TraceRange::TraceRange(const char* name) {
  if (tuning(VALI_TUNE_ROCTX) && g_roctx_push) {
    g_roctx_push(name);
    m_pushed = true;
  }
}

TraceRange::~TraceRange() {
  if (m_pushed && g_roctx_pop)
    g_roctx_pop();
}

std::string& last_error() {
  static thread_local std::string s;
  return s;
}

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

DeviceScope::DeviceScope(int device) {
  if (device < 0)
    return;
  if (hipGetDevice(&m_prev) != hipSuccess) {
    m_ok = false;
    (void)hipGetLastError();
    return;
  }
  if (m_prev != device) {
    if (hipSetDevice(device) != hipSuccess) {
      m_ok = false;
      (void)hipGetLastError();
      return;
    }
    m_switched = true;
  }
}

DeviceScope::~DeviceScope() {
  if (m_switched)
    (void)hipSetDevice(m_prev);
}

int stream_device(hipStream_t stream) {
  int dev = -1;
  if (stream) {
    hipDevice_t d;
    if (hipStreamGetDevice(stream, &d) == hipSuccess)
      return (int)d;
    (void)hipGetLastError();
  }
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return dev;
}

unsigned residency_lds_bytes(int block_threads, int waves_per_cu, unsigned min_bytes) {
  const int waves_per_block = block_threads / 64 > 0 ? block_threads / 64 : 1;
  const int blocks_per_cu = waves_per_cu / waves_per_block > 0 ? waves_per_cu / waves_per_block : 1;
  unsigned bytes = (160u * 1024u / (unsigned)blocks_per_cu) & ~1023u; // floor to 1 KiB
  if (bytes > 1024u)
    bytes -= 512u; // stay strictly inside the residency step
  if (bytes > 64u * 1024u)
    bytes = 64u * 1024u; // default per-workgroup dynamic LDS limit
  return bytes < min_bytes ? min_bytes : bytes;
}

} // namespace vali

using namespace vali;

#define VALI_DEVICE(device)                                                    \
  DeviceScope _scope(device);                                                  \
  if (!_scope.ok())                                                            \
  return fail(VALI_ERR_NO_DEVICE, "%s: cannot select device %d", __func__, device)

extern "C" {

const char* vali_last_error(void) { return last_error().c_str(); }

const char* vali_version(void) { return "vali_hip 0.2 (gfx950)"; }

int vali_tuning_set(int key, int value) {
  VALI_REQUIRE(key >= 0 && key < VALI_TUNE_COUNT, "unknown tuning key");
  tune_init();
  if (key == VALI_TUNE_ROCTX && value && !load_roctx())
    return fail(VALI_ERR_UNSUPPORTED, "vali_tuning_set: no roctx library (librocprofiler-sdk-roctx / libroctx64) to trace with");
  g_tune[key].store(value, std::memory_order_relaxed);
  return VALI_OK;
}

int vali_tuning_get(int key, int* value) {
  VALI_REQUIRE(value && key >= 0 && key < VALI_TUNE_COUNT, "unknown tuning key or null result");
  *value = tuning(key);
  return VALI_OK;
}

int vali_device_count(int* count) {
  VALI_REQUIRE(count, "null count");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *count = 0;
    return fail(VALI_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return VALI_OK;
}

int vali_device_set(int device) {
  VALI_HIP_CHECK(hipSetDevice(device));
  return VALI_OK;
}

int vali_device_get(int* device) {
  VALI_REQUIRE(device, "null argument");
  VALI_HIP_CHECK(hipGetDevice(device));
  return VALI_OK;
}

int vali_ptr_device(const void* dptr, int* device) {
  VALI_REQUIRE(dptr && device, "null argument");
  hipPointerAttribute_t attr;
  VALI_HIP_CHECK(hipPointerGetAttributes(&attr, dptr));
  *device = attr.device;
  return VALI_OK;
}

int vali_stream_create(int device, vali_stream_t* stream) {
  VALI_REQUIRE(stream, "null stream");
  VALI_DEVICE(device);
  hipStream_t s = nullptr;
  VALI_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (vali_stream_t)s;
  return VALI_OK;
}

namespace {
void drop_wait_slot(int device, hipStream_t s);   // the stream's completion word (below) goes with the stream
}

int vali_stream_destroy(int device, vali_stream_t stream) {
  VALI_DEVICE(device);
  if (stream) {
    drop_wait_slot(device, as_stream(stream));
    VALI_HIP_CHECK(hipStreamSynchronize(as_stream(stream))); // what it was given is done: no tap table is still read through it
    tap_table_forget_stream(device, as_stream(stream));
    VALI_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
  }
  return VALI_OK;
}

int vali_stream_sync(int device, vali_stream_t stream) {
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return VALI_OK;
}

// ---- vali_stream_wait: the host side of the blocking Run forms --------------------------------------------------
// hipStreamSynchronize costs 11.8-12.6 us around a 1 us kernel on this stack, 3-4 us of it inside the runtime's signal
// wait (tools/exp/sync_latency.hip: interrupts off and ROC_ACTIVE_WAIT_TIMEOUT change nothing).  A 4-byte word in
// pinned, device-mapped host memory that the stream itself writes when it gets there (hipStreamWriteValue32) and the host
// spins on takes 8.7 us -- 0.6 us above a kernel that announces its own end, the floor of a launch -> completion round
// trip on this hardware.  One word per stream; after ~150 us of spinning (a long kernel, or a stream that will never get
// there) the wait falls back to hipStreamSynchronize, which also surfaces errors.
namespace {
struct WaitSlot {
  unsigned* host = nullptr;
  unsigned* dev = nullptr;
  unsigned value = 0;   // last value handed out; read and written under `order` only
  std::mutex order;     // taking a value and enqueueing its write are ONE step: values reach the stream in order
  ~WaitSlot() {
    if (host)
      (void)hipHostFree(host);
  }
};
std::mutex g_wait_mutex;
// (device, stream) -> slot.  shared_ptr: a waiter keeps its slot alive while vali_stream_destroy drops the map's entry.
// The map itself is never destroyed (a leaked heap object): its slots would call hipHostFree from a static destructor,
// while or after the HIP runtime tears itself down -- slots of streams still alive at exit are left to the process.
std::unordered_map<uint64_t, std::shared_ptr<WaitSlot>>& g_wait_slots = *new std::unordered_map<uint64_t, std::shared_ptr<WaitSlot>>();

uint64_t wait_key(int device, hipStream_t s) { return ((uint64_t)(unsigned)device << 56) ^ (uint64_t)(uintptr_t)s; }

void drop_wait_slot(int device, hipStream_t s) {
  std::shared_ptr<WaitSlot> gone;             // (freed AFTER the lock is released: hipHostFree may synchronise the device,
  {                                           // and every vali_stream_wait on another stream takes this mutex)
    std::lock_guard<std::mutex> lock(g_wait_mutex);
    auto it = g_wait_slots.find(wait_key(device, s));
    if (it != g_wait_slots.end()) {
      gone = std::move(it->second);
      g_wait_slots.erase(it);                 // a recycled stream handle starts with a fresh slot
    }
  }
}

std::shared_ptr<WaitSlot> wait_slot(int device, hipStream_t s) {
  std::lock_guard<std::mutex> lock(g_wait_mutex);
  const uint64_t key = wait_key(device, s);
  auto it = g_wait_slots.find(key);
  if (it != g_wait_slots.end())
    return it->second;
  auto wp = std::make_shared<WaitSlot>();
  WaitSlot& w = *wp;
  void* h = nullptr;
  if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostFree(h);
    return nullptr;
  }
  w.host = (unsigned*)h;
  w.dev = (unsigned*)d;
  *w.host = 0;
  g_wait_slots.emplace(key, wp);
  return wp;
}
} // namespace

int vali_stream_wait(int device, vali_stream_t stream) {
  VALI_DEVICE(device);
  hipStream_t s = as_stream(stream);
  // Two threads blocking on the SAME stream (tasks built without a stream share their GPU's manager stream, and the GIL
  // is released around this call) take their values and enqueue their writes under the slot's lock: the word only ever
  // grows, and "word >= my value" means everything enqueued before my write has finished.
  std::shared_ptr<WaitSlot> w = tuning(VALI_TUNE_BLOCKING_WAIT) == 0 ? wait_slot(device, s) : nullptr;
  if (w) {
    unsigned v;
    hipError_t we;
    {
      std::lock_guard<std::mutex> order(w->order);
      v = w->value + 1;
      we = hipStreamWriteValue32(s, w->dev, v, 0);
      if (we == hipSuccess)
        w->value = v;
    }
    if (we == hipSuccess) {
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spin = 1;; ++spin) {
        if ((int)(__atomic_load_n((volatile unsigned*)w->host, __ATOMIC_ACQUIRE) - v) >= 0)
          return VALI_OK;
        if ((spin & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(150))
          break;
      }
    } else {
      (void)hipGetLastError();
    }
  }
  VALI_HIP_CHECK(hipStreamSynchronize(s));
  return VALI_OK;
}

int vali_event_query(int device, vali_event_t event, int* done) {
  VALI_REQUIRE(event && done, "null argument");
  VALI_DEVICE(device);
  const hipError_t e = hipEventQuery((hipEvent_t)event);
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();
    *done = 0;
    return VALI_OK;
  }
  VALI_HIP_CHECK(e);
  *done = 1;
  return VALI_OK;
}

int vali_stream_wait_event(int device, vali_stream_t stream, vali_event_t event) {
  VALI_REQUIRE(event, "null event");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)event, 0));
  return VALI_OK;
}

int vali_host_alloc(int device, size_t bytes, void** hptr) {
  VALI_REQUIRE(hptr && bytes > 0, "bad argument");
  VALI_DEVICE(device);
  void* p = nullptr;
  VALI_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
  *hptr = p;
  return VALI_OK;
}

int vali_host_free(int device, void* hptr) {
  VALI_DEVICE(device);
  if (hptr)
    VALI_HIP_CHECK(hipHostFree(hptr));
  return VALI_OK;
}

int vali_mem_info(int device, size_t* free_bytes, size_t* total_bytes) {
  VALI_REQUIRE(free_bytes && total_bytes, "null argument");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipMemGetInfo(free_bytes, total_bytes));
  return VALI_OK;
}

int vali_device_pci_bus_id(int device, char* out, int len) {
  VALI_REQUIRE(out && len >= 16, "buffer of at least 16 bytes");
  VALI_HIP_CHECK(hipDeviceGetPCIBusId(out, len, device));
  return VALI_OK;
}

int vali_event_create(int device, vali_event_t* event) {
  VALI_REQUIRE(event, "null event");
  VALI_DEVICE(device);
  hipEvent_t e = nullptr;
  VALI_HIP_CHECK(hipEventCreate(&e));
  *event = (vali_event_t)e;
  return VALI_OK;
}

int vali_event_destroy(int device, vali_event_t event) {
  VALI_DEVICE(device);
  if (event)
    VALI_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
  return VALI_OK;
}

int vali_event_record(int device, vali_event_t event, vali_stream_t stream) {
  VALI_REQUIRE(event, "null event");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipEventRecord((hipEvent_t)event, as_stream(stream)));
  return VALI_OK;
}

int vali_event_sync(int device, vali_event_t event) {
  VALI_REQUIRE(event, "null event");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
  return VALI_OK;
}

int vali_event_elapsed_ms(vali_event_t start, vali_event_t stop, float* ms) {
  VALI_REQUIRE(start && stop && ms, "null argument");
  VALI_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return VALI_OK;
}

// ---- stream capture: a chain of asynchronous vali_* calls replayed as ONE hipGraph launch ----
int vali_graph_capture_begin(int device, vali_stream_t stream) {
  VALI_REQUIRE(stream, "capture needs an explicit (non-null) stream");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return VALI_OK;
}

int vali_graph_capture_end(int device, vali_stream_t stream, vali_graph_t* graph) {
  VALI_REQUIRE(stream && graph, "null argument");
  VALI_DEVICE(device);
  hipGraph_t g = nullptr;
  VALI_HIP_CHECK(hipStreamEndCapture(as_stream(stream), &g));
  if (!g)
    return fail(VALI_ERR_RUNTIME, "vali_graph_capture_end: capture was invalidated");
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess)
    return fail(VALI_ERR_RUNTIME, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  *graph = (vali_graph_t)exec;
  return VALI_OK;
}

int vali_graph_launch(int device, vali_graph_t graph, vali_stream_t stream) {
  VALI_REQUIRE(graph, "null graph");
  VALI_DEVICE(device);
  VALI_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph, as_stream(stream)));
  return VALI_OK;
}

int vali_graph_destroy(int device, vali_graph_t graph) {
  VALI_DEVICE(device);
  if (graph)
    VALI_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph));
  return VALI_OK;
}

int vali_mem_alloc_pitch(int device, size_t width_bytes, size_t height, void** dptr,
                         size_t* pitch) {
  VALI_REQUIRE(dptr && pitch, "null argument");
  VALI_REQUIRE(width_bytes > 0 && height > 0, "empty plane");
  VALI_DEVICE(device);
  const size_t p = (width_bytes + 255u) & ~(size_t)255u;
  void* mem = nullptr;
  VALI_HIP_CHECK(hipMalloc(&mem, p * height));
  *dptr = mem;
  *pitch = p;
  return VALI_OK;
}

int vali_mem_alloc(int device, size_t bytes, void** dptr) {
  VALI_REQUIRE(dptr, "null argument");
  VALI_REQUIRE(bytes > 0, "zero-size allocation");
  VALI_DEVICE(device);
  void* mem = nullptr;
  VALI_HIP_CHECK(hipMalloc(&mem, bytes));
  *dptr = mem;
  return VALI_OK;
}

int vali_mem_free(int device, void* dptr) {
  VALI_DEVICE(device);
  if (dptr)
    VALI_HIP_CHECK(hipFree(dptr));
  return VALI_OK;
}

int vali_memcpy2d_async(int device, void* dst, size_t dst_pitch, const void* src,
                        size_t src_pitch, size_t width_bytes, size_t height,
                        int kind, vali_stream_t stream) {
  VALI_REQUIRE(dst && src, "null pointer");
  VALI_REQUIRE(kind >= 0 && kind <= 2, "bad copy kind");
  VALI_REQUIRE(dst_pitch >= width_bytes && src_pitch >= width_bytes,
               "pitch smaller than row");
  if (width_bytes == 0 || height == 0)
    return VALI_OK;
  VALI_DEVICE(device);
  const hipMemcpyKind k = kind == 0   ? hipMemcpyHostToDevice
                          : kind == 1 ? hipMemcpyDeviceToHost
                                      : hipMemcpyDeviceToDevice;
  VALI_HIP_CHECK(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, height,
                                  k, as_stream(stream)));
  return VALI_OK;
}

int vali_memset2d_async(int device, void* dst, size_t dst_pitch, int value,
                        size_t width_bytes, size_t height, vali_stream_t stream) {
  VALI_REQUIRE(dst, "null pointer");
  VALI_REQUIRE(dst_pitch >= width_bytes, "pitch smaller than row");
  if (width_bytes == 0 || height == 0)
    return VALI_OK;
  VALI_DEVICE(device);
  VALI_HIP_CHECK(
      hipMemset2DAsync(dst, dst_pitch, value, width_bytes, height, as_stream(stream)));
  return VALI_OK;
}

} // extern "C"
