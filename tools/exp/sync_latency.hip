// Measures the cost of a blocking call = tiny kernel + "wait until it has finished", per call, for the ways a host can wait:
//   sync    hipStreamSynchronize
//   event   hipEventRecord + hipEventSynchronize
//   flag    hipStreamWriteValue32 into pinned host memory + spin on it
//   kflag   a second 1-thread kernel that stores the value with a system-scope release + spin
// build: hipcc --offload-arch=gfx950 -O2 tools/exp/sync_latency.hip -o tools/exp/sync_latency ; run under different
// ROC_ACTIVE_WAIT_TIMEOUT / HSA_ENABLE_INTERRUPT settings.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void work(unsigned* p, int n) { if (threadIdx.x < n) atomicAdd(p + threadIdx.x, 1u); }
__global__ void signal_k(volatile unsigned* flag, unsigned v) { __hip_atomic_store((unsigned*)flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
// the work kernel itself announces its end: every workgroup counts itself out, the last one stores the flag
__global__ void work_end(unsigned* p, int n, unsigned* count, unsigned* flag, unsigned v) {
  if (threadIdx.x < n) atomicAdd(p + threadIdx.x, 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(count, 1u) == gridDim.x - 1) {
      *count = 0;
      __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* d; CK(hipMalloc(&d, 1024));
  unsigned* h; CK(hipHostMalloc(&h, 64, hipHostMallocMapped)); *h = 0;
  unsigned* hd; CK(hipHostGetDevicePointer((void**)&hd, h, 0));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int N = 3000;
  unsigned* cnt; CK(hipMalloc(&cnt, 4)); CK(hipMemset(cnt, 0, 4));
  for (int mode = 0; mode < 6; ++mode) {
    unsigned v = 0;
    for (int rep = 0; rep < 2; ++rep) {
      double t0 = now();
      for (int i = 0; i < N; ++i) {
        if (mode == 4) { ++v; hipLaunchKernelGGL(work_end, dim3(64), dim3(256), 0, s, d, 8, cnt, hd, v); while (__atomic_load_n((volatile unsigned*)h, __ATOMIC_ACQUIRE) != v) {} continue; }
        if (mode == 5) { hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, d, 8); continue; }
        hipLaunchKernelGGL(work, dim3(64), dim3(256), 0, s, d, 8);
        if (mode == 0) CK(hipStreamSynchronize(s));
        else if (mode == 1) { CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); }
        else if (mode == 2) { ++v; CK(hipStreamWriteValue32(s, hd, v, 0)); while (__atomic_load_n((volatile unsigned*)h, __ATOMIC_ACQUIRE) != v) {} }
        else { ++v; hipLaunchKernelGGL(signal_k, dim3(1), dim3(1), 0, s, hd, v); while (__atomic_load_n((volatile unsigned*)h, __ATOMIC_ACQUIRE) != v) {} }
      }
      if (mode == 5) CK(hipStreamSynchronize(s));
      double dt = (now() - t0) / N;
      if (rep) printf("%-6s %.2f us per blocking call\n", mode == 0 ? "sync" : mode == 1 ? "event" : mode == 2 ? "flag" : mode == 3 ? "kflag" : mode == 4 ? "kend" : "async", dt);
    }
  }
  return 0;
}
