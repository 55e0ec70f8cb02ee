#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
// does v_cvt_pk_u8_f32 honour MODE.fp_round?  MODE[1:0] = f32 round mode: 0 nearest even, 1 +inf, 2 -inf, 3 toward zero
__global__ void k(const float* in, unsigned* out_rne, unsigned* out_rtz, float* fma_after, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = in[i];
  unsigned a = __builtin_amdgcn_cvt_pk_u8_f32(v, 0, 0u);
  unsigned b;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_pk_u8_f32 %0, %1, 0, 0\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(b) : "v"(v));
  out_rne[i] = a; out_rtz[i] = b;
  fma_after[i] = __builtin_fmaf(v, 1.140f, 0.3f);   // must be computed in nearest-even again
}
int main() {
  const int n = 1 << 16;
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) h[i] = -8.0f + 272.0f * (float)i / n + 0.000123f * (i % 7);
  h[0] = 0.5f; h[1] = 1.5f; h[2] = 2.5f; h[3] = 254.5f; h[4] = 255.5f; h[5] = 255.999f; h[6] = -0.5f; h[7] = 1e9f; h[8] = NAN; h[9] = 0.999999f;
  float *d, *df; unsigned *da, *db;
  hipMalloc(&d, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&df, n * 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, da, db, df, n);
  unsigned* a = (unsigned*)malloc(n * 4); unsigned* b = (unsigned*)malloc(n * 4); float* f = (float*)malloc(n * 4);
  hipMemcpy(a, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, db, n * 4, hipMemcpyDeviceToHost); hipMemcpy(f, df, n * 4, hipMemcpyDeviceToHost);
  int bad_rne = 0, bad_rtz = 0, bad_fma = 0;
  for (int i = 0; i < n; ++i) {
    float v = h[i];
    float r = nearbyintf(v); r = !(r > 0) ? 0 : (r > 255 ? 255 : r);
    float t = truncf(v); t = !(t > 0) ? 0 : (t > 255 ? 255 : t);
    if (a[i] != (unsigned)r) bad_rne++;
    if (b[i] != (unsigned)t) { if (bad_rtz < 5) printf("rtz mismatch v=%g got %u want %g\n", v, b[i], t); bad_rtz++; }
    if (f[i] != fmaf(v, 1.140f, 0.3f) && !(v != v)) bad_fma++;
  }
  printf("first: "); for (int i = 0; i < 10; ++i) printf("[%g -> rne %u rtz %u] ", h[i], a[i], b[i]); printf("\n");
  printf("mismatches: rne %d rtz %d fma_after %d of %d\n", bad_rne, bad_rtz, bad_fma, n);
  return 0;
}
