// What v_pk_mov_b32 / v_pk_fma_f32 do with op_sel / op_sel_hi on gfx950 (one lane, printed).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k(float* out) {
  v2f a = {1.0f, 2.0f}, b = {10.0f, 20.0f}, r;
  int i = 0;
#define MOV(mods) asm volatile("v_pk_mov_b32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b)); out[i++] = r.x; out[i++] = r.y;
  MOV("")
  MOV("op_sel:[0,0] op_sel_hi:[0,0]")
  MOV("op_sel:[1,0] op_sel_hi:[0,0]")
  MOV("op_sel:[0,1] op_sel_hi:[0,0]")
  MOV("op_sel:[1,1] op_sel_hi:[0,0]")
  MOV("op_sel:[0,0] op_sel_hi:[1,1]")
  MOV("op_sel:[0,0] op_sel_hi:[0,1]")
  MOV("op_sel:[0,0] op_sel_hi:[1,0]")
  MOV("op_sel:[1,0] op_sel_hi:[0,1]")
  MOV("op_sel:[0,1] op_sel_hi:[1,0]")
  MOV("op_sel:[1,1] op_sel_hi:[1,1]")
  v2f w = {3.0f, 5.0f}, t = {1.0f, 100.0f}, c = {0.5f, 0.25f};
#define FMA(mods) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 " mods : "=v"(r) : "v"(w), "v"(t), "v"(c)); out[i++] = r.x; out[i++] = r.y;
  FMA("")
  FMA("op_sel_hi:[0,1,1]")
  FMA("op_sel:[1,0,0] op_sel_hi:[1,1,1]")
  FMA("op_sel:[1,0,0] op_sel_hi:[0,1,1]")
  asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[0,1,0]" : "=v"(r) : "v"(w), "v"(t)); out[i++] = r.x; out[i++] = r.y;
  asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(w), "v"(t)); out[i++] = r.x; out[i++] = r.y;
  // dependent back-to-back packed ops without any nop: is the RAW interlocked?
  v2f acc = {0.0f, 0.0f};
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_add_f32 %0, %0, %0" : "+v"(acc) : "v"(w), "v"(t));
  out[i++] = acc.x; out[i++] = acc.y; // expect 2*(3*3*1)=18, 2*(3*5*100)=3000
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); hipMemset(d, 0, 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d); hipDeviceSynchronize();
  float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  const char* names[] = {"mov default","mov s[0,0] h[0,0]","mov s[1,0] h[0,0]","mov s[0,1] h[0,0]","mov s[1,1] h[0,0]","mov s[0,0] h[1,1]","mov s[0,0] h[0,1]","mov s[0,0] h[1,0]","mov s[1,0] h[0,1]","mov s[0,1] h[1,0]","mov s[1,1] h[1,1]",
    "fma default (3*1+.5, 5*100+.25)","fma h[0,1,1] (3*1+.5, 3*100+.25)","fma s[1,0,0] h[1,1,1] (5*1+.5,5*100+.25)","fma s[1,0,0] h[0,1,1]","fma c=0 h[0,1,0] (3, 300)","fma c=0 s[1..] (5, 500)","dependent chain (18, 3000)"};
  for (int i = 0; i < 18; ++i) printf("%-45s -> (%g, %g)\n", names[i], h[2*i], h[2*i+1]);
  return 0;
}
