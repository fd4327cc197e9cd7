// Probe: does packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) double the f32 rate of ONE wave per SIMD
// (the step kernel's occupancy)?  Independent chains, inline asm so the compiler cannot re-pack or fuse.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE, int ILP>
__global__ __launch_bounds__(64, 1) void chain(float* out, int iters) {
  f2 a[ILP];
  for (int i = 0; i < ILP; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f - i};
  f2 b = {1.0001f, 0.9999f}, c = {0.0001f, -0.0001f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(b.x), "v"(c.x)); }
        if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c)); }
        if (MODE == 2) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
        if (MODE == 3) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c)); }
        if (MODE == 4) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(b.x)); }
      }
  }
  float s = 0; for (int i = 0; i < ILP; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE, int ILP> void run(float* d, const char* name, int threads = 64) {
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chain<MODE, ILP>), dim3(128), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain<MODE, ILP>), dim3(128), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n = (double)iters * 16 * ILP;
  printf("%-14s ILP %2d thr %2d: %.3f ns/instr\n", name, ILP, threads, ms * 1e6 / n);
}
int main() {
  float* d; hipMalloc(&d, 1 << 22);
  run<0, 1>(d, "v_fma_f32"); run<0, 4>(d, "v_fma_f32"); run<0, 8>(d, "v_fma_f32");
  run<4, 8>(d, "v_mul_f32");
  run<1, 1>(d, "v_pk_fma_f32"); run<1, 4>(d, "v_pk_fma_f32"); run<1, 8>(d, "v_pk_fma_f32");
  run<2, 8>(d, "v_pk_mul_f32"); run<3, 8>(d, "v_pk_add_f32");
  // partially filled waves: does the SIMD skip empty 16-lane passes?
  run<4, 8>(d, "v_mul_f32", 32); run<4, 8>(d, "v_mul_f32", 16); run<0, 8>(d, "v_fma_f32", 32); run<0, 8>(d, "v_fma_f32", 16);
  return 0;
}
