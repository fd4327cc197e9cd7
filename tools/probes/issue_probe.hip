// Probe: VALU issue rate of ONE wave per SIMD (64-thread workgroups, one per CU): dependent vs independent v_fma_f32
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64, 1) void fma_chain(float* out, int iters, unsigned long long* cyc) {
  float a[ILP];
  for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 0.001f + i;
  float b = 1.0001f, c = 0.0001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], b, c);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < ILP; ++i) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int ILP> void run(float* d, unsigned long long* dc, int blocks, int threads) {
  int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(fma_chain<ILP>, dim3(blocks), dim3(threads), 0, 0, d, iters, dc);
  hipEventRecord(e0);
  hipLaunchKernelGGL(fma_chain<ILP>, dim3(blocks), dim3(threads), 0, 0, d, iters, dc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * ILP;
  printf("ILP %2d blocks %4d thr %3d: %.3f ms, %.2f ns/instr, memtime %.2f ticks/instr\n", ILP, blocks, threads, ms, ms * 1e6 / n, (double)c / n);
}
int main() {
  float* d; unsigned long long* dc; hipMalloc(&d, 1 << 22); hipMalloc(&dc, 8);
  run<1>(d, dc, 128, 64); run<2>(d, dc, 128, 64); run<4>(d, dc, 128, 64); run<8>(d, dc, 128, 64); run<16>(d, dc, 128, 64);
  run<8>(d, dc, 128, 128); run<8>(d, dc, 128, 256); run<8>(d, dc, 128, 512);
  return 0;
}
