// Probe: per-instruction cost of the f32 VALU forms the step kernel is made of, ONE wave per SIMD (its occupancy).
// Eight independent accumulators per form; inline asm so the compiler cannot change the form.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64, 1) void chain(float* out, int iters, float sb, unsigned long long* cyc) {
  constexpr int ILP = 8;
  float a[ILP], e[ILP];
  for (int i = 0; i < ILP; ++i) { a[i] = threadIdx.x * 0.001f + i; e[i] = 0.f; }
  float b = 1.0001f + threadIdx.x * 1e-6f, c = 0.0001f + threadIdx.x * 1e-7f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));       // 3 VGPR sources
        if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));          // VOP2, dst is the addend
        if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));               // 2 distinct VGPRs
        if (MODE == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));      // one SGPR source
        if (MODE == 4) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(a[i]) : "v"(c));              // inline constant
        if (MODE == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (MODE == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (MODE == 7) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e[i]) : "v"(a[i]), "v"(b), "v"(c));   // dst distinct
        if (MODE == 8) asm volatile("v_fmac_f32 %0, 0x3f8ccccd, %1" : "+v"(a[i]) : "v"(c));         // literal operand
        if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(e[i]) : "v"(a[i]));
        if (MODE == 10) asm volatile("v_accvgpr_write_b32 a0, %0" :: "v"(a[i]) : "a0");
        if (MODE == 11) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(e[i]) : "v"(a[i]));
        if (MODE == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : );
        if (MODE == 13) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < ILP; ++i) s += a[i] + e[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(float* d, unsigned long long* dc, const char* name) {
  int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((chain<MODE>), dim3(128), dim3(64), 0, 0, d, iters, 1.0001f, dc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chain<MODE>), dim3(128), dim3(64), 0, 0, d, iters, 1.0001f, dc);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
  double n = (double)iters * 16 * 8;
  printf("%-34s %.3f ns/instr  %.2f memtime-ticks/instr  (%.3f ticks/ns)\n", name, ms * 1e6 / n, (double)c / n, (double)c / (ms * 1e6));
}
int main() {
  float* d; unsigned long long* dc; hipMalloc(&d, 1 << 22); hipMalloc(&dc, 8);
  run<0>(d, dc, "v_fma_f32 a,a,b,c (3 VGPR)"); run<1>(d, dc, "v_fmac_f32 a,b,c (VOP2)"); run<2>(d, dc, "v_fma_f32 a,a,b,b");
  run<3>(d, dc, "v_fma_f32 a,a,s,c (SGPR)"); run<4>(d, dc, "v_fma_f32 a,a,1.0,c"); run<5>(d, dc, "v_add_f32"); run<6>(d, dc, "v_mul_f32");
  run<7>(d, dc, "v_fma_f32 e,a,b,c (dst distinct)"); run<8>(d, dc, "v_fmac_f32 a,lit,c"); run<9>(d, dc, "v_mov_b32"); run<10>(d, dc, "v_accvgpr_write_b32");
  run<11>(d, dc, "v_mov_b32_dpp quad_perm"); run<12>(d, dc, "v_cndmask_b32"); run<13>(d, dc, "v_max_f32");
  return 0;
}
