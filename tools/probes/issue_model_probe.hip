// Issue model of one gfx950 SIMD for fp32 VALU work: the issue interval of DEPENDENT and INDEPENDENT chains of
// v_fmac_f32 (VOP2, 32-bit encoding), v_fma_f32 (VOP3, 64-bit encoding) and v_pk_fma_f32 (VOP3P) at 1, 2 and 4 wavefronts per SIMD,
// in shader clocks (s_memtime) calibrated against the 100 MHz s_memrealtime.  Answers: (a) is a SIMD 16 lanes (4 clocks per wave64
// instruction) or 32 lanes (2 clocks) wide for non-packed fp32; (b) does a second wavefront on the SIMD fill issue slots a single
// dependent chain leaves empty; (c) is a packed instruction twice the work in the same slot.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_model_probe.hip -o /tmp/issue_model_probe && /tmp/issue_model_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 128
#define ITERS 64
template <int MODE>
__global__ void probe(float* out, unsigned long long* cyc, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  f2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7};
  const float k = 0.999f, c = 0.001f;
  const f2 k2 = {k, k}, c2 = {c, c};
  __syncthreads();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (MODE == 0) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(k), "v"(c)); }                       // VOP2, dependent through dst
      if (MODE == 1) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(k), "v"(c)); }                    // VOP3, dependent
      if (MODE == 2) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(k2), "v"(c2)); }               // VOP3P, dependent
      if (MODE == 3) { asm volatile("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c)); }
      if (MODE == 4) { asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c)); }
      if (MODE == 5) { asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k2), "v"(c2)); }
      if (MODE == 6) { asm volatile("v_fmac_f32 %0, %2, %3\n v_fmac_f32 %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(k), "v"(c)); }   // two chains
      if (MODE == 7) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(k2), "v"(c2)); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, int per, int flops_per_instr) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16);
  printf("%-46s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;      // 4 SIMDs x waves_per_simd wavefronts in ONE workgroup on one CU (round-robin over the SIMDs)
    probe<MODE><<<64, threads>>>(out, cyc, 1.0f);
    probe<MODE><<<64, threads>>>(out, cyc, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    const double n = (double)ITERS * REP * per;
    const double clk = (double)h[0] / n;                   // shader clocks per instruction of ONE wavefront
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);   // s_memrealtime ticks at 100 MHz
    // per SIMD: waves_per_simd wavefronts each retire one instruction per `clk` clocks
    printf(" | %dw/SIMD %5.2f clk/instr/wave = %5.2f clk/instr/SIMD, %5.1f flop/clk/SIMD (%.2f GHz)", waves_per_simd, clk, clk / waves_per_simd,
           64.0 * flops_per_instr * waves_per_simd / clk, ghz);
  }
  printf("\n");
  hipFree(out); hipFree(cyc);
}
int main() {
  printf("# one workgroup per CU; clk = s_memtime ticks; peak fp32 vector rate of the chip = 64 flop/clk/SIMD (157.3 TFLOP/s / 1024 SIMDs / 2.4 GHz)\n");
  run<0>("v_fmac_f32 (VOP2), one dependent chain", 1, 2);
  run<1>("v_fma_f32 (VOP3), one dependent chain", 1, 2);
  run<2>("v_pk_fma_f32, one dependent chain", 1, 4);
  run<6>("v_fmac_f32, two independent chains", 2, 2);
  run<7>("v_pk_fma_f32, two independent chains", 2, 4);
  run<3>("v_fmac_f32, four independent chains", 4, 2);
  run<4>("v_fma_f32 (VOP3), four independent chains", 4, 2);
  run<5>("v_pk_fma_f32, four independent chains", 4, 4);
  return 0;
}
