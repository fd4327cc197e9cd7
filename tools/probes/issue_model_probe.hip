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
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
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
      if (MODE == 8) { asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k), "v"(c)); }
      if (MODE == 7) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(k2), "v"(c2)); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y;
  // every wavefront of block 0 reports its own window: the SIMD's rate is all its wavefronts' work over the SPAN from the first start to
  // the last end (the arbiter favours the oldest wavefront, so one wavefront's own elapsed time would flatter a shared SIMD)
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) {
    const int w = threadIdx.x >> 6;
    cyc[4 * w + 0] = t0; cyc[4 * w + 1] = t1; cyc[4 * w + 2] = r0; cyc[4 * w + 3] = r1;
  }
}
template <int MODE>
void run(const char* name, int per, int flops_per_instr) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 16 * 4 * 8);
  printf("%-46s", name);
  for (int waves_per_simd : {1, 2, 4}) {
    const int threads = 256 * waves_per_simd;      // 4 SIMDs x waves_per_simd wavefronts in ONE workgroup on one CU (round-robin over the SIMDs)
    const int waves = threads / 64;
    probe<MODE><<<64, threads>>>(out, cyc, 1.0f);
    probe<MODE><<<64, threads>>>(out, cyc, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[64]; hipMemcpy(h, cyc, waves * 32, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
    double own = 0;
    for (int w = 0; w < waves; ++w) {
      if (h[4 * w] < t0) t0 = h[4 * w];
      if (h[4 * w + 1] > t1) t1 = h[4 * w + 1];
      if (h[4 * w + 2] < r0) r0 = h[4 * w + 2];
      if (h[4 * w + 3] > r1) r1 = h[4 * w + 3];
      own += (double)(h[4 * w + 1] - h[4 * w]) / waves;
    }
    const double n = (double)ITERS * REP * per;            // instructions of ONE wavefront
    const double span = (double)(t1 - t0);
    const double ghz = span / ((double)(r1 - r0) * 10.0);  // s_memrealtime ticks at 100 MHz
    // per SIMD: waves_per_simd wavefronts x n instructions within `span` clocks
    printf(" | %dw/SIMD: %5.2f clk/instr/SIMD (a wavefront's own window: %5.2f clk/instr), %5.1f flop/clk/SIMD, %.2f GHz", waves_per_simd,
           span / (n * waves_per_simd), own / n, 64.0 * flops_per_instr * waves_per_simd * n / span, ghz);
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
  run<8>("v_fmac_f32, eight independent chains", 8, 2);
  run<4>("v_fma_f32 (VOP3), four independent chains", 4, 2);
  run<5>("v_pk_fma_f32, four independent chains", 4, 4);
  return 0;
}
