// Issue / latency probe for one wavefront per SIMD (the regime of the env kernel's main and helper wavefronts): cycles per
// instruction of dependent and independent chains of v_fma_f32 and v_pk_fma_f32, by the shader clock.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/chain_issue_probe.hip -o var/chain_issue_probe && var/chain_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE>
__global__ __launch_bounds__(64) void probe(float* out, unsigned long long* cyc, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  f2 p0 = {seed, seed + 1}, p1 = {seed + 2, seed + 3}, p2 = {seed + 4, seed + 5}, p3 = {seed + 6, seed + 7};
  const float k = 0.999f, c = 0.001f;
  const f2 k2 = {k, k}, c2 = {c, c};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; ++it) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (MODE == 0) { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(k), "v"(c)); }
      if (MODE == 1) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(k2), "v"(c2)); }
      if (MODE == 2) { asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(k), "v"(c)); }
      if (MODE == 3) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(p0), "+v"(p1) : "v"(k2), "v"(c2)); }
      if (MODE == 4) { asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                                    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(k2), "v"(c2)); }
      if (MODE == 5) { asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %4, %5" : "+v"(p0), "+v"(a0) : "v"(k2), "v"(c2), "v"(k), "v"(c)); }
      if (MODE == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p0) : "v"(k2)); }
      if (MODE == 7) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(c2)); }
      if (MODE == 8) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(k)); }
      if (MODE == 9) { asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k), "v"(c)); }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p3.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int per) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 64 * 4); hipMalloc(&cyc, 8);
  probe<MODE><<<128, 64>>>(out, cyc, 1.0f);
  probe<MODE><<<128, 64>>>(out, cyc, 1.0f);
  hipDeviceSynchronize();
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-58s %6.2f clocks / instruction\n", name, (double)h / (64.0 * REP * per));
}
int main() {
  run<0>("v_fma_f32, one dependent chain", 1);
  run<8>("v_mul_f32, one dependent chain", 1);
  run<2>("v_fma_f32, two independent chains", 2);
  run<9>("v_fma_f32, four independent chains", 4);
  run<1>("v_pk_fma_f32, one dependent chain", 1);
  run<6>("v_pk_mul_f32, one dependent chain", 1);
  run<7>("v_pk_add_f32, one dependent chain", 1);
  run<3>("v_pk_fma_f32, two independent chains", 2);
  run<4>("v_pk_fma_f32, four independent chains", 4);
  run<5>("v_pk_fma_f32 + v_fma_f32 alternating, independent", 2);
  return 0;
}
