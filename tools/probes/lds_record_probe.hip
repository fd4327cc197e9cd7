// LDS probe: cycles per ds_write_b128 / ds_read_b128 / ds_write_b32 of one wavefront (of 1 or 4 in the workgroup) with the env
// kernel's lane-private layouts, at several places of a 160 KiB allocation.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_record_probe.hip -o var/lds_record_probe && var/lds_record_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 64
template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int slot0, int active_waves) {
  __shared__ float4 lds[160 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float4 v = make_float4(lane, wave, 1.f, 2.f);
  float acc = 0.f;
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < active_waves) {
    for (int it = 0; it < 16; ++it) {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const int slot = slot0 + (k % 13) + wave * 13;
        if (MODE == 0) lds[slot * 64 + lane] = v;                                  // ds_write_b128, 16-B lane stride
        if (MODE == 1) { float4 r = lds[slot * 64 + lane]; acc += r.x + r.w; }     // ds_read_b128
        if (MODE == 2) reinterpret_cast<float*>(lds)[(slot * 4 + (k & 3)) * 64 + lane] = v.x;   // ds_write_b32, 4-B lane stride
        if (MODE == 3) { acc += reinterpret_cast<float*>(lds)[(slot * 4 + (k & 3)) * 64 + lane]; }
        v.x += 1.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc + lds[threadIdx.x].x;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE>
void run(const char* name, int slot0, int waves) {
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, 128 * 256 * 4); (void)hipMalloc(&cyc, 8);
  probe<MODE><<<128, 256>>>(out, cyc, slot0, waves);
  probe<MODE><<<128, 256>>>(out, cyc, slot0, waves);
  (void)hipDeviceSynchronize();
  unsigned long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-18s slot0 %3d  %d wave(s): %6.1f clocks / instruction\n", name, slot0, waves, (double)h / (16.0 * N));
}
int main() {
  for (int waves : {1, 4})
    for (int s0 : {0, 60, 100}) {
      run<0>("ds_write_b128", s0, waves);
      run<1>("ds_read_b128", s0, waves);
      run<2>("ds_write_b32", s0, waves);
      run<3>("ds_read_b32", s0, waves);
    }
  return 0;
}
