// Probe: can one 64-thread workgroup use all 163,840 B of static LDS as lane-private float4 columns?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SLOTS>
__global__ __launch_bounds__(64, 1) void probe(int* bad, int rounds) {
  __shared__ float4 lds[SLOTS * 64];
  int lane = threadIdx.x;
  int nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    for (int s = 0; s < SLOTS; ++s) lds[s * 64 + lane] = make_float4(s + r, lane, blockIdx.x, 1.f);
    for (int s = 0; s < SLOTS; ++s) {
      float4 v = lds[s * 64 + lane];
      if (v.x != (float)(s + r) || v.y != (float)lane || v.z != (float)blockIdx.x) nbad++;
      float* f = reinterpret_cast<float*>(&lds[s * 64 + lane]);
      f[3] = 2.f;
      if (lds[s * 64 + lane].w != 2.f) nbad++;
    }
  }
  atomicAdd(bad, nbad);
}
int main() {
  int* d; hipMalloc(&d, 4);
  for (int slots : {156, 160}) {
    hipMemset(d, 0, 4);
    if (slots == 156) hipLaunchKernelGGL(probe<156>, dim3(256), dim3(64), 0, 0, d, 4);
    else hipLaunchKernelGGL(probe<160>, dim3(256), dim3(64), 0, 0, d, 4);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipDeviceSynchronize();
    int h = -1; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("slots %d launch=%s sync=%s bad=%d\n", slots, hipGetErrorString(e), hipGetErrorString(e2), h);
  }
  return 0;
}
