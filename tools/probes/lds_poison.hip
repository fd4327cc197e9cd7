// Fills the LDS of every CU with a pattern (160 KiB per workgroup, many workgroups), so that the NEXT kernel starts on known
// garbage: a kernel that reads an LDS word before writing it then produces pattern-dependent results (tools/lds_poison_check.py).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/lds_poison.hip -o var/liblds_poison.so
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ __launch_bounds__(256) void poison_kernel(uint32_t pat, uint32_t mix, uint32_t* sink) {
  __shared__ uint32_t l[40960];
  for (int i = threadIdx.x; i < 40960; i += 256) l[i] = pat ^ (mix * (uint32_t)i);
  __syncthreads();
  uint32_t v = l[(threadIdx.x * 97 + blockIdx.x) % 40960];
  if (v == 0x9e3779b1u && mix == 0xffffffffu) *sink = v;     // keeps the stores alive
}
extern "C" int lds_poison(uint32_t pat, uint32_t mix, void* sink, void* stream) {
  hipLaunchKernelGGL(poison_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, pat, mix, (uint32_t*)sink);
  return (int)hipGetLastError();
}
