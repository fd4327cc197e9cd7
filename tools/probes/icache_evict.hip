// 160 KiB of straight-line VALU code per wavefront: run on every CU, it evicts the 64 KiB instruction cache a CU pair shares, so
// that the next kernel fetches its code cold, as the first launch of a process does (tools/first_step_hunt.py).
//   hipcc --offload-arch=gfx950 -O2 -fPIC -shared tools/probes/icache_evict.hip -o var/libicache_evict.so
#include <hip/hip_runtime.h>

__global__ void icache_evict_kernel(unsigned* out) {
  unsigned x = threadIdx.x, y = blockIdx.x;
  asm volatile(".rept 40960\n v_add_u32 %0, %0, %1\n .endr" : "+v"(x) : "v"(y));
  if (x == 0xFFFFFFFFu) out[0] = x;          // never true for the launched shapes; keeps the chain alive
}

extern "C" int icache_evict(void* stream, unsigned* out, int groups) {
  hipLaunchKernelGGL(icache_evict_kernel, dim3(groups), dim3(64), 0, (hipStream_t)stream, out);
  return (int)hipGetLastError();
}
