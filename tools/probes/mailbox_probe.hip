// Probe: round trip of an action mailbox between a stream of small kernels and a PERSISTENT kernel on another stream (DESIGN.md
// section 7: the persistent-env-kernel candidate for the policy-in-the-loop path).  128 workgroups of 256 threads spin on `act_seq`
// (agent-scope acquire loads, bounded), do `work` dependent FMAs, fence, count themselves; the last one publishes `obs_seq`.  The
// consumer stream alternates a one-thread "post" kernel (act_seq = k) and a one-thread "wait" kernel (spin until obs_seq >= k).
// Prints the time per round trip for work = 0 (pure handshake) and for a ~50 us body, next to back-to-back launches of a kernel
// with the same body (one launch per step).   hipcc --offload-arch=gfx950 -O2 mailbox_probe.hip -o /tmp/mailbox_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Mbox { unsigned act_seq, obs_seq, stop, error, done_count, pad[11]; };

__device__ float body(float x, int work) {
  for (int i = 0; i < work; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
  return x;
}
__global__ __launch_bounds__(256) void persist(Mbox* mb, float* out, int work, int max_steps, long long spin_limit) {
  __shared__ int go;
  float x = threadIdx.x;
  for (int k = 1; k <= max_steps; ++k) {
    if (threadIdx.x == 0) {
      int g = 0;
      for (long long it = 0; it < spin_limit; ++it) {
        unsigned v = __hip_atomic_load(&mb->act_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - (unsigned)k) >= 0) { g = 1; break; }
        if (__hip_atomic_load(&mb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!g && !__hip_atomic_load(&mb->stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) mb->error = 1u;
      go = g;
    }
    __syncthreads();
    if (!go) return;
    x = body(x, work);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned prev = atomicAdd(&mb->done_count, 1u);
      if ((prev + 1u) % gridDim.x == 0u) __hip_atomic_store(&mb->obs_seq, (unsigned)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
__global__ void post(Mbox* mb, unsigned k) { __hip_atomic_store(&mb->act_seq, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void stop(Mbox* mb) { __hip_atomic_store(&mb->stop, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void wait(Mbox* mb, unsigned k, long long spin_limit) {
  for (long long it = 0; it < spin_limit; ++it) {
    unsigned v = __hip_atomic_load(&mb->obs_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(v - k) >= 0) return;
    __builtin_amdgcn_s_sleep(1);
  }
  mb->error = 2u;
}
__global__ __launch_bounds__(256) void one_step(float* out, int work) {
  out[blockIdx.x * blockDim.x + threadIdx.x] = body((float)threadIdx.x, work);
}

int main() {
  Mbox* mb; float* out;
  CK(hipMalloc(&mb, sizeof(Mbox))); CK(hipMalloc(&out, 128 * 256 * 4));
  hipStream_t sp, sc;
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  const long long limit = 1ll << 21;                      // bounded spins: ~1 s
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int work : {0, 12000}) {
    const int steps = 2000;
    CK(hipMemset(mb, 0, sizeof(Mbox)));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(persist, dim3(128), dim3(256), 0, sp, mb, out, work, steps + 8, limit);
    for (int k = 1; k <= 8; ++k) { hipLaunchKernelGGL(post, dim3(1), dim3(1), 0, sc, mb, (unsigned)k); hipLaunchKernelGGL(wait, dim3(1), dim3(1), 0, sc, mb, (unsigned)k, limit); }
    CK(hipStreamSynchronize(sc));
    CK(hipEventRecord(e0, sc));
    for (int k = 9; k <= steps + 8; ++k) { hipLaunchKernelGGL(post, dim3(1), dim3(1), 0, sc, mb, (unsigned)k); hipLaunchKernelGGL(wait, dim3(1), dim3(1), 0, sc, mb, (unsigned)k, limit); }
    CK(hipEventRecord(e1, sc));
    CK(hipStreamSynchronize(sc));
    hipLaunchKernelGGL(stop, dim3(1), dim3(1), 0, sc, mb);
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Mbox h; CK(hipMemcpy(&h, mb, sizeof h, hipMemcpyDeviceToHost));
    // the same body with one launch per step
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(one_step, dim3(128), dim3(256), 0, sc, out, work);
    CK(hipEventRecord(e0, sc));
    for (int k = 0; k < steps; ++k) hipLaunchKernelGGL(one_step, dim3(128), dim3(256), 0, sc, out, work);
    CK(hipEventRecord(e1, sc));
    CK(hipStreamSynchronize(sc));
    float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
    printf("work %6d FMAs: mailbox round trip %.2f us per step (error word %u, obs_seq %u) | one launch per step %.2f us\n", work,
           1e3f * ms / steps, h.error, h.obs_seq, 1e3f * ms1 / steps);
  }
  return 0;
}
