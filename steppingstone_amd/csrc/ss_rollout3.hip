// ss_rollout3.hip -- the three-helper rollout kernel (BASELINE's 4096 envs per GPU) as its own translation unit.
//
// Why: machine-scheduling strategy is a per-translation-unit compiler option.  The max-ILP strategy is worth 7.5 % on the plain
// kernel (32768 envs: 0.0683 vs 0.0734 ms/step) and 3.5 % on the single-helper one, but costs the three-helper ROLLOUT kernel 1.7 %
// (0.0498 vs 0.0490 ms/step at 4096 envs, three interleaved runs each; the one-launch-per-step kernel does not care).  Scheduling
// reorders independent instructions only: the values are the same bits whichever strategy compiled a kernel (the bitwise tests of
// tests/test_gpu_branches.py compare this kernel with the others).  steppingstone_amd/build.py compiles this file WITHOUT
// -amdgpu-sched-strategy=max-ilp and everything else (ss_api.hip) with it; ss_api.hip declares these two instantiations extern.
//
// Round 6, the second per-unit choice: here all three helpers evaluate four cos / sin pairs each (SS_CS_PER_HELPER = 4) instead of
// helpers 1 and 2 six each -- 0.0461 against 0.0465 ms/step for this kernel, while the one-launch-per-step kernel of ss_api.hip
// LOSES 2 % with it (0.0595 against 0.0583; profiles/r06_ab_disc_vs_plank_variants.txt).  Which helper evaluates a joint's cos / sin
// changes no value: ss_sincos is the same function of the same angle.
#include <hip/hip_runtime.h>

#define SS_CS_PER_HELPER 4
#include "ss_kernels.hpp"

template __global__ void ss::rollout_kernel_helped<ss::ModelWalker3D, 3>(ss::Params, ss::StepIO);
template __global__ void ss::rollout_kernel_helped<ss::ModelMike, 3>(ss::Params, ss::StepIO);
