"""Workload for the HBM-traffic PMC passes.  Run under rocprofv3 with ONE of --pmc FETCH_SIZE / --pmc WRITE_SIZE:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/hbm_traffic.py [N [S]]
Launches (a) the calibration copy (known bytes: n*4 read, n*4 written, dword per lane like the step kernel) and
(b) 20 single-step launches (ss::step_kernel*) and 4 launches of S control steps each (ss::rollout_kernel*; S = argv[2], default 250) at N envs.
tools/hbm_traffic_report.py turns the two counter CSVs into bytes per launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd import _lib
from steppingstone_amd.envs import SteppingStoneVecEnv
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lib = _lib.load()
n = 256 * 1024 * 1024 // 4 * 2       # 512 MiB in, 512 MiB out: beyond the 256 MiB infinity cache
a = torch.zeros(n, device="cuda")
b = torch.empty(n, device="cuda")
for _ in range(3):
    lib.ss_debug_calib_copy(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n, None)
torch.cuda.synchronize()
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n_envs, seed=0, device="cuda:0")
env.reset()
STEPS_PER_LAUNCH = int(sys.argv[2]) if len(sys.argv) > 2 else 250
env.rollout_random(20, 0, steps_per_launch=1)
env.rollout_random(4 * STEPS_PER_LAUNCH, 20, steps_per_launch=STEPS_PER_LAUNCH)
torch.cuda.synchronize()
