"""Throughput vs number of envs on one GPU (random-action rollout, events on the launch stream): the multi-step rollout
kernel (300 control steps per launch) and the one-launch-per-step path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
for n in [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768, 65536, 131072]:
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=0, device="cuda:0", return_numpy=False)
    env.reset()
    env.rollout_random(100, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 300
    e0.record(); env.rollout_random(K, 100, steps_per_launch=K); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    e0.record(); env.rollout_random(K, 400, steps_per_launch=1); e1.record(); torch.cuda.synchronize()
    ms1 = e0.elapsed_time(e1) / K
    print("N %7d  rollout kernel %.4f ms/step  %.1f M env-steps/s | one launch per step %.4f ms/step  %.1f M env-steps/s"
          % (n, ms, n / ms / 1e3, ms1, n / ms1 / 1e3), flush=True)
    env.close()
