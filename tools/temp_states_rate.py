"""HBM rate of create_temp_states (N x 121 x 60 f32 written per call): the one kernel on the path that IS HBM-bound."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
    for n in (4096, 32768):
        env = SteppingStoneVecEnv(env_id, n, seed=0, device="cuda:0", return_numpy=False)
        env.update_curriculum(5)
        env.reset()
        env.rollout_random(5, 0)
        for _ in range(3):
            out = env.create_temp_states()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        K = 20
        e0.record()
        for _ in range(K):
            out = env.create_temp_states()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        by = n * 121 * 60 * 4
        print("%s N %6d: %.3f ms/call, %.0f GB/s written (%.1f%% of 8 TB/s)" % (env_id, n, ms, by / ms / 1e6, 100 * by / ms / 1e6 / 8000), flush=True)
        env.close()

# reference points: a plain fill and a copy of the same size (what the memory system gives a pure write stream)
for n in (4096, 32768):
    x = torch.empty((n, 121, 60), device="cuda:0"); y = torch.empty_like(x)
    for name, fn, mult in (("fill", lambda: x.fill_(1.0), 1), ("copy", lambda: y.copy_(x), 2)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("torch %s N %6d: %.3f ms, %.0f GB/s (read+write)" % (name, n, ms, mult * x.numel() * 4 / ms / 1e6), flush=True)
