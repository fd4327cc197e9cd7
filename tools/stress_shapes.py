"""Shape stress for the launch variants (run on the GPU box): ragged env counts around every variant boundary x steps per launch;
the K-step launch must leave the same bits as K single-step launches (observations of every step, final state), for both robots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
bad = 0
for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
    for n in (1, 2, 31, 33, 200, 4095, 4097, 8191, 8193, 16383, 16385, 20000):
        for K in (1, 2, 3, 7, 40):
            a = SteppingStoneVecEnv(env_id, n, seed=3, device="cuda:0", return_numpy=False)
            b = SteppingStoneVecEnv(env_id, n, seed=3, device="cuda:0", return_numpy=False)
            for e in (a, b):
                e.update_curriculum(5)
                e.reset()
            oa = a.rollout_random(K, t0=5, steps_per_launch=K)
            ob = b.rollout_random(K, t0=5, steps_per_launch=1)
            same = all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(a.get_state(), b.get_state())
            # and a second launch continuing from there (state handed over through HBM)
            oa = a.rollout_random(3, t0=5 + K, steps_per_launch=3)
            ob = b.rollout_random(3, t0=5 + K, steps_per_launch=1)
            same = same and all(torch.equal(x, y) for x, y in zip(oa, ob)) and torch.equal(a.get_state(), b.get_state())
            bad += 0 if same else 1
            if not same:
                print("MISMATCH", env_id, n, K, flush=True)
            a.close(); b.close()
    print(env_id, "done", flush=True)
print("mismatches:", bad)
