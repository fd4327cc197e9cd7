"""Bit-level fingerprint of what a library build computes (run on the GPU box): for both robots and an env count per launch
variant (plain, one helper, three helpers), 60 random-action steps at curriculum 5 through the K-step kernel; prints a SHA-256
of every step's observations / rewards / dones and of the final state.  Two builds that print the same lines compute the same
bits:  STEPPINGSTONE_LIB=var/libss_x.so python tools/state_hash.py"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv

for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
    for n in (4096, 16384, 40000):
        e = SteppingStoneVecEnv(env_id, n, seed=11, device="cuda:0", return_numpy=False)
        e.update_curriculum(5)
        e.reset()
        h = hashlib.sha256()
        for t0 in (0, 20, 40):
            for x in e.rollout_random(20, t0=t0, steps_per_launch=20):
                h.update(x.cpu().numpy().tobytes())
        h.update(e.get_state().cpu().numpy().tobytes())
        print(env_id, n, h.hexdigest()[:24], flush=True)
        e.close()
