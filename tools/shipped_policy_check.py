#!/usr/bin/env python3
"""Informational: roll the reference's SHIPPED Walker3D policy in this repository's env (CPU oracle
backend, this container only).  The physics here is our own spec, so failure to walk is expected and is not a bug;
the check only tells whether obs/action conventions are in the same ballpark.  Weights are read with a restricted
unpickler (no reference code is executed); nothing is written into the repository.

  python tools/shipped_policy_check.py [/root/reference/playground/models/mocca_envs:Walker3DStepperEnv-v0_latest.pt]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


from steppingstone_amd.legacy_checkpoint import read_legacy, tensors_of  # noqa: E402


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/playground/models/mocca_envs:Walker3DStepperEnv-v0_latest.pt"
    obj, storages = read_legacy(path)
    w = tensors_of(obj, storages)
    from oracle_backend import OracleBackend
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    ac = ppo.Actor()
    ac.load_state_dict({k[len("actor."):]: torch.from_numpy(v) for k, v in w.items() if k.startswith("actor.")})
    n = 16
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=0, return_numpy=False, backend=OracleBackend(0, n, 0))
    obs = env.reset()
    lens, rets, reached = [], [], []
    for t in range(600):
        with torch.no_grad():
            a = ac(obs)
        obs, r, d, info = env.step(a)
        for i in torch.nonzero(d).flatten().tolist():
            lens.append(float(info["ep_len"][i])); rets.append(float(info["ep_ret"][i])); reached.append(int(info["steps_reached"][i]))
    print("shipped policy %s in OUR env: %d episodes, mean length %.1f steps, mean return %.1f, mean stones reached %.2f"
          % (os.path.basename(path), len(lens), np.mean(lens), np.mean(rets), np.mean(reached)))
    env2 = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=0, return_numpy=False, backend=OracleBackend(0, n, 0))
    env2.reset()
    lens2 = []
    for t in range(600):
        _, _, d, info = env2.step(env2.random_actions(t))
        lens2 += [float(info["ep_len"][i]) for i in torch.nonzero(d).flatten().tolist()]
    print("random actions for comparison: mean episode length %.1f steps" % np.mean(lens2))


if __name__ == "__main__":
    main()
