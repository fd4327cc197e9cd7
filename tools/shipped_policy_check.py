#!/usr/bin/env python3
"""Informational: roll the reference's SHIPPED Walker3D policy in this repository's env (CPU oracle
backend, this container only).  The physics here is our own spec, so failure to walk is expected and is not a bug;
the check only tells whether obs/action conventions are in the same ballpark.  Weights are read with a restricted
unpickler (no reference code is executed); nothing is written into the repository.

  python tools/shipped_policy_check.py [/root/reference/playground/models/mocca_envs:Walker3DStepperEnv-v0_latest.pt]
"""
import collections
import os
import pickle
import struct
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


class _Stub:
    def __init__(self, *a, **k):
        pass

    def __setstate__(self, st):
        self.__dict__.update(st if isinstance(st, dict) else {})


def _rebuild_tensor_v2(storage, offset, size, stride, *rest):
    return ("tensor", storage, offset, tuple(size), tuple(stride))


def _rebuild_parameter(data, requires_grad, hooks):
    return data


class _U(pickle.Unpickler):
    def find_class(self, mod, name):
        if (mod, name) == ("collections", "OrderedDict"):
            return collections.OrderedDict
        if (mod, name) == ("torch._utils", "_rebuild_tensor_v2"):
            return _rebuild_tensor_v2
        if (mod, name) == ("torch._utils", "_rebuild_parameter"):
            return _rebuild_parameter
        if mod == "torch" and name.endswith("Storage"):
            return name
        return _Stub          # model classes / backends -> inert stubs

    def persistent_load(self, pid):
        if pid[0] == "module":
            return pid[1]
        if pid[0] == "storage":
            return ("storage", pid[2], pid[4])        # key, numel
        raise pickle.UnpicklingError(pid)


def read_legacy(path):
    f = open(path, "rb")
    for _ in range(3):
        pickle.load(f)                                # magic, protocol, sys info
    obj = _U(f).load()
    keys = pickle.load(f)
    storages = {}
    for k in keys:
        n = struct.unpack("<q", f.read(8))[0]
        storages[k] = np.frombuffer(f.read(4 * n), dtype="<f4").copy()
    return obj, storages


def tensors_of(obj, storages, prefix="", out=None):
    out = {} if out is None else out
    d = getattr(obj, "__dict__", {})
    for group in ("_parameters", "_buffers"):
        for k, v in (d.get(group) or {}).items():
            if isinstance(v, tuple) and v and v[0] == "tensor":
                _, st, off, size, stride = v
                flat = storages[st[1]]
                out[prefix + k] = np.lib.stride_tricks.as_strided(flat[off:], size, [s * 4 for s in stride]).copy()
    for k, m in (d.get("_modules") or {}).items():
        tensors_of(m, storages, prefix + k + ".", out)
    return out


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
