#!/usr/bin/env python3
"""tests/golden/shipped_actor_<kind>.npz: the ACTOR weights (six Linear layers, float32) of the reference's shipped policies
`playground/models/mocca_envs:{Walker3D,Mike}StepperEnv-v0_latest.pt`, read in THIS container with the restricted unpickler of
steppingstone_amd/legacy_checkpoint.py (no reference code executed) and stored as plain arrays -- data only: the pickled files also
carry the source text of the reference's classes, which is NOT copied.  They let the GPU box (where /root/reference does not exist)
run the one reference-held behavioural check of the env: the shipped deterministic policy walks the stepping-stone course
(playground/enjoy.py:143-235; tests/test_gpu_shipped_policy.py, tests/test_shipped_policy_walks.py).

  python tools/make_golden_policy.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from steppingstone_amd import legacy_checkpoint as lc  # noqa: E402

MODELS = "/root/reference/playground/models/"
FILES = {"walker3d": "mocca_envs:Walker3DStepperEnv-v0_latest.pt", "mike": "mocca_envs:MikeStepperEnv-v0_latest.pt",
         # round 6: the reference's OTHER Walker3D actor -- the flat-terrain policy its curriculum runs start from (playground/train.py:148-153)
         "walker3d_base": "mocca_envs:Walker3DStepperEnv-v0_base.pt"}

for kind, f in FILES.items():
    obj, storages = lc.read_legacy(MODELS + f)
    w = lc.tensors_of(obj, storages)
    out = {k[len("actor."):]: np.ascontiguousarray(v, np.float32) for k, v in w.items() if k.startswith("actor.")}
    assert sorted(out) == sorted("%s.%s" % (l, p) for l in ("fc1", "fc2", "fc3", "fc4", "fc5", "out") for p in ("weight", "bias")), sorted(out)
    assert out["fc1.weight"].shape == (256, 60) and out["out.weight"].shape == (21, 256)
    path = os.path.join(ROOT, "tests", "golden", "shipped_actor_%s.npz" % kind)
    np.savez_compressed(path, source=np.array(f), **out)
    print(path, os.path.getsize(path), "bytes,", sum(v.size for v in out.values()), "parameters")
