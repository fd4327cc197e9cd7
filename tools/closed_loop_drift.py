"""Closed-loop drift of the implementations of docs/PHYSICS.md over a full 1000-step episode under the committed balance
controller (tests/controllers.py): fp32 CPU oracle vs fp64 CPU oracle, and -- when a GPU is present -- the HIP kernel vs
both.  Each implementation computes its actions from ITS OWN observations; the seeded action noise is shared.
   python tools/closed_loop_drift.py [walker3d|mike] [noise=0.05] [envs=32]
Prints, every 100 steps, the median / max |obs| distance between the implementations over the robots still alive in all
of them.  (Runs on the CPU-only container for the two oracle builds; docs/HISTORY.md section 3 quotes its output.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from controllers import balance_controller, standing_state  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "walker3d"
noise = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
envs = {"f32": ol.OracleEnv(kind, n, seed=3), "f64": ol.OracleEnv(kind, n, seed=3, prec="f64")}
try:
    import torch
    if torch.cuda.is_available():
        from steppingstone_amd.envs import SteppingStoneVecEnv
        envs["hip"] = SteppingStoneVecEnv({"walker3d": "Walker3DStepperEnv-v0", "mike": "MikeStepperEnv-v0"}[kind], n, seed=3,
                                          device="cuda:0", return_numpy=True)
except ImportError:
    pass
ctrl = balance_controller(kind)
for e in envs.values():
    e.reset()
st0 = standing_state(kind, envs["f32"].get_state())       # the closed-loop tests start standing (tests/controllers.py)
obs = {}
for k, e in envs.items():
    e.set_state(st0.astype(np.float64) if k == "f64" else st0)
    o_ = e.get_obs()
    obs[k] = o_.cpu().numpy() if hasattr(o_, "cpu") else np.array(o_)
alive = np.ones(n, bool)
rng = np.random.default_rng(0)
print("%s, %d envs, action noise %.3f, implementations: %s" % (kind, n, noise, ", ".join(envs)))
for t in range(999):
    z = rng.standard_normal((n, 21)).astype(np.float32)
    for k, e in envs.items():
        o, r, d, _ = e.step(np.clip(ctrl(obs[k]) + noise * z, -1, 1).astype(np.float32))
        obs[k] = o
        alive &= ~np.asarray(d).astype(bool)
    if (t + 1) % 100 == 0 or t == 998:
        row = ["step %4d alive %2d" % (t + 1, alive.sum())]
        for a, b in (("f32", "f64"), ("hip", "f64"), ("hip", "f32")):
            if a in obs and b in obs and alive.any():
                err = np.abs(obs[a] - obs[b]).max(axis=1)[alive]
                row.append("%s vs %s: median %.2e max %.2e" % (a, b, np.median(err), err.max()))
        print(" | ".join(row), flush=True)
