"""Dump the inputs and the HIP kernel's outputs of single-step parity runs (run on the GPU box), so that the
classification of every env-step (tests/test_gpu_parity.py) can be studied offline on the CPU with the oracle builds:
one control step from identical injected states, both robots, flat and curriculum-5 terrain.  Kept per env-step:
the injected state, the actions, the kernel's observation / reward / done / info and its state after the step --
for every env-step whose observation differs from the fp32 oracle's by more than `keep_tol`, plus every integer
mismatch, plus a random sample of the rest.  Writes gpurun_out/parity_dump_<kind>_c<curriculum>.npz.
usage: python tools/parity_dump.py [envs] [steps] [keep_tol]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
from steppingstone_amd.envs import SteppingStoneVecEnv  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
keep_tol = float(sys.argv[3]) if len(sys.argv) > 3 else 2e-5
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
rng = np.random.default_rng(0)
for env_id, kind in (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")):
    for cur, seed in ((0, 11), (5, 23)):
        g = SteppingStoneVecEnv(env_id, n, seed=seed, device="cuda:0", return_numpy=True)
        o = ol.OracleEnv(kind, n, seed=seed)
        if cur:
            g.update_curriculum(cur)
            o.set_curriculum(cur)
        g.reset()
        o.reset()
        K = {k: [] for k in ("st", "act", "g_obs", "g_rew", "g_done", "g_info", "g_state", "env", "t", "e_obs")}
        total = 0
        for t in range(steps):
            st = o.get_state()
            g.set_state(st)
            a = o.random_actions(t)
            oo, ro, do, io = o.step(a)
            og, rg, dg, _ = g.step(a)
            sg, so = g.get_state().cpu().numpy(), o.get_state()
            raw = g._info.cpu().numpy()
            ints = (sg[:, 59:65] == so[:, 59:65]).all(axis=1) & (dg == do.astype(bool)) & (raw[:, 2] == io["bad_transition"]) & \
                (raw[:, 4] == io["update_terrain"])
            e = np.abs(og - oo).max(axis=1)
            keep = (e > keep_tol) | ~ints | (rng.random(n) < 0.01)
            idx = np.nonzero(keep)[0]
            total += n
            K["st"].append(st[idx]); K["act"].append(a[idx]); K["g_obs"].append(og[idx]); K["g_rew"].append(rg[idx])
            K["g_done"].append(dg[idx]); K["g_info"].append(raw[idx]); K["g_state"].append(sg[idx, :65])
            K["env"].append(idx.astype(np.int32)); K["t"].append(np.full(idx.size, t, np.int32)); K["e_obs"].append(e[idx])
        out = {k: np.concatenate(v) for k, v in K.items()}
        path = os.path.join(ROOT, "gpurun_out", "parity_dump_%s_c%d.npz" % (kind, cur))
        np.savez_compressed(path, seed=seed, n=n, curriculum=cur, total=total, **out)
        print("%s curriculum %d: kept %d of %d env-steps (%d with |obs| error > 1e-4) -> %s" % (
            kind, cur, out["env"].size, total, int((out["e_obs"] > 1e-4).sum()), path))
        g.close()
