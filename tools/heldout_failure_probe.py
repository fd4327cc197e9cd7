"""Diagnosis of a held-out parity failure (round 5: ONE env-step of 1 063 392, Mike, policy-driven, curriculum 0).
GPU half (run on the GPU box): replays the env-step from its saved state / action through ss_step and dumps what the kernel returned.
CPU half (--bruteforce): tries EVERY subset of the oracle's near-threshold decisions of that env-step (tests/parity_rule.py searches
subsets of at most 3) and reports whether one of them reproduces the kernel's result.
  python tools/heldout_failure_probe.py gpu   var/fail_state.npy var/fail_act.npy gpurun_out/r05_fail_probe.npz
  python tools/heldout_failure_probe.py brute var/fail_state.npy var/fail_act.npy gpurun_out/r05_fail_probe.npz"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
KIND, ENV_ID, SEED, GID, CUR = "mike", "MikeStepperEnv-v0", 9120, 800013 + 3776, 0

mode, fs, fa, fo = sys.argv[1:5]
st, act = np.load(fs), np.load(fa)
if mode == "gpu":
    from steppingstone_amd.envs import SteppingStoneVecEnv
    out = {}
    for rep in range(3):
        g = SteppingStoneVecEnv(ENV_ID, 1, seed=SEED, device="cuda:0", return_numpy=True, env_id_offset=GID)
        g.reset()
        g.set_state(st)
        o, r, d, _ = g.step(act)
        out["obs%d" % rep], out["rew%d" % rep], out["done%d" % rep] = o, r, d
        out["state%d" % rep], out["info%d" % rep] = g.get_state().cpu().numpy(), g._info.cpu().numpy()
        g.close()
    np.savez(fo, **out)
    print("kernel replayed 3 x; identical:", all(np.array_equal(out["obs0"], out["obs%d" % k]) for k in (1, 2)))
else:
    import oracle_lib as ol
    import parity_rule as pr
    G = np.load(fo)
    o = ol.OracleEnv(KIND, 1, seed=SEED, env_offset=GID)
    o.reset()
    o.set_state(st)
    b = o.step_ex(act, tol=pr.NEAR_TOL, record=True)
    near = [int(i) for i in b["near"][0][:b["nnear"][0]]]
    print("oracle as it ran: |obs - kernel| %.2e; near-threshold decisions: %s" % (np.abs(b["obs"] - G["obs0"]).max(), near))
    gi = pr._ints(G["state0"], G["done0"], dict(bad_transition=G["info0"][:, 2], update_terrain=G["info0"][:, 4]))
    best = (np.inf, None)
    for k in range(0, len(near) + 1):
        for S in itertools.combinations(near, k):
            force = np.zeros((1, max(len(near), 1)), np.int32)
            force[0, :len(S)] = S
            o.set_state(st)
            r = o.step_ex(act, tol=pr.NEAR_TOL, force=force, nforce=np.array([len(S)], np.int32))
            ai = pr._ints(o.get_state(), r["done"], r["info"])
            e = np.abs(r["obs"] - G["obs0"]).max()
            if (ai == gi).all() and e < best[0]:
                best = (e, S)
    print("best integer-exact subset of the near-threshold decisions: %s inverted -> |obs - kernel| %.2e (%d decisions; the rule searches <= %d)" % (
        best[1], best[0], len(best[1]) if best[1] is not None else -1, pr.MAX_DEPTH))
