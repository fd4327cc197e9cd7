"""Off-line diagnosis of a miss of the frozen parity rule (tools/parity_heldout.py dumps every failed / beyond env-step as
<json stem>_miss_cell<c>_<k>.npz: injected state, action, and everything the HIP kernel returned).

Version 1 of the frozen rule (tests/parity_rule.py; version 2 lists NEAR_LIST = 16) listed at most 6 near-threshold decisions of an env-step and searched
subsets of at most MAX_DEPTH = 3 of them.  This tool lifts both limits for ONE env-step: it lists every decision within NEAR_TOL of its
threshold and tries every subset, and it re-runs the step with the base height nudged by +-1e-5 .. 4e-5 m (a knife-edge state decides
itself one way or the other; the kernel compiled for the host -- tests/host_lib.py -- must then agree with the oracle as it runs).
Nothing here changes the rule; it says whether a miss is the kernel's or the search's.

  python tools/heldout_failure_probe.py gpurun_out/r05_v1b_parity_heldout_miss_cell7_0.npz > profiles/r05_v1b_parity_heldout_miss_diagnosis.txt"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import host_lib as hl  # noqa: E402
import oracle_lib as ol  # noqa: E402
import parity_rule as pr  # noqa: E402

CAP = 16
D = np.load(sys.argv[1])
kind, seed, gid, cur = str(D["kind"]), int(D["seed"]), int(D["global_env_id"]), int(D["curriculum"])
st, act = D["state"][None].astype(np.float32), D["action"][None]
print("miss: %s, seed %d, global env id %d, curriculum %d, control step %d of the cell; failed = %s" % (kind, seed, gid, cur, int(D["step"]), bool(D["failed"])))
o = ol.OracleEnv(kind, 1, seed=seed, env_offset=gid)
if cur:
    o.set_curriculum(cur)
o.reset()
o.set_state(st)
b = o.step_ex(act, tol=pr.NEAR_TOL, record=True, cap=CAP)
near = [int(i) for i in b["near"][0][:b["nnear"][0]]]
gi = pr._ints(D["hip_state"][None], np.array([D["hip_done"]]), dict(bad_transition=D["hip_info"][None][:, 2], update_terrain=D["hip_info"][None][:, 4]))
bi = pr._ints(o.get_state(), b["done"], b["info"])
print("oracle as it ran: |obs - kernel| %.2e, integers equal %s" % (np.abs(b["obs"][0] - D["hip_obs"]).max(), bool((gi == bi).all())))
print("decisions within %.0e of their threshold: %d -> %s   (the rule lists at most %d of them and inverts at most %d at a time; decision index = "
      "90 x substep + 42 joint-limit switches + 6 x sole corner + 2 x stone slot + {0: touches, 1: wins})" % (pr.NEAR_TOL, len(near), near, pr.NEAR_LIST, pr.MAX_DEPTH))
best, tried = (np.inf, None), 0
for k in range(0, min(len(near), 10) + 1):
    for S in itertools.combinations(near, k):
        force = np.zeros((1, CAP), np.int32)
        force[0, :len(S)] = S
        o.set_state(st)
        r = o.step_ex(act, tol=pr.NEAR_TOL, force=force, nforce=np.array([len(S)], np.int32), cap=CAP)
        tried += 1
        ai = pr._ints(o.get_state(), r["done"], r["info"])
        e = float(np.abs(r["obs"][0] - D["hip_obs"]).max())
        if (ai == gi).all() and e < best[0]:
            best = (e, S)
print("every subset of them inverted (%d oracle runs): the kernel's result is the oracle's with %s inverted -- integers equal, |obs - kernel| %.2e%s" % (
    tried, best[1], best[0], "" if best[1] is None or all(i in near[:6] for i in best[1]) else
    "; decision(s) %s are beyond the list of 6 of rule version 1, which never tried them (version 2 lists %d)" % ([i for i in best[1] if i not in near[:6]], pr.NEAR_LIST)))
out, oh, rh, dh, ih = hl.step(ol.KIND[kind], st, act, seed=seed, curriculum=cur)
print("the kernel source compiled for the host on the same state: |obs - GPU| %.2e" % np.abs(oh[0] - D["hip_obs"]).max())
for dz in (-4e-5, -2e-5, -1e-5, 1e-5, 2e-5, 4e-5):
    s2 = st.copy()
    s2[0, 2] += dz
    o.set_state(s2)
    r = o.step_ex(act, tol=pr.NEAR_TOL, cap=CAP)
    so2 = o.get_state()
    out, oh, rh, dh, ih = hl.step(ol.KIND[kind], s2, act, seed=seed, curriculum=cur)
    print("base height %+.0e m: host kernel vs oracle as it runs |obs| %.2e, contact flags %d / %d, near-threshold decisions left %d" % (
        dz, np.abs(oh - r["obs"]).max(), int(out[0, ol.S_FLAGS]), int(so2[0, ol.S_FLAGS]), int(r["nnear"][0])))
