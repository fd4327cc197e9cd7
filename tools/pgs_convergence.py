"""Convergence study of the contact solver (oracle, fp64): error of one control step's velocities against a
converged solve (400 cold sweeps) for candidate (sweeps, warm-start) settings, on states sampled from random-action
rollouts and from standing poses."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_lib as ol
lib = ol.load("f64")
lib.sso_debug_set_solver.argtypes = [C.c_int, C.c_int]
n = 64
o = ol.OracleEnv("walker3d", n, seed=1, prec="f64")
o.reset()
states, acts = [], []
for t in range(60):
    a = o.random_actions(t) * (0.3 if t % 2 else 1.0)
    states.append(o.get_state().copy()); acts.append(a)
    o.step(a)
def run(iters, warm):
    lib.sso_debug_set_solver(iters, warm)
    out = []
    for st, a in zip(states, acts):
        o.set_state(st); o.set_auto_reset(False)
        o.step(a)
        out.append(o.get_state()[:, 7:55].copy())
    return np.array(out)
ref = run(400, 0)
incontact = np.array([st[:, 64] != 0 for st in states])       # flags before the step
scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-3
for iters, warm in ((8, 0), (5, 0), (4, 0), (8, 1), (5, 1), (4, 1), (3, 1), (2, 1), (16, 0)):
    got = run(iters, warm)
    err = (np.abs(got - ref) / scale).max(axis=2)[incontact]
    print("sweeps %2d warm %d : rel. velocity error median %.2e  p90 %.2e  max %.2e" % (iters, warm, np.median(err), np.quantile(err, 0.9), err.max()))
lib.sso_debug_set_solver(8, 0)
