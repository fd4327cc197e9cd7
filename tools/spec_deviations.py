"""Measured effect of each solver choice of docs/PHYSICS.md 3.4 that deviates from SURVEY.md section 9's (unverified)
recollection of Bullet's defaults -- on the fp64 oracle, this container, no GPU.  One control step from states of
random-action rollouts (curriculum 5) and of robots standing under a PD controller, against the specified solve
(8 cold sweeps, ERP 0.2, Jacobi between the feet): relative change of the post-step velocities of the env-steps that are
in contact, plus the steady-state sole penetration of a standing robot and the joint-limit overshoot.  The table
goes into docs/HISTORY.md section 3.  usage: python tools/spec_deviations.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from steppingstone_amd import model as M  # noqa: E402

lib = ol.load("f64")
lib.sso_debug_set_solver.argtypes = [C.c_int, C.c_int]
lib.sso_debug_set_variant.argtypes = [C.c_double, C.c_int]


def setting(iters=8, warm=0, erp=0.2, seq=0):
    lib.sso_debug_set_solver(iters, warm)
    lib.sso_debug_set_variant(erp, seq)


for kind in ("walker3d", "mike"):
    n = 128
    o = ol.OracleEnv(kind, n, seed=1, prec="f64")
    o.set_curriculum(5)
    o.reset()
    setting()
    states, acts = [], []
    for t in range(50):
        a = o.random_actions(t) * (0.3 if t % 2 else 1.0)
        states.append(o.get_state().copy())
        acts.append(a)
        o.step(a)

    def run(**kw):
        setting(**kw)
        out = []
        o.set_auto_reset(False)
        for st, a in zip(states, acts):
            o.set_state(st)
            o.step(a)
            out.append(o.get_state()[:, 7:55].copy())
        o.set_auto_reset(True)
        setting()
        return np.array(out)

    ref = run()
    contact = np.array([st[:, 64] != 0 for st in states])
    double = np.array([st[:, 64] == 3 for st in states])
    scale = np.abs(ref).max(axis=2, keepdims=True) + 1e-3
    print("== %s: %d env-steps in contact (%d in double support) of %d" % (kind, contact.sum(), double.sum(), contact.size))
    rows = [("5 sweeps (SURVEY 9: numSolverIterations = 5)", dict(iters=5), contact),
            ("16 sweeps", dict(iters=16), contact),
            ("400 sweeps (converged where it converges)", dict(iters=400), contact),
            ("warm start from the previous substep", dict(warm=1), contact),
            ("ERP 0.9 (SURVEY 9: default contact ERP), same 2 m/s cap", dict(erp=0.9), contact),
            ("Gauss-Seidel between the feet instead of Jacobi", dict(seq=1), double)]
    for name, kw, mask in rows:
        err = (np.abs(run(**kw) - ref) / scale).max(axis=2)[mask]
        print("   %-58s rel. velocity change after one control step: median %.1e  p90 %.1e  max %.1e" % (
            name, np.median(err), np.quantile(err, 0.9), err.max()))
    # standing robot: steady-state penetration for ERP 0.2 / 0.9, joint-limit overshoot under random actions
    m = M.build(kind)
    for erp in (0.2, 0.9):
        setting(erp=erp)
        s = ol.OracleEnv(kind, 1, seed=3, prec="f64")
        s.reset()
        pen = []
        for k in range(6):
            s.substeps(0, np.zeros(21), 4)
        st = s.get_state()[0]
        pos, rot = ol.debug_fk(kind, st)
        zc = [(pos[b] + rot[b] @ c)[2] for b in (M.RIGHT_FOOT_BODY, M.LEFT_FOOT_BODY) for c in m["corners"]]
        print("   ERP %.1f: deepest sole corner of a torque-free robot 0.1 s after reset: %.2f mm inside the stone" % (erp, -1e3 * min(zc)))
    setting()
    lo, hi = m["range"][:, 0], m["range"][:, 1]
    over = []
    o.reset()
    for t in range(200):
        o.step(o.random_actions(t))
        q = o.get_state()[:, 13:34]
        over.append(np.maximum(np.maximum(q - hi, lo - q), 0).max(axis=1))
    over = np.concatenate(over)
    print("   soft joint limits (implicit spring-damper, PHYSICS.md 3.1) under U(-1,1) torques: %.1f %% of env-steps have a joint past its "
          "limit; overshoot median %.1f deg, 99 %% %.1f deg, max %.1f deg" % (100 * (over > 0).mean(), np.degrees(np.median(over[over > 0])),
                                                                          np.degrees(np.quantile(over[over > 0], 0.99)), np.degrees(over.max())))
