#!/usr/bin/env python3
"""INFORMATIONAL (this container only; never a parity claim, nothing here changes the specification): how the survival of the reference's
SHIPPED deterministic policies in our env responds to the free numbers of docs/PHYSICS.md -- solver knobs (run-time debug hooks of the
oracle) and global scale factors of the robot model (a private copy of the oracle built with regenerated tables under var/scan/; the
tree's tables and libraries are not touched).  A parameter whose change multiplies the survival time of BOTH policies is where our
robot differs most from the one they were trained on.

  python tools/policy_physics_scan.py > profiles/r04_policy_physics_scan.txt
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
SCAN = os.path.join(ROOT, "var", "scan")

WORKER = r'''
import sys, numpy as np, torch
sys.path[:0] = [%(root)r, %(root)r + "/tests"]
import oracle_lib as ol
from steppingstone_amd.legacy_checkpoint import load_reference_checkpoint
M = "/root/reference/playground/models/"
solver = %(solver)r
out = []
for kind, f in (("walker3d", "mocca_envs:Walker3DStepperEnv-v0_latest.pt"), ("mike", "mocca_envs:MikeStepperEnv-v0_latest.pt")):
    actor = load_reference_checkpoint(M + f).actor
    o = ol.OracleEnv(kind, 128, seed=9)
    if solver:
        import ctypes as C
        o.lib.sso_debug_set_solver.argtypes = [C.c_int, C.c_int]
        o.lib.sso_debug_set_variant.argtypes = [C.c_double, C.c_int]
        o.lib.sso_debug_set_solver(int(solver.get("iters", 8)), int(solver.get("warm", 0)))
        o.lib.sso_debug_set_variant(float(solver.get("erp", 0.2)), int(solver.get("seq", 0)))
    o.set_curriculum(0)
    obs = o.reset()
    lens, reached = [], []
    for t in range(300):
        with torch.no_grad():
            a = actor(torch.from_numpy(obs)).numpy()
        obs, _, d, info = o.step(a.astype(np.float32))
        for i in np.nonzero(d)[0]:
            lens.append(float(info["ep_len"][i])); reached.append(int(info["steps_reached"][i]))
    lens += o.get_state()[:, ol.S_ELAPSED].tolist()
    out.append("%%.1f / %%.2f" %% (np.mean(lens), np.mean(reached) if reached else 0.0))
print(" | ".join(out))
'''


def run(solver=None, lib=None):
    env = dict(os.environ)
    if lib:
        env["SS_ORACLE_LIB_F32"] = lib
    r = subprocess.run([sys.executable, "-c", WORKER % dict(root=ROOT, solver=solver or {})], capture_output=True, text=True, env=env, timeout=900)
    return r.stdout.strip() or ("failed: " + r.stderr.strip().splitlines()[-1][:120])


def build_variant(tag, mutate):
    """a private fp32 oracle with the model tables regenerated after `mutate(m)` (m: the dict steppingstone_amd.model.build returns)"""
    import gen_model_tables as gen
    from steppingstone_amd import model
    d = os.path.join(SCAN, tag)
    os.makedirs(d, exist_ok=True)
    orig = model.build

    def patched(kind):
        m = orig(kind)
        mutate(m)
        return m
    model.build = patched
    try:
        open(os.path.join(d, "ss_model_tables.h"), "w").write(gen.gen_h())
    finally:
        model.build = orig
    src = os.path.join(d, "ss_oracle.c")
    open(src, "w").write(open(os.path.join(ROOT, "oracle", "ss_oracle.c")).read())
    lib = os.path.join(d, "liboracle.so")
    subprocess.check_call(["cc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-DSSO_REAL=float",
                           "-o", lib, src, "-lm"], cwd=d)
    return lib


def scale(keys, f):
    def mutate(m):
        for k in keys:
            m[k] = np.asarray(m[k], float) * f
    return mutate


def main():
    print("# survival of the shipped deterministic policies (mean episode length in control steps / stones reached; 128 envs x 300 steps, flat")
    print("# terrain, fp32 CPU oracle), Walker3D | Mike.  Random actions 25-28 steps, zero actions 22-24.")
    print("%-46s %s" % ("specification as it is", run()))
    for label, s in (("PGS sweeps 5 (SURVEY 9: numSolverIterations)", dict(iters=5)), ("PGS sweeps 16", dict(iters=16)), ("PGS sweeps 32", dict(iters=32)),
                     ("warm start from the previous substep", dict(warm=1)), ("ERP 0.5", dict(erp=0.5)), ("ERP 0.9 (SURVEY 9)", dict(erp=0.9)),
                     ("Gauss-Seidel across the feet", dict(seq=1)), ("5 sweeps + warm start + ERP 0.9 (SURVEY 9 altogether)", dict(iters=5, warm=1, erp=0.9))):
        print("%-46s %s" % (label, run(solver=s)), flush=True)
    if "--solver-only" in sys.argv:
        return
    variants = []
    for f in (0.6, 0.8, 1.25, 1.6):
        variants.append(("mass and inertia x %g" % f, scale(["mass", "inertia_o"], f)))
    for f in (0.6, 0.8, 1.25, 1.6, 2.5):
        variants.append(("torque limits x %g" % f, scale(["torque"], f)))
    for f in (0.3, 3.0):
        variants.append(("joint damping x %g" % f, scale(["damping"], f)))
        variants.append(("joint stiffness x %g" % f, scale(["stiffness"], f)))
        variants.append(("armature x %g" % f, scale(["armature"], f)))
        variants.append(("limit spring / damper x %g" % f, scale(["k_lim", "d_lim"], f)))
    for mu in (0.5, 0.7, 1.2, 2.0):
        variants.append(("friction %g (spec: 0.9)" % mu, lambda m, mu=mu: m.__setitem__("friction", mu)))
    for i, (label, mut) in enumerate(variants):
        lib = build_variant("v%02d" % i, mut)
        print("%-46s %s" % (label, run(lib=lib)), flush=True)


if __name__ == "__main__":
    main()
