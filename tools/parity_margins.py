"""Calibration of the parity bounds (run on the GPU box): one control step from identical injected states by the HIP
kernel and by the fp32 oracle, with the oracle's decision margins (oracle_lib.OracleEnv.step_margins) per env-step.
Prints how the observation error relates to the distance of the nearest discrete decision (contact predicate, stone
choice, joint-limit switch) from its threshold; writes gpurun_out/parity_margins_<kind>.npz.
usage: python tools/parity_margins.py [envs] [steps]"""
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
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for env_id, kind in (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")):
    g = SteppingStoneVecEnv(env_id, n, seed=11, device="cuda:0", return_numpy=True)
    o = ol.OracleEnv(kind, n, seed=11)
    o64 = ol.OracleEnv(kind, n, seed=11, prec="f64")
    g.update_curriculum(5); o.set_curriculum(5); o64.set_curriculum(5)
    g.reset(); o.reset(); o64.reset()
    E, S, R, M0, M1, I, E64, G64 = [], [], [], [], [], [], [], []
    for t in range(steps):
        st = o.get_state()
        g.set_state(st)
        a = o.random_actions(t)
        o64.set_state(st.astype(np.float64))
        o6 = o64.step(a)[0]
        oo, ro, do, io, mg = o.step_margins(a)
        og, rg, dg, _ = g.step(a)
        E64.append(np.abs(oo - o6).max(axis=1)); G64.append(np.abs(og - o6).max(axis=1))
        sg, so = g.get_state().cpu().numpy(), o.get_state()
        raw = g._info.cpu().numpy()
        ints = (sg[:, 59:65] == so[:, 59:65]).all(axis=1) & (dg == do.astype(bool)) & (raw[:, 2] == io["bad_transition"]) & (raw[:, 4] == io["update_terrain"])
        keep = ~do.astype(bool)                      # a reset replaces the state: compare only continuing envs' states
        es = np.abs(sg[:, :59] - so[:, :59]) / (1.0 + np.abs(so[:, :59]))
        E.append(np.abs(og - oo).max(axis=1)); S.append(np.where(keep, es.max(axis=1), 0.0)); R.append(np.abs(rg - ro))
        M0.append(mg[:, 0]); M1.append(mg[:, 1]); I.append(ints)
    E, S, R, M0, M1, I, E64, G64 = map(np.concatenate, (E, S, R, M0, M1, I, E64, G64))
    np.savez(os.path.join(ROOT, "gpurun_out", "parity_margins_%s.npz" % kind), e_obs=E, e_state=S, e_rew=R, m0=M0, m1=M1, ints=I,
             e_o32_o64=E64, e_gpu_o64=G64)
    out = E > 1e-4
    print("   fp64 arbiter: of the %d env-steps with |gpu - o32| > 1e-4, %d have |o32 - o64| > 2.5e-5 and %d have margin0 < 1e-5; "
          "unexplained by both: %d" % (out.sum(), (out & (E64 > 2.5e-5)).sum(), (out & (M0 < 1e-5)).sum(),
                                      (out & ~(E64 > 2.5e-5) & ~(M0 < 1e-5)).sum()))
    un = out & ~(E64 > 2.5e-5) & ~(M0 < 1e-5)
    print("   unexplained rows (gpu-o32, o32-o64, gpu-o64, margin0):", [(float(E[i]), float(E64[i]), float(G64[i]), float(M0[i])) for i in np.nonzero(un)[0][:12]])
    print("   |o32 - o64| quantiles 50%% %.2e 99%% %.2e 99.9%% %.2e max %.2e ; |gpu - o64| 50%% %.2e 99%% %.2e 99.9%% %.2e max %.2e" % (
        tuple(np.quantile(E64, [.5, .99, .999, 1.0])) + tuple(np.quantile(G64, [.5, .99, .999, 1.0]))))
    print("   env-steps with |o32 - o64| > 1e-4: %d ; with |gpu - o64| > 1e-4: %d" % ((E64 > 1e-4).sum(), (G64 > 1e-4).sum()))
    print("== %s: %d env-steps" % (kind, E.size))
    print("   |obs| error quantiles  50%% %.2e  90%% %.2e  99%% %.2e  99.9%% %.2e  max %.2e" % tuple(np.quantile(E, [.5, .9, .99, .999, 1.0])))
    print("   state rel-err quantiles 50%% %.2e  90%% %.2e  99%% %.2e  99.9%% %.2e  max %.2e" % tuple(np.quantile(S, [.5, .9, .99, .999, 1.0])))
    print("   |rew| error quantiles  50%% %.2e  90%% %.2e  99%% %.2e  99.9%% %.2e  max %.2e" % tuple(np.quantile(R, [.5, .9, .99, .999, 1.0])))
    for tol in (1e-4, 3e-4, 1e-3):
        out = E > tol
        print("   obs error > %.0e: %d env-steps (%.3f %%); their state-decision margins: %s" % (tol, out.sum(), 100.0 * out.mean(), np.sort(M0[out])[-8:] if out.any() else []))
        for eps in (1e-6, 1e-5, 1e-4):
            print("        margin0 < %.0e explains %d of %d; env-steps with margin0 < eps overall: %d" % (eps, (out & (M0 < eps)).sum(), out.sum(), (M0 < eps).sum()))
    bad_r = (R > 1e-3) & (E <= 1e-4)
    print("   reward-only mismatches (> 1e-3 with obs ok): %d; their reward-decision margins: %s" % (bad_r.sum(), np.sort(M1[bad_r])[-8:] if bad_r.any() else []))
    bad_i = ~I
    print("   integer / done mismatches: %d; min(margin0, margin1) of those: %s" % (bad_i.sum(), np.sort(np.minimum(M0, M1)[bad_i])[-8:] if bad_i.any() else []))
    g.close()
