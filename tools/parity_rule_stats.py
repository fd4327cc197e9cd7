"""Large-sample statistics of the GPU parity rule (tests/parity_rule.py), run on the GPU box: 4096 envs x `steps` judged control steps
per robot and terrain (one step from identical injected states, after a burn-in that spreads the states over falls, partial
contacts and resets), HIP kernel through ss_step.  Prints the category counts, the largest error / bound, quantiles of bound and
error, and the number of env-steps farther than 1e-4 from the fp64 oracle for the kernel and for the fp32 CPU oracle.
usage: python tools/parity_rule_stats.py [steps] > profiles/<tag>_parity_rule_stats.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import parity_rule as pr  # noqa: E402
from steppingstone_amd.envs import SteppingStoneVecEnv  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 4096
for env_id, kind in (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")):
    for cur, seed in ((0, 101), (5, 202)):
        g = SteppingStoneVecEnv(env_id, n, seed=seed, device="cuda:0", return_numpy=True)
        J = pr.StepJudge(kind, n, seed=seed, curriculum=cur)
        if cur:
            g.update_curriculum(cur)
        g.reset()
        for t in range(30):
            J.o32.step(J.o32.random_actions(t))
        st = J.o32.get_state()
        res = []
        for t in range(30, 30 + steps):
            g.set_state(st)
            a = J.o32.random_actions(t)
            og, rg, dg, _ = g.step(a)
            raw = g._info.cpu().numpy()
            r = J.judge(st, a, og, rg, dg.astype(bool), g.get_state().cpu().numpy(), raw[:, 2], raw[:, 4])
            res.append(r)
            st = r["next_state"]
        g.close()
        R, txt = pr.summarize(res)
        ratio = R["matched_e"] / R["tol"]
        print("%s curriculum %d: %s" % (kind, cur, txt))
        print("   error / bound: 99 %% %.3f  99.9 %% %.3f  max %.3f | bound quantiles 50 / 90 / 99 %%: %.1e %.1e %.1e | error quantiles 50 / 99 / 99.9 %%: "
              "%.1e %.1e %.1e | farther than 1e-4 from fp64: kernel %d, fp32 CPU oracle %d" % (
                  np.quantile(ratio, .99), np.quantile(ratio, .999), ratio.max(), *np.quantile(R["tol"], [.5, .9, .99]),
                  *np.quantile(R["e_obs"], [.5, .99, .999]), int((R["e_hip_o64"] > 1e-4).sum()), int((R["e_o32_o64"] > 1e-4).sum())), flush=True)
        q = [.5, .9, .99, 1.0]
        for name, e, tol in (("reward", "e_rew", "tol_rew"), ("pose (pos, quat, q)", "e_pose", "tol_pose"), ("velocities (twist, qd)", "e_vel", "tol_vel")):
            print("   %-24s error 50 / 90 / 99 / 100 %%: %.1e %.1e %.1e %.1e | bound: %.1e %.1e %.1e %.1e | error / bound max %.3f | at the floor: %.0f %%" % (
                (name,) + tuple(np.quantile(R[e], q)) + tuple(np.quantile(R[tol], q)) + ((R[e] / R[tol])[R["category"] < 2].max(), 100.0 * (R[tol] <= np.quantile(R[tol], 0) * 1.0000001).mean())), flush=True)
        print("   observation bound: max %.1e; env-steps with a bound beyond its ceiling (%.0e obs / pose, %.0e velocities / reward): %d = %.4f %%; "
              "env-steps held to 1e-4: %.1f %%" % (R["tol"].max(), pr.OBS_CEIL, pr.VEL_CEIL, int(R["loose"].sum()), 100.0 * R["loose"].mean(),
                                                  100.0 * (R["category"] == 0).mean()), flush=True)
        for i in np.nonzero(~R["ok"] | R["beyond"])[0][:8]:
            print("   %s env-step %d: category %d near %s | obs err %.2e / bound %.2e | reward %.2e / %.2e | pose %.2e / %.2e | velocities %.2e / %.2e | "
                  "integers equal %s" % ("FAILED" if not R["ok"][i] else "beyond its bound (counted)", i, R["category"][i], bool(R["near"][i]), R["matched_e"][i],
                                        R["tol"][i], R["e_rew"][i], R["tol_rew"][i], R["e_pose"][i], R["tol_pose"][i], R["e_vel"][i], R["tol_vel"][i],
                                        bool(R["int_ok"][i])), flush=True)
