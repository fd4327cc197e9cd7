"""HELD-OUT validation of the frozen GPU parity rule (tests/parity_rule.py, locked by tests/parity_rule.lock), run on the GPU box.

The rule's constants were calibrated on rounds 3-4's kernels with seeds 101 / 202, 4096 envs, curricula 0 / 5, 30 steps of
random-action burn-in (tools/parity_rule_stats.py).  This tool judges >= 1 M FRESH env-steps the calibration never saw:

  * seeds 9001 ... (no test, tool or profile uses them), env counts 3000 (ragged: 46.9 wavefronts) and 5056, env-id offsets != 0;
  * both robots x curricula 0 / 3 / 5 (3 was never judged before);
  * two state / action distributions per cell: (a) uniform random actions after 45 control steps of burn-in, from a part of the
    Philox action stream no other run touches (t >= 5000); (b) STATES HARVESTED FROM A PPO-TRAINED POLICY'S ROLLOUTS -- a policy
    trained on this very env on the GPU (python -m steppingstone_amd.train, saved with ppo.save_checkpoint), evaluated on the CPU
    against the ORACLE's observations with its exploration noise: 80 control steps of policy-driven burn-in (walking, stepping onto
    stones, stumbling), then judged steps under the policy's own stochastic actions;
  * every judged step goes through ss_step (the launch a policy in the loop uses); every 4th config goes through
    ss_rollout_random(1) instead where the actions are the stream's (the benchmarked instantiation).

Output: per cell and in total, the fraction of env-steps HELD TO THE FLAT 1e-4 as the headline and the count of EVERY escape hatch
(sensitive / other-branch / other-branch+sensitive / integer mismatch excused / loose / beyond / failures), the err / bound tail, and
the kernel's and the fp32 CPU oracle's distance to fp64.  Exit status 1 if any env-step fails or any threshold of
tests/parity_assert.py is missed on the total.

usage: python tools/parity_heldout.py [--steps 22] [--policy-dir gpurun_out/heldout_policies] [--train-updates 100] > profiles/<tag>_parity_heldout.txt"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import parity_assert as pa  # noqa: E402
import parity_rule as pr  # noqa: E402

ROBOTS = (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike"))


def train_policy(env_id, updates, out_dir):
    """PPO on the GPU env (torch learner, the package's own training entry point); returns the checkpoint path."""
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "%s_latest.pt" % env_id)
    if os.path.exists(path):
        return path
    log = os.path.join(out_dir, "%s_train.jsonl" % env_id)
    with open(log, "w") as f:
        subprocess.check_call([sys.executable, "-m", "steppingstone_amd.train", "--env", env_id, "--num-envs", "4096", "--num-steps", "32",
                               "--updates", str(updates), "--mini-batch-size", "4096", "--seed", "77", "--test-interval", "0",
                               "--save-dir", out_dir], cwd=ROOT, stdout=f)
    return path


def judged_cell(env_id, kind, n, seed, offset, cur, steps, policy, rollout_launch, t_base):
    import torch
    from steppingstone_amd.envs import SteppingStoneVecEnv
    g = SteppingStoneVecEnv(env_id, n, seed=seed, device="cuda:0", return_numpy=not rollout_launch, env_id_offset=offset)
    J = pr.StepJudge(kind, n, seed=seed, env_offset=offset, curriculum=cur)
    if cur:
        g.update_curriculum(cur)
    g.reset()
    torch.manual_seed(seed)           # the policy's exploration noise (CPU generator)

    def policy_actions(obs):
        with torch.no_grad():
            _, a, _ = policy.act(torch.from_numpy(np.ascontiguousarray(obs, np.float32)), deterministic=False)
        return a.clamp(-1, 1).numpy().astype(np.float32)

    burn = 80 if policy is not None else 45
    obs = J.o32.get_obs()
    walked = []
    for t in range(burn):
        a = policy_actions(obs) if policy is not None else J.o32.random_actions(t_base + t)
        obs, _, d, info = J.o32.step(a)
        if d.any():
            walked += info["steps_reached"][d].tolist()
    st = J.o32.get_state()
    res, dumps = [], []
    for t in range(burn, burn + steps):
        a = policy_actions(J.o32.get_obs()) if policy is not None else J.o32.random_actions(t_base + t)
        g.set_state(st)
        if rollout_launch:
            og, rg, dg = [x.cpu().numpy() for x in g.rollout_random(1, t0=t_base + t, steps_per_launch=1)]
        else:
            og, rg, dg, _ = g.step(a)
        sg = g.get_state().cpu().numpy()
        raw = g._info.cpu().numpy()
        r = J.judge(st, a, og, rg, np.asarray(dg).astype(bool), sg, raw[:, 2], raw[:, 4])
        res.append(r)
        for e in np.nonzero(~r["ok"] | r["beyond"])[0]:          # everything needed to replay a miss off-line (tools/heldout_failure_probe.py)
            dumps.append(dict(step=int(t), env=int(e), global_env_id=int(offset + e), state=np.array(st[e]), action=np.array(a[e]),
                              hip_obs=np.array(og[e]), hip_rew=float(rg[e]), hip_done=bool(np.asarray(dg)[e]), hip_state=np.array(sg[e]),
                              hip_info=np.array(raw[e]), failed=bool(not r["ok"][e])))
        st = r["next_state"]
    g.close()
    R, txt = pr.summarize(res)
    return R, txt, (float(np.mean(walked)) if walked else float("nan"), len(walked)), dumps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=22)
    ap.add_argument("--policy-dir", default=os.path.join(ROOT, "gpurun_out", "heldout_policies"))
    ap.add_argument("--train-updates", type=int, default=100)
    ap.add_argument("--json", default="")
    ap.add_argument("--seed-base", type=int, default=9001, help="another value = another independent sample (seeds, env ids, policy noise)")
    args = ap.parse_args()
    import torch
    from steppingstone_amd import ppo
    lock = open(os.path.join(ROOT, "tests", "parity_rule.lock")).read().split()[0]
    have = hashlib.sha256(open(os.path.join(ROOT, "tests", "parity_rule.py"), "rb").read()).hexdigest()
    print("# held-out validation of the frozen parity rule: tests/parity_rule.py sha256 %s (%s the lock)" % (have[:16], "==" if have == lock else "!= "))
    assert have == lock, "the rule was edited after it was frozen"
    policies = {}
    for env_id, kind in ROBOTS:
        t0 = time.time()
        path = train_policy(env_id, args.train_updates, args.policy_dir)
        ac, ck = ppo.load_checkpoint(path)
        policies[kind] = ac.eval()
        rows = [json.loads(l) for l in open(os.path.join(args.policy_dir, "%s_train.jsonl" % env_id)) if l.startswith("{")]
        ret = [r.get("mean_rew") for r in rows if isinstance(r, dict) and "mean_rew" in r]
        print("# policy for %s: %d PPO updates on the GPU env in %.0f s, mean episode return first / last logged update: %s / %s" % (
            env_id, args.train_updates, time.time() - t0, ret[0] if ret else "?", ret[-1] if ret else "?"), flush=True)
    allR, cell_id, rows = [], 0, []
    for ri, (env_id, kind) in enumerate(ROBOTS):
        for cur in (0, 3, 5):
            for src in ("random", "policy"):
                n = 3000 if (cell_id % 2 == 0) else 5056
                seed, offset = args.seed_base + 17 * cell_id, 100000 * (cell_id + 1) + 13 + (args.seed_base - 9001) * 1000
                rollout_launch = (src == "random") and (cell_id % 4 == 0)
                t0 = time.time()
                R, txt, walked, dumps = judged_cell(env_id, kind, n, seed, offset, cur, args.steps, policies[kind] if src == "policy" else None,
                                                    rollout_launch, t_base=5000 + 1000 * cell_id + (args.seed_base - 9001) * 7)
                for k, dmp in enumerate(dumps):
                    np.savez(os.path.splitext(args.json or os.path.join(ROOT, "gpurun_out", "heldout"))[0] + "_miss_cell%d_%d.npz" % (cell_id, k),
                             kind=kind, env_id=env_id, seed=seed, curriculum=cur, **dmp)
                c = pa.counts(R)
                c.update(robot=kind, curriculum=cur, source=src, envs=n, seed=seed, env_id_offset=offset,
                         launch="ss_rollout_random(1)" if rollout_launch else "ss_step", seconds=round(time.time() - t0, 1))
                rows.append(c)
                print("%-8s curriculum %d %-6s n=%d seed=%d %s: %s" % (kind, cur, src, n, seed, c["launch"], txt))
                print("   HELD TO THE FLAT 1e-4: %.1f %% | escape hatches: sensitive %d, other branch %d, other branch + sensitive %d, integer mismatch "
                      "excused %d, loose %d, beyond %d, FAILURES %d | err / bound 99.9 %% %.3f max %.3f | within 1e-4 of the oracle as it ran: %.2f %% | "
                      "farther than 1e-4 from fp64: kernel %d, fp32 CPU oracle %d | bound median %.2e 90 %% %.2e | burn-in episodes ended: %d (mean stones "
                      "reached %.2f) | %.0f s" % (
                          100 * c["held_to_flat_1e4"], c["sensitive"], c["other_branch"], c["other_branch_sensitive"], c["int_excused"], c["loose"],
                          c["beyond"], c["failures"], c["q999_err_over_bound"], c["max_err_over_bound"], 100 * c["within_1e4_of_oracle"],
                          c["far_from_fp64_hip"], c["far_from_fp64_cpu_fp32"], c["bound_median"], c["bound_q90"], walked[1], walked[0], c["seconds"]), flush=True)
                for i in np.nonzero(~R["ok"] | R["beyond"])[0][:8]:
                    print("   %s env-step %d: category %d near %s | obs err %.2e / bound %.2e | reward %.2e / %.2e | pose %.2e / %.2e | velocities "
                          "%.2e / %.2e | integers equal %s" % ("FAILED" if not R["ok"][i] else "beyond its bound (counted)", i, R["category"][i],
                                                              bool(R["near"][i]), R["matched_e"][i], R["tol"][i], R["e_rew"][i], R["tol_rew"][i],
                                                              R["e_pose"][i], R["tol_pose"][i], R["e_vel"][i], R["tol_vel"][i], bool(R["int_ok"][i])), flush=True)
                allR.append(R)
                cell_id += 1
    T = {k: np.concatenate([r[k] for r in allR]) for k in allR[0]}
    c = pa.counts(T)
    print("\nTOTAL %d held-out env-steps, %d cells: HELD TO THE FLAT 1e-4: %.1f %% | sensitive %d (%.1f %%), other branch %d, other branch + sensitive %d, "
          "integer mismatch excused %d, loose %d (%.3f %%), beyond %d, FAILURES %d | err / bound 99.9 %% %.3f max %.3f | within 1e-4 of the oracle as it "
          "ran: %.2f %% | farther than 1e-4 from fp64: kernel %d, fp32 CPU oracle %d | bound median %.2e, 90 %% %.2e" % (
              c["env_steps"], len(rows), 100 * c["held_to_flat_1e4"], c["sensitive"], 100.0 * c["sensitive"] / c["env_steps"], c["other_branch"],
              c["other_branch_sensitive"], c["int_excused"], c["loose"], 100.0 * c["loose"] / c["env_steps"], c["beyond"], c["failures"],
              c["q999_err_over_bound"], c["max_err_over_bound"], 100 * c["within_1e4_of_oracle"], c["far_from_fp64_hip"], c["far_from_fp64_cpu_fp32"],
              c["bound_median"], c["bound_q90"]))
    if args.json:
        json.dump(dict(total=c, cells=rows, rule_sha256=have), open(args.json, "w"), indent=1)
    try:
        pa.assert_judged(T, "(total)", "held-out total", log=lambda *_: None)
        print("thresholds of tests/parity_assert.py on the total: all met")
    except AssertionError as exc:
        print("thresholds of tests/parity_assert.py on the total: MISSED: %r" % (exc,))
        raise SystemExit(1)


if __name__ == "__main__":
    main()
