"""The parity rule of the GPU tests: how ONE control step of the HIP path, started from an injected state, is judged
against the CPU oracle.  TEST INFRASTRUCTURE (imports oracle_lib).

FROZEN since round 5 (tests/parity_rule.lock holds the SHA-256 of this file; tests/test_parity_rule_frozen.py fails on any edit): the
constants below were calibrated on rounds 3-4's kernels and amended four times in round 4 after misses, so from here on a miss is
fixed in the kernel or reported as a miss -- not absorbed by the rule.  Held-out validation: tools/parity_heldout.py.

VERSION 2 (re-locked once, at the end of round 5, in a commit of its own).  Version 1 (sha256 e1a62ad8...) was validated on three
held-out samples of 1 063 392 GPU env-steps each: 1, 0 and 3 failures (profiles/r05_v1b / v1c / v1d_parity_heldout.*).  All four were
diagnosed off line (tools/heldout_failure_probe.py, profiles/*_miss_diagnosis.txt) as ONE limitation of the SEARCH, not of a bound: two
sole corners of a standing Mike sat on their touch threshold through all four substeps = 8 near-threshold decisions; version 1 listed
the first 6 (oracle_lib.NEAR_CAP) and so never inverted the one of the fourth substep that the kernel had taken -- with it inverted
the oracle equals the kernel to 1.2e-7, integers included.  Version 2 changes the search capacity and nothing else: NEAR_LIST = 16
decisions listed (4 corners x 4 substeps) and MAX_ALTERNATIVES 24 -> 40 so that the 16 single inversions do not eat the budget of the
pairs and triples.  No tolerance, factor, ceiling, fraction or depth moved; an env-step with at most 6 near-threshold decisions on
every branch visited is judged by the oracle runs version 1 made, in the same order.  Version 1's three samples stay on record as
they came out; version 2 is validated on a FRESH sample (other seeds), not by re-judging those.

VERSION 3 (re-locked once, at the START of round 6, in a commit of its own, BEFORE any new sample was drawn or looked at; VERDICT r5
item 1b).  A NARROWING ONLY, to what the six held-out runs of round 5 support (6 380 352 GPU env-steps, profiles/r05_{a,v1b,v1c,v1d,v2,v2b}
_parity_heldout.txt): two escape hatches that never fired in any of them are now hard failures --
  * the 1 x - 2 x tail (`beyond`): TAIL_FACTOR 2 -> 1, BEYOND_MAX_FRACTION 2e-4 -> 0.  The acceptance region IS max(floor, 8 s) now, for
    every quantity; the largest err / bound on a clean sample was 0.90, so SENS_FACTOR = 8 stays where it is;
  * "integer mismatch excused by an unstable probe" (`int_ok | ~stable`): an integer outcome that differs from the oracle's, on an
    env-step WITHOUT a near-threshold decision, fails whatever the sensitivity probe saw.  `int_excused` is still computed and reported
    (such an env-step is now a failure that carries this label); near-threshold env-steps are judged by the branch search as before.
Nothing was widened: floors, factor, ceilings, search depth and capacity are version 2's.  The callers' thresholds narrowed with it
(tests/parity_assert.py: 99.9 % quantile of err / bound < 0.2 instead of < 0.5, beyond == 0, int_excused == 0, a hard floor on the
plain fraction).  Lock history: tests/parity_rule.lock lists the SHA-256 of versions 1, 2 and 3.  If version 3 misses on the fresh
sample it is reported as a miss; no tolerance moves back.

Every env-step is bounded by max(floor, 8 s).  A bound above its ceiling is counted as `loose` (asserted < 1 %) rather than capped.
That hatch, the alternative-branch matches and the two retired hatches' counters (`beyond`, `int_excused`: both must read 0) are
reported by summarize() and by every caller:

  integers (next_step_index, counters, RNG counter, contact flags, done, bad_transition, update_terrain): bit-exact;
  observation: |obs_hip - obs_oracle| <= max(1e-4, 8 s)      (1e-4 = the north-star's per-step bound)
  reward:      |rew_hip - rew_oracle| <= max(1e-4, 8 s_rew)  (round 4; rounds 1-3 allowed 1e-3 flat.  The reward's progress term is
               60 x the change of a planar distance, so ONE fp32 ulp of a base position of a few metres is already 1.4e-5 of
               reward: s_rew is about 1e-4 on most env-steps and the effective bound stays near 1e-3, but it is now measured per
               step instead of granted)
  post-step state (round 4: what the NEXT step starts from, not only what the policy sees): base position, quaternion and joint
               angles <= max(1e-4, 8 s_pose); base twist and joint rates <= max(1e-3, 8 s_vel) (rates enter the observation as 0.1 q')
  tail:        err / s has no hard limit in principle (s is the response to INPUT errors; the kernel also rounds every intermediate) and
               its tail is heavy (max err / bound 0.75 and 0.90 on two clean held-out samples); versions 1-2 therefore COUNTED env-steps
               between 1 x and 2 x their bound (`beyond`) instead of failing them.  None occurred in 6.38 M held-out env-steps, so from
               version 3 on outside the factor-8 bound is a failure; the callers assert that the 99.9 % quantile of err / bound stays
               below 0.2 (measured 0.056 - 0.13).  (Round 4, another kernel: one env-step of a 15 360-step test reached 1.03.)
  loose bounds: an env-step whose bound exceeds 5e-3 (observation, pose) or 5e-2 (velocities, reward) is counted (`loose`); the tests
               assert that such steps stay below 1 % of the env-steps (measured: 0.4 % of a fall-heavy CPU sample of 10 240, see
               profiles/r04_v5_parity_rule_stats.txt for the GPU sample: a foot pivoting on one corner, a body spinning up before the
               episode ends).  A hard cap instead fails the CPU's own fp32 build on those steps (round 4 tried 5e-3 / 5e-2: 26
               "failures" of 327 680 env-steps, every one of them a capped bound with the error inside 8 s)

where s is the MEASURED first-order sensitivity of that very env-step in the fp64 build of the oracle: each of the 55
dynamic state inputs (base pose / twist, q, qd) is perturbed by 8 ulp (relative 8 * 2^-23, floor 1e-3 absolute scale), one
at a time, and the absolute changes of the observation are summed per component (a first-order worst case over the
signs of the input errors); s also includes the distance between the oracle's own fp32 and fp64 builds on that step.  An
env-step with 8 s <= 1e-4 is "plain" and held to the north-star's 1e-4; one with 8 s > 1e-4 ("sensitive": the light
foot / ankle pivoting on one or two sole corners amplifies rounding 1e2..1e5 x within the four substeps, measured on the
CPU alone) is held to 8 x what the specification itself does to an 8-ulp input error.  The classification never looks at
the HIP result.  Calibration (tools/parity_rule_stats.py: 327 680 env-steps, both robots, flat and curriculum-5 terrain,
round-3 kernels): the largest |obs| error over s is 5.4 (99.9 % of env-steps: below 0.9), hence the factor 8; about 60 % of
env-steps are plain, the bound of the others is 2.8e-4 at the 90 % and 1e-3 at the 99 % quantile of all env-steps.  The fp32 oracle itself is farther than 1e-4 from its fp64 build on 0.16 % of env-steps (the kernel:
0.17 %), so no fp32 implementation can hold 1e-4 on every step -- hence a bound that scales with the step's own
conditioning rather than a blanket allowance.

  decisions: when a discrete decision of the oracle's step (contact predicate of a sole corner, winner among two stones,
  joint-limit switch, reward / done thresholds) lies within 1e-5 of its threshold, an fp32 implementation with another
  operation order may take the other branch.  Then the HIP result must agree -- integers exactly, observation / reward
  to the bounds above -- with the oracle re-evaluated with SOME subset of the near-threshold decisions inverted
  (oracle_lib.step_ex(force=...); decisions that become near-threshold on the alternative trajectory are searched too,
  depth <= 3); the sensitivity along an alternative branch is measured with every decision frozen (replay) so that the
  probe cannot flip the branch it is measuring.

`StepJudge.judge()` returns per-env arrays; `category`: 0 plain, 1 sensitive, 2 other branch, 3 other branch + sensitive.
"""
import numpy as np

import oracle_lib as ol

OBS_TOL, REW_TOL, NEAR_TOL = 1e-4, 1e-4, 1e-5
POSE_TOL, VEL_TOL = 1e-4, 1e-3              # post-step state: pos 3 + quat 4 + q 21 | base twist 6 + qd 21
OBS_CEIL, POSE_CEIL, VEL_CEIL, REW_CEIL = 5e-3, 5e-3, 5e-2, 5e-2      # a bound above these marks the env-step `loose` (counted, asserted rare)
LOOSE_MAX_FRACTION = 1e-2
POSE_COLS = list(range(0, 7)) + list(range(13, 34))
VEL_COLS = list(range(7, 13)) + list(range(34, 55))
ULPS, SENS_FACTOR = 8.0, 8.0
TAIL_FACTOR, BEYOND_MAX_FRACTION = 1.0, 0.0    # version 3: no tail -- outside max(floor, 8 s) is a failure (versions 1-2: 2.0, 2e-4; never used)
MAX_DEPTH, MAX_ALTERNATIVES = 3, 40
NEAR_LIST = 16                              # near-threshold decisions listed per env-step (version 1: oracle_lib.NEAR_CAP = 6)
NDYN = 55                                   # pos 3, quat 4, twist 6, q 21, qd 21 of the packed state
INT_FIELDS = [ol.S_N, ol.S_COUNT, ol.S_ELAPSED, ol.S_CTRLO, ol.S_CTRHI, ol.S_FLAGS]


def _ints(state, done, info):
    """[N, 9] integer outcome of a step: the six integer state fields, done, bad_transition, update_terrain."""
    return np.concatenate([state[:, INT_FIELDS].astype(np.int64), np.asarray(done).astype(np.int64)[:, None],
                           np.asarray(info["bad_transition"]).astype(np.int64)[:, None],
                           np.asarray(info["update_terrain"]).astype(np.int64)[:, None]], axis=1)


class StepJudge:
    def __init__(self, kind, n, seed=0, env_offset=0, curriculum=0, setup=None):
        self.n = n
        self.o32 = ol.OracleEnv(kind, n, seed=seed, env_offset=env_offset)
        self.alt = ol.OracleEnv(kind, n, seed=seed, env_offset=env_offset)
        self.o64 = ol.OracleEnv(kind, n, seed=seed, env_offset=env_offset, prec="f64")
        for o in (self.o32, self.alt, self.o64):
            if curriculum:
                o.set_curriculum(curriculum)
            if setup:
                setup(o)
            o.reset()

    # ------------------------------------------------------------------------------------------------ sensitivity
    def _sensitivity(self, st64, act, ref, replay=None):
        """Sum over the 55 dynamic inputs of |change of obs| (per component, then max) and of |change of rew| under an
        8-ulp perturbation of that input, in the fp64 oracle; `stable`: no perturbation changed an integer outcome."""
        acc_o = np.zeros((self.n, ol.OBS_DIM))
        acc_r = np.zeros(self.n)
        acc_s = np.zeros((self.n, NDYN))
        stable = np.ones(self.n, bool)
        for i in range(NDYN):
            p = st64.copy()
            p[:, i] += ULPS * 2.0 ** -23 * np.maximum(np.abs(p[:, i]), 1e-3)
            self.o64.set_state(p)
            r = self.o64.step_ex(act, replay=replay)
            acc_o += np.abs(r["obs"].astype(np.float64) - ref["obs"])
            acc_r += np.abs(r["rew"].astype(np.float64) - ref["rew"])
            s_after = self.o64.get_state()
            acc_s += np.abs(s_after[:, :NDYN] - ref["state"])
            if replay is None:
                stable &= (_ints(s_after, r["done"], r["info"]) == ref["ints"]).all(axis=1)
        self.s_pose, self.s_vel = acc_s[:, POSE_COLS].max(axis=1), acc_s[:, VEL_COLS].max(axis=1)
        return acc_o.max(axis=1), acc_r, stable

    # ------------------------------------------------------------------------------------------------ the rule
    def judge(self, st, act, g_obs, g_rew, g_done, g_state, g_bad, g_upd):
        """st [N,185] f32 injected state, act [N,21]; g_*: what the HIP path returned for that step (g_state: its packed
        state after the step).  Leaves self.o32 advanced by the step (self.o32.get_state() is the next state)."""
        n = self.n
        st = np.ascontiguousarray(st, np.float32)
        st64 = st.astype(np.float64)
        g_obs = np.asarray(g_obs, np.float64)
        g_rew = np.asarray(g_rew, np.float64)
        g_int = _ints(np.asarray(g_state), g_done, dict(bad_transition=g_bad, update_terrain=g_upd))
        self.o32.set_state(st)
        b = self.o32.step_ex(act, tol=NEAR_TOL, record=True, cap=NEAR_LIST)
        so = self.o32.get_state()
        b_int = _ints(so, b["done"], b["info"])
        self.o64.set_state(st64)
        r64 = self.o64.step_ex(act)
        s64 = self.o64.get_state()
        ref = dict(obs=r64["obs"].astype(np.float64), rew=r64["rew"].astype(np.float64), state=s64[:, :NDYN].copy(),
                   ints=_ints(s64, r64["done"], r64["info"]))
        s_obs, s_rew, stable = self._sensitivity(st64, act, ref)
        s_obs = np.maximum(s_obs, np.abs(b["obs"] - ref["obs"]).max(axis=1))
        s_rew = np.maximum(s_rew, np.abs(b["rew"] - ref["rew"]))
        d32 = np.abs(so[:, :NDYN].astype(np.float64) - ref["state"])
        s_pose = np.maximum(self.s_pose, d32[:, POSE_COLS].max(axis=1))
        s_vel = np.maximum(self.s_vel, d32[:, VEL_COLS].max(axis=1))
        tol_o = np.maximum(OBS_TOL, SENS_FACTOR * s_obs)
        tol_r = np.maximum(REW_TOL, SENS_FACTOR * s_rew)
        tol_p = np.maximum(POSE_TOL, SENS_FACTOR * s_pose)
        tol_v = np.maximum(VEL_TOL, SENS_FACTOR * s_vel)
        e_obs = np.abs(g_obs - b["obs"]).max(axis=1)
        e_rew = np.abs(g_rew - b["rew"])
        g_dyn = np.asarray(g_state)[:, :NDYN].astype(np.float64)
        dg = np.abs(g_dyn - so[:, :NDYN])
        e_pose, e_vel = dg[:, POSE_COLS].max(axis=1), dg[:, VEL_COLS].max(axis=1)
        int_ok = (g_int == b_int).all(axis=1)
        near = b["nnear"] > 0
        category = np.where(SENS_FACTOR * s_obs > OBS_TOL, 1, 0)
        # plain / sensitive env-steps: the oracle as it ran, integers exactly (version 3: an unstable probe excuses nothing)
        strict = (e_obs <= tol_o) & (e_rew <= tol_r) & (e_pose <= tol_p) & (e_vel <= tol_v) & int_ok
        # `beyond` (versions 1-2; with TAIL_FACTOR = 1 the set is empty by construction): outside the factor-8 bound but inside TAIL_FACTOR x it.  err / s has no hard limit (s is a first-order response to
        # INPUT errors, the kernel also rounds thousands of intermediates) and its tail is heavy: largest err / s 5.4, 7.2, 8.2 on
        # successive samples of 1.6e5, 3.3e5, 1.5e4 env-steps.  Such env-steps are counted and must stay below BEYOND_MAX_FRACTION
        # (callers assert it, with the 99.9 % quantile of err / bound); anything beyond TAIL_FACTOR x the bound fails.
        # (the OBSERVATION's plain bound -- the north star's 1e-4 -- has no tail: only its sensitivity-scaled bounds do.  Reward, pose and
        # rates have: their floors are conventions of this rule, and the reward's is tight -- 60 x a planar position error -- one of 327 680
        # GPU env-steps had a reward error of 1.04e-4 at a floor-level bound, profiles/r04_v5_parity_rule_stats.txt)
        wide = ((e_obs <= np.where(tol_o > OBS_TOL, TAIL_FACTOR * tol_o, tol_o)) & (e_rew <= TAIL_FACTOR * tol_r) &
                (e_pose <= TAIL_FACTOR * tol_p) & (e_vel <= TAIL_FACTOR * tol_v) & int_ok)          # TAIL_FACTOR = 1: identical to `strict`
        ok = wide.copy()
        matched_e = e_obs.copy()
        int_excused = ~int_ok & ~stable & ~near
        if near.any():
            self._branches(st, st64, act, b, b_int, near, g_obs, g_rew, g_int, e_rew, ok, matched_e, category, tol_o, g_dyn, e_pose, e_vel,
                           tol_r, tol_p, tol_v)
        beyond = ok & ~strict & ~near          # (near-threshold env-steps are judged by _branches, strictly)
        return dict(ok=ok, beyond=beyond, e_obs=e_obs, e_rew=e_rew, matched_e=matched_e, tol=tol_o, s=s_obs, category=category, near=near,
                    int_ok=int_ok, int_excused=int_excused, e_o32_o64=np.abs(b["obs"] - ref["obs"]).max(axis=1),
                    e_hip_o64=np.abs(g_obs - ref["obs"]).max(axis=1), tol_rew=tol_r, g_int=g_int, b_int=b_int, stable=stable,
                    e_pose=e_pose, e_vel=e_vel, tol_pose=tol_p, tol_vel=tol_v, oracle=b, next_state=so,
                    loose=(tol_o > OBS_CEIL) | (tol_r > REW_CEIL) | (tol_p > POSE_CEIL) | (tol_v > VEL_CEIL))

    def _branches(self, st, st64, act, b, b_int, near, g_obs, g_rew, g_int, e_rew, ok, matched_e, category, tol_o, g_dyn, e_pose, e_vel,
                  tol_r, tol_p, tol_v):
        """Env-steps with a near-threshold decision: search the alternative branches (module docstring)."""
        n, cap = self.n, NEAR_LIST
        envs = np.nonzero(near)[0]
        # per env: queue of forced sets still to evaluate, the sets seen, the closest integer-exact branch so far
        queue = {e: [(int(i),) for i in b["near"][e, :min(b["nnear"][e], cap)]] for e in envs}
        seen = {e: set(queue[e]) | {()} for e in envs}
        best, resolved = {}, {}
        for e in envs:
            same_int = bool((g_int[e] == b_int[e]).all())
            best[e] = (matched_e[e] if same_int else np.inf, ())
            # the branch the oracle took: the step's own bounds for reward and state (measured on this very branch), 1e-4 for the observation
            resolved[e] = (same_int and matched_e[e] <= OBS_TOL and e_rew[e] <= tol_r[e] and e_pose[e] <= tol_p[e] and e_vel[e] <= tol_v[e])
            ok[e] = resolved[e]
            category[e] = 0
        done_runs = {e: 0 for e in envs}
        while True:
            todo = [e for e in envs if not resolved[e] and queue[e] and done_runs[e] < MAX_ALTERNATIVES]
            if not todo:
                break
            force = np.zeros((n, cap), np.int32)
            nforce = np.zeros(n, np.int32)
            cur = {}
            for e in todo:
                S = queue[e].pop(0)
                cur[e] = S
                force[e, :len(S)] = S
                nforce[e] = len(S)
                done_runs[e] += 1
            self.alt.set_state(st)
            r = self.alt.step_ex(act, tol=NEAR_TOL, cap=cap, force=force, nforce=nforce)
            a_state = self.alt.get_state()
            a_int = _ints(a_state, r["done"], r["info"])
            for e in todo:
                S = cur[e]
                ea = np.abs(g_obs[e] - r["obs"][e]).max()
                er = abs(g_rew[e] - r["rew"][e])
                da = np.abs(g_dyn[e] - a_state[e, :NDYN])
                same_int = (g_int[e] == a_int[e]).all()
                if same_int and ea < best[e][0]:
                    best[e] = (ea, S)
                # (reward and state of an alternative branch: the original branch's measured bounds serve as the scale -- the
                # fall-back below measures the alternative's own)
                if same_int and ea <= OBS_TOL and er <= tol_r[e] and da[POSE_COLS].max() <= tol_p[e] and da[VEL_COLS].max() <= tol_v[e]:
                    resolved[e] = True
                    ok[e] = True
                    matched_e[e] = ea
                    category[e] = 2
                    continue
                if len(S) < MAX_DEPTH:
                    for i in r["near"][e, :min(r["nnear"][e], cap)]:
                        T = tuple(sorted(set(S) | {int(i)}))
                        if T not in seen[e]:
                            seen[e].add(T)
                            queue[e].append(T)
        # what is left: the closest integer-exact branch, held to its own frozen-decision sensitivity
        left = [e for e in envs if not resolved[e] and np.isfinite(best[e][0])]
        if left:
            force = np.zeros((n, cap), np.int32)
            nforce = np.zeros(n, np.int32)
            for e in left:
                S = best[e][1]
                force[e, :len(S)] = S
                nforce[e] = len(S)
            self.alt.set_state(st)
            ra = self.alt.step_ex(act, cap=cap, force=force, nforce=nforce, record=True)
            ra_state = self.alt.get_state()
            self.o64.set_state(st64)
            r6 = self.o64.step_ex(act, replay=ra["trace"])
            ref = dict(obs=r6["obs"].astype(np.float64), rew=r6["rew"].astype(np.float64), state=self.o64.get_state()[:, :NDYN].copy(), ints=None)
            s_o, s_r, _ = self._sensitivity(st64, act, ref, replay=ra["trace"])
            s_o = np.maximum(s_o, np.abs(ra["obs"] - ref["obs"]).max(axis=1))
            s_r = np.maximum(s_r, np.abs(ra["rew"] - ref["rew"]))
            d32 = np.abs(ra_state[:, :NDYN].astype(np.float64) - ref["state"])
            s_p = np.maximum(self.s_pose, d32[:, POSE_COLS].max(axis=1))
            s_v = np.maximum(self.s_vel, d32[:, VEL_COLS].max(axis=1))
            for e in left:
                ea = np.abs(g_obs[e] - ra["obs"][e]).max()
                er = abs(g_rew[e] - ra["rew"][e])
                da = np.abs(g_dyn[e] - ra_state[e, :NDYN])
                lim_o, lim_r = max(OBS_TOL, SENS_FACTOR * s_o[e]), max(REW_TOL, SENS_FACTOR * s_r[e])
                lim_p, lim_v = max(POSE_TOL, SENS_FACTOR * s_p[e]), max(VEL_TOL, SENS_FACTOR * s_v[e])
                ok[e] = bool(ea <= lim_o and er <= lim_r and da[POSE_COLS].max() <= lim_p and da[VEL_COLS].max() <= lim_v)
                matched_e[e] = ea
                tol_o[e] = lim_o
                category[e] = 3 if len(best[e][1]) else 1
        for e in envs:
            if not resolved[e] and not np.isfinite(best[e][0]):
                ok[e] = False


def summarize(results):
    """Concatenate the per-step dicts of judge() and return (arrays, text)."""
    keys = ("ok", "e_obs", "e_rew", "matched_e", "tol", "s", "category", "near", "int_ok", "int_excused", "e_o32_o64", "e_hip_o64",
            "tol_rew", "e_pose", "e_vel", "tol_pose", "tol_vel", "loose", "beyond")
    r = {k: np.concatenate([x[k] for x in results]) for k in keys}
    cat = r["category"]
    plain = cat == 0
    txt = ("%d env-steps: %d plain = %.0f %% held to 1e-4 (max |obs| err %.2e), %d sensitive (max err / bound %.2f), %d matched another "
           "branch, %d another branch + sensitive; reward max err / bound %.2f, pose %.2f, velocities %.2f; integer mismatches excused by an "
           "unstable probe (version 3: these are failures): %d; loose bounds: %d; between 1 x and TAIL_FACTOR x their bound (version 3: empty): %d; failures: %d" % (
               cat.size, plain.sum(), 100.0 * plain.mean(), r["matched_e"][plain].max() if plain.any() else 0.0, (cat == 1).sum(),
               (r["matched_e"] / r["tol"])[cat == 1].max() if (cat == 1).any() else 0.0, (cat == 2).sum(), (cat == 3).sum(),
               (r["e_rew"] / r["tol_rew"])[cat < 2].max() if (cat < 2).any() else 0.0, (r["e_pose"] / r["tol_pose"])[cat < 2].max() if (cat < 2).any() else 0.0,
               (r["e_vel"] / r["tol_vel"])[cat < 2].max() if (cat < 2).any() else 0.0, r["int_excused"].sum(), r["loose"].sum(), r["beyond"].sum(), (~r["ok"]).sum()))
    return r, txt
