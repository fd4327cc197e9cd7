"""What every caller of the frozen parity rule (tests/parity_rule.py) asserts about a judged sample -- ONE set of thresholds for the GPU
tests, __graft_entry__.smoke() and tools/parity_heldout.py (the smoke had its own, tuned to its sample).
TEST INFRASTRUCTURE (imports the oracle through parity_rule)."""
import numpy as np

import parity_rule as pr

# How far the sensitivity-scaled bounds of a sample may sit above the 1e-4 floor -- the guard against "the rule quietly turns into
# everything is sensitive".  HISTORY OF THIS CRITERION (the only caller-side threshold that changed in round 5; tests/parity_rule.py is
# untouched): rounds 3-4 asserted "more than half of the env-steps are plain (bound == floor)".  That is a statement about the
# SPECIFICATION's conditioning on the sample, not about the kernel (the classification never looks at the HIP result), and it is
# discontinuous at the floor: with the robot numbers identified in round 5 (lighter joint damping, deeper crouch) the median bound of
# the same random-action samples moved from 1.00e-4 to 1.04e-4 -- 49 % "plain", every env-step's ERROR below 3.2e-5 -- and on a
# walking policy's states (held-out run) it is 35-44 % plain.  The criterion is therefore stated on the bounds' SIZE: the median
# bound at most 2 x the floor and the 90 % quantile at most 1e-3 (rounds 3-4 samples: 1.0e-4 / 3.7e-4; round 5: 1.04e-4 / 3.9e-4).
# The plain fraction stays the headline every caller prints (counts()["held_to_flat_1e4"]).
BOUND_MEDIAN_MAX, BOUND_Q90_MAX = 2e-4, 1e-3
# ROUND 6, with version 3 of the rule (a narrowing to what 6.38 M held-out env-steps of round 5 support; set before any new sample):
#   * the 99.9 % quantile of err / bound: < 0.2 (round 5 asserted < 0.5; measured 0.056 - 0.13 on the six held-out runs);
#   * `beyond` and `int_excused` must be ZERO (version 3 turns both into failures of the rule itself; asserted here as well);
#   * a hard floor on the plain fraction: at least 30 % of a sample's env-steps are held to the flat 1e-4.  Measured: 44-54 %
#     on random-action cells, 35-86 % on policy-driven cells, 45 % / 58 % in the driver's smoke.  It is a statement about the
#     specification on the sample (see above), so it guards the rule against drifting, not the kernel.
Q999_ERR_OVER_BOUND_MAX = 0.2
PLAIN_MIN_FRACTION = 0.30


def counts(R):
    """The headline numbers of a judged sample: the fraction held to the flat 1e-4 and every escape hatch, by name."""
    cat = R["category"]
    return dict(env_steps=int(cat.size), held_to_flat_1e4=float((cat == 0).mean()), sensitive=int((cat == 1).sum()),
                other_branch=int((cat == 2).sum()), other_branch_sensitive=int((cat == 3).sum()), int_excused=int(R["int_excused"].sum()),
                loose=int(R["loose"].sum()), beyond=int(R["beyond"].sum()), failures=int((~R["ok"]).sum()),
                bound_median=float(np.quantile(R["tol"], 0.5)), bound_q90=float(np.quantile(R["tol"], 0.9)),
                max_err_over_bound=float((R["matched_e"] / R["tol"]).max()), q999_err_over_bound=float(np.quantile(R["matched_e"] / R["tol"], 0.999)),
                within_1e4_of_oracle=float((R["e_obs"] <= 1e-4).mean()),
                far_from_fp64_hip=int((R["e_hip_o64"] > 1e-4).sum()), far_from_fp64_cpu_fp32=int((R["e_o32_o64"] > 1e-4).sum()))


def assert_judged(R, txt, label, log=print):
    log("%s: %s" % (label, txt))
    log("   bound quantiles over env-steps 50/90/99/100 %%: %s ; |hip - oracle| quantiles: %s" % (
        np.array2string(np.quantile(R["tol"], [.5, .9, .99, 1.0]), precision=2), np.array2string(np.quantile(R["e_obs"], [.5, .9, .99, 1.0]), precision=2)))
    assert R["ok"].all(), "%d env-steps outside their bound" % (~R["ok"]).sum()
    plain = R["category"] == 0
    assert plain.any() and R["matched_e"][plain].max() <= pr.OBS_TOL          # the north-star's 1e-4 wherever 8 s <= 1e-4
    assert plain.mean() >= PLAIN_MIN_FRACTION, "only %.1f %% of the env-steps are held to the flat 1e-4" % (100 * plain.mean())
    assert R["int_excused"].sum() == 0                                         # version 3: an unstable probe excuses nothing
    assert np.quantile(R["tol"], 0.5) <= BOUND_MEDIAN_MAX and np.quantile(R["tol"], 0.9) <= BOUND_Q90_MAX, (
        "bounds drifted away from the floor: median %.2e, 90 %% %.2e" % (np.quantile(R["tol"], 0.5), np.quantile(R["tol"], 0.9)))
    assert R["loose"].mean() <= max(pr.LOOSE_MAX_FRACTION, 2.0 / R["loose"].size)      # bounds beyond their ceilings stay rare
    # the tail of err / bound: version 3 has none (outside the bound is a failure, asserted above) and the bulk sits far inside:
    # 99.9 % of env-steps below a fifth of their bound (measured 0.056 - 0.13)
    assert R["beyond"].sum() == 0, "%d env-steps beyond their bound" % R["beyond"].sum()
    assert np.quantile(R["matched_e"] / R["tol"], 0.999) < Q999_ERR_OVER_BOUND_MAX
    # all env-steps, against the oracle as it ran: 99 % within the north-star's 1e-4 (measured: 99 % within 2e-5), at most 0.5 % beyond it
    assert np.quantile(R["e_obs"], 0.99) < 1e-4 and (R["e_obs"] > 1e-4).mean() < 5e-3
    # ... and against the fp64 evaluation the kernel is no noisier than the CPU's own fp32 build (which also leaves 1e-4 on ~0.16 %)
    far_hip, far_cpu = int((R["e_hip_o64"] > 1e-4).sum()), int((R["e_o32_o64"] > 1e-4).sum())
    log("   env-steps farther than 1e-4 from the fp64 oracle: HIP kernel %d, fp32 CPU oracle %d (of %d)" % (far_hip, far_cpu, R["ok"].size))
    assert far_hip <= 1.5 * far_cpu + 8
