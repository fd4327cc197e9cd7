"""The parity rule is frozen for the round: any edit to tests/parity_rule.py -- constants, search depth, tail factor,
a docstring -- changes its SHA-256 and fails here.  Re-locking is a deliberate, visible act: `sha256sum tests/parity_rule.py >
tests/parity_rule.lock` in a commit of its own whose message says why."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def test_parity_rule_is_the_locked_one():
    want = open(os.path.join(HERE, "parity_rule.lock")).read().split()[0]
    have = hashlib.sha256(open(os.path.join(HERE, "parity_rule.py"), "rb").read()).hexdigest()
    assert have == want, "tests/parity_rule.py was edited after it was frozen"


def test_lock_file_carries_the_history_of_every_version():
    lines = [ln.split() for ln in open(os.path.join(HERE, "parity_rule.lock")) if ln.strip() and not ln.startswith("#")]
    assert [ln[0][:8] for ln in lines] == ["%s" % h for h in (lines[0][0][:8], "f462080d", "e1a62ad8")]       # current, version 2, version 1


def test_frozen_constants_are_the_round_4_values_the_search_capacity_is_version_2s_and_the_hatches_are_closed():
    import parity_rule as pr
    assert (pr.OBS_TOL, pr.REW_TOL, pr.NEAR_TOL, pr.POSE_TOL, pr.VEL_TOL) == (1e-4, 1e-4, 1e-5, 1e-4, 1e-3)
    assert (pr.OBS_CEIL, pr.POSE_CEIL, pr.VEL_CEIL, pr.REW_CEIL, pr.LOOSE_MAX_FRACTION) == (5e-3, 5e-3, 5e-2, 5e-2, 1e-2)
    assert (pr.ULPS, pr.SENS_FACTOR, pr.MAX_DEPTH) == (8.0, 8.0, 3)
    assert (pr.TAIL_FACTOR, pr.BEYOND_MAX_FRACTION) == (1.0, 0.0)          # version 3: no tail beyond max(floor, 8 s) (versions 1-2: 2.0, 2e-4)
    import parity_assert as pa
    assert (pa.Q999_ERR_OVER_BOUND_MAX, pa.PLAIN_MIN_FRACTION) == (0.2, 0.30)
    assert (pr.NEAR_LIST, pr.MAX_ALTERNATIVES) == (16, 40)          # version 2: the search's capacity, the only change (module docstring)
