"""The one reference-held behavioural signal about the env's PHYSICS: `playground/enjoy.py:143-235`
loads `playground/models/*_latest.pt` and the policy walks the stepping-stone course.  Until round 4 the shipped deterministic actors
fell after 1-2 stones in this env; with the robot numbers identified in round 5 (steppingstone_amd/identified_<kind>.json, DESIGN.md
section 8) they walk it.  Runs WITHOUT the reference: the actors' weights are plain arrays under tests/golden/ (tools/make_golden_policy.py).
CPU: the oracle behind the package's own VecEnv class.  INFORMATIONAL about PyBullet parity (never a claim), but a regression guard for
the identified model: anything that breaks the robot the policies were trained on shows up here."""
import numpy as np
import pytest

import shipped_actor as sa
from oracle_backend import OracleBackend
from steppingstone_amd.envs import SteppingStoneVecEnv, kind_of

torch = pytest.importorskip("torch")

# (env id, kind, envs, steps, mean stones beyond the start >=, median >=) on flat terrain
# (measured at adoption -- round 5: Walker3D 18.0 / 18, Mike 17.5 / 18; round 6, bounded numbers on planks: 17.5 / 18 and 17.0 / 18 -- in
#  the oracle; a regression guard, set well below)
CASES = [("Walker3DStepperEnv-v0", "walker3d", 32, 700, 14.0, 17.0),
         ("MikeStepperEnv-v0", "mike", 32, 700, 12.0, 15.0)]


@pytest.mark.parametrize("env_id,kind,n,steps,mean_min,median_min", CASES)
def test_shipped_actor_walks_the_course_in_the_oracle(env_id, kind, n, steps, mean_min, median_min):
    from steppingstone_amd import model
    if not model.identified(kind):
        pytest.skip("no identified numbers for %s yet (steppingstone_amd/identified_%s.json)" % (kind, kind))
    env = SteppingStoneVecEnv(env_id, n, seed=31, return_numpy=False, backend=OracleBackend(kind_of(env_id), n, 31))
    stones, length, alive = sa.walk(env, sa.load_actor(kind), steps, lambda o: o, n)
    print("%s in the CPU oracle, flat terrain, %d envs: stones beyond the start mean %.2f median %.1f max %.0f, first-episode length mean %.0f" % (
        kind, n, stones.mean(), np.median(stones), stones.max(), length.mean()))
    assert stones.mean() >= mean_min and np.median(stones) >= median_min


def test_the_references_other_walker3d_actor_walks_too():
    """Round 6: `Walker3DStepperEnv-v0_base.pt` is the flat-terrain policy the reference's curriculum runs start from
    (playground/train.py:148-153) -- a different actor from `_latest` (`_best` carries `_latest`'s weights), less robust and so the
    sharper probe of flat-terrain physics.  Rounds 1-4: 0.0 stones; round 5 (fit to `_latest` alone): 2.0; round 6 (joint fit, `_base`
    on flat terrain in the score; stage 3): 12.1-13.1 stones on other seeds (median 13-16), 87-88 % of the episodes beyond 5 stones.  A regression guard, set below."""
    n = 64
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=31, return_numpy=False, backend=OracleBackend("walker3d", n, 31))
    stones, length, alive = sa.walk(env, sa.load_actor("walker3d_base"), 900, lambda o: o, n)      # (it walks the course more slowly than `_latest`: ~450 steps and more)
    print("_base in the CPU oracle, flat terrain, %d envs: stones beyond the start mean %.2f median %.1f max %.0f; %.0f %% reach 5 stones" % (
        n, stones.mean(), np.median(stones), stones.max(), 100 * (stones >= 5).mean()))
    assert stones.mean() >= 8.0 and (stones >= 5).mean() >= 0.6
