"""The reference's shipped deterministic actors in the HIP env (through the C ABI), on the GPU box -- where /root/reference does not
exist: their weights are plain arrays under tests/golden/ (tools/make_golden_policy.py).  `playground/enjoy.py:143-235` is the
reference's own use of these files: load the policy, walk the course.  With the robot numbers identified in round 5
the policies walk it in the PRODUCT env, not only in the CPU oracle; thresholds as in tests/test_shipped_policy_walks.py.
`pytest -m gpu`."""
import numpy as np
import pytest

import shipped_actor as sa
from test_shipped_policy_walks import CASES

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_id,kind,n,steps,mean_min,median_min", CASES)
def test_shipped_actor_walks_the_course_on_hip(env_id, kind, n, steps, mean_min, median_min):
    from steppingstone_amd import model
    from steppingstone_amd.envs import SteppingStoneVecEnv
    if not model.identified(kind):
        pytest.skip("no identified numbers for %s yet (steppingstone_amd/identified_%s.json)" % (kind, kind))
    n = 1024
    env = SteppingStoneVecEnv(env_id, n, seed=31, device="cuda:0", return_numpy=False)
    stones, length, alive = sa.walk(env, sa.load_actor(kind, "cuda:0"), 900, lambda o: o, n)
    env.close()
    print("%s on the MI355X, flat terrain, %d envs: stones beyond the start mean %.2f median %.1f max %.0f; %.0f %% reach 5 stones, %.0f %% the "
          "end of the course; first-episode length mean %.0f" % (kind, n, stones.mean(), np.median(stones), stones.max(),
                                                                100 * (stones >= 5).mean(), 100 * (stones >= 18).mean(), length.mean()))
    assert stones.mean() >= mean_min and np.median(stones) >= median_min


def test_the_references_other_walker3d_actor_walks_on_hip():
    """`Walker3DStepperEnv-v0_base.pt` (the reference's flat-terrain starting policy; tests/test_shipped_policy_walks.py) in the PRODUCT env."""
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 1024
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=31, device="cuda:0", return_numpy=False)
    stones, length, alive = sa.walk(env, sa.load_actor("walker3d_base", "cuda:0"), 900, lambda o: o, n)
    env.close()
    print("_base on the MI355X, flat terrain, %d envs: stones beyond the start mean %.2f median %.1f max %.0f; %.0f %% reach 5 stones; "
          "first-episode length mean %.0f" % (n, stones.mean(), np.median(stones), stones.max(), 100 * (stones >= 5).mean(), length.mean()))
    assert stones.mean() >= 8.0 and (stones >= 5).mean() >= 0.6
