"""Host-side vec-env / single-env facades (steppingstone_amd.envs) driven by an oracle-backed test double: types,
shapes, hook plumbing and error behaviour of the reference protocol (common/envs_utils.py:542-606).  CPU only."""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from oracle_backend import OracleBackend
from steppingstone_amd.envs import SteppingStoneEnv, SteppingStoneVecEnv, kind_of


def make(n, env_id="Walker3DStepperEnv-v0", seed=3, numpy_mode=True):
    return SteppingStoneVecEnv(env_id, n, seed=seed, return_numpy=numpy_mode, backend=OracleBackend(kind_of(env_id), n, seed))


def test_numpy_mode_types_and_infos():
    env = make(6)
    assert env.observation_space.shape == (60,) and env.action_space.shape[0] == 21
    obs = env.reset()
    assert obs.shape == (6, 60) and obs.dtype == np.float32
    got_episode = False
    for t in range(60):
        obs, rew, done, infos = env.step(env.random_actions(t).numpy())
        assert rew.shape == (6,) and rew.dtype == np.float64 and done.dtype == bool and len(infos) == 6
        for i, info in enumerate(infos):
            if done[i]:
                got_episode = True
                assert set(info["episode"]) == {"r", "l", "t"} and info["episode"]["l"] >= 1
                assert info["episode"]["r"] == round(info["episode"]["r"], 6)
            else:
                assert "episode" not in info.keys() and "bad_transition" not in info.keys()
    assert got_episode
    env.close()
    env.close()      # idempotent


def test_tensor_mode_and_hooks():
    env = make(5, "MikeStepperEnv-v0", numpy_mode=False)
    obs = env.reset()
    assert torch.is_tensor(obs) and obs.shape == (5, 60)
    obs, rew, done, info = env.step(torch.zeros(5, 21))
    assert done.dtype == torch.bool and set(info) == {"ep_ret", "ep_len", "bad_transition", "steps_reached", "update_terrain", "ep_ret_lo"}
    env.update_curriculum(3)
    env.update_specialist(2)
    env.update_sample_prob(np.full((5, 11, 11), 1 / 121.0))
    env.update_sample_prob(np.full((11, 11), 1 / 121.0))
    per_env = np.random.default_rng(0).random((5, 11, 11))
    env.update_sample_prob(per_env / per_env.sum(axis=(1, 2), keepdims=True))
    with pytest.raises(ValueError):
        env.update_sample_prob(np.zeros((4, 11, 11)))
    env.set_mirror(True)
    env.set_robot_params({"power": 0.8})
    env.set_env_params({"curriculum": 1})
    with pytest.raises(KeyError):
        env.set_robot_params({"mass": 2})
    assert env.create_temp_states().shape == (5, 121, 60)
    assert env.terrain_info.shape == (5, 20, 6) and env.next_step_index.shape == (5,)
    assert env.yaw_samples.shape == (11,) and np.isclose(env.pitch_samples[-1], np.deg2rad(30))
    with pytest.raises(NotImplementedError):
        env.get_images()
    with pytest.raises(AssertionError):
        env.step(torch.zeros(4, 21))


def test_state_roundtrip_matches_oracle_directly():
    env = make(4, seed=9)
    o = ol.OracleEnv("walker3d", 4, seed=9)
    assert np.array_equal(env.reset(), o.reset())
    for t in range(5):
        a = o.random_actions(t)
        eo, er, ed, _ = env.step(a)
        oo, orr, od, _ = o.step(a)
        assert np.array_equal(eo, oo) and np.array_equal(ed, od.astype(bool))
    st = env.get_state()
    env.set_state(st)
    assert np.array_equal(env.get_state().numpy(), st.numpy())


def test_single_env_facade():
    env = SteppingStoneEnv("mocca_envs:Walker3DStepperEnv-v0", seed=0, backend_factory=lambda k, n, s: OracleBackend(k, n, s))
    env.seed(1093)
    obs = env.reset()
    assert obs.shape == (60,) and env.spec.id.endswith("Walker3DStepperEnv-v0") and env._max_episode_steps == 1000
    total = 0.0
    for t in range(300):
        obs, r, d, info = env.step(np.zeros(21, np.float32))
        assert isinstance(r, float) and isinstance(d, bool)
        total += r
        if d:
            assert abs(info["episode"]["r"] - total) < 1e-3 * max(1, abs(total))
            assert abs(obs[0]) > 0.02              # terminal observation (no auto-reset for the plain env)
            break
    else:
        pytest.fail("episode did not end")
    assert env.robot.feet_contact.shape == (2,)
    assert env.create_temp_states().shape == (121, 60) and len(env.get_mirror_indices()) == 6
    obs2 = env.reset()
    assert np.all(obs2[:6] == 0)
    env.update_curriculum(2)
    env.update_sample_prob(np.full((11, 11), 1 / 121.0))
    env.close()
