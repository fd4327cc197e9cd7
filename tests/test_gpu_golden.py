"""The reference-pinned golden vectors (tests/golden/harness_golden.npz: the reference's own RolloutStorage.compute_returns,
get_mirror_function and PPO.update, tools/make_golden.py) evaluated ON THE GPU: the device-resident learner pieces of rows
f-1 / f-2 / f-4 run where they run in production.  `pytest -m gpu`."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,use_gae", [("gae", True), ("ret", False)])
def test_returns_match_reference_on_gpu(tag, use_gae):
    from test_harness_golden import check_returns
    check_returns("cuda:0", tag, use_gae)


def test_mirror_function_matches_reference_on_gpu():
    from test_harness_golden import check_mirror
    check_mirror("cuda:0")


def test_one_ppo_update_matches_reference_on_gpu():
    """eager step on the GPU: same three losses, same weights after one Adam step as the reference's PPO.update"""
    from test_ppo_golden import check_one_ppo_update
    check_one_ppo_update("cuda:0", loss_rtol=1e-4, loss_atol=1e-5, w_rtol=1e-3, w_atol=1e-5)


def test_one_ppo_update_through_the_hipgraph_step_matches_reference():
    """the captured minibatch step (gather, forward, backward, clip, capturable Adam) on the same golden batch"""
    from test_ppo_golden import check_one_ppo_update
    check_one_ppo_update("cuda:0", use_graph=True, loss_rtol=1e-4, loss_atol=1e-5, w_rtol=2e-3, w_atol=2e-5)


def test_threshold_sampler_runs_on_gpu_and_feeds_the_device_hook():
    """f-2 on the GPU: batched evaluation env -> critic ensemble over create_temp_states -> softmax grid as a device
    tensor -> ss_set_sample_prob_device; the stones drawn afterwards come from that grid."""
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    ac = ppo.ActorCritic(num_ensembles=2).to(dev)
    ev = SteppingStoneVecEnv("MikeStepperEnv-v0", 64, seed=1, device=dev, env_id_offset=10000, return_numpy=False)
    ev.update_curriculum(0)
    orig = ev.reset

    def reset_on_target():
        orig()
        st = ev.get_state()
        st[:, 0] = st[:, 65 + 6]
        ev.set_state(st)
        return ev.get_obs()

    ev.reset = reset_on_target
    for mode in ("threshold", "adaptive"):
        p = ppo.sampling_probs_from_values(ac, ev, mode=mode, as_tensor=True)
        assert p is not None and p.is_cuda and p.shape == (11, 11) and abs(float(p.sum()) - 1) < 1e-5 and bool((p > 0).all())
        pn = ppo.sampling_probs_from_values(ac, ev, mode=mode)        # a second evaluation rollout (other episodes): numpy form
        assert pn.dtype == np.float64 and pn.shape == (11, 11) and abs(pn.sum() - 1) < 1e-5 and (pn > 0).all()
    # a one-hot grid through the device hook: every drawn stone has exactly that cell's yaw
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", 256, seed=2, device=dev, return_numpy=False)
    envs.update_curriculum(5)
    onehot = torch.zeros((11, 11), device=dev)
    onehot[9, 2] = 1.0                                     # yaw +16 deg, pitch -18 deg
    envs.update_sample_prob(onehot)
    envs.reset()
    st = envs.get_state()
    st[:, 0] = st[:, 65 + 6]
    envs.set_state(st)
    act = torch.zeros((256, 21), device=dev)
    for _ in range(4):
        envs.step(act)
    st = envs.get_state()
    drawn = st[:, 59] >= 2
    assert drawn.float().mean() > 0.5
    phi = st[drawn, 65 + 3 * 6 + 3] * (180.0 / 3.141592653589793)
    assert (phi - 16.0).abs().max() < 1e-3
    ev.close(); envs.close()
