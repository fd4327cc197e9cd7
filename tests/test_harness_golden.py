"""Host-side harness pieces against golden vectors produced by the REFERENCE's own code (tools/make_golden.py ran
the reference's RolloutStorage.compute_returns, get_mirror_function, decay schedules and its real ShmemVecEnv +
Monitor + _subproc_worker on a scripted toy env).  CPU only."""
import os

import numpy as np
import pytest
import torch

from steppingstone_amd import _lib, harness
from steppingstone_amd.envs import SteppingStoneVecEnv

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "harness_golden.npz"))


def check_returns(device, tag, use_gae):
    """algorithms/storage.py:59-82 on the golden batch (shared with the -m gpu run, tests/test_gpu_golden.py)."""
    t = lambda k: torch.from_numpy(G[tag + "_" + k]).to(device)  # noqa: E731
    vals = t("values")
    ret = harness.compute_returns(t("rewards"), vals, t("masks"), t("bad"), vals[-1].clone(), use_gae, 0.99, 0.95).cpu()
    ref = G[tag + "_returns"]
    T = ref.shape[0] - 1
    assert np.allclose(ret.numpy()[:T], ref[:T], atol=1e-6)
    if not use_gae:
        assert np.allclose(ret.numpy()[T], ref[T], atol=1e-6)


@pytest.mark.parametrize("tag,use_gae", [("gae", True), ("ret", False)])
def test_returns_match_reference(tag, use_gae):
    check_returns("cpu", tag, use_gae)


def check_mirror(device):
    """common/envs_utils.py:687-740 on the golden batch with this library's index lists."""
    obs, act = torch.from_numpy(G["mirror_obs_in"]).to(device), torch.from_numpy(G["mirror_act_in"]).to(device)
    idx = _lib.mirror_indices()
    o2, a2 = harness.mirror_batch(obs, act, idx)
    assert np.array_equal(o2.cpu().numpy(), G["mirror_obs_out"])
    assert np.array_equal(a2.cpu().numpy(), G["mirror_act_out"])
    z = torch.zeros(3, 1, device=device)
    res = harness.get_mirror_function(idx)((obs, z, act, z, z, z, z, z))
    assert len(res) == 8 and res[1].shape == (6, 1)
    assert np.array_equal(res[0].cpu().numpy(), G["mirror_obs_out"]) and np.array_equal(res[2].cpu().numpy(), G["mirror_act_out"])
    # mirroring twice is the identity
    o4, a4 = harness.mirror_batch(o2[3:], a2[3:], idx)
    assert np.array_equal(o4[3:].cpu().numpy(), obs.cpu().numpy()) and np.array_equal(a4[3:].cpu().numpy(), act.cpu().numpy())


def test_mirror_function_matches_reference():
    check_mirror("cpu")


def test_decay_schedules_match_reference():
    ep = G["decay_epochs"]
    assert np.allclose([harness.exponential_decay(int(e), 0.99, 3e-4, 3e-5) for e in ep], G["exp_decay"], rtol=0, atol=0)
    assert np.allclose([harness.linear_decay(int(e), 5000, 3e-4, 0.0) for e in ep], G["lin_decay"], rtol=0, atol=0)


class ScriptedBackend:
    """Same toy dynamics as tools/make_golden.py's ToyEnv, in the shape of a HipBackend (worker auto-reset
    included), so the vec-env facade can be compared with what the reference's ShmemVecEnv returned."""

    def __init__(self, n):
        self.device = torch.device("cpu")
        self.n = n
        self.t = np.zeros(n, np.int64)
        self.ret = np.zeros(n, np.float64)

    def _obs(self, obs):
        obs.zero_()
        obs[:, 0] = torch.arange(self.n, dtype=torch.float32)
        obs[:, 1] = torch.from_numpy(self.t.astype(np.float32))

    def reset(self, obs):
        self.t[:] = 0
        self.ret[:] = 0
        self._obs(obs)

    def step(self, act, obs, rew, done, info):
        r = 0.25 * (self.t + 1) + np.arange(self.n) + act[:, 0].numpy().astype(np.float64)
        self.t += 1
        self.ret += r
        d = self.t >= 4 + np.arange(self.n)
        raw = np.zeros((self.n, 6), np.int32)
        hi = self.ret.astype(np.float32)
        raw.view(np.float32)[:, 0] = hi
        raw.view(np.float32)[:, 1] = self.t
        raw.view(np.float32)[:, 5] = (self.ret - hi.astype(np.float64)).astype(np.float32)
        info.copy_(torch.from_numpy(raw))
        rew.copy_(torch.from_numpy(r.astype(np.float32)))
        done.copy_(torch.from_numpy(d.astype(np.uint8)))
        self.t[d] = 0
        self.ret[d] = 0
        self._obs(obs)

    def close(self):
        pass


def test_vecenv_protocol_matches_reference_shmemvecenv():
    n = 3
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, return_numpy=True, backend=ScriptedBackend(n))
    obs = env.reset()
    assert obs.dtype == np.float32 and np.array_equal(obs, G["vec_obs"][0])
    for t in range(14):
        o, r, d, infos = env.step(G["vec_actions"][t])
        assert o.dtype == np.float32 and r.dtype == np.float64 and d.dtype == bool
        assert isinstance(infos, tuple) and len(infos) == n
        assert np.array_equal(o, G["vec_obs"][t + 1])            # terminal step returns the RESET observation
        assert np.allclose(r, G["vec_rew"][t], atol=1e-6)        # ... with the terminal reward
        assert np.array_equal(d, G["vec_done"][t])
        for i in range(n):
            if d[i]:
                assert infos[i]["episode"]["r"] == G["vec_ep_r"][t, i]      # fp64 sum of the step rewards, round(., 6): Monitor.update
                assert infos[i]["episode"]["l"] == G["vec_ep_l"][t, i]
                assert "t" in infos[i]["episode"]
            else:
                assert "episode" not in infos[i].keys()
                assert np.isnan(G["vec_ep_r"][t, i])
