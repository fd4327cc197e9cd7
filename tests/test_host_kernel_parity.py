"""Pre-flight for the GPU-less container: the product's kernel source (steppingstone_amd/csrc/*.hpp) compiled for the
CPU by tests/host/host_harness.cpp must agree with the oracle.  This exercises the same per-lane code the GPU runs
(ABA, contact rows, PGS, reward, reset, sampler); the real parity tests are the `-m gpu` ones.  CPU only."""
import shutil

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")

INT_FIELDS = [ol.S_N, ol.S_COUNT, ol.S_ELAPSED, ol.S_CTRLO, ol.S_CTRHI, ol.S_FLAGS]


@pytest.mark.parametrize("kind,k", [("walker3d", 0), ("mike", 1)])
def test_step_matches_oracle(kind, k):
    import host_lib as hl
    n = 48
    o = ol.OracleEnv(kind, n, seed=11)
    o.reset()
    bad = total = 0
    for t in range(25):
        st, a = o.get_state(), o.random_actions(t)
        oo, ro, do, io = o.step(a)
        so = o.get_state()
        sh, oh, rh, dh, ih = hl.step(k, st, a, seed=11)
        ok = (np.abs(oh - oo).max(axis=1) < 2e-3) & ((np.abs(sh - so) / (2e-3 + 2e-3 * np.abs(so))).max(axis=1) < 1) & \
             (np.abs(rh - ro) < 2e-2) & (dh == do) & (sh[:, INT_FIELDS] == so[:, INT_FIELDS]).all(axis=1) & \
             (ih["bad_transition"] == io["bad_transition"]) & (ih["update_terrain"] == io["update_terrain"])
        bad += int((~ok).sum())
        total += n
    assert bad <= 0.01 * total, (bad, total)


def test_target_advance_and_sampler_match_oracle():
    import host_lib as hl
    n = 40
    rng = np.random.default_rng(0)
    prob = rng.random((11, 11))
    prob /= prob.sum()
    o = ol.OracleEnv("walker3d", n, seed=5)
    o.set_curriculum(5)
    o.set_sample_prob(prob)
    o.reset()
    st = o.get_state()
    st[:, 0] = st[:, 65 + 6]
    o.set_state(st)
    zero = np.zeros((n, 21), np.float32)
    adv = np.zeros(n, bool)
    for t in range(5):          # (the robot is released 1 cm above the stone: the feet are down for two consecutive steps by step 3-5)
        st = o.get_state()
        oo, ro, do, io = o.step(zero)
        so = o.get_state()
        sh, oh, rh, dh, ih = hl.step(0, st, zero, seed=5, curriculum=5, prob=prob)
        assert np.array_equal(ih["update_terrain"], io["update_terrain"])
        assert np.array_equal(sh[:, INT_FIELDS], so[:, INT_FIELDS])
        assert np.abs(sh[:, 65:185] - so[:, 65:185]).max() < 1e-5
        assert np.abs(rh - ro).max() < 2e-2
        adv |= io["update_terrain"].astype(bool)
    assert adv.mean() > 0.5


@pytest.mark.parametrize("kind,k", [("walker3d", 0), ("mike", 1)])
def test_steps_of_a_walking_policy_on_turned_and_tilted_stones_match_oracle(kind, k):
    """Round 6: the stones' stepping surface is a plank aligned with the stone's HEADING, which the kernels carry as (cos, sin) per active
    stone in fstate / LDS (reset, advance, set_state).  Random actions from the reset pose never leave the straight start of the course, so
    here the reference's shipped actor walks the oracle at curriculum 3 until the stones in play are turned and tilted, and the kernel
    source compiled for the host must then agree with the oracle step by step (contacts on, target advances included)."""
    import torch
    import host_lib as hl
    import shipped_actor as sa
    n = 48
    o = ol.OracleEnv(kind, n, seed=21)
    o.set_curriculum(3)
    obs = o.reset()
    actor = sa.load_actor(kind)
    for t in range(110):           # (first episodes: both robots are on their third / fourth stone; Mike falls soon after at this level)
        with torch.no_grad():
            obs, _, _, _ = o.step(actor(torch.from_numpy(obs)).numpy())
    st = o.get_state()
    terr = st[:, 65:185].reshape(n, 20, 6)
    nidx = st[:, ol.S_N].astype(int)
    turned = np.abs(terr[np.arange(n), nidx, 3]) > 0.03
    assert turned.sum() >= n // 2, "the sample does not exercise turned stones"
    bad = total = advanced = contacts = 0
    for t in range(14):
        st = o.get_state()
        with torch.no_grad():
            a = actor(torch.from_numpy(o.get_obs())).numpy().astype(np.float32)
        oo, ro, do, io = o.step(a)
        so = o.get_state()
        sh, oh, rh, dh, ih = hl.step(k, st, a, seed=21, curriculum=3)
        ok = (np.abs(oh - oo).max(axis=1) < 2e-3) & ((np.abs(sh - so) / (2e-3 + 2e-3 * np.abs(so))).max(axis=1) < 1) & \
             (np.abs(rh - ro) < 2e-2) & (dh == do) & (sh[:, INT_FIELDS] == so[:, INT_FIELDS]).all(axis=1) & \
             (ih["update_terrain"] == io["update_terrain"])
        bad += int((~ok).sum())
        total += n
        advanced += int(io["update_terrain"].sum())
        contacts += int(((so[:, ol.S_FLAGS].astype(int) & 3) != 0).sum())
    assert advanced >= 3 and contacts > total // 2, (advanced, contacts)
    assert bad <= 0.02 * total, (bad, total)
