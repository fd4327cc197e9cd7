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
