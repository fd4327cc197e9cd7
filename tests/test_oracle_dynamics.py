"""Pins the CPU oracle's dynamics against (a) the Random123 Philox known answers, (b) an independent numpy
RNEA/dense-solve forward dynamics, (c) analytic invariants.  CPU only."""
import numpy as np
import pytest

import np_dynamics as npd
import oracle_lib as ol
from steppingstone_amd import model as M


def random_state(m, rng, vel_scale=1.0, outside_limits=False):
    st = np.zeros(ol.STATE_DIM)
    st[ol.S_POS] = rng.normal(size=3) * 0.3 + np.array([0, 0, 5.0])
    qt = rng.normal(size=4)
    st[ol.S_QUAT] = qt / np.linalg.norm(qt)
    st[ol.S_VEL] = rng.normal(size=6) * vel_scale
    lo, hi = m["range"][:, 0], m["range"][:, 1]
    q = lo + (hi - lo) * rng.uniform(0.05, 0.95, size=21)
    if outside_limits:
        q[::3] = hi[::3] + 0.05
        q[1::5] = lo[1::5] - 0.03
    st[ol.S_Q] = q
    st[ol.S_QD] = rng.normal(size=21) * 2.0 * vel_scale
    st[ol.S_N] = 1
    st[ol.S_TERRAIN] = 0
    return st


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [hex(x) for x in ol.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    f = 0xFFFFFFFF
    assert [hex(x) for x in ol.philox([f, f, f, f], [f, f])] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    out = ol.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])
    assert [hex(x) for x in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


@pytest.mark.parametrize("kind", ["walker3d", "mike"])
def test_fk_matches_numpy(kind):
    m = M.build(kind)
    rng = np.random.default_rng(1)
    for _ in range(5):
        st = random_state(m, rng)
        pos, rot = ol.debug_fk(kind, st)
        R, p = M.fk(m, st[ol.S_Q], st[ol.S_POS], npd.quat_rot(st[ol.S_QUAT]))
        # tables are float32-rounded copies of the float64 model
        assert np.allclose(pos, np.array(p), atol=2e-6)
        assert np.allclose(rot, np.array(R), atol=1e-6)


@pytest.mark.parametrize("kind", ["walker3d", "mike"])
@pytest.mark.parametrize("outside", [False, True])
def test_aba_matches_dense_solve(kind, outside):
    m = M.build(kind)
    rng = np.random.default_rng(2)
    for _ in range(6):
        st = random_state(m, rng, outside_limits=outside)
        tau = rng.uniform(-1, 1, 21) * m["torque"]
        qdd, a0 = ol.debug_aba(kind, st, tau, prec="f64")
        qdd_ref, a0_ref, _ = npd.forward_dynamics(m, st[ol.S_QUAT], st[ol.S_VEL], st[ol.S_Q], st[ol.S_QD], tau)
        scale = max(1.0, np.abs(qdd_ref).max())
        assert np.abs(qdd - qdd_ref).max() / scale < 2e-5, (np.abs(qdd - qdd_ref).max(), scale)
        assert np.abs(a0 - a0_ref).max() / max(1.0, np.abs(a0_ref).max()) < 2e-5


def test_momentum_balance_instantaneous():
    """Exact invariant of the continuous dynamics the ABA solves: with no contact, dP/dt = M g and dL_com/dt = 0
    whatever the (internal) joint torques are.  Checked by a central finite difference along the ABA's own
    accelerations (fp64), so it is independent of the integrator."""
    kind = "walker3d"
    m = M.build(kind)
    m0 = dict(m)
    # remove the implicit-in-h terms' O(h) bias from the comparison: they are internal torques anyway
    rng = np.random.default_rng(3)
    mass = m["mass"].sum()
    for _ in range(4):
        st = random_state(m, rng, vel_scale=0.7)
        tau = rng.uniform(-0.5, 0.5, 21) * m["torque"]
        qdd, a0 = ol.debug_aba(kind, st, tau, prec="f64")

        def advance(eps):
            s = st.copy()
            R = npd.quat_rot(st[ol.S_QUAT])
            s[ol.S_POS] += eps * (R @ st[ol.S_VEL][3:])
            w = st[ol.S_VEL][:3]
            qw, qx, qy, qz = st[ol.S_QUAT]
            dq = 0.5 * np.array([-qx * w[0] - qy * w[1] - qz * w[2], qw * w[0] + qy * w[2] - qz * w[1],
                                 qw * w[1] - qx * w[2] + qz * w[0], qw * w[2] + qx * w[1] - qy * w[0]])
            qn = st[ol.S_QUAT] + eps * dq
            s[ol.S_QUAT] = qn / np.linalg.norm(qn)
            s[ol.S_Q] += eps * st[ol.S_QD]
            s[ol.S_VEL] += eps * a0
            s[ol.S_QD] += eps * qdd
            return npd.com_and_momentum(m, s[ol.S_POS], s[ol.S_QUAT], s[ol.S_VEL], s[ol.S_Q], s[ol.S_QD])

        eps = 1e-6
        _, Pp, Lp = advance(eps)
        _, Pm, Lm = advance(-eps)
        dP, dL = (Pp - Pm) / (2 * eps), (Lp - Lm) / (2 * eps)
        assert np.allclose(dP, mass * np.array([0, 0, -9.8]), atol=1e-3), dP
        assert np.abs(dL).max() < 1e-3, dL


def test_free_flight_integrated():
    """Integrated over 0.25 s of free flight the first-order (semi-implicit Euler, body-frame) scheme keeps the
    COM on the ballistic parabola and the angular momentum about the COM to O(h)."""
    kind = "walker3d"
    m = M.build(kind)
    rng = np.random.default_rng(3)
    env = ol.OracleEnv(kind, 1, prec="f64")
    st = random_state(m, rng, vel_scale=0.5)
    env.set_state(st[None])
    com0, P0, L0 = npd.com_and_momentum(m, st[ol.S_POS], st[ol.S_QUAT], st[ol.S_VEL], st[ol.S_Q], st[ol.S_QD])
    nsub = 60
    env.substeps(0, np.zeros(21), nsub)
    s1 = env.get_state()[0]
    com1, P1, L1 = npd.com_and_momentum(m, s1[ol.S_POS], s1[ol.S_QUAT], s1[ol.S_VEL], s1[ol.S_Q], s1[ol.S_QD])
    t = nsub / 240.0
    mass = m["mass"].sum()
    g = np.array([0, 0, -9.8])
    assert np.allclose(P1 / mass, P0 / mass + g * t, atol=0.03)
    assert np.allclose(com1, com0 + P0 / mass * t + 0.5 * g * t * t, atol=0.02)
    assert np.abs(L1 - L0).max() < 0.1 * max(1.0, np.abs(L0).max())


def test_oracle_f32_tracks_f64_short_horizon():
    kind = "walker3d"
    e32, e64 = ol.OracleEnv(kind, 1, seed=5, prec="f32"), ol.OracleEnv(kind, 1, seed=5, prec="f64")
    o32, o64 = e32.reset(), e64.reset()
    assert np.allclose(o32, o64, atol=1e-5)
    for t in range(10):
        a = e32.random_actions(t)
        o32, r32, d32, _ = e32.step(a)
        o64, r64, d64, _ = e64.step(a)
        assert d32[0] == d64[0]
        assert np.allclose(o32, o64, atol=5e-3), (t, np.abs(o32 - o64).max())


@pytest.mark.parametrize("kind", ["walker3d", "mike"])
def test_resting_contact_is_bounded(kind):
    """Zero torques from the reset pose: the soles must stay on stone 0 (penetration bounded, no tunnelling, no
    blow-up) while the passive body starts to sag."""
    m = M.build(kind)
    env = ol.OracleEnv(kind, 1, seed=11, prec="f32")
    env.reset()
    flags, both = None, 0
    for k in range(12):           # 12 x 4 substeps = 0.2 s
        flags = env.substeps(0, np.zeros(21), 4)
        both += int(flags[0] == 1 and flags[1] == 1)
        st = env.get_state()[0].astype(np.float64)
        pos, rot = ol.debug_fk(kind, st)
        ec = M.env_constants()
        for b in (M.RIGHT_FOOT_BODY, M.LEFT_FOOT_BODY):
            P = [pos[b] + rot[b] @ c for c in m["corners"]]
            # corners over stone 0 (inside its plank footprint around the origin, heading +x) are held by the contact model; a corner
            # that hangs over the rim is not
            zc = [p[2] for p in P if abs(p[0]) < ec["stone_plank_half_length"] - 0.005 and abs(p[1]) < ec["stone_plank_half_width"] - 0.005]
            assert len(zc) >= 2 and min(zc) > -0.012, (k, b, zc)
        assert np.isfinite(st).all()
        assert np.abs(st[ol.S_VEL]).max() < 10 and np.abs(st[ol.S_QD]).max() < 60
    # both feet carry the robot while it stands; as the passive body sags one sole may peel off the 0.6 m long plank before 0.2 s are over
    assert both >= 6 and (flags[0] == 1 or flags[1] == 1), (both, flags)
