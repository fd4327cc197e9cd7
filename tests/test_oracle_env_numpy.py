"""Pins the ENV LOGIC of the CPU oracle's control step (docs/PHYSICS.md 4 and 5: contact flags, target logic, progress,
bonuses, termination, reward, the 60-float observation) against the independent numpy evaluation of tests/np_env.py, on
contact-rich states (curriculum-5 rollouts, robots on tilted stones).  With tests/test_oracle_dynamics.py (ABA) and
tests/test_oracle_contact.py (contact stage) every stage of the oracle's step() now has a second, independently written
evaluation.  CPU only."""
import numpy as np
import pytest

import np_contact as npc
import np_env
import oracle_lib as ol
from test_oracle_contact import contact_states

KINDS = ["walker3d", "mike"]


@pytest.mark.parametrize("kind", KINDS)
def test_control_step_outputs_match_independent_numpy(kind):
    m = npc.rounded_model(kind)
    rng = np.random.default_rng(23)
    states = np.array(contact_states(kind, rng)[:72], np.float64)
    states[:, ol.S_ELAPSED] = rng.integers(0, 990, states.shape[0])
    states[::9, ol.S_ELAPSED] = 999                               # the time limit on this very step
    states[:, ol.S_COUNT] = 0
    for e in range(0, states.shape[0], 3):                        # every third robot onto its target stone: first-touch bonus
        terrain = states[e, ol.S_TERRAIN].reshape(20, 6)
        k = int(states[e, ol.S_N])
        states[e, 0:3] += terrain[k][:3] - terrain[0][:3]
    n = states.shape[0]
    acts = rng.uniform(-1.3, 1.3, (n, 21))
    o = ol.OracleEnv(kind, n, seed=1, prec="f64")
    o.set_curriculum(5)
    o.reset()
    o.set_state(states)
    obs, rew, done, info = o.step(acts.astype(np.float32))
    so = o.get_state()
    used = plain = bonus = ended = limit = 0
    worst = dict(obs=0.0, rew=0.0, state=0.0)
    for e in range(n):
        ref = np_env.control_step(m, states[e], acts[e].astype(np.float32))
        if ref["advance"]:
            continue                                              # stone re-draw: Philox, pinned bit-exactly elsewhere
        used += 1
        assert bool(done[e]) == ref["done"], (e, done[e], ref)
        assert int(info["bad_transition"][e]) == int(ref["bad"] and ref["done"])
        worst["rew"] = max(worst["rew"], abs(float(rew[e]) - ref["rew"]))
        bonus += ref["count"] == 1 and int(states[e, ol.S_COUNT]) == 0
        limit += ref["bad"]
        if ref["done"]:
            ended += 1
            continue                                              # the returned observation / state are the reset ones
        plain += 1
        assert int(so[e, ol.S_FLAGS]) == ref["flags"] and int(so[e, ol.S_COUNT]) == ref["count"] and int(so[e, ol.S_N]) == ref["n"]
        worst["obs"] = max(worst["obs"], np.abs(obs[e].astype(np.float64) - ref["obs"]).max())
        worst["state"] = max(worst["state"], np.abs(so[e, :55] - ref["state55"]).max())
    print("%s: %d env-steps (%d continuing, %d ended, %d at the time limit, %d with a first-touch bonus): worst |obs| %.1e, |rew| %.1e, "
          "|state| %.1e" % (kind, used, plain, ended, limit, bonus, worst["obs"], worst["rew"], worst["state"]))
    assert used >= 60 and plain >= 30 and ended >= 5 and limit >= 5 and bonus >= 3
    # the oracle returns float32 observations / rewards from its fp64 evaluation: float32 rounding of O(1..30) values
    assert worst["obs"] < 1e-6 and worst["rew"] < 1e-5 and worst["state"] < 1e-10


@pytest.mark.parametrize("kind", KINDS)
def test_create_temp_states_matches_independent_numpy(kind):
    """PHYSICS.md 8: the current observation with the look-ahead stone n+1 re-placed at each of the 121 grid cells (same step
    length, same tilts), rows in grid order i*11+j."""
    import np_terrain as npt
    m = npc.rounded_model(kind)
    o = ol.OracleEnv(kind, 12, seed=4, prec="f64")
    o.set_curriculum(5)
    o.reset()
    for rnd in range(2):                                         # robots set onto their target stone: two advances, drawn stones in play
        st = o.get_state()
        for e in range(12):
            terrain = st[e, ol.S_TERRAIN].reshape(20, 6)
            k = int(st[e, ol.S_N])
            st[e, 0:2] = terrain[k][:2]
            st[e, 2] += terrain[k][2] - terrain[k - 1][2]
        o.set_state(st)
        for t in range(3):
            o.step(np.zeros((12, 21), np.float32))
    st = o.get_state()
    tmp = o.create_temp_states()
    assert tmp.shape == (12, 121, 60)
    worst, moved = 0.0, 0
    for e in range(12):
        n = int(st[e, ol.S_N])
        if n >= 19:
            continue
        moved += n > 1
        terrain = st[e, ol.S_TERRAIN].reshape(20, 6)
        base = np_env.observation(m, st[e], int(st[e, ol.S_FLAGS]), n)
        for i in range(11):
            for j in range(11):
                s2 = st[e].copy()
                t2 = s2[ol.S_TERRAIN].reshape(20, 6)
                phi, dr = terrain[n][3] + npt.YAW[i], st[e, ol.S_NNDR]
                t2[n + 1][:3] = terrain[n][:3] + dr * np.array([np.cos(npt.PITCH[j]) * np.cos(phi), np.cos(npt.PITCH[j]) * np.sin(phi), np.sin(npt.PITCH[j])])
                ref = np_env.observation(m, s2, int(st[e, ol.S_FLAGS]), n)
                assert np.array_equal(ref[:55], base[:55])
                worst = max(worst, np.abs(tmp[e, i * 11 + j].astype(np.float64) - ref).max())
    print("%s: create_temp_states of 12 envs (%d beyond their first target): worst |obs| %.1e" % (kind, moved, worst))
    assert moved >= 3 and worst < 1e-6
