"""Pins the Philox-driven parts of the CPU oracle (docs/PHYSICS.md 6 and 7: reset pose, stone draw on a target advance for the
curriculum window, the specialist ring and a custom grid) against the independent numpy evaluation of tests/np_terrain.py,
whose Philox4x32-10 is checked against the Random123 known answers here as well.  CPU only."""
import numpy as np
import pytest

import np_contact as npc
import np_terrain as npt
import oracle_lib as ol

KINDS = ["walker3d", "mike"]


def test_numpy_philox_known_answers():
    f = 0xFFFFFFFF
    assert [hex(x) for x in npt.philox4x32_10([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in npt.philox4x32_10([f, f, f, f], [f, f])] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    out = npt.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])
    assert [hex(x) for x in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


@pytest.mark.parametrize("kind", KINDS)
def test_reset_matches_independent_numpy(kind):
    m = npc.rounded_model(kind)
    seed, off, n = 0x1234567890, 7, 5
    o = ol.OracleEnv(kind, n, seed=seed, env_offset=off, prec="f64")
    o.reset()
    st = o.get_state()
    for e in range(n):
        assert int(st[e, ol.S_CTRLO]) + (int(st[e, ol.S_CTRHI]) << 16) == 6                 # six blocks consumed
        q = npt.reset_joint_angles(m, seed, 0, off + e)
        assert np.abs(st[e, ol.S_Q] - q).max() < 1e-7
        assert np.abs(st[e, ol.S_QD]).max() == 0 and np.abs(st[e, ol.S_VEL]).max() == 0
        assert np.allclose(st[e, ol.S_QUAT], [1, 0, 0, 0]) and st[e, 0] == 0 and st[e, 1] == 0
        assert int(st[e, ol.S_N]) == 1 and int(st[e, ol.S_COUNT]) == 0 and int(st[e, ol.S_ELAPSED]) == 0
        terrain = st[e, ol.S_TERRAIN].reshape(20, 6)
        assert np.allclose(terrain[:, 0], 0.75 * np.arange(20)) and np.abs(terrain[:, 1:]).max() == 0


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("mode", ["curriculum3", "curriculum5", "specialist4", "custom"])
def test_stone_draw_matches_independent_numpy(kind, mode):
    """Robots standing on their target stone for two steps: the target advances and stone n+1 is drawn."""
    seed, off, n = 99, 3, 24
    o = ol.OracleEnv(kind, n, seed=seed, env_offset=off, prec="f64")
    if mode.startswith("curriculum"):
        level = int(mode[-1]); o.set_curriculum(level); prob = npt.window_grid(level)
    elif mode == "specialist4":
        level = 4; o.set_specialist(4); prob = npt.window_grid(4, ring=True)
    else:
        level = 5; o.set_curriculum(5)
        prob = np.random.default_rng(0).random((11, 11)) ** 6
        prob = prob / prob.sum()
        o.set_sample_prob(prob)
        prob = prob.astype(np.float32)
    o.reset()
    drawn = 0
    for rnd in range(3):                                           # three advances per env: stones 3, 4, 5 from their predecessors
        st = o.get_state()
        k = st[:, ol.S_N].astype(int)
        for e in range(n):
            terrain = st[e, ol.S_TERRAIN].reshape(20, 6)
            st[e, 0:2] = terrain[k[e]][:2]                         # onto the target
            st[e, 2] += terrain[k[e]][2] - (terrain[k[e] - 1][2] if rnd else 0.0)
            st[e, ol.S_POT] = 0.0
        o.set_state(st)
        before = o.get_state()
        adv = np.zeros(n, bool)
        for t in range(4):
            _, _, done, info = o.step(np.zeros((n, 21), np.float32))
            adv |= np.asarray(info["update_terrain"]).astype(bool)
            if adv.all():
                break
        after = o.get_state()
        for e in range(n):
            if not adv[e] or int(after[e, ol.S_N]) != k[e] + 1:
                continue
            ctr = int(before[e, ol.S_CTRLO]) + (int(before[e, ol.S_CTRHI]) << 16)
            u = npt.uniforms(seed, ctr, 0, off + e)
            tb, ta = before[e, ol.S_TERRAIN].reshape(20, 6), after[e, ol.S_TERRAIN].reshape(20, 6)
            new = k[e] + 2
            if new > 19:
                continue
            ref, cell = npt.draw_stone(ta[new - 1], prob, level, u)
            assert np.abs(ta[new] - ref).max() < 1e-6, (mode, e, new, ta[new], ref, cell)
            assert int(after[e, ol.S_CTRLO]) + (int(after[e, ol.S_CTRHI]) << 16) == ctr + 1
            assert prob.reshape(-1)[cell[0] * 11 + cell[1]] > 0
            drawn += 1
    print("%s %s: %d drawn stones equal to the numpy draw" % (kind, mode, drawn))
    assert drawn >= 30
