"""The device's dynamics against the INDEPENDENT numpy evaluation, with no C oracle in between (run with `pytest -m gpu`).

tests/test_oracle_contact.py pins the C oracle's contact stage to tests/np_contact.py (dense J H^-1 J^T from recursive
Newton-Euler, impulse-space Gauss-Seidel), tests/test_oracle_dynamics.py pins its ABA to tests/np_dynamics.py, and the GPU
parity tests compare the HIP kernel with the C oracle.  This file closes the triangle: one control step (4 substeps) of the
HIP kernel from injected states against 4 numpy substeps from the same float32 inputs -- nothing of oracle/ is evaluated
between the inputs and the comparison (the oracle only supplies the contact-rich input states).

Tolerance (fp32 kernel against an fp64 evaluation, PHYSICS.md's amplification of rounding on a pivoting foot included, see
DESIGN.md section 3): joint angles / base pose within 2e-6 in the median and 5e-5 at worst (measured 1.6e-7 / 6.6e-6);
generalised velocities (O(1..10) rad/s; the observation scales them by 0.1) within 1e-4 in the median, 1e-3 for 99 % of the
env-steps and 5e-3 at worst (measured 1.7e-5 / 2.3e-4 / 4.0e-4); the contact flags of both feet equal except where a corner is
within 1e-5 of a detection threshold in the numpy evaluation (measured: equal on all)."""
import numpy as np
import pytest

import np_contact as npc
import oracle_lib as ol
from test_oracle_contact import contact_states

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

KINDS = [("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")]


def numpy_control_step(m, st, act):
    """Four substeps of PHYSICS.md 3 in numpy fp64 from a packed state; returns the 55 dynamic words, the per-foot contact flags
    of the last substep and the smallest distance of any corner to a detection threshold over the four substeps."""
    tau = np.clip(act, -1.0, 1.0) * m["torque"]
    s = st.astype(np.float64).copy()
    margin = np.inf
    feet = [False, False]
    for _ in range(4):
        out = npc.substep(m, s, tau)
        margin = min(margin, detection_margin(m, s))
        feet = [any(c is not None and c["foot"] == f for c in out["contacts"]) for f in (0, 1)]
        s[:55] = out["state"]
    return s[:55], feet, margin


def detection_margin(m, st):
    """Distance of the nearest sole corner to a switching surface of the detection (PHYSICS.md 3.3): plane distance 0 or -reach,
    disc radius, two touching stones at the same depth -- evaluated like np_contact.detect does."""
    pos, quat, q = st[0:3], st[3:7], st[13:34]
    terrain = st[65:185].reshape(20, 6)
    n = int(st[59])
    R, p = npc.M.fk(m, q, pos, npc.npd.quat_rot(quat))
    best = np.inf
    for f, body in enumerate(npc.FEET):
        for k in range(4):
            r = m["corners"][k].copy()
            if f == 1:
                r[1] = -r[1]
            P = p[body] + R[body] @ r
            hits = []
            for si in (max(n - 1, 0), n, min(n + 1, 19)):
                nrm = npc.stone_normal(terrain[si])
                d = float((P - terrain[si][:3]) @ nrm)
                rho = float(np.linalg.norm((P - terrain[si][:3]) - d * nrm))
                if rho < npc.STONE_R + 1e-3 and -npc.REACH - 1e-3 < d < 1e-3:
                    best = min(best, abs(d), abs(d + npc.REACH), abs(rho - npc.STONE_R))
                if -npc.REACH < d < 0 and rho < npc.STONE_R:
                    hits.append(d)
            for i in range(len(hits)):
                for j in range(i):
                    if hits[i] != hits[j]:
                        best = min(best, abs(hits[i] - hits[j]))
    return best


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_device_step_matches_independent_numpy(env_id, kind):
    from steppingstone_amd.envs import SteppingStoneVecEnv
    m = npc.rounded_model(kind)
    rng = np.random.default_rng(17)
    states = np.array(contact_states(kind, rng)[:160], np.float64).astype(np.float32)     # what the device will hold
    states[:, ol.S_ELAPSED] = 0                                                              # no time limit in play
    n = states.shape[0]
    acts = rng.uniform(-1.2, 1.2, (n, 21)).astype(np.float32)
    g = SteppingStoneVecEnv(env_id, n, seed=3, device="cuda:0", return_numpy=True)
    g.update_curriculum(5)
    g.reset()
    g.set_state(states)
    _, _, done, _ = g.step(acts)
    sg = g.get_state().cpu().numpy().astype(np.float64)
    g.close()
    e_pose, e_vel, flags_ok, flags_near, used = [], [], 0, 0, 0
    for e in range(n):
        if done[e]:
            continue                                           # the auto-reset replaced the state
        ref, feet, margin = numpy_control_step(m, states[e], acts[e])
        used += 1
        pose = np.r_[0:7, 13:34]
        vel = np.r_[7:13, 34:55]
        e_pose.append(np.abs(sg[e, pose] - ref[pose]).max())
        e_vel.append(np.abs(sg[e, vel] - ref[vel]).max())
        gflags = int(sg[e, ol.S_FLAGS])
        same = gflags == ((1 if feet[0] else 0) | (2 if feet[1] else 0))
        flags_ok += same
        if not same:
            flags_near += margin < 1e-5
            assert margin < 1e-5, "env %d: contact flags %d vs %s with every corner %.1e from a threshold" % (e, gflags, feet, margin)
    e_pose, e_vel = np.array(e_pose), np.array(e_vel)
    print("%s: %d env-steps against numpy (no oracle): pose error median %.1e / 99 %% %.1e / max %.1e, velocity error median %.1e / 99 %% "
          "%.1e / max %.1e; contact flags equal on %d, %d near a threshold" % (kind, used, np.median(e_pose), np.quantile(e_pose, .99), e_pose.max(),
                                                                             np.median(e_vel), np.quantile(e_vel, .99), e_vel.max(), flags_ok, flags_near))
    assert used >= 100
    assert np.median(e_pose) < 2e-6 and e_pose.max() < 5e-5
    assert np.median(e_vel) < 1e-4 and np.quantile(e_vel, 0.99) < 1e-3 and e_vel.max() < 5e-3
