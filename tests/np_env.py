"""Independent fp64 numpy evaluation of ONE CONTROL STEP's outputs (docs/PHYSICS.md 4 and 5): four substeps through
tests/np_contact.py (dense joint-space dynamics, impulse-space Gauss-Seidel), then the env logic -- contact flags, target
logic, progress, bonuses, termination, reward, the 60-float observation -- written from the specification's text with
formulas of its own (roll / pitch / yaw from the rotation MATRIX, the target block through atan2 / sin / cos as the text
states it; the oracle and the kernel use quaternion algebra and a rotated planar offset).  TEST INFRASTRUCTURE ONLY: it pins
the C oracle's env logic on the CPU (tests/test_oracle_env_numpy.py) and is compared with the device directly
(tests/test_gpu_numpy_direct.py).

Not covered (the caller skips such env-steps): a step on which the target advances (the next stone is re-drawn by the
sampler, PHYSICS.md 6) and the reset observation of a finished episode (PHYSICS.md 7) -- both are Philox-driven and pinned
bit-exactly elsewhere (tests/test_gpu_parity.py::test_target_advance_and_sampler_are_bit_exact)."""
import numpy as np

import np_contact as npc
import np_dynamics as npd
from steppingstone_amd import model as M

DT = 1.0 / 60.0
NUM_STONES = 20
MAX_EPISODE_STEPS = 1000
# packed state words (include/steppingstone.h, ss_get_state)
POS, QUAT, VEL, Q, QD = slice(0, 3), slice(3, 7), slice(7, 13), slice(13, 34), slice(34, 55)
POT, ZINIT, N, COUNT, ELAPSED = 55, 56, 59, 60, 61


def euler_from_matrix(R):
    """R = Rz(yaw) Ry(pitch) Rx(roll)."""
    pitch = -np.arcsin(np.clip(R[2, 0], -1.0, 1.0))
    roll = np.arctan2(R[2, 1], R[2, 2])
    yaw = np.arctan2(R[1, 0], R[0, 0])
    return roll, pitch, yaw


def sole_centres(m, st):
    """World positions of the two sole centres (mean of the four corners), right then left."""
    R, p = M.fk(m, st[Q], st[POS], npd.quat_rot(st[QUAT]))
    out = []
    for f, b in enumerate(npc.FEET):
        c = []
        for k in range(4):
            r = m["corners"][k].copy()
            if f == 1:
                r[1] = -r[1]
            c.append(p[b] + R[b] @ r)
        out.append(np.mean(c, axis=0))
    return out


def observation(m, st, flags, n):
    """PHYSICS.md 5 from a packed state, the contact flags (bit 0 right, bit 1 left) and the target index."""
    R = npd.quat_rot(st[QUAT])
    roll, pitch, yaw = euler_from_matrix(R)
    vw = R @ st[VEL][3:]
    c, s = np.cos(-yaw), np.sin(-yaw)
    obs = np.zeros(60)
    obs[0] = st[POS][2] - st[ZINIT]
    obs[1] = c * vw[0] - s * vw[1]
    obs[2] = s * vw[0] + c * vw[1]
    obs[3] = vw[2]
    obs[4], obs[5] = roll, pitch
    lo, hi = m["range"][:, 0], m["range"][:, 1]
    sigma = np.asarray(M.POLICY_SIGN, np.float64)         # policy coordinates, PHYSICS.md 2
    obs[6:27] = sigma * (2.0 * (st[Q] - 0.5 * (lo + hi)) / (hi - lo))
    obs[27:48] = sigma * (0.1 * st[QD])
    obs[48], obs[49] = float(flags & 1), float((flags >> 1) & 1)
    obs[:50] = np.clip(obs[:50], -5.0, 5.0)
    terrain = st[65:185].reshape(NUM_STONES, 6)
    for i, k in enumerate((n, min(n + 1, NUM_STONES - 1))):
        d = terrain[k][:3] - st[POS]
        dist = np.hypot(d[0], d[1])
        dth = np.arctan2(d[1], d[0]) - yaw
        obs[50 + 5 * i: 55 + 5 * i] = [np.sin(dth) * dist, np.cos(dth) * dist, d[2], terrain[k][4], terrain[k][5]]
    return obs


def control_step(m, st, act):
    """One step(action) without the auto-reset: dict(state55, obs, rew, done, bad, flags, count, advance, n).
    `advance` = the target would advance on this step (the caller skips the env: stone re-draw not restated here)."""
    st = np.asarray(st, np.float64).copy()
    a = np.clip(np.asarray(act, np.float64), -1.0, 1.0)
    tau = np.asarray(M.POLICY_SIGN, np.float64) * a * m["torque"]      # the action is in policy coordinates, PHYSICS.md 2
    n, count, elapsed = int(st[N]), int(st[COUNT]), int(st[ELAPSED])
    terrain = st[65:185].reshape(NUM_STONES, 6)
    contacts, soles, warm = None, None, None              # every control step starts its contact solve cold (PHYSICS.md 3.4)
    for _ in range(4):
        soles = sole_centres(m, st)                       # positions at the start of the substep, like the detector
        out = npc.substep(m, st, tau, warm=warm)
        warm = out["warm"]
        contacts = out["contacts"]
        st[:55] = out["state"]
    flags = 0
    on_target = False
    for c in contacts:
        if c is not None:
            flags |= 1 << c["foot"]
            on_target |= bool(c["on_target"])          # a corner carried by stone n (np_contact.detect)
    elapsed += 1
    pos = st[POS]
    step_bonus, advance = 0.0, False
    if on_target:
        count += 1
        if count == 1:
            rho = min(np.hypot(*(s[:2] - terrain[n][:2])) for s in soles)
            step_bonus = 50.0 * np.exp(-rho / 0.25)
        if count >= 2 and n < NUM_STONES - 1:
            advance = True
    pot = -np.hypot(*(pos[:2] - terrain[n][:2])) / DT
    progress = pot - st[POT]
    target_bonus = 2.0 if (n == NUM_STONES - 1 and np.hypot(*(pos[:2] - terrain[n][:2])) < 0.15) else 0.0
    height = pos[2] - min(s[2] for s in soles)
    tall_bonus = 2.0 if height > 0.7 else -1.0
    zlow = min(terrain[max(n - 1, 0)][2], terrain[n][2], terrain[min(n + 1, NUM_STONES - 1)][2])
    finite = bool(np.isfinite(st[:55]).all())
    timeout = elapsed >= MAX_EPISODE_STEPS
    done = (tall_bonus < 0) or (pos[2] < zlow + 0.3) or (not finite) or timeout
    roll, pitch, _ = euler_from_matrix(npd.quat_rot(st[QUAT]))
    posture = (abs(pitch) if not (-0.2 < pitch < 0.4) else 0.0) + (abs(roll) if not (-0.4 < roll < 0.4) else 0.0)
    energy = (4.5 / 21) * np.mean(np.abs(a * 0.1 * st[QD])) + (0.225 / 21) * np.mean(a * a)
    lo, hi = m["range"][:, 0], m["range"][:, 1]
    qn = 2.0 * (st[Q] - 0.5 * (lo + hi)) / (hi - lo)
    penalty = 0.1 * int((np.abs(qn) > 0.99).sum())
    rew = progress + step_bonus + target_bonus + tall_bonus - energy - posture - penalty
    if not finite or not np.isfinite(rew):
        rew = 0.0
    st[COUNT], st[ELAPSED] = count, elapsed
    return dict(state55=st[:55].copy(), obs=observation(m, st, flags, n), rew=float(rew), done=bool(done), bad=bool(timeout),
                flags=flags, count=count, advance=advance, n=n,
                # distances of the step's switching quantities from their thresholds (for callers that compare with fp32 code)
                margins=dict(height=abs(height - 0.7), low=abs(pos[2] - zlow - 0.3), pitch=min(abs(pitch + 0.2), abs(pitch - 0.4)),
                             roll=min(abs(roll + 0.4), abs(roll - 0.4)), qn=float(np.min(np.abs(np.abs(qn) - 0.99)))))
