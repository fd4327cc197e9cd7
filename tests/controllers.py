"""A committed stabilising controller for closed-loop tests (own gains, found by a cross-entropy search against the
fp32 CPU oracle -- tools/tune_balance_controller.py, re-run in round 5 for the identified robots; nothing here comes from the
reference or its shipped policies).

Joint-space PD to the nominal pose plus torso feedback: ankle-y and hip-y torques on (pitch - lean, forward speed),
hip-x on (roll, lateral speed), abdomen-x on roll.  Inputs are taken from the 60-float observation only
(docs/PHYSICS.md section 5), so every implementation under test computes its actions from its own observations.
It keeps Walker3D and Mike standing on stone 0 for the full 1000-step episode under small action noise."""
import numpy as np

from steppingstone_amd import model

#           kp      kd      ankle_p ankle_v hip_p   hip_v   roll_p  roll_v  abd_r   lean    [x_pos  y_pos]
ARM_GAIN = 0.04
GAINS = {
    "walker3d": (1.4132094, 0.050730464, 0.14066191, 0.48762243, 0.40773871, 0.076366938, 0.22644276, 0.89381611, 0.43892566, 0.025935326, 5.950433, 0.80725568),
    "mike": (3.1731444, 0.0089635373, 0.72874725, 0.77964025, 2.3102897, 0.36061148, 6.5243057, 0.094869488, 2.9437466, 0.084345045),
}


def balanced_pose(m, knee_deg=22.0):
    """A statically balanced standing pose of model m: knees bent by knee_deg, ankles keeping the soles level under an upright torso, and
    the hip pitch chosen (bisection) so that the whole body's centre of mass sits over the middle of the soles.  The reset pose q0 of
    the round-5 robot is a deep crouch with tilted soles (identified against the shipped policies, which start walking at once); a PD
    controller that is to STAND for 1000 steps needs a pose it can hold."""
    q = np.array(m["q0"], np.float64)
    c = np.asarray(m["corners"], np.float64)
    sole_x = 0.5 * (c[:, 0].max() + c[:, 0].min())
    lo_, hi_ = np.asarray(m["range"], np.float64)[:, 0], np.asarray(m["range"], np.float64)[:, 1]

    def com_minus_sole(h):
        for j0 in (3, 8):
            q[j0 + 2], q[j0 + 3], q[j0 + 4] = h, np.deg2rad(knee_deg), -(h + np.deg2rad(knee_deg))
        R, p = model.fk(m, q)
        com = sum(m["mass"][b] * (p[b] + R[b] @ m["com"][b]) for b in range(model.NB)) / m["mass"].sum()
        foot = p[model.RIGHT_FOOT_BODY] + R[model.RIGHT_FOOT_BODY] @ np.array([sole_x, 0.0, 0.0])
        return com[0] - foot[0]
    a, b = np.deg2rad(-60.0), np.deg2rad(10.0)
    fa = com_minus_sole(a)
    for _ in range(50):
        mid = 0.5 * (a + b)
        fm = com_minus_sole(mid)
        if (fm > 0) == (fa > 0):
            a, fa = mid, fm
        else:
            b = mid
    com_minus_sole(0.5 * (a + b))
    return np.clip(q, lo_ + 0.03, hi_ - 0.03)


def standing_state(kind, st):
    """The packed states `st` (oracle_lib layout) with every robot put into the balanced standing pose, soles 5 mm above stone 0, at
    rest: where the closed-loop tests start from (the reset pose of the round-5 robot is a walker's starting crouch that tips forward
    by design; a STANDING controller starts standing)."""
    m = model.build(kind)
    qb = balanced_pose(m)
    R, p = model.fk(m, qb)
    h = float(-min((p[b] + R[b] @ c)[2] for b in (model.RIGHT_FOOT_BODY, model.LEFT_FOOT_BODY) for c in m["corners"]))
    st = np.array(st, copy=True)
    st[:, 13:34] = qb[None, :]
    st[:, 34:55] = 0.0
    st[:, 7:13] = 0.0
    st[:, 0:2] = 0.0
    st[:, 2] = h + 0.005
    st[:, 3:7] = (1.0, 0.0, 0.0, 0.0)
    st[:, 56] = st[:, 2]          # z_init: the observation's height term starts at zero
    return st


def balance_controller(kind):
    m = model.build(kind)
    rng = np.asarray(m["range"], np.float64)
    lo, hi, q0 = rng[:, 0], rng[:, 1], balanced_pose(m)
    mid, span = 0.5 * (lo + hi), hi - lo
    sigma = np.asarray(model.POLICY_SIGN, np.float64)
    kp, kd, ap, av, hp, hv, kr, kv, ar, lean = GAINS[kind][:10]
    kx, ky = (GAINS[kind][10], GAINS[kind][11]) if len(GAINS[kind]) > 10 else (0.0, 0.0)

    def act(obs):
        o = np.asarray(obs, np.float64)
        q = mid + sigma * o[:, 6:27] * span / 2  # un-normalise (obs = sigma 2 (q - mid) / span, policy coordinates: PHYSICS.md 2)
        qd = sigma * o[:, 27:48] * 10.0          # obs = sigma 0.1 qd
        roll, pitch, vx, vy = o[:, 4], o[:, 5], o[:, 1], o[:, 2]
        # where the torso is over stone 0, from the first target block (stone 1 is 0.75 m straight ahead of stone 0): without it the
        # lightly damped round-5 robot creeps forward for a few hundred steps and tips over its toes
        ex, ey = 0.75 - o[:, 51], -o[:, 50]
        a = kp * (q0[None, :] - q) - kd * qd
        # the arms only need to hang still: a leg-sized position gain on the shoulders' axial (z) joints -- the arm's own axis, a few
        # g m^2 of inertia -- is beyond what a 60 Hz loop can hold (it spins them at 30+ rad/s and makes the closed loop chaotic)
        a[:, 13:21] *= ARM_GAIN
        for j in (7, 12):                        # ankle y
            a[:, j] += ap * (pitch - lean) + av * vx + kx * ex
        for j in (5, 10):                        # hip y
            a[:, j] += hp * (pitch - lean) + hv * vx
        for j in (3, 8):                         # hip x
            a[:, j] += kr * roll + kv * vy + ky * ey
        a[:, 2] += ar * roll                     # abdomen x
        return np.clip(sigma * a, -1.0, 1.0).astype(np.float32)     # torques about +axis -> policy coordinates

    return act
