"""A committed stabilising controller for closed-loop tests (own gains, found by a cross-entropy search against the
fp32 CPU oracle; nothing here comes from the reference or its shipped policies).

Joint-space PD to the nominal pose plus torso feedback: ankle-y and hip-y torques on (pitch - lean, forward speed),
hip-x on (roll, lateral speed), abdomen-x on roll.  Inputs are taken from the 60-float observation only
(docs/PHYSICS.md section 5), so every implementation under test computes its actions from its own observations.
It keeps Walker3D and Mike standing on stone 0 for the full 1000-step episode under small action noise."""
import numpy as np

from steppingstone_amd import model

#           kp      kd      ankle_p ankle_v hip_p   hip_v   roll_p  roll_v  abd_r   lean
GAINS = {
    "walker3d": (2.42481105, 0.01104716, 1.08918616, 0.90022247, 1.15000078, 0.36002104, 4.99677791, 0.84591023, 1.5380171, 0.04245738),
    "mike": (2.42626432, 0.00591566469, 1.06855227, 0.908346373, 1.10546228, 0.371718653, 6.20197915, 0.880407986, 1.48045742, 0.0384706459),
}


def balance_controller(kind):
    m = model.build(kind)
    rng = np.asarray(m["range"], np.float64)
    lo, hi, q0 = rng[:, 0], rng[:, 1], np.asarray(m["q0"], np.float64)
    mid, span = 0.5 * (lo + hi), hi - lo
    sigma = np.asarray(model.POLICY_SIGN, np.float64)
    kp, kd, ap, av, hp, hv, kr, kv, ar, lean = GAINS[kind]

    def act(obs):
        o = np.asarray(obs, np.float64)
        q = mid + sigma * o[:, 6:27] * span / 2  # un-normalise (obs = sigma 2 (q - mid) / span, policy coordinates: PHYSICS.md 2)
        qd = sigma * o[:, 27:48] * 10.0          # obs = sigma 0.1 qd
        roll, pitch, vx, vy = o[:, 4], o[:, 5], o[:, 1], o[:, 2]
        a = kp * (q0[None, :] - q) - kd * qd
        for j in (7, 12):                        # ankle y
            a[:, j] += ap * (pitch - lean) + av * vx
        for j in (5, 10):                        # hip y
            a[:, j] += hp * (pitch - lean) + hv * vx
        for j in (3, 8):                         # hip x
            a[:, j] += kr * roll + kv * vy
        a[:, 2] += ar * roll                     # abdomen x
        return np.clip(sigma * a, -1.0, 1.0).astype(np.float32)     # torques about +axis -> policy coordinates

    return act
