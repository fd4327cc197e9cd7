#!/usr/bin/env python3
"""The reference's SHIPPED policies as the one reference-held anchor of the observation / action layout (SURVEY.md 8f-3, 8f-4;
VERDICT r3 item 2).  This container only (the checkpoints live under /root/reference); restricted unpickler, no reference code
executed; CPU oracle backend; informational -- never a parity claim.

The policies were trained in the reference's own (absent) env, so their weights encode ITS conventions.  Three probes:

 1. mirror equivariance: a left/right-symmetric robot's trained policy satisfies  pi(M_o o) ~ M_a pi(o)  and  V(M_o o) ~ V(o)
    for the env's true mirror operators.  Measured for (a) the index lists ss_get_mirror_indices returns for OUR joint conventions,
    (b) the lists as SURVEY section 9 recollects them for upstream mocca_envs (only the abdomen z / x joints negated: the left limbs'
    x / z axes are themselves mirrored there), (c) random lists -- then per-index attribution (toggle one index's negation) and a
    coordinate descent over the sign bits from both starts.  The critic isolates the observation side (no M_a involved).
 2. first layer: column norms of actor.fc1 per observation entry (an input the env never varied has a small column; right / left
    twins have similar ones) and the same for the critic; 2b. the mean Jacobian d pi / d o: which 21-wide windows of the 60 inputs
    are coupled joint-by-joint to the 21 actions (the joint-rate and joint-angle blocks, in action order).
 3. survival of the deterministic shipped policy in OUR env under cheap convention adapters between env and policy (sign of the
    left limbs' x / z joints, per-joint-type signs, velocity / angle / action scale, target-block order and sin / cos order, clipping of the action), greedy over the
    per-joint-type signs.  Mean episode length in control steps; random actions and zero actions for comparison.

  python tools/checkpoint_layout_probe.py [walker3d|mike] > profiles/r04_checkpoint_layout_<robot>.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from steppingstone_amd import _lib  # noqa: E402
from steppingstone_amd.legacy_checkpoint import load_reference_checkpoint  # noqa: E402
import oracle_lib as ol  # noqa: E402

MODELS = "/root/reference/playground/models/"
FILES = {"walker3d": ["mocca_envs:Walker3DStepperEnv-v0_latest.pt", "mocca_envs:Walker3DStepperEnv-v0_best.pt",
                      "mocca_envs:Walker3DStepperEnv-v0_base.pt"],
         "mike": ["mocca_envs:MikeStepperEnv-v0_latest.pt"]}
JOINTS = ["abdomen_z", "abdomen_y", "abdomen_x", "r_hip_x", "r_hip_z", "r_hip_y", "r_knee", "r_ankle", "l_hip_x", "l_hip_z", "l_hip_y",
          "l_knee", "l_ankle", "r_shoulder_x", "r_shoulder_z", "r_shoulder_y", "r_elbow", "l_shoulder_x", "l_shoulder_z", "l_shoulder_y",
          "l_elbow"]                                      # common/render_utils.py:47-69
OBS_NAMES = (["dz", "vx", "vy", "vz", "roll", "pitch"] + ["q:" + j for j in JOINTS] + ["qd:" + j for j in JOINTS] + ["contact_r", "contact_l"] +
             ["t1:sin*d", "t1:cos*d", "t1:dz", "t1:xtilt", "t1:ytilt", "t2:sin*d", "t2:cos*d", "t2:dz", "t2:xtilt", "t2:ytilt"])
RIGHT_J = [3, 4, 5, 6, 7, 13, 14, 15, 16]
LEFT_J = [8, 9, 10, 11, 12, 17, 18, 19, 20]


class Mirror:
    """Signed permutation of observation and action: negate the listed entries, then swap right <-> left."""

    def __init__(self, neg_o, neg_a):
        self.so = np.ones(60, np.float32)
        self.so[list(neg_o)] = -1
        self.sa = np.ones(21, np.float32)
        self.sa[list(neg_a)] = -1
        self.po = np.arange(60)
        for r, l in zip(RIGHT_J, LEFT_J):
            for base in (6, 27):
                self.po[base + r], self.po[base + l] = base + l, base + r
        self.po[48], self.po[49] = 49, 48
        self.pa = np.arange(21)
        for r, l in zip(RIGHT_J, LEFT_J):
            self.pa[r], self.pa[l] = l, r

    def obs(self, o):
        return (o * self.so)[:, self.po]

    def act(self, a):
        return (a * self.sa)[:, self.pa]


def ours():
    idx = _lib.mirror_indices()
    return Mirror(set(int(i) for i in idx[0]), set(int(i) for i in idx[3]))


def recollected():
    return Mirror({2, 4, 6, 8, 27, 29, 50, 53, 55, 58}, {0, 2})


def round3():
    """the lists of rounds 1-3: every x / z joint negated (left limbs measured about the +axis like the right ones)"""
    xz = [0, 2, 3, 4, 8, 9, 13, 14, 17, 18]
    return Mirror({2, 4, 50, 53, 55, 58} | {6 + j for j in xz} | {27 + j for j in xz}, set(xz))


def random_mirror(rng):
    return Mirror(set(np.nonzero(rng.random(60) < 0.4)[0].tolist()), set(np.nonzero(rng.random(21) < 0.4)[0].tolist()))


def errors(ac, M, O):
    with torch.no_grad():
        o = torch.from_numpy(O)
        om = torch.from_numpy(np.ascontiguousarray(M.obs(O)))
        a, am = ac.actor(o).numpy(), ac.actor(om).numpy()
        v, vm = ac.get_value(o).numpy(), ac.get_value(om).numpy()
    return float(np.abs(am - M.act(a)).mean()), float(np.abs(vm - v).mean() / (v.std() + 1e-9)), float(np.abs(a).mean())


def describe(M):
    return ("obs negated: %s | act negated: %s" % ([OBS_NAMES[i] for i in np.nonzero(M.so < 0)[0]], [JOINTS[i] for i in np.nonzero(M.sa < 0)[0]]))


def descend(ac, M, O, label):
    """coordinate descent over the sign bits (observation, then action) on the ACTOR's equivariance error -- the critic is reported
    but not optimised: it is far less symmetric than the actor in every shipped file; swapped pairs keep a common bit (a sign on one
    side only is not an involution)."""
    pair_of = {}
    for r, l in zip(RIGHT_J, LEFT_J):
        for base in (6, 27):
            pair_of[base + r] = base + l
            pair_of[base + l] = base + r
    pair_of[48], pair_of[49] = 49, 48
    ea, ev, _ = errors(ac, M, O)
    for sweep in range(3):
        changed = False
        for i in range(60):
            if i in pair_of and pair_of[i] < i:
                continue
            M.so[i] = -M.so[i]
            if i in pair_of:
                M.so[pair_of[i]] = M.so[i]
            ea2, ev2, _ = errors(ac, M, O)
            if ea2 < ea - 1e-4:
                ea, ev, changed = ea2, ev2, True
            else:
                M.so[i] = -M.so[i]
                if i in pair_of:
                    M.so[pair_of[i]] = M.so[i]
        apair = dict(zip(RIGHT_J, LEFT_J))
        for j in range(21):
            if j in LEFT_J:
                continue
            M.sa[j] = -M.sa[j]
            if j in apair:
                M.sa[apair[j]] = M.sa[j]
            ea2, ev2, _ = errors(ac, M, O)
            if ea2 < ea - 1e-4:
                ea, ev, changed = ea2, ev2, True
            else:
                M.sa[j] = -M.sa[j]
                if j in apair:
                    M.sa[apair[j]] = M.sa[j]
        if not changed:
            break
    print("   descent from %-12s -> actor %.3f critic %.3f | %s" % (label, ea, ev, describe(M)))
    return M


def attribution(ac, M, O):
    ea0, ev0, _ = errors(ac, M, O)
    rows = []
    for i in range(60):
        M.so[i] = -M.so[i]
        ea, ev, _ = errors(ac, M, O)
        M.so[i] = -M.so[i]
        rows.append((ev - ev0, ea - ea0, "obs " + OBS_NAMES[i]))
    for j in range(21):
        M.sa[j] = -M.sa[j]
        ea, ev, _ = errors(ac, M, O)
        M.sa[j] = -M.sa[j]
        rows.append((0.0, ea - ea0, "act " + JOINTS[j]))
    better = [r for r in rows if r[1] < -1e-3]
    print("   of the 81 sign bits, toggling one LOWERS the actor's error for: %s" % (
        ["%s (%+.3f)" % (n, da) for dv, da, n in sorted(better, key=lambda r: r[1])] or "none -- every single bit is confirmed by the actor"))
    print("      smallest increase when a bit is toggled: %+.4f (%s); toggles that lower the critic's error by > 0.05: %s" % (
        min(r[1] for r in rows), min(rows, key=lambda r: r[1])[2], [n for dv, da, n in rows if dv < -0.05] or "none"))


def sample_observations(kind, ac, n=256, steps=24):
    """observations of OUR env: half under random actions, half under the shipped policy itself."""
    out = []
    for policy in (False, True):
        o = ol.OracleEnv(kind, n, seed=5)
        o.set_curriculum(3)
        obs = o.reset()
        for t in range(steps):
            if policy:
                with torch.no_grad():
                    a = ac.actor(torch.from_numpy(obs)).numpy()
            else:
                a = o.random_actions(t)
            obs, _, _, _ = o.step(a.astype(np.float32))
            if t % 2 == 1:
                out.append(obs.copy())
        o.close()
    return np.concatenate(out).astype(np.float32)


def jacobian_blocks(ac, O):
    """mean Jacobian d pi / d o over the observations: a trained controller couples action j most strongly to ITS joint's rate and
    angle (damping / restoring terms), so the two 21-wide windows of the 60 inputs whose 21 x 21 block is diagonal-dominant are the
    joint-rate and joint-angle blocks, in ACTION order -- a reference-held check of the layout hypothesis 6 | 21 q | 21 qd | 2 | 10."""
    Ot = torch.from_numpy(O).clone().requires_grad_(True)
    J = np.zeros((21, 60))
    for j in range(21):
        g, = torch.autograd.grad(ac.actor(Ot)[:, j].sum(), Ot)
        J[j] = g.mean(0).numpy()
    ratio = [(float(np.abs(np.diag(J[:, off:off + 21])).mean() / (np.abs(J[:, off:off + 21]).mean() + 1e-12)), off) for off in range(0, 40)]
    return J, sorted(ratio, reverse=True)


def survival(kind, ac, so, sa, n=96, steps=240, act_fn=None, clip=False, qd_scale=1.0, swap_target=False, act_scale=1.0, swap_blocks=False,
             q_scale=1.0):
    o = ol.OracleEnv(kind, n, seed=9)
    o.set_curriculum(0)
    obs = o.reset()
    lens, reached = [], []
    for t in range(steps):
        if act_fn is not None:
            a = act_fn(o, t)
        else:
            x = obs * so
            if qd_scale != 1.0:
                x[:, 27:48] *= qd_scale
            if q_scale != 1.0:
                x[:, 6:27] *= q_scale
            if swap_target:
                x[:, [50, 51, 55, 56]] = x[:, [51, 50, 56, 55]]
            if swap_blocks:
                x[:, 50:60] = x[:, [55, 56, 57, 58, 59, 50, 51, 52, 53, 54]]
            with torch.no_grad():
                a = ac.actor(torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))).numpy() * sa * act_scale
            if clip:
                a = np.clip(a, -1, 1)
        obs, _, d, info = o.step(a.astype(np.float32))
        for i in np.nonzero(d)[0]:
            lens.append(float(info["ep_len"][i]))
            reached.append(int(info["steps_reached"][i]))
    # episodes still running count with their current length (censored)
    st = o.get_state()
    lens += st[:, ol.S_ELAPSED].tolist()
    o.close()
    return float(np.mean(lens)), float(np.mean(reached)) if reached else 0.0


def type_sign_vectors(flip_types, left_xz):
    """sign vectors (obs 60, act 21) for: joint TYPES whose sign convention is flipped on both sides (abdomen_z ... elbow, 12 types),
    and optionally the left limbs' x / z joints flipped (upstream's mirrored left axes)."""
    types = {"abdomen_z": [0], "abdomen_y": [1], "abdomen_x": [2], "hip_x": [3, 8], "hip_z": [4, 9], "hip_y": [5, 10], "knee": [6, 11],
             "ankle": [7, 12], "shoulder_x": [13, 17], "shoulder_z": [14, 18], "shoulder_y": [15, 19], "elbow": [16, 20]}
    sj = np.ones(21, np.float32)
    for t in flip_types:
        sj[types[t]] *= -1
    if left_xz:
        sj[[8, 9, 17, 18]] *= -1
    so = np.ones(60, np.float32)
    so[6:27] = sj
    so[27:48] = sj
    return so, sj, list(types)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "walker3d"
    rng = np.random.default_rng(0)
    for fname in FILES[kind]:
        ac = load_reference_checkpoint(MODELS + fname)
        print("=" * 100)
        print("%s  (actor %d -> %d, %d critic(s))" % (fname, ac.actor.fc1.weight.shape[1], ac.logstd.numel(), len(ac.critics)))
        O = sample_observations(kind, ac)
        print("1. mirror equivariance on %d observations of our env (actor: mean |pi(M_o o) - M_a pi(o)|, mean |a| = %.3f; critic: mean "
              "|V(M_o o) - V(o)| / std V)" % (len(O), errors(ac, ours(), O)[2]))
        for label, M in (("ours", ours()), ("recollected", recollected()), ("round 3", round3())):
            ea, ev, _ = errors(ac, M, O)
            print("   %-12s actor %.3f critic %.3f | %s" % (label, ea, ev, describe(M)))
        r = np.array([errors(ac, random_mirror(rng), O)[:2] for _ in range(20)])
        print("   %-12s actor %.3f critic %.3f (mean of 20 random sign sets, same right/left pairing)" % ("random", r[:, 0].mean(), r[:, 1].mean()))
        ident = Mirror(set(), set())
        ident.po, ident.pa = np.arange(60), np.arange(21)
        print("   %-12s actor %.3f critic %.3f (no mirror at all: M = identity)" % (("identity",) + errors(ac, ident, O)[:2]))
        print("   attribution from ours:")
        attribution(ac, ours(), O)
        print("   attribution from recollected:")
        attribution(ac, recollected(), O)
        descend(ac, ours(), O, "ours")
        descend(ac, recollected(), O, "recollected")
        print("2. first layer: column norm of fc1 per observation entry (actor | critic 0)")
        wa = ac.actor.fc1.weight.detach().numpy()
        wc = ac.critics[0][0].weight.detach().numpy()
        na, nc = np.linalg.norm(wa, axis=0), np.linalg.norm(wc, axis=0)
        for i in range(0, 60, 6):
            print("   " + "  ".join("%-16s %.2f|%.2f" % (OBS_NAMES[k][:16], na[k], nc[k]) for k in range(i, min(i + 6, 60))))
        tw = [abs(na[6 + r] - na[6 + l]) / (na[6 + r] + na[6 + l]) for r, l in zip(RIGHT_J, LEFT_J)]
        print("   right / left twin columns (joint angles): relative norm difference mean %.3f max %.3f; contact flags %.2f vs %.2f" % (
            np.mean(tw), np.max(tw), na[48], na[49]))
        J, ratio = jacobian_blocks(ac, O)
        print("2b. mean Jacobian d pi / d o: 21-wide input windows ranked by diagonal dominance of their 21 x 21 block (mean |diag| / mean |block|): "
              + ", ".join("offset %d: %.2f" % (o, r) for r, o in ratio[:5]))
        for name, off in (("joint rates, obs[27:48]", 27), ("joint angles, obs[6:27]", 6)):
            B = J[:, off:off + 21]
            print("   d a_j / d (%s)_j: %s   negative for %d of 21; row maximum on the own joint for %d of 21" % (
                name, " ".join("%+.2f" % v for v in np.diag(B)), int((np.diag(B) < 0).sum()), int((np.abs(B).argmax(1) == np.arange(21)).sum())))
        print("3. survival of the deterministic policy in OUR env (mean episode length in control steps, 96 envs x 240 steps, flat terrain;"
              " stones reached)")
        one = np.ones(60, np.float32), np.ones(21, np.float32)
        print("   random actions            %6.1f  %.2f" % survival(kind, ac, *one, act_fn=lambda o, t: o.random_actions(t)))
        print("   zero actions              %6.1f  %.2f" % survival(kind, ac, *one, act_fn=lambda o, t: np.zeros((o.n, 21), np.float32)))
        print("   policy, our conventions   %6.1f  %.2f" % survival(kind, ac, *one))
        print("   ... action clipped to +-1 %6.1f  %.2f" % survival(kind, ac, *one, clip=True))
        so, sa, types = type_sign_vectors([], True)
        print("   left x / z joints flipped %6.1f  %.2f   (the round-3 convention: left limbs about the +axis)" % survival(kind, ac, so, sa))
        for s in (0.3, 3.0, 10.0):
            print("   joint speeds x %-4g       %6.1f  %.2f" % ((s,) + survival(kind, ac, *one, qd_scale=s)))
        print("   target sin/cos swapped    %6.1f  %.2f" % survival(kind, ac, *one, swap_target=True))
        print("   target blocks swapped     %6.1f  %.2f" % survival(kind, ac, *one, swap_blocks=True))
        for s_ in (0.5, 2.0):
            print("   action x %-4g (power)     %6.1f  %.2f" % ((s_,) + survival(kind, ac, *one, act_scale=s_)))
        for s_ in (0.5, 2.0):
            print("   joint angles x %-4g       %6.1f  %.2f   (another range normalisation)" % ((s_,) + survival(kind, ac, *one, q_scale=s_)))
        feats = []
        for grp in ([0], [1], [2], [3], [4], [5], [50, 55], [51, 56], [52, 57], [53, 58], [54, 59], [48, 49]):
            so_ = np.ones(60, np.float32)
            so_[grp] = -1
            feats.append("%s %.1f" % ("+".join(OBS_NAMES[g] for g in grp), survival(kind, ac, so_, one[1])[0]))
        print("   one non-joint observation feature negated between env and policy (both targets together): " + "  ".join(feats))
        so0, sa0, types0 = type_sign_vectors([], False)
        print("   one joint type flipped (env <-> policy sign of angle, rate and action; both sides): " + "  ".join(
            "%s %.1f" % (t, survival(kind, ac, *type_sign_vectors([t], False)[:2])[0]) for t in types0))
        for left in (False, True):
            flips, best = [], survival(kind, ac, *type_sign_vectors([], left)[:2])[0]
            for sweep in range(2):
                for t in types:
                    trial = [x for x in flips if x != t] if t in flips else flips + [t]
                    s = survival(kind, ac, *type_sign_vectors(trial, left)[:2])[0]
                    if s > best + 1.0:
                        best, flips = s, trial
            print("   greedy per-joint-type signs (left x/z flipped: %s): best %.1f steps with %s flipped" % (left, best, flips or "nothing"))


if __name__ == "__main__":
    main()
