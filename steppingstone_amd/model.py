"""Robot model definitions (Walker3D, Mike) for the stepping-stone environments.

The reference's robot assets live in the un-vendored `mocca_envs` submodule
(/root/reference/.gitmodules:1-3) and are NOT available; the only facts the
reference pins are the 21-joint action order (common/render_utils.py:47-69) and
the obs/action dims 60/21 (shipped checkpoints).  Everything
numeric below (link geometry, masses, ranges, gains) is THIS repository's own
specification, documented in docs/PHYSICS.md §2.

The model is a floating base (torso, body 0) plus 21 single-DoF revolute links.
Multi-DoF anatomical joints (hip x/z/y, shoulder x/z/y, abdomen z/y) are chains of
co-located revolute joints with massless intermediate links.  All link frames are
axis-aligned with the torso frame at q = 0 (x forward, y left, z up); each joint
rotates about one coordinate axis of its own link frame.

`build(kind)` returns a dict of numpy arrays; `tools/gen_model_tables.py` turns it
into the constexpr tables compiled into the HIP kernels and the plain-C tables
used by the CPU oracle (data only — no algorithm is shared).
"""
import numpy as np

NJ = 21          # actuated joints == non-root links
NB = NJ + 1      # bodies including the torso (body 0)
DENSITY = 1000.0

JOINT_NAMES = [
    "abdomen_z", "abdomen_y", "abdomen_x",
    "right_hip_x", "right_hip_z", "right_hip_y", "right_knee", "right_ankle",
    "left_hip_x", "left_hip_z", "left_hip_y", "left_knee", "left_ankle",
    "right_shoulder_x", "right_shoulder_z", "right_shoulder_y", "right_elbow",
    "left_shoulder_x", "left_shoulder_z", "left_shoulder_y", "left_elbow",
]

# parent BODY index of joint j's child link (child link of joint j is body j+1)
PARENT = [0, 1, 2,
          3, 4, 5, 6, 7,
          3, 9, 10, 11, 12,
          0, 14, 15, 16,
          0, 18, 19, 20]
# rotation axis of joint j in its own link frame: 0=x 1=y 2=z
AXIS = [2, 1, 0,
        0, 2, 1, 1, 1,
        0, 2, 1, 1, 1,
        0, 2, 1, 1,
        0, 2, 1, 1]
RIGHT_FOOT_BODY = 8    # child of right_ankle (joint 7)
LEFT_FOOT_BODY = 13    # child of left_ankle  (joint 12)

# POLICY-FACING joint coordinates (docs/PHYSICS.md section 2): what the action and the observation carry is
# sigma_j * (torque, angle, rate about the +axis of the link frame).  The dynamics (state, oracle, kernels) keep angles about
# the +axis; sigma is applied where actions enter and observations leave.  Both entries are pinned by the reference's SHIPPED
# actors (playground/models/*.pt; tools/checkpoint_layout_probe.py, tests/test_shipped_policy_layout.py):
#  * sigma = -1 for the LEFT limbs' x and z joints: the reference's env measures them about the mirrored axis, so that a
#    left/right mirror of the policy's view swaps the limbs WITHOUT negating them.  The actors (trained with
#    common/envs_utils.py:687-740 on the env's own get_mirror_indices()) are mirror-equivariant to 0.06-0.08 under exactly
#    these lists and to 0.36-0.45 (random: 0.43-0.54) with the left x / z joints negated.
#  * sigma = -1 for both KNEES: the reference's knee angle is negative in flexion (an MJCF knee with axis "0 -1 0", range
#    "-160 -2").  Of the 12 joint types it is the one whose sign the shipped policies reject in our env: with it
#    flipped the deterministic Walker3D / Mike policies stay up 81 / 58 control steps instead of 25 / 10 and start reaching the
#    second stone, while flipping any ONE of the other 11 types changes nothing or hurts (profiles/r04_v4_checkpoint_layout_*).
POLICY_SIGN = [1, 1, 1,
               1, 1, 1, -1, 1,
               -1, -1, 1, -1, 1,
               1, 1, 1, 1,
               -1, -1, 1, 1]
# get_mirror_indices() in policy coordinates: the spine's z and x joints negate in place, the limbs swap.
MIRROR_NEGATE_JOINTS = [0, 2]
MIRROR_RIGHT_JOINTS = [3, 4, 5, 6, 7, 13, 14, 15, 16]
MIRROR_LEFT_JOINTS = [8, 9, 10, 11, 12, 17, 18, 19, 20]


# ---------------------------------------------------------------- geometry helpers
def _sphere(c, r):
    m = DENSITY * 4.0 / 3.0 * np.pi * r ** 3
    return m, np.asarray(c, float), np.eye(3) * (0.4 * m * r * r)


def _box(c, half):
    hx, hy, hz = half
    m = DENSITY * 8 * hx * hy * hz
    I = np.diag([m / 3 * (hy * hy + hz * hz), m / 3 * (hx * hx + hz * hz), m / 3 * (hx * hx + hy * hy)])
    return m, np.asarray(c, float), I


def _capsule(p0, p1, r):
    p0 = np.asarray(p0, float)
    p1 = np.asarray(p1, float)
    L = np.linalg.norm(p1 - p0)
    a = (p1 - p0) / L
    mc = DENSITY * np.pi * r * r * L
    ms = DENSITY * 4.0 / 3.0 * np.pi * r ** 3
    i_ax = 0.5 * mc * r * r + 0.4 * ms * r * r
    i_tr = mc * (3 * r * r + L * L) / 12.0 + ms * (0.4 * r * r + 0.25 * L * L + 0.375 * r * L)
    aa = np.outer(a, a)
    I = i_ax * aa + i_tr * (np.eye(3) - aa)
    return mc + ms, 0.5 * (p0 + p1), I


def _compose(geoms):
    """mass, com, inertia about the LINK ORIGIN (link-frame axes)."""
    if not geoms:
        return 0.0, np.zeros(3), np.zeros((3, 3))
    m = sum(g[0] for g in geoms)
    com = sum(g[0] * g[1] for g in geoms) / m
    Io = np.zeros((3, 3))
    for gm, gc, gI in geoms:
        Io += gI + gm * (np.dot(gc, gc) * np.eye(3) - np.outer(gc, gc))
    return m, com, Io


def _deg(lo, hi):
    return [np.deg2rad(lo), np.deg2rad(hi)]


# ---------------------------------------------------------------- robot definitions
JOINT_TYPES = ["abdomen_z", "abdomen_y", "abdomen_x", "hip_x", "hip_z", "hip_y", "knee", "ankle",
               "shoulder_x", "shoulder_z", "shoulder_y", "elbow"]
MASS_GROUPS = ["torso", "lwaist", "pelvis", "thigh", "shin", "foot", "upper_arm", "lower_arm"]

# Every free number of the robot specification, by name (docs/PHYSICS.md section 2).  `build(kind)` evaluates DEFAULTS[kind]; a
# caller may pass `overrides` (tools/sysid_policy.py searches over them; nothing in the package does).
#   ranges are degrees about the +axis of the RIGHT side's link frame (the left side is the mirror image)
_COMMON = dict(
    density=DENSITY,
    mass_mult={g: 1.0 for g in MASS_GROUPS},          # per link group, on top of the primitive's density mass (and its inertia)
    spine_r0=(-0.01, -0.195), spine_r2=-0.13,         # abdomen_z origin in the torso frame (x, z); abdomen_x below abdomen_y
    hip_y=0.10, hip_z=-0.14,                          # hip origin in the pelvis frame (|y|, z)
    thigh=0.34, knee_gap=0.043, shin=0.30, ankle_gap=0.05,      # x leg_scale
    thigh_radius=0.06, shin_radius=0.049,
    foot_box_c=(0.04, -0.05), foot_box_half=(0.10, 0.05, 0.025),  # centre (x, z), half extents
    sole=(0.14, -0.06, 0.05, -0.075),                 # sole corners: x front, x back, |y|, z (foot frame)
    upper_arm=0.28, lower_arm=0.25, shoulder_z=0.06, shoulder_out=0.10,   # x arm_scale (lengths)
    range={"abdomen_z": (-45, 45), "abdomen_y": (-75, 30), "abdomen_x": (-35, 35),
           "hip_x": (-25, 5), "hip_z": (-60, 35), "hip_y": (-110, 20), "knee": (2, 160), "ankle": (-50, 50),
           "shoulder_x": (-120, 30), "shoulder_z": (-60, 60), "shoulder_y": (-120, 60), "elbow": (-140, -2)},
    damping={"abdomen_z": 5, "abdomen_y": 5, "abdomen_x": 5, "hip_x": 5, "hip_z": 5, "hip_y": 5,
             "knee": 1, "ankle": 1, "shoulder_x": 1, "shoulder_z": 1, "shoulder_y": 1, "elbow": 1},
    stiffness={"abdomen_z": 20, "abdomen_y": 10, "abdomen_x": 10, "hip_x": 10, "hip_z": 10, "hip_y": 20,
               "knee": 1, "ankle": 0, "shoulder_x": 1, "shoulder_z": 1, "shoulder_y": 1, "elbow": 0},
    armature={"abdomen_z": .02, "abdomen_y": .02, "abdomen_x": .02, "hip_x": .01, "hip_z": .01, "hip_y": .01,
              "knee": .006, "ankle": .004, "shoulder_x": .004, "shoulder_z": .004, "shoulder_y": .004, "elbow": .003},
    k_lim_per_torque=50.0,        # unilateral limit spring  [N m / rad] per N m of torque limit
    d_lim_per_k=0.02,             # limit damper (active only in violation) [N m s / rad] per N m / rad
    q0_deg={"hip_x": 0.0, "hip_y": -12.0, "knee": 24.0, "ankle": -12.0, "elbow": -20.0},     # nominal pose: slight crouch, sole level
    foot_on_sole=False,           # True (round 6): the foot box is DERIVED from the sole rectangle -- bottom face = the four sole corners,
                                  # height 2 x foot_box_half[2] -- so the contact points always sit on the foot's own geometry
    friction=0.9,
)
DEFAULTS = {
    "walker3d": dict(_COMMON, leg_scale=1.0, arm_scale=1.0, torso_w=0.07, mass_scale=1.0, head_extra=0.0,
                     torque={"abdomen_z": 60, "abdomen_y": 80, "abdomen_x": 60, "hip_x": 80, "hip_z": 60, "hip_y": 100, "knee": 90,
                             "ankle": 60, "shoulder_x": 60, "shoulder_z": 60, "shoulder_y": 50, "elbow": 60}),
    # Mike: stockier body, shorter limbs, stronger legs (own numbers; upstream asset absent)
    "mike": dict(_COMMON, leg_scale=0.85, arm_scale=0.9, torso_w=0.11, mass_scale=1.25, head_extra=0.04,
                 torque={"abdomen_z": 80, "abdomen_y": 100, "abdomen_x": 80, "hip_x": 100, "hip_z": 80, "hip_y": 130, "knee": 120,
                         "ankle": 80, "shoulder_x": 60, "shoulder_z": 60, "shoulder_y": 50, "elbow": 60}),
}


def params(kind, overrides=None):
    """DEFAULTS[kind] with `overrides` applied: {"thigh": 0.36, "torque.knee": 110, "range.hip_y": (-120, 20), "mass_mult.foot": 0.8}."""
    import copy
    P = copy.deepcopy(DEFAULTS[kind])
    for k, v in (overrides or {}).items():
        if "." in k:
            a, b = k.split(".", 1)
            if b not in P[a]:
                raise KeyError(k)
            P[a][b] = v
        else:
            if k not in P:
                raise KeyError(k)
            P[k] = v
    return P


def _key(name):
    return name.replace("right_", "").replace("left_", "")


def _humanoid(P):
    """Shared topology; the numbers of P distinguish Walker3D from Mike."""
    global DENSITY
    dens0, DENSITY = DENSITY, P["density"]        # the primitive helpers read the module constant
    try:
        return _humanoid_at_density(P)
    finally:
        DENSITY = dens0


def _humanoid_at_density(P):
    scale_leg, scale_arm, torso_w, mass_scale, z_extra_head = P["leg_scale"], P["arm_scale"], P["torso_w"], P["mass_scale"], P["head_extra"]
    g = {}          # body -> list of geoms
    grp = {}        # body -> mass group
    r = np.zeros((NJ, 3))
    # ---- torso (body 0): chest capsule across y, head, upper waist
    g[0] = [
        _capsule([0, -torso_w, 0], [0, torso_w, 0], 0.07),
        _sphere([0, 0, 0.19 + z_extra_head], 0.09 + z_extra_head * 0.5),
        _capsule([-0.01, -0.06, -0.12], [-0.01, 0.06, -0.12], 0.06),
    ]
    grp[0] = "torso"
    # ---- spine: abdomen_z (massless) -> abdomen_y (lwaist) -> abdomen_x (pelvis)
    r[0] = [P["spine_r0"][0], 0, P["spine_r0"][1]]
    g[1] = []
    r[1] = [0, 0, 0]
    g[2] = [_capsule([0, -0.06, -0.065], [0, 0.06, -0.065], 0.06)]
    grp[2] = "lwaist"
    r[2] = [0, 0, P["spine_r2"]]
    g[3] = [_capsule([-0.02, -0.07, -0.10], [-0.02, 0.07, -0.10], 0.09)]
    grp[3] = "pelvis"
    # ---- legs
    thigh = P["thigh"] * scale_leg
    knee_off = thigh + P["knee_gap"]
    shin = P["shin"] * scale_leg
    ankle_off = shin + P["ankle_gap"]
    fc, fh = P["foot_box_c"], P["foot_box_half"]
    if P["foot_on_sole"]:
        xf_, xb_, yh_, zs_ = P["sole"]
        fh = (0.5 * (xf_ - xb_), yh_, fh[2])
        fc = (0.5 * (xf_ + xb_), zs_ + fh[2])
    for side, j0 in ((-1.0, 3), (1.0, 8)):
        r[j0] = [0, side * P["hip_y"], P["hip_z"]]   # hip_x origin in pelvis link frame
        g[j0 + 1] = []
        r[j0 + 1] = [0, 0, 0]                 # hip_z co-located
        g[j0 + 2] = []
        r[j0 + 2] = [0, 0, 0]                 # hip_y co-located -> thigh
        g[j0 + 3] = [_capsule([0, 0, 0], [0, 0, -thigh], P["thigh_radius"])]
        grp[j0 + 3] = "thigh"
        r[j0 + 3] = [0, 0, -knee_off]         # knee -> shin
        g[j0 + 4] = [_capsule([0, 0, -0.02], [0, 0, -0.02 - shin], P["shin_radius"])]
        grp[j0 + 4] = "shin"
        r[j0 + 4] = [0, 0, -ankle_off]        # ankle -> foot
        g[j0 + 5] = [_box([fc[0], 0, fc[1]], list(fh))]
        grp[j0 + 5] = "foot"
    # ---- arms
    upper = P["upper_arm"] * scale_arm
    lower = P["lower_arm"] * scale_arm
    for side, j0 in ((-1.0, 13), (1.0, 17)):
        r[j0] = [0, side * (torso_w + P["shoulder_out"]), P["shoulder_z"]]
        g[j0 + 1] = []
        r[j0 + 1] = [0, 0, 0]
        g[j0 + 2] = []
        r[j0 + 2] = [0, 0, 0]
        g[j0 + 3] = [_capsule([0, 0, 0], [0, 0, -upper], 0.04)]
        grp[j0 + 3] = "upper_arm"
        r[j0 + 3] = [0, 0, -upper]
        g[j0 + 4] = [_capsule([0, 0, 0], [0, 0, -lower], 0.031), _sphere([0, 0, -lower - 0.02], 0.04)]
        grp[j0 + 4] = "lower_arm"

    mass = np.zeros(NB)
    com = np.zeros((NB, 3))
    inertia_o = np.zeros((NB, 3, 3))
    for b in range(NB):
        m, c, Io = _compose(g[b])
        f = mass_scale * (P["mass_mult"][grp[b]] if b in grp else 1.0)
        mass[b] = m * f
        com[b] = c
        inertia_o[b] = Io * f

    #            lo   hi          (degrees, about the +axis of the link frame); the left side's x / z joints are the mirror image
    def rng_of(name):
        lo, hi = P["range"][_key(name)]
        if name.startswith("left_") and AXIS[JOINT_NAMES.index(name)] != 1:
            lo, hi = -hi, -lo
        return _deg(lo, hi)
    rng = np.array([rng_of(n) for n in JOINT_NAMES])

    torque = np.array([P["torque"][_key(n)] for n in JOINT_NAMES], float)
    damping = np.array([P["damping"][_key(n)] for n in JOINT_NAMES], float)
    stiffness = np.array([P["stiffness"][_key(n)] for n in JOINT_NAMES], float)
    armature = np.array([P["armature"][_key(n)] for n in JOINT_NAMES], float)
    k_lim = P["k_lim_per_torque"] * torque           # unilateral limit spring  [N m / rad]
    d_lim = P["d_lim_per_k"] * k_lim                 # limit damper (active only in violation) [N m s / rad]

    # nominal pose: slight crouch so reset starts in a balanced, bent-knee stance
    q0 = np.zeros(NJ)
    for side, j0 in ((1.0, 3), (-1.0, 8)):
        q0[j0] = side * np.deg2rad(P["q0_deg"]["hip_x"]) + 0.0   # hip_x about the +x axis of the RIGHT side; the left one is its mirror image
        q0[j0 + 2] = np.deg2rad(P["q0_deg"]["hip_y"])   # hip_y (flexion is negative about +y)
        q0[j0 + 3] = np.deg2rad(P["q0_deg"]["knee"])    # knee
        q0[j0 + 4] = np.deg2rad(P["q0_deg"]["ankle"])   # ankle keeps the sole level
    for j0 in (13, 17):
        q0[j0 + 3] = np.deg2rad(P["q0_deg"]["elbow"])   # elbow inside its range

    # foot sole contact points (foot link frame): 4 corners of the box bottom
    xf, xb, yh, zs = P["sole"]
    corners = np.array([[xf, -yh, zs], [xf, yh, zs], [xb, -yh, zs], [xb, yh, zs]])
    return dict(mass=mass, com=com, inertia_o=inertia_o, r=r, range=rng, torque=torque,
                damping=damping, stiffness=stiffness, armature=armature, k_lim=k_lim, d_lim=d_lim,
                q0=q0, corners=corners, friction=float(P["friction"]))


def identified(kind):
    """The numbers identified against the reference's SHIPPED policy for this robot (tools/sysid_policy.py --emit ->
    steppingstone_amd/identified_<kind>.json; DESIGN.md section 8, docs/PHYSICS.md section 2): overrides of DEFAULTS[kind] under which the
    deterministic `playground/models/*_latest.pt` actor walks the stepping-stone course.  Round 6: re-identified inside stated
    plausibility bounds, on plank-shaped stones, Walker3D with the reference's `_base.pt` actor in the score as well (DESIGN.md section
    8.3; tools/gen_model_tables.py: assert_plausible checks the result).  {} if no file is present."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "identified_%s.json" % kind)
    if not os.path.exists(path):
        return {}
    ov = json.load(open(path))["overrides"]
    return {k: (tuple(v) if isinstance(v, list) else v) for k, v in ov.items()}


def env_constants(use_identified=True):
    """Constants of the ENV (not of a robot) that the identification touched: the stones' STEPPING SURFACE (docs/PHYSICS.md 3.3).
    Rounds 1-4: a disc of 0.25 m = the reference's `step_radius`, which its target / bonus logic uses; round 5: a disc of 0.45 m (the
    shipped policies of both robots get markedly further when the physical surface is larger than that disc) -- whose neighbours
    overlap at the stones' spacing.  Round 6: a PLANK, as SURVEY section 9 recollects the reference's step bodies: footprint
    2 x half_length along the stone's heading by 2 x half_width across it; half_length 0.30 m is the longest plank that cannot overlap
    its neighbour at the smallest stone spacing (0.65 m), the half-width is identified (steppingstone_amd/identified_env.json).  The 0.25 m
    of the step bonus (PHYSICS.md 4.5) is unchanged."""
    import json
    import os
    c = {"stone_plank_half_length": 0.30, "stone_plank_half_width": 0.40}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "identified_env.json")
    if use_identified and os.path.exists(path):
        c.update(json.load(open(path))["constants"])
    return c


def build(kind, overrides=None, use_identified=True):
    """kind: 'walker3d' | 'mike' -> dict of float64 numpy arrays + scalars.  The specification is DEFAULTS[kind] (rounds 1-4: own numbers
    after the roboschool humanoid) with the identified overrides on top (round 5); `overrides` go on top of both;
    use_identified=False evaluates the rounds-1-4 numbers (the prior the identification searches around)."""
    if kind not in DEFAULTS:
        raise ValueError("unknown robot kind %r" % (kind,))
    ov = dict(identified(kind)) if use_identified else {}
    ov.update(overrides or {})
    m = _humanoid(params(kind, ov))
    m["kind"] = kind
    m["parent"] = np.array(PARENT, np.int32)
    m["axis"] = np.array(AXIS, np.int32)
    m["stand_height"] = standing_height(m)
    return m


def _rot(axis, q):
    c, s = np.cos(q), np.sin(q)
    if axis == 0:
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == 1:
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def fk(m, q, base_pos=np.zeros(3), base_rot=np.eye(3)):
    """World pose (R, p) of every body; numpy helper for tests and stand-height."""
    R = [None] * NB
    p = [None] * NB
    R[0], p[0] = np.asarray(base_rot, float), np.asarray(base_pos, float)
    for j in range(NJ):
        b, par = j + 1, PARENT[j]
        p[b] = p[par] + R[par] @ m["r"][j]
        R[b] = R[par] @ _rot(AXIS[j], q[j])
    return R, p


def standing_height(m):
    """Torso-origin height above the sole plane in the nominal pose q0."""
    R, p = fk(m, m["q0"])
    zmin = min((p[b] + R[b] @ c)[2] for b in (RIGHT_FOOT_BODY, LEFT_FOOT_BODY) for c in m["corners"])
    return float(-zmin)
