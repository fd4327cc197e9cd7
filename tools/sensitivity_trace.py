"""Where a control step amplifies fp32 rounding (CPU only: the fp32 and the fp64 build of the oracle on the same state and
action, substep by substep).  Picks the env-steps of a curriculum-5 random-action rollout on which the two builds end farthest
apart and prints, per substep, the active sole corners and the distance between the two builds in: the Delassus operator, the
free foot twist, the solved impulses, the joint-rate change applied by the contact stage, and the state.  The pattern it shows
(docs/HISTORY.md section 3): an error of ~1e-6 in the free foot twist becomes ~1e-5 in the impulses and ~1e-4 in the ankle rate within
one substep (joints 7 / 12: the ankles), and the ankle rate feeds the next substep's foot twist.
usage: python tools/sensitivity_trace.py [kind] > profiles/r03_sensitivity_trace.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import oracle_lib as ol  # noqa: E402
from steppingstone_amd import model as M  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "walker3d"
m = M.build(kind)
n = 256
o32, o64 = ol.OracleEnv(kind, n, seed=11), ol.OracleEnv(kind, n, seed=11, prec="f64")
for o in (o32, o64):
    o.set_curriculum(5)
    o.reset()
cases = []
for t in range(40):
    st = o32.get_state()
    a = o32.random_actions(t)
    o64.set_state(st.astype(np.float64))
    d = np.abs(o32.step(a)[0] - o64.step(a)[0]).max(axis=1)
    for e in np.argsort(d)[-2:]:
        cases.append((float(d[e]), st[e].copy(), a[e].copy()))
cases.sort(key=lambda c: -c[0])
one32, one64 = ol.OracleEnv(kind, 1, seed=0), ol.OracleEnv(kind, 1, seed=0, prec="f64")
print("%s: the 6 env-steps of 10240 on which the fp32 and fp64 builds of the oracle end farthest apart (max |obs| difference)" % kind)
for dist, st, a in cases[:6]:
    tau = (np.asarray(M.POLICY_SIGN) * np.clip(a, -1, 1) * m["torque"]).astype(np.float32)
    one32.set_state(st[None])
    one64.set_state(st[None].astype(np.float64))
    print("env-step with |obs32 - obs64| = %.1e" % dist)
    for k in range(4):
        t32, t64 = one32.debug_contact(0, tau), one64.debug_contact(0, tau.astype(np.float64))
        s32, s64 = one32.get_state()[0], one64.get_state()[0]
        scale = max(np.abs(t64["Li"]).max(), 1e-30)
        jq = int(np.abs(s32[34:55] - s64[34:55]).argmax())
        print("   substep %d  corners %s  Delassus rel %.0e | free foot twist %.0e | impulses %.0e (largest %.2f N s) | contact dqd %.0e | "
              "free qd %.0e | state: qd %.0e (joint %d), q %.0e" % (
                  k, "".join(str(int(x)) for x in t32["active"]), np.abs(t32["Li"] - t64["Li"]).max() / scale,
                  np.abs(t32["V0"] - t64["V0"]).max(), np.abs(t32["lam"] - t64["lam"]).max(), np.abs(t64["lam"]).max(),
                  np.abs(t32["dqd"] - t64["dqd"]).max(), np.abs(t32["qdf"] - t64["qdf"]).max(),
                  np.abs(s32[34:55] - s64[34:55]).max(), jq, np.abs(s32[13:34] - s64[13:34]).max()))
