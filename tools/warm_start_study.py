"""Convergence study behind PHYSICS.md 3.4's round-5 solver setting: distance of one control step's velocities
from a converged contact solve (400 cold sweeps per substep) for sweep counts x warm-start policies, fp64 oracle, 64 envs x 60 steps.

  none    every substep starts from zero                                   (rounds 1-4: 8 sweeps)
  within  substeps 2..4 start from the previous substep of the SAME control step       (round 5: 5 sweeps)
  cross   ... and the first substep from the last substep of the PREVIOUS control step (Bullet's persistent manifolds; needs 24 more
          state words per env that get_state / set_state would have to carry)

"cross" exists only here: the tool builds a private copy of oracle/ss_oracle.c in a temp directory and patches a persistent warm
state into it (the repository's oracle is not touched).   python tools/warm_start_study.py > profiles/r05_warm_start_study.txt"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMP = tempfile.mkdtemp(prefix="ss_warm_study_")
s = open(os.path.join(ROOT, "oracle", "ss_oracle.c")).read()
def rep(a, b):
    global s
    assert a in s, a
    s = s.replace(a, b, 1)
rep("""typedef struct { real lam[8][3]; int stone[8]; } warm_state;   /* impulses of the previous substep of this step */""",
"""typedef struct { real lam[8][3]; int stone[8]; } warm_state;   /* impulses of the previous substep of this step */
static warm_state* g_wsp = 0; static int g_wsp_n = 0; static real g_warm_factor = 1;
void sso_debug_get_warm(void* buf) { memcpy(buf, g_wsp, sizeof(warm_state) * g_wsp_n); }
void sso_debug_set_warm(const void* buf) { memcpy(g_wsp, buf, sizeof(warm_state) * g_wsp_n); }
int sso_debug_warm_size(void) { return (int)sizeof(warm_state); }
void sso_debug_set_warm_factor(double f) { g_warm_factor = (real)f; }""")
rep("""        c->lam[d] = ws->lam[k][d];""", """        c->lam[d] = g_warm_factor * ws->lam[k][d];""")
rep("""  warm_state ws;
  for (int k = 0; k < 8; ++k) ws.stone[k] = -1;
  for (int k = 0; k < 4; ++k) substep(M, s, tau, &fr, &ws);""", """  warm_state ws;
  for (int k = 0; k < 8; ++k) ws.stone[k] = -1;
  if (g_pgs_warm == 2) ws = g_wsp[e];
  for (int k = 0; k < 4; ++k) substep(M, s, tau, &fr, &ws);
  if (g_pgs_warm == 2) g_wsp[e] = ws;""")
rep("""  E->e = (env_state*)calloc((size_t)num_envs, sizeof(env_state));""", """  E->e = (env_state*)calloc((size_t)num_envs, sizeof(env_state));
  g_wsp = (warm_state*)calloc((size_t)num_envs, sizeof(warm_state)); g_wsp_n = num_envs;
  for (int e = 0; e < num_envs; ++e) for (int k = 0; k < 8; ++k) g_wsp[e].stone[k] = -1;""")
rep("""  s->z_init = s->pos[2];
  s->ep_ret = 0;""", """  s->z_init = s->pos[2];
  if (g_wsp && e < g_wsp_n) for (int k = 0; k < 8; ++k) g_wsp[e].stone[k] = -1;
  s->ep_ret = 0;""")
open(os.path.join(TMP, "ss_oracle.c"), "w").write(s)
open(os.path.join(TMP, "ss_model_tables.h"), "w").write(open(os.path.join(ROOT, "oracle", "ss_model_tables.h")).read())
subprocess.check_call(["cc", "-O2", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-DSSO_REAL=double", "-o",
                       os.path.join(TMP, "liboracle_f64.so"), os.path.join(TMP, "ss_oracle.c"), "-lm"])
os.environ["SS_ORACLE_LIB_F64"] = os.path.join(TMP, "liboracle_f64.so")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, oracle_lib as ol
lib=ol.load("f64")
lib.sso_debug_set_solver.argtypes=[C.c_int,C.c_int]
lib.sso_debug_get_warm.argtypes=[C.c_void_p]; lib.sso_debug_set_warm.argtypes=[C.c_void_p]
lib.sso_debug_set_warm_factor.argtypes=[C.c_double]
n=64
WS=lib.sso_debug_warm_size()
def getw():
    b=np.zeros(n*WS,np.uint8); lib.sso_debug_get_warm(b.ctypes.data_as(C.c_void_p)); return b
def setw(b): lib.sso_debug_set_warm(b.ctypes.data_as(C.c_void_p))
for label, amp in (("random actions (x1.0 / x0.3 alternating)", None), ("small actions x0.15 (standing / swaying)", 0.15)):
    o=ol.OracleEnv("walker3d",n,seed=1,prec="f64"); o.reset()
    lib.sso_debug_set_solver(5,2); lib.sso_debug_set_warm_factor(1.0)
    states,acts,warms=[],[],[]
    for t in range(60):
        a=o.random_actions(t)*((0.3 if t%2 else 1.0) if amp is None else amp)
        states.append(o.get_state().copy()); acts.append(a); warms.append(getw())
        o.step(a)
    def run(iters,warm,factor=1.0):
        lib.sso_debug_set_solver(iters,warm); lib.sso_debug_set_warm_factor(factor)
        out=[]
        for st,a,w in zip(states,acts,warms):
            o.set_state(st); o.set_auto_reset(False); setw(w)
            o.step(a); out.append(o.get_state()[:,7:55].copy())
        return np.array(out)
    ref=run(400,0)
    inc=np.array([st[:,64]!=0 for st in states])
    scale=np.abs(ref).max(axis=2,keepdims=True)+1e-3
    print("==",label, "(%d env-steps in contact)"%inc.sum())
    for iters,warm,f in ((8,0,1),(5,0,1),(5,1,1),(5,2,1),(5,2,0.85),(4,1,1),(4,2,1),(3,2,1),(8,1,1),(8,2,1),(16,0,1)):
        got=run(iters,warm,f)
        err=(np.abs(got-ref)/scale).max(axis=2)[inc]
        print("sweeps %2d warm %s factor %.2f: rel. velocity error median %.2e  p75 %.2e  p90 %.2e  max %.2e"%(iters,{0:"none  ",1:"within",2:"cross "}[warm],f,np.median(err),np.quantile(err,.75),np.quantile(err,.9),err.max()))
    o.close()
