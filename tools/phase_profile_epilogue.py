"""The control step's EPILOGUE in detail (shader clocks of the main wavefront per control step).  Needs a library built with
-DSS_PROFILE_PHASES -DSS_PROFILE_EPILOGUE;  STEPPINGSTONE_LIB=.../libss_profe.so python tools/phase_profile_epilogue.py [envs] [steps per launch]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from steppingstone_amd import _lib
from steppingstone_amd.envs import SteppingStoneVecEnv
NAMES = ["four substeps (+ loop entry)", "issue of the epilogue's global loads", "LDS read-back, pair exchange of foot reports", "joint sums (energy, limits, finiteness)",
         "wait for the loads + target logic (+ draw on advance)", "progress, termination, roll / pitch, reward", "info words", "reset branch (Philox)",
         "joint values of the next state", "LDS refresh + hand-off / inline output stage", "env-level stores + membar", "", "", "", "", ""]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = 200
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=0, device="cuda:0")
env.reset()
lib = _lib.load()
lib.ss_debug_phase_cycles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
out = np.zeros(16, np.uint64)
env.rollout_random(50, 0, steps_per_launch=spl)
lib.ss_debug_phase_cycles(env.backend.h, out.ctypes.data_as(C.c_void_p), 1)
env.rollout_random(steps, 50, steps_per_launch=spl)
lib.ss_debug_phase_cycles(env.backend.h, out.ctypes.data_as(C.c_void_p), 1)
waves = (n + 31) // 32
per = out.astype(np.float64) / (waves * steps)
tot = per.sum()
print("%d envs, steps per launch %s: cycles per wave per control step: %.0f  (%.1f us at 2.4 GHz); epilogue %.0f" % (n, spl or "library default", tot, tot / 2400.0, per[1:].sum()))
for nm, v in zip(NAMES, per):
    if nm:
        print("  %-56s %9.0f  %5.1f %%" % (nm, v, 100 * v / tot))
