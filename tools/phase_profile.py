"""Per-phase shader-clock breakdown of the step kernel.  Needs a library built with -DSS_PROFILE_PHASES:
   hipcc <FLAGS> -DSS_PROFILE_PHASES csrc/ss_api.hip -o lib/libss_prof.so ; STEPPINGSTONE_LIB=.../libss_prof.so python tools/phase_profile.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from steppingstone_amd import _lib
from steppingstone_amd.envs import SteppingStoneVecEnv
NAMES = ["loop/entry + wait #0", "torques (+ cs loads)", "pass1 vel", "pass2 ABI + spine", "base chol + hand-off", "pass3 acc", "detect FK / read",
         "WAIT #3 (helper variants; else Linv)", "Vfree+rows", "PGS", "final resp", "integrate", "WAIT #0b", "reward/obs/store", "WAIT #1", "WAIT #2"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spl = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # steps per launch (0: library default; 1: the one-launch-per-step kernel)
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
waves = (n + 31) // 32      # two lanes per env: 32 envs per wavefront
per = out.astype(np.float64) / (waves * steps)
tot = per.sum()
print("cycles per wave per control step: %.0f  (%.1f us at 2.4 GHz)" % (tot, tot / 2400.0))
for nm, v in zip(NAMES, per):
    if nm:
        print("  %-18s %9.0f  %5.1f %%" % (nm, v, 100 * v / tot))
