"""Clean per-step timing in two controlled regimes: every env standing on its stone (both feet in contact) and
every env in free flight.  The state is re-injected before each timed step so ablation variants cannot drift."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
n = 4096
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=0, device="cuda:0")
env.reset()
stand = env.get_state().clone()
fly = stand.clone(); fly[:, 2] += 5.0; fly[:, 56] += 5.0
act = torch.zeros((n, 21), device="cuda")
for name, st in (("standing", stand), ("flying", fly)):
    ts = []
    for rep in range(30):
        env.set_state(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step_async(act)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[5:])
    print("%-9s median %.4f ms  min %.4f" % (name, ts[len(ts) // 2], ts[0]))
