"""Cost of a rollout launch as a function of its length (run on the GPU box): HIP-event time of launches of K control steps at 4096
envs, back to back and with idle gaps between them.  docs/HISTORY.md 4.1a quotes the table."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
g = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device="cuda:0", return_numpy=False)
g.reset()
g.rollout_random(512, 0)
torch.cuda.synchronize()
def run(K, reps, idle):
    ts = []
    t0 = 1000
    for r in range(reps):
        if idle:
            torch.cuda.synchronize(); time.sleep(idle)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.rollout_random(K, t0, steps_per_launch=K); b.record()
        t0 += K
        ts.append((a, b))
    torch.cuda.synchronize()
    v = sorted(x.elapsed_time(y) * 1000 for x, y in ts)
    return v[len(v) // 2], v[0]
for idle in (0, 0.0002, 0.002, 0.02):
    print("idle %.4f s between launches:" % idle, end=" ")
    for K in (1, 2, 5, 10, 20, 50, 200):
        med, mn = run(K, 30, idle)
        print("K=%d %.1f us (%.2f/step)" % (K, med, med / K), end=" | ")
    print(flush=True)
