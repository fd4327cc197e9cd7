import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
g = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device="cuda:0", return_numpy=False)
g.reset(); g.rollout_random(512, 0); torch.cuda.synchronize()
def run(K, reps):
    ts=[]; t0=1000
    for r in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); g.rollout_random(K,t0,steps_per_launch=K); b.record(); t0+=K; ts.append((a,b))
    torch.cuda.synchronize()
    v=sorted(x.elapsed_time(y)*1000 for x,y in ts); return v[len(v)//2]/K
print("K=1 %.2f  K=20 %.2f  K=1000 %.3f us/step" % (run(1,200), run(20,30), run(1000,5)))
