"""Where a PPO update spends its time (one GPU, Mike, 4096 envs, 32-step rollouts): graph rollout, GAE, learner."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fused_ppo
from steppingstone_amd import ppo
from steppingstone_amd.envs import SteppingStoneVecEnv
dev = torch.device("cuda:0")
n, T = 4096, 32
for learner, mb in (("torch", 1024), ("fused", 1024), ("fused", 4096), ("fused", 16384)):
    torch.manual_seed(8)
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=8, device=dev, return_numpy=False)
    ac = ppo.ActorCritic().to(dev)
    agent = (fused_ppo.FusedPPO(ac, mini_batch_size=mb) if learner == "fused" else ppo.PPO(ac, mini_batch_size=mb, use_graph=True))
    roll = ppo.Rollouts(T, n, dev)
    ring = ppo.EpisodeRing(n, dev)
    roll.obs[0].copy_(envs.reset())
    col = ppo.GraphedCollector(envs, ac, roll, T, ring=ring)
    t = {"rollout": 0.0, "gae": 0.0, "update": 0.0}
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        col(); torch.cuda.synchronize(); t1 = time.perf_counter()
        with torch.no_grad():
            nv = ac.get_value(roll.obs[-1])
        roll.compute_returns(nv, True, 0.99, 0.95); torch.cuda.synchronize(); t2 = time.perf_counter()
        agent.update(roll); roll.after_update(); torch.cuda.synchronize(); t3 = time.perf_counter()
        if it >= 3:
            t["rollout"] += t1 - t0; t["gae"] += t2 - t1; t["update"] += t3 - t2
    steps = 10 * (T * n // mb)
    print("%-5s mb %5d: rollout %.1f ms, GAE %.1f ms, update %.1f ms (%d minibatch steps, %.0f us each) -> %.0f frames/s" % (
        learner, mb, 1e3 * t["rollout"] / 3, 1e3 * t["gae"] / 3, 1e3 * t["update"] / 3, steps, 1e6 * t["update"] / 3 / steps,
        T * n / ((t["rollout"] + t["gae"] + t["update"]) / 3)), flush=True)
    envs.close()
