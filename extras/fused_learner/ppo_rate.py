"""PPO end-to-end frames/s (BASELINE configs[4] shape on one GPU: Mike, curriculum on, 32-step rollouts, 10 epochs):
eager vs hipGraph and minibatch sizes; steady state = updates 4..10 (after warm-up and graph capture)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fused_ppo
from steppingstone_amd import ppo
from steppingstone_amd.envs import SteppingStoneVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mirror = "--mirror" in sys.argv            # the reference's --mirror (use_mirror: every minibatch doubled with its mirror image)
rows = ((True, 1024, "torch"), (True, 1024, "fused"), (True, 4096, "fused")) if mirror else (
    (False, 1024, "torch"), (True, 1024, "torch"), (True, 4096, "torch"), (True, 16384, "torch"),
    (True, 1024, "fused"), (True, 4096, "fused"), (True, 16384, "fused"))
for use_graph, mb, learner in rows:
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=8, device="cuda:0", return_numpy=False)
    stamps = []
    def log(st):
        torch.cuda.synchronize()
        stamps.append((time.time(), st["total_num_steps"], st["mean_rew"]))
    ac, hist = ppo.train(envs, 10, num_steps=32, ppo_epoch=10, mini_batch_size=mb, log=log, use_graph=use_graph, agent_factory=(fused_ppo.FusedPPO if learner == "fused" else None),
                          use_mirror=mirror)
    fps = (stamps[-1][1] - stamps[3][1]) / (stamps[-1][0] - stamps[3][0])
    print("%d envs  mirror=%d graph=%d  learner=%-5s minibatch %5d: %7.0f frames/s steady state, mean episode return %.1f -> %.1f" %
          (n, mirror, use_graph, learner, mb, fps, stamps[0][2], stamps[-1][2]), flush=True)
    envs.close()
