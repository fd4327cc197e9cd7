"""Fused PPO minibatch step (libsslearner.so) on one GPU: time per step and achieved f32 MFMA rate at several minibatch
sizes, next to the torch (autograd + Adam, hipGraph) step.  FLOPs per sample: forward 2 x (weights of actor + critics),
backward data the same minus the input layers, backward weights the same as forward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import fused_ppo
from steppingstone_amd import ppo
dev = torch.device("cuda:0")
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1
R = 131072
w_actor = 60 * 256 + 4 * 256 * 256 + 256 * 21
w_critic = 60 * 256 + 3 * 256 * 256 + 256
flop_per_sample = 2 * (w_actor + E * w_critic) * 3 - 2 * (60 * 256) * (1 + E)
g = torch.Generator(device="cpu").manual_seed(0)
data = tuple(torch.randn(R, w, generator=g).to(dev).contiguous() for w in (60, 21, 1, 1, 1, 1))
for mb in (1024, 2048, 4096, 16384):
    torch.manual_seed(1)
    ac = ppo.ActorCritic(num_ensembles=E).to(dev)
    agent = fused_ppo.FusedPPO(ac, mini_batch_size=mb, use_graph=False)
    idx = torch.randperm(R, device=dev)[:mb]
    for _ in range(3):
        agent.step_minibatch(data, idx)
    K = 200
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(K):
            agent._launch(data, idx)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / K
    tf = flop_per_sample * mb / (us * 1e-6) / 1e12
    print("fused  E=%d minibatch %5d: %7.1f us per step, %5.1f TFLOP/s f32 MFMA (%.1f %% of the 157.3 TFLOP/s peak)" % (E, mb, us, tf, 100 * tf / 157.3), flush=True)
for mb in (1024, 4096):
    torch.manual_seed(1)
    ac = ppo.ActorCritic(num_ensembles=E).to(dev)
    agent = ppo.PPO(ac, mini_batch_size=mb, use_graph=True)
    d6 = (data[0], data[1], data[2], data[3], data[4], data[5])
    idx = torch.randperm(R, device=dev)[:mb]
    for k in range(6):
        agent._graph_step(d6, idx, refresh=(k == 0))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        agent._graph_step(d6, idx, refresh=False)
    e1.record(); torch.cuda.synchronize()
    print("torch  E=%d minibatch %5d: %7.1f us per step (autograd + capturable Adam, one hipGraph replay per step)" % (E, mb, 1e3 * e0.elapsed_time(e1) / 50), flush=True)
