"""What does cross-stream synchronisation cost per step next to the 121 us step kernel?  (single GPU, no RCCL)
V0 kernels only; V1 + event record per step; V2 + side stream waits and copies the 1 MB block; V3 + main stream waits
on the side copy of two steps ago (the double-buffer dependency of ShardedVecEnv)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
dev = torch.device("cuda", 0)
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device=dev, return_numpy=False)
env.reset()
packed = [torch.zeros((4096, 62), device=dev) for _ in range(2)]
dst = [torch.zeros((4096, 62), device=dev) for _ in range(2)]
side = torch.cuda.Stream()
K = 2000
for variant in (0, 1, 2, 3, 0, 1, 2, 3):
    evs = [None, None]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        s = k & 1
        if variant >= 3 and evs[s] is not None:
            torch.cuda.current_stream().wait_event(evs[s])
        env.step_packed(packed[s], actions=None, t=k)
        if variant >= 1:
            e = torch.cuda.Event(); e.record()
        if variant >= 2:
            side.wait_event(e)
            with torch.cuda.stream(side):
                dst[s].copy_(packed[s], non_blocking=True)
                e2 = torch.cuda.Event(); e2.record()
            evs[s] = e2
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("V%d  enqueue %.1f us/step   complete %.1f us/step" % (variant, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K), flush=True)

# V4: the V3 pattern captured in a hipGraph of G steps, replayed K/G times (t baked in: timing experiment only)
G = 8
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream()
with torch.cuda.graph(g, stream=cap):
    evs = [None, None]
    for k in range(G):
        s = k & 1
        if evs[s] is not None:
            torch.cuda.current_stream().wait_event(evs[s])
        env.step_packed(packed[s], actions=None, t=k)
        e = torch.cuda.Event(); e.record()
        side.wait_event(e)
        with torch.cuda.stream(side):
            dst[s].copy_(packed[s], non_blocking=True)
            e2 = torch.cuda.Event(); e2.record()
        evs[s] = e2
    for e2 in evs:
        torch.cuda.current_stream().wait_event(e2)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K // G):
        g.replay()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("V4 graph(%d)  enqueue %.1f us/step   complete %.1f us/step" % (G, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K), flush=True)
