"""Probe: does a hipGraph of G x [step kernel + all_gather_into_tensor] replay correctly and faster than the eager
double-buffered loop?  (run under torch.distributed.run; t is baked into the graph here: timing probe only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from steppingstone_amd.envs import SteppingStoneVecEnv
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
W = dist.get_world_size()
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device=dev, env_id_offset=dist.get_rank() * 4096, return_numpy=False)
env.reset()
G = 8
packed = [torch.zeros((4096, 62), device=dev) for _ in range(G)]
gathered = [torch.zeros((4096 * W, 62), device=dev) for _ in range(G)]
def body():
    works = []
    for k in range(G):
        env.step_packed(packed[k], actions=None, t=k)
        works.append(dist.all_gather_into_tensor(gathered[k], packed[k], async_op=True))
    for w in works:
        w.wait()
from steppingstone_amd.distributed import ShardedVecEnv
sh = ShardedVecEnv(env)
K = 2000
if os.environ.get("PROBE_FIRST"):
    sh.rollout_random(200, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.rollout_random(K, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("first thing: ShardedVecEnv.rollout_random  %.1f us/step" % (1e6 * (t1 - t0) / K), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    body()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
K = 2000
t0 = time.perf_counter()
for _ in range(K // G):
    body()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("eager  %.1f us/step" % (1e6 * (t1 - t0) / K), flush=True)
# variant: the library path
from steppingstone_amd.distributed import ShardedVecEnv
sh = ShardedVecEnv(env)
sh.rollout_random(64, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
sh.rollout_random(K, 0)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("ShardedVecEnv.rollout_random  %.1f us/step" % (1e6 * (t1 - t0) / K), flush=True)
# variant: same loop as body() but through sh's buffers
def body2():
    for k in range(G):
        env.step_packed(sh._packed[k], actions=None, t=k)
        sh._gather_packed(k, async_op=True)
    for k in range(G):
        sh._wait(k)
body2(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K // G):
    body2()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("body2 (sh buffers)  %.1f us/step" % (1e6 * (t1 - t0) / K), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
g.replay(); torch.cuda.synchronize()
ok = all(torch.equal(gathered[k][:4096], packed[k]) for k in range(G))
t0 = time.perf_counter()
for _ in range(K // G):
    g.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("graph  %.1f us/step  (gathered == packed: %s)" % (1e6 * (t1 - t0) / K, ok), flush=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.rollout_random(K, 0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("after graph: ShardedVecEnv.rollout_random  %.1f us/step" % (1e6 * (t1 - t0) / K), flush=True)
dist.destroy_process_group()
