"""Which stream arrangement lets the RCCL all-gather overlap the step kernel?  MODE=plain|burn|side  (one rank,
SS_FORCE_COLLECTIVE=1).  plain: everything from the default stream; burn: create and use one pool stream before the
first collective; side: run the rollout on a non-default stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from steppingstone_amd.distributed import ShardedVecEnv
from steppingstone_amd.envs import SteppingStoneVecEnv
mode = os.environ.get("MODE", "plain")
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device=dev, env_id_offset=dist.get_rank() * 4096, return_numpy=False)
if mode == "burn":
    s0 = torch.cuda.Stream()
    with torch.cuda.stream(s0):
        torch.zeros(8, device=dev).add_(1)
    torch.cuda.synchronize()
sh = ShardedVecEnv(env)
K = 2000
def run():
    sh.rollout_random(200, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sh.rollout_random(K, 0)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / K
if mode == "side":
    s1 = torch.cuda.Stream()
    with torch.cuda.stream(s1):
        us = run()
else:
    us = run()
print("MODE=%s GPU_MAX_HW_QUEUES=%s: %.1f us/step" % (mode, os.environ.get("GPU_MAX_HW_QUEUES"), us), flush=True)
dist.destroy_process_group()
