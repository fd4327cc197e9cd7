"""Host-side cost of the per-step all-gather path (run under torch.distributed.run, 1+ ranks, SS_FORCE_COLLECTIVE=1):
enqueue-only wall time vs completed wall time for K steps, with and without the collective."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
from steppingstone_amd.distributed import ShardedVecEnv
from steppingstone_amd.envs import SteppingStoneVecEnv
lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dev = torch.device("cuda", lr)
dist.init_process_group("nccl", device_id=dev)
local = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device=dev, env_id_offset=dist.get_rank() * 4096, return_numpy=False)
env = ShardedVecEnv(local)
env.reset()
K = 2000
for gather in (False, True, False, True):
    env.rollout_random(200, 0, gather=gather)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    slot = 0
    for k in range(K):
        slot = k % len(env._packed)
        if slot == 0:
            for i in range(len(env._packed)): env._wait(i)
        env.local.step_packed(env._packed[slot], actions=None, t=k)
        if gather:
            env._gather_packed(slot, async_op=True)
    t1 = time.perf_counter()
    for i in range(len(env._packed)): env._wait(i)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if dist.get_rank() == 0:
        print("gather=%d  enqueue %.1f us/step   complete %.1f us/step" % (gather, 1e6 * (t1 - t0) / K, 1e6 * (t2 - t0) / K), flush=True)
dist.destroy_process_group()
