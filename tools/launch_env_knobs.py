"""One launch per control step at 4096 envs under runtime knobs (run on the GPU box; VERDICT r3 weak item 6: the ~15 us kernel boundary).
Prints the HIP-event time per step of 2000 back-to-back one-step launches and of a replayed hipGraph of 50 one-step kernel nodes.
Called once per environment-variable setting by tools/launch_env_knobs.sh (the knobs are read when the runtime starts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
g = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 4096, seed=0, device="cuda:0", return_numpy=False)
g.reset()
g.rollout_random(512, 0)
torch.cuda.synchronize()


def timed(fn, reps=5):
    out = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); n = fn(); b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1000 / n)
    return sorted(out)[len(out) // 2]


t = [1000]
def eager():
    g.rollout_random(2000, t[0], steps_per_launch=1); t[0] += 2000
    return 2000
def k1000():
    g.rollout_random(2000, t[0], steps_per_launch=1000); t[0] += 2000
    return 2000
res = {"eager_us": timed(eager), "k1000_us": timed(k1000)}
try:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g.rollout_random(50, 0, steps_per_launch=1)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            g.rollout_random(50, 5000, steps_per_launch=1)
    def replay():
        for _ in range(40):
            gr.replay()
        return 2000
    res["graph50_us"] = timed(replay)
except Exception as e:                    # noqa: BLE001
    res["graph50_us"] = "failed: %s" % (str(e).splitlines()[0][:80],)
knobs = {k: os.environ[k] for k in ("HIP_FORCE_DEV_KERNARG", "AMD_OPT_FLUSH", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "GPU_MAX_HW_QUEUES", "AMD_DIRECT_DISPATCH", "DEBUG_HIP_GRAPH_BATCH_SIZE") if k in os.environ}
print("%-60s one launch per step %.2f us | K = 1000 per launch %.2f us | graph of 50 one-step nodes %s" % (
    knobs or "(defaults)", res["eager_us"], res["k1000_us"], res["graph50_us"] if isinstance(res["graph50_us"], str) else "%.2f us" % res["graph50_us"]), flush=True)
