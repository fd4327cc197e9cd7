"""Per-step launches of BASELINE's 4096 envs as ONE 4096-env batch on one stream against TWO 2048-env half-batches on two streams
("at 4096 envs half the chip is idle, so two 2048-env half-batches on two streams can overlap one half's boundary
with the other half's body").  Run on the GPU box:   python tools/two_stream_probe.py [steps]

Rows (HIP events over `steps` control steps, median of 5 repeats, on-device actions, one launch per control step and half):
  A  one env of 4096, one stream                                   (the bench line's per_step_launch)
  B  two envs of 2048 (global ids 0..2047 / 2048..4095), two streams, launches alternating A0 B0 A1 B1 ...
  C  the same two halves on ONE stream                             (what the split alone costs)
  D  B with a stand-in policy between the steps of each half: the actor MLP (60-256x5-21, torch) on that half's observations on the
     half's own stream, its output consumed as the next step's action tensor -- the loop a learner actually runs -- against
  E  the same policy on the whole 4096-env batch, one stream.
The union of the two halves is bit-identical to the single batch (tests/test_gpu_parity.py::test_large_batch_is_a_union_of_small_ones);
row B's final state is compared with row A's here as well."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv
from steppingstone_amd import ppo

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ENV = "Walker3DStepperEnv-v0"


def timed(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / steps)
    return sorted(ts)[len(ts) // 2]


def make(n, off):
    e = SteppingStoneVecEnv(ENV, n, seed=0, device="cuda:0", env_id_offset=off)
    e.reset()
    return e


whole = make(4096, 0)
t0 = [0]


def row_a():
    for k in range(steps):
        whole.rollout_random(1, t0=t0[0] + k, steps_per_launch=1)


ms_a = timed(row_a)
halves = [make(2048, 0), make(2048, 2048)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def row_b():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for k in range(steps):
        for h, s in zip(halves, streams):
            with torch.cuda.stream(s):
                h.rollout_random(1, t0=t0[0] + k, steps_per_launch=1)
    for s in streams:
        cur.wait_stream(s)


def row_c():
    for k in range(steps):
        for h in halves:
            h.rollout_random(1, t0=t0[0] + k, steps_per_launch=1)


ms_b = timed(row_b)
ms_c = timed(row_c)
# same number of steps from the same reset on both: 5 timed repeats each of rows A and (B + C = 10 repeats) differ, so re-run a clean pair
w2, h2 = make(4096, 0), [make(2048, 0), make(2048, 2048)]
for k in range(64):
    w2.rollout_random(1, t0=k, steps_per_launch=1)
    for h, s in zip(h2, streams):
        with torch.cuda.stream(s):
            h.rollout_random(1, t0=k, steps_per_launch=1)
torch.cuda.synchronize()
same = torch.equal(w2.get_state(), torch.cat([h.get_state() for h in h2]))

actor = ppo.Actor().cuda()


def policy_loop(envs, strs):
    cur = torch.cuda.current_stream()
    for s in strs:
        s.wait_stream(cur)
    obs = [e._obs for e in envs]
    with torch.no_grad():
        for k in range(steps):
            for i, (e, s) in enumerate(zip(envs, strs)):
                with torch.cuda.stream(s):
                    a = actor(obs[i])
                    e.step_async(a)
                    e._pending = False
    for s in strs:
        cur.wait_stream(s)


ms_d = timed(lambda: policy_loop(halves, streams))
ms_e = timed(lambda: policy_loop([whole], [torch.cuda.current_stream()]))
print("per control step of 4096 envs (ms): A one batch %.4f | B two halves, two streams %.4f | C two halves, one stream %.4f | halves == whole bitwise: %s"
      % (ms_a, ms_b, ms_c, same))
print("with the actor MLP between the steps (torch eager): E one batch %.4f | D two halves, two streams %.4f" % (ms_e, ms_d))
