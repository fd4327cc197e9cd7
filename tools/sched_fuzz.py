"""Schedule-fuzz fingerprint (run on the GPU box).

    python tools/sched_fuzz.py [ENV_STEPS_PER_CONFIG [TINY_STEPS]]                   -> one line per configuration: a position-sensitive
                                                                              checksum of EVERY step's packed [N,62] block
                                                                              (obs | rew | done), of every step's info words and
                                                                              of the final state
    STEPPINGSTONE_LIB=var/libss_fuzz.so python tools/sched_fuzz.py ...      the same with the -DSS_FUZZ_SCHED build, whose
                                                                              wavefronts sleep pseudo-random times (seeded by the
                                                                              shader clock: different in every run) at the start of
                                                                              every barrier window

Configurations: both robots x {three helpers, one helper, plain kernel} x {one launch per step with an action tensor (steps/launch=0 below), one launch per step with
on-device actions, 32 steps per launch}, curriculum 5
(stone draws, resets, target advances all occur), plus ragged tiny batches (1 ... 700 envs).  Two runs that print the same lines
computed the same bits on every env-step.  tools/sched_fuzz.sh runs plain once and fuzzed three times and diffs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from steppingstone_amd.envs import SteppingStoneVecEnv  # noqa: E402

TARGET = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
TINY_STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
CHUNK = 32
# ragged tiny batches: n mod 64 in 1..32 is where the array padding (64 envs) exceeds the launch's last 32-env workgroup -- the odd
# ones (1, 3, 5, 31, 65, 67) are the batch sizes whose packed row of env n - 1 was schedule-dependent until round 4
TINY = (1, 3, 5, 31, 33, 65, 67, 700)


class Sum:
    """Two int64 checksums over the bit patterns, the second weighted by position (wrap-around arithmetic)."""

    def __init__(self, dev):
        self.a = torch.zeros((), dtype=torch.int64, device=dev)
        self.b = torch.zeros((), dtype=torch.int64, device=dev)
        self.w = None
        self.count = 0

    def add(self, t):
        v = t.contiguous().view(-1).view(torch.int32).to(torch.int64)
        if self.w is None or self.w.numel() != v.numel():
            self.w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003) + 1
        self.count += 1
        self.a += v.sum() * self.count
        self.b += (v * self.w).sum() + self.count

    def hex(self):
        return "%016x%016x" % (int(self.a.item()) & (2 ** 64 - 1), int(self.b.item()) & (2 ** 64 - 1))


def run(env_id, n, helpers, per_launch, steps):
    os.environ["SS_HELPERS"] = str(helpers)          # read by ss_create
    e = SteppingStoneVecEnv(env_id, n, seed=21, device="cuda:0", return_numpy=False)
    e.update_curriculum(5)
    e.reset()
    S = Sum(e.device)
    t = 0
    if per_launch == 0:                              # explicit action tensor: the kernel a policy drives (ss_step_packed with act)
        packed = torch.zeros((n, 62), device=e.device)
        while t < steps:
            e.step_packed(packed, actions=e.random_actions(t), t=t)
            S.add(packed)
            S.add(e._info)
            t += 1
    elif per_launch == 1:
        packed = torch.zeros((n, 62), device=e.device)
        while t < steps:
            e.step_packed(packed, actions=None, t=t)
            S.add(packed)
            S.add(e._info)
            t += 1
    else:
        packed = torch.zeros((per_launch, n, 62), device=e.device)
        while t < steps:
            k = min(per_launch, steps - t)
            e.rollout_random_packed(packed[:k], t0=t)
            S.add(packed[:k])
            S.add(e._info)
            t += k
    S.add(e.get_state())
    e.close()
    return S.hex()


if __name__ == "__main__":
    lib = os.environ.get("STEPPINGSTONE_LIB", "in-tree")
    print("# library:", lib, " env-steps per configuration:", TARGET, flush=True)
    total = 0
    for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
        for n, helpers in ((4096, 3), (4096, 1), (4096, 0), (16384, 1), (33000, 0)):
            for per_launch in (0, 1, CHUNK):
                steps = max(CHUNK, TARGET // n)
                print("%-22s n=%-6d helpers=%d steps/launch=%-3d steps=%-6d %s" % (env_id, n, helpers, per_launch, steps,
                                                                                  run(env_id, n, helpers, per_launch, steps)), flush=True)
                total += steps * n
        for n in TINY:
            for helpers in (3, 1, 0):
                for per_launch in (0, 1, 7):
                    steps = TINY_STEPS
                    print("%-22s n=%-6d helpers=%d steps/launch=%-3d steps=%-6d %s" % (env_id, n, helpers, per_launch, steps,
                                                                                      run(env_id, n, helpers, per_launch, steps)), flush=True)
                    total += steps * n
    print("# env-steps:", total)
