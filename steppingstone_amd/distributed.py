"""Multi-GPU sharding of the vectorised env: one process per GPU (torchrun), contiguous blocks of N/G envs per
rank, rank-local state never moves.  The reference's only parallelism is one OS process per env behind pipes
(common/envs_utils.py:519-538); here the single exchange per step is an all-gather (RCCL over xGMI when the
backend is "nccl") of the packed [N/G, 62] = [obs | rew | done] block so every rank holds the (N,60) obs and (N,)
rew / done the single-learner loop of playground/train.py:363-469 expects.  Actions flow the other way by slicing.
Global env ids (env_id_offset = rank * N/G) key the RNG streams, so results do not depend on G.
"""
import os

import torch
import torch.distributed as dist

from ._lib import ACT_DIM, OBS_DIM

PACK = OBS_DIM + 2
DEPTH = 8


class ShardedVecEnv:
    """Wraps this rank's local vec env (any object with .step / .reset / .num_envs returning device tensors)."""

    def __init__(self, local_env, group=None):
        self.local = local_env
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # SS_FORCE_COLLECTIVE=1 issues the all-gather even at world size 1 (exercises the RCCL path on a 1-GPU box)
        self._collective = self.world > 1 or (dist.is_initialized() and os.environ.get("SS_FORCE_COLLECTIVE") == "1")
        self.n_local = int(local_env.num_envs)
        if self._collective and torch.device(local_env.device).type == "cuda":
            # HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) round robin, and the first
            # stream taken from PyTorch's pool lands on the default stream's queue: if that is RCCL's stream, the
            # all-gather cannot run under the step kernel (measured +26 us per 0.11 ms step instead of +9 us,
            # tools/gather_queue_probe.py).  Take and use one pool stream before the first collective.
            self._spare_stream = torch.cuda.Stream(device=local_env.device)
            with torch.cuda.stream(self._spare_stream):
                torch.zeros(8, device=local_env.device).add_(1)
            self._spare_stream.synchronize()
        self.num_envs = self.n_local * self.world
        self.observation_space = getattr(local_env, "observation_space", None)
        self.action_space = getattr(local_env, "action_space", None)
        dev = local_env.device
        # ring of DEPTH buffers: the all-gather of step t runs under the kernels of steps t+1.. and the launch stream
        # waits on the collectives once per DEPTH steps, as one batch.  Measured on one rank with the collective forced:
        # a cross-stream wait before every kernel (any depth) costs +26 us per 0.11 ms step, batched waits every 8
        # steps +7 us; a hipGraph of the 8 steps is slower (+13 us) (tools/gather_overhead.py, gather_graph_probe.py).
        self._packed = [torch.zeros((self.n_local, PACK), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
        self._gathered = [torch.zeros((self.num_envs, PACK), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
        self._work = [None] * DEPTH

    # -- helpers
    def local_slice(self):
        return slice(self.rank * self.n_local, (self.rank + 1) * self.n_local)

    @staticmethod
    def _split(g):
        return g[:, :OBS_DIM], g[:, OBS_DIM], g[:, OBS_DIM + 1] > 0.5

    def _wait(self, slot):
        if self._work[slot] is not None:
            self._work[slot].wait()
            self._work[slot] = None

    def _gather_packed(self, slot, async_op=False):
        """All-gather the local packed block of `slot` (the step kernel wrote it directly: no extra copies)."""
        if not self._collective:
            return self._packed[slot]
        w = dist.all_gather_into_tensor(self._gathered[slot], self._packed[slot], group=self.group, async_op=async_op)
        self._work[slot] = w if async_op else None
        return self._gathered[slot]

    # -- VecEnv protocol on GLOBAL arrays
    def reset(self):
        obs = self.local.reset()
        self._wait(0)
        self._packed[0].zero_()
        self._packed[0][:, :OBS_DIM] = obs
        return self._split(self._gather_packed(0))[0]

    def step(self, actions):
        """actions: (N,21) global (every rank passes the same tensor) or (N/G,21) already local."""
        a = actions
        if a.shape[0] == self.num_envs and self.world > 1:
            a = a[self.local_slice()]
        assert a.shape == (self.n_local, ACT_DIM)
        self._wait(0)
        self.local.step_packed(self._packed[0], actions=a)
        gobs, grew, gdone = self._split(self._gather_packed(0))
        return gobs, grew, gdone, self.local._info_tensors()   # info stays rank-local (episode stats reduce separately)

    def rollout_random(self, num_steps, t0=0, gather=True):
        """Benchmark path: each of num_steps steps = one local kernel launch writing the packed block + (when
        gather) one asynchronous all-gather of it, overlapped with the following steps' kernels."""
        slot = 0
        for k in range(num_steps):
            slot = k % DEPTH
            if slot == 0:                          # buffers come round again: all DEPTH collectives must be done.
                for i in range(DEPTH):             # Waits are issued as one batch so that the DEPTH kernels between
                    self._wait(i)                  # two batches are dispatched back to back
            self.local.step_packed(self._packed[slot], actions=None, t=t0 + k)
            if gather:
                self._gather_packed(slot, async_op=True)
        for i in range(DEPTH):
            self._wait(i)
        g = self._gathered[slot] if (gather and self._collective) else self._packed[slot]
        return self._split(g)

    def update_curriculum(self, c):
        self.local.update_curriculum(c)

    def update_specialist(self, c):
        self.local.update_specialist(c)

    def update_sample_prob(self, probs):
        import numpy as np
        probs = np.asarray(probs)
        if probs.ndim == 3 and probs.shape[0] == self.num_envs and self.world > 1:
            probs = probs[self.local_slice()]
        self.local.update_sample_prob(probs)

    def set_mirror(self, m):
        self.local.set_mirror(m)

    def close(self):
        self.local.close()
