"""Multi-GPU sharding of the vectorised env: one process per GPU (torchrun), contiguous blocks of N/G envs per
rank, rank-local state never moves.  The reference's only parallelism is one OS process per env behind pipes
(common/envs_utils.py:519-538); here the single exchange per step is an all-gather (RCCL over xGMI when the
backend is "nccl") of the packed [N/G, 62] = [obs | rew | done] block so every rank holds the (N,60) obs and (N,)
rew / done the single-learner loop of playground/train.py:363-469 expects.  Actions flow the other way by slicing.
Global env ids (env_id_offset = rank * N/G) key the RNG streams, so results do not depend on G.
"""
import torch
import torch.distributed as dist

from ._lib import ACT_DIM, OBS_DIM

PACK = OBS_DIM + 2


class ShardedVecEnv:
    """Wraps this rank's local vec env (any object with .step / .reset / .num_envs returning device tensors)."""

    def __init__(self, local_env, group=None):
        self.local = local_env
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_local = int(local_env.num_envs)
        self.num_envs = self.n_local * self.world
        self.observation_space = getattr(local_env, "observation_space", None)
        self.action_space = getattr(local_env, "action_space", None)
        dev = local_env.device
        self._packed = torch.zeros((self.n_local, PACK), dtype=torch.float32, device=dev)
        self._gathered = torch.zeros((self.num_envs, PACK), dtype=torch.float32, device=dev)

    # -- helpers
    def local_slice(self):
        return slice(self.rank * self.n_local, (self.rank + 1) * self.n_local)

    def _gather(self, obs, rew, done):
        self._packed[:, :OBS_DIM] = obs
        self._packed[:, OBS_DIM] = rew
        self._packed[:, OBS_DIM + 1] = done.to(torch.float32)
        if self.world > 1:
            dist.all_gather_into_tensor(self._gathered, self._packed, group=self.group)
            g = self._gathered
        else:
            g = self._packed
        return g[:, :OBS_DIM], g[:, OBS_DIM], g[:, OBS_DIM + 1] > 0.5

    # -- VecEnv protocol on GLOBAL arrays
    def reset(self):
        obs = self.local.reset()
        zero = torch.zeros(self.n_local, dtype=torch.float32, device=obs.device)
        return self._gather(obs, zero, zero)[0]

    def step(self, actions):
        """actions: (N,21) global (every rank passes the same tensor) or (N/G,21) already local."""
        a = actions
        if a.shape[0] == self.num_envs and self.world > 1:
            a = a[self.local_slice()]
        assert a.shape == (self.n_local, ACT_DIM)
        obs, rew, done, info = self.local.step(a)
        gobs, grew, gdone = self._gather(obs, rew, done)
        return gobs, grew, gdone, info          # info stays rank-local (episode stats reduce separately)

    def rollout_random(self, num_steps, t0=0, gather=True):
        """Benchmark path: each of num_steps steps = one local kernel launch (+ one all-gather when gather)."""
        out = None
        for k in range(num_steps):
            obs, rew, done = self.local.rollout_random(1, t0 + k)
            if gather:
                out = self._gather(obs, rew, done)
            else:
                out = (obs, rew, done)
        return out

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
