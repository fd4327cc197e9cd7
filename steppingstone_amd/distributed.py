"""Multi-GPU sharding of the vectorised env: one process per GPU (torchrun or steppingstone_amd.launch), contiguous
blocks of N/G envs per rank, rank-local state never moves.  The reference's only parallelism is one OS process per env
behind pipes (common/envs_utils.py:519-538); here the single exchange per step is an all-gather (RCCL over xGMI when the
backend is "nccl") so every rank holds the (N,60) obs and (N,) rew / done the single-learner loop of
playground/train.py:363-469 expects.  Actions flow the other way by slicing.  Global env ids (env_id_offset =
rank * N/G) key the RNG streams, so results do not depend on G.

Two exchange layouts, both written directly by the step kernel (no pack / copy kernels):
  * rollout_random (benchmark, BASELINE configs[3]): the packed [N/G, 62] f32 block obs | rew | done;
  * step (learner-facing): one flat buffer per rank = packed block + the [N/G, 6] info words (ep_ret, ep_len,
    bad_transition, steps_reached, update_terrain, ep_ret_lo) behind it, gathered in ONE collective, so the info dict is global
    like obs / rew / done (what collect() of steppingstone_amd.ppo needs for its masks and episode statistics).
"""
import os

import torch
import torch.distributed as dist

from ._lib import ACT_DIM, NCELL, OBS_DIM

PACK = OBS_DIM + 2
INFO = 6            # _lib.INFO_WORDS
DEPTH = 8
CHUNK = 32          # control steps per launch / per collective of the chunked rollout exchange


class ShardedVecEnv:
    """Wraps this rank's local vec env (any object with .step / .reset / .num_envs returning device tensors)."""

    def __init__(self, local_env, group=None, peer_gather=None):
        self.local = local_env
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # SS_FORCE_COLLECTIVE=1 issues the all-gather even at world size 1 (exercises the RCCL path on a 1-GPU box)
        self._collective = self.world > 1 or (dist.is_initialized() and os.environ.get("SS_FORCE_COLLECTIVE") == "1")
        self.n_local = int(local_env.num_envs)
        self.device = torch.device(local_env.device)
        if self._collective and self.device.type == "cuda":
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
        dev = self.device
        # ring of DEPTH buffers: the all-gather of step t runs under the kernels of steps t+1.. and the launch stream
        # waits on the collectives once per DEPTH steps, as one batch.  Measured on one rank with the collective forced:
        # a cross-stream wait before every kernel (any depth) costs +26 us per 0.11 ms step, batched waits every 8
        # steps +7 us; a hipGraph of the 8 steps is slower (+13 us) (tools/gather_overhead.py, gather_graph_probe.py).
        self._packed = [torch.zeros((self.n_local, PACK), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
        self._gathered = [torch.zeros((self.num_envs, PACK), dtype=torch.float32, device=dev) for _ in range(DEPTH)]
        self._work = [None] * DEPTH
        self._last = None           # (kind, buffer index, steps) of the last benchmark exchange, for verify_last_exchange()
        # learner-facing step: packed block and info words in one flat buffer, one collective
        self._chunk = self.n_local * (PACK + INFO)
        self._flat = torch.zeros(self._chunk, dtype=torch.float32, device=dev)
        self._flat_all = torch.zeros(self._chunk * self.world, dtype=torch.float32, device=dev)
        self._flat_packed = self._flat[:self.n_local * PACK].view(self.n_local, PACK)
        self._flat_info = self._flat[self.n_local * PACK:].view(torch.int32).view(self.n_local, INFO)
        # peer-store exchange instead of the RCCL all-gather (steppingstone_amd/peer.py): opt-in, one node, GPUs only
        if peer_gather is None:
            peer_gather = os.environ.get("SS_PEER_GATHER") == "1"
        self._peer = None
        if peer_gather and self._collective and self.device.type == "cuda":
            from .peer import PeerGather
            self._peer = PeerGather.connect_processes(local_env, group)
            self._info_all = torch.zeros((self.num_envs, INFO), dtype=torch.int32, device=dev)
        self._peer_steps = 0

    PEER_CHECK_EVERY = 32

    def _check_peer(self, force=False):
        """The peer-store wait kernel gives up after a bounded spin (about a second) and sets an error word instead of
        hanging the GPU; rows of a rank that was later than that are then stale.  The consumers must not train on them:
        the error word is read (one host synchronisation) every PEER_CHECK_EVERY learner-facing steps -- a rollout length,
        where the training loop synchronises anyway -- and at the end of every benchmark rollout, and a timeout raises.  Rank
        skew (first-call lazy initialisation, rank-0 checkpoints / evaluation, host stalls) must stay under the timeout, or
        the exchange must be the default RCCL all-gather."""
        if self._peer is None:
            return
        self._peer_steps += 1
        if force or self._peer_steps % self.PEER_CHECK_EVERY == 0:
            n = self._peer.error()
            if n:
                raise RuntimeError("peer-store all-gather: %d wait(s) timed out -- a rank was more than the spin bound late, the "
                                   "gathered rows of that step are stale; use the RCCL exchange (peer_gather=False) or remove the "
                                   "rank skew" % n)

    def check_exchange(self, force=True):
        """Raise if the peer-store exchange lost a step (a wait timed out).  A consumer calls this at the end of every rollout, before
        it trains on the gathered rows (ppo.collect does; ADVICE r3: the internal every-32-steps poll is not aligned with rollouts
        of another length).  No-op with the RCCL exchange, whose collectives cannot deliver stale rows silently."""
        self._check_peer(force=force)

    # -- self-proof of the exchange
    def verify_last_exchange(self):
        """Collective.  Checks that the buffers of the LAST benchmark exchange (rollout_random / rollout_random_chunked with
        gather) really hold every rank's block: each rank checksums (exact int64 sum of the bit patterns) its own packed block
        and the block of every peer as it received it; the own checksums are all-gathered and compared with the received
        ones on every rank; the verdict is all-reduced (MIN).  Returns (verified, ranks)."""
        if not self._collective or self._last is None:
            return None, self.world
        kind, idx, ns = self._last
        if kind == "chunk":
            own = self._chunk_local[idx][:ns]
            got = self._chunk_all[idx].view(self.world, -1, self.n_local, PACK)[:, :ns]
        else:
            own = self._packed[idx]
            got = self._gathered[idx].view(self.world, self.n_local, PACK)
        cs = lambda x: x.contiguous().view(torch.int32).to(torch.int64).sum().reshape(1)          # noqa: E731
        mine = cs(own)
        seen = torch.cat([cs(got[r]) for r in range(self.world)])
        alls = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(alls, mine, group=self.group)
        ok = (seen == alls).all().to(torch.int32).reshape(1)
        # a rank whose own block is all zeros (nothing was stepped) proves nothing
        ok = ok * (own.abs().sum() > 0).to(torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        return bool(ok.item()), self.world

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

    def _gather_flat(self):
        """One collective for packed block + info words; returns global (packed [N,62], info [N,6] int32)."""
        if not self._collective:
            return self._flat_packed, self._flat_info
        dist.all_gather_into_tensor(self._flat_all, self._flat, group=self.group)
        per_rank = self._flat_all.view(self.world, self._chunk)
        packed = per_rank[:, :self.n_local * PACK].reshape(self.num_envs, PACK)
        info = per_rank[:, self.n_local * PACK:].reshape(self.num_envs * INFO).view(torch.int32).view(self.num_envs, INFO)
        return packed, info

    @staticmethod
    def _info_dict(info):
        fl = info.view(torch.float32)
        return {"ep_ret": fl[:, 0], "ep_len": fl[:, 1], "bad_transition": info[:, 2], "steps_reached": info[:, 3],
                "update_terrain": info[:, 4], "ep_ret_lo": fl[:, 5]}

    # -- VecEnv protocol on GLOBAL arrays
    def reset(self):
        obs = self.local.reset()
        self._flat.zero_()
        self._flat_packed[:, :OBS_DIM] = obs
        packed, _ = self._gather_flat()
        return self._split(packed)[0]

    def step(self, actions):
        """actions: (N,21) global (every rank passes the same tensor) or (N/G,21) already local.  Returns GLOBAL obs
        (N,60), rew (N,), done (N,) and a GLOBAL info dict of (N,) tensors."""
        a = actions
        if a.shape[0] == self.num_envs and self.world > 1:
            a = a[self.local_slice()]
        assert a.shape == (self.n_local, ACT_DIM)
        if self._peer is not None:
            # kernel -> every peer's gather buffer; the five info words per env follow in one small collective
            slot = self._peer.step(actions=a, info=self._flat_info)
            dist.all_gather_into_tensor(self._info_all, self._flat_info, group=self.group)
            gobs, grew, gdone = self._split(self._peer.wait(slot))
            self._check_peer()
            return gobs, grew, gdone, self._info_dict(self._info_all)
        self.local.step_packed(self._flat_packed, actions=a, info=self._flat_info)
        packed, info = self._gather_flat()
        gobs, grew, gdone = self._split(packed)
        return gobs, grew, gdone, self._info_dict(info)

    def rollout_random_chunked(self, num_steps, t0=0, gather=True, chunk=CHUNK):
        """Benchmark path, K steps per launch: each chunk of `chunk` control steps is ONE launch of the rollout kernel (state
        resident in LDS between the steps) writing every step's packed block into a [chunk, N/G, 62] buffer, followed by ONE
        asynchronous all-gather of the whole chunk -- the same bytes per step on the wire as the per-step exchange, `chunk`
        times fewer launches and collectives; the collective of chunk c runs under the kernel of chunk c+1 (two buffers).
        Nothing consumes the observations between the steps of a random-action rollout, so every rank still ends up with every
        step's global [N, 62] block.  Returns the last step's global (obs, rew, done)."""
        dev = self.device
        if getattr(self, "_chunk_local", None) is None or self._chunk_local[0].shape[0] != chunk:
            self._chunk_local = [torch.zeros((chunk, self.n_local, PACK), dtype=torch.float32, device=dev) for _ in range(2)]
            self._chunk_all = [torch.zeros((self.world * chunk, self.n_local, PACK), dtype=torch.float32, device=dev) for _ in range(2)]
            self._chunk_work = [None, None]
        done_steps, c, last = 0, 0, None
        while done_steps < num_steps:
            ns = min(chunk, num_steps - done_steps)
            b = c & 1
            if self._chunk_work[b] is not None:            # the collective that last used this pair of buffers
                self._chunk_work[b].wait()
                self._chunk_work[b] = None
            loc = self._chunk_local[b]
            self.local.rollout_random_packed(loc[:ns], t0=t0 + done_steps)
            if gather and self._collective:
                self._chunk_work[b] = dist.all_gather_into_tensor(self._chunk_all[b], loc, group=self.group, async_op=True)
            last = (b, ns)
            done_steps += ns
            c += 1
        for b in range(2):
            if self._chunk_work[b] is not None:
                self._chunk_work[b].wait()
                self._chunk_work[b] = None
        b, ns = last
        self._last = ("chunk", b, ns) if (gather and self._collective) else None
        if gather and self._collective:
            g = self._chunk_all[b].view(self.world, chunk, self.n_local, PACK)[:, ns - 1].reshape(self.num_envs, PACK)
        else:
            g = self._chunk_local[b][ns - 1]
        return self._split(g)

    def rollout_random(self, num_steps, t0=0, gather=True):
        """Benchmark path, one launch per step: each of num_steps steps = one local kernel launch writing the packed block +
        (when gather) one asynchronous all-gather of it, overlapped with the following steps' kernels."""
        if self._peer is not None and gather:
            # In-order on the launch stream: [step t] [wait t] [step t+1] ...  Seeing peer p's flag of step t+1 implies p
            # is past its wait of step t, so writing step t+2 into the slot of step t (ring of 2) cannot race p's reads.
            g = None
            for k in range(num_steps):
                g = self._peer.wait(self._peer.step(actions=None, t=t0 + k))
            self._check_peer(force=True)
            return self._split(g)
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
        self._last = ("step", slot, 1) if (gather and self._collective and num_steps > 0) else None
        g = self._gathered[slot] if (gather and self._collective) else self._packed[slot]
        return self._split(g)

    # -- hooks: every rank applies the same value (broadcast-free)
    def update_curriculum(self, c):
        self.local.update_curriculum(c)

    def update_specialist(self, c):
        self.local.update_specialist(c)

    def update_sample_prob(self, probs):
        """(11,11) shared grid, or (N,11,11) GLOBAL per-env grids (this rank takes its block)."""
        if hasattr(probs, "shape") and len(probs.shape) == 3 and probs.shape[0] == self.num_envs and self.world > 1:
            probs = probs[self.local_slice()]
        self.local.update_sample_prob(probs)

    def set_mirror(self, m):
        self.local.set_mirror(m)

    def set_env_params(self, params):
        self.local.set_env_params(params)

    def set_robot_params(self, params):
        self.local.set_robot_params(params)

    def get_mirror_indices(self):
        return self.local.get_mirror_indices()

    def create_temp_states(self):
        """(N,121,60): every rank's block gathered (119 MB per 4096 envs: the sampler calls this on small eval batches)."""
        loc = self.local.create_temp_states()
        if not self._collective:
            return loc
        out = torch.empty((self.num_envs, NCELL, OBS_DIM), dtype=loc.dtype, device=loc.device)
        dist.all_gather_into_tensor(out, loc.contiguous(), group=self.group)
        return out

    def close(self):
        if self._peer is not None:
            self._peer.close()
            self._peer = None
        self.local.close()
