"""Device-resident PPO driver for the vectorised stepping-stone env.

Counterpart of the reference's training loop, re-organised so that nothing leaves the GPU between env.step() and the
learner:
  * actor / critic-ensemble networks     common/controller.py:55-145,217-261   (same architecture and init)
  * rollout storage + GAE                algorithms/storage.py:5-82            (tensors on the env's device)
  * clipped-surrogate update             algorithms/ppo.py:40-108              (same loss, Adam, grad clip)
  * rollout / curriculum / LR schedule   playground/train.py:211-222,363-469,503-506
The per-env Python loops of train.py:446-456 (bad_masks, episode rewards, masks) are tensor ops here.  The loss and
one optimiser step are pinned against the reference's own PPO.update (tests/test_ppo_golden.py).
"""
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import harness
from ._lib import ACT_DIM, OBS_DIM


class Actor(nn.Module):
    """SoftsignActor (common/controller.py:217-261): 60 -> 256 x5 -> 21, softsign x3, relu x2, tanh."""

    def __init__(self, state_dim=OBS_DIM, action_dim=ACT_DIM, h_size=256):
        super().__init__()
        self.state_dim, self.action_dim = state_dim, action_dim
        self.fc1 = nn.Linear(state_dim, h_size)
        self.fc2 = nn.Linear(h_size, h_size)
        self.fc3 = nn.Linear(h_size, h_size)
        self.fc4 = nn.Linear(h_size, h_size)
        self.fc5 = nn.Linear(h_size, h_size)
        self.out = nn.Linear(h_size, action_dim)

    def forward(self, x):
        x = F.softsign(self.fc1(x))
        x = F.softsign(self.fc2(x))
        x = F.softsign(self.fc3(x))
        x = F.relu(self.fc4(x))
        x = F.relu(self.fc5(x))
        return torch.tanh(self.out(x))


def _critic(state_dim, h_size=256):
    gain = nn.init.calculate_gain("relu")
    layers, d = [], state_dim
    for width in (h_size, h_size, h_size, h_size, 1):
        lin = nn.Linear(d, width)
        nn.init.orthogonal_(lin.weight.data, gain=gain)
        nn.init.constant_(lin.bias.data, 0)
        layers += [lin, nn.ReLU()]
        d = width
    return nn.Sequential(*layers[:-1])


class ActorCritic(nn.Module):
    """Policy (common/controller.py:55-145): tanh-mean diagonal Gaussian with a state-independent log-std (init -1.5)
    and an ensemble of value networks whose mean is the value estimate."""

    def __init__(self, state_dim=OBS_DIM, action_dim=ACT_DIM, num_ensembles=1, noise=-1.5):
        super().__init__()
        self.actor = Actor(state_dim, action_dim)
        self.logstd = nn.Parameter(torch.full((action_dim,), float(noise)))
        self.critics = nn.ModuleList([_critic(state_dim) for _ in range(num_ensembles)])

    def reset_dist(self):
        self.logstd.data.fill_(-2.5)

    def get_ensemble_values(self, obs):
        return torch.cat([c(obs) for c in self.critics], dim=-1)

    def get_value(self, obs):
        return self.get_ensemble_values(obs).mean(dim=-1, keepdim=True)

    def _logp(self, mean, action):
        var = (2 * self.logstd).exp()
        return (-((action - mean) ** 2) / (2 * var) - self.logstd - 0.5 * math.log(2 * math.pi)).sum(-1, keepdim=True)

    def act(self, obs, deterministic=False):
        mean = self.actor(obs)
        action = mean if deterministic else mean + self.logstd.exp() * torch.randn_like(mean)
        return self.get_value(obs), action, self._logp(mean, action)

    def evaluate_actions(self, obs, action):
        mean = self.actor(obs)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + self.logstd).sum()      # per-sample entropy is constant
        return self.get_ensemble_values(obs), self._logp(mean, action), entropy


def ppo_loss(ac, obs, act, value_preds, returns, old_logp, adv, clip_param=0.2, use_clipped_value_loss=False):
    """The three loss terms of algorithms/ppo.py:64-85 for one minibatch."""
    values, logp, entropy = ac.evaluate_actions(obs, act)
    ratio = torch.exp(logp - old_logp)
    surr1 = ratio * adv
    surr2 = torch.clamp(ratio, 1.0 - clip_param, 1.0 + clip_param) * adv
    action_loss = -torch.min(surr1, surr2).mean()
    if use_clipped_value_loss:
        clipped = value_preds + (values - value_preds).clamp(-clip_param, clip_param)
        value_loss = 0.5 * torch.max((values - returns).pow(2), (clipped - returns).pow(2)).mean()
    else:
        value_loss = 0.5 * (returns - values).pow(2).mean()
    return value_loss, action_loss, entropy


class PPO:
    """algorithms/ppo.py:6-108 on device tensors (defaults of playground/train.py:72-82)."""

    def __init__(self, ac, clip_param=0.2, ppo_epoch=10, mini_batch_size=1024, value_loss_coef=1.0, entropy_coef=0.0,
                 lr=3e-4, eps=1e-5, max_grad_norm=2.0, use_clipped_value_loss=False, mirror_indices=None,
                 use_graph=False, graph_collectives=None, force_collective=False):
        self.ac = ac
        self.clip_param, self.ppo_epoch, self.mini_batch_size = clip_param, ppo_epoch, mini_batch_size
        self.value_loss_coef, self.entropy_coef, self.max_grad_norm = value_loss_coef, entropy_coef, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        dev = next(ac.parameters()).device
        # index lists on the parameters' device once: no host-to-device copies inside the (capturable) minibatch step
        self.mirror_indices = None if mirror_indices is None else [
            torch.as_tensor(i, dtype=torch.long, device=dev) for i in mirror_indices]
        # use_graph: the minibatch step (gather, forward, backward, clip, Adam) is ~90 small launches; captured once
        # in a hipGraph it replays as one submission.  Needs CUDA/HIP parameters and a single rank.
        self.use_graph = bool(use_graph) and dev.type == "cuda"
        if self.use_graph:
            self.optimizer = torch.optim.Adam(ac.parameters(), lr=torch.tensor(float(lr), device=dev), eps=eps, capturable=True)
        else:
            self.optimizer = torch.optim.Adam(ac.parameters(), lr=lr, eps=eps)
        self._graph, self._warm, self._static = None, 0, None
        # Data-parallel learner (configs[4], one process per GPU): the minibatch step contains ONE RCCL all-reduce of the flat
        # gradient.  graph_collectives (default: env SS_GRAPH_COLLECTIVES, "1" = on): capture that step -- collective included -- in
        # a hipGraph as on one rank (RCCL supports stream capture; torch's ProcessGroupNCCL joins its stream to the capturing one).
        # The capture is attempted once, after three eager warm-up steps (communicator set up); if it raises, the step stays eager
        # and says so.  force_collective: issue the all-reduce at world size 1 too (how a one-GPU box tests the captured collective).
        import os
        self.graph_collectives = (os.environ.get("SS_GRAPH_COLLECTIVES", "1") == "1") if graph_collectives is None else bool(graph_collectives)
        self.force_collective = bool(force_collective)
        self.graph_fallback = None          # the exception text if a capture was attempted and abandoned

    def _allreduce_grads(self):
        """Data-parallel learner: one RCCL all-reduce of the flattened gradient per minibatch (no-op on one rank)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not self.force_collective):
            return
        grads = [p.grad for p in self.ac.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat /= dist.get_world_size()
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()

    def set_lr(self, lr):
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g["lr"]):
                g["lr"].fill_(float(lr))
            else:
                g["lr"] = lr

    def step_minibatch(self, obs, act, value_preds, returns, old_logp, adv):
        if self.mirror_indices is not None:
            obs, act = harness.mirror_batch(obs, act, self.mirror_indices)
            value_preds, returns, old_logp, adv = (t.repeat((2, 1)) for t in (value_preds, returns, old_logp, adv))
        vl, al, ent = ppo_loss(self.ac, obs, act, value_preds, returns, old_logp, adv, self.clip_param,
                               self.use_clipped_value_loss)
        self.optimizer.zero_grad()
        (vl * self.value_loss_coef + al - ent * self.entropy_coef).backward()
        self._allreduce_grads()
        nn.utils.clip_grad_norm_(self.ac.parameters(), self.max_grad_norm)
        self.optimizer.step()
        return vl.detach(), al.detach(), ent.detach()

    # -- hipGraph path ------------------------------------------------------------------------------------------------
    def _graph_ok(self):
        import torch.distributed as dist
        if not self.use_graph or self.graph_fallback is not None:
            return False
        multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force_collective)
        # several ranks: only a backend whose collectives run on a stream can be captured (RCCL; gloo's are host calls)
        return (not multi) or (self.graph_collectives and dist.get_backend() == "nccl")

    def _gathered_step(self, data, idx):
        return self.step_minibatch(*(t[idx] for t in data))

    def _graph_step(self, data, idx, refresh=True):
        """data: the six flat rollout tensors of this update (copied into static storage when refresh is set, i.e.
        once per update); idx: minibatch indices."""
        if self._static is None:
            self._static = (tuple(torch.empty_like(t) for t in data), torch.zeros_like(idx), torch.zeros(3, device=idx.device))
            refresh = True
        if self.graph_fallback is not None:        # capture was abandoned once: never retried, straight to the eager step
            return torch.stack(self._gathered_step(data, idx))
        sdata, sidx, sout = self._static
        if refresh:
            for dst, src in zip(sdata, data):
                dst.copy_(src)
        if self._graph is None:
            self._warm += 1
            if self._warm <= 3:                         # eager warm-up steps on a side stream (real data, real steps)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    out = self._gathered_step(data, idx)
                torch.cuda.current_stream().wait_stream(side)
                return torch.stack(out)
            self.optimizer.zero_grad(set_to_none=True)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    out = self._gathered_step(sdata, sidx)
                    sout.copy_(torch.stack(out))
            except Exception as exc:          # (a collective that cannot be captured in this stack: stay eager, loudly)
                self.graph_fallback = repr(exc)[:300]
                import warnings
                warnings.warn("PPO minibatch step: hipGraph capture abandoned, running eagerly (%s)" % self.graph_fallback)
                torch.cuda.synchronize()
                return torch.stack(self._gathered_step(data, idx))
            self._graph = g
        sidx.copy_(idx)
        self._graph.replay()
        return sout.clone()

    def update(self, roll):
        adv = roll.returns[:-1] - roll.value_preds[:-1]
        mean, std = _global_mean_std(adv)          # over ALL ranks' transitions (= the reference's statistics on one rank)
        adv = (adv - mean) / (std + 1e-5)
        T, N = roll.rewards.shape[:2]
        flat = lambda t: t.reshape(T * N, -1)   # noqa: E731
        data = (flat(roll.obs[:-1]), flat(roll.actions), flat(roll.value_preds[:-1]), flat(roll.returns[:-1]),
                flat(roll.logp), flat(adv))
        dev = data[0].device
        graph = self._graph_ok() and (T * N) % self.mini_batch_size == 0
        stats = torch.zeros(3, device=dev)
        count = 0
        for _ in range(self.ppo_epoch):
            perm = torch.randperm(T * N, device=dev)
            for s in range(0, T * N, self.mini_batch_size):
                idx = perm[s:s + self.mini_batch_size]
                if graph:
                    stats += self._graph_step(data, idx, refresh=(count == 0))
                else:
                    stats += torch.stack(self._gathered_step(data, idx))
                count += 1
        return (stats / max(count, 1)).tolist()


def _dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _global_mean_std(x):
    """mean and unbiased std of x over every rank's elements (algorithms/ppo.py:42-43 computes them on the single
    process's whole batch; a data-parallel learner must use the same global statistics)."""
    if _dist_world() == 1:
        return x.mean(), x.std()
    import torch.distributed as dist
    x64 = x.double()
    s = torch.stack([x64.sum(), (x64 * x64).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
    dist.all_reduce(s)
    n = s[2]
    mean = s[0] / n
    var = (s[1] - n * mean * mean) / (n - 1)
    return mean.to(x.dtype), var.clamp_min(0).sqrt().to(x.dtype)


class EpisodeRing:
    """The reference's `episode_rewards = deque(maxlen=num_processes)` (playground/train.py:193,454-456) as a device
    ring buffer written with capturable tensor ops only (no host sync, no data-dependent shapes): finished episodes'
    returns go to consecutive slots, everything else to a dump slot.  Eager and hipGraph rollouts therefore gate the
    curriculum on the SAME statistic (mean of the last `size` episode returns)."""

    def __init__(self, size, device):
        self.size = int(size)
        self.buf = torch.zeros(self.size + 1, device=device)          # [size] = dump slot
        self.ptr = torch.zeros((), dtype=torch.long, device=device)
        self.count = torch.zeros((), dtype=torch.long, device=device)

    def push(self, ep_ret, done):
        d = done.reshape(-1).to(torch.bool)
        k = torch.cumsum(d.to(torch.long), 0) - 1
        idx = torch.where(d, (self.ptr + k) % self.size, torch.full_like(k, self.size))
        self.buf.scatter_(0, idx, torch.where(d, ep_ret.reshape(-1).to(self.buf.dtype), self.buf[self.size].expand_as(k)))
        nd = d.sum()
        self.ptr.copy_((self.ptr + nd) % self.size)
        self.count.copy_(torch.clamp(self.count + nd, max=self.size))

    def values(self):
        """Host copy of the stored returns (one synchronisation), oldest-agnostic order."""
        c = int(self.count)
        if c == 0:
            return torch.empty(0)
        b = self.buf[:self.size].cpu()
        return b if c == self.size else b[:c]       # before the first wrap the filled slots are 0..count-1

    def all_ranks_values(self):
        """Host copy of the stored returns of EVERY rank's ring (collective under torch.distributed): what the CSV row of a
        multi-rank run reports, so that it is the same statistic that gates the curriculum."""
        if _dist_world() <= 1:
            return self.values()
        import torch.distributed as dist
        w = _dist_world()
        flat = torch.zeros(w * self.size, device=self.buf.device, dtype=self.buf.dtype)
        cnts = torch.zeros(w, dtype=torch.long, device=self.buf.device)
        dist.all_gather_into_tensor(flat, self.buf[:self.size].contiguous())
        dist.all_gather_into_tensor(cnts, self.count.reshape(1).contiguous())
        bufs, cnts = flat.view(w, self.size).cpu(), cnts.cpu().tolist()
        parts = [bufs[r] if c == self.size else bufs[r, :c] for r, c in enumerate(cnts) if c]
        return torch.cat(parts) if parts else torch.empty(0)

    def all_ranks_sum_count(self):
        """(sum, count) over every rank's ring: the curriculum gate must be one decision for all ranks."""
        v = torch.stack([self.buf[:self.size].sum().double(), self.count.double()])
        if _dist_world() > 1:
            import torch.distributed as dist
            dist.all_reduce(v)
        s, c = v.tolist()
        return s, int(c)


class Rollouts:
    """RolloutStorage (algorithms/storage.py:5-57) on the env's device."""

    def __init__(self, num_steps, num_envs, device):
        T, N = num_steps, num_envs
        z = lambda *s: torch.zeros(*s, device=device)   # noqa: E731
        self.obs, self.actions = z(T + 1, N, OBS_DIM), z(T, N, ACT_DIM)
        self.rewards, self.logp = z(T, N, 1), z(T, N, 1)
        self.value_preds, self.returns = z(T + 1, N, 1), z(T + 1, N, 1)
        self.masks, self.bad_masks = torch.ones(T + 1, N, 1, device=device), torch.ones(T + 1, N, 1, device=device)
        self.step = 0

    def insert(self, obs, action, logp, value, reward, mask, bad_mask):
        t = self.step
        self.obs[t + 1].copy_(obs); self.actions[t].copy_(action); self.logp[t].copy_(logp)
        self.value_preds[t].copy_(value); self.rewards[t].copy_(reward)
        self.masks[t + 1].copy_(mask); self.bad_masks[t + 1].copy_(bad_mask)
        self.step = (t + 1) % self.rewards.shape[0]

    def after_update(self):
        self.obs[0].copy_(self.obs[-1]); self.masks[0].copy_(self.masks[-1]); self.bad_masks[0].copy_(self.bad_masks[-1])

    def compute_returns(self, next_value, use_gae=True, gamma=0.99, gae_lambda=0.95):
        self.returns = harness.compute_returns(self.rewards, self.value_preds, self.masks, self.bad_masks, next_value,
                                               use_gae, gamma, gae_lambda)
        if use_gae:
            self.value_preds[-1] = next_value


def collect(envs, ac, roll, num_steps, ep_returns=None, ep_stats=None, ring=None, deterministic=False):
    """The rollout loop of playground/train.py:363-469 with tensorised bookkeeping.  `envs` returns device tensors with
    GLOBAL shapes equal to roll's env dimension (SteppingStoneVecEnv(return_numpy=False), or ShardedVecEnv whose step()
    all-gathers obs / rew / done AND the info words).  Finished episodes are reported through `ring` (EpisodeRing, the
    reference's deque; capturable), and optionally through ep_stats, a device tensor [2] accumulating (sum of returns,
    count), or the host list ep_returns (one host sync per step)."""
    for _ in range(num_steps):
        with torch.no_grad():
            value, action, logp = ac.act(roll.obs[roll.step], deterministic=deterministic)
        obs, rew, done, info = envs.step(action)
        d = done.to(torch.float32).unsqueeze(1)
        mask = 1.0 - d
        bad_mask = 1.0 - info["bad_transition"].to(torch.float32).unsqueeze(1)
        if ring is not None:
            ring.push(info["ep_ret"], done)
        if ep_stats is not None:
            ep_stats[0] += (info["ep_ret"] * d[:, 0]).sum()
            ep_stats[1] += d.sum()
        elif ep_returns is not None and bool(done.any()):
            ep_returns.append(info["ep_ret"][done].clone())
        roll.insert(obs, action, logp, value, rew.unsqueeze(1), mask, bad_mask)
    capturing = roll.obs.is_cuda and torch.cuda.is_current_stream_capturing()
    if hasattr(envs, "check_exchange") and not capturing:
        envs.check_exchange()          # a multi-GPU exchange that lost a step must not be trained on (ShardedVecEnv, peer stores)


class GraphedCollector:
    """collect() captured once in a hipGraph (policy inference, env step kernel, storage writes of all num_steps
    steps: ~50 launches per step) and replayed per update.  The rollout storage, the env's I/O buffers and the
    episode accumulators are static, and every call starts at roll.step == 0, so all addresses are replay-stable."""

    def __init__(self, envs, ac, roll, num_steps, ring=None):
        self.envs, self.ac, self.roll, self.num_steps, self.ring = envs, ac, roll, num_steps, ring
        self.ep_stats = torch.zeros(2, device=roll.obs.device)
        self.graph, self.warm = None, 0

    def __call__(self):
        assert self.roll.step == 0 and self.num_steps == self.roll.rewards.shape[0]
        self.ep_stats.zero_()
        if self.graph is None:
            if self.warm < 1:                              # one eager rollout on a side stream first
                self.warm += 1
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    collect(self.envs, self.ac, self.roll, self.num_steps, ep_stats=self.ep_stats, ring=self.ring)
                torch.cuda.current_stream().wait_stream(side)
                return self.ep_stats
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                collect(self.envs, self.ac, self.roll, self.num_steps, ep_stats=self.ep_stats, ring=self.ring)
            self.graph = g
        self.graph.replay()
        return self.ep_stats


def sampling_probs_from_values(ac, eval_envs, mode="threshold", curriculum_threshold=0.85, events=5, max_steps=400,
                               as_tensor=False):
    """Adaptive / threshold curriculum sampler (playground/train.py:229-272, 320-361), batched: the evaluation envs
    are rolled with the deterministic policy (auto-reset on, like the reference's `if done: reset`); every env that
    advances to a new target contributes the critic-ensemble mean value of its 121 hypothetical next stones
    (create_temp_states); the first `events` such contributions (env order within a step) are summed, the sum is
    normalised by its absolute maximum and becomes softmax(-10 |m - threshold|) ("threshold", train.py:262) or
    softmax(-10 m) ("adaptive", train.py:354).  Returns the (11,11) grid as float64 numpy (what update_sample_prob gets in
    the reference) or, as_tensor=True, as a float32 tensor on the env's device (stream-ordered hook, no host copy);
    None if fewer than one event happened within max_steps."""
    dev = eval_envs.device
    obs = eval_envs.reset()
    total = torch.zeros(121, device=dev)
    seen = 0
    for _ in range(max_steps):
        with torch.no_grad():
            _, action, _ = ac.act(obs, deterministic=True)
        obs, _, _, info = eval_envs.step(action)
        hit = info["update_terrain"] > 0
        if bool(hit.any()):
            temp = eval_envs.create_temp_states()[hit][:events - seen]      # [k,121,60], at most the missing events
            with torch.no_grad():
                v = ac.get_ensemble_values(temp.reshape(-1, temp.shape[-1])).mean(dim=-1)
            total += v.view(-1, 121).sum(dim=0)
            seen += temp.shape[0]
            if seen >= events:
                break
    if seen == 0:
        return None
    m = total / total.abs().max()
    logits = -10.0 * (m - curriculum_threshold).abs() if mode == "threshold" else -10.0 * m
    p = torch.softmax(logits, dim=0).view(11, 11)
    return p if as_tensor else p.double().cpu().numpy()


def evaluate(test_envs, ac, max_steps):
    """The deterministic test loop of playground/train.py:472-500: reset the test envs, run the deterministic policy for
    max_steps steps (auto-reset on), collect the returns of the episodes that finish.  Returns a 1-D CPU tensor."""
    obs = test_envs.reset()
    rets = []
    for _ in range(max_steps):
        with torch.no_grad():
            _, action, _ = ac.act(obs, deterministic=True)
        obs, _, done, info = test_envs.step(action)
        rets.append(torch.where(done, info["ep_ret"], torch.full_like(info["ep_ret"], float("nan"))))
    r = torch.stack(rets).reshape(-1).cpu()
    return r[~torch.isnan(r)]


def save_checkpoint(ac, path, **meta):
    """{env}_latest.pt / {env}_best.pt / {env}_{frames}.pt (playground/train.py:523-562).  The reference pickles the
    whole module (its class source travels with the file); here the file holds the CPU state_dict plus the few numbers
    needed to rebuild the module -- torch.load(path)["state_dict"] -> ActorCritic(...).load_state_dict."""
    sd = {k: v.detach().cpu() for k, v in ac.state_dict().items()}
    torch.save({"state_dict": sd, "num_ensembles": len(ac.critics), "state_dim": ac.actor.state_dim,
                "action_dim": ac.actor.action_dim, "policy_convention": policy_convention(), "env_fingerprint": env_fingerprint(), **meta}, path)


def env_fingerprint():
    """SHA-256 over the numbers of the env a policy was trained in (both robots' tables as model.build() produces them + the env
    constants).  Warning-level on load: a policy still means the same joints in a re-identified robot, but it was trained
    for another body."""
    import hashlib
    import numpy as np
    from . import model
    h = hashlib.sha256()
    for kind in ("walker3d", "mike"):
        m = model.build(kind)
        for k in sorted(m):
            if isinstance(m[k], np.ndarray):
                h.update(k.encode() + np.ascontiguousarray(m[k], np.float32).tobytes())
        h.update(repr((round(m["friction"], 6), round(m["stand_height"], 6))).encode())
    h.update(repr(sorted(model.env_constants().items())).encode())
    return h.hexdigest()


def policy_convention():
    """What a trained policy's inputs and outputs MEAN: the env's ABI version and the per-joint sign of the policy coordinates
    (docs/PHYSICS.md 2).  ABI 3 -> 4 flipped the sign of the left limbs' x / z joints and of both knees under the same layout, so a
    file without this stamp (rounds 1-3) or with another one would load silently and drive the robot with flipped joints."""
    from . import model
    from ._lib import ABI_VERSION
    return {"abi_version": int(ABI_VERSION), "policy_sign": [int(s) for s in model.POLICY_SIGN]}


def load_checkpoint(path, device="cpu", allow_convention_mismatch=False):
    """Refuses a file whose policy convention is missing or differs from this build's (INTEGRATION.md "checkpoints"); WARNS when the
    convention matches but the file was trained in an env with other numbers (another `env_fingerprint`, or none: a file written
    before round 6)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    have, want = ck.get("policy_convention"), policy_convention()
    if have != want and not allow_convention_mismatch:
        raise ValueError("%s was trained under policy convention %r, this build presents %r: its actions and observations would be "
                         "misread (pass allow_convention_mismatch=True to load it anyway)" % (path, have, want))
    if ck.get("env_fingerprint") != env_fingerprint():
        import warnings
        warnings.warn("%s was trained in an env with other robot / terrain numbers than this build's (env_fingerprint %s): it loads, but "
                      "expect it to need re-training" % (path, "absent" if ck.get("env_fingerprint") is None else "differs"), RuntimeWarning)
    ac = ActorCritic(ck["state_dim"], ck["action_dim"], num_ensembles=ck["num_ensembles"])
    ac.load_state_dict(ck["state_dict"])
    return ac.to(device), ck


def train(envs, num_updates, num_steps=32, num_ensembles=1, seed=8, use_curriculum=True, use_mirror=False, lr=3e-4,
          gamma=0.99, gae_lambda=0.95, ppo_epoch=10, mini_batch_size=1024, log=print, use_graph="auto",
          sampling="none", eval_envs=None, curriculum_threshold=0.85, uniform_every=500000,
          test_envs=None, test_interval=1, logger=None, save_dir="", save_every=1e7, env_name="env",
          use_specialist=False, agent_factory=None, on_rollout=None):
    """The training loop of playground/train.py:211-578 on device tensors.  Returns (actor_critic, per-update stats).

      use_curriculum   fixed-order curriculum: level += 1 while mean(recent episode returns) > 1000 (train.py:115-118,503-506)
      use_specialist   same gate, ring windows + a `{env}_specialist_{k}.pt` file per level (train.py:119-122,538-545)
      sampling         "threshold" (train.py:123-133,229-272,460-469) / "adaptive" (train.py:134-137,320-361): the grid of
                       the next-next stone is re-estimated EVERY update from the critic ensemble on `eval_envs` (a small
                       batch at curriculum 0; the reference uses one env); "threshold" starts with one uniform update
                       (curriculum 5) and repeats it every `uniform_every` updates
      test_envs        deterministic evaluation every `test_interval` updates for max_episode_steps steps (train.py:472-500;
                       the reference does it every update)
      logger           ConsoleCSVLogger-compatible object (steppingstone_amd.csv_logger): log_epoch(dict) per update
      save_dir         `{env}_latest.pt` every update, `{env}_{frames}.pt` every save_every frames, `{env}_best.pt` on a new
                       best mean return (train.py:523-562)
      on_rollout       optional `f(update_index, rollouts)` called after every collection, before the returns are computed (tests: the
                       rollout a rank trained on is compared with a single-process env replaying its actions)
      agent_factory    None: the learner is steppingstone_amd.ppo.PPO (BASELINE configs[4]: "actor/critic on PyTorch-ROCm": autograd +
                       torch.optim.Adam).  A caller may pass `f(actor_critic, ppo_epoch=, mini_batch_size=, lr=, mirror_indices=, use_graph=)`
                       returning an object with PPO's `update` / `set_lr` contract (the out-of-scope fused learner under
                       extras/fused_learner does; nothing in this package provides one)
    use_graph ("auto": on a GPU with a single rank): rollout and minibatch step replay as hipGraphs.  Episode returns go
    through a device ring of the last num_envs episodes in both modes (EpisodeRing = the reference's deque).
    Under torch.distributed (one rank per GPU) every rank passes its LOCAL envs: gradients and the advantage statistics
    are all-reduced, parameters start identical (broadcast from rank 0), exploration noise is seeded per rank, and the
    curriculum gate uses the all-reduced episode statistics so that every rank takes the same decision."""
    import os
    import torch.distributed as dist
    dev = torch.device(envs.device if hasattr(envs, "device") else envs.local.device)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0
    world = dist.get_world_size() if multi else 1
    # hipGraphs: the rollout (policy + env step + storage writes) on one rank; the minibatch step on one rank AND, with RCCL as the
    # transport, at several ranks -- its single all-reduce is captured with it (PPO.graph_collectives; falls back to eager by itself)
    graph_update = use_graph
    if use_graph == "auto":
        use_graph = dev.type == "cuda" and not multi
        graph_update = dev.type == "cuda" and (not multi or dist.get_backend() == "nccl")
    torch.manual_seed(seed)
    ac = ActorCritic(num_ensembles=num_ensembles).to(dev)
    if multi:
        for p_ in ac.parameters():
            dist.broadcast(p_.data, src=0)
        torch.manual_seed(seed + 7919 * rank)            # decorrelate exploration noise / minibatch order across ranks
    mirror = envs.get_mirror_indices() if use_mirror and hasattr(envs, "get_mirror_indices") else None
    n = envs.num_envs
    if agent_factory is not None:
        agent = agent_factory(ac, ppo_epoch=ppo_epoch, mini_batch_size=mini_batch_size, lr=lr, mirror_indices=mirror, use_graph=bool(use_graph))
    else:
        agent = PPO(ac, ppo_epoch=ppo_epoch, mini_batch_size=mini_batch_size, lr=lr, mirror_indices=mirror, use_graph=graph_update)
    roll = Rollouts(num_steps, n, dev)
    ring = EpisodeRing(n, dev)
    any_logger = logger is not None
    if multi:                                             # one decision for all ranks: the CSV row needs a collective
        flag = torch.tensor([1.0 if logger is not None else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        any_logger = bool(flag.item())
    curriculum = specialist = 0
    if use_curriculum:
        envs.update_curriculum(curriculum)
    if use_specialist:
        envs.update_specialist(specialist)
    uniform_sampling, uniform_counter = True, 1            # train.py:124-128
    if sampling != "none":
        assert eval_envs is not None, "adaptive / threshold sampling needs eval_envs"
        eval_envs.update_curriculum(0)                    # train.py:131,137
    roll.obs[0].copy_(envs.reset())
    collector = GraphedCollector(envs, ac, roll, num_steps, ring=ring) if use_graph else None
    history, start = [], time.time()
    test_rets = torch.empty(0)
    next_checkpoint, max_ep_reward = save_every, float("-inf")
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
    for j in range(num_updates):
        agent.set_lr(harness.exponential_decay(j, 0.99, lr, 3e-5))
        # -- sampling grid of this update (train.py:229-272, 320-361)
        grid_updated = False
        if sampling == "threshold" and uniform_sampling:
            envs.update_curriculum(5)
        elif sampling in ("threshold", "adaptive"):
            g = None
            if rank == 0:
                g = sampling_probs_from_values(ac, eval_envs, mode=sampling, curriculum_threshold=curriculum_threshold,
                                               as_tensor=dev.type == "cuda")
            if multi:                                     # one grid for the whole job: rank 0's
                gt, ok = torch.zeros((11, 11), device=dev), torch.zeros(1, device=dev)
                if g is not None:
                    gt.copy_(torch.as_tensor(g, dtype=torch.float32))
                    ok.fill_(1)
                dist.broadcast(ok, src=0)
                dist.broadcast(gt, src=0)
                g = (gt if dev.type == "cuda" else gt.double().numpy()) if bool(ok.item()) else None
            if g is not None:
                envs.update_sample_prob(g)
                grid_updated = True
        # -- rollout
        if collector is not None:
            collector()
        else:
            collect(envs, ac, roll, num_steps, ring=ring)
        if on_rollout is not None:
            on_rollout(j, roll)
        if sampling == "threshold":                       # train.py:460-469
            uniform_sampling = (uniform_counter % uniform_every == 0)
            uniform_counter = 0 if uniform_sampling else uniform_counter
            uniform_counter += 1
            if uniform_sampling:
                envs.update_curriculum(5)
        # -- deterministic test episodes (train.py:472-500)
        if test_envs is not None and test_interval and j % test_interval == 0:
            r = evaluate(test_envs, ac, getattr(test_envs, "_max_episode_steps", 1000))
            if r.numel():
                test_rets = torch.cat([test_rets, r])[-test_envs.num_envs:]      # deque(maxlen=num_tests)
        # -- curriculum gate on the recent-episode mean, one decision for all ranks
        ssum, cnt = ring.all_ranks_sum_count()            # the one host sync of the rollout
        mean_ret = ssum / cnt if cnt else float("nan")
        if use_curriculum and cnt and mean_ret > 1000 and curriculum <= 4:
            curriculum += 1
            envs.update_curriculum(curriculum)
        with torch.no_grad():
            next_value = ac.get_value(roll.obs[-1])
        roll.compute_returns(next_value, True, gamma, gae_lambda)
        vl, al, ent = agent.update(roll)
        roll.after_update()
        frames = (j + 1) * num_steps * n * world
        # -- checkpoints (rank 0)
        if save_dir and rank == 0:
            if frames >= next_checkpoint or j == num_updates - 1:
                name = "%s_%d.pt" % (env_name, int(next_checkpoint))
                next_checkpoint += save_every
            else:
                name = "%s_latest.pt" % env_name
            save_checkpoint(ac, os.path.join(save_dir, name), frames=frames, update=j + 1)
            if cnt > 1 and mean_ret > max_ep_reward:
                max_ep_reward = mean_ret
                save_checkpoint(ac, os.path.join(save_dir, "%s_best.pt" % env_name), frames=frames, update=j + 1, mean_rew=mean_ret)
        if use_specialist and cnt and mean_ret > 1000 and specialist <= 4:
            if save_dir and rank == 0:
                save_checkpoint(ac, os.path.join(save_dir, "%s_specialist_%d.pt" % (env_name, specialist)), frames=frames)
            specialist += 1
            envs.update_specialist(specialist)
        stats = {"iter": j + 1, "total_num_steps": frames, "fps": int(frames / (time.time() - start)), "entropy": ent,
                 "value_loss": vl, "action_loss": al, "mean_rew": mean_ret, "curriculum": curriculum,
                 "grid_updated": grid_updated}
        history.append(stats)
        # collective: EVERY rank calls it when any rank logs (only rank 0 holds a logger: steppingstone_amd/train.py)
        vals = ring.all_ranks_values() if (any_logger and cnt > 1) else None
        if logger is not None and rank == 0 and cnt > 1 and vals.numel():      # train.py:564: only once episodes have finished
            logger.log_epoch({"iter": j + 1, "total_num_steps": frames, "fps": stats["fps"], "entropy": ent,
                              "value_loss": vl, "action_loss": al, "stats": {"rew": vals.numpy()},
                              "test_stats": {"rew": (test_rets if test_rets.numel() else torch.full((1,), float("nan"))).numpy()}})
        if log:
            log(stats)
    return ac, history
