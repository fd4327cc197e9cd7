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
                 use_graph=False):
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

    def _allreduce_grads(self):
        """Data-parallel learner: one RCCL all-reduce of the flattened gradient per minibatch (no-op on one rank)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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
        return self.use_graph and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

    def _gathered_step(self, data, idx):
        return self.step_minibatch(*(t[idx] for t in data))

    def _graph_step(self, data, idx, refresh=True):
        """data: the six flat rollout tensors of this update (copied into static storage when refresh is set, i.e.
        once per update); idx: minibatch indices."""
        if self._static is None:
            self._static = (tuple(torch.empty_like(t) for t in data), torch.zeros_like(idx), torch.zeros(3, device=idx.device))
            refresh = True
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
            with torch.cuda.graph(g):
                out = self._gathered_step(sdata, sidx)
                sout.copy_(torch.stack(out))
            self._graph = g
        sidx.copy_(idx)
        self._graph.replay()
        return sout.clone()

    def update(self, roll):
        adv = roll.returns[:-1] - roll.value_preds[:-1]
        adv = (adv - adv.mean()) / (adv.std() + 1e-5)
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


def collect(envs, ac, roll, num_steps, ep_returns=None, ep_stats=None):
    """The rollout loop of playground/train.py:363-469 with tensorised bookkeeping.  `envs` returns device tensors
    (SteppingStoneVecEnv(return_numpy=False) or ShardedVecEnv).  Finished episodes are reported either through
    ep_stats, a device tensor [2] accumulating (sum of returns, count) with no host synchronisation (the form a
    hipGraph can capture), or appended to the list ep_returns (one host sync per step)."""
    for _ in range(num_steps):
        with torch.no_grad():
            value, action, logp = ac.act(roll.obs[roll.step])
        obs, rew, done, info = envs.step(action)
        d = done.to(torch.float32).unsqueeze(1)
        mask = 1.0 - d
        bad_mask = 1.0 - info["bad_transition"].to(torch.float32).unsqueeze(1)
        if ep_stats is not None:
            ep_stats[0] += (info["ep_ret"] * d[:, 0]).sum()
            ep_stats[1] += d.sum()
        elif ep_returns is not None and bool(done.any()):
            ep_returns.append(info["ep_ret"][done].clone())
        roll.insert(obs, action, logp, value, rew.unsqueeze(1), mask, bad_mask)


class GraphedCollector:
    """collect() captured once in a hipGraph (policy inference, env step kernel, storage writes of all num_steps
    steps: ~50 launches per step) and replayed per update.  The rollout storage, the env's I/O buffers and the
    episode accumulators are static, and every call starts at roll.step == 0, so all addresses are replay-stable."""

    def __init__(self, envs, ac, roll, num_steps):
        self.envs, self.ac, self.roll, self.num_steps = envs, ac, roll, num_steps
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
                    collect(self.envs, self.ac, self.roll, self.num_steps, ep_stats=self.ep_stats)
                torch.cuda.current_stream().wait_stream(side)
                return self.ep_stats
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                collect(self.envs, self.ac, self.roll, self.num_steps, ep_stats=self.ep_stats)
            self.graph = g
        self.graph.replay()
        return self.ep_stats


def sampling_probs_from_values(ac, eval_envs, mode="threshold", curriculum_threshold=0.85, events=5, max_steps=400):
    """Adaptive / threshold curriculum sampler (playground/train.py:229-272, 320-361), batched: the evaluation envs
    are rolled with the deterministic policy; every env that advances to a new target contributes the critic-ensemble
    value of its 121 hypothetical next stones (create_temp_states); after `events` such contributions the summed
    11x11 metric, normalised by its absolute maximum, becomes softmax(-10 |m - threshold|) ("threshold") or
    softmax(-10 m) ("adaptive").  Returns an (11,11) float64 numpy grid for update_sample_prob."""
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
            temp = eval_envs.create_temp_states()[hit]                      # [k,121,60]
            with torch.no_grad():
                v = ac.get_ensemble_values(temp.reshape(-1, temp.shape[-1])).mean(dim=-1)
            total += v.view(-1, 121).sum(dim=0)
            seen += int(hit.sum())
            if seen >= events:
                break
    if seen == 0:
        return None
    m = total / total.abs().max()
    logits = -10.0 * (m - curriculum_threshold).abs() if mode == "threshold" else -10.0 * m
    return torch.softmax(logits, dim=0).view(11, 11).double().cpu().numpy()


def train(envs, num_updates, num_steps=32, num_ensembles=1, seed=8, use_curriculum=True, use_mirror=False, lr=3e-4,
          gamma=0.99, gae_lambda=0.95, ppo_epoch=10, mini_batch_size=1024, log=print, use_graph="auto"):
    """Fixed-order-curriculum PPO (playground/train.py:115-118,211-222,503-521).  Returns the list of per-update stats.
    use_graph ("auto": on a GPU with a single rank): rollout and minibatch step replay as hipGraphs, and the episode
    statistics stay on the device (mean return of the episodes finished during the update, carried over when none
    finished) instead of the reference's host-side deque."""
    import torch.distributed as dist
    torch.manual_seed(seed)
    dev = envs.device if hasattr(envs, "device") else envs.local.device
    dev = torch.device(dev)
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if use_graph == "auto":
        use_graph = dev.type == "cuda" and not multi
    ac = ActorCritic(num_ensembles=num_ensembles).to(dev)
    mirror = envs.get_mirror_indices() if use_mirror and hasattr(envs, "get_mirror_indices") else None
    agent = PPO(ac, ppo_epoch=ppo_epoch, mini_batch_size=mini_batch_size, lr=lr, mirror_indices=mirror, use_graph=use_graph)
    n = envs.num_envs
    roll = Rollouts(num_steps, n, dev)
    curriculum = 0
    if use_curriculum:
        envs.update_curriculum(curriculum)
    roll.obs[0].copy_(envs.reset())
    collector = GraphedCollector(envs, ac, roll, num_steps) if use_graph else None
    recent, history, start = [], [], time.time()
    mean_ret = float("nan")
    for j in range(num_updates):
        agent.set_lr(harness.exponential_decay(j, 0.99, lr, 3e-5))
        if collector is not None:
            ssum, cnt = collector().tolist()               # the one host sync of the rollout
            if cnt > 0:
                mean_ret = ssum / cnt
            have = not math.isnan(mean_ret)
        else:
            ep = []
            collect(envs, ac, roll, num_steps, ep)
            recent = (recent + ep)[-50:]
            mean_ret = float(torch.cat(recent).mean()) if recent else float("nan")
            have = bool(recent)
        if use_curriculum and have and mean_ret > 1000 and curriculum <= 4:
            curriculum += 1
            envs.update_curriculum(curriculum)
        with torch.no_grad():
            next_value = ac.get_value(roll.obs[-1])
        roll.compute_returns(next_value, True, gamma, gae_lambda)
        vl, al, ent = agent.update(roll)
        roll.after_update()
        frames = (j + 1) * num_steps * n
        stats = {"iter": j + 1, "total_num_steps": frames, "fps": int(frames / (time.time() - start)), "entropy": ent,
                 "value_loss": vl, "action_loss": al, "mean_rew": mean_ret, "curriculum": curriculum}
        history.append(stats)
        if log:
            log(stats)
    return ac, history
