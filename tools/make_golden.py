#!/usr/bin/env python3
"""Generate tests/golden/harness_golden.npz by importing the REFERENCE's own pure-Python harness code from
/root/reference (under a ~30-line stub `gym`) in THIS container.  The fixture holds inputs and the
reference's outputs only (data, no reference source); the reference never travels to the GPU box.

  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.dont_write_bytecode = True


def install_stub_gym():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")
    core = types.ModuleType("gym.core")

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.asarray(low).shape
            self.dtype = np.dtype(dtype)
            self.low, self.high = low, high

    class Dict(dict):
        pass

    class MultiDiscrete:
        pass

    class Wrapper:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.env, name)

        def close(self):
            return self.env.close()

    registry = {}

    def make(env_id, **kw):
        return registry[env_id](**kw)

    spaces.Box, spaces.Dict, spaces.MultiDiscrete = Box, Dict, MultiDiscrete
    core.Wrapper = Wrapper
    gym.spaces, gym.core, gym.Wrapper, gym.make, gym.registry = spaces, core, Wrapper, make, registry
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.core": core})
    return gym


class ToyEnv:
    """Scripted env: obs[0]=rank, obs[1]=t; reward 0.25*(t+1)+rank; episode length 4+rank."""

    class _Spec:
        id = "Toy-v0"

    def __init__(self, render=False):
        import gym
        self.observation_space = gym.spaces.Box(-np.inf, np.inf, shape=(60,), dtype=np.float32)
        self.action_space = gym.spaces.Box(-1, 1, shape=(21,), dtype=np.float32)
        self.spec = self._Spec()
        self.rank = 0
        self.t = 0

    def seed(self, s):
        self.rank = int(s) - 100

    def _obs(self):
        o = np.zeros(60, np.float32)
        o[0], o[1] = self.rank, self.t
        return o

    def reset(self):
        self.t = 0
        return self._obs()

    def step(self, a):
        r = 0.25 * (self.t + 1) + self.rank + float(a[0])
        self.t += 1
        done = self.t >= 4 + self.rank
        return self._obs(), r, done, {}

    def close(self):
        pass


def main():
    gym = install_stub_gym()
    gym.registry["Toy-v0"] = ToyEnv
    sys.path.insert(0, REF)
    sys.path.insert(0, ROOT)
    from algorithms.storage import RolloutStorage
    from common import envs_utils, misc_utils
    from steppingstone_amd import _lib

    out = {}
    rng = np.random.default_rng(20260928)

    # ---- 1. GAE / returns  (algorithms/storage.py:59-82)
    T, N = 8, 4
    for tag, use_gae in (("gae", True), ("ret", False)):
        st = RolloutStorage(T, N, (60,), 21, 1)
        rewards = rng.normal(size=(T, N, 1)).astype(np.float32)
        values = rng.normal(size=(T + 1, N, 1)).astype(np.float32)
        masks = (rng.random((T + 1, N, 1)) > 0.2).astype(np.float32)
        bad = (rng.random((T + 1, N, 1)) > 0.1).astype(np.float32)
        st.rewards.copy_(torch.from_numpy(rewards))
        st.value_preds.copy_(torch.from_numpy(values))
        st.masks.copy_(torch.from_numpy(masks))
        st.bad_masks.copy_(torch.from_numpy(bad))
        nv = torch.from_numpy(values[-1].copy())
        st.compute_returns(nv, use_gae, 0.99, 0.95)
        out.update({tag + "_rewards": rewards, tag + "_values": values, tag + "_masks": masks, tag + "_bad": bad,
                    tag + "_returns": st.returns.numpy().copy()})

    # ---- 2. mirror function (common/envs_utils.py:687-740) with THIS repository's index lists
    idx = _lib.mirror_indices()
    mf = envs_utils.get_mirror_function(idx)
    obs = torch.from_numpy(rng.normal(size=(3, 60)).astype(np.float32))
    act = torch.from_numpy(rng.normal(size=(3, 21)).astype(np.float32))
    z1 = torch.zeros(3, 1)
    res = mf((obs, z1, act, z1, z1, z1, z1, z1))
    out.update({"mirror_obs_in": obs.numpy(), "mirror_act_in": act.numpy(), "mirror_obs_out": res[0].numpy(),
                "mirror_act_out": res[2].numpy()})

    # ---- 3. LR schedules (common/misc_utils.py:20-27; playground/train.py:213-216)
    ep = np.arange(0, 400, 7)
    out["decay_epochs"] = ep
    out["exp_decay"] = np.array([misc_utils.exponential_decay(int(e), 0.99, 3e-4, 3e-5) for e in ep])
    out["lin_decay"] = np.array([misc_utils.linear_decay(int(e), 5000, 3e-4, 0.0) for e in ep])

    # ---- 4. vec-env protocol through the reference's ShmemVecEnv + Monitor (envs_utils.py:48-56,486-675)
    with tempfile.TemporaryDirectory() as d:
        envs = envs_utils.make_vec_envs("Toy-v0", 100, 3, d)
        obs_seq = [envs.reset()]
        rew_seq, done_seq, epr, epl = [], [], [], []
        acts = rng.uniform(-1, 1, size=(14, 3, 21)).astype(np.float32)
        for t in range(14):
            o, r, dn, infos = envs.step(acts[t])
            obs_seq.append(o)
            rew_seq.append(r)
            done_seq.append(dn)
            epr.append([i["episode"]["r"] if "episode" in i else np.nan for i in infos])
            epl.append([i["episode"]["l"] if "episode" in i else -1 for i in infos])
        out.update({"vec_actions": acts, "vec_obs": np.stack(obs_seq), "vec_rew": np.stack(rew_seq),
                    "vec_done": np.stack(done_seq), "vec_ep_r": np.array(epr), "vec_ep_l": np.array(epl)})
        assert obs_seq[0].dtype == np.float32 and rew_seq[0].dtype == np.float64 and done_seq[0].dtype == bool
        envs.close()

    # ---- 5. one PPO update (algorithms/ppo.py:40-108) from a fixed initialisation on a fixed batch
    from algorithms.ppo import PPO
    from common.controller import Policy, SoftsignActor
    torch.manual_seed(1234)
    ac = Policy(SoftsignActor(ToyEnv()), num_ensembles=2)
    sd0 = {k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}
    T, N = 4, 8
    st = RolloutStorage(T, N, (60,), 21, 1)
    obs = torch.from_numpy(rng.normal(size=(T + 1, N, 60)).astype(np.float32))
    act = torch.from_numpy(rng.normal(size=(T, N, 21)).astype(np.float32) * 0.5)
    with torch.no_grad():
        _, logp0, _, _ = ac.evaluate_actions(obs[:-1].view(-1, 60), None, None, act.view(-1, 21))
    old_logp = logp0.view(T, N, 1) + torch.from_numpy(rng.normal(size=(T, N, 1)).astype(np.float32)) * 0.3
    vpred = torch.from_numpy(rng.normal(size=(T + 1, N, 1)).astype(np.float32))
    rets = torch.from_numpy(rng.normal(size=(T + 1, N, 1)).astype(np.float32))
    st.observations.copy_(obs); st.actions.copy_(act); st.action_log_probs.copy_(old_logp)
    st.value_preds.copy_(vpred); st.returns.copy_(rets)
    agent = PPO(ac, clip_param=0.2, ppo_epoch=1, num_mini_batch=1, value_loss_coef=1.0, entropy_coef=0.0, lr=3e-4, eps=1e-5,
                max_grad_norm=2.0, use_clipped_value_loss=False)
    vl, al, ent = agent.update(st)
    sd1 = {k: v.detach().clone().numpy() for k, v in ac.state_dict().items()}
    out.update({"ppo_obs": obs.numpy(), "ppo_act": act.numpy(), "ppo_old_logp": old_logp.numpy(), "ppo_vpred": vpred.numpy(),
                "ppo_returns": rets.numpy(), "ppo_losses": np.array([vl, al, ent])})
    for k, v in sd0.items():
        out["ppo_w0/" + k] = v
    for k, v in sd1.items():
        out["ppo_w1/" + k] = v

    path = os.path.join(ROOT, "tests", "golden", "harness_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
