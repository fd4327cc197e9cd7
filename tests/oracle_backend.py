"""Test double: an object with HipBackend's call surface backed by the CPU oracle, so the host-side protocol code
(steppingstone_amd.envs / .distributed) can be exercised in the GPU-less container.  TESTS ONLY -- the product
never constructs this."""
import numpy as np
import torch

import oracle_lib as ol


class OracleBackend:
    def __init__(self, kind, num_envs, seed, device=None, env_id_offset=0):
        self.device = torch.device("cpu")
        self.n = int(num_envs)
        self.o = ol.OracleEnv(kind, num_envs, seed=seed, env_offset=env_id_offset)

    def close(self):
        self.o.close()

    def reset(self, obs):
        obs.copy_(torch.from_numpy(self.o.reset()))

    def step(self, act, obs, rew, done, info):
        o, r, d, i = self.o.step(act.numpy())
        obs.copy_(torch.from_numpy(o))
        rew.copy_(torch.from_numpy(r))
        done.copy_(torch.from_numpy(d))
        raw = np.zeros((self.n, 6), np.int32)
        raw.view(np.float32)[:, 0] = i["ep_ret"]
        raw.view(np.float32)[:, 1] = i["ep_len"]
        raw.view(np.float32)[:, 5] = i["ep_ret_lo"]
        raw[:, 2], raw[:, 3], raw[:, 4] = i["bad_transition"], i["steps_reached"], i["update_terrain"]
        info.copy_(torch.from_numpy(raw))

    def rollout_random(self, num_steps, t0, obs, rew, done, info, steps_per_launch=0):
        act = torch.zeros((self.n, 21))
        for k in range(num_steps):
            act.copy_(torch.from_numpy(self.o.random_actions(t0 + k)))
            self.step(act, obs, rew, done, info)

    def step_packed(self, act, use_random, t, packed, info):
        a = torch.from_numpy(self.o.random_actions(t)) if use_random else act
        obs, rew = torch.zeros((self.n, 60)), torch.zeros(self.n)
        done = torch.zeros(self.n, dtype=torch.uint8)
        self.step(a, obs, rew, done, info)
        packed[:, :60] = obs
        packed[:, 60] = rew
        packed[:, 61] = done.float()

    def rollout_random_packed(self, num_steps, t0, packed, info):
        for k in range(num_steps):
            self.step_packed(None, True, t0 + k, packed[k], info)

    def random_actions(self, t, act):
        act.copy_(torch.from_numpy(self.o.random_actions(t)))

    def set_curriculum(self, level):
        self.o.set_curriculum(level)

    def set_specialist(self, level):
        self.o.set_specialist(level)

    def set_sample_prob(self, prob, per_env):
        self.o.set_sample_prob(np.asarray(prob, np.float64))

    def set_mirror(self, on):
        pass

    def set_power(self, power):
        self.o.set_power(power)

    def set_auto_reset(self, on):
        self.o.set_auto_reset(on)

    def create_temp_states(self, out):
        out.copy_(torch.from_numpy(self.o.create_temp_states()))

    def get_state(self, packed):
        packed.copy_(torch.from_numpy(self.o.get_state().astype(np.float32)))

    def set_state(self, packed):
        self.o.set_state(packed.numpy())

    def get_obs(self, obs):
        obs.copy_(torch.from_numpy(self.o.get_obs()))
