"""Host-side mirror of the reference's env / vec-env protocol for the stepping-stone hot path.

Drop-in surface (names, argument meaning, error behaviour) follows what `playground/train.py` and
`playground/enjoy.py` of the reference consume through `common/envs_utils.py`:

  make_env(env_id)                      common/envs_utils.py:43-45
  make_vec_envs(env_id, seed, n, dir)   common/envs_utils.py:48-56
  VecEnv.reset/step_async/step_wait/step, update_curriculum, update_specialist, update_sample_prob,
  create_temp_states, set_mirror, set_env_params, set_robot_params, close   common/envs_utils.py:542-606

Instead of N worker processes with pipes and shared memory, all N environments live in HBM on one MI355X and one
kernel launch advances them together.  PyTorch is plumbing only (device buffers + current stream); the arithmetic
is in libsteppingstone.so (hand-written HIP).  There is NO CPU implementation in this package.
"""
import ctypes as C
import time
import types

import numpy as np
import torch

from . import _lib
from ._lib import ACT_DIM, GRID, INFO_WORDS, MAX_EPISODE_STEPS, NCELL, NUM_STONES, OBS_DIM, STATE_DIM, SteppingStoneError

DEG = np.pi / 180.0
ENV_KINDS = {
    "Walker3DStepperEnv-v0": _lib.WALKER3D, "mocca_envs:Walker3DStepperEnv-v0": _lib.WALKER3D,
    "MikeStepperEnv-v0": _lib.MIKE, "mocca_envs:MikeStepperEnv-v0": _lib.MIKE,
}
_EMPTY_INFO = types.MappingProxyType({})


class Box:
    """Minimal stand-in for gym.spaces.Box (gym is not a dependency): shape/dtype/low/high."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, self.dtype)
        self.high = np.full(self.shape, high, self.dtype)

    def sample(self):
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return np.random.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return "Box%s" % (self.shape,)


class EnvSpec:
    def __init__(self, env_id):
        self.id = env_id
        self.max_episode_steps = MAX_EPISODE_STEPS


def kind_of(env_id):
    if env_id not in ENV_KINDS:
        raise SteppingStoneError("unknown env id %r (known: %s)" % (env_id, ", ".join(sorted(ENV_KINDS))))
    return ENV_KINDS[env_id]


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class HipBackend:
    """Owns one ss_env handle on one GPU.  All tensor arguments must be contiguous tensors on `self.device`."""

    def __init__(self, kind, num_envs, seed, device, env_id_offset=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise SteppingStoneError("no GPU visible to torch: the stepping-stone env runs on MI355X only "
                                     "(there is no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise SteppingStoneError("device must be a cuda (HIP) device, got %s" % (self.device,))
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.n = int(num_envs)
        h = C.c_void_p()
        _lib.check(self.lib.ss_create(C.byref(h), int(kind), self.n, idx, int(seed) & (2 ** 64 - 1), int(env_id_offset)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.ss_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, obs):
        _lib.check(self.lib.ss_reset(self.h, _ptr(obs), _stream(self.device)))

    def step(self, act, obs, rew, done, info):
        _lib.check(self.lib.ss_step(self.h, _ptr(act), _ptr(obs), _ptr(rew), _ptr(done), _ptr(info), _stream(self.device)))

    def rollout_random(self, num_steps, t0, obs, rew, done, info, steps_per_launch=0):
        _lib.check(self.lib.ss_rollout_random(self.h, int(num_steps), int(steps_per_launch), int(t0), _ptr(obs), _ptr(rew),
                                              _ptr(done), _ptr(info), _stream(self.device)))

    def step_packed(self, act, use_random, t, packed, info):
        _lib.check(self.lib.ss_step_packed(self.h, _ptr(act) if act is not None else None, 1 if use_random else 0, int(t),
                                           _ptr(packed), _ptr(info), _stream(self.device)))

    def rollout_random_packed(self, num_steps, t0, packed, info):
        _lib.check(self.lib.ss_rollout_random_packed(self.h, int(num_steps), int(t0), _ptr(packed), _ptr(info), _stream(self.device)))

    def random_actions(self, t, act):
        _lib.check(self.lib.ss_random_actions(self.h, int(t), _ptr(act), _stream(self.device)))

    def set_curriculum(self, level):
        _lib.check(self.lib.ss_set_curriculum(self.h, int(level)))

    def set_specialist(self, level):
        _lib.check(self.lib.ss_set_specialist(self.h, int(level)))

    def set_sample_prob(self, prob, per_env):
        prob = np.ascontiguousarray(prob, np.float64)
        _lib.check(self.lib.ss_set_sample_prob(self.h, prob.ctypes.data_as(C.c_void_p), 1 if per_env else 0))

    def set_sample_prob_device(self, prob, per_env):
        _lib.check(self.lib.ss_set_sample_prob_device(self.h, _ptr(prob), 1 if per_env else 0, _stream(self.device)))

    def set_mirror(self, on):
        _lib.check(self.lib.ss_set_mirror(self.h, 1 if on else 0))

    def set_power(self, power):
        _lib.check(self.lib.ss_set_power(self.h, float(power)))

    def set_auto_reset(self, on):
        _lib.check(self.lib.ss_set_auto_reset(self.h, 1 if on else 0))

    def create_temp_states(self, out):
        _lib.check(self.lib.ss_create_temp_states(self.h, _ptr(out), _stream(self.device)))

    def get_state(self, packed):
        _lib.check(self.lib.ss_get_state(self.h, _ptr(packed), _stream(self.device)))

    def set_state(self, packed):
        _lib.check(self.lib.ss_set_state(self.h, _ptr(packed), _stream(self.device)))

    def get_obs(self, obs):
        _lib.check(self.lib.ss_get_obs(self.h, _ptr(obs), _stream(self.device)))


class SteppingStoneVecEnv:
    """N stepping-stone environments advanced by one HIP kernel launch per step.

    return_numpy=True reproduces the reference's ShmemVecEnv return types exactly (obs float32 (N,60) ndarray,
    rews float64 (N,), dones bool (N,), infos sequence of N dicts; common/envs_utils.py:555-558) so that
    playground/train.py consumes it unchanged; return_numpy=False keeps everything on the GPU (obs/rew/done
    tensors, infos as a dict of tensors) for a device-resident PPO loop.
    """

    closed = False

    def __init__(self, env_id, num_envs, seed=0, device=None, env_id_offset=0, return_numpy=False, backend=None, log_dir=None):
        self.env_id = env_id
        self.kind = kind_of(env_id)
        self.num_envs = int(num_envs)
        self.seed_value = int(seed)
        self.return_numpy = bool(return_numpy)
        self.spec = EnvSpec(env_id)
        high = np.inf
        self.observation_space = Box(-high, high, (OBS_DIM,), np.float32)
        self.action_space = Box(-1.0, 1.0, (ACT_DIM,), np.float32)
        # `backend` is a test seam (tests inject an oracle-backed object); the product path is always HIP.
        self.backend = backend if backend is not None else HipBackend(self.kind, num_envs, seed, device, env_id_offset)
        dev = self.backend.device
        self.device = dev
        n = self.num_envs
        self._obs = torch.zeros((n, OBS_DIM), dtype=torch.float32, device=dev)
        self._rew = torch.zeros((n,), dtype=torch.float32, device=dev)
        self._done = torch.zeros((n,), dtype=torch.uint8, device=dev)
        self._info = torch.zeros((n, INFO_WORDS), dtype=torch.int32, device=dev)
        self._act = torch.zeros((n, ACT_DIM), dtype=torch.float32, device=dev)
        self._pending = False
        # numpy drop-in mode on a GPU: pinned staging buffers, one H->D copy of the actions and one D->H copy of the
        # packed [N,62] = obs | rew | done block per step (instead of four pageable copies)
        self._pinned = None
        if self.return_numpy and torch.device(dev).type == "cuda":
            self._pinned = {"act": torch.zeros((n, ACT_DIM), dtype=torch.float32).pin_memory(),
                            "out": torch.zeros((n, OBS_DIM + 2), dtype=torch.float32).pin_memory(),
                            "dev": torch.zeros((n, OBS_DIM + 2), dtype=torch.float32, device=dev),
                            "info": torch.zeros((n, INFO_WORDS), dtype=torch.int32).pin_memory(),
                            "event": torch.cuda.Event()}
        self._tstart = time.time()
        # Monitor's files (common/envs_utils.py:36-38,172-194): <log_dir>/<rank>.monitor.csv per env, a row per finished episode
        self._monitor = None
        if log_dir is not None:
            from .monitor_csv import MonitorFiles
            self._monitor = MonitorFiles(log_dir, self.num_envs, env_id, first_rank=env_id_offset, t_start=self._tstart)
        self.yaw_samples = np.linspace(-20.0, 20.0, GRID) * DEG
        self.pitch_samples = np.linspace(-30.0, 30.0, GRID) * DEG
        self.yaw_sample_size = GRID
        self.pitch_sample_size = GRID
        self._max_episode_steps = MAX_EPISODE_STEPS
        self.curriculum = 0

    # ------------------------------------------------------------------ gym / VecEnv protocol
    def reset(self):
        if self._pending:
            print("Called reset() while waiting for the step to complete")
            self.step_wait()
        self.backend.reset(self._obs)
        return self._out_obs()

    def step_async(self, actions):
        if self._pinned is not None and not torch.is_tensor(actions):
            pb = self._pinned
            a = np.asarray(actions, dtype=np.float32)
            assert a.shape[0] == self.num_envs, "expected %d actions, got %d" % (self.num_envs, a.shape[0])
            pb["act"].numpy()[...] = a.reshape(self.num_envs, ACT_DIM)
            self._act.copy_(pb["act"], non_blocking=True)
            self.backend.step_packed(self._act, False, 0, pb["dev"], self._info)
            pb["out"].copy_(pb["dev"], non_blocking=True)
            pb["info"].copy_(self._info, non_blocking=True)      # the step report travels with the block: no second, synchronous copy
            pb["event"].record()
            self._pending = "pinned"
            return
        if torch.is_tensor(actions):
            a = actions.to(device=self.device, dtype=torch.float32)
        else:
            a = torch.from_numpy(np.ascontiguousarray(actions, dtype=np.float32)).to(self.device)
        assert a.shape[0] == self.num_envs, "expected %d actions, got %d" % (self.num_envs, a.shape[0])
        self._act.copy_(a.reshape(self.num_envs, ACT_DIM))
        self.backend.step(self._act, self._obs, self._rew, self._done, self._info)
        self._pending = True

    def step_wait(self):
        if self._pending == "pinned":
            self._pending = False
            pb = self._pinned
            pb["event"].synchronize()
            out = pb["out"].numpy()
            obs = out[:, :OBS_DIM].copy()                       # fresh arrays every step, like common/envs_utils.py:619
            rew = out[:, OBS_DIM].astype(np.float64)
            done = out[:, OBS_DIM + 1] > 0.5
            self._info_host = pb["info"].numpy()
            return obs, rew, done, self._info_dicts(done, self._info_host)
        self._pending = False
        if not self.return_numpy:
            if self._monitor is not None:      # opt-in (log_dir): the episode rows need the finished envs on the host every step
                self._info_dicts(self._done.cpu().numpy().astype(bool))
            return self._obs, self._rew, self._done.bool(), self._info_tensors()
        obs = self._obs.cpu().numpy()
        rew = self._rew.cpu().numpy().astype(np.float64)
        done = self._done.cpu().numpy().astype(bool)
        self._info_host = self._info.cpu().numpy()
        return obs, rew, done, self._info_dicts(done, self._info_host)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def rollout_random(self, num_steps, t0=0, steps_per_launch=0):
        """BASELINE metric path: num_steps control steps with on-device U(-1,1) actions (Philox stream 1),
        steps_per_launch of them per kernel launch (0: the library default of 1000; 1: one launch per step)."""
        self.backend.rollout_random(num_steps, t0, self._obs, self._rew, self._done, self._info, steps_per_launch)
        return self._obs, self._rew, self._done

    def step_packed(self, packed, actions=None, t=0, info=None):
        """One step written into the caller's [N,62] buffer (obs | rew | done): the block ShardedVecEnv all-gathers.
        actions=None draws them from the benchmark Philox stream at index t.  info: optional caller-owned [N,5] int32
        buffer for the per-env step report (default: this env's own)."""
        if actions is not None:
            self._act.copy_(actions.reshape(self.num_envs, ACT_DIM))
        self.backend.step_packed(self._act if actions is not None else None, actions is None, t, packed,
                                 self._info if info is None else info)
        return packed

    def rollout_random_packed(self, packed, t0=0):
        """packed: [K, N, 62] device buffer; ONE launch advances K control steps and writes step k's obs | rew | done block at
        packed[k] (the multi-GPU rollout ships such a chunk per collective)."""
        assert packed.dim() == 3 and packed.shape[1] == self.num_envs and packed.shape[2] == OBS_DIM + 2 and packed.is_contiguous()
        self.backend.rollout_random_packed(packed.shape[0], t0, packed, self._info)
        return packed

    def random_actions(self, t):
        self.backend.random_actions(t, self._act)
        return self._act.clone()

    def close(self):
        if self.closed:
            return
        if self._monitor is not None:
            self._monitor.close()
        self.backend.close()
        self.closed = True

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope for the GPU env")

    def get_images(self):
        raise NotImplementedError("rendering is out of scope for the GPU env")

    @property
    def unwrapped(self):
        return self

    # ------------------------------------------------------------------ curriculum hooks
    def update_curriculum(self, curriculum):
        self.curriculum = int(curriculum)
        self.backend.set_curriculum(min(max(self.curriculum, 0), 5))

    def update_specialist(self, specialist):
        self.curriculum = int(specialist)
        self.backend.set_specialist(min(max(int(specialist), 0), 5))

    def update_sample_prob(self, probs):
        """probs: (N,11,11) one grid per env (playground/train.py:267-271) or a single (11,11) grid.  A torch tensor on
        the env's device takes the stream-ordered path (no host copy, no synchronisation)."""
        if torch.is_tensor(probs) and probs.device.type == "cuda":
            if tuple(probs.shape) not in ((GRID, GRID), (self.num_envs, GRID, GRID)):
                raise ValueError("sample_prob must have shape (%d,%d,%d) or (%d,%d), got %s"
                                 % (self.num_envs, GRID, GRID, GRID, GRID, tuple(probs.shape)))
            p = probs.to(device=self.device, dtype=torch.float32).contiguous()
            self._prob_keepalive = p           # the kernel reads it asynchronously
            self.backend.set_sample_prob_device(p, p.dim() == 3)
            return
        probs = np.asarray(probs.cpu() if torch.is_tensor(probs) else probs, np.float64)
        if probs.shape == (GRID, GRID):
            self.backend.set_sample_prob(probs, False)
        elif probs.shape == (self.num_envs, GRID, GRID):
            same = self.num_envs == 1 or bool(np.all(probs == probs[0]))
            self.backend.set_sample_prob(probs[0] if same else probs, not same)
        else:
            raise ValueError("sample_prob must have shape (%d,%d,%d) or (%d,%d), got %s"
                             % (self.num_envs, GRID, GRID, GRID, GRID, probs.shape))

    _warned_set_mirror = False

    def create_temp_states(self):
        out = torch.empty((self.num_envs, NCELL, OBS_DIM), dtype=torch.float32, device=self.device)
        self.backend.create_temp_states(out)
        return out.cpu().numpy() if self.return_numpy else out

    def set_mirror(self, mirror):
        """common/envs_utils.py:588-590.  Accepted for protocol compatibility only: no effect on Walker3D / Mike (their observation has
        no gait-phase term; include/steppingstone.h ss_set_mirror).  The symmetry itself is get_mirror_indices().
        `use_phase_mirror=True` (playground/train.py:48,109-111) is therefore UNSUPPORTED: set_mirror(True) warns once per process
        instead of silently accepting; the reference's default is False."""
        if mirror and not SteppingStoneVecEnv._warned_set_mirror:
            import warnings
            SteppingStoneVecEnv._warned_set_mirror = True
            warnings.warn("set_mirror(True): phase mirroring (use_phase_mirror, playground/train.py:48,109) is not supported by this env -- the "
                          "call is accepted for protocol compatibility and has no effect; use get_mirror_indices() with mirror "
                          "augmentation (common/envs_utils.py:687-740) instead", RuntimeWarning, stacklevel=2)
        self.backend.set_mirror(bool(mirror))

    def set_env_params(self, params_dict):
        for k, v in dict(params_dict).items():
            if k == "curriculum":
                self.update_curriculum(v)
            else:
                raise KeyError("unsupported env param %r" % (k,))

    def set_robot_params(self, params_dict):
        for k, v in dict(params_dict).items():
            if k == "power":
                self.backend.set_power(float(v))
            else:
                raise KeyError("unsupported robot param %r" % (k,))

    def get_mirror_indices(self):
        return _lib.mirror_indices(self.kind)

    # ------------------------------------------------------------------ state access
    def get_state(self):
        st = torch.empty((self.num_envs, STATE_DIM), dtype=torch.float32, device=self.device)
        self.backend.get_state(st)
        return st

    def set_state(self, packed):
        st = torch.as_tensor(packed, dtype=torch.float32).to(self.device).contiguous().reshape(self.num_envs, STATE_DIM)
        self.backend.set_state(st)

    def get_obs(self):
        self.backend.get_obs(self._obs)
        return self._out_obs()

    @property
    def terrain_info(self):
        """(N,20,6) x,y,z,phi,x_tilt,y_tilt (playground/enjoy.py:60-64)."""
        return self.get_state()[:, 65:185].reshape(self.num_envs, NUM_STONES, 6).cpu().numpy()

    @property
    def next_step_index(self):
        return self.get_state()[:, 59].cpu().numpy().astype(np.int64)

    # ------------------------------------------------------------------ helpers
    def _out_obs(self):
        return self._obs.cpu().numpy() if self.return_numpy else self._obs

    def _info_tensors(self):
        fl = self._info.view(torch.float32)
        return {"ep_ret": fl[:, 0], "ep_len": fl[:, 1], "bad_transition": self._info[:, 2],
                "steps_reached": self._info[:, 3], "update_terrain": self._info[:, 4], "ep_ret_lo": fl[:, 5]}

    def _info_dicts(self, done, raw=None):
        """Sequence of N info dicts with the reference's keys (common/envs_utils.py:59-65,131-153).  raw: the [N,6] info words
        on the host already (numpy drop-in mode); else they are fetched when any env finished."""
        infos = [_EMPTY_INFO] * self.num_envs
        idx = np.nonzero(done)[0]
        if idx.size:
            if raw is None:
                raw = self._info.cpu().numpy()
            fl = raw.view(np.float32)
            now = round(time.time() - self._tstart, 6)
            # Monitor.update: eprew = sum(self.rewards) in Python floats, "r": round(eprew, 6) (common/envs_utils.py:134-138); the
            # kernel's (ep_ret, ep_ret_lo) pair is that fp64 sum of the fp32 step rewards.  Columns are pulled out as Python lists
            # once (per-element numpy indexing made this loop 200 us per step at 4096 envs, two thirds of the drop-in mode's step).
            rets = (fl[idx, 0].astype(np.float64) + fl[idx, 5].astype(np.float64)).tolist()
            lens = fl[idx, 1].astype(np.int64).tolist()
            bad, reached = raw[idx, 2].tolist(), raw[idx, 3].tolist()
            for k, i in enumerate(idx.tolist()):
                d = {"episode": {"r": round(rets[k], 6), "l": lens[k], "t": now}, "steps_reached": reached[k]}
                if self._monitor is not None:
                    self._monitor.write_row(i, d["episode"])
                if bad[k]:
                    d["bad_transition"] = True
                infos[i] = d
        return tuple(infos)


class _Robot:
    def __init__(self):
        self.feet_contact = np.zeros(2, np.float32)


class SteppingStoneEnv:
    """Single-env gym-style facade over a 1-env SteppingStoneVecEnv (what `make_env` returns in the reference;
    playground/train.py:96,129-131,231-247 and playground/enjoy.py:101-102,231-235 use these members)."""

    def __init__(self, env_id, seed=0, device=None, render=False, backend_factory=None):
        if render:
            raise NotImplementedError("rendering is out of scope for the GPU env")
        self._env_id = env_id
        self._device = device
        self._backend_factory = backend_factory
        self._make(seed)
        self.robot = _Robot()
        self.update_terrain = False
        self.camera = None

    def _make(self, seed):
        backend = self._backend_factory(kind_of(self._env_id), 1, seed) if self._backend_factory else None
        self.vec = SteppingStoneVecEnv(self._env_id, 1, seed=seed, device=self._device, return_numpy=True, backend=backend)
        self.vec.backend.set_auto_reset(False)
        self.observation_space = self.vec.observation_space
        self.action_space = self.vec.action_space
        self.spec = self.vec.spec
        self.yaw_samples, self.pitch_samples = self.vec.yaw_samples, self.vec.pitch_samples
        self.yaw_sample_size, self.pitch_sample_size = GRID, GRID
        self._max_episode_steps = MAX_EPISODE_STEPS

    @property
    def unwrapped(self):
        return self

    def seed(self, seed=None):
        curriculum = self.vec.curriculum
        self.vec.close()
        self._make(0 if seed is None else int(seed))
        self.vec.update_curriculum(curriculum)
        return [seed]

    def reset(self):
        self.update_terrain = False
        return self.vec.reset()[0]

    def step(self, action):
        obs, rew, done, infos = self.vec.step(np.asarray(action, np.float32).reshape(1, ACT_DIM))
        raw = self.vec._info_host[0]          # the step's info words, already on the host (no second copy)
        self.update_terrain = bool(raw[4])
        self.robot.feet_contact[:] = obs[0, 48:50]
        info = dict(infos[0])
        return obs[0], float(rew[0]), bool(done[0]), info

    def render(self, mode="human"):
        raise NotImplementedError("rendering is out of scope for the GPU env")

    def close(self):
        self.vec.close()

    def update_curriculum(self, curriculum):
        self.vec.update_curriculum(curriculum)

    def update_specialist(self, specialist):
        self.vec.update_specialist(specialist)

    def update_sample_prob(self, prob):
        self.vec.update_sample_prob(np.asarray(prob, np.float64).reshape(GRID, GRID))

    def set_mirror(self, mirror):
        self.vec.set_mirror(mirror)

    def create_temp_states(self):
        return self.vec.create_temp_states()[0]

    def get_mirror_indices(self):
        return self.vec.get_mirror_indices()

    @property
    def terrain_info(self):
        return self.vec.terrain_info[0]

    @property
    def next_step_index(self):
        return int(self.vec.next_step_index[0])


def make_env(env_id, render=False, seed=0, device=None):
    """Counterpart of common/envs_utils.py:43-45."""
    return SteppingStoneEnv(env_id, seed=seed, device=device, render=render)


def make_vec_envs(env_id, seed, num_processes, log_dir=None, device=None, return_numpy=True, env_id_offset=0):
    """Counterpart of common/envs_utils.py:48-56: `num_processes` environments with seeds seed+rank.  The reference
    forks one process per env; here they are lanes of one kernel on `device`.  log_dir: as in make_env_fns (:36-38), env `rank`
    logs its episodes to <log_dir>/<rank>.monitor.csv (steppingstone_amd/monitor_csv.py); None: no files, as in the reference."""
    assert num_processes > 1
    return SteppingStoneVecEnv(env_id, num_processes, seed=seed, device=device, env_id_offset=env_id_offset,
                               return_numpy=return_numpy, log_dir=log_dir)
