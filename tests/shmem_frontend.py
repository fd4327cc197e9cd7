"""CPU-baseline infrastructure: a process-per-env vec env with the ARCHITECTURE of the
reference's ShmemVecEnv (common/envs_utils.py:486-675) -- one worker process per environment, one pipe per worker
for commands and (reward, done, info), one shared-memory block per worker for the observation, the parent sending
one message per env per step and collecting the replies in order, worker-side auto-reset -- written from scratch
around the CPU oracle (tests/oracle_lib.py) or a no-op env (IPC-only ceiling).

TEST / BENCH INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg and tests import it; nothing under steppingstone_amd/ does.
"""
import multiprocessing
import os
import time

import numpy as np

OBS_DIM, ACT_DIM = 60, 21


class _NoopEnv:
    """Zero-cost stand-in: cached zero observation, done every 1000 steps."""

    def __init__(self):
        self.obs = np.zeros(OBS_DIM, np.float32)
        self.t = 0

    def reset(self):
        self.t = 0
        return self.obs

    def step(self, act):
        self.t += 1
        done = self.t >= 1000
        return self.obs, 0.0, done, ({"episode": {"r": 0.0, "l": self.t}} if done else {})


class _OracleEnv1:
    """One oracle environment (tests/oracle_lib.OracleEnv with num_envs=1, auto-reset off: the worker resets)."""

    def __init__(self, kind, seed, index):
        import oracle_lib as ol
        self.env = ol.OracleEnv(kind, 1, seed=seed, env_offset=index)
        self.env.set_auto_reset(0)

    def reset(self):
        return self.env.reset()[0]

    def step(self, act):
        obs, rew, done, info = self.env.step(np.asarray(act, np.float32).reshape(1, ACT_DIM))
        d = bool(done[0])
        return obs[0], float(rew[0]), d, ({"episode": {"r": float(info["ep_ret"][0]), "l": int(info["ep_len"][0])}} if d else {})


def _worker(pipe, parent_pipe, shm, kind, seed, index):
    os.environ["OMP_NUM_THREADS"] = "1"
    parent_pipe.close()
    env = _NoopEnv() if kind == "noop" else _OracleEnv1(kind, seed, index)
    dst = np.frombuffer(shm.get_obj(), dtype=np.float32)
    try:
        while True:
            cmd, data = pipe.recv()
            if cmd == "step":
                obs, rew, done, info = env.step(data)
                if done:
                    obs = env.reset()
                np.copyto(dst, obs)
                pipe.send((rew, done, info))
            elif cmd == "reset":
                np.copyto(dst, env.reset())
                pipe.send(None)
            elif cmd == "close":
                pipe.send(None)
                break
    except (KeyboardInterrupt, EOFError):
        pass


class ShmemFrontEnd:
    def __init__(self, kind, num_envs, seed=0, context="spawn"):
        ctx = multiprocessing.get_context(context)
        self.num_envs = int(num_envs)
        self.bufs = [ctx.Array("f", OBS_DIM) for _ in range(self.num_envs)]
        self.pipes, self.procs = [], []
        for i, buf in enumerate(self.bufs):
            parent, child = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(child, parent, buf, kind, seed, i), daemon=True)
            p.start()
            child.close()
            self.pipes.append(parent)
            self.procs.append(p)

    def _obs(self):
        return np.array([np.frombuffer(b.get_obj(), dtype=np.float32) for b in self.bufs])

    def reset(self):
        for p in self.pipes:
            p.send(("reset", None))
        for p in self.pipes:
            p.recv()
        return self._obs()

    def step(self, actions):
        for p, a in zip(self.pipes, actions):
            p.send(("step", a))
        outs = [p.recv() for p in self.pipes]
        rews, dones, infos = zip(*outs)
        return self._obs(), np.array(rews), np.array(dones), infos

    def close(self):
        for p in self.pipes:
            try:
                p.send(("close", None))
                p.recv()
            except (BrokenPipeError, EOFError):
                pass
        for pr in self.procs:
            pr.join(timeout=5)


def measure(kind, num_envs, seconds=5.0, warmup=5, seed=0):
    """env-steps/s of the front end: actions U(-1,1)^21 f32 from np.random.default_rng(0) (BASELINE.md C1 inputs)."""
    fe = ShmemFrontEnd(kind, num_envs, seed=seed)
    try:
        rng = np.random.default_rng(0)
        acts = rng.uniform(-1, 1, (8, num_envs, ACT_DIM)).astype(np.float32)
        fe.reset()
        for k in range(warmup):
            fe.step(acts[k % 8])
        t0 = time.perf_counter()
        steps = 0
        while True:
            fe.step(acts[steps % 8])
            steps += 1
            el = time.perf_counter() - t0
            if el > seconds:
                break
        return num_envs * steps / el, steps, el
    finally:
        fe.close()
