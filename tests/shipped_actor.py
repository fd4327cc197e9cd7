"""The reference's shipped actors as plain arrays (tests/golden/shipped_actor_<kind>.npz, made by tools/make_golden_policy.py) and the
rollout both behavioural tests share.  TEST INFRASTRUCTURE.  Architecture: common/controller.py:217-261 (SoftsignActor: three softsign
layers, two relu layers, tanh output); `playground/enjoy.py:143-235` runs it deterministically."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_actor(kind, device="cpu"):
    import torch
    w = np.load(os.path.join(GOLD, "shipped_actor_%s.npz" % kind))
    W = {k: torch.from_numpy(w[k]).to(device) for k in w.files if k != "source"}

    def actor(x):
        for i, act in ((1, "softsign"), (2, "softsign"), (3, "softsign"), (4, "relu"), (5, "relu")):
            x = torch.nn.functional.linear(x, W["fc%d.weight" % i], W["fc%d.bias" % i])
            x = torch.nn.functional.softsign(x) if act == "softsign" else torch.relu(x)
        return torch.tanh(torch.nn.functional.linear(x, W["out.weight"], W["out.bias"]))
    return actor


def walk(env, actor, steps, to_tensor, n):
    """Deterministic actor in `env` (VecEnv protocol, auto-reset on): stones reached beyond the start and length of the FIRST episode of
    each env (an env still walking after `steps` counts with where it is)."""
    import torch
    obs = env.reset()
    alive = np.ones(n, bool)
    reached, length = np.ones(n), np.full(n, float(steps))
    last_n = np.ones(n)
    for t in range(steps):
        with torch.no_grad():
            a = actor(to_tensor(obs))
        obs, rew, done, info = env.step(a)
        d = np.asarray(done.cpu() if hasattr(done, "cpu") else done).astype(bool)
        sr = info["steps_reached"]
        sr = np.asarray(sr.cpu() if hasattr(sr, "cpu") else sr)
        el = info["ep_len"]
        el = np.asarray(el.cpu() if hasattr(el, "cpu") else el)
        fin = alive & d
        reached[fin] = sr[fin]
        length[fin] = el[fin]
        alive &= ~fin
        if not alive.any():
            break
    return reached - 1.0, length, alive
