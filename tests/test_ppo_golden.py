"""The device-resident PPO pieces (steppingstone_amd/ppo.py) against the reference's own Policy + PPO.update run on
a fixed batch from fixed weights (tools/make_golden.py section 5), plus an end-to-end smoke of the driver on the oracle-
backed env.  CPU only."""
import os

import numpy as np
import pytest
import torch

from steppingstone_amd import ppo

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "harness_golden.npz"))


def load_reference_weights(ac, prefix):
    sd = {}
    for k in G.files:
        if not k.startswith(prefix):
            continue
        name = k[len(prefix):]
        t = torch.from_numpy(G[k])
        if name == "dist.logstd._bias":
            sd["logstd"] = t.reshape(-1)
        elif name.startswith("c"):
            i, rest = name[1:].split(".", 1)
            sd["critics.%s.%s" % (i, rest)] = t
        else:
            sd[name] = t
    ac.load_state_dict(sd, strict=True)


def check_one_ppo_update(device, **tol):
    """algorithms/ppo.py:40-108 on the golden batch: three losses and every weight after one Adam step.  Shared by the
    CPU test below and the -m gpu test (tests/test_gpu_golden.py runs it on cuda, eagerly and through the hipGraph step)."""
    device = torch.device(device)
    ac = ppo.ActorCritic(num_ensembles=2)
    load_reference_weights(ac, "ppo_w0/")
    ac = ac.to(device)
    T, N = G["ppo_act"].shape[:2]
    t = lambda k: torch.from_numpy(G["ppo_" + k]).to(device)  # noqa: E731
    roll = ppo.Rollouts(T, N, device)
    roll.obs.copy_(t("obs")); roll.actions.copy_(t("act")); roll.logp.copy_(t("old_logp"))
    roll.value_preds.copy_(t("vpred")); roll.returns.copy_(t("returns"))
    agent = ppo.PPO(ac, clip_param=0.2, ppo_epoch=1, mini_batch_size=T * N, lr=3e-4, eps=1e-5, max_grad_norm=2.0,
                    use_graph=tol.get("use_graph", False))
    if tol.get("use_graph"):
        agent._warm = 3                                   # capture on the very first (and only) minibatch step
    vl, al, ent = agent.update(roll)
    assert np.allclose([vl, al, ent], G["ppo_losses"], rtol=tol.get("loss_rtol", 2e-5), atol=tol.get("loss_atol", 2e-6)), \
        ([vl, al, ent], G["ppo_losses"])
    ref = ppo.ActorCritic(num_ensembles=2)
    load_reference_weights(ref, "ppo_w1/")
    for (name, p), (_, q) in zip(ac.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(p.cpu(), q, rtol=tol.get("w_rtol", 1e-4), atol=tol.get("w_atol", 2e-6)), name
    # the step moved the weights at all
    moved = ppo.ActorCritic(num_ensembles=2)
    load_reference_weights(moved, "ppo_w0/")
    assert (moved.actor.fc1.weight - ac.actor.fc1.weight.cpu()).abs().max() > 1e-5


def test_one_ppo_update_matches_reference():
    check_one_ppo_update("cpu")


def test_actor_critic_shapes_and_logp():
    torch.manual_seed(0)
    ac = ppo.ActorCritic(num_ensembles=3)
    obs = torch.randn(7, 60)
    v, a, lp = ac.act(obs)
    assert v.shape == (7, 1) and a.shape == (7, 21) and lp.shape == (7, 1)
    vals, lp2, ent = ac.evaluate_actions(obs, a)
    assert vals.shape == (7, 3) and torch.allclose(lp, lp2, atol=1e-5)
    d = torch.distributions.Normal(ac.actor(obs), ac.logstd.exp())
    assert torch.allclose(d.log_prob(a).sum(-1, keepdim=True), lp2, atol=1e-5)
    assert torch.allclose(d.entropy().sum(-1).mean(), ent, atol=1e-5)
    _, a_det, _ = ac.act(obs, deterministic=True)
    assert torch.equal(a_det, ac.actor(obs))


def test_driver_runs_on_oracle_backed_env():
    from oracle_backend import OracleBackend
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 16
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=1, return_numpy=False, backend=OracleBackend(1, n, 1))
    logs = []
    ac, hist = ppo.train(envs, num_updates=2, num_steps=8, num_ensembles=2, ppo_epoch=2, mini_batch_size=64,
                         use_mirror=True, log=logs.append)
    assert len(hist) == 2 and hist[-1]["total_num_steps"] == 2 * 8 * n
    assert all(np.isfinite([h["value_loss"], h["action_loss"], h["entropy"]]).all() for h in hist)
    assert hist[-1]["curriculum"] == 0


def test_adaptive_sampler_grid():
    from oracle_backend import OracleBackend
    from steppingstone_amd.envs import SteppingStoneVecEnv
    import oracle_lib as ol
    n = 12
    be = OracleBackend(0, n, 3)
    envs = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=3, return_numpy=False, backend=be)
    envs.update_curriculum(5)
    torch.manual_seed(0)
    ac = ppo.ActorCritic(num_ensembles=2)
    # make the target advance quickly: start every env standing on its target stone
    orig_reset = envs.reset

    def reset_on_target():
        orig_reset()
        st = be.o.get_state()
        st[:, 0] = st[:, 65 + 6]
        be.o.set_state(st)
        return envs.get_obs()

    envs.reset = reset_on_target
    for mode in ("threshold", "adaptive"):
        p = ppo.sampling_probs_from_values(ac, envs, mode=mode, events=3, max_steps=6)
        assert p is not None and p.shape == (11, 11) and p.dtype == np.float64
        assert abs(p.sum() - 1.0) < 1e-6 and (p > 0).all()      # softmax is evaluated in fp32 like the reference
        envs.update_sample_prob(np.repeat(p[None], n, axis=0))      # what train.py:267-271 does


REF_MODELS = "/root/reference/playground/models/"


@pytest.mark.parametrize("name,dims,critic_prefixes", [
    ("mocca_envs:Walker3DStepperEnv-v0_latest.pt", (60, 21), ["critic."]),       # older layout: one module `critic`
    ("mocca_envs:Walker3DStepperEnv-v0_best.pt", (60, 21), ["critic."]),
    ("mocca_envs:Walker3DStepperEnv-v0_base.pt", (60, 21), ["critic."]),
    ("mocca_envs:MikeStepperEnv-v0_latest.pt", (60, 21), ["c0."]),               # current layout: ensemble modules c0, c1, ... (controller.py:94-95)
    ("CassieStepper-v1_base.pt", (51, 10), ["c0.", "c1."]),                       # train.py:37's default env: other dims, two critics
])
def test_reference_checkpoint_loader(name, dims, critic_prefixes):
    """steppingstone_amd.legacy_checkpoint: EVERY policy the reference ships (legacy torch.save of the pickled Policy module) read
    with the restricted unpickler -- no reference code executed -- and mapped onto ActorCritic; the deterministic action and every
    ensemble member's value must equal a numpy forward pass over the raw arrays of the file (this container only: the files live
    under /root/reference)."""
    import os
    path = REF_MODELS + name
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    from steppingstone_amd import legacy_checkpoint as lc
    ac = lc.load_reference_checkpoint(path)
    obj, st = lc.read_legacy(path)
    w = lc.tensors_of(obj, st)
    assert ac.actor.fc1.weight.shape[1] == dims[0] and ac.logstd.numel() == dims[1] and len(ac.critics) == len(critic_prefixes)
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, dims[0])).astype(np.float32)
    h = x.astype(np.float64)
    for i, act in ((1, "softsign"), (2, "softsign"), (3, "softsign"), (4, "relu"), (5, "relu")):
        h = h @ w["actor.fc%d.weight" % i].T.astype(np.float64) + w["actor.fc%d.bias" % i]
        h = h / (1 + np.abs(h)) if act == "softsign" else np.maximum(h, 0)
    ref = np.tanh(h @ w["actor.out.weight"].T.astype(np.float64) + w["actor.out.bias"])
    vals = []
    for pre in critic_prefixes:
        h = x.astype(np.float64)
        for li in (0, 2, 4, 6):
            h = np.maximum(h @ w["%s%d.weight" % (pre, li)].T.astype(np.float64) + w["%s%d.bias" % (pre, li)], 0)
        vals.append(h @ w[pre + "8.weight"].T.astype(np.float64) + w[pre + "8.bias"])
    with torch.no_grad():
        v, a, _ = ac.act(torch.from_numpy(x), deterministic=True)
        ve = ac.get_ensemble_values(torch.from_numpy(x))
    assert np.abs(a.numpy() - ref).max() < 1e-5
    assert np.abs(ve.numpy() - np.concatenate(vals, axis=1)).max() < 1e-3 * max(1.0, np.abs(np.concatenate(vals, axis=1)).max())
    assert np.allclose(ac.logstd.detach().numpy(), w["dist.logstd._bias"].reshape(-1)) and v.shape == (5, 1)


def test_legacy_reader_rejects_other_files(tmp_path):
    """Not a legacy torch.save stream -> ValueError (the magic number is checked), a truncated one -> an error, never garbage."""
    import os
    from steppingstone_amd import legacy_checkpoint as lc
    p = tmp_path / "x.pt"
    torch.save({"a": torch.zeros(3)}, str(p))                       # the zip format
    with pytest.raises(Exception):
        lc.read_legacy(str(p))
    src = REF_MODELS + "mocca_envs:MikeStepperEnv-v0_latest.pt"
    if os.path.exists(src):
        raw = open(src, "rb").read()
        q = tmp_path / "cut.pt"
        q.write_bytes(raw[: len(raw) - 1000])
        with pytest.raises(Exception):
            lc.read_legacy(str(q))
    # a crafted file whose HEADER records name a global (the classic os.system pickle) is refused before anything runs
    import pickle

    class Boom:
        def __reduce__(self):
            return (os.mkdir, (str(tmp_path / "pwned"),))
    for pos in (1, 2):
        recs = [lc._MAGIC, 1001, {"little_endian": True}]
        recs[pos] = Boom()
        evil = tmp_path / ("evil%d.pt" % pos)
        with open(evil, "wb") as f:
            for r in recs:
                pickle.dump(r, f, protocol=2)
        with pytest.raises(pickle.UnpicklingError, match="refused"):
            lc.read_legacy(str(evil))
        assert not (tmp_path / "pwned").exists()


def test_export_in_the_form_the_reference_loads_back(tmp_path):
    """tools/export_reference_checkpoint.py: a checkpoint of this package becomes a pickled `Policy` module of the REFERENCE's own classes
    (playground/train.py:551 saves one, playground/enjoy.py:148 `torch.load`s one).  Here, where the reference checkout is present: the
    exported file loads with plain torch.load, is an instance of common.controller.Policy, and its deterministic action, ensemble values
    and log-std equal this package's; and it reads back through legacy_checkpoint-free means only -- the round trip ours -> theirs."""
    import os
    import subprocess
    import sys
    if not os.path.exists("/root/reference/common/controller.py"):
        pytest.skip("reference checkout not present")
    from steppingstone_amd import ppo
    torch.manual_seed(4)
    ac = ppo.ActorCritic(num_ensembles=2)
    with torch.no_grad():
        ac.logstd.copy_(torch.linspace(-2.0, -1.0, 21))
    ours = str(tmp_path / "ours.pt")
    ppo.save_checkpoint(ac, ours, update=7)
    out = str(tmp_path / "theirs.pt")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "export_reference_checkpoint.py"), ours, "--reference", "/root/reference",
                        "--out", out], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    # load it the way playground/enjoy.py does, in a process that can import the reference's classes
    code = ("import sys, types, torch\n"
            "sys.dont_write_bytecode = True\n"
            "sys.path.insert(0, '/root/reference')\n"
            "g, sp = types.ModuleType('gym'), types.ModuleType('gym.spaces')\n"
            "sp.Box, sp.Dict, sp.MultiDiscrete = type('Box', (), {}), type('Dict', (dict,), {}), type('MultiDiscrete', (), {})\n"
            "g.spaces = sp; sys.modules.update({'gym': g, 'gym.spaces': sp})\n"
            "m = torch.load(%r, weights_only=False)\n"
            "from common.controller import Policy\n"
            "assert isinstance(m, Policy) and len(m.critics) == 2\n"
            "x = torch.linspace(-1, 1, 5 * 60).reshape(5, 60)\n"
            "v, a, lp, _ = m.act(x, None, None, deterministic=True)\n"
            "torch.save({'a': a, 'v': m.get_ensemble_values(x, None, None), 'logstd': m.dist.logstd._bias.reshape(-1)}, %r)\n"
            % (out, str(tmp_path / "ref_out.pt")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    ref = torch.load(str(tmp_path / "ref_out.pt"))
    x = torch.linspace(-1, 1, 5 * 60).reshape(5, 60)
    with torch.no_grad():
        _, a, _ = ac.act(x, deterministic=True)
        ve = ac.get_ensemble_values(x)
    assert torch.allclose(ref["a"], a, atol=1e-6) and torch.allclose(ref["v"], ve, atol=1e-5)
    assert torch.equal(ref["logstd"], ac.logstd.detach())
