"""Fused PPO learner (extras/fused_learner/fused_ppo.py -> lib/libsslearner.so, MFMA f32 kernels) against torch autograd and
against the reference's own PPO.update golden vectors (tests/golden/harness_golden.npz).  `pytest -m gpu`."""
import numpy as np
import pytest

import os

torch = pytest.importorskip("torch")
# OUT OF SURVEY section 8's SCOPE (the PPO learner is SURVEY section 2 #5-#7, "no custom kernel warranted"): frozen, and since round 6
# OUTSIDE the product package and outside tests/ -- `python extras/fused_learner/build.py && python -m pytest extras/fused_learner -m gpu`.
import sys
_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
for _p in (_HERE, _ROOT, os.path.join(_ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
_LEARNER = os.path.join(_HERE, "lib", "libsslearner.so")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not os.path.exists(_LEARNER), reason="fused learner not built (python extras/fused_learner/build.py)")]


def _batch(R, dev, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)    # noqa: E731
    obs, act = r(R, 60), (r(R, 21) * 0.5).clamp(-1, 1)
    vpred, ret, adv = r(R, 1), r(R, 1), r(R, 1)
    return obs, act, vpred, ret, adv


def _torch_reference(ac, data, idx, clip=0.2, lr=3e-4, eps=1e-5, max_norm=2.0):
    """one minibatch step with autograd + clip_grad_norm_ + torch Adam; returns (grads by name, losses)"""
    from steppingstone_amd import ppo
    obs, act, vpred, ret, logp, adv = data
    opt = torch.optim.Adam(ac.parameters(), lr=lr, eps=eps)
    vl, al, ent = ppo.ppo_loss(ac, obs[idx], act[idx], vpred[idx], ret[idx], logp[idx], adv[idx], clip)
    opt.zero_grad()
    (vl + al).backward()
    grads = {n: p.grad.detach().clone() for n, p in ac.named_parameters()}
    torch.nn.utils.clip_grad_norm_(ac.parameters(), max_norm)
    opt.step()
    return grads, (float(vl.detach()), float(al.detach()), float(ent.detach()))


@pytest.mark.parametrize("E,B", [(1, 1024), (2, 512), (3, 96)])
def test_fused_step_matches_autograd_and_adam(E, B):
    import fused_ppo
    from steppingstone_amd import ppo
    dev = torch.device("cuda:0")
    R = 4096
    torch.manual_seed(1)
    ref = ppo.ActorCritic(num_ensembles=E).to(dev)
    with torch.no_grad():                                   # make every branch non-trivial: biases and log-std off their init
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    fus = ppo.ActorCritic(num_ensembles=E).to(dev)
    fus.load_state_dict(ref.state_dict())
    obs, act, vpred, ret, adv = _batch(R, dev)
    with torch.no_grad():
        _, logp0, _ = ref.evaluate_actions(obs, act)
    logp = logp0 + 0.3 * torch.randn_like(logp0)           # ratios on both sides of the clip range
    data = (obs, act, vpred, ret, logp, adv)
    idx = torch.randperm(R, device=dev)[:B]
    agent = fused_ppo.FusedPPO(fus, mini_batch_size=B, use_graph=False)
    w0 = agent.flat.clone()
    stats = agent.step_minibatch(tuple(t.contiguous() for t in data), idx).clone()
    g_fused = agent.grad()
    grads, losses = _torch_reference(ref, data, idx)
    # losses
    assert np.allclose(stats.cpu().numpy(), losses, rtol=2e-4, atol=2e-6), (stats.tolist(), losses)
    # gradients, tensor by tensor
    worst = 0.0
    for name, (off, shape) in agent.layout.items():
        n = int(np.prod(shape))
        gf, gt = g_fused[off:off + n].view(shape), grads[name]
        scale = float(gt.abs().max()) + 1e-12
        err = float((gf - gt).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-4, (name, err, scale)
    # the update itself: same weights as clip_grad_norm_ + Adam
    moved = float((agent.flat - w0).abs().max())
    assert moved > 1e-5
    for name, p in ref.named_parameters():
        off, shape = agent.layout[name]
        q = agent.flat[off:off + p.numel()].view(shape)
        # the first Adam step moves a weight by lr * g / (|g| + eps): where |g| is of the size of eps = 1e-5 the step follows the
        # gradient's relative error, hence 2 % of the step and not 2e-4
        assert float((q - p.detach()).abs().max()) < 2e-2 * moved + 1e-7, name
    # the module's own parameters ARE the flat vector
    for name, p in fus.named_parameters():
        off, shape = agent.layout[name]
        assert p.data_ptr() == agent.flat[off:].data_ptr()
    print("fused learner E=%d B=%d: worst relative gradient error %.2e, losses %s" % (E, B, worst, stats.tolist()))


def test_fused_update_matches_reference_golden():
    """the reference's PPO.update on the golden batch: three losses and every weight after one Adam step"""
    import os
    import fused_ppo
    from steppingstone_amd import ppo
    from test_ppo_golden import G, load_reference_weights
    dev = torch.device("cuda:0")
    ac = ppo.ActorCritic(num_ensembles=2)
    load_reference_weights(ac, "ppo_w0/")
    ac = ac.to(dev)
    T, N = G["ppo_act"].shape[:2]
    t = lambda k: torch.from_numpy(G["ppo_" + k]).to(dev)  # noqa: E731
    roll = ppo.Rollouts(T, N, dev)
    roll.obs.copy_(t("obs")); roll.actions.copy_(t("act")); roll.logp.copy_(t("old_logp"))
    roll.value_preds.copy_(t("vpred")); roll.returns.copy_(t("returns"))
    agent = fused_ppo.FusedPPO(ac, clip_param=0.2, ppo_epoch=1, mini_batch_size=T * N, lr=3e-4, eps=1e-5, max_grad_norm=2.0,
                               use_graph=False)
    vl, al, ent = agent.update(roll)
    assert np.allclose([vl, al, ent], G["ppo_losses"], rtol=1e-4, atol=1e-5), ([vl, al, ent], G["ppo_losses"])
    ref = ppo.ActorCritic(num_ensembles=2)
    load_reference_weights(ref, "ppo_w1/")
    for (name, p), (_, q) in zip(ac.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(p.cpu(), q, rtol=1e-3, atol=1e-5), name


def test_fused_update_equals_torch_update_over_many_minibatches_and_replays_as_a_graph():
    import fused_ppo
    from steppingstone_amd import ppo
    dev = torch.device("cuda:0")
    T, N, mb = 8, 512, 1024
    finals = []
    for kind in ("torch", "fused_eager", "fused_graph"):
        torch.manual_seed(7)
        ac = ppo.ActorCritic(num_ensembles=2).to(dev)
        roll = ppo.Rollouts(T, N, dev)
        g = torch.Generator(device="cpu").manual_seed(3)
        roll.obs.copy_(torch.randn(T + 1, N, 60, generator=g)); roll.actions.copy_(torch.randn(T, N, 21, generator=g) * 0.3)
        roll.value_preds.copy_(torch.randn(T + 1, N, 1, generator=g)); roll.returns.copy_(torch.randn(T + 1, N, 1, generator=g))
        with torch.no_grad():
            _, lp, _ = ac.evaluate_actions(roll.obs[:-1].reshape(-1, 60), roll.actions.reshape(-1, 21))
        roll.logp.copy_(lp.view(T, N, 1) + 0.2 * torch.randn(T, N, 1, generator=g).to(dev))
        if kind == "torch":
            agent = ppo.PPO(ac, ppo_epoch=2, mini_batch_size=mb)
        else:
            agent = fused_ppo.FusedPPO(ac, ppo_epoch=2, mini_batch_size=mb, use_graph=(kind == "fused_graph"))
        torch.manual_seed(11)                               # same minibatch permutations
        losses = agent.update(roll)
        torch.manual_seed(12)
        losses2 = agent.update(roll)                        # a second update: graph replay with a re-used capture
        finals.append((torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).clone(), losses, losses2))
    w_t, w_e, w_g = finals[0][0], finals[1][0], finals[2][0]
    assert torch.equal(w_e, w_g)                            # graph replay == eager launches, bit for bit
    step = 16 * 3e-4                                        # 16 Adam steps can move a weight by at most ~16 lr
    assert float((w_e - w_t).abs().max()) < 0.05 * step, float((w_e - w_t).abs().max())
    assert np.allclose(finals[0][1], finals[1][1], rtol=2e-3, atol=1e-5) and np.allclose(finals[0][2], finals[1][2], rtol=2e-3, atol=1e-5)


def test_fused_step_with_mirror_augmentation_matches_torch():
    """use_mirror (common/envs_utils.py:687-740 inside PPO.update): the doubled minibatch is formed inside the kernels (mirrored
    rows read through permutation / sign tables); gradients and losses equal autograd on harness.mirror_batch's batch."""
    from steppingstone_amd import _lib, fused_ppo, harness, ppo
    dev = torch.device("cuda:0")
    R, B, E = 4096, 512, 2
    idxs = _lib.mirror_indices()
    torch.manual_seed(3)
    ref = ppo.ActorCritic(num_ensembles=E).to(dev)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    fus = ppo.ActorCritic(num_ensembles=E).to(dev)
    fus.load_state_dict(ref.state_dict())
    obs, act, vpred, ret, adv = _batch(R, dev, seed=5)
    with torch.no_grad():
        _, logp0, _ = ref.evaluate_actions(obs, act)
    logp = logp0 + 0.3 * torch.randn_like(logp0)
    idx = torch.randperm(R, device=dev)[:B]
    agent = fused_ppo.FusedPPO(fus, mini_batch_size=B, mirror_indices=idxs, use_graph=False)
    stats = agent.step_minibatch((obs, act, vpred, ret, logp, adv), idx).clone()
    g_fused = agent.grad()
    # torch: the reference's order of operations -- gather, mirror (doubles the batch), loss
    o2, a2 = harness.mirror_batch(obs[idx], act[idx], [torch.as_tensor(i, dtype=torch.long, device=dev) for i in idxs])
    rep = lambda t: t[idx].repeat((2, 1))    # noqa: E731
    vl, al, ent = ppo.ppo_loss(ref, o2, a2, rep(vpred), rep(ret), rep(logp), rep(adv), 0.2)
    (vl + al).backward()
    assert np.allclose(stats.cpu().numpy(), [float(vl.detach()), float(al.detach()), float(ent.detach())], rtol=2e-4, atol=2e-6)
    for name, p in ref.named_parameters():
        off, shape = agent.layout[name]
        gf = g_fused[off:off + p.numel()].view(shape)
        scale = float(p.grad.abs().max()) + 1e-12
        # weights seeded with 3, not 2: with seed 2 one sample's pre-activation of critics.1.4 neuron 247 lies within rounding of
        # zero, its ReLU derivative flips between the two implementations, and that sample's delta differs in critics.1.{0,2,4}
        # (1.3e-3 of the largest gradient; later layers and the other nets agree to 1e-6) -- tools/_dbg history in DESIGN 4.3
        assert float((gf - p.grad).abs().max()) / scale < 2e-4, (name, float((gf - p.grad).abs().max()), scale)


def test_fused_data_parallel_halves_equal_the_global_minibatch():
    """configs[4] (one process per GPU): ssl_grad on each rank's minibatch, the sum of the gradients (what the RCCL all-reduce
    leaves on every rank), ssl_apply with 1 / world.  Two 'ranks' are played on one GPU: the result must be the torch step
    (autograd, clip_grad_norm_, Adam) on the concatenated minibatch, and both ranks end with the same weights."""
    import ctypes as C
    import fused_ppo
    from steppingstone_amd import ppo
    dev = torch.device("cuda:0")
    R, B, E = 4096, 256, 2
    torch.manual_seed(4)
    ref = ppo.ActorCritic(num_ensembles=E).to(dev)
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.05 * torch.randn_like(p))
    obs, act, vpred, ret, adv = _batch(R, dev, seed=9)
    with torch.no_grad():
        _, logp0, _ = ref.evaluate_actions(obs, act)
    logp = logp0 + 0.3 * torch.randn_like(logp0)
    data = tuple(t.contiguous() for t in (obs, act, vpred, ret, logp, adv))
    perm = torch.randperm(R, device=dev)
    idx = [perm[:B].contiguous(), perm[B:2 * B].contiguous()]
    ranks = []
    for r in range(2):
        ac = ppo.ActorCritic(num_ensembles=E).to(dev)
        ac.load_state_dict(ref.state_dict())
        ranks.append(fused_ppo.FusedPPO(ac, mini_batch_size=B, use_graph=False, data_parallel=True))
    p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for a, ix in zip(ranks, idx):
        a.step_t.add_(1.0)
        fused_ppo._check(a.lib.ssl_grad(a.h, p(a.flat), p(obs), p(act), p(logp), p(adv), p(ret), p(vpred), p(ix), B, 0.2, 0, p(a.stats),
                                        p(a.gbuf), st, None, None, None, None))
    total = ranks[0].gbuf + ranks[1].gbuf                    # the all-reduce (sum)
    for a in ranks:
        a.gbuf.copy_(total)
        fused_ppo._check(a.lib.ssl_apply(a.h, p(a.flat), p(a.m), p(a.v), p(a.lr_t), p(a.step_t), p(a.gbuf), 0.5, 2.0, 1e-5, st))
    torch.cuda.synchronize()
    assert torch.equal(ranks[0].flat, ranks[1].flat)         # same inputs, deterministic kernels: the replicas stay in step
    w0 = torch.cat([q.detach().reshape(-1) for q in ref.parameters()]).clone()
    grads, losses = _torch_reference(ref, data, torch.cat(idx))
    a = ranks[0]
    moved = float((torch.cat([q.detach().reshape(-1) for q in ref.parameters()]) - w0).abs().max())
    assert moved > 1e-5
    for name, q in ref.named_parameters():
        off, shape = a.layout[name]
        n = q.numel()
        scale = float(grads[name].abs().max()) + 1e-12
        assert float((a.gbuf[off:off + n].view(shape) - grads[name]).abs().max()) / scale < 2e-4, name   # gbuf now holds the mean
        assert float((a.flat[off:off + n].view(shape) - q.detach()).abs().max()) < 2e-2 * moved + 1e-7, name
    # the mean of the two ranks' losses is the loss of the global minibatch
    s = 0.5 * (ranks[0].stats + ranks[1].stats)
    assert np.allclose(s.cpu().numpy(), losses, rtol=2e-4, atol=2e-6)
    # and the forced two-call path on one rank is the one-call step
    one = ppo.ActorCritic(num_ensembles=E).to(dev); two = ppo.ActorCritic(num_ensembles=E).to(dev)
    one.load_state_dict(ranks[0].ac.state_dict()); two.load_state_dict(ranks[0].ac.state_dict())
    a1 = fused_ppo.FusedPPO(one, mini_batch_size=B, use_graph=False)
    a2 = fused_ppo.FusedPPO(two, mini_batch_size=B, use_graph=False, data_parallel=True)
    a1.step_minibatch(data, idx[0]); a2.step_minibatch(data, idx[0])
    torch.cuda.synchronize()
    assert float((a1.flat - a2.flat).abs().max()) < 1e-7
