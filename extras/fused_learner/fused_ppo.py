"""Fused PPO learner: the minibatch loop body of algorithms/ppo.py:55-100 (evaluate_actions, clipped surrogate + value
loss, backward, clip_grad_norm_, Adam) as 15 launches of hand-written gfx950 kernels (extras/fused_learner/
ss_learner.hip, steppingstone_learner.h beside it; exact-f32 MFMA GEMMs) instead of ~90 launches of generic framework
kernels.  Drop-in for steppingstone_amd.ppo.PPO (mirror augmentation and the data-parallel update included):

    agent = FusedPPO(actor_critic, ppo_epoch=10, mini_batch_size=1024, lr=3e-4, ...)
    value_loss, action_loss, entropy = agent.update(rollouts)          # same contract as PPO.update

The ActorCritic's parameters are re-bound as views into one flat f32 vector (the layout of steppingstone_learner.h), so
policy inference (`ac.act`), checkpoints and state_dicts keep working on the very memory the kernels update.
There is no CPU path: the library raises without a GPU.
"""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEPPINGSTONE_LEARNER_LIB") or os.path.join(HERE, "lib", "libsslearner.so")
SYMBOLS = ["ssl_last_error", "ssl_num_params", "ssl_create", "ssl_destroy", "ssl_step", "ssl_step_mirror", "ssl_grad", "ssl_apply",
           "ssl_debug_grad"]
OBS, HID, ACT = 60, 256, 21

_lib = None


class FusedLearnerError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FusedLearnerError("libsslearner.so is missing (%s): build it with `python extras/fused_learner/build.py`" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
        lib.ssl_last_error.restype = C.c_char_p
        lib.ssl_num_params.argtypes = [i32]
        lib.ssl_num_params.restype = C.c_int64
        lib.ssl_create.argtypes = [C.POINTER(vp), C.c_int, i32, i32]
        lib.ssl_destroy.argtypes = [vp]
        lib.ssl_destroy.restype = None
        lib.ssl_step.argtypes = [vp] * 13 + [i32, f32, f32, f32, i32, vp, vp]
        lib.ssl_step_mirror.argtypes = [vp] * 13 + [i32, f32, f32, f32, i32, vp, vp] + [vp] * 4
        lib.ssl_grad.argtypes = [vp] * 9 + [i32, f32, i32, vp, vp, vp] + [vp] * 4
        lib.ssl_apply.argtypes = [vp] * 7 + [f32, f32, f32, vp]
        lib.ssl_debug_grad.argtypes = [vp]
        lib.ssl_debug_grad.restype = vp
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        msg = load().ssl_last_error()
        raise FusedLearnerError("libsslearner error %d: %s" % (rc, msg.decode() if msg else "?"))


def _al4(x):
    return (x + 3) // 4 * 4


def layout(num_ensembles):
    """name -> (offset, shape) of every parameter inside the flat vector (same rule as make_net in ss_learner.hip: log-std,
    actor layers, critic layers; weight then bias; every tensor at a multiple of 4 floats)."""
    out, o = {}, 0
    out["logstd"] = (o, (ACT,))
    o = _al4(o + ACT)
    dims = [("fc1", OBS, HID), ("fc2", HID, HID), ("fc3", HID, HID), ("fc4", HID, HID), ("fc5", HID, HID), ("out", HID, ACT)]
    for name, n_in, n_out in dims:
        out["actor.%s.weight" % name] = (o, (n_out, n_in))
        b = _al4(o + n_in * n_out)
        out["actor.%s.bias" % name] = (b, (n_out,))
        o = _al4(b + n_out)
    for e in range(num_ensembles):
        for k, (n_in, n_out) in zip((0, 2, 4, 6, 8), ((OBS, HID), (HID, HID), (HID, HID), (HID, HID), (HID, 1))):
            out["critics.%d.%d.weight" % (e, k)] = (o, (n_out, n_in))
            b = _al4(o + n_in * n_out)
            out["critics.%d.%d.bias" % (e, k)] = (b, (n_out,))
            o = _al4(b + n_out)
    return out, o


class FusedPPO:
    """Same constructor arguments and update() contract as steppingstone_amd.ppo.PPO (defaults of playground/train.py:72-82)."""

    def __init__(self, ac, clip_param=0.2, ppo_epoch=10, mini_batch_size=1024, value_loss_coef=1.0, entropy_coef=0.0, lr=3e-4,
                 eps=1e-5, max_grad_norm=2.0, use_clipped_value_loss=False, mirror_indices=None, use_graph=True,
                 data_parallel=None):
        """data_parallel: None = follow torch.distributed (world size > 1: every minibatch step is ssl_grad on the rank's own
        minibatch, ONE all-reduce of the flat gradient, ssl_apply with 1 / world -- the reference's update on the global
        minibatch); True forces the two-call path on a single rank as well (tests)."""
        if value_loss_coef != 1.0 or entropy_coef != 0.0:
            raise FusedLearnerError("the fused learner implements the reference's defaults value_loss_coef=1, entropy_coef=0")
        if mini_batch_size % 32:
            raise FusedLearnerError("mini_batch_size must be a multiple of 32")
        dev = next(ac.parameters()).device
        if dev.type != "cuda":
            raise FusedLearnerError("the fused learner runs on an MI355X only (parameters are on %s)" % dev)
        self.lib = load()
        self.ac, self.device = ac, dev
        self.clip_param, self.ppo_epoch, self.mini_batch_size = clip_param, ppo_epoch, mini_batch_size
        self.max_grad_norm, self.eps, self.use_clipped_value_loss = max_grad_norm, eps, use_clipped_value_loss
        E = len(ac.critics)
        lay, n = layout(E)
        assert n == self.lib.ssl_num_params(E), "flat layout mismatch between fused_ppo.py and ss_learner.hip"
        self.layout, self.n_params = lay, n
        self.flat = torch.zeros(n, device=dev)
        for name, p in ac.named_parameters():
            off, shape = lay[name]
            assert tuple(p.shape) == tuple(shape), (name, tuple(p.shape), shape)
            view = self.flat[off:off + p.numel()].view(shape)
            view.copy_(p.data)
            p.data = view                                   # the module now reads and writes the flat vector
        self.m, self.v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self.lr_t = torch.tensor(float(lr), device=dev)
        self.step_t = torch.zeros((), device=dev)
        self.stats = torch.zeros(3, device=dev)
        # mirror augmentation (common/envs_utils.py:687-740): column c of a mirrored row = sgn[src] * x[src], src = perm[c]
        self.mirror = None
        if mirror_indices is not None:
            neg_o, right_o, left_o, neg_a, right_a, left_a = [torch.as_tensor(i, dtype=torch.long).cpu() for i in mirror_indices]

            def tables(neg, right, left, n):
                perm, sgn = torch.arange(n), torch.ones(n)
                perm[right], perm[left] = left.clone(), right.clone()
                sgn[neg] = -1.0
                return perm.to(torch.int32).to(dev), sgn.to(dev)

            self.mirror = tables(neg_o, right_o, left_o, OBS) + tables(neg_a, right_a, left_a, ACT)
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _check(self.lib.ssl_create(C.byref(h), dev.index if dev.index is not None else torch.cuda.current_device(), E,
                                       mini_batch_size * (2 if self.mirror else 1)))
        self.h = h
        import torch.distributed as dist
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.data_parallel = (self.world > 1) if data_parallel is None else bool(data_parallel)
        self.gbuf = torch.zeros(n, device=dev) if self.data_parallel else None
        self.use_graph = bool(use_graph) and self.world == 1        # the collective is not captured
        self._graph, self._static_idx, self._data_ptrs, self._warm = None, None, None, 0

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.ssl_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_lr(self, lr):
        self.lr_t.fill_(float(lr))

    # -- one minibatch
    def _launch(self, data, idx):
        obs, act, vpred, ret, logp, adv = data
        p = lambda t: C.c_void_p(t.data_ptr())   # noqa: E731
        self.step_t.add_(1.0)
        if self.data_parallel:
            st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            tabs = [p(t) for t in self.mirror] if self.mirror else [None] * 4
            _check(self.lib.ssl_grad(self.h, p(self.flat), p(obs), p(act), p(logp), p(adv), p(ret), p(vpred), p(idx), int(idx.numel()),
                                     float(self.clip_param), 1 if self.use_clipped_value_loss else 0, p(self.stats), p(self.gbuf), st,
                                     *tabs))
            if self.world > 1:
                import torch.distributed as dist
                dist.all_reduce(self.gbuf)                  # RCCL, one 1.3 MB message per minibatch step
            _check(self.lib.ssl_apply(self.h, p(self.flat), p(self.m), p(self.v), p(self.lr_t), p(self.step_t), p(self.gbuf),
                                      1.0 / self.world, float(self.max_grad_norm), float(self.eps), st))
            return self.stats
        args = [self.h, p(self.flat), p(self.m), p(self.v), p(self.lr_t), p(self.step_t), p(obs), p(act), p(logp), p(adv), p(ret),
                p(vpred), p(idx), int(idx.numel()), float(self.clip_param), float(self.max_grad_norm), float(self.eps),
                1 if self.use_clipped_value_loss else 0, p(self.stats), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)]
        if self.mirror:
            _check(self.lib.ssl_step_mirror(*args, *[p(t) for t in self.mirror]))
        else:
            _check(self.lib.ssl_step(*args))
        return self.stats

    def step_minibatch(self, data, idx):
        """data: the six flat, contiguous f32 rollout tensors (obs [R,60], act [R,21], vpred, ret, old_logp, adv [R] or
        [R,1]); idx: int64 [mini_batch_size].  With use_graph the 15 launches (+ the step counter) replay as one hipGraph;
        the graph is re-captured when the rollout tensors move."""
        ptrs = tuple(t.data_ptr() for t in data)
        if not self.use_graph:
            return self._launch(data, idx)
        if self._graph is None or ptrs != self._data_ptrs or idx.numel() != self._static_idx.numel():
            self._warm += 1
            if self._warm <= 1 and self._graph is None:           # first call eagerly (module load, lazy init)
                return self._launch(data, idx)
            self._static_idx = torch.zeros_like(idx)
            self._static_idx.copy_(idx)
            g = torch.cuda.CUDAGraph()
            step_before = self.step_t.clone()
            m0, v0, f0 = self.m.clone(), self.v.clone(), self.flat.clone()
            with torch.cuda.graph(g):
                self._launch(data, self._static_idx)
            # capture does not execute: nothing to undo (the clones keep the allocator from reusing live memory)
            del step_before, m0, v0, f0
            self._graph, self._data_ptrs = g, ptrs
        self._static_idx.copy_(idx)
        self._graph.replay()
        return self.stats

    def update(self, roll):
        """algorithms/ppo.py:40-108: advantages normalised over the whole rollout, ppo_epoch passes of random minibatches.
        With use_graph one hipGraph holds a WHOLE epoch (every minibatch step of it reads its slice of a static permutation
        buffer and adds its losses to a device accumulator), so the host launches ppo_epoch graphs per update instead of
        ppo_epoch x num_mini_batch steps."""
        from steppingstone_amd.ppo import _global_mean_std
        adv = roll.returns[:-1] - roll.value_preds[:-1]
        mean, std = _global_mean_std(adv)                   # over every rank's transitions
        adv = (adv - mean) / (std + 1e-5)
        T, N = roll.rewards.shape[:2]
        R = T * N
        flat = lambda t, w: t.reshape(R, w).contiguous()   # noqa: E731
        if getattr(self, "_adv_buf", None) is None or self._adv_buf.numel() != R:
            self._adv_buf = torch.empty(R, device=self.device)
            self._perm = torch.zeros(R, dtype=torch.long, device=self.device)
            self._tot = torch.zeros(3, device=self.device)
            self._epoch_graph, self._epoch_ptrs = None, None
        self._adv_buf.copy_(adv.reshape(R))                # stable address for the captured graph
        data = (flat(roll.obs[:-1], OBS), flat(roll.actions, ACT), flat(roll.value_preds[:-1], 1), flat(roll.returns[:-1], 1),
                flat(roll.logp, 1), self._adv_buf)
        mb = self.mini_batch_size
        nsteps = R // mb                                    # full minibatches only (the kernels need multiples of 32)
        self._tot.zero_()

        def epoch():
            for s in range(nsteps):
                self._tot.add_(self._launch(data, self._perm[s * mb:(s + 1) * mb]))

        for _ in range(self.ppo_epoch):
            self._perm.copy_(torch.randperm(R, device=self.device))
            if not self.use_graph:
                epoch()
                continue
            ptrs = tuple(t.data_ptr() for t in data)
            if self._epoch_graph is None or ptrs != self._epoch_ptrs:
                if self._warm < 1:                          # the very first epoch runs eagerly (lazy module loads)
                    self._warm += 1
                    epoch()
                    continue
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    epoch()
                self._epoch_graph, self._epoch_ptrs = g, ptrs
            self._epoch_graph.replay()
        return (self._tot / max(nsteps * self.ppo_epoch, 1)).tolist()

    def grad(self):
        """Gradient of the last minibatch step (slices reduced, before clipping) as a flat tensor copy (tests)."""
        if self.data_parallel:
            torch.cuda.synchronize(self.device)
            return self.gbuf.clone()                        # after ssl_apply: scaled by 1 / world
        ptr = self.lib.ssl_debug_grad(self.h)

        class _Dev:
            __cuda_array_interface__ = {"shape": (self.n_params,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}

        torch.cuda.synchronize(self.device)
        return torch.as_tensor(_Dev(), device=self.device).clone()
