"""Hunt for the non-reproducible FIRST step (run on the GPU box).  Round 3 saw, twice, the first step of a new
environment come out different from identical repetitions after it (numpy drop-in mode both times).  Two in-process reproductions
of "first":

    python tools/first_step_hunt.py cold   REPS [numpy]   the same injected state and actions stepped REPS times per robot and env
                                                          count, the instruction cache of every CU evicted by a 160 KiB dummy kernel
                                                          (tools/probes/icache_evict.hip -> var/libicache_evict.so) before every
                                                          step; every result compared bit for bit with the first
    python tools/first_step_hunt.py newenv ITERS [numpy]  ITERS times: a NEW environment (fresh ss_create: hipMalloc + memsets, fresh
                                                          I/O and staging buffers), its first step from an injected state, then the
                                                          same step three more times; all four results must be bit-equal

On a mismatch the differing output indices, both bit patterns and the env they belong to are printed.  Output layout per env (columns
of the compared row): 0..59 obs | 60 rew | 61 done | 62..247 state (ss_get_state) | 248..253 info words."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle_lib as ol  # noqa: E402
from steppingstone_amd.envs import SteppingStoneVecEnv  # noqa: E402

MODE = sys.argv[1]
COUNT = int(sys.argv[2])
NUMPY = len(sys.argv) > 3 and sys.argv[3] == "numpy"
KINDS = (("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike"))


def inputs(kind, n, seed=2):
    """A contact-rich state set from the CPU oracle (12 curriculum-5 random-action steps) and one action array."""
    o = ol.OracleEnv(kind, n, seed=seed)
    o.set_curriculum(5)
    o.reset()
    for t in range(12):
        o.step(o.random_actions(t))
    return o.get_state().astype(np.float32), o.random_actions(50)


def one_step(g, st, act, before_step=None):
    """set_state + step; returns the [n, 254] row block described in the header (a device tensor)."""
    g.set_state(st)
    if before_step is not None:
        before_step()
    ob, rw, dn, _ = g.step(act)
    if NUMPY:
        ob = torch.as_tensor(np.array(ob)).cuda()
        rw = torch.as_tensor(np.array(rw, np.float32)).cuda()
        dn = torch.as_tensor(np.array(dn)).cuda()
    return torch.cat([ob.reshape(g.num_envs, -1).clone(), rw.reshape(-1, 1).float(), dn.reshape(-1, 1).float(), g.get_state(),
                      g._info.view(torch.float32).clone()], dim=1)


def report(tag, out, ref):
    ne = (out.view(torch.int32) != ref.view(torch.int32))
    idx = torch.nonzero(ne)[:12].tolist()
    print("DIFF", tag, "count", int(ne.sum()), "first (env, column, got, expected):",
          [(e, c, hex(out.view(torch.int32)[e, c].item() & 0xffffffff), hex(ref.view(torch.int32)[e, c].item() & 0xffffffff)) for e, c in idx], flush=True)


def cold():
    lib = C.CDLL(os.path.join(ROOT, "var", "libicache_evict.so"))
    lib.icache_evict.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    sink = torch.zeros(4, dtype=torch.int32, device="cuda:0")
    bad = 0
    total = 0
    for env_id, kind in KINDS:
        for n in (1, 33, 700, 4096):
            st, act = inputs(kind, n)
            if not NUMPY:
                st, act = torch.as_tensor(st).cuda(), torch.as_tensor(act).cuda()
            g = SteppingStoneVecEnv(env_id, n, seed=2, device="cuda:0", return_numpy=NUMPY)
            g.update_curriculum(5)
            g.reset()
            ref = None
            nbad = torch.zeros((), dtype=torch.int64, device="cuda:0")
            keep = None
            for r in range(COUNT):
                ob = one_step(g, st, act, lambda: lib.icache_evict(C.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                                   C.c_void_p(sink.data_ptr()), 1024))
                if ref is None:
                    ref = ob.clone()
                    keep = ob.clone()
                else:
                    d = (ob.view(torch.int32) != ref.view(torch.int32)).any()
                    keep = torch.where(d & (nbad == 0), ob, keep)
                    nbad += d
                if (r + 1) % 2000 == 0 or r + 1 == COUNT:
                    k = int(nbad.item())
                    if k:
                        report("%s n=%d rep<=%d (%d differing results so far)" % (env_id, n, r, k), keep, ref)
                        bad += k
                        nbad.zero_()
            total += COUNT * n
            g.close()
            print(env_id, n, "done", flush=True)
    print("cold-instruction-cache loop%s: %d repetitions per configuration, %d env-steps, differing results: %d" % (
        " (numpy mode)" if NUMPY else "", COUNT, total, bad))


def newenv():
    bad = 0
    total = 0
    rng = np.random.default_rng(0)
    cache = {}
    for it in range(COUNT):
        env_id, kind = KINDS[it % 2]
        n = int(rng.choice([1, 3, 33, 64, 700, 2048]))
        if (kind, n) not in cache:
            cache[(kind, n)] = inputs(kind, n, seed=2)
        st, act = cache[(kind, n)]
        if not NUMPY:
            st, act = torch.as_tensor(st).cuda(), torch.as_tensor(act).cuda()
        g = SteppingStoneVecEnv(env_id, n, seed=2, device="cuda:0", return_numpy=NUMPY)
        g.update_curriculum(5)
        g.reset()
        outs = [one_step(g, st, act) for _ in range(4)]
        for k in range(1, 4):
            if not torch.equal(outs[k].view(torch.int32), outs[0].view(torch.int32)):
                bad += 1
                report("%s n=%d iteration %d: step %d vs the first step of the new env" % (env_id, n, it, k), outs[0], outs[k])
        total += 4 * n
        g.close()
        del g
        if (it + 1) % 500 == 0:
            print("iteration", it + 1, "differing:", bad, flush=True)
    print("new-env loop%s: %d new environments, %d env-steps, first steps that differ from their repetitions: %d" % (
        " (numpy mode)" if NUMPY else "", COUNT, total, bad))


if __name__ == "__main__":
    {"cold": cold, "newenv": newenv}[MODE]()
