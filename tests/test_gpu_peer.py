"""Peer-store all-gather (steppingstone_amd/peer.py, ss_step_packed_peers): the per-step exchange written by the step
kernel itself into every rank's gather buffer, completion through flag words.  A 1-GPU box can run the whole protocol:
  * self-peering: several env handles of one process act as the ranks;
  * two PROCESSES on the same GPU, buffers shared through HIP IPC handles (the real cross-process path; gloo is only the
    control plane that ships the 64-byte handles).
Both must reproduce exactly what the plain packed step writes.  `pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_peering_three_ranks_on_one_gpu():
    from steppingstone_amd.envs import SteppingStoneVecEnv
    from steppingstone_amd.peer import PACK, PeerGather
    n, world, steps = 200, 3, 12                    # ragged last wavefront on purpose
    envs = [SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=4, device="cuda:0", env_id_offset=r * n, return_numpy=False)
            for r in range(world)]
    ref = [SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=4, device="cuda:0", env_id_offset=r * n, return_numpy=False)
           for r in range(world)]
    for e in envs + ref:
        e.update_curriculum(5)
        e.reset()
    peers = PeerGather.connect_in_process(envs)
    packed = torch.zeros((world * n, PACK), device="cuda:0")
    for t in range(steps):
        slots = [p.step(actions=None, t=t) for p in peers]
        assert len(set(slots)) == 1
        for r, e in enumerate(ref):
            e.step_packed(packed[r * n:(r + 1) * n], actions=None, t=t)
        for p in peers:
            got = p.wait(slots[0])
            assert torch.equal(got, packed), (t, p.rank)
    # explicit actions path
    act = torch.rand((world * n, 21), device="cuda:0") * 2 - 1
    slots = [p.step(actions=act[p.rank * n:(p.rank + 1) * n], t=0) for p in peers]
    for r, e in enumerate(ref):
        e.step_packed(packed[r * n:(r + 1) * n], actions=act[r * n:(r + 1) * n])
    for p in peers:
        assert torch.equal(p.wait(slots[0]), packed)
        assert p.error() == 0
    for p in peers:
        p.close()
    for e in envs + ref:
        e.close()


def _proc(rank, world, port, n, steps, ret):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from steppingstone_amd.envs import SteppingStoneVecEnv
    from steppingstone_amd.peer import PeerGather
    torch.cuda.set_device(0)
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=9, device="cuda:0", env_id_offset=rank * n, return_numpy=False)
    env.reset()
    pg = PeerGather.connect_processes(env)
    outs = []
    for t in range(steps):
        slot = pg.step(actions=None, t=t)
        g = pg.wait(slot)
        outs.append(g.clone().cpu().numpy())
        torch.cuda.synchronize()
        dist.barrier()                               # nobody overwrites a slot a slower rank still reads (ring of 2)
    ret[rank] = (outs, pg.error())
    pg.close()
    env.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_on_one_gpu_through_ipc_handles():
    import torch.multiprocessing as mp
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n, world, steps = 256, 2, 6
    port = 37500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_proc, args=(r, world, port, n, steps, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=240)
        assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
        res = {k: v for k, v in ret.items()}
    # reference: the same global envs stepped in this process
    ref = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n * world, seed=9, device="cuda:0", return_numpy=False)
    ref.reset()
    packed = torch.zeros((n * world, 62), device="cuda:0")
    for t in range(steps):
        ref.step_packed(packed, actions=None, t=t)
        exp = packed.cpu().numpy()
        for r in range(world):
            assert np.array_equal(res[r][0][t], exp), (t, r)
    assert res[0][1] == 0 and res[1][1] == 0
    ref.close()
