"""N>1 path on CPU: gloo at world sizes 2 AND 8 (BASELINE configs[3] / [4] run 8 ranks), one process per 'GPU',
each rank owning a contiguous block of envs (oracle-backed test double), per-step all-gather of the packed [obs|rew|done] block.
The gathered result must equal a single-process run over all envs (global env ids key the RNG, so sharding is invisible: the
partition arithmetic and the RNG invariance at G = 8, not only G = 2).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_LOCAL, STEPS = 6, 12


def _worker(rank, port, ret, WORLD):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from oracle_backend import OracleBackend
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv
    local = SteppingStoneVecEnv("Walker3DStepperEnv-v0", N_LOCAL, seed=5, return_numpy=False, env_id_offset=rank * N_LOCAL,
                                backend=OracleBackend(0, N_LOCAL, 5, env_id_offset=rank * N_LOCAL))
    env = ShardedVecEnv(local)
    assert env.num_envs == N_LOCAL * WORLD
    out = [env.reset().clone().numpy()]
    gen = torch.Generator().manual_seed(0)
    for t in range(STEPS):
        acts = torch.rand((env.num_envs, 21), generator=gen) * 2 - 1       # same global actions on every rank
        obs, rew, done, _ = env.step(acts)
        out.append(np.concatenate([obs.numpy(), rew.numpy()[:, None], done.numpy()[:, None].astype(np.float32)], 1))
    # benchmark path too: more steps than the ring of gather buffers is deep (asynchronous collectives, batched waits)
    o2, r2, d2 = env.rollout_random(11, t0=100)
    out.append(np.concatenate([o2.numpy(), r2.numpy()[:, None], d2.numpy()[:, None].astype(np.float32)], 1))
    proof = [env.verify_last_exchange()]                               # self-proof of the per-step exchange (bench.py)
    # chunked exchange (K steps per launch, one collective per chunk): 2 full chunks + a ragged one
    o3, r3, d3 = env.rollout_random_chunked(11, t0=200, chunk=4)
    out.append(np.concatenate([o3.numpy(), r3.numpy()[:, None], d3.numpy()[:, None].astype(np.float32)], 1))
    proof.append(env.verify_last_exchange())                           # ... and of the chunked one
    # a block that did not arrive as sent must be noticed by EVERY rank: rank 1 damages the copy it received from rank 0
    if rank == 1:
        env._chunk_all[env._last[1]].view(WORLD, -1, N_LOCAL, 62)[0, 0, 0, 3] += 1.0
    proof.append(env.verify_last_exchange())
    ret[rank] = out
    ret[100 + rank] = proof
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("WORLD", [2, 8])
def test_sharding_equals_single_process(WORLD):
    import oracle_lib as ol
    port = 29500 + (os.getpid() + 17 * WORLD) % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(port, ret, WORLD), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    o = ol.OracleEnv("walker3d", N_LOCAL * WORLD, seed=5)
    ref = [o.reset()]
    gen = torch.Generator().manual_seed(0)
    for t in range(STEPS):
        acts = (torch.rand((N_LOCAL * WORLD, 21), generator=gen) * 2 - 1).numpy()
        ob, r, d, _ = o.step(acts)
        ref.append(np.concatenate([ob, r[:, None], d[:, None].astype(np.float32)], 1))
    for k in range(11):
        ob, r, d, _ = o.step(o.random_actions(100 + k))
    ref.append(np.concatenate([ob, r[:, None], d[:, None].astype(np.float32)], 1))
    for k in range(11):
        ob, r, d, _ = o.step(o.random_actions(200 + k))
    ref.append(np.concatenate([ob, r[:, None], d[:, None].astype(np.float32)], 1))
    for rank in range(WORLD):
        assert len(res[rank]) == len(ref)
        for a, b in zip(res[rank], ref):
            assert np.array_equal(a, b)
        assert res[100 + rank] == [(True, WORLD), (True, WORLD), (False, WORLD)]
