"""Data-parallel learner and sharded collection on CPU: world_size-2 gloo.
  * PPO.update with each rank holding half of the transitions == one process holding all of them (gradient all-reduce +
    global advantage statistics; algorithms/ppo.py:40-108 is the single-process definition);
  * ppo.collect on a ShardedVecEnv: obs / rew / done AND the info words arrive with global shapes;
  * ppo.train under torch.distributed: identical weights on every rank after updates, one curriculum decision.
CPU only."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2
T, N_LOCAL = 6, 5


def _paths():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _batch():
    g = torch.Generator().manual_seed(11)
    n = N_LOCAL * WORLD
    r = lambda *s: torch.randn(*s, generator=g)     # noqa: E731
    return {"obs": r(T + 1, n, 60), "act": r(T, n, 21) * 0.5, "logp": -20 + r(T, n, 1), "vpred": r(T + 1, n, 1), "ret": r(T + 1, n, 1)}


def _fill(roll, b, sl):
    roll.obs.copy_(b["obs"][:, sl]); roll.actions.copy_(b["act"][:, sl]); roll.logp.copy_(b["logp"][:, sl])
    roll.value_preds.copy_(b["vpred"][:, sl]); roll.returns.copy_(b["ret"][:, sl])


def _dp_worker(rank, port, ret):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from steppingstone_amd import ppo
    torch.manual_seed(3)
    ac = ppo.ActorCritic(num_ensembles=2)
    agent = ppo.PPO(ac, ppo_epoch=2, mini_batch_size=T * N_LOCAL, lr=3e-4)
    roll = ppo.Rollouts(T, N_LOCAL, torch.device("cpu"))
    _fill(roll, _batch(), slice(rank * N_LOCAL, (rank + 1) * N_LOCAL))
    losses = agent.update(roll)
    ret[rank] = (torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy(), losses)
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_update_equals_single_process_on_the_concatenated_batch():
    _paths()
    from steppingstone_amd import ppo
    port = 31500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_dp_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    torch.manual_seed(3)
    ac = ppo.ActorCritic(num_ensembles=2)
    w0 = torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy().copy()
    agent = ppo.PPO(ac, ppo_epoch=2, mini_batch_size=T * N_LOCAL * WORLD, lr=3e-4)
    roll = ppo.Rollouts(T, N_LOCAL * WORLD, torch.device("cpu"))
    _fill(roll, _batch(), slice(None))
    losses = agent.update(roll)
    w = torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy()
    assert np.array_equal(res[0][0], res[1][0])                       # replicas stay identical, bit for bit
    moved = np.abs(w - w0).max()
    assert moved > 1e-4
    assert np.abs(res[0][0] - w).max() < 2e-3 * moved + 1e-7, (np.abs(res[0][0] - w).max(), moved)
    # every rank reports its own half's losses; their mean is the full batch's
    for k in range(2):
        assert abs(0.5 * (res[0][1][k] + res[1][1][k]) - losses[k]) < 1e-4 * (1 + abs(losses[k]))


def _collect_worker(rank, port, ret):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from oracle_backend import OracleBackend
    from steppingstone_amd import ppo
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 6
    local = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=5, return_numpy=False, env_id_offset=rank * n,
                                backend=OracleBackend(0, n, 5, env_id_offset=rank * n))
    env = ShardedVecEnv(local)
    assert env.device == torch.device("cpu") and len(env.get_mirror_indices()) == 6
    torch.manual_seed(0)                               # same policy and the same exploration noise on both ranks
    ac = ppo.ActorCritic()
    roll = ppo.Rollouts(30, env.num_envs, env.device)
    roll.obs[0].copy_(env.reset())
    ring = ppo.EpisodeRing(env.num_envs, env.device)
    st = torch.zeros(2)
    ppo.collect(env, ac, roll, 30, ep_stats=st, ring=ring)
    ret[rank] = (roll.obs.numpy().copy(), roll.rewards.numpy().copy(), roll.masks.numpy().copy(), roll.bad_masks.numpy().copy(),
                 st.numpy().copy(), ring.values().numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_collect_on_a_sharded_env_sees_global_info():
    _paths()
    port = 33500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_collect_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)                     # every rank holds the same global rollout
    obs, rew, masks, bad, st, ring = res[0]
    assert obs.shape == (31, 12, 60) and masks.shape == (31, 12, 1)
    n_done = int((masks[1:] == 0).sum())
    assert n_done > 0 and st[1] == n_done and ring.size == min(n_done, 12)
    # single-process reference over all 12 envs with the same policy / noise
    from oracle_backend import OracleBackend
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 12, seed=5, return_numpy=False, backend=OracleBackend(0, 12, 5))
    torch.manual_seed(0)
    ac = ppo.ActorCritic()
    roll = ppo.Rollouts(30, 12, torch.device("cpu"))
    roll.obs[0].copy_(env.reset())
    st1 = torch.zeros(2)
    ppo.collect(env, ac, roll, 30, ep_stats=st1)
    assert np.array_equal(roll.obs.numpy(), obs) and np.array_equal(roll.rewards.numpy(), rew)
    assert np.array_equal(roll.masks.numpy(), masks) and np.allclose(st1.numpy(), st)


def _train_worker(rank, port, ret):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from oracle_backend import OracleBackend
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 6
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=8, return_numpy=False, env_id_offset=rank * n,
                               backend=OracleBackend(1, n, 8, env_id_offset=rank * n))
    # only rank 0 holds a logger (as in steppingstone_amd/train.py): the CSV row's episode statistics are gathered from every
    # rank, so the gather must be entered by the rank WITHOUT a logger too (it deadlocked when gated on `logger is not None`)
    import tempfile
    from steppingstone_amd.csv_logger import ConsoleCSVLogger
    logdir = tempfile.mkdtemp() if rank == 0 else None
    logger = ConsoleCSVLogger(log_dir=logdir, console_log_interval=1000) if rank == 0 else None
    ac, hist = ppo.train(envs, num_updates=3, num_steps=24, ppo_epoch=2, mini_batch_size=24, log=None, logger=logger)
    rows = len(open(os.path.join(logdir, "progress.csv")).read().strip().splitlines()) - 1 if rank == 0 else -1
    ret[rank] = (torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).numpy(), [h["curriculum"] for h in hist],
                 hist[-1]["total_num_steps"], [h["mean_rew"] for h in hist], rows)
    dist.barrier()
    dist.destroy_process_group()


def test_train_under_two_ranks_keeps_replicas_identical():
    port = 35500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_train_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1] == res[1][1] and res[0][2] == 3 * 24 * 6 * WORLD
    assert res[0][4] >= 1                                             # rank 0 wrote CSV rows; nobody hung
    assert np.allclose(res[0][3], res[1][3], equal_nan=True)          # the gate statistic is all-reduced: same on every rank
