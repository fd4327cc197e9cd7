"""Two ranks on ONE MI355X (both on cuda:0, gloo carries the collectives: the driver's GPU box has a single GPU, RCCL needs one
GPU per rank).  What a multi-GPU run does, with the real HIP kernels under it:
  * env sharding: each rank steps its block of envs (global env ids key every random stream), the packed blocks are all-gathered
    per step and per chunk of K steps -- every rank ends up with exactly what ONE process stepping all envs produces, bit for bit;
  * data-parallel PPO (the default torch learner: gradients all-reduced between backward and Adam) -- replicas stay identical.
`pytest -m gpu`."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, N_LOCAL, STEPS = 2, 384, 10


def _setup(rank, port):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    return dist


def _shard_worker(rank, port, ret):
    dist = _setup(rank, port)
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv
    local = SteppingStoneVecEnv("MikeStepperEnv-v0", N_LOCAL, seed=5, device="cuda:0", return_numpy=False, env_id_offset=rank * N_LOCAL)
    env = ShardedVecEnv(local)
    env.update_curriculum(5)
    out = [env.reset().cpu().numpy()]
    gen = torch.Generator().manual_seed(0)
    for t in range(STEPS):
        acts = (torch.rand((env.num_envs, 21), generator=gen) * 2 - 1).to("cuda:0")     # the same global actions on every rank
        obs, rew, done, infos = env.step(acts)
        out.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], 1))
    o2, r2, d2 = env.rollout_random(9, t0=100)                                          # per-step all-gather
    out.append(np.concatenate([o2.cpu().numpy(), r2.cpu().numpy()[:, None], d2.cpu().numpy()[:, None].astype(np.float32)], 1))
    o3, r3, d3 = env.rollout_random_chunked(11, t0=200, chunk=4)                        # K-step launches, one gather per chunk
    out.append(np.concatenate([o3.cpu().numpy(), r3.cpu().numpy()[:, None], d3.cpu().numpy()[:, None].astype(np.float32)], 1))
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process_bit_for_bit():
    import torch.multiprocessing as mp
    from steppingstone_amd.envs import SteppingStoneVecEnv
    port = 35500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_shard_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    env = SteppingStoneVecEnv("MikeStepperEnv-v0", N_LOCAL * WORLD, seed=5, device="cuda:0", return_numpy=False)
    env.update_curriculum(5)
    ref = [env.reset().cpu().numpy()]
    gen = torch.Generator().manual_seed(0)
    for t in range(STEPS):
        acts = (torch.rand((N_LOCAL * WORLD, 21), generator=gen) * 2 - 1).to("cuda:0")
        obs, rew, done, _ = env.step(acts)
        ref.append(np.concatenate([obs.cpu().numpy(), rew.cpu().numpy()[:, None], done.cpu().numpy()[:, None].astype(np.float32)], 1))
    for t0, k in ((100, 9), (200, 11)):
        o, r, d = env.rollout_random(k, t0=t0)
        ref.append(np.concatenate([o.cpu().numpy(), r.cpu().numpy()[:, None], d.cpu().numpy()[:, None].astype(np.float32)], 1))
    for rank in range(WORLD):
        assert len(res[rank]) == len(ref)
        for i, (a, b) in enumerate(zip(res[rank], ref)):
            assert np.array_equal(a, b), (rank, i)


def _train_worker(rank, port, ret):
    dist = _setup(rank, port)
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    envs = SteppingStoneVecEnv("Walker3DStepperEnv-v0", 256, seed=8, device="cuda:0", return_numpy=False, env_id_offset=rank * 256)
    ac, hist = ppo.train(envs, num_updates=3, num_steps=8, ppo_epoch=2, mini_batch_size=512, log=None)
    ret[rank] = (torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).cpu().numpy(),
                 [(h["value_loss"], h["action_loss"]) for h in hist], hist[-1]["total_num_steps"])
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_training_keeps_the_replicas_identical():
    import torch.multiprocessing as mp
    port = 37500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_train_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: v for k, v in ret.items()}
    w0, w1 = res[0][0], res[1][0]
    assert np.isfinite(w0).all() and np.array_equal(w0, w1)          # same all-reduced gradient, same deterministic step
    assert res[0][2] == res[1][2] == 3 * 8 * 256 * WORLD             # frames of the whole job
    assert all(np.isfinite(l).all() for l in res[0][1])
    # the weights moved away from the initialisation (ppo.train seeds with 8)
    from steppingstone_amd import ppo
    torch.manual_seed(8)
    fresh = torch.cat([p.detach().reshape(-1) for p in ppo.ActorCritic(num_ensembles=1).parameters()]).numpy()
    assert fresh.shape == w0.shape and np.abs(w0 - fresh).max() > 1e-5


def test_bench_two_rank_path_runs_end_to_end_on_one_gpu():
    """`python bench.py --gpus 2` exactly as the driver starts it (self-launch, chunked exchange, max over ranks, one JSON line
    from rank 0) -- with gloo as the transport and both ranks on cuda:0, because this box has one GPU.  Checks the line's
    shape, not its numbers."""
    import json
    import subprocess
    env = dict(os.environ, SS_BENCH_TEST_TRANSPORT="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "32",
                          "--envs-per-gpu", "1024"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1                                             # rank 0 only
    d = rows[0]
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["warmup"] == 32 and d["scaling"] == "weak"
    assert d["config"]["envs_total"] == 2048 and d["config"]["ranks"] == 2 and d["config"]["parallelism"] == "env-shard x2+allgather"
    assert "test_transport" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    assert "no_gather" in d and "per_step_gather" in d and d["roofline"]["frac"] > 0
    # the self-proof of the exchange and the policy-in-the-loop row (round 3)
    assert d["gather_verified"] is True and d["rccl_ranks"] == 2 and "rccl_version" in d and d["transport"] == "gloo"
    assert d["per_step_gather"]["value"] > 0 and d["policy_in_the_loop"]["ms_per_step"] == d["per_step_gather"]["ms_per_step"]
    assert d["repeats"] >= 9 and d["timed_region_s"] >= 0.2 and d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    # round 6: the per-step exchange (north_star's / SURVEY 8d-4's shape) is a first-class value beside the chunked one, and the line carries
    # the process group's own view
    assert d["value_per_step_exchange"] == d["per_step_gather"]["value"] and d["ms_per_step_per_step_exchange"] == d["per_step_gather"]["ms_per_step"]
    assert d["rccl"]["ranks"] == 2 and d["rccl"]["backend"] == "gloo" and len(d["rccl"]["devices_per_rank"]) == 2
    assert "32-step" in d["config"]["value_is"] and "per 32-step chunk" in d["config"]["collective"]
