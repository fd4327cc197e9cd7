"""BASELINE.json configs[3] -- Walker3DStepperEnv-v0, 32768 envs sharded over 8 ranks, all-gather of the packed [N/G,62] blocks --
EXECUTED with 8 ranks at full size on ONE MI355X (all ranks on cuda:0, gloo carries the collectives: RCCL needs one GPU per rank and
this box has one).  What has never run with more than 2-3 ranks on hardware before round 5: the 8-entry peer tables and gather
slots, the `--gpus 8` self-launch of bench.py, its watchdog and gather_verified self-check, ShardedVecEnv with world_size 8 on the
real HIP kernels.  It measures nothing (8 processes share one GPU); it proves the 8-rank code path end to end.  `pytest -m gpu`."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, N_LOCAL, STEPS = 8, 4096, 5
ENV_ID = "Walker3DStepperEnv-v0"


def _digest(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def _worker(rank, port, ret):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv
    local = SteppingStoneVecEnv(ENV_ID, N_LOCAL, seed=0, device="cuda:0", return_numpy=False, env_id_offset=rank * N_LOCAL)
    env = ShardedVecEnv(local)
    assert env.num_envs == WORLD * N_LOCAL
    out = [_digest(env.reset())]
    gen = torch.Generator().manual_seed(1)
    for t in range(STEPS):                       # policy-in-the-loop shape: global actions in, gathered obs / rew / done out
        acts = (torch.rand((env.num_envs, 21), generator=gen) * 2 - 1).to("cuda:0")
        obs, rew, done, infos = env.step(acts)
        out.append(_digest(obs, rew, done.to(torch.float32), infos["ep_ret"], infos["steps_reached"].to(torch.float32)))
    out.append(_digest(*env.rollout_random(6, t0=100)))                      # per-step all-gather under on-device actions
    out.append(_digest(*env.rollout_random_chunked(40, t0=200, chunk=32)))   # K-step launches, one gather per 32-step chunk (+ a ragged one)
    env.check_exchange()
    ret[rank] = out
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_x_4096_envs_equal_one_32768_env_process_bit_for_bit():
    import torch.multiprocessing as mp
    from steppingstone_amd.envs import SteppingStoneVecEnv
    port = 33500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(port, ret), nprocs=WORLD, join=True)
        res = {k: list(v) for k, v in ret.items()}
    env = SteppingStoneVecEnv(ENV_ID, N_LOCAL * WORLD, seed=0, device="cuda:0", return_numpy=False)
    ref = [_digest(env.reset())]
    gen = torch.Generator().manual_seed(1)
    for t in range(STEPS):
        acts = (torch.rand((N_LOCAL * WORLD, 21), generator=gen) * 2 - 1).to("cuda:0")
        obs, rew, done, info = env.step(acts)
        ref.append(_digest(obs, rew, done.to(torch.float32), info["ep_ret"], info["steps_reached"].to(torch.float32)))
    ref.append(_digest(*env.rollout_random(6, t0=100)))
    ref.append(_digest(*env.rollout_random(40, t0=200)))
    env.close()
    assert sorted(res) == list(range(WORLD))
    for rank in range(WORLD):
        assert res[rank] == ref, (rank, [i for i, (a, b) in enumerate(zip(res[rank], ref)) if a != b])


def test_bench_gpus_8_runs_the_configs3_workload_end_to_end_on_one_gpu():
    """`python bench.py --gpus 8 --envs-per-gpu 4096` exactly as the driver starts it for SCALE (self-launch of 8 ranks, chunked
    exchange, max over ranks, one JSON line from rank 0), gloo as the transport and all ranks on cuda:0.  The line is marked
    `test_transport`; its numbers mean nothing (8 ranks share one GPU), its shape and its self-checks do."""
    env = dict(os.environ, SS_BENCH_TEST_TRANSPORT="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "64", "--warmup", "32",
                          "--envs-per-gpu", "4096", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1                                             # rank 0 only
    d = rows[0]
    assert "error" not in d, d
    assert d["n_gpus"] == 8 and d["steps"] == 64 and d["warmup"] == 32 and d["scaling"] == "weak"
    assert d["config"]["envs_total"] == 32768 and d["config"]["ranks"] == 8 and d["config"]["parallelism"] == "env-shard x8+allgather"
    assert d["config"]["workload"].startswith("Walker3DStepperEnv-v0") and "test_transport" in d["config"]
    assert d["gather_verified"] is True and d["rccl_ranks"] == 8 and d["transport"] == "gloo"
    assert d["value"] > 0 and d["no_gather"]["value"] > 0 and d["per_step_gather"]["value"] > 0
    assert d["value_per_step_exchange"] == d["per_step_gather"]["value"] and d["rccl"]["ranks"] == 8
    assert len(d.get("pci_bus_ids", d["config"].get("pci_bus_ids", [None] * 8))) == 8
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "gather_verified", "transport")}))
