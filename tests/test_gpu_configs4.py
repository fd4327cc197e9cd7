"""BASELINE.json configs[4] -- MikeStepperEnv-v0, 32768 envs over 8 ranks, the loop of playground/train.py:363-521 with the actor / critic on
PyTorch-ROCm -- EXECUTED at its stated shape on ONE MI355X: eight processes x 4096 Mike envs on cuda:0, gloo carrying
the collectives (RCCL needs one GPU per rank; this box has one).  `python -m steppingstone_amd.train`'s loop (ppo.train) runs 2 updates of
32 steps with the fixed-order curriculum on; asserted:
  * the eight replicas are bit-identical after the all-reduced updates, and every rank reports the same job-wide statistics (frames of
    the whole job, all-reduced episode mean, curriculum level) -- one decision for all ranks;
  * what each rank trained on IS the env: one 32768-env process replaying the eight ranks' recorded actions reproduces every rank's
    rollout storage of the first update (observations, rewards, masks; 32 steps x 8 ranks) bit for bit -- sharding by global env id;
  * `python bench.py --ppo --gpus 8` end to end as the driver would start it (self-launch, one JSON line from rank 0, marked
    `test_transport`).
It measures nothing (eight ranks share a GPU); it proves the 8-rank data-parallel training path at full size.  `pytest -m gpu`."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, N_LOCAL, T, UPDATES = 8, 4096, 32, 2
ENV_ID, SEED = "MikeStepperEnv-v0", 8


def _digest(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


FULL = dict(world=WORLD, n_local=N_LOCAL, T=T, device="cuda:0", mb=1024)          # configs[4]
SMALL = dict(world=2, n_local=24, T=8, device="cpu", mb=96)                        # the same code on the CPU stand-in (no GPU here)


def _make_env(cfg, n, offset):
    from steppingstone_amd.envs import SteppingStoneVecEnv
    if cfg["device"] == "cpu":
        from oracle_backend import OracleBackend
        return SteppingStoneVecEnv(ENV_ID, n, seed=SEED, return_numpy=False, env_id_offset=offset, backend=OracleBackend("mike", n, SEED, env_id_offset=offset))
    return SteppingStoneVecEnv(ENV_ID, n, seed=SEED, device=cfg["device"], return_numpy=False, env_id_offset=offset)


def _worker(rank, port, tmp, ret, cfg):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    WORLD, N_LOCAL, T = cfg["world"], cfg["n_local"], cfg["T"]
    if cfg["device"] != "cpu":
        torch.cuda.set_device(0)
    else:
        torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from steppingstone_amd import ppo
    envs = _make_env(cfg, N_LOCAL, rank * N_LOCAL)
    seen = {}

    def on_rollout(j, roll):
        if j == 0:          # what this rank trains its first update on
            np.save(os.path.join(tmp, "actions_rank%d.npy" % rank), roll.actions.cpu().numpy())
            seen["obs"] = [_digest(roll.obs[t]) for t in range(T + 1)]
            seen["rew_mask"] = [_digest(roll.rewards[t], roll.masks[t + 1], roll.bad_masks[t + 1]) for t in range(T)]

    ac, hist = ppo.train(envs, num_updates=UPDATES, num_steps=T, ppo_epoch=1, mini_batch_size=cfg["mb"], use_curriculum=True, log=None,
                         on_rollout=on_rollout)
    ret[rank] = dict(params=_digest(*[p for p in ac.parameters()]), finite=bool(all(torch.isfinite(p).all() for p in ac.parameters())),
                     stats=[(h["total_num_steps"], h["curriculum"], repr(h["mean_rew"])) for h in hist],
                     losses=[(h["value_loss"], h["action_loss"]) for h in hist], obs=seen["obs"], rew_mask=seen["rew_mask"])
    envs.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_eight_ranks_x_4096_mike_envs_train_as_one_job(tmp_path):
    _one_job(tmp_path, FULL)


def test_the_same_job_at_two_ranks_on_the_cpu_stand_in(tmp_path):
    """No GPU: 2 ranks x 24 envs over the oracle-backed stand-in env -- the test's own plumbing (recorded actions, per-rank digests, the
    single-process replay) exercised here."""
    _one_job(tmp_path, SMALL)


def _one_job(tmp_path, cfg):
    import torch.multiprocessing as mp
    WORLD, N_LOCAL, T = cfg["world"], cfg["n_local"], cfg["T"]
    dev = cfg["device"]
    port = 35500 + os.getpid() % 2000
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(port, str(tmp_path), ret, cfg), nprocs=WORLD, join=True)
        res = {k: dict(v) for k, v in ret.items()}
    assert sorted(res) == list(range(WORLD))
    # one job: identical replicas, identical job-wide statistics
    for r in range(1, WORLD):
        assert res[r]["params"] == res[0]["params"], "replica %d diverged from replica 0" % r
        assert res[r]["stats"] == res[0]["stats"], (r, res[r]["stats"], res[0]["stats"])
    assert all(res[r]["finite"] for r in range(WORLD))
    assert [s[0] for s in res[0]["stats"]] == [(j + 1) * T * N_LOCAL * WORLD for j in range(UPDATES)]      # frames of the WHOLE job: 1 048 576 per update
    assert all(np.isfinite(l).all() for l in res[0]["losses"])
    from steppingstone_amd import ppo
    torch.manual_seed(SEED)
    fresh = _digest(*[p for p in ppo.ActorCritic(num_ensembles=1).to(dev).parameters()])
    assert fresh != res[0]["params"]                                   # the weights moved
    # the env under the job: ONE 32768-env process, the ranks' actions replayed
    acts = torch.from_numpy(np.concatenate([np.load(os.path.join(str(tmp_path), "actions_rank%d.npy" % r)) for r in range(WORLD)], axis=1)).to(dev)
    assert acts.shape == (T, WORLD * N_LOCAL, 21)
    env = _make_env(cfg, WORLD * N_LOCAL, 0)
    env.update_curriculum(0)                                           # ppo.train(use_curriculum=True) starts at level 0
    obs = env.reset()
    sl = [slice(r * N_LOCAL, (r + 1) * N_LOCAL) for r in range(WORLD)]
    for r in range(WORLD):
        assert _digest(obs[sl[r]]) == res[r]["obs"][0], ("reset", r)
    for t in range(T):
        obs, rew, done, info = env.step(acts[t])
        mask = (1.0 - done.to(torch.float32)).unsqueeze(1)
        bad = (1.0 - info["bad_transition"].to(torch.float32)).unsqueeze(1)
        for r in range(WORLD):
            assert _digest(obs[sl[r]]) == res[r]["obs"][t + 1], ("obs", t, r)
            assert _digest(rew[sl[r]].unsqueeze(1), mask[sl[r]], bad[sl[r]]) == res[r]["rew_mask"][t], ("rew / masks", t, r)
    env.close()


@pytest.mark.gpu
def test_bench_ppo_gpus_8_runs_the_configs4_workload_end_to_end_on_one_gpu():
    # (10 PPO epochs of 128 minibatches with a gloo all-reduce each took 582 s for the 5 updates on the first run: the test transport runs
    # ONE epoch per update -- the line says so -- because this run proves the path, and measures nothing either way)
    env = dict(os.environ, SS_BENCH_TEST_TRANSPORT="gloo", SS_BENCH_TEST_PPO_EPOCHS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--ppo", "--gpus", "8", "--updates", "5", "--envs-per-gpu", "4096",
                          "--ppo-rows", "torch"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-3000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1
    d = rows[0]
    assert "error" not in d and d["learner_torch"].get("error") is None, d
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["unit"] == "frames/s" and d["value"] > 0
    assert d["config"]["envs_total"] == 32768 and d["config"]["workload"].startswith("MikeStepperEnv-v0") and "test_transport" in d["config"]
    assert d["config"]["parallelism"] == "data-parallel x8" and d["transport"] == "gloo"
    assert d["learner_torch"]["mini_batch_size"] == 1024 and d["learner_torch"]["updates"] == 5 and d["config"]["test_ppo_epochs"] == 1
    print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "transport")}))
