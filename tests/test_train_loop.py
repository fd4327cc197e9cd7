"""Host side of the training driver on the oracle-backed env: progress.csv against the file the
reference's own ConsoleCSVLogger wrote (tests/golden/progress_golden.*, tools/make_golden_csv.py), the threshold /
adaptive sampler wired into the loop, checkpoints, the deterministic test loop, the episode ring.  CPU only."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

from oracle_backend import OracleBackend
from steppingstone_amd import ppo
from steppingstone_amd.csv_logger import ConsoleCSVLogger
from steppingstone_amd.envs import SteppingStoneVecEnv

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_progress_csv_is_byte_identical_to_the_reference_logger(tmp_path):
    g = json.load(open(os.path.join(GOLD, "progress_golden.json")))
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        lg = ConsoleCSVLogger(log_dir=str(tmp_path), console_log_interval=2)
        for r in g["records"]:
            lg.log_epoch(r)
        lg.close()
    assert open(os.path.join(str(tmp_path), "progress.csv")).read() == open(os.path.join(GOLD, "progress_golden.csv")).read()
    assert out.getvalue() == g["console"]


def test_episode_ring_is_the_reference_deque():
    from collections import deque
    rng = np.random.default_rng(0)
    n = 7
    ring, dq = ppo.EpisodeRing(n, torch.device("cpu")), deque(maxlen=n)
    for _ in range(40):
        done = rng.random(n) < 0.3
        ret = rng.normal(size=n).astype(np.float32)
        ring.push(torch.from_numpy(ret), torch.from_numpy(done))
        for i in range(n):                       # train.py:446-456: env order within a step
            if done[i]:
                dq.append(ret[i])
        s, c = ring.all_ranks_sum_count()
        assert c == len(dq) and abs(s - float(np.sum(dq))) < 1e-4
        assert sorted(ring.values().tolist()) == sorted(float(x) for x in dq)


def _envs(n, seed, offset=0, kind=1):
    return SteppingStoneVecEnv("MikeStepperEnv-v0" if kind else "Walker3DStepperEnv-v0", n, seed=seed, return_numpy=False,
                               env_id_offset=offset, backend=OracleBackend(kind, n, seed, env_id_offset=offset))


def test_training_loop_with_threshold_sampler_logs_and_checkpoints(tmp_path):
    n = 12
    envs, eval_envs, test_envs = _envs(n, 1), _envs(4, 1, 1000), _envs(2, 1, 2000)
    calls = []
    orig = envs.update_sample_prob
    envs.update_sample_prob = lambda p: (calls.append(np.asarray(p).copy()), orig(p))[1]
    # make the evaluation env advance quickly: start standing on the target
    be = eval_envs.backend
    orig_reset = eval_envs.reset

    def reset_on_target():
        orig_reset()
        st = be.o.get_state()
        st[:, 0] = st[:, 65 + 6]
        be.o.set_state(st)
        return eval_envs.get_obs()

    eval_envs.reset = reset_on_target
    test_envs._max_episode_steps = 40
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        logger = ConsoleCSVLogger(log_dir=str(tmp_path / "log"))
        ac, hist = ppo.train(envs, num_updates=3, num_steps=40, num_ensembles=2, ppo_epoch=1, mini_batch_size=120,
                             use_curriculum=False, sampling="threshold", eval_envs=eval_envs, test_envs=test_envs,
                             test_interval=1, logger=logger, save_dir=str(tmp_path / "ckpt"), env_name="MikeStepperEnv-v0",
                             log=None)
        logger.close()
    # threshold sampling: first update uniform (curriculum 5), then a critic-derived grid every update (train.py:229-272,460-469)
    assert [h["grid_updated"] for h in hist] == [False, True, True]
    assert len(calls) == 2 and all(c.shape == (11, 11) and abs(c.sum() - 1) < 1e-5 for c in calls)
    rows = open(str(tmp_path / "log" / "progress.csv")).read().strip().split("\n")
    assert rows[0] == open(os.path.join(GOLD, "progress_golden.csv")).readline().strip()      # the reference's columns
    assert len(rows) >= 2                                    # random-policy episodes finish within 40 steps
    files = sorted(os.listdir(str(tmp_path / "ckpt")))
    assert "MikeStepperEnv-v0_latest.pt" in files and "MikeStepperEnv-v0_best.pt" in files
    assert "MikeStepperEnv-v0_10000000.pt" in files          # the last update is always saved under the next checkpoint name
    ac2, ck = ppo.load_checkpoint(str(tmp_path / "ckpt" / "MikeStepperEnv-v0_10000000.pt"))
    for (k, a), (_, b) in zip(ac.state_dict().items(), ac2.state_dict().items()):
        assert torch.equal(a, b), k
    assert ck["update"] == 3 and ck["num_ensembles"] == 2
    # the policy convention (ABI version + per-joint signs) travels with the file; a file without it, or with the rounds-1..3
    # convention (same layout, left x / z joints and knees of the other sign), is refused instead of driving flipped joints
    assert ck["policy_convention"] == ppo.policy_convention() and ck["policy_convention"]["abi_version"] >= 4
    old = dict(ck)
    old.pop("policy_convention")
    torch.save(old, str(tmp_path / "ckpt" / "unstamped.pt"))
    with pytest.raises(ValueError, match="policy convention"):
        ppo.load_checkpoint(str(tmp_path / "ckpt" / "unstamped.pt"))
    old["policy_convention"] = {"abi_version": 3, "policy_sign": [1] * 21}
    torch.save(old, str(tmp_path / "ckpt" / "abi3.pt"))
    with pytest.raises(ValueError, match="policy convention"):
        ppo.load_checkpoint(str(tmp_path / "ckpt" / "abi3.pt"))
    ppo.load_checkpoint(str(tmp_path / "ckpt" / "abi3.pt"), allow_convention_mismatch=True)
    # round 6: the env's numbers travel too (warning-level): a file from an env with other robot tables loads, loudly
    assert ck["env_fingerprint"] == ppo.env_fingerprint()
    other = dict(ck, env_fingerprint="0" * 64)
    torch.save(other, str(tmp_path / "ckpt" / "other_robot.pt"))
    with pytest.warns(RuntimeWarning, match="other robot / terrain numbers"):
        ppo.load_checkpoint(str(tmp_path / "ckpt" / "other_robot.pt"))


def test_adaptive_sampler_and_specialist_switches():
    n = 8
    envs, eval_envs = _envs(n, 2, kind=0), _envs(4, 2, 500, kind=0)
    be = eval_envs.backend
    orig_reset = eval_envs.reset

    def reset_on_target():
        orig_reset()
        st = be.o.get_state()
        st[:, 0] = st[:, 65 + 6]
        be.o.set_state(st)
        return eval_envs.get_obs()

    eval_envs.reset = reset_on_target
    ac, hist = ppo.train(envs, num_updates=2, num_steps=8, ppo_epoch=1, mini_batch_size=64, use_curriculum=False,
                         sampling="adaptive", eval_envs=eval_envs, log=None)
    assert [h["grid_updated"] for h in hist] == [True, True]       # adaptive: every update (train.py:320-361)
    envs2 = _envs(n, 3, kind=0)
    levels = []
    envs2.update_specialist = lambda s: levels.append(s)
    ppo.train(envs2, num_updates=1, num_steps=4, ppo_epoch=1, mini_batch_size=32, use_curriculum=False, use_specialist=True, log=None)
    assert levels == [0]
