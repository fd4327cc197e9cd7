"""Host-side vec-env / single-env facades (steppingstone_amd.envs) driven by an oracle-backed test double: types,
shapes, hook plumbing and error behaviour of the reference protocol (common/envs_utils.py:542-606).  CPU only."""
import numpy as np
import pytest
import torch

import oracle_lib as ol
from oracle_backend import OracleBackend
from steppingstone_amd.envs import SteppingStoneEnv, SteppingStoneVecEnv, kind_of


def make(n, env_id="Walker3DStepperEnv-v0", seed=3, numpy_mode=True):
    return SteppingStoneVecEnv(env_id, n, seed=seed, return_numpy=numpy_mode, backend=OracleBackend(kind_of(env_id), n, seed))


def test_numpy_mode_types_and_infos():
    env = make(6)
    assert env.observation_space.shape == (60,) and env.action_space.shape[0] == 21
    obs = env.reset()
    assert obs.shape == (6, 60) and obs.dtype == np.float32
    got_episode = False
    for t in range(60):
        obs, rew, done, infos = env.step(env.random_actions(t).numpy())
        assert rew.shape == (6,) and rew.dtype == np.float64 and done.dtype == bool and len(infos) == 6
        for i, info in enumerate(infos):
            if done[i]:
                got_episode = True
                assert set(info["episode"]) == {"r", "l", "t"} and info["episode"]["l"] >= 1
                assert info["episode"]["r"] == round(info["episode"]["r"], 6)
            else:
                assert "episode" not in info.keys() and "bad_transition" not in info.keys()
    assert got_episode
    env.close()
    env.close()      # idempotent


def test_tensor_mode_and_hooks():
    env = make(5, "MikeStepperEnv-v0", numpy_mode=False)
    obs = env.reset()
    assert torch.is_tensor(obs) and obs.shape == (5, 60)
    obs, rew, done, info = env.step(torch.zeros(5, 21))
    assert done.dtype == torch.bool and set(info) == {"ep_ret", "ep_len", "bad_transition", "steps_reached", "update_terrain", "ep_ret_lo"}
    env.update_curriculum(3)
    env.update_specialist(2)
    env.update_sample_prob(np.full((5, 11, 11), 1 / 121.0))
    env.update_sample_prob(np.full((11, 11), 1 / 121.0))
    per_env = np.random.default_rng(0).random((5, 11, 11))
    env.update_sample_prob(per_env / per_env.sum(axis=(1, 2), keepdims=True))
    with pytest.raises(ValueError):
        env.update_sample_prob(np.zeros((4, 11, 11)))
    import warnings
    from steppingstone_amd.envs import SteppingStoneVecEnv
    SteppingStoneVecEnv._warned_set_mirror = False
    with pytest.warns(RuntimeWarning, match="phase mirroring"):       # use_phase_mirror is unsupported: said once, not silently accepted
        env.set_mirror(True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        env.set_mirror(True)                                           # ... once per process
        env.set_mirror(False)
    env.set_robot_params({"power": 0.8})
    env.set_env_params({"curriculum": 1})
    with pytest.raises(KeyError):
        env.set_robot_params({"mass": 2})
    assert env.create_temp_states().shape == (5, 121, 60)
    assert env.terrain_info.shape == (5, 20, 6) and env.next_step_index.shape == (5,)
    assert env.yaw_samples.shape == (11,) and np.isclose(env.pitch_samples[-1], np.deg2rad(30))
    with pytest.raises(NotImplementedError):
        env.get_images()
    with pytest.raises(AssertionError):
        env.step(torch.zeros(4, 21))


def test_state_roundtrip_matches_oracle_directly():
    env = make(4, seed=9)
    o = ol.OracleEnv("walker3d", 4, seed=9)
    assert np.array_equal(env.reset(), o.reset())
    for t in range(5):
        a = o.random_actions(t)
        eo, er, ed, _ = env.step(a)
        oo, orr, od, _ = o.step(a)
        assert np.array_equal(eo, oo) and np.array_equal(ed, od.astype(bool))
    st = env.get_state()
    env.set_state(st)
    assert np.array_equal(env.get_state().numpy(), st.numpy())


def test_single_env_facade():
    env = SteppingStoneEnv("mocca_envs:Walker3DStepperEnv-v0", seed=0, backend_factory=lambda k, n, s: OracleBackend(k, n, s))
    env.seed(1093)
    obs = env.reset()
    assert obs.shape == (60,) and env.spec.id.endswith("Walker3DStepperEnv-v0") and env._max_episode_steps == 1000
    total = 0.0
    for t in range(300):
        obs, r, d, info = env.step(np.zeros(21, np.float32))
        assert isinstance(r, float) and isinstance(d, bool)
        total += r
        if d:
            assert abs(info["episode"]["r"] - total) < 1e-3 * max(1, abs(total))
            assert abs(obs[0]) > 0.02              # terminal observation (no auto-reset for the plain env)
            break
    else:
        pytest.fail("episode did not end")
    assert env.robot.feet_contact.shape == (2,)
    assert env.create_temp_states().shape == (121, 60) and len(env.get_mirror_indices()) == 6
    obs2 = env.reset()
    assert np.all(obs2[:6] == 0)
    env.update_curriculum(2)
    env.update_sample_prob(np.full((11, 11), 1 / 121.0))
    env.close()


def test_monitor_csv_is_byte_identical_to_the_reference_writer(tmp_path):
    """steppingstone_amd.monitor_csv against the file the reference's own Monitor + ResultsWriter wrote for a scripted episode
    sequence (tests/golden/monitor_golden.json, tools/make_golden_monitor.py): same file name, header line, csv header, rows and
    line terminators; `r` = round(sum of the Python-float step rewards, 6) as Monitor.update computes it."""
    import json, os, re
    from steppingstone_amd.monitor_csv import MonitorFiles
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "monitor_golden.json")))
    text = G["text"]
    t_start = json.loads(text.split("\n")[0][2:])["t_start"]
    ts = [float(l.split(",")[2]) for l in text.split("\r\n")[1:] if l]          # the time column as the reference wrote it
    m = MonitorFiles(str(tmp_path), 8, G["env_id"], first_rank=0, max_open=2, t_start=t_start)
    assert sorted(os.listdir(str(tmp_path))) == sorted("%d.monitor.csv" % i for i in range(8))
    for ep, t in zip(G["episodes"], ts):
        m.write_row(5, {"r": round(sum(ep), 6), "l": len(ep), "t": t})
        m.write_row(1, {"r": 1.0, "l": 2, "t": t})                               # other envs interleaved, handles recycled (max_open 2)
        m.write_row(7, {"r": 2.0, "l": 3, "t": t})
    m.close()
    assert G["file_name"] == "5.monitor.csv"
    assert open(os.path.join(str(tmp_path), "5.monitor.csv"), newline="").read() == text
    assert re.fullmatch(r'# \{"t_start": [0-9.e+]+, "env_id": "Toy-v0"\} \nr,l,t\r\n', open(os.path.join(str(tmp_path), "0.monitor.csv"), newline="").read())


def test_make_vec_envs_log_dir_writes_monitor_files(tmp_path):
    """make_vec_envs(log_dir=...) (common/envs_utils.py:36-38,48-56): one <rank>.monitor.csv per env, a row per finished episode with
    the values of info["episode"]; numpy and tensor mode."""
    import csv, os
    for mode in (True, False):
        d = str(tmp_path / ("np" if mode else "tensor"))
        n = 6
        envs = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=4, return_numpy=mode, backend=OracleBackend(kind_of("Walker3DStepperEnv-v0"), n, 4),
                                   log_dir=d, env_id_offset=10)
        envs.reset()
        rows = {i: [] for i in range(n)}
        rng = np.random.default_rng(0)
        for t in range(80):
            a = rng.uniform(-1, 1, (n, 21)).astype(np.float32)
            obs, rew, done, infos = envs.step(a if mode else torch.from_numpy(a))
            if mode:
                for i, inf in enumerate(infos):
                    if "episode" in inf:
                        rows[i].append(inf["episode"])
            else:
                fl = envs._info.view(torch.float32)
                for i in torch.nonzero(done).flatten().tolist():
                    rows[i].append({"r": round(float(fl[i, 0].double() + fl[i, 5].double()), 6), "l": int(fl[i, 1])})
        envs.close()
        assert sum(len(v) for v in rows.values()) >= 6
        for i in range(n):
            f = open(os.path.join(d, "%d.monitor.csv" % (10 + i)), newline="")
            assert f.readline().startswith('# {"t_start": ')
            got = list(csv.DictReader(f))
            assert len(got) == len(rows[i])
            for g, e in zip(got, rows[i]):
                assert float(g["r"]) == e["r"] and int(g["l"]) == e["l"]
