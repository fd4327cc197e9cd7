"""GPU parity tests: the HIP path (through the C ABI, via steppingstone_amd.envs) against the CPU oracle on the
same seeded inputs.  Run with `pytest -m gpu` on an MI355X.

Tolerances (fp32, stated here as required by the task brief; the rule itself is tests/parity_rule.py -- FROZEN, see its header and
tests/parity_rule.lock -- and the thresholds every caller asserts are tests/parity_assert.py):
  * integer / index / RNG-driven quantities (next_step_index, counters, rng counter, sampled grid cell, contact flags,
    done, bad_transition, update_terrain): bit-exact;
  * one control step from an identical injected state (4 substeps, different operation order, own sincos / reciprocal):
    |obs| error <= max(1e-4, 8 s), |rew| error <= max(1e-4, 8 s_rew), post-step pose <= max(1e-4, 8 s_pose), post-step rates <=
    max(1e-3, 8 s_vel), where 1e-4 is the north-star's per-step bound and s is the measured first-order response of the fp64 oracle
    to an 8-ulp error in each of that step's 55 dynamic state inputs (summed).  An env-step whose oracle evaluation has a discrete
    decision within 1e-5 of its threshold must match the oracle re-evaluated on one of the alternative branches (integers exactly).
    Version 3 of the rule (round 6): outside max(floor, 8 s) is a FAILURE (no 1 x - 2 x tail) and an integer mismatch is never
    excused by an unstable probe.  What parity_assert.assert_judged asserts about a sample on top of "no failure": the plain env-steps
    within 1e-4; at least 30 % of the env-steps plain (held to the flat 1e-4); the bounds' median <= 2e-4 and 90 % quantile <= 1e-3;
    a bound above its ceiling (`loose`) on < 1 % of the env-steps; `beyond` == 0 and `int_excused` == 0; the 99.9 % quantile of
    err / bound < 0.2; 99 % of all env-steps within 1e-4 of the oracle as it ran;
  * launch shapes and kernel variants of the SAME step are compared bitwise (array_equal), never with a tolerance;
  * free-running drift is chaotic (contacts make/break): characterised against the fp64 build, see
    tests/test_gpu_branches.py::test_closed_loop_1000_step_drift.
"""
import os
import numpy as np
import pytest

import oracle_lib as ol
import parity_rule as pr

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

KINDS = [("Walker3DStepperEnv-v0", "walker3d"), ("MikeStepperEnv-v0", "mike")]
INT_FIELDS = [ol.S_N, ol.S_COUNT, ol.S_ELAPSED, ol.S_CTRLO, ol.S_CTRHI, ol.S_FLAGS]


def gpu_env(env_id, n, seed=0, **kw):
    from steppingstone_amd.envs import SteppingStoneVecEnv
    return SteppingStoneVecEnv(env_id, n, seed=seed, device="cuda:0", return_numpy=True, **kw)


def test_library_loaded_is_in_tree():
    from steppingstone_amd import _lib
    lib = _lib.load()
    assert lib.ss_version() >= 1
    assert _lib.LIB_PATH.endswith("steppingstone_amd/lib/libsteppingstone.so") or os.environ.get("STEPPINGSTONE_LIB")


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_reset_matches_oracle(env_id, kind):
    n = 200   # not a multiple of 64 on purpose (ragged last wavefront)
    g = gpu_env(env_id, n, seed=7)
    o = ol.OracleEnv(kind, n, seed=7)
    og, oo = g.reset(), o.reset()
    assert og.shape == (n, 60) and og.dtype == np.float32
    assert np.abs(og - oo).max() < 1e-6
    sg, so = g.get_state().cpu().numpy(), o.get_state()
    assert np.array_equal(sg[:, INT_FIELDS], so[:, INT_FIELDS])
    assert np.abs(sg - so).max() < 1e-6
    g.close()


def test_random_action_stream_is_bit_exact():
    g = gpu_env("Walker3DStepperEnv-v0", 130, seed=3)
    o = ol.OracleEnv("walker3d", 130, seed=3)
    for t in (0, 1, 17, 100000):
        assert np.array_equal(g.random_actions(t).cpu().numpy(), o.random_actions(t))
    g.close()


def _judged_steps(env_id, kind, n, steps, seed, curriculum=0, on_device_actions=False, burn_in=0):
    """`steps` single control steps from identical injected states, HIP kernel vs oracle, every env-step judged by
    parity_rule.StepJudge.  on_device_actions: the step is ss_rollout_random(1 step at t) -- the benchmarked kernel
    instantiation with its own Philox actions -- instead of ss_step with the oracle's action array."""
    if on_device_actions:
        from steppingstone_amd.envs import SteppingStoneVecEnv
        g = SteppingStoneVecEnv(env_id, n, seed=seed, device="cuda:0", return_numpy=False)
    else:
        g = gpu_env(env_id, n, seed=seed)
    J = pr.StepJudge(kind, n, seed=seed, curriculum=curriculum)
    if curriculum:
        g.update_curriculum(curriculum)
    g.reset()
    for t in range(burn_in):                       # diverse states (falls, partial contacts, resets) before the judged steps
        J.o32.step(J.o32.random_actions(t))
    st = J.o32.get_state()
    res = []
    for t in range(burn_in, burn_in + steps):
        g.set_state(st)
        a = J.o32.random_actions(t)
        if on_device_actions:
            og, rg, dg = [x.cpu().numpy() for x in g.rollout_random(1, t0=t, steps_per_launch=1)]
        else:
            og, rg, dg, _ = g.step(a)
        sg = g.get_state().cpu().numpy()
        raw = g._info.cpu().numpy()
        r = J.judge(st, a, og, rg, dg.astype(bool), sg, raw[:, 2], raw[:, 4])
        res.append(r)
        st = r["next_state"]
    g.close()
    return pr.summarize(res)


from parity_assert import assert_judged as _assert_judged  # noqa: E402  (one set of thresholds for tests, smoke and the held-out tool)


def _judged_step(J, g, st, a):
    """One injected step of the HIP env g judged by J (parity_rule): asserts that every env is inside its bound."""
    g.set_state(st)
    og, rg, dg, _ = g.step(a)
    raw = g._info.cpu().numpy()
    r = J.judge(st, a, og, rg, np.asarray(dg).astype(bool), g.get_state().cpu().numpy(), raw[:, 2], raw[:, 4])
    bad = ~r["ok"]
    assert r["ok"].all(), ("env-steps outside their bound: %s (obs errors %s, bounds %s; reward errors %s, bounds %s; integers equal %s, "
                           "near a threshold %s, probe stable %s; integers [n count elapsed ctr_lo ctr_hi flags done bad update] hip %s oracle %s)" % (
        np.nonzero(bad)[0][:8], r["matched_e"][bad][:8], r["tol"][bad][:8], r["e_rew"][bad][:8], r["tol_rew"][bad][:8], r["int_ok"][bad][:8],
        r["near"][bad][:8], r["stable"][bad][:8], r["g_int"][bad][:4].tolist(), r["b_int"][bad][:4].tolist()))
    return r, og, rg, dg


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_single_step_parity_from_injected_state(env_id, kind):
    R, txt = _judged_steps(env_id, kind, n=256, steps=60, seed=11)
    _assert_judged(R, txt, "single-step parity %s, flat terrain" % kind)


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_single_step_parity_on_curriculum_terrain(env_id, kind):
    """Same rule on tilted / turned stones (curriculum 5, grid drawn stones in play after the first resets)."""
    R, txt = _judged_steps(env_id, kind, n=256, steps=40, seed=23, curriculum=5)
    _assert_judged(R, txt, "single-step parity %s, curriculum 5" % kind)


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_benchmarked_rollout_step_against_the_oracle_at_full_size(env_id, kind):
    """BASELINE size, the benchmarked instantiation: ss_rollout_random(1 step at index t) -- on-device Philox actions --
    against o.step(o.random_actions(t)) from injected states, 4096 envs, same rule (no chain through ss_step)."""
    R, txt = _judged_steps(env_id, kind, n=4096, steps=3, seed=5, curriculum=5, on_device_actions=True, burn_in=25)
    _assert_judged(R, txt, "rollout kernel %s, 4096 envs" % kind)


def _stand_on_target(o, n_envs):
    """Oracle state with every robot standing on its target stone (stone 1), so the target-advance / stone-draw
    path runs within two steps."""
    st = o.get_state()
    st[:, 0] = st[:, 65 + 6 + 0]          # x := stone 1 x
    st[:, ol.S_POT] = 0.0
    o.set_state(st)
    return st


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_target_advance_and_sampler_are_bit_exact(env_id, kind):
    n = 192
    g = gpu_env(env_id, n, seed=5)
    o = ol.OracleEnv(kind, n, seed=5)
    rng = np.random.default_rng(0)
    probs = rng.random((n, 11, 11))
    probs /= probs.sum(axis=(1, 2), keepdims=True)
    for env in (g,):
        env.update_curriculum(5)
        env.update_sample_prob(probs)
    o.set_curriculum(5)
    o.set_sample_prob(probs)
    g.reset()
    o.reset()
    g.set_state(_stand_on_target(o, n))
    zero = np.zeros((n, 21), np.float32)
    advanced = np.zeros(n, bool)
    for t in range(5):          # (released 1 cm above the stone: the feet are down for two consecutive steps by step 3-5)
        oo, ro, do, io = o.step(zero)
        og, rg, dg, ig = g.step(zero)
        raw = g._info.cpu().numpy()
        assert np.array_equal(raw[:, 4], io["update_terrain"])
        assert np.array_equal(raw[:, 3], io["steps_reached"])
        advanced |= io["update_terrain"].astype(bool)
        sg, so = g.get_state().cpu().numpy(), o.get_state()
        assert np.array_equal(sg[:, INT_FIELDS], so[:, INT_FIELDS])
        # the drawn stone (terrain rows) must agree to rounding; the sampled cell is discrete, so a wrong cell
        # would show up as a >= 4 degree / 6 degree jump
        assert np.abs(sg[:, 65:185] - so[:, 65:185]).max() < 1e-5
        g.set_state(so)
    assert advanced.mean() > 0.5     # a torque-free robot keeps a foot on the stone for 2 steps in most envs
    # step bonus was paid on first touch: reward parity covers it
    g.close()


@pytest.mark.parametrize("env_id,kind", KINDS)
def test_create_temp_states_matches_oracle(env_id, kind):
    n = 70
    g = gpu_env(env_id, n, seed=9)
    o = ol.OracleEnv(kind, n, seed=9)
    g.update_curriculum(3)
    o.set_curriculum(3)
    g.reset()
    o.reset()
    for t in range(3):
        a = o.random_actions(t)
        o.step(a)
    g.set_state(o.get_state())
    tg, to = g.create_temp_states(), o.create_temp_states()
    assert tg.shape == (n, 121, 60)
    assert np.abs(tg - to).max() < 2e-5
    # only the look-ahead block differs between grid cells
    assert np.abs(tg[:, 1:, :55] - tg[:, :1, :55]).max() == 0.0
    g.close()


def test_free_run_drift_is_characterised():
    """Both sides free-run from the same seed; divergence is chaotic once contacts differ.  Assert the short
    horizon and print the curve for DESIGN.md."""
    n = 256
    g = gpu_env("Walker3DStepperEnv-v0", n, seed=21)
    o = ol.OracleEnv("walker3d", n, seed=21)
    g.reset()
    o.reset()
    curve = []
    alive = np.ones(n, bool)
    for t in range(200):
        a = o.random_actions(t)
        oo, ro, do, _ = o.step(a)
        og, rg, dg, _ = g.step(a)
        alive &= ~(do.astype(bool) | dg)
        if alive.sum() == 0:
            break
        err = np.abs(og - oo).max(axis=1)
        curve.append((t, int(alive.sum()), float(np.median(err[alive])), float(err[alive].max())))
    print("free-run |obs| divergence (step, alive, median, max):", curve[:5], "...", curve[-3:])
    assert curve[4][3] < 5e-2 and curve[4][2] < 2e-3
    g.close()


@pytest.mark.parametrize("env_id", ["Walker3DStepperEnv-v0", "MikeStepperEnv-v0"])
def test_full_size_rollout_properties(env_id):
    """BASELINE size (4096 envs): size-independent invariants of a random-action rollout."""
    n = 4096
    from steppingstone_amd.envs import SteppingStoneVecEnv
    g = SteppingStoneVecEnv(env_id, n, seed=1, device="cuda:0", return_numpy=False)
    g.update_curriculum(5)
    g.reset()
    ndone = 0
    for t in range(80):
        obs, rew, done = g.rollout_random(1, t0=t)
        info = g._info_tensors()
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (obs[:, :50].abs() <= 5.0).all()
        d = done.bool()
        ndone += int(d.sum())
        st = g.get_state()
        qn = st[:, 3:7].norm(dim=1)
        assert (qn - 1).abs().max() < 1e-4
        # auto-reset: finished envs restart at elapsed 0 with the reset pose, the others count up
        assert (st[d, 61] == 0).all()
        assert (info["ep_len"][d] >= 1).all()
        assert (st[:, 59] >= 1).all() and (st[:, 59] <= 19).all()
    assert ndone > n        # every env fell at least once in 80 random steps on average
    g.close()


def test_rollout_matches_step_with_explicit_actions():
    """ss_step with an explicit action array (what a policy in the loop calls) and ss_rollout_random(1) (the benchmarked kernel's
    instantiation, which draws the same Philox actions on the device) are the same step: bitwise, both robots, batch sizes that select
    each kernel variant by themselves (DESIGN.md section 2 item 1 states bit-identity across launch shapes as a product requirement)."""
    from steppingstone_amd.envs import SteppingStoneVecEnv
    for env_id, n in (("Walker3DStepperEnv-v0", 128), ("MikeStepperEnv-v0", 200), ("Walker3DStepperEnv-v0", 4096)):
        a_env = SteppingStoneVecEnv(env_id, n, seed=4, device="cuda:0", return_numpy=False)
        b_env = SteppingStoneVecEnv(env_id, n, seed=4, device="cuda:0", return_numpy=False)
        for e in (a_env, b_env):
            e.update_curriculum(5)
            e.reset()
        for t in range(40):
            oa, ra, da = a_env.rollout_random(1, t0=t)
            ob, rb, db, _ = b_env.step(b_env.random_actions(t))
            assert torch.equal(oa, ob) and torch.equal(ra, rb) and torch.equal(da, db), (env_id, n, t)
        assert torch.equal(a_env.get_state(), b_env.get_state())
        a_env.close()
        b_env.close()


def test_single_env_facade_and_errors():
    from steppingstone_amd import SteppingStoneError
    from steppingstone_amd.envs import make_env
    env = make_env("mocca_envs:Walker3DStepperEnv-v0")
    env.seed(1093)
    obs = env.reset()
    assert obs.shape == (60,) and obs.dtype == np.float32
    o2, r, d, info = env.step(np.zeros(21, np.float32))
    assert isinstance(r, float) and isinstance(d, bool) and isinstance(info, dict)
    assert env.create_temp_states().shape == (121, 60)
    assert env.yaw_samples.shape == (11,) and env.pitch_samples.shape == (11,)
    assert env.terrain_info.shape == (20, 6)
    # no auto-reset for the plain gym facade: run to termination and check the terminal obs is NOT a reset obs
    for _ in range(200):
        o2, r, d, info = env.step(np.zeros(21, np.float32))
        if d:
            break
    assert d and "episode" in info
    assert abs(o2[0]) > 0.05     # torso height changed: terminal observation, not the reset pose
    env.close()
    with pytest.raises(SteppingStoneError):
        make_env("NoSuchEnv-v0")


def test_ppo_driver_end_to_end_on_gpu():
    """BASELINE configs[4] in miniature: Mike, curriculum on, device-resident PPO for two updates."""
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", 512, seed=8, device="cuda:0", return_numpy=False)
    ac, hist = ppo.train(envs, num_updates=2, num_steps=16, ppo_epoch=2, mini_batch_size=1024, use_mirror=True, log=None)
    assert len(hist) == 2 and all(np.isfinite([h["value_loss"], h["action_loss"], h["entropy"]]).all() for h in hist)
    assert next(ac.parameters()).is_cuda
    envs.close()


def test_gpu_is_deterministic_and_handles_tiny_batches():
    outs = []
    for rep in range(2):
        g = gpu_env("MikeStepperEnv-v0", 33, seed=17)
        g.update_curriculum(4)
        g.reset()
        acc = []
        for t in range(40):
            o, r, d, _ = g.step(g.random_actions(t))
            acc.append(np.concatenate([o, r[:, None], d[:, None].astype(np.float64)], axis=1))
        outs.append(np.array(acc))
        g.close()
    assert np.array_equal(outs[0], outs[1])           # bitwise run-to-run
    one = gpu_env("Walker3DStepperEnv-v0", 1, seed=2)            # a single env: one lane pair of one wavefront
    J = pr.StepJudge("walker3d", 1, seed=2)
    assert np.abs(one.reset() - J.o32.get_obs()).max() < 1e-6
    st = J.o32.get_state()
    for t in range(8):
        r, _, _, _ = _judged_step(J, one, st, J.o32.random_actions(t))
        st = r["next_state"]
    one.close()


def test_episode_statistics_match_oracle():
    """Distribution-level parity over free-running rollouts (chaotic trajectories diverge, statistics must not):
    episode lengths and returns of 1024 envs x 150 random-action steps."""
    n, steps = 1024, 150
    g = gpu_env("Walker3DStepperEnv-v0", n, seed=31)
    o = ol.OracleEnv("walker3d", n, seed=31)
    g.reset()
    o.reset()
    lg, lo_, rg_, ro_ = [], [], [], []
    for t in range(steps):
        a = o.random_actions(t)
        _, _, dg, _ = g.step(a)
        raw = g._info.cpu().numpy()
        fl = raw.view(np.float32)
        lg += list(fl[dg, 1]); rg_ += list(fl[dg, 0])
        _, _, do, io = o.step(a)
        m = do.astype(bool)
        lo_ += list(io["ep_len"][m]); ro_ += list(io["ep_ret"][m])
    assert len(lg) > 2000 and abs(len(lg) - len(lo_)) < 0.03 * len(lo_)
    assert abs(np.mean(lg) - np.mean(lo_)) < 0.03 * np.mean(lo_), (np.mean(lg), np.mean(lo_))
    assert abs(np.mean(rg_) - np.mean(ro_)) < 0.05 * abs(np.mean(ro_)) + 0.5, (np.mean(rg_), np.mean(ro_))
    g.close()


def test_packed_step_equals_plain_step():
    """The [N,62] packed output (what a multi-GPU shard all-gathers) carries exactly obs | rew | done of ss_step."""
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 200
    a_env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=6, device="cuda:0", return_numpy=False)
    b_env = SteppingStoneVecEnv("Walker3DStepperEnv-v0", n, seed=6, device="cuda:0", return_numpy=False)
    sh = ShardedVecEnv(b_env)                 # world size 1: no collective, same code path otherwise
    assert torch.equal(a_env.reset(), sh.reset())
    for t in range(6):
        act = a_env.random_actions(t)
        o1, r1, d1, _ = a_env.step(act)
        o2, r2, d2, _ = sh.step(act)
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    o1, r1, d1 = a_env.rollout_random(3, t0=50)
    o2, r2, d2 = sh.rollout_random(3, t0=50)
    assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1.bool(), d2)
    a_env.close()
    b_env.close()


def test_bench_rccl_path_on_one_rank():
    """bench.py under torch.distributed.run with one rank and SS_FORCE_COLLECTIVE=1: process-group init on RCCL,
    the per-step asynchronous all-gather of the packed block, barriers and the max-over-ranks reduction all run."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SS_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29517", os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "20",
           "--no-cpu-baseline", "--peer-store"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    js = json.loads(line)
    assert js["config"]["parallelism"].endswith("+allgather")
    assert js["value"] > 1e6 and js["n_gpus"] == 1
    assert js["no_gather"]["ms_per_step"] > 0 and js["peer_store"]["ms_per_step"] > 0 and js["peer_store"]["wait_timeouts"] == 0
    # the exchange proved itself over real RCCL (one rank: the rank's own block came back as sent), and says what it ran on
    assert js["gather_verified"] is True and js["rccl_ranks"] == 1 and js["transport"] == "nccl (RCCL)" and js["rccl_version"]
    assert js["per_step_gather"]["value"] > 0 and "all-gather of [N/G,62]" in js["policy_in_the_loop"]["what"]


def test_ppo_graph_replay_matches_eager():
    """The hipGraph-captured minibatch step (gather, forward, backward, clip, capturable Adam) and the captured
    rollout give the same numbers as the eager path: same weights after 8 minibatch steps, same storage after a rollout."""
    from steppingstone_amd import ppo
    from steppingstone_amd.envs import SteppingStoneVecEnv
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    data = (torch.randn(4096, 60, device=dev), torch.randn(4096, 21, device=dev).clamp(-1, 1), torch.randn(4096, 1, device=dev),
            torch.randn(4096, 1, device=dev), -20 + torch.randn(4096, 1, device=dev), torch.randn(4096, 1, device=dev))
    finals = []
    for use_graph in (False, True):
        torch.manual_seed(5)
        ac = ppo.ActorCritic(num_ensembles=2).to(dev)
        agent = ppo.PPO(ac, mini_batch_size=512, use_graph=use_graph)
        g = torch.Generator(device=dev); g.manual_seed(11)
        for k in range(8):
            idx = torch.randperm(4096, device=dev, generator=g)[:512]
            out = agent._graph_step(data, idx, refresh=(k == 0)) if use_graph else torch.stack(agent._gathered_step(data, idx))
        assert (not use_graph) or agent._graph is not None
        finals.append((torch.cat([p.detach().reshape(-1) for p in ac.parameters()]).clone(), out.clone()))
    # capturable Adam orders its arithmetic differently from the default one: weights agree to a fraction of the
    # 8 x lr = 2.4e-3 they can have moved, the last minibatch's losses to 1 %
    assert torch.allclose(finals[0][0], finals[1][0], atol=5e-4), float((finals[0][0] - finals[1][0]).abs().max())
    assert torch.allclose(finals[0][1], finals[1][1], rtol=1e-2, atol=1e-4)
    # rollout: three collector calls (eager warm-up, capture + replay, replay).  The sampling noise of a replayed graph comes from other
    # generator offsets than the eager loop's, so the captured rollout is checked against what it must be, not against another noise
    # stream: (a) the whole run repeated from the same seeds reproduces every stored tensor bit for bit (replay determinism);
    # (b) the ACTIONS the graphed run stored, fed step by step to a fresh env from the same reset through plain ss_step launches,
    # reproduce the stored observations, rewards and masks bit for bit -- i.e. the captured graph really is policy -> env -> storage.
    def graphed_run():
        torch.manual_seed(9)
        envs = SteppingStoneVecEnv("MikeStepperEnv-v0", 256, seed=4, device=dev, return_numpy=False)
        ac = ppo.ActorCritic().to(dev)
        roll = ppo.Rollouts(8, 256, dev)
        roll.obs[0].copy_(envs.reset())
        col = ppo.GraphedCollector(envs, ac, roll, 8)
        chunks = []
        for _ in range(3):
            col()
            chunks.append((roll.obs.clone(), roll.actions.clone(), roll.rewards.clone(), roll.masks.clone(), roll.logp.clone(), roll.value_preds.clone()))
            roll.after_update()
        assert col.graph is not None
        envs.close()
        return chunks
    run1, run2 = graphed_run(), graphed_run()
    for c1, c2 in zip(run1, run2):
        assert all(torch.equal(a, b) for a, b in zip(c1, c2)), "graph replay is not deterministic"
    envs = SteppingStoneVecEnv("MikeStepperEnv-v0", 256, seed=4, device=dev, return_numpy=False)
    obs = envs.reset()
    for obs_c, act_c, rew_c, mask_c, _, _ in run1:
        assert torch.equal(obs_c[0], obs)
        for t in range(8):
            obs, rew, done, _ = envs.step(act_c[t])
            assert torch.equal(obs, obs_c[t + 1]) and torch.equal(rew, rew_c[t, :, 0]), t
            assert torch.equal(1.0 - done.to(torch.float32), mask_c[t + 1, :, 0])
    envs.close()
    assert all(torch.isfinite(c[4]).all() and torch.isfinite(c[5][:-1]).all() for c in run1)      # log-probabilities and values were stored


def test_large_batch_is_a_union_of_small_ones():
    """Maximum-size property (131072 envs = 4096 wavefronts, four per CU): every env's trajectory is keyed by its
    global id only, so a 64-env instance created at env_id_offset=k reproduces envs [k, k+64) of the big batch
    bit for bit -- across wavefront placement, occupancy and batch size."""
    from steppingstone_amd.envs import SteppingStoneVecEnv
    big = SteppingStoneVecEnv("MikeStepperEnv-v0", 131072, seed=21, device="cuda:0", return_numpy=False)
    big.update_curriculum(5)
    big.reset()
    for t in range(6):
        ob, rb, db = big.rollout_random(1, t0=t)
    assert torch.isfinite(ob).all() and torch.isfinite(rb).all()
    for k in (0, 70000, 131072 - 64):
        small = SteppingStoneVecEnv("MikeStepperEnv-v0", 64, seed=21, device="cuda:0", env_id_offset=k, return_numpy=False)
        small.update_curriculum(5)
        small.reset()
        for t in range(6):
            osm, rsm, dsm = small.rollout_random(1, t0=t)
        assert torch.equal(osm, ob[k:k + 64]) and torch.equal(rsm, rb[k:k + 64]) and torch.equal(dsm, db[k:k + 64])
        small.close()
    big.close()


def test_non_finite_and_out_of_range_actions_are_contained():
    """PHYSICS.md 4.8 / 2: actions are clipped to [-1,1] (so is +-Inf) and a NaN action may not poison the state: the
    episode ends (done, zero reward), the env auto-resets, every output stays finite, neighbours are unaffected, and the
    oracle makes the same decisions."""
    n = 64
    g = gpu_env("Walker3DStepperEnv-v0", n, seed=12)
    o = ol.OracleEnv("walker3d", n, seed=12)
    assert np.abs(g.reset() - o.reset()).max() < 1e-6
    a = o.random_actions(0).copy()
    a[3, 5] = np.nan
    a[7, :] = np.inf
    a[9, 2] = -np.inf
    a[11, :] = 50.0            # far out of range: clipped, not an error
    st0 = o.get_state()
    g.set_state(st0)           # identical inputs on both sides (the reset states agree to rounding only)
    og, rg, dg, _ = g.step(a)
    raw, sg = g._info.cpu().numpy(), g.get_state().cpu().numpy()
    oo, ro, do, _ = o.step(a)
    assert np.isfinite(og).all() and np.isfinite(rg).all()
    assert np.array_equal(dg, do)
    assert dg[3] and not dg[7] and not dg[9] and not dg[11]      # NaN ends the episode; +-Inf and 50 are clipped to +-1
    assert rg[3] == 0.0
    clean = np.ones(n, bool)
    clean[3] = False                                            # the NaN env: both sides return the reset observation
    assert np.abs(og[3] - oo[3]).max() < 1e-6
    # every other env -- the saturated ones (+-Inf, 50) included -- inside the parity rule's bound of ITS env-step (a flat 1e-4 does
    # not hold on a full-torque step of a robot standing on its soles: 1.4e-4 seen on one of the 63 envs)
    J = pr.StepJudge("walker3d", n, seed=12)
    a_judge = a.copy()
    a_judge[3] = 0.0                                            # (the judge's sensitivity probes need finite inputs; env 3 is not judged)
    r0 = J.judge(st0, a_judge, og, rg, np.asarray(dg).astype(bool), sg, raw[:, 2], raw[:, 4])
    assert r0["ok"][clean].all(), (np.nonzero(~r0["ok"] & clean)[0], r0["matched_e"][~r0["ok"] & clean], r0["tol"][~r0["ok"] & clean])
    assert np.quantile(np.abs(og - oo)[clean].max(axis=1), 0.9) < 1e-4
    # the following step runs normally for everyone (the poisoned env was reset): judged by the parity rule
    r, og, rg, dg = _judged_step(J, g, o.get_state(), o.random_actions(1))
    assert np.isfinite(og).all()
    g.close()


def test_gpu_rounding_error_is_comparable_to_the_fp32_cpu_path():
    """Accuracy against an fp64 evaluation of the same specification: one control step from identical injected
    states and actions by (a) the HIP kernel, (b) the fp32 oracle, (c) the fp64 oracle.  The kernel's distance to (c)
    must be of the size of (b)'s distance to (c) -- i.e. its deviations from the fp32 oracle are rounding, not
    algorithm.  Medians and 95th percentiles over 256 envs x 40 states (contact-set flips live in the tail)."""
    n, steps = 256, 40
    g = gpu_env("Walker3DStepperEnv-v0", n, seed=31)
    o32 = ol.OracleEnv("walker3d", n, seed=31)
    o64 = ol.OracleEnv("walker3d", n, seed=31, prec="f64")
    g.reset(); o32.reset(); o64.reset()
    eg, e32 = [], []
    for t in range(steps):
        st = o32.get_state()
        g.set_state(st)
        o64.set_state(st.astype(np.float64) if st.dtype != np.float64 else st)
        a = o32.random_actions(t)
        og = g.step(a)[0]
        o_32 = o32.step(a)[0]
        o_64 = o64.step(a)[0]
        eg.append(np.abs(og - o_64).max(axis=1))
        e32.append(np.abs(o_32 - o_64).max(axis=1))
    eg, e32 = np.concatenate(eg), np.concatenate(e32)
    med_g, med_32 = np.median(eg), np.median(e32)
    p95_g, p95_32 = np.percentile(eg, 95), np.percentile(e32, 95)
    print("max |obs - fp64| per env-step: kernel median %.2e p95 %.2e ; fp32 oracle median %.2e p95 %.2e" % (med_g, p95_g, med_32, p95_32))
    assert med_g < 4 * med_32 + 1e-7 and p95_g < 4 * p95_32 + 1e-6
    assert med_g < 1e-4          # the north-star's per-step bound, against fp64 of OUR specification (PyBullet is absent)
    g.close()


def test_numpy_mode_pinned_path_equals_tensor_mode():
    """The reference-style caller (numpy in / numpy out + info dicts) goes through pinned staging and the packed block;
    it must return exactly what the tensor mode returns, fresh arrays every step."""
    from steppingstone_amd.envs import SteppingStoneVecEnv
    n = 300
    a_env = gpu_env("MikeStepperEnv-v0", n, seed=5)
    b_env = SteppingStoneVecEnv("MikeStepperEnv-v0", n, seed=5, device="cuda:0", return_numpy=False)
    assert a_env._pinned is not None
    assert np.array_equal(a_env.reset(), b_env.reset().cpu().numpy())
    prev = None
    for t in range(40):
        act = b_env.random_actions(t)
        o1, r1, d1, infos = a_env.step(act.cpu().numpy())
        o2, r2, d2, it = b_env.step(act)
        assert o1.dtype == np.float32 and r1.dtype == np.float64 and d1.dtype == np.bool_ and len(infos) == n
        assert np.array_equal(o1, o2.cpu().numpy()) and np.array_equal(r1, r2.cpu().numpy().astype(np.float64))
        assert np.array_equal(d1, d2.cpu().numpy())
        for i in np.nonzero(d1)[0]:
            assert abs(infos[i]["episode"]["r"] - float(it["ep_ret"][i])) < 1e-5 and infos[i]["episode"]["l"] == int(it["ep_len"][i])
        assert prev is None or prev is not o1
        prev = o1
    a_env.close(); b_env.close()


def test_remaining_hooks_match_oracle():
    """update_specialist (ring window of the sampler), set_robot_params(power), auto-reset off (plain gym semantics of
    make_env callers): same decisions and numbers as the oracle."""
    n = 128
    g = gpu_env("Walker3DStepperEnv-v0", n, seed=41)
    o = ol.OracleEnv("walker3d", n, seed=41)
    g.update_specialist(3); o.set_specialist(3)
    g.set_robot_params({"power": 0.5}); o.set_power(0.5)
    assert np.abs(g.reset() - o.reset()).max() < 1e-6
    J = pr.StepJudge("walker3d", n, seed=41, setup=lambda x: (x.set_specialist(3), x.set_power(0.5)))
    st = o.get_state()
    for t in range(12):                                         # every env-step inside its bound under the hooks too
        r, _, _, _ = _judged_step(J, g, st, o.random_actions(t))
        st = r["next_state"]
    o.set_state(st)
    # specialist ring: stones drawn while standing on the target come from cells at Chebyshev distance 3 only
    st = _stand_on_target(o, n)
    zero = np.zeros((n, 21), np.float32)
    advanced = 0
    for t in range(3):
        # judged by the rule (a robot sagging under zero torques puts sole corners on their touch threshold: an env-step whose
        # near-threshold decision the kernel takes the other way is matched on that branch by the rule, not by bit equality --
        # round 6: with the re-identified Walker3D one of 128 envs does exactly that here); the envs on the oracle's own branch
        # must have the oracle's integers and the oracle's freshly drawn stones
        r, _, _, _ = _judged_step(J, g, st, zero)
        sg, so = g.get_state().cpu().numpy(), r["next_state"]
        same = r["category"] < 2
        assert same.mean() >= 0.9
        assert np.array_equal(sg[same][:, INT_FIELDS], so[same][:, INT_FIELDS]) and np.abs(sg[same][:, 65:185] - so[same][:, 65:185]).max() < 1e-5
        advanced += int(r["oracle"]["info"]["update_terrain"].sum())
        st = so
    assert advanced >= n // 2                                    # the ring was actually drawn from
    o.set_state(st)
    # auto-reset off: a finished env reports done and keeps its terminal observation until reset() is called
    g.backend.set_auto_reset(False); o.set_auto_reset(0)
    st = o.get_state(); st[:8, 2] -= 3.0         # drop eight robots far below the fall threshold
    o.set_state(st); g.set_state(st)
    J2 = pr.StepJudge("walker3d", n, seed=41, setup=lambda x: (x.set_specialist(3), x.set_power(0.5), x.set_auto_reset(0)))
    r, og, rg, dg = _judged_step(J2, g, st, zero)
    assert np.asarray(dg)[:8].all() and np.array_equal(np.asarray(dg).astype(bool), r["oracle"]["done"].astype(bool))
    g.close()


def test_kernel_variants_are_bitwise_identical():
    """The plain step kernel and the helper-wavefront variants (1 or 3 extra wavefronts computing the contact operators;
    chosen by batch size) must produce the same bits: results may not depend on batch size or GPU count."""
    from steppingstone_amd.envs import SteppingStoneVecEnv
    outs = []
    old = os.environ.get("SS_HELPERS")
    try:
        for h in ("0", "1", "3"):
            os.environ["SS_HELPERS"] = h
            e = SteppingStoneVecEnv("MikeStepperEnv-v0", 1000, seed=21, device="cuda:0", return_numpy=False)
            e.update_curriculum(5)
            e.reset()
            for t in range(40):
                o, r, d = e.rollout_random(1, t0=t)
            outs.append((o.clone(), r.clone(), d.clone(), e.get_state().clone()))
            e.close()
    finally:
        if old is None:
            os.environ.pop("SS_HELPERS", None)
        else:
            os.environ["SS_HELPERS"] = old
    for k in (1, 2):
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[k])), "variant %d differs" % k


def test_data_parallel_minibatch_step_is_captured_with_its_all_reduce():
    """the data-parallel minibatch step of ppo.train (gather, forward, backward, ONE RCCL all-reduce of the flat
    gradient, clip, capturable Adam) as a hipGraph.  No multi-GPU box here, so the collective is forced on a 1-rank RCCL group
    (force_collective): the capture must succeed (no fallback) and replays must reproduce the eager data-parallel steps.
    Runs in a process of its own (tests/dp_graph_capture_check.py): a native abort inside RCCL / the graph capture (round 5 saw one
    SIGABRT in autograd's backward under capture, not reproduced) then fails THIS test instead of taking the whole suite down."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "dp_graph_capture_check.py")], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "dp graph capture ok" in out.stdout, (out.returncode, out.stdout[-1500:], out.stderr[-3000:])
