"""`python bench.py --gpus N` / `python -m steppingstone_amd.train --gpus N` start their own N ranks when no launcher
did (the reference's make_vec_envs forks its own workers, common/envs_utils.py:519-538).
CPU only: --dry-launch makes every rank report RANK / WORLD_SIZE and exit before touching a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(text):
    """JSON objects of the ranks' shared stdout (robust against two ranks' lines landing on one line)"""
    dec, rows = json.JSONDecoder(), []
    for line in text.splitlines():
        line = line.strip()
        while line.startswith("{"):
            obj, end = dec.raw_decode(line)
            rows.append(obj)
            line = line[end:].strip()
    return rows


def _env():
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return e


def test_bench_self_launches_two_ranks():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], cwd=ROOT, env=_env(),
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = _rows(out.stdout)
    assert sorted(r["rank"] for r in rows) == [0, 1] and all(r["world_size"] == 2 for r in rows)
    assert all(r["master"].startswith("127.0.0.1:") for r in rows)


def test_bench_under_an_external_launcher_does_not_relaunch():
    from steppingstone_amd import launch
    cmd = launch.launcher_command(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"])
    out = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = _rows(out.stdout)
    assert len(rows) == 2                                   # two ranks, each printed once: no nested launch


def test_rank_count_mismatch_is_an_error():
    e = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], cwd=ROOT, env=e,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "launcher started 3 ranks" in (out.stderr + out.stdout)


def test_bench_watchdog_prints_a_json_error_line_instead_of_hanging():
    """bench.py's watchdog: a rank still running after --watchdog seconds prints ONE JSON line naming its phase and
    exits with code 3 -- so that a hung collective at N > 1 leaves a reason in the driver's SCALE file, not a timeout."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time, types; sys.path.insert(0, %r); import bench\n"
            "bench.PHASE['name'] = 'timed rollout (test)'\n"
            "bench.start_watchdog(types.SimpleNamespace(watchdog=0.5, gpus=8), 3, 8)\n"
            "time.sleep(30)\n" % root)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["rank"] == 3 and line["world_size"] == 8 and "timed rollout (test)" in line["error"] and line["value"] is None
    # default: on at N > 1 (900 s), off at N = 1
    import bench, types
    assert bench.start_watchdog(types.SimpleNamespace(watchdog=-1.0, gpus=1), 0, 1) is None
    t = bench.start_watchdog(types.SimpleNamespace(watchdog=-1.0, gpus=8), 0, 8)
    assert t is not None and t.interval == 900.0
    t.cancel()
