"""The first launch of a fresh process, checked from inside the launch.

Round 3 twice saw the FIRST step of a new environment differ from identical repetitions after it (about one step in 1e5).  A first
launch fetches its code through a cold instruction cache, so the main and the helper wavefronts of a workgroup run with a relative
timing no steady-state test produces.  This test makes first launches cheap and self-checking: tests/host/first_launch_check.c (plain
C over the C ABI, no Python / torch in the process) creates 2N environments in which env e and env e + N share their global id and
their injected state, runs ONE launch of the step kernel (or the K-step rollout kernel) as the first launch of its process, and
requires (a) the two copies of every env to be bit-equal -- observation, reward, done, info words, the full state afterwards -- and
(b) three further launches from the same state to reproduce the first bit for bit.  56 fresh processes per run: both robots, one
workgroup (2 x 16 envs) and the benchmark's shape (2 x 2048 envs = 128 workgroups with three helper wavefronts each), explicit-action
step kernel and rollout kernel.

The -m "not gpu" half builds the client and checks the duplicate-id arithmetic of the inputs on the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROCESSES_PER_CASE = 7
sys.path.insert(0, os.path.join(ROOT, "tools"))
from build_c_client import build_client  # noqa: E402  (tools/build_c_client.py: ROCm root resolved, not hard-coded)


def write_inputs(path, kind, n):
    """n contact-rich states (curriculum 5, 12 random-action steps of the oracle) followed by n actions, float32."""
    o = ol.OracleEnv(kind, n, seed=2)
    o.set_curriculum(5)
    o.reset()
    for t in range(12):
        o.step(o.random_actions(t))
    st = o.get_state().astype(np.float32)
    act = o.random_actions(50).astype(np.float32)
    with open(path, "wb") as f:
        f.write(st.tobytes())
        f.write(act.tobytes())
    return st, act


def test_client_builds_and_inputs_are_well_formed(tmp_path):
    exe = build_client()
    assert os.access(exe, os.X_OK)
    st, act = write_inputs(str(tmp_path / "in.bin"), "walker3d", 16)
    assert st.shape == (16, 186) and act.shape == (16, 21) and os.path.getsize(str(tmp_path / "in.bin")) == 16 * (186 + 21) * 4
    assert np.isfinite(st).all() and (np.abs(act) <= 1).all()
    # without a GPU the client must stop at ss_create (no CPU fallback), not crash
    import torch
    if not torch.cuda.is_available():
        from steppingstone_amd import build
        out = subprocess.run([exe, build.build(), "0", "16", str(tmp_path / "in.bin"), "step"], capture_output=True, text=True, timeout=120)
        assert out.returncode in (2, 3), (out.returncode, out.stdout, out.stderr)


@pytest.mark.gpu
def test_first_launch_of_fresh_processes_is_self_consistent(tmp_path):
    from steppingstone_amd import _lib
    exe = build_client()
    lines, failures = [], []
    for kind_i, kind in enumerate(("walker3d", "mike")):
        for n in (16, 2048):
            path = str(tmp_path / ("%s_%d.bin" % (kind, n)))
            write_inputs(path, kind, n)
            for mode in ("step", "rollout"):
                for rep in range(PROCESSES_PER_CASE):
                    out = subprocess.run([exe, _lib.LIB_PATH, str(kind_i), str(n), path, mode], capture_output=True, text=True, timeout=300)
                    lines.append(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "(no output) " + out.stderr.strip())
                    if out.returncode != 0:
                        failures.append((kind, n, mode, rep, out.returncode, out.stdout[-2000:], out.stderr[-500:]))
    print("%d fresh processes; last line of each kind:" % len(lines))
    for l in sorted(set(lines)):
        print("   %3d x %s" % (lines.count(l), l))
    assert len(lines) >= 50
    assert not failures, failures
