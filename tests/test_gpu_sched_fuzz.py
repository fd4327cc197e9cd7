"""Schedule fuzzing as a standing test: the -DSS_FUZZ_SCHED build of the product's own kernel
sources (steppingstone_amd/lib/libsteppingstone_fuzz.so, built by steppingstone_amd.build.build_fuzz) makes every wavefront sleep a
pseudo-random time -- seeded by the shader clock, so different in every run -- at the start of every barrier window of the main /
helper schedule and at the hand-over points of a control step.  Its results must be the bits of the product library on every
env-step: tools/sched_fuzz.py prints a checksum of every step's packed block, info words and the final state per configuration
(both robots x plain / one-helper / three-helper kernels x explicit actions / on-device actions / multi-step launches, and ragged tiny
batches); this test runs it once with the product library and twice with the fuzzed one and compares the lines.

Round 4's full-size run of the same tool (>= 1e7 env-steps per configuration, profiles/r04_sched_fuzz_*.txt) is what found the
schedule-dependent reward / done words of env n - 1's packed row at odd batch sizes with n mod 64 < 32."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lines_of(lib, target, tiny_steps):
    env = dict(os.environ)
    if lib:
        env["STEPPINGSTONE_LIB"] = lib
    else:
        env.pop("STEPPINGSTONE_LIB", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sched_fuzz.py"), str(target), str(tiny_steps)], env=env,
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l and not l.startswith("#") and "amdgpu.ids" not in l]


@pytest.mark.gpu
def test_fuzzed_schedule_produces_the_same_bits():
    from steppingstone_amd import build
    fuzz = build.build_fuzz()
    plain = lines_of(None, 1_000_000, 200)
    assert len(plain) >= 30 + 48
    # within the product library: explicit actions == on-device actions (same Philox stream), and every helper variant agrees
    by_key = {}
    for l in plain:
        f = l.split()
        key = (f[0], f[1], "multi" if f[3] not in ("steps/launch=0", "steps/launch=1") else "single")
        by_key.setdefault(key, set()).add(f[-1])
    assert all(len(v) == 1 for v in by_key.values()), {k: v for k, v in by_key.items() if len(v) > 1}
    for run in range(2):
        fuzzed = lines_of(fuzz, 1_000_000, 200)
        diff = [(a, b) for a, b in zip(plain, fuzzed) if a != b]
        assert len(fuzzed) == len(plain) and not diff, "schedule-dependent results (run %d): %s" % (run, diff[:6])
