"""bench.py's host-side helpers that need no GPU: the launch list of the dominant kernel (run-length form since round 6: the driver's
20-step shape repeats its launch thousands of times) and the per-launch average a `rocprofv3 --stats` summary of the same command must
show for it (the contract's cross-check between the line and the committed rocprof summary).  CPU only."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(m)          # main() is guarded by __name__
    finally:
        sys.argv = argv
    return m


def test_launch_shape_is_run_length_coded_and_prices_the_rocprof_average():
    b = _bench()
    d = b.launch_shape(5, 20 * 5249, 20, 0.0472, 256)              # the driver's shape: 5 warm-up steps, 5249 regions of one 20-step launch
    assert d["launch_steps"] == {"prewarm": 256, "warmup": [5], "timed_steps_x_launches": [[20, 5249]]}
    launches = [256, 5] + [20] * 5249
    assert abs(d["rocprofv3_stats_average_ms_expected"] - 0.0472 * sum(launches) / len(launches)) < 1e-12
    d = b.launch_shape(200, 2000 * 3 + 500, 1000, 0.0453, 256)     # ragged tail: a 500-step launch after six 1000-step ones
    assert d["launch_steps"]["timed_steps_x_launches"] == [[1000, 6], [500, 1]] and d["launch_steps"]["warmup"] == [200]
    d = b.launch_shape(5, 20, 1, 0.056)                            # one launch per step: no clock-ramp launch of this kernel
    assert d["launch_steps"]["timed_steps_x_launches"] == [[1, 20]] and abs(d["rocprofv3_stats_average_ms_expected"] - 0.056) < 1e-12


def test_algorithmic_bytes_follow_the_state_layout():
    b = _bench()
    # 90 f32 + 4 i32 read; 60 f32 + 5 i32 state, obs 240 B, rew 4 B, done 1 B, info 24 B written
    assert b.ALGO_READ_B == 90 * 4 + 4 * 4 and b.ALGO_WRITE_B == 60 * 4 + 5 * 4 + 240 + 4 + 1 + 24
