#!/bin/bash
# GPU call A of round 5: issue-model probe, full GPU suite (with the new 8-rank and bitwise tests), smoke, held-out parity, bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 120 var/issue_model_probe > $O/issue_model_probe.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q -rs --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 2400 python tools/parity_heldout.py --json $O/parity_heldout.json > $O/parity_heldout.txt 2> $O/parity_heldout.err; echo "heldout rc $?" >> $O/parity_heldout.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_shape.json 2>> $O/bench.err
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -3 $O/parity_heldout.txt; cut -c1-300 $O/bench.json
