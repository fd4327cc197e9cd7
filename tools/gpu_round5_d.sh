#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=8 -s > $O/pytest_gpu_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
grep -E "passed|failed|FAILED|on the MI355X|rc " $O/pytest_gpu_all.log | cut -c1-300 | tail -30; tail -2 $O/smoke.log | cut -c1-200; cut -c1-200 $O/bench.json
