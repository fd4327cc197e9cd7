#!/bin/bash
# Round 6, GPU call B: first run of the PLANK kernels on the MI355X -- GPU suite, smoke, bench (default + driver shape).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-r06_b}
O=gpurun_out
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q -rs --durations=15 2>&1; echo "pytest rc $?" ) > $O/${tag}_pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${tag}_smoke.log 2>&1
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${tag}_bench_driver_shape.json 2>/dev/null
tail -4 $O/${tag}_pytest_gpu.log; tail -2 $O/${tag}_smoke.log | cut -c1-300; cut -c1-300 $O/${tag}_bench.json; cut -c1-300 $O/${tag}_bench_driver_shape.json
