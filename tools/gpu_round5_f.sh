#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=r05_v1
timeout 2400 python -m pytest tests -m gpu -x -q -rs --durations=12 -s > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/${tag}_pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/${tag}_smoke.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_shape.json 2>> gpurun_out/${tag}_bench.err
grep -E "passed|failed|on the MI355X|rc |closed-loop .* step  *10 |closed-loop .* step  999" gpurun_out/${tag}_pytest_gpu.log | cut -c1-260 | tail -12; tail -2 gpurun_out/${tag}_smoke.log | cut -c1-200; cut -c1-220 gpurun_out/${tag}_bench.json; cut -c1-220 gpurun_out/${tag}_bench_driver_shape.json
