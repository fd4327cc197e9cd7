#!/bin/bash
# Round 6, first GPU call (run ON the GPU box via gpurun): the round-5 kernels and specification, unchanged --
#   GPU suite with the new policy-driven parity cells and the configs[4] 8-rank job, smoke, bench (default + driver shape),
#   rocprofv3 --kernel-trace --stats of the bench command, and the FRESH held-out sample that validates parity rule version 3.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-r06_a}
O=gpurun_out
mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -x -q -rs --durations=15 2>&1; echo "pytest rc $?" ) > $O/${tag}_pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${tag}_smoke.log 2>&1
python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err
python bench.py --steps 20 --warmup 5 > $O/${tag}_bench_driver_shape.json 2>/dev/null
( cd /tmp && rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OLDPWD/$O/${tag}_bench_driver_shape_under_rocprof.json 2>/dev/null; cp $(ls /tmp/kt/*/*kernel_stats.csv | head -1) $OLDPWD/$O/${tag}_rocprofv3_kernel_stats_driver_shape.csv )
rm -rf $O/heldout_policies
timeout 2700 python tools/parity_heldout.py --seed-base ${2:-59001} --json $O/${tag}_parity_heldout.json > $O/${tag}_parity_heldout.txt 2> $O/${tag}_parity_heldout.err
echo "heldout rc $?" >> $O/${tag}_parity_heldout.txt
tail -4 $O/${tag}_pytest_gpu.log; tail -2 $O/${tag}_smoke.log | cut -c1-300; tail -3 $O/${tag}_parity_heldout.txt | cut -c1-500; cut -c1-300 $O/${tag}_bench.json; cut -c1-300 $O/${tag}_bench_driver_shape.json
