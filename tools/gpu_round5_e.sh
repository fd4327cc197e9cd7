#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=r05_v1
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=12 -s > gpurun_out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/${tag}_pytest_gpu.log
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu_as_the_driver_runs_it.log 2>&1; echo "pytest rc $?" >> gpurun_out/${tag}_pytest_gpu_as_the_driver_runs_it.log
rm -rf gpurun_out/heldout_policies
timeout 2700 python tools/parity_heldout.py --json gpurun_out/${tag}b_parity_heldout.json > gpurun_out/${tag}b_parity_heldout.txt 2> gpurun_out/${tag}b_parity_heldout.err
echo "heldout rc $?" >> gpurun_out/${tag}b_parity_heldout.txt
grep -E "passed|failed|on the MI355X|rc " gpurun_out/${tag}_pytest_gpu.log | cut -c1-300 | tail; tail -2 gpurun_out/${tag}_pytest_gpu_as_the_driver_runs_it.log; tail -3 gpurun_out/${tag}b_parity_heldout.txt | cut -c1-700
