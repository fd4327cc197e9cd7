#!/bin/bash
# GPU call C of round 5: the whole GPU suite on the 5-sweep warm-started spec (no -x: every failure), then once more exactly as the driver runs it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -rs --durations=15 > $O/pytest_gpu_all.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu_all.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
tail -40 $O/pytest_gpu_all.log | cut -c1-300; tail -2 $O/smoke.log | cut -c1-200
