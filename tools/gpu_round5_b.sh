#!/bin/bash
# GPU call B of round 5: issue-model probe (span over all wavefronts), A/B of the PGS settings, GPU suite on the 5-sweep warm-started spec.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 120 var/issue_model_probe > $O/issue_model_probe.txt 2>&1
ab() { # label lib envs
  local lib=""; [ -n "$2" ] && lib="$PWD/$2"
  STEPPINGSTONE_LIB=$lib timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra --envs-per-gpu $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('%-34s envs %6d  rollout %.4f ms/step  per-step-launch %.4f ms/step' % ('$1', $3, d['ms_per_step'], d['per_step_launch']['ms_per_step']))"
}
{
for r in 1 2; do
  ab "8 cold (rounds 1-4)" var/libss_base.so 4096
  ab "5 cold" var/libss_pgs5.so 4096
  ab "4 cold" var/libss_pgs4.so 4096
  ab "5 warm (HEAD; LDS below 3 helpers)" "" 4096
  ab "5 warm, impulses in registers" var/libss_warmreg.so 4096
done
for n in 16384 32768; do
  ab "8 cold (rounds 1-4)" var/libss_base.so $n
  ab "5 warm (HEAD; LDS below 3 helpers)" "" $n
  ab "5 warm, impulses in registers" var/libss_warmreg.so $n
done
} > $O/ab_pgs.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q -rs --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_shape.json 2>> $O/bench.err
cat $O/ab_pgs.txt; tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log | cut -c1-300; cut -c1-200 $O/bench.json
