#!/bin/bash
# Run ON the GPU box (via gpurun): collects everything profiles/ holds for one kernel version.
#   usage: tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*
# rocprofv3 passes: --kernel-trace --stats on its own; each --pmc group on its own with --kernel-trace only.
tag=${1:-r01_vX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 2400 python -m pytest tests -m gpu -x -q -rs --durations=12 2>&1; echo "pytest rc $?" ) > $O/${tag}_pytest_gpu.log
( cd $R && python bench.py ) > $O/${tag}_bench.json 2> $O/${tag}_bench.err
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $R/bench.py --no-cpu-baseline > $O/${tag}_bench_under_rocprof.json 2>/dev/null
cp $(ls /tmp/kt/*/*kernel_stats.csv | head -1) $O/${tag}_rocprofv3_kernel_stats.csv
# per-dispatch durations of the multi-step kernel (the --stats average mixes the 200-step warm-up launch with the timed 1000-step ones)
python - "$(ls /tmp/kt/*/*kernel_trace.csv | head -1)" > $O/${tag}_rocprofv3_rollout_dispatches.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rollout_kernel" in r["Kernel_Name"]]
print("kernel,start_ns,duration_ns")
for r in rows:
    print('"%s",%s,%d' % (r["Kernel_Name"], r["Start_Timestamp"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
for N in 4096 32768; do
  rm -rf /tmp/hb
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/hb/$C -- python $R/tools/hbm_traffic.py $N > /dev/null 2>&1
  done
  python $R/tools/hbm_traffic_report.py /tmp/hb $N step > $O/${tag}_hbm_traffic_$N.json
  python $R/tools/hbm_traffic_report.py /tmp/hb $N rollout > $O/${tag}_hbm_traffic_rollout_$N.json
done
# a second launch length of the rollout kernel: bench.py fits bytes(launch of k steps) = const + per_step * k from the two
rm -rf /tmp/hb25
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/hb25/$C -- python $R/tools/hbm_traffic.py 4096 25 > /dev/null 2>&1
done
python $R/tools/hbm_traffic_report.py /tmp/hb25 4096 rollout 25 > $O/${tag}_hbm_traffic_rollout25_4096.json
python $R/tools/pmc_profile.py 4096 $O/${tag}_pmc_4096.json rollout > /dev/null
python $R/tools/pmc_profile.py 4096 $O/${tag}_pmc_step_4096.json step > /dev/null
python $R/tools/pmc_profile.py 32768 $O/${tag}_pmc_32768.json rollout > /dev/null
python $R/tools/scaling_n.py > $O/${tag}_scaling_envs.txt
python $R/tools/contact_cost.py > $O/${tag}_regimes.txt
( cd $R && python bench.py --ppo --no-cpu-baseline ) > $O/${tag}_bench_ppo.json 2>/dev/null
( cd $R && python bench.py --steps 20 --warmup 5 ) > $O/${tag}_bench_driver_shape.json 2>/dev/null
( cd $R && python -c "import __graft_entry__ as g; g.smoke()" ) > $O/${tag}_smoke.log 2>&1
ls -la $O | tail -20
