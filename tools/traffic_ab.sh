#!/bin/bash
# Run ON the GPU box: HBM traffic per launch (PMC FETCH_SIZE / WRITE_SIZE, separate passes, calibrated) of the step kernel and the
# 250-step rollout kernel at 4096 envs for library variants:  tools/traffic_ab.sh <outdir> name=path[,ENV=VALUE] ...
#   e.g. tools/traffic_ab.sh gpurun_out/r03d prev=var/libss_prev.so new= new_plain=,SS_HELPERS=0
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  name="${spec%%=*}"; rest="${spec#*=}"; lib="${rest%%,*}"; envs=""
  [[ "$rest" == *,* ]] && envs="${rest#*,}"
  rm -rf /tmp/hb_$name
  for C in FETCH_SIZE WRITE_SIZE; do
    ( [ -n "$lib" ] && export STEPPINGSTONE_LIB=$R/$lib; [ -n "$envs" ] && export ${envs//,/ }; \
      timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/hb_$name/$C -- python $R/tools/hbm_traffic.py 4096 > /dev/null 2>&1 )
  done
  python $R/tools/hbm_traffic_report.py /tmp/hb_$name 4096 step > $out/traffic_${name}_step.json
  python $R/tools/hbm_traffic_report.py /tmp/hb_$name 4096 rollout > $out/traffic_${name}_rollout.json
  python - $out/traffic_${name}_step.json $out/traffic_${name}_rollout.json $name <<'PY'
import json, sys
s, r = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
print("%-12s step kernel: fetch %.2f MB write %.2f MB total %.2f MB | 250-step rollout launch: fetch %.2f write %.2f total %.2f MB" % (
    sys.argv[3], s["FETCH_SIZE"]["step_bytes"] / 1e6, s["WRITE_SIZE"]["step_bytes"] / 1e6, s["hbm_bytes_per_launch"] / 1e6,
    r["FETCH_SIZE"]["step_bytes"] / 1e6, r["WRITE_SIZE"]["step_bytes"] / 1e6, r["hbm_bytes_per_launch"] / 1e6))
PY
done
