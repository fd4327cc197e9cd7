#!/bin/bash
# A/B timing of prebuilt library variants on the GPU box: tools/ab_libs.sh var/libss_a.so var/libss_b.so ...
# (each run: bench.py N=1, 2000 steps; prints rollout-kernel and per-step-launch ms/step), two interleaved rounds.
cmd='for r in 1 2; do'
for lib in "$@"; do
  cmd+=" printf \"%-28s \" $lib; STEPPINGSTONE_LIB=\$PWD/$lib python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-extra 2>/dev/null | python -c \"import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('rollout %.4f  per-step-launch %.4f' % (d['ms_per_step'], d['per_step_launch']['ms_per_step']))\";"
done
cmd+=' done'
/usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] sending\|amdgpu.ids"
