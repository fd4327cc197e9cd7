#!/bin/bash
# A/B timing of kernel build variants on the GPU box.  usage: tools/ab_variants.sh name1:"-DX=1" name2:"-DY" ...
# (Both translation units get the same flags here, max-ILP scheduling included: the in-tree build compiles ss_rollout3.hip without it.)
# Builds each variant into var/ (git-ignored, travels with the gpurun snapshot), then one gpurun call runs the
# configs[1] bench on every variant (and the in-tree build) twice, interleaved.
cd "$(dirname "$0")/.." || exit 1
mkdir -p var
names=()
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -fno-signed-zeros -ffp-contract=on -mllvm -amdgpu-sched-strategy=max-ilp $flags \
    -Rpass-analysis=kernel-resource-usage steppingstone_amd/csrc/ss_api.hip steppingstone_amd/csrc/ss_rollout3.hip -o var/libss_$name.so 2> var/$name.res || { echo "build $name failed"; exit 1; }
  printf "%-12s " "$name"; grep -A12 "step_kernelINS_13ModelWalker3DELb1" var/$name.res | grep -E "VGPRs:|AGPRs|ScratchSize" | sed 's/.*remark: *//; s/ \[-R.*//' | tr '\n' ' '; echo
  names+=("$name")
done
cmd='for r in 1 2; do printf "%-12s " base; python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | grep -oE "\"ms_per_step\": [0-9.]+";'
for n in "${names[@]}"; do
  cmd+=" printf \"%-12s \" $n; STEPPINGSTONE_LIB=\$PWD/var/libss_$n.so python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>&1 | tail -1 | grep -oE \"\\\"ms_per_step\\\": [0-9.]+\";"
done
cmd+=' done'
/usr/local/graft/bin/gpurun --timeout 900 -- "$cmd" 2>&1 | grep -v "^\[gpurun\] sending"
