#!/bin/bash
# usage: bv.sh name flags...   (rollout3 unit only changes matter for the K=1000 number at 4096 envs, but build both)
name=$1; shift
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-signed-zeros -ffp-contract=on $*"
hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp -c steppingstone_amd/csrc/ss_api.hip -o var/${name}_api.o 2>var/${name}.err &
hipcc $F -c steppingstone_amd/csrc/ss_rollout3.hip -o var/${name}_r3.o 2>>var/${name}.err
wait
hipcc --offload-arch=gfx950 -shared -fPIC var/${name}_api.o var/${name}_r3.o -o var/libss_${name}.so && echo built $name || echo FAIL $name
