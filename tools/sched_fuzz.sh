#!/bin/bash
# Builds the -DSS_FUZZ_SCHED variant (var/libss_fuzz.so) and, on the GPU box, prints the fingerprints of the in-tree build once and of
# the fuzzed build three times (its sleeps are seeded by the shader clock: every run is a different schedule); any differing line is a
# result that depended on the relative timing of the wavefronts.   usage: tools/sched_fuzz.sh [env-steps per configuration]
cd "$(dirname "$0")/.." || exit 1
mkdir -p var gpurun_out
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -fno-signed-zeros -ffp-contract=on -DSS_FUZZ_SCHED"
hipcc $F -mllvm -amdgpu-sched-strategy=max-ilp -c steppingstone_amd/csrc/ss_api.hip -o var/fuzz_api.o &
hipcc $F -c steppingstone_amd/csrc/ss_rollout3.hip -o var/fuzz_r3.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC var/fuzz_api.o var/fuzz_r3.o -o var/libss_fuzz.so || exit 1
N=${1:-10000000}
/usr/local/graft/bin/gpurun --timeout 2400 -- "python tools/sched_fuzz.py $N > gpurun_out/fuzz_plain.txt 2>&1; for r in 1 2 3; do STEPPINGSTONE_LIB=\$PWD/var/libss_fuzz.so python tools/sched_fuzz.py $N > gpurun_out/fuzz_run\$r.txt 2>&1; done; cd gpurun_out; for r in 1 2 3; do echo \"== fuzzed run \$r vs plain\"; diff <(grep -v '^#' fuzz_plain.txt) <(grep -v '^#' fuzz_run\$r.txt) && echo identical; done; tail -2 fuzz_plain.txt fuzz_run1.txt"
