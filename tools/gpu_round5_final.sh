#!/bin/bash
# Final GPU call of round 5 (run ON the GPU box via gpurun): everything profiles/r05_v1_* holds, at the final build.
#   GPU suite + smoke + bench + rocprofv3 stats + PMC + HBM traffic + scaling + regimes + --ppo (tools/collect_profiles.sh),
#   then the held-out validation of the frozen parity rule on the final specification.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-r05_v1}
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
rm -rf gpurun_out/heldout_policies
timeout 2700 python tools/parity_heldout.py --json gpurun_out/${tag}_parity_heldout.json > gpurun_out/${tag}_parity_heldout.txt 2> gpurun_out/${tag}_parity_heldout.err
echo "heldout rc $?" >> gpurun_out/${tag}_parity_heldout.txt
tail -4 gpurun_out/${tag}_pytest_gpu.log; tail -2 gpurun_out/${tag}_smoke.log | cut -c1-200; tail -3 gpurun_out/${tag}_parity_heldout.txt | cut -c1-400; cut -c1-250 gpurun_out/${tag}_bench.json
