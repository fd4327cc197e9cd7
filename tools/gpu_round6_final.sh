#!/bin/bash
# Round 6, evidence call (run ON the GPU box via gpurun): everything profiles/<tag>_* holds for one build --
#   GPU suite + smoke + bench (default, driver shape, --ppo) + rocprofv3 stats / per-dispatch CSV + PMC + HBM traffic + scaling + regimes
#   (tools/collect_profiles.sh), the phase profile with barrier waits, then the held-out validation of the frozen parity rule (version 3)
#   on a FRESH sample, and a training-sanity run of the package's own PPO on each robot.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
tag=${1:-r06_v1}
seed=${2:-69001}
bash tools/collect_profiles.sh $tag > gpurun_out/${tag}_collect.log 2>&1
rm -rf gpurun_out/heldout_policies
timeout 2700 python tools/parity_heldout.py --seed-base $seed --json gpurun_out/${tag}_parity_heldout.json > gpurun_out/${tag}_parity_heldout.txt 2> gpurun_out/${tag}_parity_heldout.err
echo "heldout rc $?" >> gpurun_out/${tag}_parity_heldout.txt
if [ "${3:-train}" = "train" ]; then
  for env in Walker3DStepperEnv-v0 MikeStepperEnv-v0; do
    timeout 900 python -m steppingstone_amd.train --env $env --num-envs 4096 --mini-batch-size 4096 --updates 150 --mirror --test-interval 0 > gpurun_out/${tag}_ppo_${env}_150_updates.jsonl 2> gpurun_out/${tag}_ppo_${env}.err
  done
fi
tail -4 gpurun_out/${tag}_pytest_gpu.log; tail -2 gpurun_out/${tag}_smoke.log | cut -c1-200; tail -3 gpurun_out/${tag}_parity_heldout.txt | cut -c1-500; cut -c1-300 gpurun_out/${tag}_bench.json; cut -c1-300 gpurun_out/${tag}_bench_driver_shape.json
for env in Walker3DStepperEnv-v0 MikeStepperEnv-v0; do tail -1 gpurun_out/${tag}_ppo_${env}_150_updates.jsonl 2>/dev/null | cut -c1-300; done
