#!/bin/bash
# training sanity on the identified robots: the package's own PPO (torch learner), 4096 envs, fixed-order curriculum on
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for env in Walker3DStepperEnv-v0 MikeStepperEnv-v0; do
  timeout 900 python -m steppingstone_amd.train --env $env --num-envs 4096 --num-steps 32 --updates 300 --mini-batch-size 4096 --mirror --test-interval 0 > gpurun_out/r05_v1_ppo_${env}_300_updates.jsonl 2> gpurun_out/r05_v1_ppo_${env}.err
  python - $env <<'PY'
import json, sys
rows=[json.loads(l) for l in open("gpurun_out/r05_v1_ppo_%s_300_updates.jsonl" % sys.argv[1]) if l.startswith("{")]
for r in rows[::30] + rows[-1:]:
    print(sys.argv[1], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ("update", "total_num_steps", "mean_rew", "curriculum", "fps")})
PY
done
