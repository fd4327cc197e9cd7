"""Soak: long random-action rollouts at full size, both robots, all curricula; every output finite, rewards bounded,
episodes end (no env stuck beyond the step limit), determinism across two runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv, MAX_EPISODE_STEPS
n, K = 8192, 6000
for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
    for cur in (0, 3, 5):
        sums = []
        for rep in range(2):
            env = SteppingStoneVecEnv(env_id, n, seed=123 + cur, device="cuda:0", return_numpy=False)
            env.update_curriculum(cur)
            env.reset()
            bad = 0; rmin, rmax = 1e9, -1e9; ndone = 0; acc = torch.zeros((), device="cuda:0", dtype=torch.float64)
            maxlen = 0
            for t in range(K):
                obs, rew, done = env.rollout_random(1, t0=t)
                bad += int((~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum()) if t % 50 == 0 else 0
                if t % 50 == 0:
                    rmin = min(rmin, float(rew.min())); rmax = max(rmax, float(rew.max()))
                    info = env._info_tensors()
                    maxlen = max(maxlen, int(info["ep_len"].max()))
                ndone += int(done.sum()) if t % 50 == 0 else 0
                acc += obs.double().sum() + rew.double().sum()
            sums.append(float(acc))
            env.close()
        print("%s curriculum %d: non-finite %d, reward range [%.2f, %.2f], sampled dones %d, max episode length %d (limit %d), deterministic %s"
              % (env_id, cur, bad, rmin, rmax, ndone, maxlen, MAX_EPISODE_STEPS, sums[0] == sums[1]), flush=True)
