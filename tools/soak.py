"""Soak: long random-action rollouts at full size, both robots, all curricula; every output finite, rewards bounded,
episodes end (no env stuck beyond the step limit), determinism across two runs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from steppingstone_amd.envs import SteppingStoneVecEnv, MAX_EPISODE_STEPS
n, K = 8192, 20000
for env_id in ("Walker3DStepperEnv-v0", "MikeStepperEnv-v0"):
    for cur in (0, 3, 5):
        sums = []
        for rep in range(2):
            env = SteppingStoneVecEnv(env_id, n, seed=123 + cur, device="cuda:0", return_numpy=False)
            env.update_curriculum(cur)
            env.reset()
            bad = 0; rmin, rmax = 1e9, -1e9; ndone = 0; acc = torch.zeros((), device="cuda:0", dtype=torch.float64)
            maxlen = 0
            for t in range(0, K, 50):               # 50 control steps per launch (rollout kernel), checks after every launch
                obs, rew, done = env.rollout_random(50, t0=t, steps_per_launch=50)
                bad += int((~torch.isfinite(obs)).sum() + (~torch.isfinite(rew)).sum())
                rmin = min(rmin, float(rew.min())); rmax = max(rmax, float(rew.max()))
                info = env._info_tensors()
                maxlen = max(maxlen, int(info["ep_len"].max()))
                ndone += int(done.sum())
                acc += obs.double().sum() + rew.double().sum()
                st = env.get_state()
                bad += int((~torch.isfinite(st)).sum()) + int(((st[:, 3:7].norm(dim=1) - 1).abs() > 1e-3).sum())
            sums.append(float(acc))
            env.close()
        print("%s curriculum %d: non-finite %d, reward range [%.2f, %.2f], sampled dones %d, max episode length %d (limit %d), deterministic %s"
              % (env_id, cur, bad, rmin, rmax, ndone, maxlen, MAX_EPISODE_STEPS, sums[0] == sums[1]), flush=True)
