"""Evaluate a policy trained by steppingstone_amd.train (state_dict) with deterministic actions.
   python tools/eval_policy.py policy.pt --backend hip|oracle [--env Walker3DStepperEnv-v0] [--envs 256] [--steps 600]
--backend oracle runs the CPU oracle behind the same VecEnv class (tests/oracle_backend.py): a policy learned on the GPU
environment has to behave the same way in the independent CPU implementation of the specification."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from steppingstone_amd import ppo
from steppingstone_amd.envs import SteppingStoneVecEnv, kind_of
ap = argparse.ArgumentParser()
ap.add_argument("policy"); ap.add_argument("--backend", default="hip"); ap.add_argument("--env", default="Walker3DStepperEnv-v0")
ap.add_argument("--envs", type=int, default=256); ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--curriculum", type=int, default=5); ap.add_argument("--seed", type=int, default=77)
a = ap.parse_args()
if a.backend == "oracle":
    from oracle_backend import OracleBackend
    dev = torch.device("cpu")
    env = SteppingStoneVecEnv(a.env, a.envs, seed=a.seed, return_numpy=False, backend=OracleBackend(kind_of(a.env), a.envs, a.seed))
else:
    dev = torch.device("cuda:0")
    env = SteppingStoneVecEnv(a.env, a.envs, seed=a.seed, device=dev, return_numpy=False)
ac = ppo.ActorCritic().to(dev)
ac.load_state_dict(torch.load(a.policy, map_location=dev))
env.update_curriculum(a.curriculum)
obs = env.reset()
rets, lens, reached = [], [], []
for t in range(a.steps):
    with torch.no_grad():
        _, act, _ = ac.act(obs, deterministic=True)
    obs, rew, done, info = env.step(act)
    if bool(done.any()):
        rets += info["ep_ret"][done].tolist(); lens += info["ep_len"][done].tolist(); reached += info["steps_reached"][done].tolist()
print("%s backend, %s, curriculum %d, %d envs x %d steps: %d episodes, mean return %.1f, mean length %.1f, mean stones reached %.2f, median stones %.0f"
      % (a.backend, a.env, a.curriculum, a.envs, a.steps, len(rets), np.mean(rets), np.mean(lens), np.mean(reached), np.median(reached)))
