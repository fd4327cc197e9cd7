"""env-steps/s seen by a reference-style caller: numpy actions in, numpy obs/rew/done + info dicts out (the
drop-in mode of INTEGRATION.md A), i.e. including H->D of actions, D->H of results and Python info assembly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from steppingstone_amd.envs import make_vec_envs
n = 4096
envs = make_vec_envs("Walker3DStepperEnv-v0", 0, n, None)
obs = envs.reset()
rng = np.random.default_rng(0)
acts = rng.uniform(-1, 1, size=(8, n, 21)).astype(np.float32)
for t in range(20):
    envs.step(acts[t % 8])
t0 = time.perf_counter()
K = 300
for t in range(K):
    obs, rew, done, infos = envs.step(acts[t % 8])
el = time.perf_counter() - t0
print("numpy drop-in mode: %.3f ms/step, %.2f M env-steps/s (PCIe + Python inclusive)" % (1e3 * el / K, n * K / el / 1e6))
