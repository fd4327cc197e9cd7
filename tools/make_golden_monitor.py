#!/usr/bin/env python3
"""Generate tests/golden/monitor_golden.json: the per-env `N.monitor.csv` the REFERENCE's own Monitor + ResultsWriter
(common/envs_utils.py:71-194) write for a scripted toy env behind make_env_fns' wrapping (`Monitor(env, os.path.join(log_dir,
str(rank)), allow_early_resets=True)`, :36-38) -- file name, header line, csv header, rows with their line terminators.  The fixture
holds the step rewards that were fed (inputs) and the file text (expected output); the two time-dependent fields (`t_start` of the
header and the `t` column) are recorded as they came out and masked by the test.  Runs in THIS container only (imports /root/reference
under the stub gym of tools/make_golden.py); data only, no reference source.

  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_monitor.py
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402  (install_stub_gym, nothing else runs on import)

gym = mg.install_stub_gym()
sys.path.insert(0, "/root/reference")
from common.envs_utils import Monitor  # noqa: E402


class Spec:
    id = "Toy-v0"


class Scripted:
    """episodes of fixed lengths with fixed float32 step rewards"""
    spec = Spec()

    def __init__(self, rewards):
        self.rewards, self.k, self.i = rewards, 0, 0

    def reset(self):
        self.i = 0
        return np.zeros(60, np.float32)

    def step(self, a):
        r = float(self.rewards[self.k][self.i])
        self.i += 1
        done = self.i == len(self.rewards[self.k])
        if done:
            self.k += 1
        return np.zeros(60, np.float32), r, done, {}

    def close(self):
        pass


rng = np.random.default_rng(3)
episodes = [rng.normal(1.5, 1.0, size=n).astype(np.float32) for n in (7, 1, 33, 1000, 12)]
with tempfile.TemporaryDirectory() as d:
    env = Monitor(Scripted(episodes), os.path.join(d, "5"), allow_early_resets=True)      # rank 5 of make_env_fns
    for ep in episodes:
        env.reset()
        for _ in ep:
            env.step(None)
    names = sorted(os.listdir(d))
    text = open(os.path.join(d, names[0]), newline="").read()
out = os.path.join(ROOT, "tests", "golden", "monitor_golden.json")
json.dump({"file_name": names[0], "env_id": "Toy-v0", "episodes": [[float(x) for x in ep] for ep in episodes], "text": text}, open(out, "w"), indent=1)
print(names, repr(text))
