#!/usr/bin/env python3
"""Generate tests/golden/progress_golden.{json,csv}: a fixed sequence of per-update log records (inputs, JSON) and the
progress.csv the REFERENCE's own ConsoleCSVLogger (common/csv_utils.py:42-68) writes for them (expected output).
Runs in THIS container only (imports /root/reference); the fixture is data, no reference source.

  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_csv.py
"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
from common.csv_utils import ConsoleCSVLogger  # noqa: E402

rng = np.random.default_rng(7)
records = []
for j in range(6):
    records.append({"iter": j + 1, "total_num_steps": (j + 1) * 4096 * 32, "fps": int(300000 + 1000 * j),
                    "entropy": float(rng.normal()), "value_loss": float(abs(rng.normal())), "action_loss": float(rng.normal() * 0.01),
                    "stats": {"rew": [float(x) for x in rng.normal(100 * j, 30, size=5 + j)]},
                    "test_stats": {"rew": [float(x) for x in rng.normal(90 * j, 10, size=4)]}})
with tempfile.TemporaryDirectory() as d:
    console = io.StringIO()
    with contextlib.redirect_stdout(console):
        lg = ConsoleCSVLogger(log_dir=d, console_log_interval=2)
        for r in records:
            lg.log_epoch(json.loads(json.dumps(r)))      # the reference's log_epoch consumes (deletes keys of) its argument
        lg.csvfile.flush()
    text = open(os.path.join(d, "progress.csv")).read()
out = os.path.join(ROOT, "tests", "golden")
json.dump({"records": records, "console": console.getvalue()}, open(os.path.join(out, "progress_golden.json"), "w"), indent=1)
open(os.path.join(out, "progress_golden.csv"), "w").write(text)
print(text)
