"""Per-env `N.monitor.csv` files, the on-disk side of the reference's `Monitor` wrapper (common/envs_utils.py:71-194): `make_env_fns`
wraps env `rank` in `Monitor(env, os.path.join(log_dir, str(rank)), allow_early_resets=True)` (:36-38), whose `ResultsWriter` creates
`<log_dir>/<rank>.monitor.csv` with one JSON header line (`# {"t_start": ..., "env_id": ...} \\n`), a csv header `r,l,t` and one row
per finished episode, flushed as it is written.  Byte-identical to the reference's writer on the golden episode sequence
(tests/golden/monitor_golden.json, tools/make_golden_monitor.py).

The reference holds one open file per worker process; with thousands of envs in one process that would exceed the descriptor limit,
so at most `max_open` files stay open (least recently written closed first; rows are flushed either way)."""
import collections
import csv
import json
import os
import time


class MonitorFiles:
    EXT = "monitor.csv"

    def __init__(self, log_dir, num_envs, env_id, first_rank=0, max_open=256, t_start=None):
        self.log_dir = log_dir
        self.first_rank = int(first_rank)
        self.max_open = int(max_open)
        self._open = collections.OrderedDict()      # env index -> (file, csv writer)
        os.makedirs(log_dir, exist_ok=True)
        header = "# {} \n".format(json.dumps({"t_start": time.time() if t_start is None else t_start, "env_id": env_id}))
        for i in range(int(num_envs)):
            with open(self.path(i), "wt", newline="") as f:      # csv supplies its own \r\n (ResultsWriter opens in text mode too)
                f.write(header)
                csv.DictWriter(f, fieldnames=("r", "l", "t")).writeheader()

    def path(self, i):
        return os.path.join(self.log_dir, "%d.%s" % (self.first_rank + i, self.EXT))

    def _writer(self, i):
        if i in self._open:
            self._open.move_to_end(i)
            return self._open[i]
        if len(self._open) >= self.max_open:
            _, (f, _w) = self._open.popitem(last=False)
            f.close()
        f = open(self.path(i), "at", newline="")
        self._open[i] = (f, csv.DictWriter(f, fieldnames=("r", "l", "t")))
        return self._open[i]

    def write_row(self, i, epinfo):
        """epinfo: the dict Monitor.update builds, {"r": round(sum, 6), "l": length, "t": round(seconds since start, 6)}."""
        f, w = self._writer(i)
        w.writerow({k: epinfo[k] for k in ("r", "l", "t")})
        f.flush()

    def close(self):
        for f, _w in self._open.values():
            f.close()
        self._open.clear()
