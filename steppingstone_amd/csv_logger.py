"""progress.csv writer with the columns of the reference's logger (common/csv_utils.py:16-68 fed by
playground/train.py:564-578), so that its plotting scripts (playground/plot_from_csv.py --columns mean_rew
test_mean_rew ...) read this framework's runs unchanged:

    iter,total_num_steps,fps,entropy,value_loss,action_loss,
    mean_rew,median_rew,min_rew,max_rew,test_mean_rew,test_median_rew,test_min_rew,test_max_rew
"""
import csv
import os

import numpy as np


class CSVLogger:
    def __init__(self, log_dir, filename="progress.csv"):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, filename)
        self.csvfile = open(self.path, "w", newline="")
        self.writer = None

    def log_epoch(self, data):
        row = {k: v for k, v in data.items() if k not in ("stats", "test_stats")}
        for prefix, key in (("", "stats"), ("test_", "test_stats")):
            for name, values in data.get(key, {}).items():
                v = np.asarray(values, dtype=np.float64)
                row[prefix + "mean_" + name] = np.mean(v)
                row[prefix + "median_" + name] = np.median(v)
                row[prefix + "min_" + name] = np.min(v)
                row[prefix + "max_" + name] = np.max(v)
        if self.writer is None:
            self.writer = csv.DictWriter(self.csvfile, fieldnames=list(row.keys()))
            self.writer.writeheader()
        self.writer.writerow(row)
        self.csvfile.flush()
        return row

    def close(self):
        if not self.csvfile.closed:
            self.csvfile.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ConsoleCSVLogger(CSVLogger):
    def __init__(self, console_log_interval=1, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.console_log_interval = console_log_interval

    def log_epoch(self, data):
        row = super().log_epoch(data)
        if row["iter"] % self.console_log_interval == 0:
            print("Updates {}, num timesteps {}, FPS {}, mean/median reward {:.1f}/{:.1f}, min/max reward {:.1f}/{:.1f}, "
                  "test_mean/median reward {:.1f}/{:.1f}, test_min/max reward {:.1f}/{:.1f}, entropy {:.5f}, value loss "
                  "{:.5f}, policy loss {:.5f}".format(
                      row["iter"], row["total_num_steps"], row["fps"], row["mean_rew"], row["median_rew"], row["min_rew"],
                      row["max_rew"], row["test_mean_rew"], row["test_median_rew"], row["test_min_rew"], row["test_max_rew"],
                      row["entropy"], row["value_loss"], row["action_loss"]), flush=True)
        return row
