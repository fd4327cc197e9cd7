"""usage: hbm_traffic_report.py <dir with *counter_collection.csv from the FETCH_SIZE and WRITE_SIZE passes> N [step|rollout [S]]
Prints calibrated HBM bytes per launch of the single-step kernel (default) or of the multi-step rollout kernel
(tools/hbm_traffic.py launches it with S control steps per launch, default 250).  FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3); both
are calibrated on the dword-per-lane copy kernel of known size (MI355X_MICROARCH.md 'HBM': other access widths than
16 B/lane are uncalibrated -> calibrate on your own access pattern)."""
import csv, glob, sys, collections, json
d, n_envs = sys.argv[1], int(sys.argv[2])
which = sys.argv[3] if len(sys.argv) > 3 else "step"
steps_per_launch = (int(sys.argv[4]) if len(sys.argv) > 4 else 250) if which == "rollout" else 1
pat = "rollout_kernel" if which == "rollout" else "step_kernel"
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "calib" if "calib_copy" in r["Kernel_Name"] else ("step" if pat in r["Kernel_Name"] else None)
        if k:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
calib_bytes = 256 * 1024 * 1024 * 2.0
out = {"kernel": pat, "steps_per_launch": steps_per_launch, "envs": n_envs}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = vals["calib"][c]
    st = vals["step"][c]
    if not cal or not st:
        continue
    kb_cal = sum(cal) / len(cal)
    factor = calib_bytes / (kb_cal * 1024.0)            # true bytes per reported byte
    st = st[-10:] if which == "step" else st[1:] or st   # skip the first launch (cold L2 / code fetch of a fresh process)
    kb = sum(st) / len(st)
    out[c] = {"calib_reported_KB": kb_cal, "correction": factor, "step_reported_KB": kb, "step_bytes": kb * 1024.0 * factor}
tot = sum(v["step_bytes"] for k, v in out.items() if isinstance(v, dict))
out["hbm_bytes_per_launch"] = tot
out["hbm_bytes_per_env_step"] = tot / n_envs / steps_per_launch
print(json.dumps(out, indent=1))
