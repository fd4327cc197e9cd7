"""usage: hbm_traffic_report.py <dir with *counter_collection.csv from the FETCH_SIZE and WRITE_SIZE passes> N
Prints calibrated HBM bytes per step_kernel launch.  FETCH_SIZE / WRITE_SIZE are in KB (rocprofv3); both are
calibrated on the dword-per-lane copy kernel of known size (MI355X_MICROARCH.md 'HBM': other access widths than
16 B/lane are uncalibrated -> calibrate on your own access pattern)."""
import csv, glob, sys, collections, json
d, n_envs = sys.argv[1], int(sys.argv[2])
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "calib" if "calib_copy" in r["Kernel_Name"] else ("step" if "step_kernel" in r["Kernel_Name"] else None)
        if k:
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
calib_bytes = 256 * 1024 * 1024 * 2.0
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = vals["calib"][c]
    st = vals["step"][c]
    if not cal or not st:
        continue
    kb_cal = sum(cal) / len(cal)
    factor = calib_bytes / (kb_cal * 1024.0)            # true bytes per reported byte
    kb = sum(st[-10:]) / len(st[-10:])
    out[c] = {"calib_reported_KB": kb_cal, "correction": factor, "step_reported_KB": kb, "step_bytes": kb * 1024.0 * factor}
tot = sum(v["step_bytes"] for v in out.values())
out["hbm_bytes_per_launch"] = tot
out["hbm_bytes_per_env_step"] = tot / n_envs
print(json.dumps(out, indent=1))
