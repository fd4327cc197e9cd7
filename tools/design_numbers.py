#!/usr/bin/env python3
"""Prints the numbers DESIGN.md section 6 / README quote, straight from the evidence files of one tag under profiles/ (or gpurun_out/):
   python tools/design_numbers.py r06_v1 [dir]"""
import csv
import json
import os
import sys

tag = sys.argv[1]
d = sys.argv[2] if len(sys.argv) > 2 else "profiles"
P = lambda name: os.path.join(d, "%s_%s" % (tag, name))      # noqa: E731


def last_json(path):
    rows = [l for l in open(path) if l.startswith("{")]
    return json.loads(rows[-1])


b = last_json(P("bench.json"))
print("bench: value %.2f M env-steps/s, ms_per_step %.5f, repeats %d; per_step_launch %.5f ms (%.2f M)" % (
    b["value"] / 1e6, b["ms_per_step"], b["repeats"], b["per_step_launch"]["ms_per_step"], b["per_step_launch"]["value"] / 1e6))
r = b["roofline"]
print("  roofline: achieved %.2f GB/s, frac %.5f, kernel %s, kernel_ms %.4f per %d-step launch, algorithmic %.1f B/env-step; state round trip: %.2f GB/s frac %.5f" % (
    r["achieved"], r["frac"], r["kernel"], r["kernel_ms"], b["config"]["steps_per_launch"], r["algorithmic_bytes_per_env_step"],
    r["state_roundtrip_every_step"]["achieved"], r["state_roundtrip_every_step"]["frac"]))
print("  traffic (PMC model) %s B/launch" % r.get("traffic"))
if "binding_roofline" in b:
    print("  binding roofline: %.2f TFLOP/s = %.4f of %.1f" % (b["binding_roofline"]["achieved"], b["binding_roofline"]["frac"], b["binding_roofline"]["peak"]))
c = b.get("cpu_baseline")
if c:
    print("  cpu_baseline: %s" % json.dumps({k: c[k] for k in c if not isinstance(c[k], (dict, list))})[:400])
for k, v in (b.get("extra") or {}).items():
    print("  extra %s: %s" % (k, json.dumps(v)[:300]))
try:
    ds = last_json(P("bench_driver_shape.json"))
    print("driver shape: value %.2f M, ms_per_step %.5f (min %.5f max %.5f), repeats %d, timed_region_s %.3f, events in %s regions; kernel_ms %.4f; per_step_launch %.5f" % (
        ds["value"] / 1e6, ds["ms_per_step"], ds["ms_per_step_min"], ds["ms_per_step_max"], ds["repeats"], ds["timed_region_s"], ds.get("hip_events_in_regions"),
        ds["roofline"]["kernel_ms"], ds["per_step_launch"]["ms_per_step"]))
    print("  driver shape roofline frac %.5f binding %.4f" % (ds["roofline"]["frac"], ds.get("binding_roofline", {}).get("frac", float("nan"))))
except Exception as exc:
    print("driver shape: %r" % exc)
try:
    pp = last_json(P("bench_ppo.json"))
    for k in ("learner_torch", "learner_torch_scaled_minibatch"):
        v = pp.get(k) or {}
        print("ppo %s: %.0f frames/s, %.0f ms per update, minibatch %s" % (k, v.get("value") or 0, v.get("ms_per_update") or 0, v.get("mini_batch_size")))
except Exception as exc:
    print("ppo: %r" % exc)
try:
    rows = list(csv.DictReader(open(P("rocprofv3_kernel_stats.csv"))))
    for r_ in rows[:3]:
        print("rocprofv3 stats: %s calls %s avg %.1f us" % (r_["Name"][:70], r_["Calls"], float(r_["AverageNs"]) / 1e3))
    rows = list(csv.DictReader(open(P("rocprofv3_rollout_dispatches.csv"))))
    print("rocprofv3 rollout dispatches (ms): %s" % [round(int(r_["duration_ns"]) / 1e6, 3) for r_ in rows])
except Exception as exc:
    print("rocprof: %r" % exc)
for name in ("scaling_envs.txt", "regimes.txt"):
    try:
        print("--- %s\n%s" % (name, open(P(name)).read()[-1500:]))
    except Exception as exc:
        print(name, repr(exc))
for name in ("hbm_traffic_4096.json", "hbm_traffic_rollout_4096.json", "hbm_traffic_rollout25_4096.json", "hbm_traffic_32768.json", "hbm_traffic_rollout_32768.json", "pmc_4096.json", "pmc_32768.json", "pmc_step_4096.json"):
    try:
        j = json.load(open(P(name)))
        print("--- %s: %s" % (name, json.dumps(j)[:600]))
    except Exception as exc:
        print(name, repr(exc))
try:
    print("--- pytest: %s" % [l.strip() for l in open(P("pytest_gpu.log")) if "passed" in l or "failed" in l][-1])
    t = open(P("parity_heldout.txt")).read()
    print("--- heldout: %s" % t[t.index("TOTAL"):][:900])
except Exception as exc:
    print(repr(exc))
