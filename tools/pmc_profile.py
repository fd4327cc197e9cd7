"""SQ counter profile of the step kernel (run ON the GPU box):  python tools/pmc_profile.py [N] [out.json]
Runs tools/hbm_traffic.py (20 rollout steps at N envs) under rocprofv3 once per counter group (8 SQ slots per pass;
--pmc only with --kernel-trace, as the pool requires) and prints per-wave, per-launch means for the step kernel.
Units: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)."""
import collections, csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else "4096"
OUT = sys.argv[2] if len(sys.argv) > 2 else None
WHICH = sys.argv[3] if len(sys.argv) > 3 else "rollout"      # rollout: ss::rollout_kernel* (250 steps per launch); step: ss::step_kernel*
PAT = "rollout_kernel" if WHICH == "rollout" else "step_kernel"
STEPS_PER_LAUNCH = 250 if WHICH == "rollout" else 1
GROUPS = [
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY",
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH",
    "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_IFETCH",
    "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_IFETCH_LEVEL SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES",
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU_FLOPS_FP32",
]
env = dict(os.environ, TMPDIR="/tmp")
res = {}
for g in GROUPS:
    d = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + g.split() + ["--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(ROOT, "tools", "hbm_traffic.py"), N]
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if PAT in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
waves = res.get("SQ_WAVES", 1.0) or 1.0
per_wave = {k: round(v / waves, 1) for k, v in sorted(res.items())}
summary = {"envs": int(N), "kernel": PAT, "steps_per_launch": STEPS_PER_LAUNCH, "waves_per_launch": waves, "per_wave_per_launch": per_wave}
wc = per_wave.get("SQ_WAVE_CYCLES")
if wc:
    summary["fractions_of_wave_cycles"] = {k: round(per_wave[k] / wc, 3) for k in
        ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA") if k in per_wave}
print(json.dumps(summary, indent=1))
if OUT:
    json.dump(summary, open(OUT, "w"), indent=1)
