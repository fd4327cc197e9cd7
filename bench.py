#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched random-action rollout (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: ONE step_kernel launch advancing 4096 Walker3DStepperEnv-v0
environments per GPU by one control step (4 physics substeps, contact solve, reward, auto-reset, 60-float
observation), actions drawn on the device (Philox, U(-1,1)).  Workload = BASELINE.json configs[1]
("Walker3DStepperEnv-v0, 4096 envs on 1 MI355X, flat terrain (curriculum off), random actions").  With N>1 GPUs
each rank owns 4096 envs (weak scaling) and every step ends with the RCCL all-gather of the packed
[4096,62] obs|rew|done block (BASELINE configs[3]).  State is resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task brief) including `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
ENV_ID = "Walker3DStepperEnv-v0"
# ALGORITHMIC HBM bytes per env-step of the rollout kernel with this repository's state layout (DESIGN.md
# "bytes"): read 83 f32 state/stone-cache fields + 4 i32 = 348 B; write 59 f32 + 5 i32 state = 256 B,
# obs 240 B, rew 4 B, done 1 B, info 20 B = 521 B.  (Actions are generated on the device: 0 B.)
ALGO_BYTES_PER_ENV_STEP = 348 + 521
HBM_PEAK_GBS = 8000.0
VALU_FP32_PEAK_TFLOPS = 157.3          # packed-f32 vector peak (MI355X_MICROARCH.md): 256 CUs x 2.4 GHz x 256 flop/clk


def recorded_traffic(n_envs):
    """HBM bytes per step_kernel launch from the latest committed PMC run (separate FETCH_SIZE / WRITE_SIZE passes,
    calibrated on a dword-per-lane copy: tools/hbm_traffic.py + tools/hbm_traffic_report.py).  PMC collection needs
    rocprofv3 around the process, so bench.py reports the recorded figure and names its source; null if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*hbm_traffic_%d.json" % n_envs)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return float(json.load(f)["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)


def recorded_flops_per_env_step():
    """fp32 VALU flops per env-step from the committed SQ-counter profile (SQ_INSTS_VALU_FLOPS_FP32 counts flops per
    wavefront-instruction lane); null if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_4096.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    v = d["per_wave_per_launch"].get("SQ_INSTS_VALU_FLOPS_FP32")
    if not v:
        return None, None
    # per-wavefront mean x wavefronts per launch (main + helper wavefronts) x 64 lanes, per env of the launch
    return float(v) * float(d["waves_per_launch"]) * 64.0 / float(d["envs"]), os.path.relpath(files[-1], ROOT)


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (a port of docs/PHYSICS.md, NOT PyBullet) on the host cores, same workload, bounded."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as ol
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    env = ol.OracleEnv("walker3d", ENVS_PER_GPU, seed=0)
    env.reset()
    acts = [env.random_actions(t) for t in range(4)]
    env.step(acts[0])                       # warm-up
    t0 = time.perf_counter()
    steps = 0
    while True:
        env.step(acts[steps % 4])
        steps += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or steps >= 400:
            break
    return {"value": ENVS_PER_GPU * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d control steps of %d Walker3D envs, oracle/ss_oracle.c fp32, OpenMP over %d host threads, "
                      "%.1f s" % (steps, ENVS_PER_GPU, cores, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--env", default=ENV_ID)
    ap.add_argument("--curriculum", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the per-step all-gather")
    args = ap.parse_args()

    # more hardware queues than HIP's default 4, so that RCCL's stream never shares one with the launch stream
    # (steppingstone_amd/distributed.py); must be set before the HIP runtime starts
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist
    from steppingstone_amd.distributed import ShardedVecEnv
    from steppingstone_amd.envs import SteppingStoneVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force = os.environ.get("SS_FORCE_COLLECTIVE") == "1"      # world 1 under torchrun: still go through RCCL
    use_dist = world > 1 or (force and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    n_local = args.envs_per_gpu
    local = SteppingStoneVecEnv(args.env, n_local, seed=0, device=dev, env_id_offset=rank * n_local, return_numpy=False)
    if args.curriculum:
        local.update_curriculum(args.curriculum)
    env = ShardedVecEnv(local)
    env.reset()
    gather = use_dist and not args.no_gather

    def sync():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(dev)

    env.rollout_random(args.warmup, t0=0, gather=gather)
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # kernel-only time: events around each launch would perturb the stream; instead time a second, gather-free
    # pass of the same K launches with events on the launch stream (back-to-back launches => sum of durations).
    t_start = time.perf_counter()
    env.rollout_random(args.steps, t0=args.warmup, gather=gather)
    sync()
    elapsed = time.perf_counter() - t_start
    ev0.record()
    local.rollout_random(args.steps, t0=args.warmup + args.steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        total_envs = n_local * world
        value = total_envs * args.steps / elapsed
        achieved = ALGO_BYTES_PER_ENV_STEP * n_local / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = recorded_traffic(n_local)
        out = {
            "metric": "env-steps/sec (batched random-action rollout)",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d envs per MI355X, curriculum %d (flat terrain), on-device Philox U(-1,1) "
                                   "actions, auto-reset on" % (args.env, n_local, args.curriculum),
                       "envs_total": total_envs, "parallelism": "env-shard x%d%s" % (world, "+allgather" if gather else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "B/launch",
                         "traffic_source": traffic_src,
                         "kernel": "ss::step_kernel_helped<ModelWalker3D,true,3> (ss::step_kernel above 16384 envs)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_ENV_STEP * n_local,
                         "note": "VALU-issue-bound per-lane rigid-body dynamics (80 % VALU-busy, one wavefront per SIMD), not HBM-bound (DESIGN.md 4.1)"},
        }
        flop, flop_src = recorded_flops_per_env_step()
        if flop:
            tf = flop * n_local / (kernel_ms * 1e-3) / 1e12
            out["roofline"]["valu_fp32"] = {"flop_per_env_step": flop, "source": flop_src, "achieved": tf, "peak": VALU_FP32_PEAK_TFLOPS,
                                            "unit": "TFLOP/s", "frac": tf / VALU_FP32_PEAK_TFLOPS,
                                            "note": "secondary roofline: packed-f32 vector peak of the chip; 4096 envs occupy 128 of its 1024 SIMDs"}
        if world == 1 and not use_dist and n_local < 32768:
            # not the metric: the same kernel with every SIMD of the chip occupied (4 wavefronts per CU)
            big = SteppingStoneVecEnv(args.env, 32768, seed=0, device=dev, return_numpy=False)
            big.reset()
            big.rollout_random(50, 0)
            torch.cuda.synchronize(dev)
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record(); big.rollout_random(200, 50); b1.record()
            torch.cuda.synchronize(dev)
            bms = b0.elapsed_time(b1) / 200
            out["capacity"] = {"envs_per_gpu": 32768, "ms_per_step": bms, "value": 32768 / (bms * 1e-3), "unit": "env-steps/s",
                               "roofline_frac": ALGO_BYTES_PER_ENV_STEP * 32768 / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                               "valu_fp32_frac": (flop * 32768 / (bms * 1e-3) / 1e12 / VALU_FP32_PEAK_TFLOPS) if flop else None,
                               "note": "same kernel at 32768 envs on this GPU (all 1024 SIMDs occupied); not the BASELINE config"}
            big.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
