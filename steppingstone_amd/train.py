"""python -m steppingstone_amd.train --env MikeStepperEnv-v0 --num-envs 4096 --num-steps 32 --updates 50 [--gpus N]

PPO end-to-end on the GPU env (BASELINE.json configs[4]): counterpart of `python -m playground.train with ...`
(scripts/local_run_playground_train.sh:23; switches of playground/train.py:44-70).  One process per GPU: with --gpus N > 1
and no launcher the script starts its own N ranks (steppingstone_amd/launch.py), the way the reference forks its env
workers itself; every rank owns --num-envs environments and an identical policy replica, gradients and advantage
statistics are all-reduced per minibatch (RCCL), env state never leaves its GPU.
The reference derives num_steps = episode_steps // num_processes (playground/train.py:59-61), which is 1 at 4096 envs;
an explicit --num-steps (default 32) is used instead."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

from . import launch, ppo
from .csv_logger import ConsoleCSVLogger
from .envs import SteppingStoneVecEnv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="MikeStepperEnv-v0")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--num-envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--num-steps", type=int, default=32)
    ap.add_argument("--updates", type=int, default=20)
    ap.add_argument("--seed", type=int, default=8)                  # playground/train.py:41
    ap.add_argument("--num-ensembles", type=int, default=1)
    ap.add_argument("--ppo-epoch", type=int, default=10)
    ap.add_argument("--mini-batch-size", type=int, default=1024)     # playground/train.py:62
    ap.add_argument("--no-curriculum", action="store_true")         # use_curriculum
    ap.add_argument("--specialist", action="store_true")            # use_specialist
    ap.add_argument("--adaptive", action="store_true")              # use_adaptive_sampling (train.py:48,134-137,320-361)
    ap.add_argument("--threshold", action="store_true")             # use_threshold_sampling (train.py:50,123-133,229-272)
    ap.add_argument("--curriculum-threshold", type=float, default=0.85)   # train.py:68
    ap.add_argument("--num-eval-envs", type=int, default=16, help="batch of the sampler's evaluation env (reference: 1)")
    ap.add_argument("--num-tests", type=int, default=4)             # train.py:64
    ap.add_argument("--test-interval", type=int, default=10, help="deterministic test episodes every K updates (reference: 1; 0 = off)")
    ap.add_argument("--mirror", action="store_true")                # use_mirror
    ap.add_argument("--no-graph", action="store_true", help="run rollout and minibatch steps eagerly (no hipGraph)")
    ap.add_argument("--log-dir", default="", help="progress.csv with the reference's columns (common/csv_utils.py)")
    ap.add_argument("--save-dir", default="", help="{env}_latest.pt / _best.pt / _{frames}.pt (train.py:523-562)")
    ap.add_argument("--save-every", type=float, default=1e7)        # train.py:43
    ap.add_argument("--save", default="", help="also write the final state_dict here")
    args = ap.parse_args()
    assert not (args.adaptive and args.threshold), "choose one of --adaptive / --threshold"

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # see steppingstone_amd/distributed.py (RCCL stream vs launch stream)
    rc = launch.ensure_ranks(args.gpus, sys.argv, module="steppingstone_amd.train")
    if rc is not None:
        raise SystemExit(rc)
    rank, local_rank, world = launch.rank_info()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    envs = SteppingStoneVecEnv(args.env, args.num_envs, seed=args.seed, device=dev, env_id_offset=rank * args.num_envs,
                               return_numpy=False)
    sampling = "adaptive" if args.adaptive else ("threshold" if args.threshold else "none")
    # evaluation / test envs live beyond the training envs' global id range (their own RNG streams)
    far = world * args.num_envs
    eval_envs = (SteppingStoneVecEnv(args.env, args.num_eval_envs, seed=args.seed, device=dev, env_id_offset=far,
                                     return_numpy=False) if sampling != "none" else None)
    test_envs = (SteppingStoneVecEnv(args.env, args.num_tests, seed=args.seed, device=dev, env_id_offset=far + 65536,
                                     return_numpy=False) if args.test_interval and rank == 0 else None)
    logger = ConsoleCSVLogger(log_dir=args.log_dir) if args.log_dir and rank == 0 else None

    def log(stats):
        if rank == 0:
            print(json.dumps(stats), flush=True)

    env_name = args.env.split(":")[-1]
    ac, hist = ppo.train(envs, args.updates, num_steps=args.num_steps, num_ensembles=args.num_ensembles, seed=args.seed,
                         use_curriculum=not (args.no_curriculum or args.specialist or sampling != "none"),
                         use_specialist=args.specialist, use_mirror=args.mirror, ppo_epoch=args.ppo_epoch,
                         mini_batch_size=args.mini_batch_size, log=None if logger else log,
                         use_graph=False if args.no_graph else "auto", sampling=sampling, eval_envs=eval_envs,
                         curriculum_threshold=args.curriculum_threshold, test_envs=test_envs,
                         test_interval=args.test_interval, logger=logger, save_dir=args.save_dir,
                         save_every=args.save_every, env_name=env_name)
    if args.save and rank == 0:
        torch.save(ac.state_dict(), args.save)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
