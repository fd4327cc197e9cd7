"""python -m steppingstone_amd.train --env MikeStepperEnv-v0 --num-envs 4096 --num-steps 32 --updates 50

PPO end-to-end on the GPU env (BASELINE.json configs[4]): counterpart of `python -m playground.train with ...`
(scripts/local_run_playground_train.sh:23).  Under torchrun every rank owns --num-envs environments and an identical
policy replica; gradients are all-reduced per minibatch (RCCL), env state never leaves its GPU.
The reference derives num_steps = episode_steps // num_processes (playground/train.py:59-61), which is 1 at 4096 envs;
an explicit --num-steps (default 32) is used instead."""
import argparse
import json
import os

import torch
import torch.distributed as dist

from . import ppo
from .envs import SteppingStoneVecEnv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="MikeStepperEnv-v0")
    ap.add_argument("--num-envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--num-steps", type=int, default=32)
    ap.add_argument("--updates", type=int, default=20)
    ap.add_argument("--seed", type=int, default=8)                  # playground/train.py:41
    ap.add_argument("--num-ensembles", type=int, default=1)
    ap.add_argument("--ppo-epoch", type=int, default=10)
    ap.add_argument("--mini-batch-size", type=int, default=1024)     # playground/train.py:62
    ap.add_argument("--no-curriculum", action="store_true")
    ap.add_argument("--mirror", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="run rollout and minibatch steps eagerly (no hipGraph)")
    ap.add_argument("--save", default="")
    args = ap.parse_args()

    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")     # see steppingstone_amd/distributed.py (RCCL stream vs launch stream)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    envs = SteppingStoneVecEnv(args.env, args.num_envs, seed=args.seed, device=dev, env_id_offset=rank * args.num_envs,
                               return_numpy=False)

    def log(stats):
        if rank == 0:
            stats = dict(stats, total_num_steps=stats["total_num_steps"] * world, fps=stats["fps"] * world)
            print(json.dumps(stats), flush=True)

    ac, hist = ppo.train(envs, args.updates, num_steps=args.num_steps, num_ensembles=args.num_ensembles, seed=args.seed,
                         use_curriculum=not args.no_curriculum, use_mirror=args.mirror, ppo_epoch=args.ppo_epoch,
                         mini_batch_size=args.mini_batch_size, log=log,
                         use_graph=False if args.no_graph else "auto")
    if args.save and rank == 0:
        torch.save(ac.state_dict(), args.save)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
