#!/usr/bin/env python3
"""BASELINE configs[3] / configs[4] launcher: the evaluation arena (libriichi.arena.OneVsThree.py_vs_py, champion vs three
baselines) with the tables sharded over the GPUs of one node — one process per GPU, started by torchrun:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
      tools/arena_sharded.py --tables 131072 --version 4 [--channels 192 --blocks 40] [--log-dir DIR]

Every rank plays its contiguous range of games (4-aligned: the four seat rotations of a seed stay on one GPU) with its own
replica of the two engines; the only collective is the all-reduce of the challenger's rank histogram inside py_vs_py
(mortal_amd/arena.py).  Prints throughput and the rating statistics of mortal/one_vs_three.py:99-103 on rank 0.
With a single process (no torchrun) it runs the whole arena on one GPU."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables", type=int, default=131072, help="hanchan in total (seed_count = tables / 4)")
    ap.add_argument("--version", type=int, default=4)
    ap.add_argument("--channels", type=int, default=192)
    ap.add_argument("--blocks", type=int, default=40)
    ap.add_argument("--seed-start", type=int, default=10000)
    ap.add_argument("--key", type=lambda s: int(s, 0), default=0xD5DFAA4CEF265CD7)
    ap.add_argument("--epsilon", type=float, default=0.0, help="Boltzmann epsilon of the challenger (engine.py:72-81)")
    ap.add_argument("--log-dir", default=None)
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from libriichi.arena import OneVsThree

    from mortal_amd.policy import DeviceEngine, PolicyNet

    def engine(seed, name, eps):  # random-init networks of the reference's architecture (no checkpoints offline)
        torch.manual_seed(seed)
        net = PolicyNet(version=args.version, conv_channels=args.channels, num_blocks=args.blocks)
        return DeviceEngine(net, args.version, dev, name=name, enable_amp=True, boltzmann_epsilon=eps, top_p=0.9, seed=seed + rank)

    chal, cham = engine(1, "challenger", args.epsilon), engine(2, "champion", 0.0)
    env = OneVsThree(disable_progress_bar=rank != 0, log_dir=args.log_dir)
    t0 = time.perf_counter()
    rankings = np.array(env.py_vs_py(challenger=chal, champion=cham, seed_start=(args.seed_start, args.key),
                                     seed_count=args.tables // 4))
    dt = time.perf_counter() - t0
    if rank == 0:
        n = rankings.sum()
        avg_rank = rankings @ np.arange(1, 5) / n
        pts = np.array([90, 45, 0, -135])
        avg_pt = rankings @ pts / n
        var_pt = rankings @ (pts - avg_pt) ** 2 / n  # rating variance of the evaluation (configs[4])
        print(f"{n} hanchan on {world} GPU(s) in {dt:.1f}s = {n / dt:.1f} hanchan/s")
        print(f"challenger rankings: {rankings.tolist()} (avg rank {avg_rank:.4f}, {avg_pt:.3f}pt, "
              f"pt std error {float(np.sqrt(var_pt / n)):.3f})")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
