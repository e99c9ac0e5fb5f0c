#!/usr/bin/env python3
"""BASELINE configs[2] context: how fast can PyTorch-ROCm run the reference-shaped Brain + DQN stand-in
(mortal_amd/policy.py: 192 channels x 40 blocks, obs v4) on one MI355X, and with which cheap settings?

Not a kernel task: the net stays PyTorch's (north_star).  Each variant runs the forward of a 16,384-row batch (the bench's chunk
size) several times on random data and reports rows/s; the best one is what `bench.py --policy brain` / `workloads.brain_v4` can use.

  python tools/brain_tune.py [--rows 16384] [--reps 3] > gpurun_out/brain_tune.jsonl
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch

    from mortal_amd.policy import PolicyNet

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = PolicyNet(version=4).to(dev).eval()
    obs = torch.rand(args.rows, 1012, 34, device=dev)
    mask = torch.rand(args.rows, 46, device=dev) < 0.3
    mask[:, 45] = True

    def timed(fn, label, rows=args.rows):
        try:
            with torch.inference_mode():
                fn()  # warm-up (MIOpen find, compile)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(args.reps):
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
            print(json.dumps({"variant": label, "rows": rows, "ms": best * 1e3, "rows_per_s": rows / best}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"variant": label, "error": repr(e)[:300]}), flush=True)

    def fwd(n_chunk, dtype=torch.float16, model=None, amp=True):
        m = net if model is None else model

        def run():
            for i in range(0, args.rows, n_chunk):
                with torch.autocast("cuda", dtype=dtype, enabled=amp):
                    q = m(obs[i:i + n_chunk], mask[i:i + n_chunk])
                q.argmax(-1)
        return run

    timed(fwd(16384), "autocast fp16, chunk 16384 (bench default)")
    if os.environ.get("BRAIN_TUNE_QUICK"):
        torch.backends.cudnn.benchmark = True
    for c in (() if os.environ.get("BRAIN_TUNE_QUICK") else (2048, 4096, 8192)):
        timed(fwd(c), f"autocast fp16, chunk {c}")
    if not os.environ.get("BRAIN_TUNE_QUICK"):
        timed(fwd(16384, torch.bfloat16), "autocast bf16, chunk 16384")
        torch.backends.cudnn.benchmark = True
        timed(fwd(16384), "autocast fp16, chunk 16384, cudnn.benchmark (MIOpen find)")
        timed(fwd(8192), "autocast fp16, chunk 8192, cudnn.benchmark (MIOpen find)")
    # whole model in half precision (no autocast casts per layer)
    try:
        net_h = PolicyNet(version=4).to(dev).eval().half()
        obs_h = obs.half()

        def run_h():
            for i in range(0, args.rows, 8192):
                q = net_h(obs_h[i:i + 8192], mask[i:i + 8192])
                q.argmax(-1)
        timed(run_h, "model.half(), input cast once, chunk 8192, cudnn.benchmark")
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": "model.half()", "error": repr(e)[:300]}), flush=True)
    try:
        comp = torch.compile(net)
        timed(fwd(8192, model=comp), "torch.compile (inductor), autocast fp16, chunk 8192")
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": "torch.compile", "error": repr(e)[:300]}), flush=True)
    try:
        comp_h = torch.compile(net_h)

        def run_ch():
            for i in range(0, args.rows, 8192):
                q = comp_h(obs_h[i:i + 8192], mask[i:i + 8192])
                q.argmax(-1)
        timed(run_ch, "torch.compile (inductor) of model.half(), chunk 8192")
    except Exception as e:  # noqa: BLE001
        print(json.dumps({"variant": "torch.compile half", "error": repr(e)[:300]}), flush=True)


if __name__ == "__main__":
    main()
