#!/usr/bin/env python3
"""BASELINE config 4: conv-AE training steps/s, data-parallel (secondary measurement; bench.py is the headline).

    python tools/train_bench.py [--batch 8] [--steps 30] [--warmup 3]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py

A step = forward + backward + flat-gradient all-reduce (RCCL) + Adam on a per-GPU micro-batch of
`--batch` synthetic (10x10x1 -> 400x400x1) pairs (the notebook trains with batch 8,
sr-ae-conv.ipynb:c558).  Weak scaling: global batch = batch x N; `--global-batch 256` = strong scaling
(SURVEY.md 8d config 4).  Prints one JSON line on rank 0."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--global-batch", type=int, default=0, help="strong scaling: split this global batch over the ranks")
    ap.add_argument("--profile", action="store_true", help="per-phase HIP-event timing on rank 0")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    backend = os.environ.get("SRCFD_BENCH_BACKEND", "nccl")    # "gloo": rehearse N ranks on a box with fewer GPUs
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    srcfd = importlib.import_module("sr-for-cfd_amd")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    tr = importlib.import_module("sr-for-cfd_amd.train")
    ds = importlib.import_module("sr-for-cfd_amd.datasets")
    enc, dec = synth.keras_default_init(0)                     # identical replicas: same seed on every rank
    model = srcfd.SRModel.from_weights(enc, dec, device=local)
    batch = args.batch if not args.global_batch else args.global_batch // world   # strong scaling: fixed global batch
    t = tr.Trainer(model, max_batch=batch)
    # the notebook's own dummy recipe (sr-ae-conv.ipynb:c72-91): x_hr ~ N(0,1), x_lr = avg_pool(x_hr, 40); seed 0 (+rank)
    rng = np.random.default_rng(rank)
    y_h = rng.standard_normal((batch, 400, 400, 1)).astype(np.float32)
    x = torch.from_numpy(ds.avg_pool(y_h, 40)).to(dev)
    y = torch.from_numpy(y_h).to(dev)
    args.batch = batch
    gb = batch * world
    losses = []
    for _ in range(args.warmup):
        t.step(x, y, gb)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t.step(x, y, gb, return_loss=False)     # forward + backward (gradients stored, nothing zero-filled) + all-reduce + Adam
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    d = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
    dt = float(d.item())
    phases = None
    if args.profile and rank == 0:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        acc = np.zeros(3)
        for _ in range(5):
            t.grads.zero_()
            ev[0].record(); t.forward_backward(x, y, gb)
            ev[1].record(); tr.allreduce_sum_(t.grads)
            ev[2].record(); t.apply_adam()
            ev[3].record(); torch.cuda.synchronize()
            acc += [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        phases = dict(zip(("forward_backward_ms", "allreduce_ms", "adam_ms"), (acc / 5).round(4).tolist()))
    if rank == 0:
        ms = dt / args.steps * 1e3
        flops = 3 * 2 * 140_024_128 * gb  # fwd + dgrad + wgrad
        print(json.dumps({"metric": "conv-AE training samples/sec (10x10->400x400, f32, Adam)", "value": round(gb / (ms * 1e-3), 2),
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
                          "scaling": "strong" if args.global_batch else "weak", "dtype": "f32", "data": "synthetic", "micro_batch": args.batch, "global_batch": gb,
                          "params": t.n_params, "tflops_model": round(flops / (ms * 1e-3) / 1e12, 2), "phases": phases}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
