#!/usr/bin/env python3
"""Soak: many solver-side calls (graph replay active) and training steps; device memory must not grow, results must not
drift.  python tools/soak.py [--calls 20000] [--steps 2000]"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=2000)
    args = ap.parse_args()
    import torch
    srcfd = importlib.import_module("sr-for-cfd_amd")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    tr = importlib.import_module("sr-for-cfd_amd.train")
    enc, dec = synth.keras_default_init(0)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 10, 10, 1)).astype(np.float32)
    free0 = torch.cuda.mem_get_info()[0]
    for prec in ("bf16", "fp32"):
        m = srcfd.SRModel.from_weights(enc, dec, device=0)
        m.precision = prec
        ref = m.predict(x).copy()
        out = np.empty_like(ref)
        m.predict(x, out=out)
        base = torch.cuda.mem_get_info()[0]
        t0 = time.perf_counter()
        for i in range(args.calls):
            m.predict(x, out=out)
            if i % 5000 == 0:
                assert np.array_equal(out, ref), (prec, i)
        dt = time.perf_counter() - t0
        assert np.array_equal(out, ref)
        leak = base - torch.cuda.mem_get_info()[0]
        print(f"{prec}: {args.calls} calls, {dt / args.calls * 1e3:.4f} ms/call, device memory change {leak} B")
        assert leak <= 0, leak
        del m
    # random batch sizes and sample subsets: graph capture / drop / re-capture, buffer growth, tail segmentation choices --
    # every row must equal the row of a one-off reference run, bit for bit (batch invariance)
    pool = rng.standard_normal((40, 10, 10, 1)).astype(np.float32)
    for prec in ("bf16", "f16", "fp32"):
        m = srcfd.SRModel.from_weights(enc, dec, device=0)
        m.precision = prec
        ref = m.predict(pool)
        sizes = [1, 1, 3, 3, 3, 2, 7, 7, 7, 40, 3, 3, 3, 16, 16, 16, 5, 130, 3, 3, 3]
        for it in range(args.calls // 40):
            n = sizes[it % len(sizes)] if it < 3 * len(sizes) else int(rng.integers(1, 24))
            idx = rng.integers(0, 40, size=min(n, 40)) if n <= 40 else rng.integers(0, 40, size=n)
            y = m.predict(pool[idx])
            assert np.array_equal(y, ref[idx]), (prec, it, n)
        print(f"{prec}: {args.calls // 40} random-size calls bit-identical to the reference rows")
        del m
    t = tr.Trainer(srcfd.SRModel.from_weights(enc, dec, device=0), max_batch=8)
    xs = torch.from_numpy(rng.standard_normal((8, 10, 10, 1)).astype(np.float32)).cuda()
    ys = torch.from_numpy(rng.standard_normal((8, 400, 400, 1)).astype(np.float32)).cuda()
    l0 = t.step(xs, ys)
    for _ in range(3):  # the step graph is captured on the second call: take the memory baseline after it
        t.step(xs, ys, return_loss=False)
    torch.cuda.synchronize()
    base = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    for i in range(args.steps):
        t.step(xs, ys, return_loss=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    l1 = t.step(xs, ys)
    leak = base - torch.cuda.mem_get_info()[0]
    print(f"training: {args.steps} steps, {dt / args.steps * 1e3:.4f} ms/step, loss {l0:.5f} -> {l1:.5f}, device memory change {leak} B")
    assert np.isfinite(l1) and l1 < l0 and leak <= 0
    print("free at start", free0, "now", torch.cuda.mem_get_info()[0])


if __name__ == "__main__":
    main()
