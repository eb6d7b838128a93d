#!/usr/bin/env python3
"""Soak of the full-batch launches (races between waves show up as rare mismatches): N repeats of the 768-sample batch at
every precision, each compared bit for bit with the first.  python tools/soak_batch.py [--reps 300]"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    args = ap.parse_args()
    import torch
    srcfd = importlib.import_module("sr-for-cfd_amd")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    enc, dec = synth.keras_default_init(0)
    rng = np.random.default_rng(0)
    for n in (768, 300, 7):
        x = torch.from_numpy(rng.standard_normal((n, 10, 10, 1)).astype(np.float32)).cuda()
        for prec in ("bf16", "f16", "fp32"):
            m = srcfd.SRModel.from_weights(enc, dec, device=0)
            m.precision = prec
            y0 = torch.empty((n, 400, 400, 1), dtype=torch.float32, device="cuda")
            y = torch.empty_like(y0)
            m.predict_device(x, y0)
            bad = 0
            for i in range(args.reps):
                m.predict_device(x, y)
                if i % 10 == 0 and not torch.equal(y, y0):
                    bad += 1
            torch.cuda.synchronize()
            print(f"n={n} {prec}: {args.reps} repeats, {bad} mismatching checks", flush=True)
            assert bad == 0


if __name__ == "__main__":
    main()
