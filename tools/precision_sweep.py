#!/usr/bin/env python3
"""Device-resident predict at 3 ... 768 samples for the f32-grade precisions: where does fp32x3 (split-bf16 GEMMs for ConvT#0 / #1) pass plain fp32?
    python tools/precision_sweep.py"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
srcfd = importlib.import_module('sr-for-cfd_amd')
synth = importlib.import_module('sr-for-cfd_amd.synth')
enc = srcfd.SRModel.load_h5(os.path.join(ROOT, 'tests/golden/vanilla_encoder10_to_400_swish_trained_upto_700_multiBC.h5'), None, device=-1).weights()
m = srcfd.SRModel.from_weights(enc, synth.synthetic_decoder_weights(1), device=0)
for n in (3, 6, 12, 24, 48, 96, 192, 384, 768):
    x = torch.randn((n, 10, 10, 1), device="cuda")
    y = torch.empty((n, 400, 400, 1), device="cuda")
    res = {}
    for prec in ("fp32", "fp32x3"):
        m.precision = prec
        for _ in range(5):
            m.predict_device(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.predict_device(x, y)
        torch.cuda.synchronize()
        res[prec] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"n={n:4d}: fp32 {res['fp32']:.3f} ms   fp32x3 {res['fp32x3']:.3f} ms   ratio {res['fp32x3'] / res['fp32']:.2f}")
