import sys, importlib, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
srcfd = importlib.import_module('sr-for-cfd_amd'); synth = importlib.import_module('sr-for-cfd_amd.synth')
enc = srcfd.SRModel.load_h5('/root/repo/tests/golden/vanilla_encoder10_to_400_swish_trained_upto_700_multiBC.h5', None, device=-1).weights()
m = srcfd.SRModel.from_weights(enc, synth.synthetic_decoder_weights(1), device=0)
m.precision = "bf16"
import os
for n in (3, 48, 128, 200, 256, 257, 300, 400, 512, 600, 768):
    x = torch.randn((n, 10, 10, 1), device="cuda"); y = torch.empty((n, 400, 400, 1), device="cuda")
    res = []
    for segenv in ("1", None):
        if segenv: os.environ["SRCFD_TAIL_SEG"] = segenv
        else: os.environ.pop("SRCFD_TAIL_SEG", None)
        for _ in range(5): m.predict_device(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): m.predict_device(x, y)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e3)
    print(f"n={n:4d}: unsegmented {res[0]:.3f} ms, auto {res[1]:.3f} ms -> {n / 3 / res[1] * 1e3:.0f} fields/s")
