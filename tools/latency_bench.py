#!/usr/bin/env python3
"""BASELINE configs 1 and 3: latency of ONE `ml_super_resolution` call as the solvers make it
(host numpy in, host numpy out, 3 components of one 10x10 field -> 400x400), end to end through the
drop-in Python surface.  (The host-core comparator for the same call is timed by bench.py's cpu_baseline leg,
`single_field_call_ms`: only that leg may use oracle/.)

    python tools/latency_bench.py [--calls 50]
Prints one JSON line per variant."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ENC = os.path.join(GOLDEN, "vanilla_encoder10_to_400_swish_trained_upto_700_multiBC.h5")
STATS = os.path.join(GOLDEN, "standardization_stats_10to400_swish_trained_upto_700_multiBC.txt")


def timeit(fn, calls, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts = np.array(ts)
    return {"median_ms": round(float(np.median(ts)), 4), "p90_ms": round(float(np.percentile(ts, 90)), 4), "min_ms": round(float(ts.min()), 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=50)
    args = ap.parse_args()
    srcfd = importlib.import_module("sr-for-cfd_amd")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    pl = importlib.import_module("sr-for-cfd_amd.pipeline")
    h5 = importlib.import_module("sr-for-cfd_amd.h5")
    dec_w = synth.synthetic_decoder_weights(1)
    tmp = tempfile.mkdtemp()
    dec = os.path.join(tmp, "vanilla_decoder400_from_10_synthetic.h5")
    srcfd.SRModel.from_weights(None, dec_w, device=-1).save_h5(None, dec)
    ldc = h5.read_coarse_fields(os.path.join(GOLDEN, "coarse_ldc_Re1000_double_lid.h5"))
    bfs = h5.read_coarse_fields(os.path.join(GOLDEN, "coarse_bfs_Re400.h5"))
    for prec in ("fp32", "fp32x3", "bf16"):
        r = timeit(lambda: pl.ml_super_resolution(ldc, 10, 400, STATS, ENC, dec, precision=prec), args.calls)
        print(json.dumps({"call": "ml_super_resolution (LDC, PyCFD_ML_accelerated.py:764)", "precision": prec, **r}))
        r = timeit(lambda: pl.ml_super_resolution_bfs(bfs, 10, 400, STATS, ENC, dec, use_aspect_ratio_correction=True, lx=10.0, ly=3.0,
                                                      precision=prec), args.calls)
        print(json.dumps({"call": "ml_super_resolution (BFS: spline resample + adaptive blend, bfs_ml_accelerated.py:979)", "precision": prec, **r}))
    # the hand-off that follows the call in the solvers (PyCFD_ML_accelerated.py:936-943): host recipe vs one device pass
    bc = {"u": {"left": ("dirichlet", 0.0), "right": ("dirichlet", 0.0), "top": ("dirichlet", 1.0), "bottom": ("dirichlet", 0.0)},
          "v": {k: ("dirichlet", 0.0) for k in ("left", "right", "top", "bottom")},
          "p": {k: ("neumann", 0.0) for k in ("left", "right", "top", "bottom")}}
    types, values = pl.bc_arrays(bc)
    Var = np.zeros((3, 402, 402))

    def host_handoff():
        hr = pl.ml_super_resolution(ldc, 10, 400, STATS, ENC, dec, precision="bf16")
        pl.inject_into_solver_state(hr, Var)
        for k in range(3):  # apply_bc_configured, vectorised
            t, v = types[k], values[k]
            Var[k, 0, 1:-1] = 2 * v[0] - Var[k, 1, 1:-1] if t[0] == 0 else Var[k, 1, 1:-1]
            Var[k, -1, 1:-1] = 2 * v[1] - Var[k, -2, 1:-1] if t[1] == 0 else Var[k, -2, 1:-1]
            Var[k, 1:-1, -1] = 2 * v[2] - Var[k, 1:-1, -2] if t[2] == 0 else Var[k, 1:-1, -2]
            Var[k, 1:-1, 0] = 2 * v[3] - Var[k, 1:-1, 1] if t[3] == 0 else Var[k, 1:-1, 1]
    r = timeit(host_handoff, args.calls)
    print(json.dumps({"call": "ml_super_resolution + host injection into Var + ghost cells (numpy)", "precision": "bf16", **r}))
    r = timeit(lambda: pl.ml_super_resolution_into_solver(ldc, 10, 400, STATS, ENC, dec, (types, values), Var=Var, precision="bf16"), args.calls)
    print(json.dumps({"call": "ml_super_resolution_into_solver (fused device hand-off, float64 Var)", "precision": "bf16", **r}))


if __name__ == "__main__":
    main()
