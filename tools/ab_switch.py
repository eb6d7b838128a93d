#!/usr/bin/env python3
"""A / B of one library switch inside one process on one box: outputs compared bit for bit, kernel times from the engine's HIP events.

    python tools/ab_switch.py SRCFD_MID 1 2 [--fields 256] [--precision bf16] [--reps 30]

The switches are read per call (engine.h Switches), so both arms run on the same handle, inputs and clocks, interleaved.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("values", nargs="+")
    ap.add_argument("--fields", type=int, default=256)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=30)
    a = ap.parse_args()
    args = bench.parse_args(["--fields", str(a.fields)])
    job = bench.Job(args)
    torch = job.torch
    job.model.precision = a.precision
    y = job.out_buffer("f32")

    def step():
        job.model.predict_device(job.x, y, in_affine=job.ain, out_affine=job.aout, nan_guard=True, nonfinite=job.bad)

    try:
        pr = torch.cuda.get_device_properties(job.dev)
        pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:   # noqa: BLE001
        pci = None
    outs, times = {}, {v: {} for v in a.values}
    for v in a.values:
        os.environ[a.name] = v
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        outs[v] = y.clone()
        print(v, job.model.last_plan(), flush=True)
    for rnd in range(3):
        for v in a.values:
            os.environ[a.name] = v
            for _ in range(10):
                step()
            k = job.kernel_profile(step, a.reps)
            for name, ms in k.items():
                times[v].setdefault(name, []).append(ms)
    import time
    wall = {v: [] for v in a.values}
    state = {v: [] for v in a.values}
    for rnd in range(6):
        for v in a.values:
            os.environ[a.name] = v
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                step()
            torch.cuda.synchronize()
            wall[v].append((time.perf_counter() - t0) * 10.0)
            st = bench.gpu_state(pci)
            state[v].append(f"{st.get('sclk')}/{st.get('power_W')}W")
    for v in a.values:
        print(f"{a.name}={v}: wall ms per step over 100 steps, 6 interleaved rounds: " + " ".join(f"{t:.4f}" for t in wall[v]) + f"; median {float(np.median(wall[v])):.4f}; sclk/power after each: " + " ".join(state[v]))
    ref = outs[a.values[0]]
    for v in a.values:
        same = bool(torch.equal(outs[v], ref))
        tot = sum(float(np.median(t)) for t in times[v].values())
        print(f"{a.name}={v}: bit-identical to {a.values[0]}: {same}; kernels (median of 3 rounds x {a.reps}): "
              + ", ".join(f"{n} {float(np.median(t)):.4f}" for n, t in times[v].items()) + f"; sum {tot:.4f} ms")


if __name__ == "__main__":
    main()
