#!/usr/bin/env python3
"""Step time of the bf16 headline leg against the time since the device was idle, with the driver's clock / power readings beside it.

    python tools/clock_profile.py [--seconds 3] [--window 20] [--idle 2]

After `--idle` seconds of sleep the 768-field step is issued back to back for `--seconds`; every `--window` steps one line:
ms since start, ms per step over the window, sclk, power.  Shows (a) the ramp after an idle spell and (b) the level the clocks
settle at under this load -- the two things bench.py's untimed pre-warm has to be sized against.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--window", type=int, default=20)
    ap.add_argument("--idle", type=float, default=2.0)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    args = bench.parse_args([])
    job = bench.Job(args)
    torch = job.torch
    job.model.precision = "bf16"
    y = job.out_buffer("f32")
    try:
        pr = torch.cuda.get_device_properties(job.dev)
        pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    except Exception:
        pci = None

    def step():
        job.model.predict_device(job.x, y, in_affine=job.ain, out_affine=job.aout, nan_guard=True, nonfinite=job.bad)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    for rep in range(a.repeat):
        time.sleep(a.idle)
        print(f"# pass {rep}: after {a.idle:.1f} s idle; state {bench.gpu_state(pci)}", flush=True)
        t0 = time.perf_counter()
        while True:
            t1 = time.perf_counter()
            for _ in range(a.window):
                step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            st = bench.gpu_state(pci)
            print(f"{(t2 - t0) * 1e3:8.1f} ms  {(t2 - t1) / a.window * 1e3:7.4f} ms/step  sclk {st.get('sclk')}  {st.get('power_W')} W", flush=True)
            if t2 - t0 > a.seconds:
                break


if __name__ == "__main__":
    main()
