#!/usr/bin/env python3
"""profiles/pmc_traffic_{tail,fp32}.json from the per-launch PMC averages of tools/pmc_summary.py.

    python tools/pmc_traffic.py profiles/r04/h_bf16_pmc_per_launch_avg.json profiles/r04/h_fp32_pmc_per_launch_avg.json

bench.py reports `roofline.traffic` from these files (it cannot run the profiler on itself).  Each file is stamped with a
hash of the kernel sources it describes (bench.py: TRAFFIC_SOURCES / kernel_source_stamp) and with the git commit it was
written at; bench.py refuses a file whose stamp differs from the tree it runs in.
Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts wide coalesced
reads at half their bytes -> doubled; WRITE_SIZE is exact for 16-byte stores."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


def bytes_of(c):
    return int(round(c.get("FETCH_SIZE", 0.0) * 1024 * 2)), int(round(c.get("WRITE_SIZE", 0.0) * 1024))


def main():
    bf, fp = sys.argv[1], sys.argv[2]
    rel = lambda p: os.path.relpath(os.path.abspath(p), ROOT)   # noqa: E731
    b = json.load(open(bf))
    tail = next(k for k in b if "tail16<" in k or k.endswith("tail16"))
    f, w = bytes_of(b[tail])
    others = {k.replace("void srcfd::", ""): dict(zip(("fetch_bytes", "write_bytes"), bytes_of(v))) for k, v in b.items() if "srcfd" in k and k != tail}
    out = {"source": f"{rel(bf)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes via tools/profile_run.sh, per-launch averages by "
                     "tools/pmc_summary.py; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction, KB -> bytes)",
           "config": {"fields": 256, "precision": "bf16", "out_dtype": "f32"}, "kernel": "tail(convT2-4+out)", "fetch_bytes": f, "write_bytes": w,
           "hbm_bytes_per_launch": f + w, "other_kernels": others, "kernel_source_stamp": bench.kernel_source_stamp("pmc_traffic_tail.json"),
           "git_commit_when_written": commit()}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_tail.json"), "w"), indent=1)
    p = json.load(open(fp))
    by = {k.replace("void srcfd::", ""): sum(bytes_of(v)) for k, v in p.items() if "srcfd" in k}
    out = {"source": f"{rel(fp)} (same collection and corrections as pmc_traffic_tail.json); sum over the kernels of one f32 step, one launch each at 768 samples",
           "config": {"fields": 256, "precision": "fp32", "out_dtype": "f32"}, "kernel": "all f32 kernels of one step",
           "hbm_bytes_per_launch": sum(by.values()), "by_kernel": by, "kernel_source_stamp": bench.kernel_source_stamp("pmc_traffic_fp32.json"),
           "git_commit_when_written": commit()}
    json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic_fp32.json"), "w"), indent=1)
    print("wrote profiles/pmc_traffic_tail.json, profiles/pmc_traffic_fp32.json")


if __name__ == "__main__":
    main()
