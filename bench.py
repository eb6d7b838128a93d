#!/usr/bin/env python3
"""Headline benchmark: batched super-resolution inference, BASELINE.json config 2.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the SR hot path over one batch of 256 synthetic fields
(256 x (10,10,3) -> (400,400,3) = 768 single-channel samples through
encoder_10 + decoder_400) per GPU, inputs resident in HBM, including the
per-channel standardise / de-standardise / NaN guard the reference does around
`predict` (PyCFD_ML_accelerated.py:841-876).  Weak scaling: every rank
processes its own 256 fields, no data-path collective.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks
itself as CHILD processes (`python -m torch.distributed.run --nproc-per-node N
bench.py ...`) before this process imports torch or touches a GPU, relays rank
0's JSON line and exits with the launcher's code.  Under an external
`torch.distributed.run` (WORLD_SIZE set) it is one rank; WORLD_SIZE != --gpus
is an error.  A rank exits non-zero when fewer devices than ranks are visible.

Prints ONE JSON line (rank 0).  Besides the contract's keys it carries
  roofline      dominant kernel (tail16), HIP events on the launch stream
  cpu_baseline  torch-CPU (oneDNN) restatement timed on the host cores (N = 1)
  parity_path   the same batch through the f32 kernels (the <= 1e-5 path):
                fields/s, ms/step, its own roofline, rel-L2 vs the f64 oracle
  train         BASELINE config 4: conv-AE training step, micro-batch 8 per
                GPU, flat-gradient all-reduce (RCCL) + Adam, samples/s
  tiled         BASELINE config 5: 40x40x3 -> 1600x1600x3 through 4x4 tiles, f16
  host_io       srcfd_predict with host buffers (PCIe + page faults included)
  env           every SRCFD_* variable seen (diagnostic switches are refused)
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ENCODER_H5 = os.path.join(GOLDEN, "vanilla_encoder10_to_400_swish_trained_upto_700_multiBC.h5")
STATS_TXT = os.path.join(GOLDEN, "standardization_stats_10to400_swish_trained_upto_700_multiBC.txt")

FIELDS = 256                      # BASELINE.json config 2
MACS_PER_SAMPLE = 140_024_128
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16/f16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3          # f32-input MFMA
PEAK_HBM_GBS = 8000.0
REFUSED_ENV = ("SRCFD_TAIL_ABLATE", "SRCFD_MID_ABLATE", "SRCFD_TAIL_PROF")   # switch work off / add syncs: never a headline


def srcfd_env():
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("SRCFD_")}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f16", "fp32"])
    ap.add_argument("--out-dtype", default="f32", choices=["f32", "bf16", "f16"])
    ap.add_argument("--fields", type=int, default=FIELDS)
    ap.add_argument("--workload", default="sr", choices=["sr", "tiled"],
                    help="sr: BASELINE config 2 (headline); tiled: config 5 as the headline line (40x40x3 -> 1600x1600x3, f16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity_path / train / tiled / host_io sub-records")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """Parent side of `--gpus N`: nothing here imports torch or initialises HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    print(line)
    return 0


def build_inputs(fields, seed, stats_lr, stats_hr):
    """x ~ N(0,1) in standardised space, mapped back to physical
    units so the engine's fused standardise does real work.  Sample order is
    field-major, component-minor: sample 3*f + c."""
    import numpy as np
    rng = np.random.default_rng(seed)
    xs = rng.standard_normal((fields, 10, 10, 3)).astype(np.float32)
    comps = ("u", "v", "p")
    lr = np.array([stats_lr[c] for c in comps], np.float32)  # (3,2) mean,std
    hr = np.array([stats_hr[c] for c in comps], np.float32)
    raw = xs * lr[:, 1] + lr[:, 0]
    x = np.ascontiguousarray(raw.transpose(0, 3, 1, 2).reshape(fields * 3, 10, 10, 1))
    ain = np.ascontiguousarray(np.tile(lr, (fields, 1)))
    aout = np.ascontiguousarray(np.tile(hr, (fields, 1)))
    return x, ain, aout


def measured_traffic(fields, precision, out_dtype):
    """HBM bytes per launch of the dominant kernel (16-bit paths) or per step (f32 path) from the committed rocprofv3 PMC
    passes (bench.py cannot run the profiler on itself); None unless the profile was taken on this exact configuration."""
    name = "pmc_traffic_fp32.json" if precision == "fp32" else "pmc_traffic_tail.json"
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            t = json.load(f)
    except OSError:
        return None, None
    c = t.get("config", {})
    if (c.get("fields"), c.get("precision"), c.get("out_dtype")) != (fields, precision, out_dtype):
        return None, None
    return t["hbm_bytes_per_launch"], t["source"]


def tail_flops(n):
    macs = 3 * 20_480_000 + 11_520_000  # ConvT#2..#4 + output conv
    return 2.0 * macs * n


def cpu_baseline(x, ain, aout, enc_w, dec_w, y_gpu, budget_s=12.0):
    """Reference stand-in on the host cores: the oracle's torch-CPU (oneDNN, the
    conv backend family TensorFlow uses) port of the network, Keras' default
    predict batch of 32, plus numpy pre/post as the reference does them.  The
    same leg checks the GPU outputs in `y_gpu` ({precision: first 8 samples})
    against the float64 oracle (relative L2 in standardised space)."""
    import numpy as np
    import torch
    from oracle.sr_oracle_torch import TorchSR
    model = TorchSR(enc_w, dec_w, torch.float32)
    bs = 32

    def std(lo, hi):
        return ((x[lo:hi] - ain[lo:hi, 0].reshape(-1, 1, 1, 1)) / ain[lo:hi, 1].reshape(-1, 1, 1, 1)).astype(np.float32)

    model.forward(std(0, bs), batch_size=bs)  # warm-up (oneDNN primitive creation)
    done, t0 = 0, time.perf_counter()
    while True:
        lo = done % (len(x) - bs + 1)
        y = model.forward(std(lo, lo + bs), batch_size=bs)
        y = y * aout[lo:lo + bs, 1].reshape(-1, 1, 1, 1) + aout[lo:lo + bs, 0].reshape(-1, 1, 1, 1)
        if np.isnan(y).any() or np.isinf(y).any():
            y = np.nan_to_num(y, nan=0.0, posinf=0.0, neginf=0.0)
        done += bs
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    # the solver-side shape of the call (configs 1 and 3): three batch-1 predicts per field, like PyCFD_ML_accelerated.py:841-876
    calls = []
    for _ in range(7):
        t1 = time.perf_counter()
        for c in range(3):
            xs1 = ((x[c:c + 1] - ain[c, 0]) / ain[c, 1]).astype(np.float32)
            _ = model.forward(xs1, batch_size=1) * aout[c, 1] + aout[c, 0]
        calls.append((time.perf_counter() - t1) * 1e3)
    # parity of the GPU paths: float64 network on the float32-standardised inputs, compared before de-standardisation
    k = 8
    ref = TorchSR(enc_w, dec_w, torch.float64).forward(std(0, k).astype(np.float64), batch_size=k).reshape(k, -1)
    rel = {}
    for prec, yg in y_gpu.items():
        ys = ((yg[:k].astype(np.float64) - aout[:k, 0].reshape(-1, 1, 1, 1)) / aout[:k, 1].reshape(-1, 1, 1, 1)).reshape(k, -1)
        rel[prec] = float(np.max(np.linalg.norm(ys - ref, axis=1) / np.linalg.norm(ref, axis=1)))
    return {
        "value": round(done / 3.0 / el, 3), "unit": "fields/s", "cores": int(torch.get_num_threads()), "kind": "port",
        "sample": f"{done} single-channel samples ({done // 3} fields) in batches of 32, f32, torch-CPU/oneDNN port of the network "
                  f"(TensorFlow/Keras not installable; SURVEY.md 8c), {el:.1f} s",
        "host_cpus": os.cpu_count(),
        "single_field_call_ms": round(float(np.median(calls)), 2),
        "gpu_rel_l2_vs_f64_oracle": rel,
    }


class Job:
    """One rank's device state."""

    def __init__(self, args):
        import numpy as np
        import torch
        import torch.distributed as dist
        self.np, self.torch, self.dist, self.args = np, torch, dist, args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device: libsrcfd has no CPU fallback")
        self.backend = os.environ.get("SRCFD_BENCH_BACKEND", "nccl")  # nccl = RCCL; gloo only to rehearse the rank logic on fewer GPUs
        ndev = torch.cuda.device_count()
        if self.world > ndev and self.backend == "nccl":
            raise SystemExit(f"bench.py: {self.world} ranks requested but only {ndev} HIP device(s) visible")
        local_rank %= ndev
        torch.cuda.set_device(local_rank)
        self.local_rank = local_rank
        self.dev = torch.device("cuda", local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)
        self.world_reported = dist.get_world_size() if self.world > 1 else 1
        self.srcfd = importlib.import_module("sr-for-cfd_amd")
        self.synth = importlib.import_module("sr-for-cfd_amd.synth")
        self.shard = importlib.import_module("sr-for-cfd_amd.shard")
        self.enc_w = self.srcfd.SRModel.load_h5(ENCODER_H5, None, device=-1).weights()   # real trained encoder (reference checkout)
        self.dec_w = self.synth.synthetic_decoder_weights(1)                              # decoder .h5 absent upstream -> random init
        self.model = self.srcfd.SRModel.from_weights(self.enc_w, self.dec_w, device=local_rank)
        self.stats_lr, self.stats_hr = self.srcfd.load_stats(STATS_TXT, 10, 400)
        self.x_h, self.ain_h, self.aout_h = build_inputs(args.fields, seed=self.rank, stats_lr=self.stats_lr, stats_hr=self.stats_hr)
        self.n = self.x_h.shape[0]
        self.x = torch.from_numpy(self.x_h).to(self.dev)
        self.ain = torch.from_numpy(self.ain_h).to(self.dev)
        self.aout = torch.from_numpy(self.aout_h).to(self.dev)
        self.bad = torch.zeros(1, dtype=torch.int64, device=self.dev)
        # One-time set-up of every leg's precision (operand packs, activation workspaces) happens here, through the ABI's
        # explicit reserve call, not lazily inside a leg's first warm-up step: a leg then starts on a GPU that the previous
        # leg has just left, instead of one that idled through host-side weight packing and hipMalloc.
        for prec in dict.fromkeys(([] if args.no_extras or args.precision == "fp32" else ["fp32"]) + [args.precision]):
            self.model.precision = prec
            self.model.reserve(self.n)
        self.y_out = {}

    def out_buffer(self, out_dtype):
        torch = self.torch
        if out_dtype not in self.y_out:
            odt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[out_dtype]
            self.y_out[out_dtype] = torch.empty((self.n, 400, 400, 1), dtype=odt, device=self.dev)
        return self.y_out[out_dtype]

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def timed(self, step, steps, warmup):
        """`warmup` untimed + exactly `steps` timed calls of `step`, bracketed by barrier + synchronize on both
        sides; returns the MAX over ranks of the seconds the timed calls took."""
        torch = self.torch
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        return self.shard.max_over_ranks(time.perf_counter() - t0, device=self.dev if self.backend == "nccl" else None)

    def kernel_profile(self, step, reps):
        """Per-step kernel times from HIP events recorded by the engine on the launch stream around every
        launch: a step's launches of one name are SUMMED (the f32 path runs 768 samples as 3 chunks of 256),
        then averaged over `reps` steps."""
        np = self.np
        self.model.set_profiling(True)
        acc = {}
        for _ in range(reps):
            step()
            per_step = {}
            for name, ms in self.model.get_profile():
                per_step[name] = per_step.get(name, 0.0) + ms
            for name, ms in per_step.items():
                acc.setdefault(name, []).append(ms)
        self.model.set_profiling(False)
        return {k: round(float(np.mean(v)), 4) for k, v in acc.items()}

    # -- the SR batch (config 2) at one precision -------------------------------------------------------------
    def run_sr(self, precision, out_dtype, steps, warmup):
        torch, args = self.torch, self.args
        self.model.precision = precision
        y = self.out_buffer(out_dtype)     # one result buffer per output type, shared by the legs (allocated once)
        self.bad.zero_()

        def step():
            self.model.predict_device(self.x, y, in_affine=self.ain, out_affine=self.aout, nan_guard=True, nonfinite=self.bad)

        dt = self.timed(step, steps, warmup)
        ms = dt / steps * 1e3
        rec = {"value": round(self.shard.aggregate_throughput(args.fields, self.world, ms * 1e-3), 2), "unit": "fields/s",
               "ms_per_step": round(ms, 4), "steps": steps, "dtype": precision, "out_dtype": out_dtype,
               "tflops_model": round(2.0 * MACS_PER_SAMPLE * self.n * self.world / (ms * 1e-3) / 1e12, 2),
               "nonfinite": int(self.bad.item())}
        if self.rank == 0:
            kernels = self.kernel_profile(step, max(3, min(steps, 10)))
            tot = sum(kernels.values())
            rec["kernels_ms"] = kernels
            rec["kernels_ms_sum"] = round(tot, 4)
            rec["launch_gap_ms"] = round(ms - tot, 4)
            if precision in ("bf16", "f16"):
                dom = "tail(convT2-4+out)"
                fl = tail_flops(self.n)
                ach = fl / (kernels[dom] * 1e-3) / 1e12
                traffic, traffic_src = measured_traffic(args.fields, precision, out_dtype)
                floor = self.n * 2_240_000 / 64 * 23.6 / 1024 / 2.4e9 * 1e3
                rec["roofline"] = {
                    "bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "bytes/launch (HBM, PMC)",
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": self.n * 160000 * (2 + (4 if out_dtype == "f32" else 2)),
                    "avg_launch_ms": kernels[dom], "algorithmic_flops_per_launch": fl,
                    # what actually binds (not expressible as "hbm" | "mfma"): 2 240 000 swish activations per sample in this
                    # kernel at the measured 23.6 SIMD-cycles per 64 (profiles/r01/microbench6_valu_throughput.txt), 1024 SIMDs, 2.4 GHz
                    "valu_swish_floor_ms": round(floor, 4), "frac_of_valu_swish_floor": round(floor / kernels[dom], 4),
                    "note": "swish = 2 quarter-rate transcendentals per activation: exact swish caps this network at ~0.28 of the MFMA "
                            "peak on the VALU transcendental rate (DESIGN.md 4.2)"}
            else:
                # whole f32 step: all 768 samples' FLOPs over the SUM of every launch of the step (all chunks)
                fl = 2.0 * MACS_PER_SAMPLE * self.n
                ach = fl / (tot * 1e-3) / 1e12
                if tot > ms * 1.10:   # event records between launches add a few per cent
                    raise SystemExit(f"bench.py: kernel times ({tot:.3f} ms) exceed the step ({ms:.3f} ms): profile is not per step")
                traffic, traffic_src = measured_traffic(args.fields, precision, out_dtype)
                rec["roofline"] = {"bound": "mfma", "kernel": "all f32 kernels of one step (sum over all launches)", "achieved": round(ach, 2),
                                   "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4), "traffic": traffic,
                                   "traffic_unit": "bytes/step (HBM, PMC)", "traffic_source": traffic_src,
                                   "algorithmic_bytes_per_launch": self.n * (100 + 160000) * 4,
                                   "avg_launch_ms": round(tot, 4), "algorithmic_flops_per_launch": fl,
                                   "note": "f32 MFMAs and vector instructions of a SIMD do not overlap on gfx950 (profiles/r02/d_microbench8...): "
                                           "the swish / output-conv vector work of this path is paid on top of the MFMA time"}
        return rec, y

    # -- config 4: training step ---------------------------------------------------------------------------------
    def run_train(self, steps=30, warmup=3, batch=8):
        np, torch = self.np, self.torch
        tr = importlib.import_module("sr-for-cfd_amd.train")
        ds = importlib.import_module("sr-for-cfd_amd.datasets")
        enc, dec = self.synth.keras_default_init(0)                     # identical replicas: same seed on every rank
        model = self.srcfd.SRModel.from_weights(enc, dec, device=self.local_rank)
        t = tr.Trainer(model, max_batch=batch)
        # the notebook's own dummy recipe (sr-ae-conv.ipynb:c72-91): x_hr ~ N(0,1), x_lr = avg_pool(x_hr, 40)
        rng = np.random.default_rng(1000 + self.rank)
        y_h = rng.standard_normal((batch, 400, 400, 1)).astype(np.float32)
        x = torch.from_numpy(ds.avg_pool(y_h, 40)).to(self.dev)
        y = torch.from_numpy(y_h).to(self.dev)
        gb = batch * self.world

        def step():
            t.grads.zero_()
            t.forward_backward(x, y, gb)
            tr.allreduce_sum_(t.grads)      # the step's only collective (RCCL when backend == nccl)
            t.apply_adam()

        dt = self.timed(step, steps, warmup)
        ms = dt / steps * 1e3
        t.sse.zero_()
        t.forward_backward(x, y, gb)
        loss = float(t.sse.item()) / (batch * 160000)
        rec = {"metric": "conv-AE training samples/s (10x10->400x400, f32, Adam; BASELINE config 4)", "value": round(gb / (ms * 1e-3), 2),
               "unit": "samples/s", "ms_per_step": round(ms, 4), "steps": steps, "warmup": warmup, "micro_batch": batch, "global_batch": gb,
               "scaling": "weak", "dtype": "f32", "params": t.n_params, "collective": "none" if self.world == 1 else
               f"all_reduce(sum) of {t.n_params} f32 per step, backend {self.backend}",
               "tflops_model": round(3 * 2 * MACS_PER_SAMPLE * gb / (ms * 1e-3) / 1e12, 2),
               "frac_f32_mfma_peak": round(3 * 2 * MACS_PER_SAMPLE * batch / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 4),
               "loss_finite": bool(np.isfinite(loss))}
        t.close()
        model.close()
        return rec

    # -- config 5: tiled SR -------------------------------------------------------------------------------------
    def run_tiled(self, steps=20, warmup=3):
        """One 40x40x3 field = 4x4 tiles x 3 components = 48 samples -> 1600x1600x3, f16 operands; the tile samples are
        sharded contiguously over the ranks (shard.shard_range), no collective in the timed region."""
        np, torch = self.np, self.torch
        rng = np.random.default_rng(5)
        field = rng.standard_normal((40, 40, 3)).astype(np.float32)
        tiles = np.ascontiguousarray(field.reshape(4, 10, 4, 10, 3).transpose(0, 2, 4, 1, 3).reshape(48, 10, 10, 1))
        lo, hi = self.shard.shard_range(48, self.rank, self.world)
        n = hi - lo
        self.model.precision = "f16"
        x = torch.from_numpy(tiles[lo:hi]).to(self.dev)
        y = torch.empty((max(n, 1), 400, 400, 1), dtype=torch.float32, device=self.dev)

        def step():
            if n:
                self.model.predict_device(x, y[:n], nan_guard=True, nonfinite=self.bad)

        dt = self.timed(step, steps, warmup)
        ms = dt / steps * 1e3
        return {"metric": "tiled SR fields/s (40x40x3 -> 1600x1600x3 via 4x4 tiles; BASELINE config 5)", "value": round(1.0 / (ms * 1e-3), 2),
                "unit": "fields/s", "ms_per_field": round(ms, 4), "steps": steps, "warmup": warmup, "dtype": "f16", "tile_samples": 48,
                "tile_samples_this_rank": n, "scaling": "strong", "mpix_per_s": round(1600 * 1600 * 3 / (ms * 1e-3) / 1e6, 1)}

    # -- host-buffer entry ---------------------------------------------------------------------------------------
    def run_host_io(self):
        """srcfd_predict (numpy in -> numpy out): what a solver gets when it hands over host arrays.  `value` of the
        headline never includes any of this."""
        np = self.np
        out = {}
        self.model.precision = "bf16"
        for n in (3, self.n):
            x = self.x_h[:n]
            ai, ao = self.ain_h[:n], self.aout_h[:n]
            self.model.predict(x, in_affine=ai, out_affine=ao, nan_guard=True)
            fresh = []
            for _ in range(3):
                t0 = time.perf_counter()
                y = self.model.predict(x, in_affine=ai, out_affine=ao, nan_guard=True)
                fresh.append(time.perf_counter() - t0)
            reused = []
            for _ in range(3):
                t0 = time.perf_counter()
                self.model.predict(x, in_affine=ai, out_affine=ao, nan_guard=True, out=y)
                reused.append(time.perf_counter() - t0)
            out[f"samples_{n}"] = {"fresh_result_ms": round(min(fresh) * 1e3, 3), "reused_result_ms": round(min(reused) * 1e3, 3),
                                   "fields_per_s_fresh": round(n / 3 / min(fresh), 1), "fields_per_s_reused": round(n / 3 / min(reused), 1),
                                   "d2h_GBps_reused": round(y.nbytes / min(reused) / 1e9, 2)}
        out["note"] = "pageable host memory; a fresh 491.5 MB result pays first-touch page faults on top of the D2H copy"
        return out


def dry_run(args, env):
    """SRCFD_BENCH_DRYRUN=1: the rank plumbing only (process group, barrier, MAX over ranks, rank 0's line) with no
    device work, so that `--gpus N` can be rehearsed on a box without GPUs (tests/test_distributed.py).  `value` is null."""
    import torch
    import torch.distributed as dist
    shard = importlib.import_module("sr-for-cfd_amd.shard")
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    t = shard.max_over_ranks(float(rank + 1))
    lo, hi = shard.shard_range(48, rank, world)
    cnt = torch.tensor([hi - lo], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(cnt)
    if rank == 0:
        print(json.dumps({"metric": "SR fields/sec (10x10->400x400, 3-ch) @batch256", "value": None, "unit": "fields/s", "dry_run": True,
                          "n_gpus": world, "world_size_reported": dist.get_world_size() if world > 1 else 1, "steps": args.steps,
                          "warmup": args.warmup, "max_over_ranks": t, "tile_samples_covered": int(cnt.item()), "env": env}))
    if world > 1:
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    env = srcfd_env()
    bad_env = [k for k in env if k in REFUSED_ENV and env[k] not in ("", "0")]
    if bad_env and os.environ.get("SRCFD_BENCH_ALLOW_DIAG", "0") in ("", "0"):
        raise SystemExit(f"bench.py: diagnostic switches {bad_env} are set: they skip work or add synchronisation; refusing to report a number")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args, argv))
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}; launch one rank per GPU (or let --gpus start them)")

    if os.environ.get("SRCFD_BENCH_DRYRUN", "0") not in ("", "0"):
        return dry_run(args, env)

    job = Job(args)
    torch, np = job.torch, job.np
    extras = not args.no_extras

    if args.workload == "tiled":
        rec = job.run_tiled(args.steps, args.warmup)
        if job.rank == 0:
            print(json.dumps({"metric": rec["metric"], "value": rec["value"], "unit": rec["unit"], "n_gpus": job.world,
                              "world_size_reported": job.world_reported, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": rec["ms_per_field"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": "f16", "data": "synthetic",
                              "config": {"workload": "BASELINE config 5: 40x40x3 -> 1600x1600x3 via 4x4 non-overlapping 10x10 tiles "
                                                     "(48 single-channel samples), same weights, tile samples sharded over the ranks",
                                         "parallelism": f"tile-sharded x{job.world}, no collective"},
                              "detail": rec, "env": env}))
        if job.world > 1:
            job.dist.destroy_process_group()
        return

    y_gpu = {}
    parity = train = tiled = host_io = None
    extra_errors = {}

    def leg(name, fn):
        """A sub-record must never cost the headline: an exception in one (on every rank alike: same code, same data shapes)
        is recorded in the line instead of ending the run."""
        try:
            return fn()
        except Exception as e:   # noqa: BLE001
            extra_errors[name] = f"{type(e).__name__}: {e}"[:300]
            return None

    # Order of the legs: the f32 parity path first (parity is the gate; it also means the headline's W warm-up steps do not
    # start on a GPU that has idled through model loading and weight packing), then the headline, then the other sub-records.
    if extras and args.precision != "fp32":
        def _parity():
            rec, y_par = job.run_sr("fp32", "f32", max(5, min(args.steps, 20)), min(args.warmup, 3))
            if job.rank == 0 and job.world == 1 and not args.no_cpu_baseline:
                y_gpu["fp32"] = y_par[:8].cpu().numpy()
            return rec
        parity = leg("parity_path", _parity)

    head, y_head = job.run_sr(args.precision, args.out_dtype, args.steps, args.warmup)
    if job.rank == 0 and job.world == 1 and not args.no_cpu_baseline and args.out_dtype == "f32":
        y_gpu[args.precision] = y_head[:8].cpu().numpy()
    del y_head

    if extras:
        torch.cuda.empty_cache()
        train = leg("train", job.run_train)
        tiled = leg("tiled", job.run_tiled)
        if job.rank == 0 and job.world == 1:
            host_io = leg("host_io", job.run_host_io)

    cpu = None
    if job.rank == 0 and job.world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(job.x_h, job.ain_h, job.aout_h, job.enc_w, job.dec_w, y_gpu)
        if parity is not None and "fp32" in cpu["gpu_rel_l2_vs_f64_oracle"]:
            parity["rel_l2_vs_f64_oracle"] = cpu["gpu_rel_l2_vs_f64_oracle"]["fp32"]
            parity["tolerance"] = 1e-5

    if job.rank == 0:
        out = {
            "metric": "SR fields/sec (10x10->400x400, 3-ch) @batch256", "value": head["value"], "unit": "fields/s",
            "n_gpus": job.world, "world_size_reported": job.world_reported, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: batched SR inference, 256 fields x (10,10,3)->(400,400,3) per GPU = 768 "
                                   "single-channel encoder_10+decoder_400 passes, fused standardise/de-standardise/NaN guard",
                       "fields_per_gpu": args.fields, "samples_per_gpu": job.n, "out_dtype": args.out_dtype,
                       "weights": "encoder: reference multiBC .h5; decoder: random init seed 1 (reference decoder .h5 absent)",
                       "parallelism": f"sample-sharded x{job.world}, no collective", "backend": job.backend if job.world > 1 else None},
            "tflops_model": head["tflops_model"], "nonfinite": head["nonfinite"],
            "kernels_ms": head.get("kernels_ms"), "kernels_ms_sum": head.get("kernels_ms_sum"), "launch_gap_ms": head.get("launch_gap_ms"),
            "roofline": head.get("roofline"),
            "cpu_baseline": cpu,
            "parity_path": parity, "train": train, "tiled": tiled, "host_io": host_io,
            "env": env,
        }
        if extra_errors:
            out["sub_record_errors"] = extra_errors
        if bad_env:
            out["INVALID_diagnostic_run"] = bad_env   # SRCFD_BENCH_ALLOW_DIAG=1 with a DIAG=1 build: tools/ablate*.sh only
        print(json.dumps(out))
    if job.world > 1:
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
