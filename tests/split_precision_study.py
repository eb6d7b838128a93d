#!/usr/bin/env python3
"""Time-boxed experiment of round 4: can 16-bit MFMAs carry the <= 1e-5 parity path?

    python tests/split_precision_study.py [--samples 4] > profiles/r04/n_split_precision_study.txt

f32-input MFMAs run at 1/16 of the bf16 / f16 rate on gfx950 and do not overlap vector work; 16-bit MFMAs do.  If every
operand is SPLIT into 16-bit terms (x = hi + lo [+ lo2], weights split on the host) and the cross products that matter
are accumulated in f32, a GEMM costs 3 (two terms) or 6 (three terms) 16-bit MFMAs instead of one f32 MFMA: 3/16 or
6/16 of the f32 matrix time.  This script EMULATES that arithmetic on the CPU (numpy; test infrastructure, nothing here
is on the product path): products of two 16-bit values are exact in float32, so a float32 matmul of the split operands
is the MFMA result up to summation order.  Everything else (bias, swish, the activations handed on) stays float32, as a
kernel would keep it.

Modes
  f32      plain float32 matmuls (what the shipped parity path computes, up to summation order)
  bf16x2   x = hi + lo in bfloat16:  hi.hi + hi.lo + lo.hi                                   (3 MFMAs)
  bf16x3   x = hi + mid + lo:        hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid      (6 MFMAs)
  f16x2    x = hi + lo in float16, lo kept as lo * 2^11 (stays in f16's normal range):
           hi.hi + 2^-11 (hi.lo' + lo'.hi)                                                   (3 MFMAs, two accumulators)
  f16x2u   the same without the scaling: lo underflows into f16 denormals                    (3 MFMAs)
Reported: relative L2 against the float64 oracle, max over samples -- for every layer in isolation (its float64-exact
input rounded to float32) and for the whole network, on three weight / input sets: the reference's trained encoder + a
seeded synthetic decoder on N(0,1) inputs; Keras-default-initialised weights (an untrained model); the first set on
inputs 200 sigma out.  Bar: 1e-5 on the whole network, all three sets."""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sr_oracle as o            # noqa: E402
from oracle.sr_oracle_lowp import round_bf16, round_f16   # noqa: E402

F = np.float32


def split(a, rnd, terms, scale=1.0):
    """a = t0 + t1/scale (+ t2/scale^2): list of float32 arrays holding 16-bit values."""
    out, rest = [], a.astype(F)
    for i in range(terms):
        t = rnd(rest * F(scale ** i))
        out.append(t)
        rest = (rest - t / F(scale ** i)).astype(F)
    return out


def make_mm(mode):
    if mode == "f32":
        return lambda x, w: x.astype(F) @ w.astype(F)
    if mode in ("bf16x2", "bf16x3"):
        n = 2 if mode == "bf16x2" else 3

        def mm(x, w):
            xs, ws = split(x, round_bf16, n), split(w, round_bf16, n)
            acc = np.zeros(x.shape[:-1] + (w.shape[-1],), F)
            for i in range(n):            # smallest terms first would be kinder; MFMAs accumulate in issue order: big first
                for j in range(n):
                    if i + j <= n - 1 or (n == 3 and i == 1 and j == 1):
                        acc += xs[i] @ ws[j]
            return acc
        return mm
    if mode in ("f16x2", "f16x2u"):
        sc = 2048.0 if mode == "f16x2" else 1.0

        def mm(x, w):
            xs, ws = split(x, round_f16, 2, sc), split(w, round_f16, 2, sc)
            return (xs[0] @ ws[0] + (xs[0] @ ws[1] + xs[1] @ ws[0]) * F(1.0 / sc)).astype(F)
        return mm
    raise ValueError(mode)


def swish32(z):
    z = z.astype(F)
    return (z / (F(1) + np.exp(-z, dtype=F))).astype(F)


def layers_of(enc_w, dec_w):
    """(name, kind, kernel, bias, activation) in execution order."""
    L = [("conv2d", "conv_s2", enc_w["conv2d/kernel"], enc_w["conv2d/bias"], True),
         ("conv2d_1", "conv_s1", enc_w["conv2d_1/kernel"], enc_w["conv2d_1/bias"], True),
         ("dense", "dense", enc_w["dense/kernel"], enc_w["dense/bias"], True),
         ("latent_vector", "dense", enc_w["latent_vector/kernel"], enc_w["latent_vector/bias"], False),
         ("dense_1", "dense", dec_w["dense_1/kernel"], dec_w["dense_1/bias"], True)]
    for name in o.DECODER_LAYERS[1:6]:
        L.append((name, "convt", dec_w[f"{name}/kernel"], dec_w[f"{name}/bias"], True))
    L.append(("output_image_400", "conv_s1", dec_w["output_image_400/kernel"], dec_w["output_image_400/bias"], False))
    return L


def run_layer(kind, x, w, b, act, mm):
    """One layer with matmul `mm` (float32 in / out); same index conventions as oracle/sr_oracle.py."""
    x = x.astype(F)
    if kind == "dense":
        z = mm(x.reshape(x.shape[0], -1), w) + b.astype(F)
    elif kind == "convt":
        n, h, wd, _ = x.shape
        kh, kw, cout, _ = w.shape
        z = np.zeros((n, (h - 1) * 2 + kh, (wd - 1) * 2 + kw, cout), F)
        for a in range(kh):
            for bb in range(kw):
                z[:, a:a + (h - 1) * 2 + 1:2, bb:bb + (wd - 1) * 2 + 1:2, :] += mm(x, np.ascontiguousarray(w[a, bb].T))
        z += b.astype(F)
    else:
        stride = 2 if kind == "conv_s2" else 1
        n, h, wd, cin = x.shape
        kh, kw, _, cout = w.shape
        oh, pt, pb = o.same_padding(h, kh, stride)
        ow, pl, pr = o.same_padding(wd, kw, stride)
        xp = np.zeros((n, h + pt + pb, wd + pl + pr, cin), F)
        xp[:, pt:pt + h, pl:pl + wd, :] = x
        # one matmul over the whole (tap, cin) contraction, as the implicit GEMM does
        cols = np.concatenate([xp[:, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride, :]
                               for ky in range(kh) for kx in range(kw)], axis=-1)
        z = mm(cols, w.reshape(kh * kw * cin, cout)) + b.astype(F)
    return swish32(z) if act else z.astype(F)


def ref_layer(kind, x, w, b, act):
    x = x.astype(np.float64)
    a = "swish" if act else "linear"
    if kind == "dense":
        return o.dense(x.reshape(x.shape[0], -1), w, b, a)
    if kind == "convt":
        return o.conv2d_transpose(x, w, b, 2, "valid", a)
    return o.conv2d(x, w, b, 2 if kind == "conv_s2" else 1, "same", a)


def rel(y, r):
    y, r = y.reshape(y.shape[0], -1).astype(np.float64), r.reshape(r.shape[0], -1)
    return float(np.max(np.linalg.norm(y - r, axis=1) / np.linalg.norm(r, axis=1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4)
    args = ap.parse_args()
    srcfd = importlib.import_module("sr-for-cfd_amd")
    synth = importlib.import_module("sr-for-cfd_amd.synth")
    golden = os.path.join(ROOT, "tests", "golden")
    enc_tr = srcfd.SRModel.load_h5(os.path.join(golden, "vanilla_encoder10_to_400_swish_trained_upto_700_multiBC.h5"), None, device=-1).weights()
    dec_sy = synth.synthetic_decoder_weights(1)
    enc_k, dec_k = synth.keras_default_init(0)
    rng = np.random.default_rng(0)
    x1 = rng.standard_normal((args.samples, 10, 10, 1)).astype(F)
    sets = [("trained encoder + synthetic decoder, x ~ N(0,1)", enc_tr, dec_sy, x1),
            ("Keras-default-init weights (untrained), x ~ N(0,1)", enc_k, dec_k, x1),
            ("trained encoder + synthetic decoder, x = 200 sigma", enc_tr, dec_sy, (200.0 * x1).astype(F))]
    modes = ["f32", "bf16x2", "bf16x3", "f16x2", "f16x2u"]
    print(__doc__.split("Modes")[0].strip().splitlines()[0])
    print(f"samples per set: {args.samples}; relative L2 vs the float64 oracle, max over samples\n")
    verdict = {m: True for m in modes}
    for title, enc_w, dec_w, x in sets:
        L = layers_of(enc_w, dec_w)
        # float64 reference activations
        refs, a = [], x.astype(np.float64)
        for name, kind, w, b, act in L:
            a = ref_layer(kind, a.reshape(-1, 12, 12, 256) if name == "conv2d_transpose" else a, w, b, act)
            refs.append(a)
        print(f"== {title}")
        print(f"{'layer':22s}" + "".join(f"{m:>12s}" for m in modes))
        full = {m: x for m in modes}
        for li, (name, kind, w, b, act) in enumerate(L):
            row = []
            xin_ref = (x if li == 0 else refs[li - 1]).astype(F)
            if name == "conv2d_transpose":
                xin_ref = xin_ref.reshape(-1, 12, 12, 256)
            for m in modes:
                mm = make_mm(m)
                row.append(rel(run_layer(kind, xin_ref, w, b, act, mm), refs[li]))        # the layer alone
                xi = full[m].reshape(-1, 12, 12, 256) if name == "conv2d_transpose" else full[m]
                full[m] = run_layer(kind, xi, w, b, act, mm)                              # the chain
            print(f"{name:22s}" + "".join(f"{v:12.2e}" for v in row))
        row = [rel(full[m], refs[-1]) for m in modes]
        for m, v in zip(modes, row):
            verdict[m] &= v <= 1e-5
        print(f"{'WHOLE NETWORK':22s}" + "".join(f"{v:12.2e}" for v in row) + "\n")
    print("whole network <= 1e-5 on all three sets: " + ", ".join(f"{m}: {'yes' if ok else 'NO'}" for m, ok in verdict.items()))


if __name__ == "__main__":
    main()
