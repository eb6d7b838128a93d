"""Emulation of the 16-bit throughput path on the CPU (numpy).

TEST INFRASTRUCTURE ONLY (same import rule as sr_oracle.py).

Same layer semantics as sr_oracle.py, but weights and the activations handed
from one layer to the next are rounded to bfloat16 / float16 (round to nearest
even) while sums run in float64 -- i.e. what libsrcfd's SRCFD_PREC_BF16/F16
path computes up to accumulation order and the hardware exp/rcp.  It separates
"the kernel is wrong" from "16-bit operands cost accuracy": GPU-vs-emulation
must agree to ~1e-3, while emulation-vs-float64 shows the precision price.
The reference itself has no reduced-precision path.
"""
from __future__ import annotations

import numpy as np

from . import sr_oracle as o


def round_bf16(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def round_f16(a: np.ndarray) -> np.ndarray:
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


LOG2E = 1.4426950408889634


def superres_forward_lowp(x, enc_w, dec_w, kind="bf16", return_all=False):
    """(N,10,10,1) -> (N,400,400,1); every kernel operand rounded to `kind`.

    Mirrors libsrcfd's log2(e) folding (kernels_bf16.hip header): swish outputs
    are *stored* as round(log2e * a); a linear layer that consumes them uses
    round(W / log2e); dense_1, which follows the linear latent, round(W * log2e).
    """
    rnd = round_bf16 if kind == "bf16" else round_f16
    q = lambda w: rnd(w).astype(np.float64)
    q_div = lambda w: LOG2E * rnd(np.asarray(w, np.float64) / LOG2E).astype(np.float64)   # consumer of scaled activations
    q_mul = lambda w: rnd(np.asarray(w, np.float64) * LOG2E).astype(np.float64) / LOG2E   # producer after a linear layer
    r = lambda a: rnd(a.astype(np.float32)).astype(np.float64)
    rs = lambda a: rnd((a * LOG2E).astype(np.float32)).astype(np.float64) / LOG2E          # stored scaled by log2e
    acts = {}
    x = np.asarray(x, dtype=np.float64)
    # conv1 runs in f32 on the VALU with unrounded weights; its output is stored 16-bit
    a = rs(o.conv2d(x, enc_w["conv2d/kernel"], enc_w["conv2d/bias"], 2, "same", "swish"))
    a = rs(o.conv2d(a, q(enc_w["conv2d_1/kernel"]), enc_w["conv2d_1/bias"], 1, "same", "swish"))
    a = rs(o.dense(a.reshape(a.shape[0], -1), q(enc_w["dense/kernel"]), enc_w["dense/bias"], "swish"))
    z = r(o.dense(a, q_div(enc_w["latent_vector/kernel"]), enc_w["latent_vector/bias"], "linear"))
    acts["latent"] = z
    h = rs(o.dense(z, q_mul(dec_w["dense_1/kernel"]), dec_w["dense_1/bias"], "swish")).reshape(-1, 12, 12, 256)
    for i, name in enumerate(o.DECODER_LAYERS[1:6]):
        h = rs(o.conv2d_transpose(h, q(dec_w[f"{name}/kernel"]), dec_w[f"{name}/bias"], 2, "valid", "swish"))
        acts[f"t{i}"] = h
    y = o.conv2d(h, q_div(dec_w["output_image_400/kernel"]), dec_w["output_image_400/bias"], 1, "same", "linear")
    if return_all:
        return y, acts
    return y
