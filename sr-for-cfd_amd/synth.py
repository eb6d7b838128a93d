"""Seeded weights: stand-ins for the absent decoder_400 file, and Keras' default initialisation for training.

The reference's trained decoder files are absent from its checkout
(.MISSING_LARGE_BLOBS:29-34), so benchmarks and the smoke test run
random-init weights of the same architecture (sr-ae-conv.ipynb:c277-287).
Variance-preserving uniform init (variance 2.4 / fan_eff) keeps activations
O(1) through the six swish layers, as trained weights do.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

DECODER_SHAPES = {
    "dense_1": (50, 36864),
    "conv2d_transpose": (3, 3, 128, 256),
    "conv2d_transpose_1": (2, 2, 64, 128),
    "conv2d_transpose_2": (2, 2, 32, 64),
    "conv2d_transpose_3": (2, 2, 16, 32),
    "conv2d_transpose_4": (2, 2, 8, 16),
    "output_image_400": (3, 3, 8, 1),
}


def synthetic_decoder_weights(seed: int = 1, bias_scale: float = 0.1) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in DECODER_SHAPES.items():
        tr = name.startswith("conv2d_transpose")
        if len(shape) == 2:
            fan_in = shape[0]
        else:
            rf = shape[0] * shape[1]
            fan_in = rf * (shape[3] if tr else shape[2])
        fan_eff = fan_in / 4.0 if (tr and shape[0] == 2) else (fan_in / 2.25 if tr else fan_in)
        limit = math.sqrt(3.0 * 2.4 / fan_eff)
        w[f"{name}/kernel"] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        nb = shape[2] if tr else shape[-1]
        w[f"{name}/bias"] = (bias_scale * rng.standard_normal(nb)).astype(np.float32)
    return w


ENCODER_SHAPES = {
    "conv2d": (3, 3, 1, 64),
    "conv2d_1": (3, 3, 64, 128),
    "dense": (3200, 128),
    "latent_vector": (128, 50),
}


def _glorot(rng, shape):
    """Keras `glorot_uniform` + `compute_fans`: the last two axes are (fan_in, fan_out) units, also for the
    (kh,kw,Cout,Cin) kernels of Conv2DTranspose; limit = sqrt(6 / (fan_in + fan_out))."""
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    limit = math.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def keras_default_init(seed: int = 0):
    """Fresh encoder_10 / decoder_400 weights as `build_encoder_10` / `build_decoder_400` create them
    (sr-ae-conv.ipynb:c162-169, c277-287): glorot_uniform kernels, zero biases.  -> (enc_w, dec_w)."""
    rng = np.random.default_rng(seed)
    out = []
    for shapes in (ENCODER_SHAPES, DECODER_SHAPES):
        w = {}
        for name, shape in shapes.items():
            w[f"{name}/kernel"] = _glorot(rng, shape)
            nb = shape[2] if name.startswith("conv2d_transpose") else shape[-1]
            w[f"{name}/bias"] = np.zeros(nb, np.float32)
        out.append(w)
    return out[0], out[1]
