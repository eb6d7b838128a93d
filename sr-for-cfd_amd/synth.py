"""Seeded stand-in weights for the decoder_400 half.

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
