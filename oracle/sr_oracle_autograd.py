"""Gradient oracle: the SR network in differentiable torch CPU ops (float64) + reference Adam.

TEST INFRASTRUCTURE ONLY (same import rule as sr_oracle.py).  Restates
sr-ae-conv.ipynb:c306-320 (`train_step`: loss = reduce_mean(mse), tape.gradient) and Keras'
Adam update rule; PARITY UNPINNED for the same reasons as the forward oracle.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .sr_oracle import DECODER_LAYERS, ENCODER_LAYERS, same_padding


def flat_order(enc_w, dec_w) -> List[str]:
    """Keras trainable_weights order: per layer kernel then bias, encoder then decoder."""
    names = []
    for l in list(ENCODER_LAYERS if enc_w is not None else ()) + list(DECODER_LAYERS if dec_w is not None else ()):
        names += [f"{l}/kernel", f"{l}/bias"]
    return names


def loss_and_grads(x: np.ndarray, y: np.ndarray, enc_w: Dict[str, np.ndarray], dec_w: Dict[str, np.ndarray]) -> Tuple[float, np.ndarray]:
    w = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in {**enc_w, **dec_w}.items()}
    t = torch.tensor(np.asarray(x, np.float64)).permute(0, 3, 1, 2)
    _, pt, pb = same_padding(t.shape[2], 3, 2)
    _, pl, pr = same_padding(t.shape[3], 3, 2)
    h = F.silu(F.conv2d(F.pad(t, (pl, pr, pt, pb)), w["conv2d/kernel"].permute(3, 2, 0, 1).contiguous(), w["conv2d/bias"], stride=2))
    h = F.silu(F.conv2d(h, w["conv2d_1/kernel"].permute(3, 2, 0, 1).contiguous(), w["conv2d_1/bias"], padding=1))
    h = h.permute(0, 2, 3, 1).reshape(h.shape[0], -1)
    h = F.silu(h @ w["dense/kernel"] + w["dense/bias"])
    z = h @ w["latent_vector/kernel"] + w["latent_vector/bias"]
    h = F.silu(z @ w["dense_1/kernel"] + w["dense_1/bias"]).reshape(-1, 12, 12, 256).permute(0, 3, 1, 2)
    for name in DECODER_LAYERS[1:6]:
        h = F.silu(F.conv_transpose2d(h, w[f"{name}/kernel"].permute(3, 2, 0, 1).contiguous(), w[f"{name}/bias"], stride=2))
    pred = F.conv2d(h, w["output_image_400/kernel"].permute(3, 2, 0, 1).contiguous(), w["output_image_400/bias"], padding=1).permute(0, 2, 3, 1)
    loss = torch.mean((torch.tensor(np.asarray(y, np.float64)) - pred) ** 2)
    loss.backward()
    flat = np.concatenate([w[k].grad.numpy().reshape(-1) for k in flat_order(enc_w, dec_w)])
    return float(loss.item()), flat


def adam_reference(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam, float64."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    alpha = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return p - alpha * m / (np.sqrt(v) + eps), m, v
