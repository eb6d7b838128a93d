"""Second, independent CPU implementation of the SR network on torch CPU ops.

TEST INFRASTRUCTURE ONLY (see sr_oracle.py header; same import rule).

Purpose: (1) cross-check ``sr_oracle.py`` with code written by different
authors (``torch.nn.functional`` / oneDNN) -- SURVEY.md 8c "independent check";
(2) the ``cpu_baseline`` leg of ``bench.py``: oneDNN is the CPU convolution
backend family the reference's TensorFlow build uses, so this is the closest
runnable stand-in for "the reference CPU Keras path" (TensorFlow itself is not
installable here).

Keras -> torch layout conversions (sr-ae-conv.ipynb:c162-169, c277-287):
  Conv2D kernel (kh,kw,Cin,Cout)          -> conv2d weight (Cout,Cin,kh,kw)
  Conv2DTranspose kernel (kh,kw,Cout,Cin) -> conv_transpose2d weight (Cin,Cout,kh,kw)
  TF SAME, stride 2, 10->5                -> F.pad(x,(0,1,0,1)) then VALID
  Flatten is NHWC                         -> permute before reshape
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .sr_oracle import DECODER_LAYERS, same_padding


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


class TorchSR:
    """Holds converted weights; forward in NCHW internally, NHWC at the boundary."""

    def __init__(self, enc_w: Dict[str, np.ndarray], dec_w: Dict[str, np.ndarray],
                 dtype=torch.float32):
        self.dtype = dtype
        g = lambda w, k: _t(w[k], dtype)
        self.c1w = g(enc_w, "conv2d/kernel").permute(3, 2, 0, 1).contiguous()
        self.c1b = g(enc_w, "conv2d/bias")
        self.c2w = g(enc_w, "conv2d_1/kernel").permute(3, 2, 0, 1).contiguous()
        self.c2b = g(enc_w, "conv2d_1/bias")
        self.d1w = g(enc_w, "dense/kernel").t().contiguous()
        self.d1b = g(enc_w, "dense/bias")
        self.d2w = g(enc_w, "latent_vector/kernel").t().contiguous()
        self.d2b = g(enc_w, "latent_vector/bias")
        self.d3w = g(dec_w, "dense_1/kernel").t().contiguous()
        self.d3b = g(dec_w, "dense_1/bias")
        self.tw, self.tb = [], []
        for name in DECODER_LAYERS[1:6]:
            self.tw.append(g(dec_w, f"{name}/kernel").permute(3, 2, 0, 1).contiguous())
            self.tb.append(g(dec_w, f"{name}/bias"))
        self.ow = g(dec_w, "output_image_400/kernel").permute(3, 2, 0, 1).contiguous()
        self.ob = g(dec_w, "output_image_400/bias")

    @torch.no_grad()
    def encode(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        x = x_nhwc.permute(0, 3, 1, 2)
        _, pt, pb = same_padding(x.shape[2], 3, 2)
        _, pl, pr = same_padding(x.shape[3], 3, 2)
        x = F.silu(F.conv2d(F.pad(x, (pl, pr, pt, pb)), self.c1w, self.c1b, stride=2))
        x = F.silu(F.conv2d(x, self.c2w, self.c2b, padding=1))
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
        x = F.silu(F.linear(x, self.d1w, self.d1b))
        return F.linear(x, self.d2w, self.d2b)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        h = F.silu(F.linear(z, self.d3w, self.d3b))
        h = h.reshape(-1, 12, 12, 256).permute(0, 3, 1, 2)
        for w, b in zip(self.tw, self.tb):
            h = F.silu(F.conv_transpose2d(h, w, b, stride=2))
        y = F.conv2d(h, self.ow, self.ob, padding=1)
        return y.permute(0, 2, 3, 1)

    @torch.no_grad()
    def forward(self, x_nhwc: np.ndarray, batch_size: int = 32) -> np.ndarray:
        """Keras ``predict`` default batch_size=32."""
        x = _t(x_nhwc, self.dtype)
        outs = []
        for i in range(0, x.shape[0], batch_size):
            outs.append(self.decode(self.encode(x[i:i + batch_size])))
        return torch.cat(outs, 0).contiguous().numpy()
