"""Aspect-ratio resampling of the BFS path as two matrix products (bfs_ml_accelerated.py:59-145).

The reference calls `scipy.interpolate.RectBivariateSpline(y, x, field, kx=3, ky=3)(y_new, x_new)`
per component.  With s=0 that is the interpolating bicubic spline (not-a-knot in both directions),
linear in `field` and a tensor product, and FITPACK clamps evaluation points to the data interval,
so   out = Ry @ field @ Rx.T   with the 1-D matrices built here once per geometry (float64,
`make_interp_spline` on the identity: the same spline, agreeing with RectBivariateSpline to ~1e-15).
`Resampler` keeps the pair on the device; `SRModel.predict_resampled` applies it to the network
output before it leaves the GPU.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache
from typing import Tuple

import numpy as np

from . import _lib as L


@lru_cache(maxsize=64)
def interp_matrix(n_src: int, src_hi: float, n_dst: int, dst_hi: float) -> np.ndarray:
    """[n_dst][n_src] cubic-spline interpolation matrix from nodes linspace(0, src_hi, n_src) to points
    linspace(0, dst_hi, n_dst) (clamped to the source interval, as FITPACK's bispev does)."""
    from scipy.interpolate import make_interp_spline
    src = np.linspace(0.0, src_hi, n_src)
    dst = np.clip(np.linspace(0.0, dst_hi, n_dst), src[0], src[-1])
    R = make_interp_spline(src, np.eye(n_src), k=3)(dst)
    R.setflags(write=False)
    return R


def square_to_rect_matrices(n_sq: int, nx_rect: int, ny_rect: int, lx: float, ly: float) -> Tuple[np.ndarray, np.ndarray]:
    """(Ry, Rx) of `reshape_square_to_rectangular` (bfs_ml_accelerated.py:104-145)."""
    Lmax = max(lx, ly)
    return interp_matrix(n_sq, Lmax, ny_rect, ly), interp_matrix(n_sq, Lmax, nx_rect, lx)


def rect_to_square_matrices(nx_rect: int, ny_rect: int, lx: float, ly: float) -> Tuple[np.ndarray, np.ndarray]:
    """(Ry, Rx) of `reshape_rectangular_to_square` (bfs_ml_accelerated.py:59-101): target nx_rect x nx_rect."""
    Lmax = max(lx, ly)
    return interp_matrix(ny_rect, ly, nx_rect, Lmax), interp_matrix(nx_rect, lx, nx_rect, Lmax)


class Resampler:
    """Device-resident (Ry, Rx) pair: out[z] = Ry @ in[z] @ Rx.T in float64."""

    def __init__(self, Ry: np.ndarray, Rx: np.ndarray, device: int = 0):
        Ry = np.ascontiguousarray(Ry, np.float64)
        Rx = np.ascontiguousarray(Rx, np.float64)
        self.out_h, self.in_h = Ry.shape
        self.out_w, self.in_w = Rx.shape
        self.device = device
        self._h = C.c_void_p()
        L.check(L.lib.srcfd_resampler_create(device, Ry.ctypes.data_as(C.c_void_p), Rx.ctypes.data_as(C.c_void_p), self.in_h, self.in_w,
                                             self.out_h, self.out_w, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib.srcfd_resampler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def apply_device(self, x, out=None, stream=None):
        """x: float32 CUDA tensor (n,in_h,in_w) -> float64 CUDA tensor (n,out_h,out_w)."""
        import torch
        n = int(x.shape[0])
        if out is None:
            out = torch.empty((n, self.out_h, self.out_w), dtype=torch.float64, device=x.device)
        st = stream if stream is not None else torch.cuda.current_stream(x.device).cuda_stream
        L.check(L.lib.srcfd_resample_device(self._h, C.c_void_p(x.data_ptr()), n, C.c_void_p(out.data_ptr()), C.c_void_p(st)))
        return out


@lru_cache(maxsize=8)
def square_to_rect_resampler(n_sq: int, nx_rect: int, ny_rect: int, lx: float, ly: float, device: int) -> Resampler:
    Ry, Rx = square_to_rect_matrices(n_sq, nx_rect, ny_rect, lx, ly)
    return Resampler(Ry, Rx, device)
