"""Counterpart of the reference's `ml_super_resolution` (the one call the solvers make
between the coarse and the fine solve), on libsrcfd.

    PyCFD_ML_accelerated.py:764-879   ml_super_resolution(coarse_fields, lr_dim, hr_dim,
                                          stats_file, encoder_file, decoder_file)
    bfs_ml_accelerated.py:979-1137    same + use_aspect_ratio_correction, lx, ly,
                                          use_adaptive_normalization=True, blend_factor=0.3

Order of operations is the reference's: float32 cast, [BFS: spline resample to a
square], stats lookup, [BFS: adaptive blend], standardise, predict, de-standardise
with the *training* HR stats, NaN/Inf zero-fill, [BFS: resample back].  Standardise,
the network, de-standardise and the guard run as one device call for the three
components.  The BFS resample back to the rectangle (a bicubic spline fit of a 400x400
field per component in the reference) runs on the device as two float64 matrix products
(resample.py); the 10x10 statistics and the 10x10 pre-resampling stay on the host.
"""
from __future__ import annotations

import os
import warnings
from typing import Dict, Optional

import numpy as np

from . import keras_compat as kc
from .stats import load_stats

COMPONENTS = ("u", "v", "p")


def standardize_with_stats(arr, mean, std):
    """PyCFD_ML_accelerated.py:665-668."""
    std = 1e-8 if std == 0 else std
    return (arr - mean) / std


def inverse_standardize(arr, mean, std):
    """PyCFD_ML_accelerated.py:671-673."""
    return arr * std + mean


def reshape_rectangular_to_square(fields, nx_rect, ny_rect, lx, ly):
    """bfs_ml_accelerated.py:59-101 (bicubic RectBivariateSpline onto linspace(0,max(lx,ly))^2)."""
    from scipy import interpolate
    x_rect, y_rect = np.linspace(0, lx, nx_rect), np.linspace(0, ly, ny_rect)
    L = max(lx, ly)
    x_sq, y_sq = np.linspace(0, L, nx_rect), np.linspace(0, L, nx_rect)
    out = {}
    for c in COMPONENTS:
        out[c] = interpolate.RectBivariateSpline(y_rect, x_rect, fields[c], kx=3, ky=3)(y_sq, x_sq)
    return out


def reshape_square_to_rectangular(fields, nx_rect, ny_rect, lx, ly):
    """bfs_ml_accelerated.py:104-145."""
    from scipy import interpolate
    n = fields["u"].shape[0]
    L = max(lx, ly)
    x_sq, y_sq = np.linspace(0, L, n), np.linspace(0, L, n)
    x_rect, y_rect = np.linspace(0, lx, nx_rect), np.linspace(0, ly, ny_rect)
    out = {}
    for c in COMPONENTS:
        out[c] = interpolate.RectBivariateSpline(y_sq, x_sq, fields[c], kx=3, ky=3)(y_rect, x_rect)
    return out


def _prepare(coarse_fields, lr_dim, hr_dim, stats_file, encoder_file, decoder_file, use_aspect_ratio_correction, lx, ly,
             use_adaptive_normalization, blend_factor, precision, say):
    """Everything before `predict` (PyCFD...:787-857 / bfs...:1025-1108): model handle, the (3,lr,lr,1) float32
    batch, the per-component (mean,std) pairs in and out, and the resampler for the way back (or None)."""
    fields = coarse_fields
    if use_aspect_ratio_correction and lx != ly:
        # reshape_rectangular_to_square (bfs_ml_accelerated.py:59-101) as two cached 1-D spline matrices: the same
        # interpolating bicubic spline as scipy's RectBivariateSpline to ~1e-15 (tests/test_resample.py), without three
        # FITPACK fits per call
        from . import resample as rs
        Ry, Rx = rs.rect_to_square_matrices(lr_dim, lr_dim, float(lx), float(ly))
        fields = {c: Ry @ np.asarray(coarse_fields[c], np.float64) @ Rx.T for c in COMPONENTS}
    stats_lr, stats_hr = load_stats(stats_file, lr_dim, hr_dim)  # FileNotFoundError / KeyError like :819-825
    for f in (encoder_file, decoder_file):  # the callers pre-check this (:1080-1087); load_model would raise too
        if not os.path.exists(f):
            raise FileNotFoundError(f"model file '{f}' not found")
    model = kc._device_handle((os.fspath(encoder_file), os.fspath(decoder_file)), precision or kc._DEFAULT_PRECISION)
    x = np.empty((3, lr_dim, lr_dim, 1), np.float32)
    ain = np.empty((3, 2), np.float32)
    aout = np.empty((3, 2), np.float32)
    for i, c in enumerate(COMPONENTS):
        x[i, :, :, 0] = fields[c]  # == np.asarray(fields[c]).astype(np.float32)
    if use_adaptive_normalization:
        # np.mean / np.std of each component's float32 array (bfs_ml_accelerated.py:1091-1092), as two row reductions over
        # the stacked batch instead of six calls: same pairwise sums, bit-identical (tests/test_dropin.py)
        rows = x.reshape(3, -1)
        input_means, input_stds = rows.mean(axis=1), rows.std(axis=1)
    for i, c in enumerate(COMPONENTS):
        mean_lr, std_lr = stats_lr[c]
        if use_adaptive_normalization:  # bfs_ml_accelerated.py:1093-1097, same expressions
            input_mean, input_std = input_means[i], input_stds[i]
            mean_lr = (1 - blend_factor) * mean_lr + blend_factor * input_mean
            std_lr = (1 - blend_factor) * std_lr + blend_factor * max(input_std, 1e-8)
            if say is not _quiet:
                say(f"  {c.upper()}: adaptive norm (blend={blend_factor:.2f}) mean={float(mean_lr):.6f} std={float(std_lr):.6f}")
        ain[i] = (mean_lr, std_lr)
        aout[i] = stats_hr[c]
    back = None
    if use_aspect_ratio_correction and lx != ly:
        from . import resample as rs
        back = rs.square_to_rect_resampler(hr_dim, hr_dim, hr_dim, float(lx), float(ly), model.device)
    return model, x, ain, aout, back, fields


def _quiet(*a, **k):
    pass


def _warn_nonfinite(bad):
    if bad:
        warnings.warn(f"super-resolved fields contained {bad} NaN/Inf values; replaced with zeros "
                      "(PyCFD_ML_accelerated.py:869-876)", RuntimeWarning)


def ml_super_resolution(coarse_fields: Dict[str, np.ndarray], lr_dim: int, hr_dim: int,
                        stats_file: str, encoder_file: str, decoder_file: str,
                        use_aspect_ratio_correction: bool = False, lx: float = 1.0, ly: float = 1.0,
                        use_adaptive_normalization: bool = False, blend_factor: float = 0.3,
                        precision: Optional[str] = None, verbose: bool = False) -> Dict[str, np.ndarray]:
    """Lid-driven-cavity defaults (PyCFD_ML_accelerated.py:764); `ml_super_resolution_bfs`
    has the backward-facing-step defaults.  Returns {'u','v','p'} -> (hr_dim, hr_dim) float32
    (float64 after the BFS back-resampling, as scipy returns it)."""
    say = print if verbose else _quiet
    model, x, ain, aout, back, fields = _prepare(coarse_fields, lr_dim, hr_dim, stats_file, encoder_file, decoder_file,
                                                 use_aspect_ratio_correction, lx, ly, use_adaptive_normalization, blend_factor, precision, say)
    if back is not None:
        y, bad = model.predict_resampled(x, back, in_affine=ain, out_affine=aout, nan_guard=True, return_nonfinite=True)
        y = y[..., None]
    else:
        y, bad = model.predict(x, in_affine=ain, out_affine=aout, nan_guard=True, return_nonfinite=True)
    _warn_nonfinite(bad)
    hr = {c: y[i, :, :, 0] for i, c in enumerate(COMPONENTS)}
    if verbose:  # the range scan costs more than the device work: only when it is printed
        for c in COMPONENTS:
            say(f"  {c.upper()}: {fields[c].shape} -> {hr[c].shape}, range [{hr[c].min():.6f}, {hr[c].max():.6f}]")
    return hr


def bc_arrays(bc):
    """(3,4) int types and (3,4) float values, [left,right,top,bottom] for u, v, p, from the solvers'
    `BoundaryConditions` object (`_get_bc_arrays`, PyCFD_ML_accelerated.py:~350 / bfs_ml_accelerated.py:497-521) or from a
    plain {'u': {'left': ('dirichlet', 0.0), ...}, ...} dict."""
    types = np.zeros((3, 4), np.int32)
    values = np.zeros((3, 4), np.float64)
    for k, c in enumerate(COMPONENTS):
        d = getattr(bc, f"{c}_boundaries", None)
        if d is None:
            d = bc[c]
        for s, side in enumerate(("left", "right", "top", "bottom")):
            e = d[side]
            t, v = (e.type, e.value) if hasattr(e, "type") else e
            types[k, s] = 0 if t == "dirichlet" else 1
            values[k, s] = v
    return types, values


def bfs_inlet_profiles(ny: int, dy: float, step_height: float, h: float, Ub: float) -> Dict[int, np.ndarray]:
    """Row-wise Dirichlet values of the BFS left boundary (bfs_ml_accelerated.py:524-562): wall (0) below the step,
    parabolic U = 6 Ub (y'/h)(1 - y'/h) above it with y' clamped to [0,h]; V = 0 everywhere."""
    y = (np.arange(1, ny + 1) - 0.5) * dy
    yp = np.clip(y - step_height, 0.0, h)
    u = np.where(y < step_height, 0.0, 6.0 * Ub * (yp / h) * (1.0 - (yp / h)))
    return {0: u, 1: np.zeros(ny)}


def ml_super_resolution_into_solver(coarse_fields, lr_dim: int, hr_dim: int, stats_file: str, encoder_file: str, decoder_file: str, bc,
                                    Var: Optional[np.ndarray] = None, left_profiles=None,
                                    use_aspect_ratio_correction: bool = False, lx: float = 1.0, ly: float = 1.0,
                                    use_adaptive_normalization: bool = False, blend_factor: float = 0.3,
                                    precision: Optional[str] = None) -> np.ndarray:
    """`ml_super_resolution` + the hand-off that follows it in the solvers (PyCFD_ML_accelerated.py:936-943,
    bfs_ml_accelerated.py:1211-1218) as one device pass: returns/fills the float64 `Var (3, nx+2, ny+2)` with the SR
    fields transposed into the interior and the ghost cells set from `bc`."""
    model, x, ain, aout, back, _ = _prepare(coarse_fields, lr_dim, hr_dim, stats_file, encoder_file, decoder_file,
                                            use_aspect_ratio_correction, lx, ly, use_adaptive_normalization, blend_factor, precision,
                                            _quiet)
    types, values = bc if isinstance(bc, tuple) else bc_arrays(bc)
    Var, bad = model.predict_into_solver_state(x, types, values, left_profiles=left_profiles, resampler=back, in_affine=ain, out_affine=aout,
                                               nan_guard=True, Var=Var, return_nonfinite=True)
    _warn_nonfinite(bad)
    return Var


def ml_super_resolution_batch(coarse_batch, lr_dim: int, hr_dim: int, stats_file: str, encoder_file: str, decoder_file: str,
                              use_aspect_ratio_correction: bool = False, lx: float = 1.0, ly: float = 1.0,
                              use_adaptive_normalization: bool = False, blend_factor: float = 0.3,
                              precision: Optional[str] = None, return_device: bool = False):
    """`ml_super_resolution` for a LIST of coarse fields in one device pass (the batched form of the BFS / LDC call; the
    reference makes one call per field).  Everything between the float64 coarse fields and the result runs on the GPU:
    the 10x10 aspect-ratio resampling and the adaptive blend of the statistics (`srcfd_prepare_inputs_device`,
    bfs_ml_accelerated.py:59-101, 1086-1097), the network with its fused standardise / de-standardise / NaN guard, and the
    resampling back to the rectangle (bfs_ml_accelerated.py:104-145).  Returns a list of {'u','v','p'} dicts like the
    per-field call (float32 (hr,hr); float64 after the BFS back-resampling), or with return_device=True the
    (N, 3, hr, hr) CUDA tensor."""
    import ctypes as C
    import torch
    from . import _lib as L
    from . import resample as rs
    n_f = len(coarse_batch)
    stats_lr, stats_hr = load_stats(stats_file, lr_dim, hr_dim)
    for f in (encoder_file, decoder_file):
        if not os.path.exists(f):
            raise FileNotFoundError(f"model file '{f}' not found")
    model = kc._device_handle((os.fspath(encoder_file), os.fspath(decoder_file)), precision or kc._DEFAULT_PRECISION)
    dev = torch.device("cuda", model.device)
    with torch.cuda.device(dev):     # the preparation kernel is launched on this device's current stream
        return _super_resolution_batch_on_device(coarse_batch, n_f, model, dev, stats_lr, stats_hr, lr_dim, hr_dim, use_aspect_ratio_correction,
                                                 lx, ly, use_adaptive_normalization, blend_factor, return_device)


def _super_resolution_batch_on_device(coarse_batch, n_f, model, dev, stats_lr, stats_hr, lr_dim, hr_dim, use_aspect_ratio_correction, lx, ly,
                                      use_adaptive_normalization, blend_factor, return_device):
    import ctypes as C
    import torch
    from . import _lib as L
    from . import resample as rs
    st = torch.cuda.current_stream(dev)
    h, w = np.asarray(coarse_batch[0]["u"]).shape
    fields = np.stack([np.stack([np.asarray(cf[c], np.float64) for c in COMPONENTS]) for cf in coarse_batch]).reshape(3 * n_f, h, w)
    f_dev = torch.from_numpy(np.ascontiguousarray(fields)).to(dev)
    train = torch.from_numpy(np.tile(np.array([stats_lr[c] for c in COMPONENTS], np.float64), (n_f, 1))).to(dev)
    aout = torch.from_numpy(np.tile(np.array([stats_hr[c] for c in COMPONENTS], np.float32), (n_f, 1))).to(dev)
    resample = use_aspect_ratio_correction and lx != ly
    Ry = Rx = None
    if resample:
        ry, rx = rs.rect_to_square_matrices(w, h, float(lx), float(ly))
        Ry, Rx = torch.from_numpy(np.array(ry)).to(dev), torch.from_numpy(np.array(rx)).to(dev)
    side = lr_dim if resample else h
    x = torch.empty((3 * n_f, side, side, 1), dtype=torch.float32, device=dev)
    ain = torch.empty((3 * n_f, 2), dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    L.check(L.lib.srcfd_prepare_inputs_device(p(f_dev), 3 * n_f, h, w, p(Ry), p(Rx), lr_dim, p(train), int(bool(use_adaptive_normalization)),
                                              float(blend_factor), p(x), p(ain), C.c_void_p(st.cuda_stream)))
    y = torch.empty((3 * n_f, hr_dim, hr_dim, 1), dtype=torch.float32, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    model.predict_device(x, y, in_affine=ain, out_affine=aout, nan_guard=True, nonfinite=bad)
    if resample:
        back = rs.square_to_rect_resampler(hr_dim, hr_dim, hr_dim, float(lx), float(ly), model.device)
        out = back.apply_device(y.view(3 * n_f, hr_dim, hr_dim))
    else:
        out = y.view(3 * n_f, hr_dim, hr_dim)
    _warn_nonfinite(int(bad.item()))
    out = out.view(n_f, 3, out.shape[-2], out.shape[-1])
    if return_device:
        return out
    host = out.cpu().numpy()
    return [{c: host[i, k] for k, c in enumerate(COMPONENTS)} for i in range(n_f)]


def ml_super_resolution_bfs(coarse_fields, lr_dim, hr_dim, stats_file, encoder_file, decoder_file,
                            use_aspect_ratio_correction: bool = False, lx: float = 1.0, ly: float = 1.0,
                            use_adaptive_normalization: bool = True, blend_factor: float = 0.3, **kw):
    """bfs_ml_accelerated.py:979-985 defaults."""
    return ml_super_resolution(coarse_fields, lr_dim, hr_dim, stats_file, encoder_file, decoder_file,
                               use_aspect_ratio_correction, lx, ly, use_adaptive_normalization, blend_factor, **kw)


def inject_into_solver_state(hr_fields: Dict[str, np.ndarray], Var: np.ndarray) -> None:
    """The consumer side of the boundary (PyCFD_ML_accelerated.py:936-938): write the SR
    fields, transposed, into the interior of the solver's float64 `Var[k, 1:-1, 1:-1]`."""
    for k, c in enumerate(COMPONENTS):
        Var[k, 1:-1, 1:-1] = hr_fields[c].T


def tiled_super_resolution(field: np.ndarray, model, lr_dim: int = 10, in_affine=None, out_affine=None, distributed: bool = False) -> np.ndarray:
    """BASELINE config 5: a (T*lr, T*lr, C) coarse field is cut into T x T non-overlapping
    lr x lr tiles, each component of each tile super-resolved independently with the same
    weights, and the 400x400 results stitched to (T*400, T*400, C).  No overlap or blending:
    the reference defines none.

    distributed=True (inside an initialised `torch.distributed` group, one process per GPU): every rank
    super-resolves its contiguous block of the T*T*C tile samples (shard.shard_range) and the blocks are
    exchanged with one all_gather, so each rank returns the whole stitched field.  The tiles are independent:
    there is no other collective."""
    H, W, C = field.shape
    ty, tx = H // lr_dim, W // lr_dim
    if ty * lr_dim != H or tx * lr_dim != W:
        raise ValueError("field size must be a multiple of the tile size")
    tiles = field.reshape(ty, lr_dim, tx, lr_dim, C).transpose(0, 2, 4, 1, 3).reshape(ty * tx * C, lr_dim, lr_dim, 1)
    tiles = np.ascontiguousarray(tiles, dtype=np.float32)
    n = tiles.shape[0]
    ai = np.tile(np.asarray(in_affine, np.float32), (ty * tx, 1)) if in_affine is not None else None
    ao = np.tile(np.asarray(out_affine, np.float32), (ty * tx, 1)) if out_affine is not None else None
    if not distributed:
        y = model.predict(tiles, in_affine=ai, out_affine=ao)
    else:
        import torch
        import torch.distributed as dist
        from .shard import shard_range
        rank, world = dist.get_rank(), dist.get_world_size()
        lo, hi = shard_range(n, rank, world)
        y_loc = model.predict(tiles[lo:hi], in_affine=None if ai is None else ai[lo:hi], out_affine=None if ao is None else ao[lo:hi])
        hr = y_loc.shape[1] if hi > lo else None
        # blocks differ by at most one sample: pad to the largest, gather, cut
        per = -(-n // world)
        shape = torch.tensor([0 if hr is None else hr], dtype=torch.int64)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        shape = shape.to(dev)
        dist.all_reduce(shape, op=dist.ReduceOp.MAX)
        hr = int(shape.item())
        buf = torch.zeros((per, hr, hr, 1), dtype=torch.float32, device=dev)
        if hi > lo:
            buf[:hi - lo] = torch.from_numpy(np.ascontiguousarray(y_loc)).to(dev)
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        y = np.concatenate([parts[r][:shard_range(n, r, world)[1] - shard_range(n, r, world)[0]].cpu().numpy() for r in range(world)])
    hr = y.shape[1]
    return y.reshape(ty, tx, C, hr, hr).transpose(0, 3, 1, 4, 2).reshape(ty * hr, tx * hr, C)
