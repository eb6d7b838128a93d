"""Coarse-mesh solves: `run_coarse_simulation` of the reference's lid-driven-cavity solver (PyCFD_ML_accelerated.py:694-761)
and of its backward-facing-step solver (bfs_ml_accelerated.py:893-976, `run_bfs_coarse_simulation` here) on libsrcfd's host
solver (csrc/coarse_solver.cpp, C ABI `srcfd_coarse_solve`).

The SR hot path starts from a converged 10x10 coarse field.  The reference checkout holds such fields for Re = 800 and
1000 only (tests/golden/coarse_ldc_*.h5); BASELINE config 1 names Re = 400, so the field has to be produced here.
No plots, no timestamped directory: the fields come back as the dict `ml_super_resolution` takes; `save` writes the
reference's own HDF5 layout (group `Re{Re}_mesh{nx}x{ny}`, flat float64 datasets, PyCFD...:517-544).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from . import _lib as L
from .h5 import H5Writer

SIDES = ("left", "right", "top", "bottom")

# BoundaryConditions() defaults (PyCFD_ML_accelerated.py:45-66): single moving lid
LDC_SINGLE_LID = {
    "u": {"left": ("dirichlet", 0.0), "right": ("dirichlet", 0.0), "top": ("dirichlet", 1.0), "bottom": ("dirichlet", 0.0)},
    "v": {s: ("dirichlet", 0.0) for s in SIDES},
    "p": {s: ("neumann", 0.0) for s in SIDES},
}
# the custom case the reference's __main__ runs (PyCFD_ML_accelerated.py:1385-1405): lid and floor both move with u = 1
LDC_DOUBLE_LID = {
    "u": {"left": ("dirichlet", 0.0), "right": ("dirichlet", 0.0), "top": ("dirichlet", 1.0), "bottom": ("dirichlet", 1.0)},
    "v": {s: ("dirichlet", 0.0) for s in SIDES},
    "p": {s: ("neumann", 0.0) for s in SIDES},
}
# the backward-facing-step case of the reference's __main__ (bfs_ml_accelerated.py:1789-1812): the left boundary is a
# placeholder (the solver overrides it with the wall / parabolic-inlet mix), pressure outlet on the right, no-slip walls
BFS_DEFAULT = {
    "u": {"left": ("dirichlet", 0.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
    "v": {"left": ("dirichlet", 0.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
    "p": {"left": ("neumann", 0.0), "right": ("dirichlet", 0.0), "top": ("neumann", 0.0), "bottom": ("neumann", 0.0)},
}
# `BoundaryConditions()` of the BFS file plus the three overrides run_coarse_simulation applies when bc is None
# (bfs_ml_accelerated.py:158-178, 945-951): u_left stays dirichlet 1.0 -- irrelevant, the inlet override rewrites that side
BFS_RUN_COARSE_DEFAULT = {
    "u": {"left": ("dirichlet", 1.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
    "v": {"left": ("dirichlet", 0.0), "right": ("neumann", 0.0), "top": ("dirichlet", 0.0), "bottom": ("dirichlet", 0.0)},
    "p": {"left": ("neumann", 0.0), "right": ("dirichlet", 0.0), "top": ("neumann", 0.0), "bottom": ("neumann", 0.0)},
}


def _bc_entry(e) -> Tuple[str, float]:
    return (e.type, float(e.value)) if hasattr(e, "type") else (e[0], float(e[1]))


def _bc_dicts(bc, default=None):
    if bc is None:
        return default if default is not None else LDC_SINGLE_LID
    if hasattr(bc, "u_boundaries"):      # the solvers' BoundaryConditions object
        return {"u": bc.u_boundaries, "v": bc.v_boundaries, "p": bc.p_boundaries}
    return bc


def solve_coarse(Re: float, nx: int = 10, ny: int = 10, lx: float = 1.0, ly: float = 1.0, dt: float = 0.001, scheme: str = "QUICK",
                 convergence_criteria: Optional[Dict[str, float]] = None, max_iterations: int = 100000, bc=None, rho: float = 1.0,
                 bfs: Optional[Dict[str, float]] = None, relaxation_factors: Optional[Dict[str, float]] = None):
    """Returns (Var (3, nx+2, ny+2) float64, iterations, rms residuals) of the converged (or capped) solve.
    `bfs` = {'step_height', 'h', 'Ub'} selects the backward-facing-step loop (inlet override + under-relaxation with
    `relaxation_factors`, default u 0.5 / v 0.5 / p 0.2 as SolverSettings, bfs_ml_accelerated.py:222-229)."""
    if scheme not in ("QUICK", "UPWIND"):
        raise ValueError(f"scheme must be 'QUICK' or 'UPWIND', not {scheme!r}")
    cc = {"u": 1e-6, "v": 1e-6, "p": 1e-6}
    cc.update(convergence_criteria or {})
    pb = L.CoarseProblem()
    pb.nx, pb.ny, pb.lx, pb.ly = int(nx), int(ny), float(lx), float(ly)
    pb.reynolds, pb.rho, pb.dt = float(Re), float(rho), float(dt)
    pb.scheme = 0 if scheme == "QUICK" else 1
    pb.max_iterations = int(max_iterations)
    for k, c in enumerate("uvp"):
        pb.tolerance[k] = float(cc[c])
        d = _bc_dicts(bc, BFS_RUN_COARSE_DEFAULT if bfs is not None else None)[c]
        for s_, side in enumerate(SIDES):
            t, v = _bc_entry(d[side])
            pb.bc_type[k][s_] = 0 if t == "dirichlet" else 1
            pb.bc_value[k][s_] = v
    if bfs is not None:
        rf = {"u": 0.5, "v": 0.5, "p": 0.2} if relaxation_factors is None else relaxation_factors
        pb.case_type = 1
        for k, c in enumerate("uvp"):
            pb.relax[k] = float(rf.get(c, 0.2 if c == "p" else 0.5))     # `.get(name, default)` as _implicit_solve does
        pb.step_height, pb.channel_height, pb.bulk_velocity = float(bfs["step_height"]), float(bfs["h"]), float(bfs["Ub"])
    var = np.zeros((3, nx + 2, ny + 2), np.float64)
    it = C.c_int(0)
    rms = (C.c_double * 3)()
    try:
        L.check(L.lib.srcfd_coarse_solve(C.byref(pb), var.ctypes.data_as(C.c_void_p), C.byref(it), rms))
    except (ValueError, L.SrcfdError) as e:
        if "NaN" in str(e):
            raise ValueError("Solver failed: NaN/Inf in residuals") from e    # what the reference raises (PyCFD...:487-492)
        raise
    return var, int(it.value), np.array(list(rms))


def run_coarse_simulation(Re: float, lr_dim: int = 10, dt: float = 0.001, scheme: str = "QUICK",
                          convergence_criteria: Optional[Dict[str, float]] = None, max_iterations: int = 100000,
                          output_dir: Optional[str] = None, bc=None) -> Dict[str, np.ndarray]:
    """Same signature and return value as the reference's function: {'u','v','p'} arrays of shape (lr_dim, lr_dim),
    Var[k, 1:-1, 1:-1].T (PyCFD_ML_accelerated.py:755-759).  With `output_dir` the result is also saved under the
    reference's file name."""
    var, it, _ = solve_coarse(Re, lr_dim, lr_dim, 1.0, 1.0, dt, scheme, convergence_criteria, max_iterations, bc)
    fields = {c: var[k, 1:-1, 1:-1].T.copy() for k, c in enumerate("uvp")}
    if output_dir is not None:
        import os
        os.makedirs(output_dir, exist_ok=True)
        save_coarse_fields(os.path.join(output_dir, f"coarse_Re{Re}_{lr_dim}x{lr_dim}_{max_iterations}_coarse_iterations.h5"), fields, Re)
    return fields


def run_bfs_coarse_simulation(Re: float, lr_dim: int = 10, dt: float = 0.002, scheme: str = "UPWIND",
                              convergence_criteria: Optional[Dict[str, float]] = None, max_iterations: int = 100000,
                              output_dir: Optional[str] = None, bc=None, step_height: float = 1.0, h: float = 2.0, Ub: float = 1.0,
                              lx: float = 10.0, ly: float = 3.0,
                              relaxation_factors: Optional[Dict[str, float]] = None) -> Dict[str, np.ndarray]:
    """`run_coarse_simulation` of bfs_ml_accelerated.py:893-976 (same arguments, defaults and return value): the coarse
    backward-facing-step field that `ml_super_resolution(..., use_aspect_ratio_correction=True)` starts from."""
    var, it, _ = solve_coarse(Re, lr_dim, lr_dim, lx, ly, dt, scheme, convergence_criteria, max_iterations, bc,
                              bfs={"step_height": step_height, "h": h, "Ub": Ub}, relaxation_factors=relaxation_factors)
    fields = {c: var[k, 1:-1, 1:-1].T.copy() for k, c in enumerate("uvp")}
    if output_dir is not None:
        import os
        os.makedirs(output_dir, exist_ok=True)
        save_coarse_fields(os.path.join(output_dir, f"bfs_coarse_Re{Re}_{lr_dim}x{lr_dim}_{max_iterations}_coarse_iterations.h5"), fields, Re,
                           lx, ly, bfs_step_height=step_height)
    return fields


def save_coarse_fields(path: str, fields: Dict[str, np.ndarray], Re: float, lx: float = 1.0, ly: float = 1.0,
                       bfs_step_height: Optional[float] = None) -> None:
    """`CFDSolver._save_results_hdf5` (PyCFD_ML_accelerated.py:517-544; bfs_ml_accelerated.py:726-757 with its extra
    lx / ly / step_height attributes): group Re{Re}_mesh{nx}x{ny} with flat x, y, u, v, p."""
    ny, nx = fields["u"].shape
    grp = f"Re{Re}_mesh{nx}x{ny}"
    w = H5Writer()
    w.group(grp)
    w.attr(grp, "case_name", "lid driven cavity" if bfs_step_height is None else "backward facing step")
    w.attr(grp, "reynolds_number", float(Re))
    w.attr(grp, "nx", int(nx))
    w.attr(grp, "ny", int(ny))
    if bfs_step_height is not None:
        w.attr(grp, "lx", float(lx))
        w.attr(grp, "ly", float(ly))
        w.attr(grp, "step_height", float(bfs_step_height))
    w.attr(grp, "total_points", int(nx * ny))
    X, Y = np.meshgrid(np.linspace(0, lx, nx), np.linspace(0, ly, ny))
    w.dataset(f"{grp}/x", X.flatten())
    w.dataset(f"{grp}/y", Y.flatten())
    for c in "uvp":
        w.dataset(f"{grp}/{c}", np.ascontiguousarray(fields[c], np.float64).flatten())
    w.save(path)
