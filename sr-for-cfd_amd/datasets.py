"""Training-data side of the SR path: the `simulation_result*.h5` reader, the
per-boundary-condition Reynolds split, component-wise standardisation and the stats-file writer,
as the training notebook does them (sr-ae-conv.ipynb:c16-113, c400-541, c589-603), on libsrcfd's
own HDF5 reader (no h5py).

File schema (sr-simulation-data-creation.ipynb cell 2, save_solution r262-285): one group per run,
`Re{Re}_mesh{n}x{n}`, attrs `bc_type` (str), `nx`, `ny`, ...; datasets `u`, `v`, `p` = flat float64
of length n*n (row-major (ny, nx)).
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np

from .h5 import H5File, H5Writer
from .stats import save_stats

COMPONENTS = ("u", "v", "p")
_GROUP = re.compile(r"^Re(\d+)_mesh(\d+)x(\d+)$")


def dataset_standardize(arr):
    """sr-ae-conv.ipynb:c111 -- float64 mean / population std over every pixel of every sample.  The
    arithmetic on the float32 array stays float32 (the notebook's NumPy 1.x value-based casting; Python
    floats keep it so under NumPy 2 as well)."""
    mean, std = float(np.mean(arr, dtype=np.float64)), float(np.std(arr, dtype=np.float64))
    std = 1e-8 if std == 0 else std
    return (arr - mean) / std, mean, std


def standardize_with_stats(arr, mean, std):
    """sr-ae-conv.ipynb:c112."""
    std = 1e-8 if std == 0 else std
    return (arr - mean) / std


def inverse_standardize(arr, mean, std):
    """sr-ae-conv.ipynb:c113."""
    return arr * std + mean


def avg_pool(x: np.ndarray, factor: int) -> np.ndarray:
    """`tf.nn.avg_pool(x, ksize=f, strides=f, 'VALID')` on (N,H,W,C) float32 (dummy-data recipe, c80)."""
    n, h, w, c = x.shape
    hh, ww = h // factor, w // factor
    return x[:, :hh * factor, :ww * factor].reshape(n, hh, factor, ww, factor, c).mean(axis=(2, 4), dtype=np.float32)


def dummy_pairs(lr_dim: int, hr_dim: int, n_per_component: int = 20, seed: int = 0):
    """The notebook's fallback when no file can be read (c72-91): x_hr ~ N(0,1), x_lr = avg_pool(x_hr)."""
    if hr_dim % lr_dim:
        raise ValueError("For dummy data, hr_dim must be a multiple of lr_dim.")
    rng = np.random.default_rng(seed)
    xs_lr, xs_hr, res, comps, bcs = [], [], [], [], []
    for comp in COMPONENTS:
        x_hr = rng.standard_normal((n_per_component, hr_dim, hr_dim, 1)).astype(np.float32)
        xs_hr.append(x_hr)
        xs_lr.append(avg_pool(x_hr, hr_dim // lr_dim))
        res.extend(np.arange(50, 50 * n_per_component + 1, 50))
        comps.extend([comp] * n_per_component)
        bcs.extend(["dummy"] * n_per_component)
    return np.concatenate(xs_lr), np.concatenate(xs_hr), np.array(res), np.array(comps), np.array(bcs)


def load_paired_reynolds_multi(file_paths: Iterable[str], lr_dim: int, hr_dim: int, verbose: bool = False, dummy_seed: int = 0):
    """sr-ae-conv.ipynb:c16-109.  For every file and every Reynolds number that has both the
    lr_dim and the hr_dim mesh, one (lr, hr) float32 pair per component.  Unreadable files are
    skipped; with nothing loaded the dummy recipe is returned, like the notebook.
    -> x_lr (N,lr,lr,1), x_hr (N,hr,hr,1), used_res (N,), components (N,), bc_types (N,)."""
    say = print if verbose else (lambda *a, **k: None)
    xs_lr: List[np.ndarray] = []
    xs_hr: List[np.ndarray] = []
    used_res: List[int] = []
    comps: List[str] = []
    bcs: List[str] = []
    for path in file_paths:
        try:
            with H5File(path) as f:
                keys = f.keys("/")
                if not keys:
                    say(f"{path}: empty, skipped")
                    continue
                res_in_file = sorted({int(m.group(1)) for m in map(_GROUP.match, keys) if m})
                first = keys[0]
                bc_type = f.attr_str(first, "bc_type")[0] if "bc_type" in f.attr_names(first) else "unknown"
                for Re in res_in_file:
                    g_lr, g_hr = f"Re{Re}_mesh{lr_dim}x{lr_dim}", f"Re{Re}_mesh{hr_dim}x{hr_dim}"
                    if g_lr in keys and g_hr in keys:
                        for comp in COMPONENTS:
                            if f"{g_lr}/{comp}" in f and f"{g_hr}/{comp}" in f:
                                xs_lr.append(f.read(f"{g_lr}/{comp}", np.float64).astype(np.float32).reshape(lr_dim, lr_dim))
                                xs_hr.append(f.read(f"{g_hr}/{comp}", np.float64).astype(np.float32).reshape(hr_dim, hr_dim))
                                used_res.append(Re)
                                comps.append(comp)
                                bcs.append(bc_type)
                say(f"{path}: Re {res_in_file}, bc_type {bc_type}")
        except (IOError, OSError, FileNotFoundError) as e:
            say(f"{path}: {e}; skipped")
            continue
    if not xs_lr:
        say("no data loaded from any file: dummy data")
        return dummy_pairs(lr_dim, hr_dim, seed=dummy_seed)
    return (np.array(xs_lr, dtype=np.float32)[..., None], np.array(xs_hr, dtype=np.float32)[..., None],
            np.array(used_res), np.array(comps), np.array(bcs))


ReList = Union[str, Sequence[int]]


def split_by_reynolds(used_res: np.ndarray, bc_types: np.ndarray, reynolds_config: Dict[str, Dict[str, ReList]]):
    """Per-boundary-condition train/test masks (c430-465); a list of Re or "ALL" per BC type.
    -> (train_mask, test_mask, reynolds_to_evaluate)."""
    train = np.zeros(len(used_res), dtype=bool)
    test = np.zeros(len(used_res), dtype=bool)
    evaluate: List[int] = []
    for bc, cfg in reynolds_config.items():
        m = bc_types == bc
        tr = np.unique(used_res[m]) if isinstance(cfg["train"], str) and cfg["train"] == "ALL" else cfg["train"]
        te = np.unique(used_res[m]) if isinstance(cfg["test"], str) and cfg["test"] == "ALL" else cfg["test"]
        train |= m & np.isin(used_res, tr)
        test |= m & np.isin(used_res, te)
        evaluate.extend(cfg.get("evaluate", []))
    return train, test, sorted(set(evaluate))


def component_standardize(x_lr_raw: np.ndarray, x_hr_raw: np.ndarray, comps: np.ndarray, stats=None):
    """Component-wise standardisation (c486-541).  stats=None: compute them from these arrays (training set);
    otherwise apply the given (stats_lr, stats_hr).  -> x_lr, x_hr, stats_lr, stats_hr."""
    x_lr = np.zeros_like(x_lr_raw)
    x_hr = np.zeros_like(x_hr_raw)
    stats_lr: Dict[str, Tuple[float, float]] = {}
    stats_hr: Dict[str, Tuple[float, float]] = {}
    for c in COMPONENTS:
        m = comps == c
        if stats is None:
            if not m.any():
                stats_lr[c] = stats_hr[c] = (0.0, 1.0)
                continue
            x_lr[m], ml, sl = dataset_standardize(x_lr_raw[m])
            x_hr[m], mh, sh = dataset_standardize(x_hr_raw[m])
            stats_lr[c], stats_hr[c] = (ml, sl), (mh, sh)
        else:
            stats_lr[c], stats_hr[c] = stats[0][c], stats[1][c]
            x_lr[m] = standardize_with_stats(x_lr_raw[m], *stats_lr[c])
            x_hr[m] = standardize_with_stats(x_hr_raw[m], *stats_hr[c])
    return x_lr, x_hr, stats_lr, stats_hr


def save_component_stats(path, lr_dim: int, hr_dim: int, stats_lr, stats_hr) -> None:
    """`standardization_stats_{lr}to{hr}_{suffix}.txt` (c589-603)."""
    save_stats(path, lr_dim, hr_dim, stats_lr, stats_hr)


def append_solution(writer: H5Writer, Re: int, n: int, fields: Dict[str, np.ndarray], bc_type: str, case_name: str = "") -> None:
    """One `Re{Re}_mesh{n}x{n}` group in the data-creation notebook's schema (save_solution, r262-285)."""
    g = f"Re{Re}_mesh{n}x{n}"
    writer.group(g)
    writer.attr(g, "bc_type", bc_type)
    if case_name:
        writer.attr(g, "case_name", case_name)
    writer.attr(g, "reynolds_number", np.float64(Re))
    writer.attr(g, "nx", np.int64(n))
    writer.attr(g, "ny", np.int64(n))
    writer.attr(g, "total_points", np.int64(n * n))
    for c in COMPONENTS:
        writer.dataset(f"{g}/{c}", np.asarray(fields[c], np.float64).reshape(-1))


def prepare_training_set(file_paths: Iterable[str], lr_dim: int, hr_dim: int, reynolds_config=None, verbose: bool = False):
    """Everything between the files and `fit` (c430-541).  reynolds_config=None: every sample trains
    (what happens for the dummy recipe).  -> dict with standardised train/test arrays and the stats."""
    x_lr, x_hr, res, comps, bcs = load_paired_reynolds_multi(file_paths, lr_dim, hr_dim, verbose)
    if reynolds_config is None:
        train = np.ones(len(res), dtype=bool)
        test = np.zeros(len(res), dtype=bool)
        evaluate: List[int] = []
    else:
        train, test, evaluate = split_by_reynolds(res, bcs, reynolds_config)
    xl, xh, stats_lr, stats_hr = component_standardize(x_lr[train], x_hr[train], comps[train])
    tl, th, _, _ = component_standardize(x_lr[test], x_hr[test], comps[test], (stats_lr, stats_hr))
    return dict(x_lr_train=xl, x_hr_train=xh, res_train=res[train], comps_train=comps[train],
                x_lr_test=tl, x_hr_test=th, res_test=res[test], comps_test=comps[test],
                x_lr_test_raw=x_lr[test], x_hr_test_raw=x_hr[test],
                stats_lr=stats_lr, stats_hr=stats_hr, reynolds_to_evaluate=evaluate)
