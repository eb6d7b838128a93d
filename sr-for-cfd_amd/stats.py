"""Standardisation-stats text files (``key value`` lines, ``#`` comments).

Reader = PyCFD_ML_accelerated.py:787-809 (raises FileNotFoundError / KeyError
the same way); writer = sr-ae-conv.ipynb:c589-603.  Parsing runs in libsrcfd
(``srcfd_stats_load``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Tuple

from . import _lib as L

COMPONENTS = ("u", "v", "p")


_CACHE: Dict[tuple, tuple] = {}


def load_stats(path, lr_dim: int, hr_dim: int) -> Tuple[Dict[str, Tuple[float, float]], Dict[str, Tuple[float, float]]]:
    """-> (stats_lr, stats_hr), each {'u'|'v'|'p': (mean, std)}.  The solvers re-read the file on every SR call
    (PyCFD_ML_accelerated.py:787); an unchanged file (same size and mtime) is parsed once."""
    try:
        st = os.stat(path)
        key = (os.fspath(path), st.st_mtime_ns, st.st_size, int(lr_dim), int(hr_dim))
    except OSError:
        key = None  # let the library produce the FileNotFoundError with its message
    hit = _CACHE.get(key) if key else None
    if hit is None:
        out = (C.c_double * 12)()
        L.check(L.lib.srcfd_stats_load(L.enc(path), int(lr_dim), int(hr_dim), out))
        hit = tuple(out)
        if key:
            if len(_CACHE) > 64:
                _CACHE.clear()
            _CACHE[key] = hit
    lr = {c: (hit[i * 2], hit[i * 2 + 1]) for i, c in enumerate(COMPONENTS)}
    hr = {c: (hit[6 + i * 2], hit[6 + i * 2 + 1]) for i, c in enumerate(COMPONENTS)}
    return lr, hr


def save_stats(path, lr_dim: int, hr_dim: int, stats_lr, stats_hr) -> None:
    vals = []
    for st in (stats_lr, stats_hr):
        for c in COMPONENTS:
            vals += [float(st[c][0]), float(st[c][1])]
    L.check(L.lib.srcfd_stats_save(L.enc(path), int(lr_dim), int(hr_dim), (C.c_double * 12)(*vals)))
