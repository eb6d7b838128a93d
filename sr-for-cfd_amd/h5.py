"""Thin h5py-like reader/writer over libsrcfd's HDF5 subset.

Covers what the hot path touches: legacy Keras-H5 weight files
(PyCFD_ML_accelerated.py:831-832) and the solver's field dumps
(PyCFD_ML_accelerated.py:517-544, groups ``Re{Re}_mesh{n}x{n}`` with flat
float64 ``u,v,p,x,y``).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List

import numpy as np

from . import _lib as L

_NP = {L.F32: np.float32, L.F64: np.float64, L.I32: np.int32, L.I64: np.int64, L.U8: np.uint8}
_CODE = {np.dtype(v): k for k, v in _NP.items()}


def _string_call(fn, *args) -> str:
    need = C.c_size_t(0)
    L.check(fn(*args, None, 0, C.byref(need)))
    buf = C.create_string_buffer(max(need.value, 1))
    L.check(fn(*args, buf, len(buf), C.byref(need)))
    return buf.value.decode("utf-8", "replace")


class H5File:
    """Read-only view of an HDF5 file (libver-earliest subset)."""

    def __init__(self, path):
        self._h = C.c_void_p()
        L.check(L.lib.srcfd_h5_open(L.enc(path), C.byref(self._h)))

    def close(self):
        if self._h:
            L.lib.srcfd_h5_close(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def kind(self, path: str) -> str:
        return {0: "absent", 1: "group", 2: "dataset"}[L.lib.srcfd_h5_kind(self._h, path.encode())]

    def __contains__(self, path: str) -> bool:
        return self.kind(path) != "absent"

    def keys(self, group: str = "/") -> List[str]:
        s = _string_call(L.lib.srcfd_h5_list, self._h, group.encode())
        return s.split("\n") if s else []

    def attr_names(self, obj: str = "/") -> List[str]:
        s = _string_call(L.lib.srcfd_h5_attr_names, self._h, obj.encode())
        return s.split("\n") if s else []

    def attr_str(self, obj: str, name: str) -> List[str]:
        s = _string_call(L.lib.srcfd_h5_attr_string, self._h, obj.encode(), name.encode())
        return s.split("\n") if s else []

    def attr_num(self, obj: str, name: str) -> np.ndarray:
        cnt = C.c_int(0)
        L.check(L.lib.srcfd_h5_attr_numeric(self._h, obj.encode(), name.encode(), None, 0, C.byref(cnt)))
        out = np.zeros(max(cnt.value, 1), dtype=np.float64)
        L.check(L.lib.srcfd_h5_attr_numeric(self._h, obj.encode(), name.encode(),
                                            out.ctypes.data_as(C.POINTER(C.c_double)), len(out), C.byref(cnt)))
        return out[:cnt.value]

    def shape_dtype(self, path: str):
        dt, rank = C.c_int(), C.c_int()
        dims = (C.c_uint64 * 8)()
        L.check(L.lib.srcfd_h5_dataset_info(self._h, path.encode(), C.byref(dt), C.byref(rank), dims))
        return tuple(int(dims[i]) for i in range(rank.value)), _NP.get(dt.value)

    def read(self, path: str, dtype=None) -> np.ndarray:
        shape, native = self.shape_dtype(path)
        if native is None:
            raise OSError(f"dataset '{path}' has an unsupported element type")
        dtype = np.dtype(dtype or native)
        out = np.empty(shape, dtype=dtype)
        if out.size:
            L.check(L.lib.srcfd_h5_read(self._h, path.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes, _CODE[dtype]))
        return out

    __getitem__ = read


class H5Writer:
    """Builds a libver-earliest HDF5 file in memory, then saves it."""

    def __init__(self):
        self._h = C.c_void_p()
        L.check(L.lib.srcfd_h5w_create(C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                L.lib.srcfd_h5w_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def group(self, path: str):
        L.check(L.lib.srcfd_h5w_group(self._h, path.encode()))

    def dataset(self, path: str, data):
        a = np.ascontiguousarray(data)
        if a.dtype not in _CODE:
            raise ValueError(f"unsupported dtype {a.dtype}")
        dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
        L.check(L.lib.srcfd_h5w_dataset(self._h, path.encode(), _CODE[a.dtype], a.ndim, dims, a.ctypes.data_as(C.c_void_p)))

    def attr(self, obj: str, name: str, value, utf8: bool = False):
        if isinstance(value, str):
            arr = (C.c_char_p * 1)(value.encode())
            L.check(L.lib.srcfd_h5w_attr_strings(self._h, obj.encode(), name.encode(), arr, 1, 1, int(utf8)))
        elif isinstance(value, (list, tuple)) and all(isinstance(v, str) for v in value):
            arr = (C.c_char_p * max(len(value), 1))(*[v.encode() for v in value])
            L.check(L.lib.srcfd_h5w_attr_strings(self._h, obj.encode(), name.encode(), arr, len(value), 0, int(utf8)))
        else:
            a = np.ascontiguousarray(value)
            if a.dtype.kind == "i" and a.dtype != np.int32:
                a = a.astype(np.int64)
            if a.dtype.kind == "f" and a.dtype != np.float32:
                a = a.astype(np.float64)
            L.check(L.lib.srcfd_h5w_attr_numeric(self._h, obj.encode(), name.encode(), _CODE[a.dtype],
                                                 a.ctypes.data_as(C.c_void_p), a.size, int(a.ndim == 0)))

    def save(self, path):
        L.check(L.lib.srcfd_h5w_save(self._h, L.enc(path)))


def read_coarse_fields(path) -> Dict[str, np.ndarray]:
    """Coarse-solution file written by the solvers (PyCFD_ML_accelerated.py:517-544):
    one group ``Re{Re}_mesh{nx}x{ny}`` holding flat float64 ``u,v,p`` (x fastest?
    no: flattened from the (ny,nx) arrays of :755-759).  Returns (n,n) arrays."""
    with H5File(path) as f:
        groups = [k for k in f.keys("/") if f.kind(k) == "group"]
        if not groups:
            raise KeyError("no solution group in " + str(path))
        g = groups[0]
        out = {}
        nx = int(f.attr_num(g, "nx")[0]) if "nx" in f.attr_names(g) else None
        for c in ("u", "v", "p"):
            a = f.read(f"{g}/{c}", np.float64)
            n = nx or int(round(a.size ** 0.5))
            out[c] = a.reshape(a.size // n, n)
        return out
