"""Python handle over ``srcfd_model`` (include/srcfd.h).

``SRModel`` is the host-side object the Keras-compatible surface
(``keras_compat.py``) and the solver harness (``pipeline.py``) sit on.  It
replaces what ``tf.keras.models.load_model`` + ``SuperResolutionAE`` +
``Model.predict`` do in PyCFD_ML_accelerated.py:831-833,858.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib as L

PRECISIONS = {"fp32": L.PREC_FP32, "float32": L.PREC_FP32, "bf16": L.PREC_BF16, "bfloat16": L.PREC_BF16,
              "fp32_naive": L.PREC_FP32_NAIVE, "f16": L.PREC_F16, "fp16": L.PREC_F16, "float16": L.PREC_F16,
              "fp32x3": L.PREC_FP32X3}   # f32-grade: the wide decoder GEMMs as six bf16 MFMAs on exactly split operands (csrc/kernels_x3.hip)
_ACT = {"linear": L.ACT_LINEAR, None: L.ACT_LINEAR, "swish": L.ACT_SWISH, "silu": L.ACT_SWISH, "relu": L.ACT_RELU,
        "sigmoid": L.ACT_SIGMOID, "tanh": L.ACT_TANH}
_KIND = {"conv2d": L.LAYER_CONV2D, "conv2d_transpose": L.LAYER_CONV2D_TRANSPOSE, "dense": L.LAYER_DENSE,
         "flatten": L.LAYER_FLATTEN, "reshape": L.LAYER_RESHAPE}


def device_count() -> int:
    return int(L.lib.srcfd_device_count())


def layers_from_weights(enc_w: Optional[Dict[str, np.ndarray]], dec_w: Optional[Dict[str, np.ndarray]]) -> List[dict]:
    """Layer specs of encoder_10 and/or decoder_400 (sr-ae-conv.ipynb:c162-169,
    c277-287) from ``{'<layer>/kernel', '<layer>/bias'}`` dicts."""
    specs: List[dict] = []
    if enc_w is not None:
        specs += [
            dict(kind="conv2d", name="conv2d", k=3, stride=2, same=True, act="swish", w=enc_w["conv2d/kernel"], b=enc_w["conv2d/bias"]),
            dict(kind="conv2d", name="conv2d_1", k=3, stride=1, same=True, act="swish", w=enc_w["conv2d_1/kernel"], b=enc_w["conv2d_1/bias"]),
            dict(kind="flatten", name="flatten"),
            dict(kind="dense", name="dense", act="swish", w=enc_w["dense/kernel"], b=enc_w["dense/bias"]),
            dict(kind="dense", name="latent_vector", act="linear", w=enc_w["latent_vector/kernel"], b=enc_w["latent_vector/bias"]),
        ]
    if dec_w is not None:
        specs += [dict(kind="dense", name="dense_1", act="swish", w=dec_w["dense_1/kernel"], b=dec_w["dense_1/bias"]),
                  dict(kind="reshape", name="reshape", shape=(12, 12, 256))]
        for i, k in enumerate((3, 2, 2, 2, 2)):
            name = "conv2d_transpose" + ("" if i == 0 else f"_{i}")
            specs.append(dict(kind="conv2d_transpose", name=name, k=k, stride=2, same=False, act="swish",
                              w=dec_w[f"{name}/kernel"], b=dec_w[f"{name}/bias"]))
        specs.append(dict(kind="conv2d", name="output_image_400", k=3, stride=1, same=True, act="linear",
                          w=dec_w["output_image_400/kernel"], b=dec_w["output_image_400/bias"]))
    return specs


class _PinnedPool:
    """Recycling pool of page-locked result buffers (srcfd_host_alloc).  `empty(shape)` returns a float32 ndarray over a pool buffer;
    the buffer goes back to the pool when the array and every view of it are gone (a weakref finalizer on the ctypes object all
    the views hang from).
    * Sizes are rounded up to a few classes (four per octave: at most 25 % over-allocation), so callers with varying batch sizes
      share buffers instead of growing one bucket per exact size.
    * Bounded: at most `cap_bytes` stay cached (default 1 GiB, `SRCFD_RESULT_POOL_MB`); a released buffer that does not fit EVICTS the
      least recently used cached ones to make room (it is the most likely to be asked for again) instead of being freed itself.
    * Small results (< `min_bytes`) and hosts where the allocation fails use ordinary numpy memory.  SRCFD_RESULT_POOL=0 switches
      the pool off."""

    def __init__(self, min_bytes: int = 4 << 20, cap_bytes: Optional[int] = None):
        import threading
        from collections import OrderedDict
        if cap_bytes is None:
            try:
                cap_bytes = int(float(os.environ.get("SRCFD_RESULT_POOL_MB", "1024")) * (1 << 20))
            except ValueError:
                cap_bytes = 1 << 30
        self.min_bytes, self.cap_bytes = min_bytes, max(0, cap_bytes)
        self.free = OrderedDict()   # ptr -> class bytes, least recently released first
        self.cached = 0
        self.lock = threading.Lock()
        self.enabled = os.environ.get("SRCFD_RESULT_POOL", "1") not in ("0", "")
        self.stats = {"allocated": 0, "reused": 0, "fallback": 0, "evicted": 0}

    @staticmethod
    def size_class(nbytes: int) -> int:
        """nbytes rounded up to a multiple of a quarter of its leading power of two (and of 4 KiB)."""
        step = max(4096, 1 << max(nbytes.bit_length() - 3, 0))
        return -(-nbytes // step) * step

    def _release(self, ptr: int, cls_bytes: int) -> None:
        drop = []
        with self.lock:
            if cls_bytes > self.cap_bytes:
                drop.append(ptr)
            else:
                while self.cached + cls_bytes > self.cap_bytes and self.free:
                    old, ob = self.free.popitem(last=False)
                    self.cached -= ob
                    self.stats["evicted"] += 1
                    drop.append(old)
                self.free[ptr] = cls_bytes
                self.cached += cls_bytes
        for q in drop:
            L.lib.srcfd_host_free(C.c_void_p(q))

    def empty(self, shape) -> np.ndarray:
        import weakref
        nbytes = int(np.prod(shape)) * 4
        if not self.enabled or nbytes < self.min_bytes:
            return np.empty(shape, dtype=np.float32)
        cls_bytes = self.size_class(nbytes)
        ptr = None
        with self.lock:
            for q in reversed(self.free):            # most recently released first: its pages are the warmest
                if self.free[q] == cls_bytes:
                    ptr = q
                    break
            if ptr is not None:
                del self.free[ptr]
                self.cached -= cls_bytes
                self.stats["reused"] += 1
        if ptr is None:
            out = C.c_void_p()
            ok = L.lib.srcfd_host_alloc(cls_bytes, C.byref(out)) == 0 and bool(out.value)
            with self.lock:
                self.stats["allocated" if ok else "fallback"] += 1
            if not ok:
                return np.empty(shape, dtype=np.float32)
            ptr = out.value
        buf = (C.c_char * nbytes).from_address(ptr)
        weakref.finalize(buf, self._release, ptr, cls_bytes)
        return np.frombuffer(buf, dtype=np.float32).reshape(shape)

    def trim(self) -> None:
        """Frees every cached buffer (buffers still referenced by arrays stay alive until those die)."""
        with self.lock:
            items = list(self.free)
            self.free.clear()
            self.cached = 0
        for ptr in items:
            L.lib.srcfd_host_free(C.c_void_p(ptr))


_result_pool = _PinnedPool()


class SRModel:
    """Owns one ``srcfd_model*``.  ``device=None`` -> GPU 0 when one exists,
    else a host-only handle (shape / weight queries; predict raises)."""

    def __init__(self, handle: C.c_void_p, device: int):
        self._h = handle
        self.device = device
        self._keep = None

    # -- construction -------------------------------------------------------
    @staticmethod
    def _pick_device(device) -> int:
        if device is None:
            return 0 if device_count() > 0 else -1
        return int(device)

    @classmethod
    def load_h5(cls, encoder_h5=None, decoder_h5=None, device=None) -> "SRModel":
        dev = cls._pick_device(device)
        h = C.c_void_p()
        L.check(L.lib.srcfd_model_load_h5(L.enc(encoder_h5) if encoder_h5 else None,
                                          L.enc(decoder_h5) if decoder_h5 else None, dev, C.byref(h)))
        return cls(h, dev)

    @classmethod
    def load_superres_h5(cls, superres_h5, device=None) -> "SRModel":
        """The whole-model file `superres_model.save(...)` writes (sr-ae-conv.ipynb:c586): encoder + decoder in one .h5."""
        dev = cls._pick_device(device)
        h = C.c_void_p()
        L.check(L.lib.srcfd_model_load_superres_h5(L.enc(superres_h5), dev, C.byref(h)))
        return cls(h, dev)

    @classmethod
    def from_layers(cls, specs: Sequence[dict], in_shape: Tuple[int, int, int], device=None) -> "SRModel":
        dev = cls._pick_device(device)
        arr = (L.Layer * len(specs))()
        keep = []
        for i, s in enumerate(specs):
            l = arr[i]
            l.kind = _KIND[s["kind"]]
            l.activation = _ACT[s.get("act", "linear")]
            l.kh = l.kw = int(s.get("k", 1))
            l.stride = int(s.get("stride", 1))
            l.same_padding = int(bool(s.get("same", False)))
            if s.get("name"):
                nm = s["name"].encode()
                keep.append(nm)
                l.name = nm
            if "shape" in s:
                for j in range(3):
                    l.reshape[j] = int(s["shape"][j])
            if "w" in s:
                w = np.ascontiguousarray(s["w"], dtype=np.float32)
                b = np.ascontiguousarray(s["b"], dtype=np.float32)
                keep += [w, b]
                if l.kind == L.LAYER_DENSE:
                    l.cin, l.cout = w.shape
                elif l.kind == L.LAYER_CONV2D:
                    l.kh, l.kw, l.cin, l.cout = w.shape
                else:
                    l.kh, l.kw, l.cout, l.cin = w.shape
                l.kernel = w.ctypes.data_as(C.POINTER(C.c_float))
                l.bias = b.ctypes.data_as(C.POINTER(C.c_float))
        shp = (C.c_int * 3)(*in_shape)
        h = C.c_void_p()
        L.check(L.lib.srcfd_model_create(arr, len(specs), shp, dev, C.byref(h)))
        return cls(h, dev)

    @classmethod
    def from_weights(cls, enc_w, dec_w, device=None) -> "SRModel":
        in_shape = (10, 10, 1) if enc_w is not None else (1, 1, 50)
        return cls.from_layers(layers_from_weights(enc_w, dec_w), in_shape, device)

    def close(self):
        if getattr(self, "_h", None):
            L.lib.srcfd_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- queries ------------------------------------------------------------
    @property
    def input_shape(self) -> Tuple[int, int, int]:
        s = (C.c_int * 3)()
        L.check(L.lib.srcfd_model_input_shape(self._h, s))
        return tuple(s)

    @property
    def output_shape(self) -> Tuple[int, int, int]:
        s = (C.c_int * 3)()
        L.check(L.lib.srcfd_model_output_shape(self._h, s))
        return tuple(s)

    @property
    def macs_per_sample(self) -> int:
        return int(L.lib.srcfd_model_macs_per_sample(self._h))

    @property
    def has_fused_path(self) -> bool:
        return bool(L.lib.srcfd_model_has_fused_path(self._h))

    @property
    def precision(self) -> str:
        p = L.lib.srcfd_model_get_precision(self._h)
        return {L.PREC_FP32: "fp32", L.PREC_BF16: "bf16", L.PREC_FP32_NAIVE: "fp32_naive", L.PREC_F16: "f16", L.PREC_FP32X3: "fp32x3"}[p]

    @precision.setter
    def precision(self, name: str):
        L.check(L.lib.srcfd_model_set_precision(self._h, PRECISIONS[name]))

    def reserve(self, n: int) -> None:
        """Allocate / pack now what the first n-sample forward at the current precision would set up lazily."""
        L.check(L.lib.srcfd_model_reserve(self._h, int(n)))

    def footprint(self, n: int, precision: Optional[str] = None) -> dict:
        """Bytes an n-sample forward keeps allocated, computed without allocating (srcfd_model_footprint; works on a host-only
        handle): device activation workspace, device weights / operand packs (upper bound), device staging of the host-buffer
        entry, host bytes of one result."""
        arr = (C.c_size_t * 4)()
        prec = PRECISIONS[precision] if precision is not None else L.check(L.lib.srcfd_model_get_precision(self._h))
        L.check(L.lib.srcfd_model_footprint(self._h, int(n), prec, arr))
        return {"device_workspace": int(arr[0]), "device_weights": int(arr[1]), "device_host_entry_staging": int(arr[2]), "host_result": int(arr[3])}

    def layers(self) -> List[dict]:
        out = []
        for i in range(L.check(L.lib.srcfd_model_num_layers(self._h))):
            l = L.Layer()
            name = C.create_string_buffer(128)
            L.check(L.lib.srcfd_model_get_layer(self._h, i, C.byref(l), name, len(name)))
            d = dict(name=name.value.decode(), kind=l.kind, activation=l.activation, kh=l.kh, kw=l.kw, stride=l.stride,
                     same=bool(l.same_padding), cin=l.cin, cout=l.cout, reshape=tuple(l.reshape))
            if l.kernel:
                if l.kind == L.LAYER_DENSE:
                    shape = (l.cin, l.cout)
                elif l.kind == L.LAYER_CONV2D:
                    shape = (l.kh, l.kw, l.cin, l.cout)
                else:
                    shape = (l.kh, l.kw, l.cout, l.cin)
                d["kernel"] = np.ctypeslib.as_array(l.kernel, shape=(int(np.prod(shape)),)).reshape(shape).copy()
                d["bias"] = np.ctypeslib.as_array(l.bias, shape=(l.cout,)).copy()
            out.append(d)
        return out

    def weights(self) -> Dict[str, np.ndarray]:
        w = {}
        for d in self.layers():
            if "kernel" in d:
                w[f"{d['name']}/kernel"] = d["kernel"]
                w[f"{d['name']}/bias"] = d["bias"]
        return w

    def save_h5(self, encoder_h5=None, decoder_h5=None):
        L.check(L.lib.srcfd_model_save_h5(self._h, L.enc(encoder_h5) if encoder_h5 else None,
                                          L.enc(decoder_h5) if decoder_h5 else None))

    def save_superres_h5(self, superres_h5):
        """`superres_model.save(...)` (sr-ae-conv.ipynb:c586): both sub-models in one legacy Keras-H5 file."""
        L.check(L.lib.srcfd_model_save_superres_h5(self._h, L.enc(superres_h5)))

    # -- forward ------------------------------------------------------------
    def predict(self, x: np.ndarray, in_affine=None, out_affine=None, nan_guard: bool = False,
                return_nonfinite: bool = False, out: Optional[np.ndarray] = None):
        """numpy in / numpy out, like ``Model.predict`` (PyCFD...:858).  `out`: optional C-contiguous float32
        (n, oh, ow, oc) array to fill instead of allocating (a fresh 491 MB result costs more in first-touch page
        faults than in PCIe time; callers that predict repeatedly can reuse one)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        ih, iw, ic = self.input_shape
        if x.ndim == 2 and ih == 1 and iw == 1:
            x = x.reshape(x.shape[0], 1, 1, ic)
        if x.ndim != 4 or x.shape[1:] != (ih, iw, ic):
            raise ValueError(f"input shape {x.shape} incompatible with model input (None, {ih}, {iw}, {ic})")
        n = x.shape[0]
        oh, ow, oc = self.output_shape
        if out is None:
            # a result of this size moves at the PCIe rate only into page-locked memory, and a FRESH pageable array pays more in
            # first-touch page faults than in transfer time: large results come from a recycling pool of page-locked buffers
            # (returned to it when the last view of the array dies); `Model.predict` still hands back a new ndarray every call
            y = _result_pool.empty((n, oh, ow, oc)) if self.device >= 0 else np.empty((n, oh, ow, oc), dtype=np.float32)
        else:
            if out.shape != (n, oh, ow, oc) or out.dtype != np.float32 or not out.flags.c_contiguous:
                raise ValueError(f"out must be a C-contiguous float32 array of shape {(n, oh, ow, oc)}")
            y = out

        def aff(a):
            if a is None:
                return None, None
            a = np.ascontiguousarray(a, dtype=np.float32).reshape(n, 2)
            return a, a.ctypes.data_as(C.c_void_p)

        ain, pin = aff(in_affine)
        aout, pout = aff(out_affine)
        bad = C.c_int64(0)
        L.check(L.lib.srcfd_predict(self._h, x.ctypes.data_as(C.c_void_p), n, pin, pout, y.ctypes.data_as(C.c_void_p),
                                    L.FLAG_NAN_GUARD if nan_guard else 0, C.byref(bad)))
        if return_nonfinite:
            return y, int(bad.value)
        return y

    def predict_resampled(self, x: np.ndarray, resampler, in_affine=None, out_affine=None, nan_guard: bool = False,
                          return_nonfinite: bool = False):
        """`predict` + resampling of every (H,W) output by `resampler` (float64 result, like scipy's), the
        float32 network output never leaving the device (bfs_ml_accelerated.py:1109-1135)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        if tuple(x.shape[1:]) != self.input_shape:
            raise ValueError(f"expected input (N,{self.input_shape}), got {x.shape}")

        def aff(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != (n, 2):
                raise ValueError("affine must have shape (N, 2)")
            return a
        ai, ao = aff(in_affine), aff(out_affine)
        y = np.empty((n, resampler.out_h, resampler.out_w), np.float64)
        bad = C.c_int64(0)
        L.check(L.lib.srcfd_predict_resampled(self._h, resampler._h, x.ctypes.data_as(C.c_void_p), n,
                                              ai.ctypes.data_as(C.c_void_p) if ai is not None else None,
                                              ao.ctypes.data_as(C.c_void_p) if ao is not None else None,
                                              y.ctypes.data_as(C.c_void_p), L.FLAG_NAN_GUARD if nan_guard else 0, C.byref(bad)))
        return (y, int(bad.value)) if return_nonfinite else y

    def predict_into_solver_state(self, x: np.ndarray, bc_types, bc_values, left_profiles=None, resampler=None, in_affine=None,
                                  out_affine=None, nan_guard: bool = False, Var: Optional[np.ndarray] = None, return_nonfinite: bool = False):
        """`predict` for the u, v, p samples of one field + the transposed float64 injection into the solver's
        `Var (3, nx+2, ny+2)` + the ghost-cell pass, on the device (PyCFD_ML_accelerated.py:936-943).
        bc_types (3,4) int [left,right,top,bottom] 0=Dirichlet 1=Neumann, bc_values (3,4); left_profiles: optional
        {k: (ny,) float64} Dirichlet rows for the left boundary (BFS inlet, bfs_ml_accelerated.py:524-562)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if x.shape[0] != 3 or tuple(x.shape[1:]) != self.input_shape:
            raise ValueError(f"expected input (3,{self.input_shape}), got {x.shape}")
        oh, ow, _ = self.output_shape
        ny, nx = (resampler.out_h, resampler.out_w) if resampler is not None else (oh, ow)
        bt = np.ascontiguousarray(bc_types, dtype=np.int32).reshape(3, 4)
        bv = np.ascontiguousarray(bc_values, dtype=np.float64).reshape(3, 4)
        arr = (L.SolverBC * 3)()
        keep = []
        for k in range(3):
            for s_ in range(4):
                arr[k].type[s_] = int(bt[k, s_])
                arr[k].value[s_] = float(bv[k, s_])
            prof = None if left_profiles is None else left_profiles.get(k)
            if prof is not None:
                prof = np.ascontiguousarray(prof, dtype=np.float64)
                if prof.shape != (ny,):
                    raise ValueError(f"left profile of variable {k} must have shape ({ny},)")
                keep.append(prof)
                arr[k].left_profile = prof.ctypes.data_as(C.POINTER(C.c_double))

        def aff(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float32)
            if a.shape != (3, 2):
                raise ValueError("affine must have shape (3, 2)")
            return a
        ai, ao = aff(in_affine), aff(out_affine)
        if Var is None:
            Var = np.empty((3, nx + 2, ny + 2), np.float64)
        if Var.shape != (3, nx + 2, ny + 2) or Var.dtype != np.float64 or not Var.flags.c_contiguous:
            raise ValueError(f"Var must be a C-contiguous float64 array of shape (3, {nx + 2}, {ny + 2})")
        bad = C.c_int64(0)
        L.check(L.lib.srcfd_predict_into_solver_state(self._h, resampler._h if resampler is not None else None, x.ctypes.data_as(C.c_void_p),
                                                      ai.ctypes.data_as(C.c_void_p) if ai is not None else None,
                                                      ao.ctypes.data_as(C.c_void_p) if ao is not None else None,
                                                      arr, Var.ctypes.data_as(C.c_void_p), L.FLAG_NAN_GUARD if nan_guard else 0, C.byref(bad)))
        return (Var, int(bad.value)) if return_nonfinite else Var

    def predict_device(self, x, y, in_affine=None, out_affine=None, nan_guard=False, nonfinite=None, stream=None):
        """Device-resident forward on torch CUDA tensors (plumbing only):
        x float32 (n,h,w,c); y float32/bfloat16/float16 (n,oh,ow,oc); affines
        float32 (n,2).  Enqueues on ``stream`` (torch stream) or the current one."""
        import torch

        if not (x.is_cuda and y.is_cuda and x.is_contiguous() and y.is_contiguous()):
            raise ValueError("predict_device needs contiguous CUDA tensors")
        if x.dtype != torch.float32:
            raise ValueError("x must be float32")
        dt = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[y.dtype]
        n = x.shape[0]
        st = stream if stream is not None else torch.cuda.current_stream(x.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        L.check(L.lib.srcfd_predict_device(self._h, p(x), n, p(in_affine), p(out_affine), p(y), dt,
                                           L.FLAG_NAN_GUARD if nan_guard else 0, p(nonfinite), C.c_void_p(st.cuda_stream)))

    def debug_activation(self, index: int, shape, dtype=np.uint16) -> np.ndarray:
        """Test hook (srcfd_model_debug_activation): raw 16-bit inter-kernel activations."""
        out = np.empty(shape, dtype=dtype)
        L.check(L.lib.srcfd_model_debug_activation(self._h, index, out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def last_plan(self) -> dict:
        """Test hook (srcfd_model_last_plan): which implementation of each stage the last forward ran, e.g.
        {'precision': 'bf16', 'encoder': 'enc16', 'dense_1': 'dense1_16', 'middle': 'mid16_4x64', 'tail': 'tail16', 'tail_seg': '10', 'graph': 'eager'}."""
        buf = C.create_string_buffer(512)
        L.check(L.lib.srcfd_model_last_plan(self._h, buf, len(buf)))
        return dict(w.split("=", 1) for w in buf.value.decode().split())

    def set_profiling(self, on: bool):
        L.check(L.lib.srcfd_model_set_profiling(self._h, int(on)))

    def get_profile(self) -> List[Tuple[str, float]]:
        names = C.create_string_buffer(1 << 16)
        ms = (C.c_float * 4096)()
        cnt = C.c_int(0)
        L.check(L.lib.srcfd_model_get_profile(self._h, names, len(names), ms, C.byref(cnt), 4096))
        ns = names.value.decode().split("\n") if cnt.value else []
        return [(ns[i], float(ms[i])) for i in range(cnt.value)]
