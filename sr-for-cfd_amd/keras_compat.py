"""The slice of the Keras API the solver scripts use for the SR call, on libsrcfd.

Reference surface (PyCFD_ML_accelerated.py:4-5, 676-689, 831-833, 858;
bfs_ml_accelerated.py:14-15, 873-886, 1069-1071, 1109):

    import tensorflow as tf
    from tensorflow.keras import Model
    class SuperResolutionAE(Model):
        def __init__(self, encoder_lr, decoder_hr, **kw): super().__init__(**kw); ...
        def call(self, inputs, training=False):
            return self.decoder_hr(self.encoder_lr(inputs, training=training), training=training)
    enc = tf.keras.models.load_model(encoder_file, compile=False)
    dec = tf.keras.models.load_model(decoder_file, compile=False)
    y = SuperResolutionAE(enc, dec).predict(x, verbose=0)        # float32 NHWC in/out

`load_model` returns a `LoadedModel`; calling a model on a symbolic tensor records
the chain of loaded sub-models, and `predict` runs that chain as ONE device
handle (`srcfd_model_load_h5(encoder, decoder)`): weights are uploaded once per
file pair and cached, instead of the reference's load-on-every-call.
`compat/tensorflow` re-exports this module under the names the scripts import.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .engine import SRModel, device_count

_DEFAULT_PRECISION = os.environ.get("SRCFD_PRECISION", "fp32")
# (paths, precision, device) -> (mtimes of the files when loaded, handle).  One entry per file pair: a re-saved file
# (new mtime) closes and replaces the stale handle instead of piling up device memory next to it.
_HANDLE_CACHE: Dict[Tuple, Tuple[Tuple, SRModel]] = {}
_DEFAULT_DEVICE: Optional[int] = None


def set_default_device(index: Optional[int]) -> None:
    """GPU the Keras-style surface runs on.  None (default): torch's current CUDA device when torch is already imported
    and initialised, else LOCAL_RANK (one process per GPU under torch.distributed.run), else device 0."""
    global _DEFAULT_DEVICE
    _DEFAULT_DEVICE = None if index is None else int(index)


def _pick_device() -> int:
    n = device_count()
    if n <= 0:
        return -1
    if _DEFAULT_DEVICE is not None:
        return _DEFAULT_DEVICE
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        return int(torch.cuda.current_device())
    try:
        return int(os.environ.get("LOCAL_RANK", "0")) % n
    except ValueError:
        return 0


def clear_handle_cache() -> None:
    """Closes every cached device handle (weights + workspace)."""
    for _, m in _HANDLE_CACHE.values():
        m.close()
    _HANDLE_CACHE.clear()


def set_default_precision(name: str) -> None:
    """'fp32' (default: the <=1e-5 parity path), 'bf16' or 'f16' (fused throughput path)."""
    global _DEFAULT_PRECISION
    if name not in ("fp32", "bf16", "f16", "fp32_naive"):
        raise ValueError(name)
    _DEFAULT_PRECISION = name


class _Symbolic:
    """Stands for a tensor flowing through `Model.call`; only records which loaded
    sub-models it passed through, in order."""

    def __init__(self, value: np.ndarray, chain: Tuple["LoadedModel", ...] = ()):
        self.value = value
        self.chain = chain

    def _no(self, *a, **k):
        raise NotImplementedError(
            "sr-for-cfd_amd runs `call` symbolically: only compositions of loaded sub-models "
            "(decoder(encoder(x))) are supported, not tensor arithmetic inside `call`")

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __truediv__ = __getitem__ = _no


def _device_handle(paths: Tuple[Optional[str], ...], precision: str, device: Optional[int] = None) -> SRModel:
    dev = _pick_device() if device is None else int(device)
    key = (paths, precision, dev)
    stamp = tuple(os.path.getmtime(p) if p else None for p in paths)
    hit = _HANDLE_CACHE.get(key)
    if hit is not None and hit[0] == stamp:
        return hit[1]
    if hit is not None:       # the file was re-saved: drop the stale handle and its device memory
        hit[1].close()
        del _HANDLE_CACHE[key]
    enc, dec = (paths + (None,))[:2]
    m = SRModel.load_h5(enc, dec, device=dev)
    if precision in ("bf16", "f16") and not m.has_fused_path:
        precision = "fp32"  # the fused path needs the encoder_10 + decoder_400 pair
    m.precision = precision
    _HANDLE_CACHE[key] = (stamp, m)
    return m


class Model:
    """Base class for user subclasses (`class SuperResolutionAE(Model)`)."""

    def __init__(self, *args, **kwargs):
        self.name = kwargs.pop("name", type(self).__name__.lower())
        self._precision = kwargs.pop("precision", None)
        self.built = True

    def call(self, inputs, training=False):  # pragma: no cover - overridden
        raise NotImplementedError

    def __call__(self, inputs, training=False, **kwargs):
        if isinstance(inputs, _Symbolic):
            return self.call(inputs, training=training)
        return self.predict(inputs, verbose=0)

    def compile(self, *args, **kwargs):
        return None

    def _chain(self, x: np.ndarray) -> Tuple["LoadedModel", ...]:
        out = self.call(_Symbolic(x), training=False)
        if not isinstance(out, _Symbolic) or not out.chain:
            raise NotImplementedError("`call` must return the output of a chain of loaded sub-models")
        return out.chain

    def predict(self, x, batch_size=None, verbose=0, steps=None, callbacks=None, **kwargs):
        """float32 NHWC in -> new float32 NHWC out (Keras `Model.predict`, PyCFD...:858).
        `batch_size` is accepted for compatibility; results do not depend on it."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        chain = self._chain(x)
        if len(chain) > 2:
            raise NotImplementedError("at most two chained sub-models (encoder, decoder) are supported")
        paths = tuple(m.path for m in chain)
        handle = _device_handle(paths, self._precision or _DEFAULT_PRECISION)
        return handle.predict(x)


class _LayerView:
    def __init__(self, d: dict):
        self.name = d["name"]
        self._w = [d[k] for k in ("kernel", "bias") if k in d]

    def get_weights(self) -> List[np.ndarray]:
        return [w.copy() for w in self._w]


class LoadedModel(Model):
    """What `load_model` returns: one legacy Keras-H5 sub-model file."""

    def __init__(self, path: str):
        super().__init__(name=os.path.splitext(os.path.basename(path))[0])
        self.path = os.fspath(path)
        self._host = SRModel.load_h5(self.path, None, device=-1)  # parses + validates now, like load_model does

    def call(self, inputs, training=False):
        if isinstance(inputs, _Symbolic):
            return _Symbolic(inputs.value, inputs.chain + (self,))
        return self.predict(inputs)

    @property
    def input_shape(self):
        h, w, c = self._host.input_shape
        return (None, c) if (h, w) == (1, 1) else (None, h, w, c)

    @property
    def output_shape(self):
        h, w, c = self._host.output_shape
        return (None, c) if (h, w) == (1, 1) else (None, h, w, c)

    @property
    def layers(self) -> List[_LayerView]:
        return [_LayerView(d) for d in self._host.layers()]

    def get_weights(self) -> List[np.ndarray]:
        out = []
        for l in self.layers:
            out += l.get_weights()
        return out

    def count_params(self) -> int:
        return int(sum(w.size for w in self.get_weights()))

    def summary(self, print_fn=print):
        print_fn(f'Model: "{self.name}"  input {self.input_shape} -> output {self.output_shape}, {self.count_params():,} params')
        for d in self._host.layers():
            print_fn(f"  {d['name']:<24} kind={d['kind']} cin={d['cin']} cout={d['cout']}")

    def save(self, path, **kwargs):
        self._host.save_h5(os.fspath(path))

    def predict(self, x, batch_size=None, verbose=0, **kwargs):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = _device_handle((self.path,), "fp32" if not self._host.has_fused_path else (self._precision or _DEFAULT_PRECISION)).predict(x)
        oh, ow, oc = self._host.output_shape
        return y.reshape(y.shape[0], oc) if (oh, ow) == (1, 1) else y


def load_model(filepath, custom_objects=None, compile=True, safe_mode=True, **kwargs) -> LoadedModel:
    """`tf.keras.models.load_model(path, compile=False)` for legacy `.h5` sub-models.
    Missing file -> FileNotFoundError, unreadable -> OSError (both are what the
    reference's `except (IOError, OSError)` expects, PyCFD...:835-837)."""
    return LoadedModel(filepath)


class _Namespace:
    def __init__(self, **kw):
        self.__dict__.update(kw)


models = _Namespace(load_model=load_model, Model=Model)
