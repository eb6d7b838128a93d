"""ctypes binding of libsrcfd.so (C ABI in include/srcfd.h).

The library is the product: if it is missing or cannot be loaded this module
raises -- there is no Python/CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRCFD_LIB") or os.path.join(_HERE, "lib", "libsrcfd.so")  # SRCFD_LIB: A/B builds of the same ABI

OK, ENOENT, EIO, ENOMEM, ENODEV, EINVAL, EKEY, EHIP = 0, -2, -5, -12, -19, -22, -126, -1000

F32, F64, I32, I64, U8, STR, BF16, F16 = 0, 1, 2, 3, 4, 5, 16, 17
PREC_FP32, PREC_BF16, PREC_FP32_NAIVE, PREC_F16, PREC_FP32X3 = 0, 1, 2, 3, 4
TRAIN_OVERWRITE = 1   # srcfd_trainer_forward_backward_ex: grads and sse are written, not added into
TRAIN_SAME_PARAMS = 2  # ... params unchanged since this trainer's previous call: the operand re-packing is skipped
LAYER_CONV2D, LAYER_CONV2D_TRANSPOSE, LAYER_DENSE, LAYER_FLATTEN, LAYER_RESHAPE = 1, 2, 3, 4, 5
ACT_LINEAR, ACT_SWISH, ACT_RELU, ACT_SIGMOID, ACT_TANH = 0, 1, 2, 3, 4
FLAG_NAN_GUARD = 1


class SrcfdError(RuntimeError):
    """Any libsrcfd failure that has no closer built-in exception type."""


class NoDeviceError(SrcfdError):
    """No HIP device (SRCFD_ENODEV): the extension cannot compute here."""


class Layer(C.Structure):
    _fields_ = [
        ("kind", C.c_int), ("activation", C.c_int),
        ("kh", C.c_int), ("kw", C.c_int), ("stride", C.c_int), ("same_padding", C.c_int),
        ("cin", C.c_int), ("cout", C.c_int),
        ("reshape", C.c_int * 3),
        ("kernel", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
        ("name", C.c_char_p),
    ]


class CoarseProblem(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("lx", C.c_double), ("ly", C.c_double),
                ("reynolds", C.c_double), ("rho", C.c_double), ("dt", C.c_double),
                ("scheme", C.c_int), ("max_iterations", C.c_int), ("tolerance", C.c_double * 3),
                ("bc_type", (C.c_int * 4) * 3), ("bc_value", (C.c_double * 4) * 3),
                ("case_type", C.c_int), ("relax", C.c_double * 3),
                ("step_height", C.c_double), ("channel_height", C.c_double), ("bulk_velocity", C.c_double)]


class SolverBC(C.Structure):
    _fields_ = [("type", C.c_int * 4), ("value", C.c_double * 4), ("left_profile", C.POINTER(C.c_double))]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C sr-for-cfd_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    return C.CDLL(LIB_PATH)


def _share_hip_runtime_with_torch() -> None:
    """One HIP runtime per process.  PyTorch-ROCm ships its own libamdhip64.so (soname
    libamdhip64.so.7) and links it by the unversioned name; libsrcfd needs `libamdhip64.so.7`.
    Whichever loads first decides whether the second resolves to the same object: torch first is
    fine, libsrcfd first would bring in /opt/rocm's copy and torch then loads a second runtime that
    finds no devices.  So when torch is installed (not necessarily imported), map its copy first."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


lib = _load()

_p = C.c_void_p
_sz = C.c_size_t
_protos = {
    "srcfd_last_error": (C.c_char_p, []),
    "srcfd_version": (C.c_char_p, []),
    "srcfd_device_count": (C.c_int, []),
    "srcfd_model_load_h5": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_p)]),
    "srcfd_model_create": (C.c_int, [C.POINTER(Layer), C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(_p)]),
    "srcfd_model_destroy": (None, [_p]),
    "srcfd_model_input_shape": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "srcfd_model_output_shape": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "srcfd_model_num_layers": (C.c_int, [_p]),
    "srcfd_model_get_layer": (C.c_int, [_p, C.c_int, C.POINTER(Layer), C.c_char_p, _sz]),
    "srcfd_model_macs_per_sample": (C.c_int64, [_p]),
    "srcfd_model_set_precision": (C.c_int, [_p, C.c_int]),
    "srcfd_model_get_precision": (C.c_int, [_p]),
    "srcfd_model_reserve": (C.c_int, [_p, C.c_int]),
    "srcfd_model_has_fused_path": (C.c_int, [_p]),
    "srcfd_predict": (C.c_int, [_p, _p, C.c_int, _p, _p, _p, C.c_int, C.POINTER(C.c_int64)]),
    "srcfd_predict_device": (C.c_int, [_p, _p, C.c_int, _p, _p, _p, C.c_int, C.c_int, _p, _p]),
    "srcfd_model_workspace": (C.c_int, [_p, C.c_int, C.POINTER(_sz)]),
    "srcfd_model_footprint": (C.c_int, [_p, C.c_int, C.c_int, C.POINTER(_sz)]),
    "srcfd_model_set_profiling": (C.c_int, [_p, C.c_int]),
    "srcfd_model_get_profile": (C.c_int, [_p, C.c_char_p, _sz, C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int]),
    "srcfd_model_debug_activation": (C.c_int, [_p, C.c_int, _p, _sz]),
    "srcfd_model_last_plan": (C.c_int, [_p, C.c_char_p, _sz]),
    "srcfd_host_alloc": (C.c_int, [_sz, C.POINTER(C.c_void_p)]),
    "srcfd_host_free": (None, [_p]),
    "srcfd_model_save_h5": (C.c_int, [_p, C.c_char_p, C.c_char_p]),
    "srcfd_resampler_create": (C.c_int, [C.c_int, _p, _p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_p)]),
    "srcfd_resampler_destroy": (None, [_p]),
    "srcfd_resample_device": (C.c_int, [_p, _p, C.c_int, _p, _p]),
    "srcfd_predict_resampled": (C.c_int, [_p, _p, _p, C.c_int, _p, _p, _p, C.c_int, C.POINTER(C.c_int64)]),
    "srcfd_predict_into_solver_state": (C.c_int, [_p, _p, _p, _p, _p, C.POINTER(SolverBC), _p, C.c_int, C.POINTER(C.c_int64)]),
    "srcfd_trainer_create": (C.c_int, [_p, C.c_int, C.POINTER(_p)]),
    "srcfd_trainer_destroy": (None, [_p]),
    "srcfd_trainer_num_params": (C.c_int64, [_p]),
    "srcfd_trainer_get_params": (C.c_int, [_p, _p]),
    "srcfd_trainer_forward_backward": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_float, _p, _p, _p]),
    "srcfd_trainer_forward_backward_ex": (C.c_int, [_p, _p, _p, _p, C.c_int, C.c_float, _p, _p, C.c_int, _p]),
    "srcfd_model_save_superres_h5": (C.c_int, [_p, C.c_char_p]),
    "srcfd_model_load_superres_h5": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_p)]),
    "srcfd_prepare_inputs_device": (C.c_int, [_p, C.c_int, C.c_int, C.c_int, _p, _p, C.c_int, _p, C.c_int, C.c_double, _p, _p, _p]),
    "srcfd_coarse_solve": (C.c_int, [C.POINTER(CoarseProblem), _p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "srcfd_adam_step": (C.c_int, [_p, _p, _p, _p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, _p]),
    "srcfd_stats_load": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "srcfd_stats_save": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "srcfd_h5_open": (C.c_int, [C.c_char_p, C.POINTER(_p)]),
    "srcfd_h5_close": (None, [_p]),
    "srcfd_h5_list": (C.c_int, [_p, C.c_char_p, C.c_char_p, _sz, C.POINTER(_sz)]),
    "srcfd_h5_kind": (C.c_int, [_p, C.c_char_p]),
    "srcfd_h5_dataset_info": (C.c_int, [_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "srcfd_h5_read": (C.c_int, [_p, C.c_char_p, _p, _sz, C.c_int]),
    "srcfd_h5_attr_string": (C.c_int, [_p, C.c_char_p, C.c_char_p, C.c_char_p, _sz, C.POINTER(_sz)]),
    "srcfd_h5_attr_numeric": (C.c_int, [_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]),
    "srcfd_h5_attr_names": (C.c_int, [_p, C.c_char_p, C.c_char_p, _sz, C.POINTER(_sz)]),
    "srcfd_h5w_create": (C.c_int, [C.POINTER(_p)]),
    "srcfd_h5w_free": (None, [_p]),
    "srcfd_h5w_group": (C.c_int, [_p, C.c_char_p]),
    "srcfd_h5w_dataset": (C.c_int, [_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), _p]),
    "srcfd_h5w_attr_strings": (C.c_int, [_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int]),
    "srcfd_h5w_attr_numeric": (C.c_int, [_p, C.c_char_p, C.c_char_p, C.c_int, _p, C.c_int, C.c_int]),
    "srcfd_h5w_save": (C.c_int, [_p, C.c_char_p]),
}
for _name, (_res, _args) in _protos.items():
    _fn = getattr(lib, _name)  # AttributeError here = ABI drift, fail loudly
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED = tuple(_protos)


def last_error() -> str:
    return (lib.srcfd_last_error() or b"").decode("utf-8", "replace")


def check(rc: int) -> int:
    """Map a status code to the exception the reference's Python would raise
    (SURVEY.md 8b 'errors')."""
    if rc >= 0:
        return rc
    msg = last_error()
    if rc == ENOENT:
        raise FileNotFoundError(msg)
    if rc == EIO:
        raise OSError(msg)
    if rc == EKEY:
        raise KeyError(msg)
    if rc == EINVAL:
        raise ValueError(msg)
    if rc == ENOMEM:
        raise MemoryError(msg)
    if rc == ENODEV:
        raise NoDeviceError(msg)
    raise SrcfdError(f"libsrcfd error {rc}: {msg}")


def enc(path) -> bytes:
    return os.fsencode(path)
