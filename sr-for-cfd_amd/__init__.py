"""sr-for-cfd_amd: MI355X-native super-resolution hot path of bitseal02/SR-for-CFD.

Import with ``importlib.import_module("sr-for-cfd_amd")`` or through the
``srcfd_amd`` alias module at the repo root (the directory name is not a
Python identifier).
"""
from ._lib import (LIB_PATH, NoDeviceError, SrcfdError)  # noqa: F401  (raises ImportError if the .so is missing)
from .engine import SRModel, device_count, layers_from_weights  # noqa: F401
from .h5 import H5File, H5Writer, read_coarse_fields  # noqa: F401
from .stats import load_stats, save_stats  # noqa: F401

__all__ = ["SRModel", "device_count", "layers_from_weights", "H5File", "H5Writer", "read_coarse_fields",
           "load_stats", "save_stats", "LIB_PATH", "NoDeviceError", "SrcfdError"]
