"""Stand-in `tensorflow` package: ONLY the names the SR-for-CFD solver scripts import
(PyCFD_ML_accelerated.py:4-5, bfs_ml_accelerated.py:14-15), routed to libsrcfd.

Put this directory's parent on sys.path *instead of* a real TensorFlow:
    PYTHONPATH=/path/to/repo/sr-for-cfd_amd/compat python PyCFD_ML_accelerated.py
"""
from . import keras  # noqa: F401

__version__ = "0.0-srcfd"
