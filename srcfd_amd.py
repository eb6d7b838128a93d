"""Importable alias of the ``sr-for-cfd_amd`` package (hyphenated directory)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("sr-for-cfd_amd")
sys.modules[__name__] = _pkg
