import importlib
import os
import sys

_repo = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
if _repo not in sys.path:
    sys.path.insert(0, _repo)
_kc = importlib.import_module("sr-for-cfd_amd.keras_compat")

Model = _kc.Model
load_model = _kc.load_model
from . import models  # noqa: E402,F401
