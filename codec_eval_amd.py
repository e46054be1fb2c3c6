"""Import shim: the package directory is ``codec-eval_amd`` (not a Python identifier), so
``import codec_eval_amd`` resolves here and re-exports that package as this module."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("codec-eval_amd")
sys.modules[__name__] = _pkg
