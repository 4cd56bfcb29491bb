"""Importable alias of the hyphen-named package ``point-cloud-registration-with-global-refinement_amd``:
``import pcr_amd`` returns that package (same module object)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("point-cloud-registration-with-global-refinement_amd")
sys.modules[__name__] = _pkg
