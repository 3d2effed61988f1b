"""Import shim: the package directory is ``cl-drd_amd/`` (not a valid Python identifier),
so ``import cldrd_amd`` loads it from there and replaces this module in ``sys.modules``."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cl-drd_amd")
_spec = importlib.util.spec_from_file_location(
    "cldrd_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cldrd_amd"] = _mod
_spec.loader.exec_module(_mod)
