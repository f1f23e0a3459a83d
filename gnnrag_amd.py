"""Import shim: ``import gnnrag_amd`` loads the package that lives in ``gnn-rag_amd/``.

The package directory keeps the repository's name (``gnn-rag_amd``), which is not a
valid Python identifier; this module registers it under ``gnnrag_amd`` instead.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "gnn-rag_amd")
_spec = _ilu.spec_from_file_location(
    "gnnrag_amd", _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["gnnrag_amd"] = _mod
_spec.loader.exec_module(_mod)
