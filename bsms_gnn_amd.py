"""Import alias: the package directory is `bsms-gnn_amd/` (not a valid Python identifier), so this
one-file module loads it under the importable name `bsms_gnn_amd` and replaces itself with it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bsms-gnn_amd")
_spec = importlib.util.spec_from_file_location("bsms_gnn_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bsms_gnn_amd"] = _mod
_spec.loader.exec_module(_mod)
