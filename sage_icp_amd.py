"""Loader: makes the package directory `sage-icp_amd/` (hyphen, not importable by name) available
as the module `sage_icp_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sage-icp_amd")
_spec = importlib.util.spec_from_file_location(
    "sage_icp_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sage_icp_amd"] = _mod
_spec.loader.exec_module(_mod)
