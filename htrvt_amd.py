"""Importable alias for the package directory `htr-vt_amd/` (a hyphen is not a
valid Python identifier).  `import htrvt_amd` loads that directory as a package."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "htr-vt_amd")
_spec = importlib.util.spec_from_file_location("htrvt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["htrvt_amd"] = _mod
_spec.loader.exec_module(_mod)
