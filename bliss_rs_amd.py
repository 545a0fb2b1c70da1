"""Import shim: the package lives in ./bliss-rs_amd/ (a directory name Python cannot import
directly).  `import bliss_rs_amd` loads that directory as the package `bliss_rs_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bliss-rs_amd")
_spec = importlib.util.spec_from_file_location(
    "bliss_rs_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["bliss_rs_amd"] = _mod
_spec.loader.exec_module(_mod)
