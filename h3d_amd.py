"""Import alias: the product package lives in the directory `human-3d-reconstruction_amd/`
(not a valid Python identifier), so `import h3d_amd` loads that directory as package `h3d_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "human-3d-reconstruction_amd")
_spec = importlib.util.spec_from_file_location(
    "h3d_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["h3d_amd"] = _mod
_spec.loader.exec_module(_mod)
