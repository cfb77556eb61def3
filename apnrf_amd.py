"""Import alias: `import apnrf_amd` loads the package directory
`active-perception-using-neural-radiance-fields_amd/` (whose name is not a valid Python identifier)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "active-perception-using-neural-radiance-fields_amd")
_spec = importlib.util.spec_from_file_location(
    "apnrf_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["apnrf_amd"] = _mod
_spec.loader.exec_module(_mod)
