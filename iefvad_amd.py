"""Import shim: the package lives in the directory `ief-vad_amd/` (a name Python cannot
import because of the hyphen); `import iefvad_amd` loads it from there under this name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ief-vad_amd")
_spec = importlib.util.spec_from_file_location(
    "iefvad_amd", os.path.join(_pkg_dir, "__init__.py"), submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["iefvad_amd"] = _mod
_spec.loader.exec_module(_mod)
