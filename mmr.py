"""Import shim: ``import mmr`` loads the package kept in ``multimodal-registration_amd/``
(the directory name the project mandates is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodal-registration_amd")
_spec = importlib.util.spec_from_file_location(
    "mmr", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mmr"] = _mod
_spec.loader.exec_module(_mod)
