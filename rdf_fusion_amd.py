"""Import shim: the package directory is ``rdf-fusion_amd/`` (a hyphen cannot be imported), so
``import rdf_fusion_amd`` loads it under this importable name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rdf-fusion_amd")
_spec = importlib.util.spec_from_file_location(
    "rdf_fusion_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rdf_fusion_amd"] = _mod
_spec.loader.exec_module(_mod)
