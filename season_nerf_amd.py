"""Import shim: the package directory is named `season-nerf_amd/` (not a valid Python identifier), this module makes
`import season_nerf_amd` resolve to it when the repo root is on sys.path."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "season-nerf_amd")
_spec = importlib.util.spec_from_file_location("season_nerf_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["season_nerf_amd"] = _mod
_spec.loader.exec_module(_mod)
