"""Import alias: the package directory is named ``coot-videotext_amd`` (layout contract), which is not
a valid Python identifier; ``import coot_videotext_amd`` loads it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "coot-videotext_amd")
_spec = importlib.util.spec_from_file_location("coot_videotext_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["coot_videotext_amd"] = _mod
_spec.loader.exec_module(_mod)
