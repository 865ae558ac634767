"""Registers the source directory ``robust-multimodal-contrastive-learning_amd/`` (not a valid Python
identifier) as the importable package ``rmcl_amd``.  ``import rmcl_pkg`` once, then
``from rmcl_amd.vilt.modules import ViLTransformerSS``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_ROOT, "robust-multimodal-contrastive-learning_amd")


def load():
    if "rmcl_amd" in sys.modules:
        return sys.modules["rmcl_amd"]
    spec = importlib.util.spec_from_file_location(
        "rmcl_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rmcl_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


load()
