"""Load the product package.  Its directory is named `r-vio_amd` (hyphenated,
per the repo contract), which is not a valid Python identifier, so it is
registered under the module name `rvio_amd`."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "r-vio_amd")


def load_pkg():
    if "rvio_amd" in sys.modules:
        return sys.modules["rvio_amd"]
    spec = importlib.util.spec_from_file_location(
        "rvio_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["rvio_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
