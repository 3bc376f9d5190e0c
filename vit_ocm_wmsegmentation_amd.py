"""Import alias: the package directory is named `vit-ocm-wmsegmentation_amd` (hyphen), which
Python cannot import by name. Importing `vit_ocm_wmsegmentation_amd` loads that directory as
a regular package under this name (sub-modules resolve inside it)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vit-ocm-wmsegmentation_amd")
_spec = importlib.util.spec_from_file_location(
    __name__, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
