"""Import shim: the package directory is named ``ims-toucan-prosody-variance_amd`` (hyphens,
as the project layout prescribes), which Python cannot import by name.  Importing
``ims_toucan_prosody_variance_amd`` loads that directory as a regular package."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ims-toucan-prosody-variance_amd")
_spec = importlib.util.spec_from_file_location(
    "ims_toucan_prosody_variance_amd",
    os.path.join(_PKG_DIR, "__init__.py"),
    submodule_search_locations=[_PKG_DIR],
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ims_toucan_prosody_variance_amd"] = _mod
_spec.loader.exec_module(_mod)
