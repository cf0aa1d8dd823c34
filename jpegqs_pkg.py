"""Import helper: the package directory is named `jpeg-quantsmooth_amd` (with a
hyphen, as the project layout prescribes), which is not a Python identifier, so
it is loaded by path and registered as `jpeg_quantsmooth_amd`."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent
PKG_DIR = ROOT / "jpeg-quantsmooth_amd"
NAME = "jpeg_quantsmooth_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(
        NAME, PKG_DIR / "__init__.py", submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
