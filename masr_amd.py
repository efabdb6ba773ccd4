"""Alias: `import masr_amd` == the `metaasr-crossaccent_amd` package (its name is not a Python identifier).

Submodules are aliased too: `masr_amd.engine` IS `metaasr-crossaccent_amd.engine` (one module object under two names).  Without
that, `from masr_amd.engine import X` would import the same file a second time under the alias name, with its own copy of every
class and module-level state."""
import importlib
import importlib.abc
import importlib.util
import sys
from pathlib import Path

_REAL = "metaasr-crossaccent_amd"
_root = str(Path(__file__).resolve().parent)
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(__name__ + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(__name__):])     # the one real module object

    def exec_module(self, module):
        pass                                                                   # already executed under its real name


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
