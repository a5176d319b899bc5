"""``import holocron`` -> the MI355X-native package (``holocron_amd``).

The reference's scripts (`references/classification/train.py:30-36`, `references/detection/train.py:28-33`) and user code import
``holocron``, ``holocron.models``, ``holocron.trainer``, ``holocron.utils.misc`` ...; this alias makes every ``holocron[.x.y]`` name
resolve to the SAME module object as ``holocron_amd[.x.y]`` (no second copy of any module state), so those scripts run on this
package unchanged.  Put the repository root on ``sys.path`` in front of any installed reference package."""
import importlib
import importlib.abc
import importlib.util
import sys

import holocron_amd as _real

_PREFIX, _TARGET = "holocron", "holocron_amd"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real_name):
        self.real_name = real_name

    def create_module(self, spec):
        return importlib.import_module(self.real_name)

    def exec_module(self, module):       # the real module is already initialised
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _PREFIX and not fullname.startswith(_PREFIX + "."):
            return None
        real_name = _TARGET + fullname[len(_PREFIX):]
        try:
            real_spec = importlib.util.find_spec(real_name)
        except (ImportError, AttributeError, ValueError):
            real_spec = None
        if real_spec is None:
            return None
        spec = importlib.util.spec_from_loader(fullname, _AliasLoader(real_name), is_package=real_spec.submodule_search_locations is not None)
        return spec


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
for _name, _mod in list(sys.modules.items()):
    if _name == _TARGET or _name.startswith(_TARGET + "."):
        sys.modules[_PREFIX + _name[len(_TARGET):]] = _mod
sys.modules[__name__] = _real
