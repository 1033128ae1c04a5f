"""Importable alias of the `3d-reconstruction-tool_amd` package, whose directory name is not a
valid Python identifier:  `import amvs; amvs.PatchMatchMVS(...)`, `from amvs.engine import ...`.

`amvs` and every `amvs.<sub>` name resolve to the SAME module objects as
`3d-reconstruction-tool_amd[.<sub>]` (a meta-path finder, so nothing is imported twice).
"""
import importlib
import importlib.abc
import importlib.util
import sys

_REAL = "3d-reconstruction-tool_amd"
_ALIAS = __name__


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
