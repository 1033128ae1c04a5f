"""Importable alias of the `3d-reconstruction-tool_amd` package (whose directory name is not
a valid Python identifier):  `import amvs; amvs.PatchMatchMVS(...)`."""
import importlib
import sys

_pkg = importlib.import_module("3d-reconstruction-tool_amd")
sys.modules[__name__] = _pkg
