"""Load the reference's dense modules from /root/reference for golden-vector capture.

Runs ONLY in the build container (the reference never travels to the GPU box).
Recipe from SURVEY.md section 8(c): an empty `cv2` stub module (cv2 is only used by
`_prepare_images`, which golden capture bypasses) and a synthetic package whose
__path__ points at the reference's src/core so `core/__init__.py` (which imports
cv2 users) is not executed.
"""
import importlib
import os
import sys
import types

REF_CORE = "/root/reference/src/core"


def available() -> bool:
    return os.path.isdir(REF_CORE)


def load():
    if not available():
        raise RuntimeError("reference tree not present at " + REF_CORE)
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "refcore" not in sys.modules:
        pkg = types.ModuleType("refcore")
        pkg.__path__ = [REF_CORE]
        sys.modules["refcore"] = pkg
    mvs = importlib.import_module("refcore.mvs_patchmatch")
    stereo = importlib.import_module("refcore.dense_stereo")
    cam = importlib.import_module("refcore.camera")
    return mvs, stereo, cam
