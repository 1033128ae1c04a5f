"""MI355X-native PatchMatch-MVS / plane-sweep dense-reconstruction backend.

Drop-in for the dense stage of dackey-wav/3d-reconstruction-tool
(src/core/mvs_patchmatch.py, src/core/dense_stereo.py): same Python call surface,
per-pixel work in hand-written gfx950 kernels behind a C ABI (include/amvs.h).
The directory name is not a Python identifier; import it with
importlib.import_module("3d-reconstruction-tool_amd") or through the `amvs` shim at the
repository root.
"""
from .core.camera import Camera, CameraPose, load_calibration  # noqa: F401

__all__ = ["Camera", "CameraPose", "load_calibration", "PatchMatchMVS", "DepthNormalMap",
           "DenseStereoReconstructor", "Engine", "AmvsError"]


def __getattr__(name):
    # resolved lazily so importing the package (e.g. for the synthetic generator or the
    # camera types) does not require the shared library
    if name in ("PatchMatchMVS", "DepthNormalMap"):
        from .core import mvs_patchmatch
        return getattr(mvs_patchmatch, name)
    if name == "DenseStereoReconstructor":
        from .core import dense_stereo
        return dense_stereo.DenseStereoReconstructor
    if name == "Engine":
        from .engine import Engine
        return Engine
    if name == "AmvsError":
        from ._lib import AmvsError
        return AmvsError
    raise AttributeError(name)
