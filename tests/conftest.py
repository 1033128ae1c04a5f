import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


class GoldenScene:
    """A committed synthetic scene (tests/golden/scene_*.npz): u8 images, poses, intrinsics."""

    def __init__(self, name):
        g = load_golden(name)
        self.gray_u8 = g["gray_u8"]
        self.grays = [(x.astype(np.float32) / np.float32(255.0)) for x in self.gray_u8]
        self.colors = list(g["color_u8"])
        self.K = g["K"]
        self.R = g["R"]
        self.t = g["t"]
        self.depth_min = float(g["depth_min"])
        self.depth_max = float(g["depth_max"])
        self.gt_depth = g["gt_depth"]
        self.n = len(self.grays)
        self.H, self.W = self.grays[0].shape

    def poses(self):
        import amvs
        return {i: amvs.CameraPose(R=self.R[i].copy(), t=self.t[i].copy()) for i in range(self.n)}

    def K32(self):
        return self.K.astype(np.float32)

    def oracle_ctx(self, ref, srcs, patch, mode="exact"):
        from oracle import oracle
        return oracle.ViewContext(self.K32(), self.grays[ref], self.R[ref], self.t[ref],
                                  [self.grays[i] for i in srcs], [self.R[i] for i in srcs],
                                  [self.t[i] for i in srcs], patch, mode=mode)

    def engine(self, mode="exact"):
        import amvs
        eng = amvs.Engine(self.H, self.W, self.n, self.K32(), mode=mode)
        for i in range(self.n):
            eng.set_view(i, self.grays[i], self.R[i], self.t[i])
        return eng


@pytest.fixture(scope="session")
def scene_a():
    return GoldenScene("scene_a")


@pytest.fixture(scope="session")
def scene_b():
    return GoldenScene("scene_b")


@pytest.fixture(scope="session")
def scene_c():
    return GoldenScene("scene_c")


@pytest.fixture(scope="session")
def scene_d():
    return GoldenScene("scene_d")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import oracle
    oracle.build()


def assert_cost_close(got, want, atol, what=""):
    """Cost maps: identical +inf / NaN pattern, finite values within atol."""
    got = np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape
    assert np.array_equal(np.isposinf(got), np.isposinf(want)), f"{what}: +inf pattern differs"
    fin = np.isfinite(want) & np.isfinite(got)
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    err = np.abs(got[fin] - want[fin])
    assert err.size == 0 or err.max() <= atol, f"{what}: max abs err {err.max():.3e} > {atol:.1e}"


def chamfer(a, b):
    """Symmetric Chamfer distance of two point clouds (N,3), (M,3): the mean nearest-neighbour
    distance a -> b and b -> a, averaged (exact search, scipy cKDTree)."""
    from scipy.spatial import cKDTree
    a = np.asarray(a, np.float64).reshape(-1, 3)
    b = np.asarray(b, np.float64).reshape(-1, 3)
    if len(a) == 0 or len(b) == 0:
        return 0.0 if len(a) == len(b) else float("inf")
    dab = cKDTree(b).query(a)[0]
    dba = cKDTree(a).query(b)[0]
    return 0.5 * (float(dab.mean()) + float(dba.mean()))


# Acceptance thresholds shared by the CPU (oracle) and GPU (HIP) parity tests.
#   E2E_MIN_FRACTION  pixels whose depth is within 1e-3 relative of the reference's on identical RNG
#                     streams (>= 98 %)
#   CONF_HIST_TOL     largest difference between the confidence histograms (<= 1 %)
#   CHAMFER_TOL       symmetric Chamfer distance between our fused cloud and the cloud the reference
#                     fuses from ITS OWN maps, in scene units (depths are ~5, the fusion's voxel is
#                     0.01): 1e-3 = 0.02 % of the scene depth, a tenth of a voxel
E2E_MIN_FRACTION = 0.98
CONF_HIST_TOL = 0.01
CHAMFER_TOL = 1e-3


@pytest.fixture(autouse=True)
def _no_index_violations(request):
    """With an index-checked build of libamvs (AMVS_LIB=build/variants/libamvs_check.so; csrc/amvs_check.h) every
    GPU test also asserts that no kernel formed an out-of-range global index -- the device-side substitute for an
    address sanitizer.  With the shipped library this is a no-op."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    from amvs import _lib
    if not _lib.index_checks_enabled():
        return
    count, tu, line, index, extent = _lib.index_check(reset=True)
    assert count == 0, (f"{count} out-of-range global accesses; first in translation unit {tu} line {line}: "
                        f"index {index}, extent {extent}")
