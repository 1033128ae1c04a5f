"""ctypes front-end of the CPU oracle (oracle/amvs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under 3d-reconstruction-tool_amd/ imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("AMVS_ORACLE_SO") or os.path.join(_HERE, "libamvs_oracle.so")   # (sanitizer build: oracle/Makefile)
_lib = None

f32p = C.POINTER(C.c_float)
u8p = C.POINTER(C.c_ubyte)


def build(force=False):
    src = os.path.join(_HERE, "amvs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_rng_fill.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, f32p, f32p]
        L.orc_expf_export.argtypes = [C.c_float]
        L.orc_expf_export.restype = C.c_float
        L.orc_box_stats.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p, f32p]
        L.orc_ncc.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        L.orc_ctx_create.argtypes = [C.c_int, C.c_int, C.c_int, f32p, f32p, f32p, f32p, f32p,
                                     C.c_int, f32p, f32p, f32p]
        L.orc_ctx_create.restype = C.c_void_p
        L.orc_ctx_destroy.argtypes = [C.c_void_p]
        L.orc_ctx_set_mode.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_set_mode.restype = C.c_int
        L.orc_sample.argtypes = [C.c_void_p, C.c_int, f32p, C.c_int, f32p, u8p]
        L.orc_patch_cost.argtypes = [C.c_void_p, f32p, f32p]
        L.orc_confidence.argtypes = [C.c_void_p, f32p, f32p]
        L.orc_propagate_step.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_int, C.c_int, C.c_float]
        L.orc_spatial_propagation.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_int, C.c_float]
        L.orc_refine_step.argtypes = [C.c_void_p, f32p, f32p, f32p, f32p, f32p,
                                      C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_init_state.argtypes = [C.c_int64, f32p, f32p, f32p, C.c_float, C.c_float,
                                     f32p, f32p, f32p]
        L.orc_patchmatch_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                          C.c_float, C.c_float, C.c_uint64, C.c_uint32,
                                          f32p, f32p, f32p]
        L.orc_plane_sweep.argtypes = [C.c_void_p, f32p, C.c_int, C.c_float, f32p, f32p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        _lib = L
        # a GPU box exposes every host core but grants a 16-CPU share: cap the OpenMP team
        # (AMVS_ORACLE_THREADS overrides), tiny test images gain nothing from more
        set_threads(int(os.environ.get("AMVS_ORACLE_THREADS", "0")) or min(os.cpu_count() or 1, 8))
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def rng_fill(seed, view, draw, n):
    u = np.empty(n, np.float32)
    nz = np.empty((n, 3), np.float32)
    lib().orc_rng_fill(seed, view, draw, n, u.ctypes.data_as(f32p), nz.ctypes.data_as(f32p))
    return u, nz


def expf(x):
    return lib().orc_expf_export(float(x))


def box_stats(img, k):
    img, p = _f(img)
    H, W = img.shape
    mean = np.empty((H, W), np.float32)
    var = np.empty((H, W), np.float32)
    lib().orc_box_stats(p, H, W, k, mean.ctypes.data_as(f32p), var.ctypes.data_as(f32p))
    return mean, var


def ncc(img1, img2, k, variant=0):
    img1, p1 = _f(img1)
    img2, p2 = _f(img2)
    H, W = img1.shape
    out = np.empty((H, W), np.float32)
    lib().orc_ncc(p1, p2, H, W, k, variant, out.ctypes.data_as(f32p))
    return out


class ViewContext:
    """One reference view + its source views (the arguments of
    PatchMatchMVS._compute_patch_cost, mvs_patchmatch.py:323-329)."""

    def __init__(self, K, ref, R_ref, t_ref, srcs, Rs, ts, patch, K_inv=None, mode="exact"):
        self.K, kp = _f(np.asarray(K, np.float32).reshape(3, 3))
        if K_inv is None:
            K_inv = np.linalg.inv(self.K)
        self.K_inv, kip = _f(np.asarray(K_inv, np.float32).reshape(3, 3))
        self.ref, rp = _f(ref)
        self.R_ref, rrp = _f(np.asarray(R_ref, np.float32).reshape(3, 3))
        self.t_ref, trp = _f(np.asarray(t_ref, np.float32).reshape(3))
        self.srcs, sp = _f(np.stack([np.asarray(s, np.float32) for s in srcs]))
        self.Rs, rsp = _f(np.stack([np.asarray(r, np.float32).reshape(3, 3) for r in Rs]))
        self.ts, tsp = _f(np.stack([np.asarray(t, np.float32).reshape(3) for t in ts]))
        self.H, self.W = self.ref.shape
        self.S = self.srcs.shape[0]
        self.patch = patch
        self._h = lib().orc_ctx_create(self.H, self.W, patch, kp, kip, rp, rrp, trp,
                                       self.S, sp, rsp, tsp)
        if not self._h:
            raise RuntimeError("orc_ctx_create failed")
        self.mode = "exact"
        if mode != "exact":
            self.set_mode(mode)

    def set_mode(self, mode):
        """'exact' (the reference's arithmetic) or 'fast' (the HIP backend's tolerance mode,
        8-bit images only); applies to every later call."""
        if mode not in ("exact", "fast"):
            raise ValueError(mode)
        if lib().orc_ctx_set_mode(self._h, 1 if mode == "fast" else 0) != 0:
            raise ValueError("fast mode needs 8-bit images (every pixel exactly code/255)")
        self.mode = mode

    def close(self):
        if self._h:
            lib().orc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _out(self):
        return np.empty((self.H, self.W), np.float32)

    def sample(self, s, depth, bounds=0):
        depth, dp = _f(depth)
        out = self._out()
        valid = np.empty((self.H, self.W), np.uint8)
        lib().orc_sample(self._h, s, dp, bounds, out.ctypes.data_as(f32p), valid.ctypes.data_as(u8p))
        return out, valid.astype(bool)

    def patch_cost(self, depth):
        depth, dp = _f(depth)
        out = self._out()
        lib().orc_patch_cost(self._h, dp, out.ctypes.data_as(f32p))
        return out

    def confidence(self, depth):
        depth, dp = _f(depth)
        out = self._out()
        lib().orc_confidence(self._h, dp, out.ctypes.data_as(f32p))
        return out

    def _state(self, depth, normal, cost):
        d = np.array(depth, np.float32, order="C", copy=True)
        n = np.array(normal, np.float32, order="C", copy=True)
        c = np.array(cost, np.float32, order="C", copy=True)
        return d, n, c

    def propagate_step(self, depth, normal, cost, oy, ox, depth_min):
        d, n, c = self._state(depth, normal, cost)
        lib().orc_propagate_step(self._h, d.ctypes.data_as(f32p), n.ctypes.data_as(f32p),
                                 c.ctypes.data_as(f32p), oy, ox, depth_min)
        return d, n, c

    def spatial_propagation(self, depth, normal, cost, forward, depth_min):
        d, n, c = self._state(depth, normal, cost)
        lib().orc_spatial_propagation(self._h, d.ctypes.data_as(f32p), n.ctypes.data_as(f32p),
                                      c.ctypes.data_as(f32p), int(bool(forward)), depth_min)
        return d, n, c

    def refine_step(self, depth, normal, cost, u, nz, depth_range, normal_range, dmin, dmax):
        d, n, c = self._state(depth, normal, cost)
        u, up = _f(u)
        nz, nzp = _f(nz)
        lib().orc_refine_step(self._h, d.ctypes.data_as(f32p), n.ctypes.data_as(f32p),
                              c.ctypes.data_as(f32p), up, nzp, depth_range, normal_range, dmin, dmax)
        return d, n, c

    def patchmatch(self, iters, samples, depth_min, depth_max, seed, view):
        log_min = np.log(float(depth_min))
        log_max = np.log(float(depth_max))
        d = self._out()
        n = np.empty((self.H, self.W, 3), np.float32)
        conf = self._out()
        lib().orc_patchmatch_view(self._h, iters, samples, depth_min, depth_max,
                                  np.float32(log_max - log_min), np.float32(log_min),
                                  seed, view, d.ctypes.data_as(f32p), n.ctypes.data_as(f32p),
                                  conf.ctypes.data_as(f32p))
        return d, n, conf

    def plane_sweep(self, depths, thresh):
        depths, dp = _f(depths)
        d = self._out()
        conf = self._out()
        lib().orc_plane_sweep(self._h, dp, len(depths), thresh, d.ctypes.data_as(f32p),
                              conf.ctypes.data_as(f32p))
        return d, conf


def init_state(u, n0, n1, depth_min, depth_max):
    u, up = _f(u)
    n0, n0p = _f(n0)
    n1, n1p = _f(n1)
    H, W = u.shape
    log_min = np.log(float(depth_min))
    log_max = np.log(float(depth_max))
    d = np.empty((H, W), np.float32)
    n = np.empty((H, W, 3), np.float32)
    c = np.empty((H, W), np.float32)
    lib().orc_init_state(H * W, up, n0p, n1p, np.float32(log_max - log_min), np.float32(log_min),
                         d.ctypes.data_as(f32p), n.ctypes.data_as(f32p), c.ctypes.data_as(f32p))
    return d, n, c


def set_threads(n):
    lib().orc_set_threads(int(n))


def num_threads():
    return lib().orc_num_threads()
