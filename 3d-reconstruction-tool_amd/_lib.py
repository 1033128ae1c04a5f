"""ctypes binding of libamvs.so (the C ABI declared in include/amvs.h).

There is no CPU fallback: if the shared library is missing or no HIP device is
present, loading / context creation raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMVS_LIB selects an alternative build of the same ABI (A/B runs of kernel variants)
LIB_PATH = os.environ.get("AMVS_LIB") or os.path.join(_HERE, "libamvs.so")

AMVS_MAX_SRC = 6
MODES = {"default": 0, "exact": 1, "fast": 2}
SCHEDULES = {"auto": 0, "view-major": 1, "band-major": 2, "split": 3, "paired": 4}
SUPPORTED_PATCH_SIZES = tuple(range(3, 32, 2))        # compiled: 3 ... 29; 31 runs the run-time-k kernels

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int)


class PmParams(C.Structure):
    _fields_ = [("patch_size", C.c_int32), ("num_iterations", C.c_int32),
                ("num_samples", C.c_int32), ("tile_rows", C.c_int32), ("views_per_launch", C.c_int32),
                ("depth_min", C.c_float), ("depth_max", C.c_float),
                ("log_depth_scale", C.c_float), ("log_depth_min", C.c_float), ("mode", C.c_int32),
                ("schedule", C.c_int32), ("first_iteration", C.c_int32), ("flags", C.c_int32)]


PM_NO_CONFIDENCE = 1


class XpmParams(C.Structure):
    _fields_ = [("patch_size", C.c_int32), ("window_stride", C.c_int32), ("num_refine", C.c_int32),
                ("view_propagation", C.c_int32), ("depth_min", C.c_float), ("depth_max", C.c_float),
                ("log_depth_scale", C.c_float), ("log_depth_min", C.c_float),
                ("consistency_px", C.c_float), ("consistency_rel", C.c_float)]


class Timing(C.Structure):
    _fields_ = [("init_ms", C.c_double), ("sweep_ms", C.c_double), ("confidence_ms", C.c_double),
                ("sweep_launches", C.c_int64), ("pixel_hypotheses", C.c_int64)]


# name -> (restype, argtypes); every symbol include/amvs.h declares
SIGNATURES = {
    "amvs_version": (C.c_char_p, []),
    "amvs_last_error": (C.c_char_p, [C.c_void_p]),
    "amvs_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, f32p, f32p, C.POINTER(C.c_void_p)]),
    "amvs_destroy": (C.c_int, [C.c_void_p]),
    "amvs_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "amvs_sync": (C.c_int, [C.c_void_p]),
    "amvs_set_view": (C.c_int, [C.c_void_p, C.c_int, f32p, f32p, f32p]),
    "amvs_set_view_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, f32p, f32p]),
    "amvs_set_view_bgr8": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.c_int, f32p, f32p,
                                     C.POINTER(C.c_uint8)]),
    "amvs_patchmatch": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, C.POINTER(PmParams),
                                  C.c_uint64, f32p, f32p, f32p]),
    "amvs_patchmatch_device": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int,
                                         C.POINTER(PmParams), C.c_uint64,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "amvs_get_timing": (C.c_int, [C.c_void_p, C.POINTER(Timing)]),
    "amvs_sampling_mode": (C.c_int, [C.c_void_p]),
    "amvs_set_sampling": (C.c_int, [C.c_void_p, C.c_int]),
    "amvs_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "amvs_get_mode": (C.c_int, [C.c_void_p]),
    "amvs_set_sweep_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "amvs_set_split_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "amvs_set_step_tuning": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p]),
    "amvs_set_step_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "amvs_get_step_times": (C.c_int, [C.c_void_p, f32p, C.c_int, i32p]),
    "amvs_last_tile_rows": (C.c_int, [C.c_void_p]),
    "amvs_last_views_per_launch": (C.c_int, [C.c_void_p]),
    "amvs_plane_sweep": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, f32p, C.c_int, C.c_int,
                                   C.c_float, f32p, f32p]),
    "amvs_plane_sweep_device": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, f32p, C.c_int,
                                          C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "amvs_plane_sweep_batch": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, f32p, C.c_int, C.c_int, C.c_float]),
    "amvs_fetch_sweep_maps": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f32p, f32p]),
    "amvs_stereo_backproject": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint8),
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_float,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "amvs_cloud_knn_mean_distance": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "amvs_cloud_voxel_downsample": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_double, C.POINTER(C.c_int64)]),
    "amvs_cloud_take": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int64]),
    "amvs_knn_supported": (C.c_int, [C.c_int]),
    "amvs_xpm_init": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, C.POINTER(XpmParams), C.c_uint64,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "amvs_xpm_iterate": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, C.POINTER(XpmParams), C.c_int, C.c_uint64,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "amvs_xpm_step": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, C.POINTER(XpmParams), C.c_int, C.c_uint64, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "amvs_xpm_fetch_candidates": (C.c_int, [C.c_void_p, C.c_int, f32p, f32p]),
    "amvs_xpm_consistency": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, C.c_int, C.POINTER(XpmParams),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "amvs_eval_cost": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, f32p, f32p]),
    "amvs_sample_sources": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, f32p, f32p,
                                      C.POINTER(C.c_uint8)]),
    "amvs_confidence": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, f32p, f32p]),
    "amvs_propagate_step": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, f32p, f32p, f32p,
                                      C.c_int, C.c_int, C.c_float]),
    "amvs_refine_step": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_int, C.c_int, f32p, f32p, f32p,
                                   C.c_uint64, C.c_uint32, C.c_uint32,
                                   C.c_float, C.c_float, C.c_float, C.c_float]),
    "amvs_init_state": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_float,
                                  f32p, f32p, f32p]),
    "amvs_box_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_int, f32p, f32p]),
    "amvs_fuse_filter": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint8),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_float, C.c_int,
                                   C.POINTER(C.c_int64)]),
    "amvs_fuse_filter_views": (C.c_int, [C.c_void_p, C.c_int, i32p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                               C.POINTER(C.c_double), C.c_float, C.c_int, C.POINTER(C.c_int64)]),
    "amvs_stereo_backproject_views": (C.c_int, [C.c_void_p, C.c_int, i32p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.c_float, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "amvs_set_view_colors": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint8)]),
    "amvs_fetch_cloud": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint8)]),
    "amvs_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "amvs_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint8)]),
    "amvs_allgather_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "amvs_comm_destroy": (C.c_int, [C.c_void_p]),
    "amvs_write_ply": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int64]),
    "amvs_knn_mean_distance": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int, C.POINTER(C.c_double)]),
    "amvs_selftest_lean_math": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "amvs_index_check": (C.c_int, [C.POINTER(C.c_uint64), C.c_int]),
    "amvs_rng_fill": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int64, f32p, f32p]),
}

_lib = None


class AmvsError(RuntimeError):
    pass


def _share_torch_hip_runtime():
    """One HIP runtime per process.  libamvs.so links /opt/rocm's libamdhip64.so.7; a PyTorch-ROCm
    wheel bundles its own copy under the same SONAME, so whichever is loaded first serves both -- and
    PyTorch cannot see the GPU ("No HIP GPUs are available") once the system copy got in first.  When
    torch is installed its copy is therefore loaded first, whatever the import order (torch is located
    without importing it).  Without torch the system runtime is used as linked."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:  # noqa: BLE001  (best effort: the linked runtime still works on its own)
        pass


def load():
    """Load libamvs.so and bind every declared entry point (raises if absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AmvsError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(or `make -C 3d-reconstruction-tool_amd/csrc`). There is no CPU fallback.")
        _share_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def index_check(reset=True):
    """(violations, translation unit, source line, index, extent) of the index-checked build (include/amvs.h
    amvs_index_check); zeros from the shipped build."""
    r = (C.c_uint64 * 4)()
    load().amvs_index_check(r, int(bool(reset)))
    return int(r[0]), int(r[1]) >> 32, int(r[1]) & 0xFFFFFFFF, int(r[2]), int(r[3])


def index_checks_enabled():
    return b"+index-checks" in load().amvs_version()
