"""Engine: one device context holding every view of a scene (thin wrapper of the C ABI)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import AmvsError, PmParams, Timing, XpmParams, f32p, i32p


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _p(a):
    return a.ctypes.data_as(f32p)


def _ids(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


def make_pm_params(patch_size, num_iterations, num_samples, depth_min, depth_max, tile_rows=0,
                   views_per_launch=0, mode="default", schedule="auto", first_iteration=0, confidence=True):
    """amvs_pm_params with the log-range formed in double as mvs_patchmatch.py:268-271 does.
    mode: "default" (the engine's), "exact" or "fast" (include/amvs.h AMVS_MODE_*).
    first_iteration > 0 continues the previous call's sweep (include/amvs.h); confidence=False skips the
    confidence pass of this call."""
    log_min = np.log(float(depth_min))
    log_max = np.log(float(depth_max))
    return PmParams(int(patch_size), int(num_iterations), int(num_samples), int(tile_rows),
                    int(views_per_launch), float(depth_min), float(depth_max),
                    float(np.float32(log_max - log_min)), float(np.float32(log_min)), _lib.MODES[mode],
                    _lib.SCHEDULES[schedule], int(first_iteration), 0 if confidence else _lib.PM_NO_CONFIDENCE)


def make_xpm_params(patch_size, depth_min, depth_max, window_stride=2, num_refine=2, view_propagation=True,
                    consistency_px=1.0, consistency_rel=0.01):
    """amvs_xpm_params of the extended mode (include/amvs.h)."""
    log_min = np.log(float(depth_min))
    log_max = np.log(float(depth_max))
    return XpmParams(int(patch_size), int(window_stride), int(num_refine), int(bool(view_propagation)),
                     float(depth_min), float(depth_max), float(np.float32(log_max - log_min)), float(np.float32(log_min)),
                     float(consistency_px), float(consistency_rel))


class Engine:
    def __init__(self, H, W, n_views, K, K_inv=None, device=0, mode="exact"):
        self._lib = _lib.load()
        self.H, self.W, self.n_views, self.device = int(H), int(W), int(n_views), int(device)
        self.K = _f32(np.asarray(K, np.float32).reshape(3, 3))
        if K_inv is None:
            # float32 inverse, as torch.inverse(K) in mvs_patchmatch.py:237-238
            K_inv = np.linalg.inv(self.K)
        self.K_inv = _f32(np.asarray(K_inv, np.float32).reshape(3, 3))
        h = C.c_void_p()
        rc = self._lib.amvs_create(self.device, self.H, self.W, self.n_views, _p(self.K), _p(self.K_inv),
                                   C.byref(h))
        if rc != 0:
            raise AmvsError(f"amvs_create failed ({rc}): {self._lib.amvs_last_error(None).decode()}")
        self._h = h
        if mode != "exact":
            self.set_mode(mode)

    def reusable_for(self, H, W, n_views, K, device, mode="exact"):
        """True if this (open) context was created for exactly this problem: the classes then upload the next call's
        views into it instead of destroying and re-creating it (some twenty device buffers: 1.5 ms a call)."""
        return (self._h is not None and (self.H, self.W, self.n_views, self.device) == (int(H), int(W), int(n_views), int(device))
                and np.array_equal(self.K, np.asarray(K, np.float32).reshape(3, 3)) and self.mode() == mode)

    # -- arithmetic mode ------------------------------------------------------
    def set_mode(self, mode):
        """'exact' (bit-identical to the reference's float32 chain) or 'fast' (tolerance mode,
        8-bit images only): applies to every later sweep call of this engine."""
        self._chk(self._lib.amvs_set_mode(self._h, _lib.MODES[mode]))

    def mode(self):
        return {1: "exact", 2: "fast"}[int(self._lib.amvs_get_mode(self._h))]

    def set_sampling(self, force_f32):
        self._chk(self._lib.amvs_set_sampling(self._h, int(bool(force_f32))))

    def set_sweep_tuning(self, tile_rows=0, planes_per_wave=0):
        self._chk(self._lib.amvs_set_sweep_tuning(self._h, int(tile_rows), int(planes_per_wave)))

    def set_split_tuning(self, groups=0, sample_rows=0, sample_lds_bytes=0):
        """Split schedule (schedule="split"): view groups pipelined against each other, rows per strip
        of the sampling kernel, unused LDS bytes per sampling workgroup; 0 = automatic."""
        self._chk(self._lib.amvs_set_split_tuning(self._h, int(groups), int(sample_rows), int(sample_lds_bytes)))

    def set_step_tuning(self, tile_rows=None, wgs_per_cu=None):
        """Launch shape of the PatchMatch sweep steps by iteration: lists of (propagation, refinement)
        pairs, one per iteration (0 = automatic; later iterations repeat the last pair); both None
        clears the table.  Performance only."""
        n = max(len(tile_rows or []), len(wgs_per_cu or []))
        if n == 0:
            self._chk(self._lib.amvs_set_step_tuning(self._h, 0, None, None))
            return

        def table(t):
            t = list(t or [])
            t = t + [t[-1] if t else (0, 0)] * (n - len(t))
            return np.ascontiguousarray(np.asarray(t, dtype=np.int32).reshape(n, 2))
        r, w = table(tile_rows), table(wgs_per_cu)
        self._chk(self._lib.amvs_set_step_tuning(self._h, n, r.ctypes.data_as(i32p), w.ctypes.data_as(i32p)))

    def set_step_timing(self, enable=True):
        self._chk(self._lib.amvs_set_step_timing(self._h, int(bool(enable))))

    def step_times(self):
        """Device time (ms) of every sweep launch of the last PatchMatch call (set_step_timing)."""
        n = C.c_int(0)
        self._chk(self._lib.amvs_get_step_times(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.float32)
        self._chk(self._lib.amvs_get_step_times(self._h, _p(out), out.size, C.byref(n)))
        return out[: n.value]

    # -- native exchange (RCCL behind the C ABI; the classes use torch.distributed) ------------
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 creates it, the caller hands it to the other ranks)."""
        lib = _lib.load()
        buf = (C.c_uint8 * 128)()
        rc = lib.amvs_comm_unique_id(buf)
        if rc != 0:
            raise AmvsError(f"amvs_comm_unique_id failed ({rc}): {lib.amvs_last_error(None).decode()}")
        return bytes(buf)

    def comm_init(self, rank, world, unique_id):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self._lib.amvs_comm_init(self._h, int(rank), int(world), buf))

    def allgather_maps(self, local_ptr, full_ptr, floats_per_rank):
        """ncclAllGather of floats_per_rank float32 per rank, device pointers, on the engine's stream."""
        self._chk(self._lib.amvs_allgather_maps(self._h, C.c_void_p(local_ptr), C.c_void_p(full_ptr), int(floats_per_rank)))

    def comm_destroy(self):
        self._chk(self._lib.amvs_comm_destroy(self._h))

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.amvs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise AmvsError(f"amvs call failed ({rc}): {self._lib.amvs_last_error(self._h).decode()}")

    def sync(self):
        self._chk(self._lib.amvs_sync(self._h))

    def set_stream(self, stream_ptr):
        self._chk(self._lib.amvs_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    # -- scene -------------------------------------------------------------
    def set_view(self, view, gray, R, t):
        gray = _f32(gray, (self.H, self.W))
        R = _f32(np.asarray(R, np.float64).astype(np.float32).reshape(3, 3))
        t = _f32(np.asarray(t, np.float64).astype(np.float32).reshape(3))
        self._chk(self._lib.amvs_set_view(self._h, int(view), _p(gray), _p(R), _p(t)))

    def set_view_bgr8(self, view, image_bgr_u8, R, t, want_color=True):
        """Upload the 8-bit BGR image as it is and prepare it on the device (resize to the engine's
        H x W, BGR -> gray, /255: mvs_patchmatch.py:167-191).  Returns the resized colour image
        (H, W, 3) uint8, or None."""
        img = np.ascontiguousarray(image_bgr_u8, dtype=np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("expected an (h, w, 3) uint8 BGR image")
        R = _f32(np.asarray(R, np.float64).astype(np.float32).reshape(3, 3))
        t = _f32(np.asarray(t, np.float64).astype(np.float32).reshape(3))
        out = np.empty((self.H, self.W, 3), np.uint8) if want_color else None
        self._chk(self._lib.amvs_set_view_bgr8(
            self._h, int(view), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[0], img.shape[1], _p(R), _p(t),
            out.ctypes.data_as(C.POINTER(C.c_uint8)) if want_color else None))
        return out

    def set_view_colors(self, view, image_bgr_u8):
        """Keep the (H, W, 3) uint8 BGR image of a view on the device for fuse_filter_views /
        stereo_backproject_views (views prepared with set_view_bgr8 have theirs already)."""
        img = np.ascontiguousarray(image_bgr_u8, dtype=np.uint8)
        if img.shape != (self.H, self.W, 3):
            raise ValueError(f"expected a ({self.H}, {self.W}, 3) uint8 BGR image")
        self._chk(self._lib.amvs_set_view_colors(self._h, int(view), img.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_view_device(self, view, gray_ptr, R, t):
        R = _f32(np.asarray(R, np.float64).astype(np.float32).reshape(3, 3))
        t = _f32(np.asarray(t, np.float64).astype(np.float32).reshape(3))
        self._chk(self._lib.amvs_set_view_device(self._h, int(view), C.c_void_p(gray_ptr), _p(R), _p(t)))

    # -- PatchMatch --------------------------------------------------------
    def patchmatch(self, ref_ids, src_ids, params, seed):
        """Returns depth (n,H,W), normal (n,H,W,3), confidence (n,H,W) as numpy arrays."""
        ref, refp = _ids(ref_ids)
        src, srcp = _ids(src_ids)
        n = ref.shape[0]
        src = src.reshape(n, -1)
        depth = np.empty((n, self.H, self.W), np.float32)
        normal = np.empty((n, self.H, self.W, 3), np.float32)
        conf = np.empty((n, self.H, self.W), np.float32)
        self._chk(self._lib.amvs_patchmatch(self._h, n, refp, srcp, src.shape[1], C.byref(params),
                                            int(seed), _p(depth), _p(normal), _p(conf)))
        return depth, normal, conf

    def patchmatch_device(self, ref_ids, src_ids, params, seed, depth_ptr, normal_ptr, conf_ptr):
        ref, refp = _ids(ref_ids)
        src, srcp = _ids(src_ids)
        n = ref.shape[0]
        src = src.reshape(n, -1)
        self._chk(self._lib.amvs_patchmatch_device(self._h, n, refp, srcp, src.shape[1], C.byref(params),
                                                   int(seed), C.c_void_p(depth_ptr),
                                                   C.c_void_p(normal_ptr), C.c_void_p(conf_ptr)))

    def timing(self):
        t = Timing()
        self._chk(self._lib.amvs_get_timing(self._h, C.byref(t)))
        return {"init_ms": t.init_ms, "sweep_ms": t.sweep_ms, "confidence_ms": t.confidence_ms,
                "sweep_launches": t.sweep_launches, "pixel_hypotheses": t.pixel_hypotheses}

    def sampling_mode(self):
        """'u8-pairs' when the packed 8-bit maps are sampled, 'f32' otherwise."""
        return "u8-pairs" if self._lib.amvs_sampling_mode(self._h) else "f32"

    def last_tile_rows(self):
        return int(self._lib.amvs_last_tile_rows(self._h))

    def last_views_per_launch(self):
        return int(self._lib.amvs_last_views_per_launch(self._h))

    # -- plane sweep -------------------------------------------------------
    def plane_sweep(self, ref, nbr_ids, depths, patch_size, thresh):
        nbr, nbrp = _ids(nbr_ids)
        depths = _f32(np.asarray(depths, np.float64).astype(np.float32))
        d = np.empty((self.H, self.W), np.float32)
        conf = np.empty((self.H, self.W), np.float32)
        self._chk(self._lib.amvs_plane_sweep(self._h, int(ref), nbrp, nbr.size, _p(depths), depths.size,
                                             int(patch_size), float(thresh), _p(d), _p(conf)))
        return d, conf

    def plane_sweep_device(self, ref_ids, nbr_ids, depths, patch_size, thresh, depth_ptr, conf_ptr):
        ref, refp = _ids(ref_ids)
        nbr, nbrp = _ids(nbr_ids)
        n = ref.shape[0]
        nbr = nbr.reshape(n, -1)
        depths = _f32(np.asarray(depths, np.float64).astype(np.float32))
        self._chk(self._lib.amvs_plane_sweep_device(self._h, n, refp, nbrp, nbr.shape[1], _p(depths),
                                                    depths.size, int(patch_size), float(thresh),
                                                    C.c_void_p(depth_ptr), C.c_void_p(conf_ptr)))

    def plane_sweep_batch(self, ref_ids, nbr_ids, depths, patch_size, thresh):
        """All reference views in one launch; the maps stay in the context (fetch_sweep_maps,
        stereo_backproject(resident=True))."""
        ref, refp = _ids(ref_ids)
        nbr, nbrp = _ids(nbr_ids)
        n = ref.shape[0]
        nbr = nbr.reshape(n, -1)
        depths = _f32(np.asarray(depths, np.float64).astype(np.float32))
        self._chk(self._lib.amvs_plane_sweep_batch(self._h, n, refp, nbrp, nbr.shape[1], _p(depths), depths.size,
                                                   int(patch_size), float(thresh)))
        return n

    def fetch_sweep_maps(self, first, count):
        d = np.empty((count, self.H, self.W), np.float32)
        c = np.empty((count, self.H, self.W), np.float32)
        self._chk(self._lib.amvs_fetch_sweep_maps(self._h, int(first), int(count), _p(d), _p(c)))
        return d, c

    # -- extended mode ----------------------------------------------------------
    def _xpm_call(self, fn, ref_ids, src_ids, params, extra, ptrs):
        ref, refp = _ids(ref_ids)
        src, srcp = _ids(src_ids)
        n = ref.shape[0]
        src = src.reshape(n, -1)
        self._chk(fn(self._h, n, refp, srcp, src.shape[1], C.byref(params), *extra, *[C.c_void_p(p) for p in ptrs]))

    def xpm_init(self, ref_ids, src_ids, params, seed, depth_ptr, normal_ptr, cost_ptr):
        self._xpm_call(self._lib.amvs_xpm_init, ref_ids, src_ids, params, (int(seed),), (depth_ptr, normal_ptr, cost_ptr))

    def xpm_iterate(self, ref_ids, src_ids, params, iteration, seed, depth_ptr, normal_ptr, cost_ptr,
                    snapshot_depth_ptr=0, snapshot_normal_ptr=0):
        """One iteration (view candidates, red and black half sweeps).  snapshot_*: device copies of the
        depth / normal maps the view candidates read (0 = the live maps); pass them when the views of one
        iteration are split over several calls."""
        self._xpm_call(self._lib.amvs_xpm_iterate, ref_ids, src_ids, params, (int(iteration), int(seed)),
                       (depth_ptr, normal_ptr, cost_ptr, snapshot_depth_ptr or None, snapshot_normal_ptr or None))

    XPM_PHASES = {"candidates": 0, "red": 1, "black": 2, "eval": 3}

    def xpm_step(self, ref_ids, src_ids, params, iteration, seed, phase, depth_ptr, normal_ptr, cost_ptr,
                 snapshot_depth_ptr=0, snapshot_normal_ptr=0, cost_out_ptr=0):
        """One phase of an iteration ("candidates", "red", "black") or the test hook "eval" (cost of the
        current planes into cost_out_ptr)."""
        ref, refp = _ids(ref_ids)
        src, srcp = _ids(src_ids)
        n = ref.shape[0]
        src = src.reshape(n, -1)
        self._chk(self._lib.amvs_xpm_step(self._h, n, refp, srcp, src.shape[1], C.byref(params), int(iteration), int(seed),
                                          self.XPM_PHASES[phase], C.c_void_p(depth_ptr), C.c_void_p(normal_ptr),
                                          C.c_void_p(cost_ptr), C.c_void_p(snapshot_depth_ptr or None),
                                          C.c_void_p(snapshot_normal_ptr or None), C.c_void_p(cost_out_ptr or None)))

    def xpm_fetch_candidates(self, n_ref):
        d = np.empty((n_ref, self.H, self.W), np.float32)
        n = np.empty((n_ref, self.H, self.W, 3), np.float32)
        self._chk(self._lib.amvs_xpm_fetch_candidates(self._h, int(n_ref), _p(d), _p(n)))
        return d, n

    def xpm_consistency(self, ref_ids, src_ids, params, depth_ptr, normal_ptr, cost_ptr, conf_ptr):
        self._xpm_call(self._lib.amvs_xpm_consistency, ref_ids, src_ids, params, (),
                       (depth_ptr, normal_ptr, cost_ptr, conf_ptr))

    # -- stereo post-steps on the device --------------------------------------
    def stereo_backproject(self, colors_bgr, K_inv64, poses, min_confidence, depth=None, conf=None, fetch=False,
                           device_ptrs=None):
        """dense_stereo.py:407-437 for all views at once.  depth / conf None: the resident maps of the
        last plane_sweep_batch; device_ptrs=(depth_ptr, conf_ptr): maps in caller device memory, (n, H*W)
        float32 each.  Returns (per-view point counts, total); the cloud stays on the device
        (fetch=True additionally returns points, colours)."""
        cols = np.ascontiguousarray(colors_bgr, dtype=np.uint8)
        n = cols.shape[0]
        cols = cols.reshape(n, self.H, self.W, 3)
        kinv = np.ascontiguousarray(K_inv64, dtype=np.float64).reshape(9)
        pp = np.ascontiguousarray(np.stack([np.concatenate([np.asarray(R, np.float64).reshape(9),
                                                            np.asarray(t, np.float64).reshape(3)])
                                            for R, t in poses]))
        if device_ptrs is not None:
            dptr, cptr, where = C.c_void_p(device_ptrs[0]), C.c_void_p(device_ptrs[1]), 1
        elif depth is None:
            dptr, cptr, where = C.c_void_p(0), C.c_void_p(0), 2
        else:
            depth, conf = _f32(depth), _f32(conf)
            dptr, cptr, where = depth.ctypes.data_as(C.c_void_p), conf.ctypes.data_as(C.c_void_p), 0
        per = (C.c_int64 * n)()
        total = C.c_int64(0)
        self._chk(self._lib.amvs_stereo_backproject(
            self._h, n, dptr, cptr, where, cols.ctypes.data_as(C.POINTER(C.c_uint8)),
            kinv.ctypes.data_as(C.POINTER(C.c_double)), pp.ctypes.data_as(C.POINTER(C.c_double)),
            float(min_confidence), per, C.byref(total)))
        counts = [int(x) for x in per]
        if not fetch:
            return counts, int(total.value)
        return (counts, int(total.value)) + self.fetch_cloud(int(total.value))

    def fetch_cloud(self, m):
        pts = np.empty((m, 3), np.float64)
        rgb = np.empty((m, 3), np.uint8)
        if m:
            self._chk(self._lib.amvs_fetch_cloud(self._h, pts.ctypes.data_as(C.POINTER(C.c_double)),
                                                 rgb.ctypes.data_as(C.POINTER(C.c_uint8))))
        return pts, rgb

    def cloud_knn_mean_distance(self, n_points, k=20):
        out = np.empty(int(n_points), np.float64)
        self._chk(self._lib.amvs_cloud_knn_mean_distance(self._h, int(k), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def cloud_voxel_downsample(self, voxel_size, keep_mask=None):
        """dense_stereo.py:475-492 on the resident cloud (after the optional boolean keep mask);
        returns the new point count."""
        cnt = C.c_int64(0)
        if keep_mask is None:
            mp = None
        else:
            km = np.ascontiguousarray(keep_mask, dtype=np.uint8)
            mp = km.ctypes.data_as(C.POINTER(C.c_uint8))
        self._chk(self._lib.amvs_cloud_voxel_downsample(self._h, mp, float(voxel_size), C.byref(cnt)))
        return int(cnt.value)

    def cloud_take(self, indices):
        """The resident cloud <- its rows `indices` in that order (points[chosen], dense_stereo.py:449-455)."""
        idx = np.ascontiguousarray(indices, dtype=np.int64)
        self._chk(self._lib.amvs_cloud_take(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), idx.size))
        return int(idx.size)

    def knn_supported(self, k):
        return bool(self._lib.amvs_knn_supported(int(k)))

    # -- single steps (parity tests) ----------------------------------------
    def eval_cost(self, ref, src_ids, patch_size, depth):
        src, srcp = _ids(src_ids)
        depth = _f32(depth, (self.H, self.W))
        out = np.empty((self.H, self.W), np.float32)
        self._chk(self._lib.amvs_eval_cost(self._h, int(ref), srcp, src.size, int(patch_size), _p(depth), _p(out)))
        return out

    def sample_sources(self, ref, src_ids, patch_size, depth, bounds=0):
        """(sampled (S,H,W) float32, valid (S,H,W) bool) of the stage before the box filter; in
        fast mode the samples are in 8-bit code units (gray * 255)."""
        src, srcp = _ids(src_ids)
        depth = _f32(depth, (self.H, self.W))
        out = np.empty((src.size, self.H, self.W), np.float32)
        bits = np.empty((self.H, self.W), np.uint8)
        self._chk(self._lib.amvs_sample_sources(self._h, int(ref), srcp, src.size, int(patch_size), int(bounds),
                                                _p(depth), _p(out), bits.ctypes.data_as(C.POINTER(C.c_uint8))))
        valid = np.stack([(bits >> s) & 1 for s in range(src.size)]).astype(bool)
        return out, valid

    def confidence(self, ref, src_ids, patch_size, depth):
        src, srcp = _ids(src_ids)
        depth = _f32(depth, (self.H, self.W))
        out = np.empty((self.H, self.W), np.float32)
        self._chk(self._lib.amvs_confidence(self._h, int(ref), srcp, src.size, int(patch_size), _p(depth), _p(out)))
        return out

    def _state(self, depth, normal, cost):
        d = np.array(depth, np.float32, order="C", copy=True).reshape(self.H, self.W)
        n = np.array(normal, np.float32, order="C", copy=True).reshape(self.H, self.W, 3)
        c = np.array(cost, np.float32, order="C", copy=True).reshape(self.H, self.W)
        return d, n, c

    def propagate_step(self, ref, src_ids, patch_size, depth, normal, cost, oy, ox, depth_min):
        src, srcp = _ids(src_ids)
        d, n, c = self._state(depth, normal, cost)
        self._chk(self._lib.amvs_propagate_step(self._h, int(ref), srcp, src.size, int(patch_size),
                                                _p(d), _p(n), _p(c), int(oy), int(ox), float(depth_min)))
        return d, n, c

    def refine_step(self, ref, src_ids, patch_size, depth, normal, cost, seed, stream_view, draw,
                    depth_range, normal_range, depth_min, depth_max):
        src, srcp = _ids(src_ids)
        d, n, c = self._state(depth, normal, cost)
        self._chk(self._lib.amvs_refine_step(self._h, int(ref), srcp, src.size, int(patch_size),
                                             _p(d), _p(n), _p(c), int(seed), int(stream_view), int(draw),
                                             float(depth_range), float(normal_range),
                                             float(depth_min), float(depth_max)))
        return d, n, c

    def init_state(self, seed, stream_view, depth_min, depth_max):
        p = make_pm_params(7, 0, 0, depth_min, depth_max)
        d = np.empty((self.H, self.W), np.float32)
        n = np.empty((self.H, self.W, 3), np.float32)
        c = np.empty((self.H, self.W), np.float32)
        self._chk(self._lib.amvs_init_state(self._h, int(seed), int(stream_view), p.log_depth_scale,
                                            p.log_depth_min, _p(d), _p(n), _p(c)))
        return d, n, c

    def box_stats(self, view, patch_size):
        m = np.empty((self.H, self.W), np.float32)
        v = np.empty((self.H, self.W), np.float32)
        self._chk(self._lib.amvs_box_stats(self._h, int(view), int(patch_size), _p(m), _p(v)))
        return m, v

    # -- fusion / filter ----------------------------------------------------
    def fuse_filter(self, depth, conf, colors_bgr, K_inv64, poses, min_views, do_filter=True, device_ptrs=None):
        """Device fusion (+ filter): depth/conf (n,H,W) float32 host arrays -- or, with
        device_ptrs=(depth_ptr, conf_ptr, n), maps already resident on the GPU --, colors (n,H,W,3)
        uint8 BGR, poses = list of (R, t) float64.  Returns (points (M,3) float64, colors (M,3)
        uint8 RGB, raw_count)."""
        if device_ptrs is None:
            depth = _f32(depth)
            conf = _f32(conf)
            n = depth.shape[0]
            dptr, cptr, on_dev = depth.ctypes.data_as(C.c_void_p), conf.ctypes.data_as(C.c_void_p), 0
        else:
            dptr, cptr, n = C.c_void_p(device_ptrs[0]), C.c_void_p(device_ptrs[1]), int(device_ptrs[2])
            on_dev = 1
        cols = np.ascontiguousarray(colors_bgr, dtype=np.uint8).reshape(n, self.H, self.W, 3)
        kinv = np.ascontiguousarray(K_inv64, dtype=np.float64).reshape(9)
        pp = np.ascontiguousarray(np.stack([np.concatenate([np.asarray(R, np.float64).reshape(9),
                                                            np.asarray(t, np.float64).reshape(3)])
                                            for R, t in poses]))
        counts = (C.c_int64 * 2)()
        self._chk(self._lib.amvs_fuse_filter(
            self._h, n, dptr, cptr, on_dev,
            cols.ctypes.data_as(C.POINTER(C.c_uint8)), kinv.ctypes.data_as(C.POINTER(C.c_double)),
            pp.ctypes.data_as(C.POINTER(C.c_double)), float(min_views), int(bool(do_filter)), counts))
        m = int(counts[1])
        pts = np.empty((m, 3), np.float64)
        rgb = np.empty((m, 3), np.uint8)
        if m:
            self._chk(self._lib.amvs_fetch_cloud(self._h, pts.ctypes.data_as(C.POINTER(C.c_double)),
                                                 rgb.ctypes.data_as(C.POINTER(C.c_uint8))))
        return pts, rgb, int(counts[0])

    def stereo_backproject_views(self, view_ids, K_inv64, poses, min_confidence):
        """stereo_backproject for the resident maps of the last plane_sweep_batch with the colour images
        set_view_bgr8 left on the device (map j belongs to view view_ids[j]).  Returns (per-view point
        counts, total); the cloud stays on the device."""
        ids, idp = _ids(view_ids)
        n = ids.shape[0]
        kinv = np.ascontiguousarray(K_inv64, dtype=np.float64).reshape(9)
        pp = np.ascontiguousarray(np.stack([np.concatenate([np.asarray(R, np.float64).reshape(9),
                                                            np.asarray(t, np.float64).reshape(3)])
                                            for R, t in poses]))
        per = (C.c_int64 * n)()
        total = C.c_int64(0)
        self._chk(self._lib.amvs_stereo_backproject_views(
            self._h, n, idp, kinv.ctypes.data_as(C.POINTER(C.c_double)), pp.ctypes.data_as(C.POINTER(C.c_double)),
            float(min_confidence), per, C.byref(total)))
        return [int(x) for x in per], int(total.value)

    def fuse_filter_views(self, view_ids, depth_ptr, conf_ptr, K_inv64, poses, min_views, do_filter=True):
        """Device fusion (+ filter) of resident maps whose colour images are resident as well
        (set_view_bgr8): map j belongs to view view_ids[j].  Returns (points, colors RGB, raw_count)."""
        ids, idp = _ids(view_ids)
        n = ids.shape[0]
        kinv = np.ascontiguousarray(K_inv64, dtype=np.float64).reshape(9)
        pp = np.ascontiguousarray(np.stack([np.concatenate([np.asarray(R, np.float64).reshape(9),
                                                            np.asarray(t, np.float64).reshape(3)])
                                            for R, t in poses]))
        counts = (C.c_int64 * 2)()
        self._chk(self._lib.amvs_fuse_filter_views(
            self._h, n, idp, C.c_void_p(depth_ptr), C.c_void_p(conf_ptr), kinv.ctypes.data_as(C.POINTER(C.c_double)),
            pp.ctypes.data_as(C.POINTER(C.c_double)), float(min_views), int(bool(do_filter)), counts))
        m = int(counts[1])
        pts = np.empty((m, 3), np.float64)
        rgb = np.empty((m, 3), np.uint8)
        if m:
            self._chk(self._lib.amvs_fetch_cloud(self._h, pts.ctypes.data_as(C.POINTER(C.c_double)),
                                                 rgb.ctypes.data_as(C.POINTER(C.c_uint8))))
        return pts, rgb, int(counts[0])

    def knn_mean_distance(self, points, k=20):
        """Mean distance of every point to its k-1 nearest other points, bit-identical to
        np.mean(NearestNeighbors(n_neighbors=k).fit(p).kneighbors(p)[0][:, 1:], axis=1)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty(pts.shape[0], np.float64)
        self._chk(self._lib.amvs_knn_mean_distance(self._h, pts.ctypes.data_as(C.POINTER(C.c_double)),
                                                   pts.shape[0], int(k), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def selftest_lean_math(self):
        """(reciprocal mismatches, sqrt mismatches) against IEEE over all 2^32 float patterns."""
        out = (C.c_uint64 * 2)()
        self._chk(self._lib.amvs_selftest_lean_math(self._h, out))
        return int(out[0]), int(out[1])

    def rng_fill(self, seed, stream_view, draw, n):
        u = np.empty(n, np.float32)
        nz = np.empty((n, 3), np.float32)
        self._chk(self._lib.amvs_rng_fill(self._h, int(seed), int(stream_view), int(draw), int(n), _p(u), _p(nz)))
        return u, nz
