"""Plane-sweep dense stereo: host side of the MI355X backend.

Mirrors the call surface of the reference's src/core/dense_stereo.py
(`DenseStereoReconstructor(camera, scale, num_depths, patch_size, min_views,
consistency_thresh).reconstruct(images, poses, max_pairs) -> (points, colors)`,
reference :32-37, :61-63).  The per-pixel plane sweep (`_plane_sweep_torch`,
reference :222-316) runs in the gfx950 plane_sweep kernel, which keeps a running
best plane per pixel instead of the reference's (D,H,W) vote volume.

`reconstruct` sweeps ALL reference views in one batched launch (the reference loops over
them, :105-130), keeps the maps on the GPU, back-projects them there (:407-437), runs the
neighbour search of the outlier filter there (:456-460) and the voxel down-sampling there
(:475-492); only the per-point statistic of the outlier filter and the final cloud cross
PCIe.  Under an initialised torch.distributed process group the reference views are
sharded over the ranks (one process per GPU) and the 8 B/pixel maps all-gathered before
the fusion.  Every step is bit-identical to the host restatement below, which in turn is
pinned by the reference's golden vectors (g11, g12, g16).
"""
import threading
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .camera import Camera, CameraPose
from .imageprep import prepare_views
from .. import engine as _engine
from .. import parallel as _parallel


def _torch_cuda():
    """torch with a usable HIP device, or None (then maps travel through host arrays)."""
    try:
        import torch
    except Exception:  # noqa: BLE001
        return None
    return torch if torch.cuda.is_available() else None


_DRAW_BUF = threading.local()     # .pair = (arange, scratch) of DenseStereoReconstructor._draw_without_replacement


class DenseStereoReconstructor:
    NUM_NEIGHBORS = 6        # reference :109

    def __init__(self, camera: Camera, scale: float = 0.25, num_depths: int = 64,
                 patch_size: int = 5, min_views: int = 3, consistency_thresh: float = 0.8, *,
                 device: Optional[int] = None, device_filter: bool = True, mode: str = "exact",
                 process_group=None, device_prep: Optional[bool] = None):
        self.camera = camera
        self.scale = scale
        self.num_depths = num_depths
        self.patch_size = patch_size
        self.min_views = min_views
        self.consistency_thresh = consistency_thresh
        self.device_id = _parallel.local_device() if device is None else int(device)
        self.device_filter = device_filter       # outlier filter's neighbour search on the GPU
        self.process_group = process_group       # torch.distributed group the reference views are sharded over
        # resize / gray conversion on the GPU (amvs_set_view_bgr8): the default only where cv2 is not
        # importable (see PatchMatchMVS.__init__)
        from . import imageprep as _ip
        self.device_prep = (_ip._cv is None) if device_prep is None else bool(device_prep)
        if mode not in ("exact", "fast"):
            raise ValueError("mode must be 'exact' or 'fast'")
        self.mode = mode                         # arithmetic of the sweep (include/amvs.h AMVS_MODE_*)
        print(f"Dense stereo using GPU: HIP device {self.device_id} (gfx950 kernels)")
        # fx, fy, cx, cy scaled (reference :55-59)
        self.K_scaled = camera.K.copy()
        for r, c in ((0, 0), (1, 1), (0, 2), (1, 2)):
            self.K_scaled[r, c] *= scale
        self._engine = None
        self._engine_key = None
        self._engine_images = None
        self._resident_colors = False    # the engine holds the prepared colour images (device image prep)
        self._slot = {}

    def reconstruct(self, images: List[dict], poses: Dict[int, CameraPose],
                    max_pairs: int = 30) -> Tuple[np.ndarray, np.ndarray]:
        print("\n" + "=" * 60)
        print("GPU DENSE STEREO")
        print(f"  Scale: {self.scale}x, Depths: {self.num_depths}, Min views: {self.min_views}")
        print("=" * 60)
        t0 = time.time()
        camera_indices = sorted(poses.keys())
        n_cameras = len(camera_indices)
        if n_cameras < 3:                                       # reference :77-79
            print("Need at least 3 cameras for multi-view stereo")
            return np.array([]), np.array([])

        print("\nPreparing images...")
        if self.device_prep:
            processed = self._prepare_images_device(images, camera_indices, poses)
        else:
            processed = self._prepare_images(images, camera_indices)

        # depth range from the spread of the camera centres (reference :86-91)
        centers = np.array([poses[idx].center for idx in camera_indices])
        radius = np.percentile(np.linalg.norm(centers - np.median(centers, axis=0), axis=1), 90)
        depth_min = max(0.1, radius * 0.1)
        depth_max = radius * 5.0
        print(f"  Depth range: {depth_min:.2f} - {depth_max:.2f}")

        ref_indices = camera_indices[::max(1, n_cameras // max_pairs)]      # reference :100-101
        print(f"\nProcessing {len(ref_indices)} reference views...")
        t1 = time.time()
        jobs = []
        for ref_idx in ref_indices:
            neighbors = self._find_neighbors(ref_idx, camera_indices, poses, k=self.NUM_NEIGHBORS)
            if len(neighbors) >= 2:                            # reference :111-112
                jobs.append((ref_idx, neighbors))
        if not jobs:
            print("No points reconstructed!")
            return np.array([]), np.array([])

        H, W = processed[camera_indices[0]]["shape"]
        depths = 1.0 / np.linspace(1 / depth_max, 1 / depth_min, self.num_depths)      # reference :204-205
        eng = self._ensure_engine(processed, poses)
        counts, total, resident = self._sweep_and_backproject(eng, jobs, processed, poses, depths, H, W)
        per_view = (time.time() - t1) / len(jobs)
        for i, ((ref_idx, _), cnt) in enumerate(zip(jobs, counts)):
            print(f"  [{i+1}/{len(jobs)}] Cam {ref_idx}: {cnt:,} pts ({per_view:.1f}s)")
        if total == 0:
            print("No points reconstructed!")
            return np.array([]), np.array([])
        print("\nMerging point clouds...")
        print(f"  Raw points: {total:,}")
        points, colors = self._filter_and_downsample_device(eng, total, voxel_size=0.02)
        print(f"\nDense stereo completed in {time.time() - t0:.1f}s")
        return points, colors

    def _sweep_and_backproject(self, eng, jobs, processed, poses, depths, H, W):
        """Batched plane sweep of this rank's reference views (maps stay on the GPU), all-gather of
        the maps when ranks share the work, back-projection on the device.  Returns the per-view
        point counts, their sum, and whether the cloud is resident (always, here)."""
        rank, world = _parallel.rank_world(self.process_group)
        mine = _parallel.shard(len(jobs), rank, world)
        groups = {}
        for j in mine:                                     # a batch has one neighbour count
            groups.setdefault(len(jobs[j][1]), []).append(j)
        K_inv = np.linalg.inv(self.K_scaled)
        min_conf = self.min_views - 0.5                    # reference :121
        single = world == 1 and len(groups) == 1
        if single:
            js = next(iter(groups.values()))
            eng.plane_sweep_batch([self._slot[jobs[j][0]] for j in js],
                                  [[self._slot[i] for i in jobs[j][1]] for j in js],
                                  depths, self.patch_size, self.consistency_thresh)
            view_poses = [(poses[jobs[j][0]].R, poses[jobs[j][0]].t) for j in js]
            if self._resident_colors and processed is self._engine_images:
                counts, total = eng.stereo_backproject_views([self._slot[jobs[j][0]] for j in js], K_inv, view_poses, min_conf)
            else:
                cols = np.stack([processed[jobs[j][0]]["color"] for j in js])
                counts, total = eng.stereo_backproject(cols, K_inv, view_poses, min_conf)
            return counts, total, True
        # several batches and / or several ranks: the maps of every view are collected first -- in
        # device tensors when torch-ROCm is there (job j in row j; the sweeps write their rows, RCCL
        # gathers the rank blocks in place, the back-projection reads them: nothing crosses PCIe)
        torch = _torch_cuda()
        runs, cur = [], []
        for j in mine:                                     # launches = runs of consecutive jobs, one neighbour count
            if cur and (len(jobs[j][1]) != len(jobs[cur[-1]][1]) or j != cur[-1] + 1):
                runs.append(cur)
                cur = []
            cur.append(j)
        if cur:
            runs.append(cur)
        order = list(range(len(jobs))) if world > 1 else mine
        view_poses = [(poses[jobs[j][0]].R, poses[jobs[j][0]].t) for j in order]
        cols = np.stack([processed[jobs[j][0]]["color"] for j in order])
        if torch is not None:
            dev = torch.device("cuda", self.device_id)
            hw = H * W
            per = (len(jobs) + world - 1) // world
            base = rank * per
            dmaps = torch.zeros((world * per, hw), dtype=torch.float32, device=dev)
            cmaps = torch.zeros((world * per, hw), dtype=torch.float32, device=dev)
            torch.cuda.synchronize(dev)
            for js in runs:
                eng.plane_sweep_device([self._slot[jobs[j][0]] for j in js],
                                       [[self._slot[i] for i in jobs[j][1]] for j in js],
                                       depths, self.patch_size, self.consistency_thresh,
                                       dmaps[js[0]].data_ptr(), cmaps[js[0]].data_ptr())
            eng.sync()
            if world > 1:
                direct = torch.distributed.get_backend(self.process_group) == "nccl"
                for t in (dmaps, cmaps):
                    block = t[base: base + per]
                    if direct:
                        torch.distributed.all_gather_into_tensor(t, block.clone(), group=self.process_group)
                    else:                                  # gloo (tests): staged through the host
                        full = torch.empty((world * per, hw), dtype=torch.float32)
                        torch.distributed.all_gather_into_tensor(full, block.cpu(), group=self.process_group)
                        t.copy_(full)
                torch.cuda.synchronize(dev)
            first = 0 if world > 1 else mine[0]
            counts, total = eng.stereo_backproject(cols, K_inv, view_poses, min_conf,
                                                   device_ptrs=(dmaps[first].data_ptr(), cmaps[first].data_ptr()))
            return counts, total, True
        dmaps = np.zeros((len(mine), H, W), np.float32)
        cmaps = np.zeros((len(mine), H, W), np.float32)
        row = {j: n for n, j in enumerate(mine)}
        for js in runs:
            eng.plane_sweep_batch([self._slot[jobs[j][0]] for j in js],
                                  [[self._slot[i] for i in jobs[j][1]] for j in js],
                                  depths, self.patch_size, self.consistency_thresh)
            d, c = eng.fetch_sweep_maps(0, len(js))
            for n, j in enumerate(js):
                dmaps[row[j]], cmaps[row[j]] = d[n], c[n]
        counts, total = eng.stereo_backproject(cols, K_inv, view_poses, min_conf, depth=dmaps, conf=cmaps)
        return counts, total, True

    def _filter_and_downsample_device(self, eng, total, voxel_size, k=20, std_ratio=2.0):
        """_filter_outliers (:439-473) + _voxel_down_sample (:475-492) on the resident cloud.  The
        neighbour statistic comes from the GPU; mean + std_ratio * std and the comparison stay in numpy
        (as in the reference), the selection and the voxel grid run on the GPU again.  Clouds the
        reference sub-samples at random (> 500k points, unseeded np.random.choice) are sub-sampled with the
        same draw on the device (amvs_cloud_take); neighbour counts the device search is not compiled for take
        the host path."""
        host_path = not self.device_filter or not eng.knn_supported(k) or k >= min(total, 500000) // 2
        if total < k + 1:
            keep = None
        elif total > 500000 and not host_path and not getattr(self, "_subsample_on_host", False):
            # the reference sub-samples clouds above 500 000 points at random (unseeded np.random.choice, :449-451)
            # and filters the sample: the same draw, the sample taken on the device (amvs_cloud_take), so that the
            # cloud never travels to the host -- the same points as the host path returns for the same draw
            chosen = self._draw_without_replacement(total, 500000)
            total = eng.cloud_take(chosen)
            mean_d = eng.cloud_knn_mean_distance(total, k)
            keep = mean_d < np.mean(mean_d) + std_ratio * np.std(mean_d)
        elif total > 500000 or host_path:
            points, colors = eng.fetch_cloud(total)
            points, colors = self._filter_outliers(points, colors, k, std_ratio)
            print(f"  After outlier removal: {len(points):,}")
            points, colors = self._voxel_down_sample(points, colors, voxel_size)
            print(f"  After voxel downsample: {len(points):,}")
            return points, colors
        else:
            mean_d = eng.cloud_knn_mean_distance(total, k)
            keep = mean_d < np.mean(mean_d) + std_ratio * np.std(mean_d)
        print(f"  After outlier removal: {int(total if keep is None else keep.sum()):,}")
        m = eng.cloud_voxel_downsample(voxel_size, keep)
        print(f"  After voxel downsample: {m:,}")
        return eng.fetch_cloud(m)

    def _draw_without_replacement(self, total: int, size: int) -> np.ndarray:
        """`np.random.choice(total, size, replace=False)` -- the reference's unseeded draw (dense_stereo.py:449-451) --
        made in buffers the module keeps (one pair per thread).  NumPy's legacy generator defines that call as `permutation(total)[:size]`
        and `permutation(n)` as `arange(n)` shuffled in place, so shuffling a kept copy of `arange(total)` consumes the
        global generator identically and returns the same indices (checked against np.random.choice in
        tests/test_host_logic.py).  Why: the 4.8 MB array a fresh `choice` allocates is handed to the driver's
        host-to-device copy straight after it is written, and on this platform such a copy from a just-mapped host
        range takes 13-24 ms instead of 0.1 ms in every second call or so (measured); a buffer
        that is reused does not show it."""
        buf = getattr(_DRAW_BUF, "pair", None)        # one pair of buffers per thread
        if buf is None or buf[0].size < total:
            cap = max(int(total), 1 << 20)
            buf = _DRAW_BUF.pair = (np.arange(cap, dtype=np.int64), np.empty(cap, np.int64))
        idx = buf[1][:total]
        np.copyto(idx, buf[0][:total])
        np.random.shuffle(idx)
        return idx[:size]

    # ------------------------------------------------------------------ host ------
    def _prepare_images(self, images: List[dict], indices: List[int]) -> Dict:
        return dict(zip(indices, prepare_views([images[idx]["image"] for idx in indices], self.scale)))

    def _prepare_images_device(self, images: List[dict], indices: List[int], poses: Dict[int, CameraPose]) -> Dict:
        """_prepare_images on the GPU (amvs_set_view_bgr8): upload the 8-bit BGR images, resize and
        convert there; the engine is cached for the returned dict ('gray' is None)."""
        h, w = images[indices[0]]["image"].shape[:2]
        H, W = int(h * self.scale), int(w * self.scale)
        eng = self._engine
        if eng is None or not eng.reusable_for(H, W, len(indices), self.K_scaled, self.device_id, self.mode):
            if eng is not None:
                eng.close()
                self._engine = None
            eng = _engine.Engine(H, W, len(indices), self.K_scaled.astype(np.float32), device=self.device_id, mode=self.mode)
        self._slot = {idx: s for s, idx in enumerate(indices)}
        prepared = {}
        for idx in indices:
            img = images[idx]["image"]
            same = (H, W) == tuple(img.shape[:2])
            color = eng.set_view_bgr8(self._slot[idx], img, poses[idx].R, poses[idx].t, want_color=not same)
            prepared[idx] = {"color": img if same else color, "gray": None, "shape": (H, W)}
        self._engine, self._engine_images, self._resident_colors = eng, prepared, True
        self._engine_key = self._make_engine_key(prepared, poses)
        return prepared

    def _make_engine_key(self, processed: Dict, poses: Dict[int, CameraPose]):
        indices = sorted(processed.keys())
        H, W = processed[indices[0]]["shape"]
        pose_print = b"".join(np.asarray(poses[i].R, np.float64).tobytes() + np.asarray(poses[i].t, np.float64).tobytes()
                              for i in indices)
        return (tuple(indices), (int(H), int(W)), pose_print, self.K_scaled.tobytes(), self.device_id)

    def _find_neighbors(self, ref_idx: int, all_indices: List[int],
                        poses: Dict[int, CameraPose], k: int = 6) -> List[int]:
        """k nearest camera centres, stable order (reference :178-191)."""
        c_ref = poses[ref_idx].center
        ranked = sorted(((idx, np.linalg.norm(poses[idx].center - c_ref))
                         for idx in all_indices if idx != ref_idx), key=lambda item: item[1])
        return [idx for idx, _ in ranked[:k]]

    def _ensure_engine(self, processed: Dict, poses: Dict[int, CameraPose]):
        indices = sorted(processed.keys())
        H, W = processed[indices[0]]["shape"]
        # strong reference to the dict + pose fingerprint (see PatchMatchMVS._ensure_engine)
        key = self._make_engine_key(processed, poses)
        if self._engine is not None and self._engine_images is processed and self._engine_key == key:
            return self._engine
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        eng = _engine.Engine(H, W, len(indices), self.K_scaled.astype(np.float32), device=self.device_id,
                             mode=self.mode)
        self._slot = {idx: s for s, idx in enumerate(indices)}
        for idx in indices:
            eng.set_view(self._slot[idx], processed[idx]["gray"], poses[idx].R, poses[idx].t)
        self._engine, self._engine_key, self._engine_images = eng, key, processed
        self._resident_colors = False
        return eng

    def _compute_depth_map_gpu(self, ref_idx: int, neighbor_indices: List[int], processed: Dict,
                               poses: Dict[int, CameraPose], depth_min: float, depth_max: float):
        """Inverse-depth plane list far -> near (reference :204-205), then the sweep."""
        H, W = processed[ref_idx]["shape"]
        depths = 1.0 / np.linspace(1 / depth_max, 1 / depth_min, self.num_depths)
        return self._plane_sweep_torch(processed[ref_idx]["gray"], processed[ref_idx]["color"],
                                       poses[ref_idx], neighbor_indices, processed, poses, depths, H, W,
                                       ref_idx=ref_idx)

    def _plane_sweep_torch(self, ref_gray, ref_color, ref_pose, neighbor_indices, processed, poses,
                           depths, H, W, ref_idx=None):
        """Reference :222-316 on the device.  Same positional signature; the reference view is
        identified by `ref_idx` (or by matching `ref_gray` against `processed`)."""
        eng = self._ensure_engine(processed, poses)
        if ref_idx is None:
            ref_idx = next(i for i in processed if processed[i]["gray"] is ref_gray)
        depth_map, confidence = eng.plane_sweep(self._slot[ref_idx],
                                                [self._slot[i] for i in neighbor_indices],
                                                depths, self.patch_size, self.consistency_thresh)
        return depth_map, confidence, ref_color

    def _backproject(self, depth_map: np.ndarray, confidence: np.ndarray, color_map: np.ndarray,
                     pose: CameraPose, min_confidence: float):
        """Pixels with enough votes and positive depth -> world points (reference :407-437)."""
        keep = (confidence >= min_confidence) & (depth_map > 0)
        if not np.any(keep):
            return np.array([]).reshape(0, 3), np.array([]).reshape(0, 3)
        ys, xs = np.where(keep)
        K_inv = np.linalg.inv(self.K_scaled)
        pix = np.stack([xs, ys, np.ones_like(xs)], axis=-1).astype(np.float32)
        cam_pts = (pix @ K_inv.T) * depth_map[keep][:, np.newaxis]
        world = (cam_pts - pose.t) @ pose.R
        return world, color_map[ys, xs][:, ::-1]

    def _filter_outliers(self, points: np.ndarray, colors: np.ndarray, k: int = 20, std_ratio: float = 2.0):
        """Mean distance to the k nearest neighbours must stay below mean + std_ratio*std
        (reference :439-473; unseeded random 500k subsample above that size, as there).  The
        neighbour search -- scikit-learn on the host in the reference, 93 % of the stereo path's
        wall time -- runs on the GPU when this object has an engine (amvs_knn_mean_distance:
        the same mean distances bit for bit); threshold and selection stay in numpy."""
        if len(points) < k + 1:
            return points, colors
        if len(points) > 500000:
            chosen = np.random.choice(len(points), 500000, replace=False)
        else:
            chosen = np.arange(len(points))
        sample = points[chosen]
        # (scikit-learn answers k >= n // 2 with its brute-force kernel, whose rounding differs from
        # the KD-tree expression the device reproduces: such tiny clouds stay on the host)
        if self._engine is not None and self.device_filter and k < len(sample) // 2 and self._engine.knn_supported(k):
            mean_d = self._engine.knn_mean_distance(sample, k)
        else:
            try:
                from sklearn.neighbors import NearestNeighbors
            except ImportError:
                dist = np.linalg.norm(points - np.median(points, axis=0), axis=1)
                keep = dist < np.percentile(dist, 95)
                return points[keep], colors[keep]
            dists, _ = NearestNeighbors(n_neighbors=k).fit(sample).kneighbors(sample)
            mean_d = np.mean(dists[:, 1:], axis=1)
        inlier = mean_d < np.mean(mean_d) + std_ratio * np.std(mean_d)
        return points[chosen[inlier]], colors[chosen[inlier]]

    def _voxel_down_sample(self, points: np.ndarray, colors: np.ndarray, voxel_size: float):
        """First point of every voxel in key order (reference :475-492)."""
        if len(points) == 0:
            return points, colors
        cell = np.floor(points / voxel_size).astype(np.int32).astype(np.int64)
        keys = cell[:, 0] * 1000000000 + cell[:, 1] * 1000000 + cell[:, 2]
        _, first = np.unique(keys, return_index=True)
        return points[first], colors[first]


def create_combined_dense_cloud(camera: Camera, images: List[dict], poses: Dict[int, CameraPose],
                                use_stereo: bool = True):
    """Reference :495-505."""
    if use_stereo:
        return DenseStereoReconstructor(camera).reconstruct(images, poses)
    return np.array([]), np.array([])
