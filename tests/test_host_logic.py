"""Host-side mirror of the reference's class surface (no GPU needed): source selection, depth
range, fusion, filters, contracts -- against golden vectors captured from the reference -- and
the C ABI's export table."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, GoldenScene, load_golden

import amvs
from amvs.core.dense_stereo import DenseStereoReconstructor
from amvs.core.imageprep import prepare_view
from amvs.core.mvs_patchmatch import DepthNormalMap, PatchMatchMVS


def _pm(scene, **kw):
    pm = PatchMatchMVS(amvs.Camera(K=scene.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7, **kw)
    pm.depth_min, pm.depth_max = scene.depth_min, scene.depth_max
    return pm


def test_abi_exports_every_declared_symbol():
    """libamvs.so loads without a GPU and exports exactly what include/amvs.h declares."""
    hdr = open(os.path.join(ROOT, "include", "amvs.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(amvs_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    from amvs import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert b"amvs" in _lib.load().amvs_version()


def test_no_gpu_fails_loudly():
    """Without a HIP device the product path raises; it never falls back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(amvs._lib.AmvsError, match="no HIP device"):
        amvs.Engine(32, 32, 3, np.eye(3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "3d-reconstruction-tool_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_g08_select_source_views():
    g = load_golden("g08_select_sources")
    pm = PatchMatchMVS.__new__(PatchMatchMVS)
    for Rk, tk, sk in (("R", "t", "selected"), ("Rb", "tb", "selected_b")):
        poses = {i: amvs.CameraPose(R=g[Rk][i], t=g[tk][i]) for i in range(8)}
        got = np.array([pm._select_source_views(r, sorted(poses), poses, k=4) for r in sorted(poses)])
        assert np.array_equal(got, g[sk])


def test_g09_depth_range(scene_b):
    g = load_golden("g09_depth_range")
    pm = _pm(scene_b)
    pm._estimate_depth_range(scene_b.poses(), g["sparse"])
    assert (pm.depth_min, pm.depth_max) == tuple(g["with_sparse"])
    pm._estimate_depth_range(scene_b.poses(), None)
    assert (pm.depth_min, pm.depth_max) == tuple(g["fallback"])
    pm._estimate_depth_range(scene_b.poses(), np.zeros((0, 3)))
    assert (pm.depth_min, pm.depth_max) == tuple(g["fallback"])


def test_g10_fuse_and_filter(scene_b):
    g = load_golden("g10_fuse_filter")
    pm = _pm(scene_b)
    poses = scene_b.poses()
    proc = {i: {"gray": scene_b.grays[i], "color": scene_b.colors[i], "shape": (scene_b.H, scene_b.W)}
            for i in range(scene_b.n)}
    maps = {int(r): DepthNormalMap(depth=scene_b.gt_depth[int(r)],
                                   normal=np.zeros((scene_b.H, scene_b.W, 3), np.float32),
                                   confidence=g["confidence"][n]) for n, r in enumerate(g["refs"])}
    pts, cols = pm._fuse_depth_maps(maps, proc, poses)
    assert pts.dtype == np.float64 and cols.dtype == np.uint8
    assert np.array_equal(pts, g["points"]) and np.array_equal(cols, g["colors"])
    fp, fc = pm._filter_points(pts, cols)
    assert np.array_equal(fp, g["f_points"]) and np.array_equal(fc, g["f_colors"])
    # nothing confident -> (0,3) arrays (reference :567-568)
    empty = {0: DepthNormalMap(depth=scene_b.gt_depth[0], normal=maps[0].normal,
                               confidence=np.zeros((scene_b.H, scene_b.W), np.float32))}
    p0, c0 = pm._fuse_depth_maps(empty, proc, poses)
    assert p0.shape == (0, 3) and c0.shape == (0, 3)


def test_g12_stereo_post(scene_c):
    g = load_golden("g12_stereo_post")
    s11 = load_golden("g11_plane_sweep")
    ds = DenseStereoReconstructor(amvs.Camera(K=scene_c.K.copy(), dist=np.zeros(5)), scale=1.0,
                                  num_depths=16, patch_size=5)
    poses = scene_c.poses()
    assert ds._find_neighbors(2, sorted(poses), poses, k=6) == list(s11["nbrs"])
    bp, bc = ds._backproject(s11["depth_map"], s11["confidence"], scene_c.colors[2], poses[2],
                             min_confidence=ds.min_views - 0.5)
    assert np.array_equal(bp, g["bp_points"]) and np.array_equal(bc, g["bp_colors"])
    gp, gc = ds._backproject(scene_c.gt_depth[2], np.full((scene_c.H, scene_c.W), 4.0, np.float32),
                             scene_c.colors[2], poses[2], min_confidence=2.5)
    assert np.array_equal(gp, g["gt_points"])
    vp, vc = ds._voxel_down_sample(gp, gc, voxel_size=0.02)
    assert np.array_equal(vp, g["vox_points"]) and np.array_equal(vc, g["vox_colors"])
    op, oc = ds._filter_outliers(gp, gc)
    assert np.array_equal(op, g["out_points"]) and np.array_equal(oc, g["out_colors"])
    few = ds._filter_outliers(gp[:10], gc[:10])
    assert len(few[0]) == 10


def test_g13_contracts_fewer_than_three_cameras(scene_b, scene_c, capsys):
    g = load_golden("g13_contracts")
    poses = scene_b.poses()
    pm = _pm(scene_b)
    p, c = pm.reconstruct([{"image": scene_b.colors[0]}, {"image": scene_b.colors[1]}], {0: poses[0], 1: poses[1]})
    assert tuple(p.shape) == tuple(g["pm_points_shape"]) and tuple(c.shape) == tuple(g["pm_colors_shape"])
    out = capsys.readouterr().out
    assert "PATCHMATCH MULTI-VIEW STEREO" in out and "Need at least 3 cameras" in out
    ds = DenseStereoReconstructor(amvs.Camera(K=scene_c.K.copy(), dist=np.zeros(5)), scale=1.0)
    pc = scene_c.poses()
    p2, c2 = ds.reconstruct([{"image": scene_c.colors[0]}, {"image": scene_c.colors[1]}], {0: pc[0], 1: pc[1]})
    assert tuple(p2.shape) == tuple(g["st_points_shape"]) and tuple(c2.shape) == tuple(g["st_colors_shape"])
    assert "GPU DENSE STEREO" in capsys.readouterr().out


def test_constructor_surface_matches_reference_defaults():
    cam = amvs.Camera(K=np.array([[100.0, 0, 50], [0, 100, 40], [0, 0, 1]]), dist=np.zeros(5))
    pm = PatchMatchMVS(cam)
    assert (pm.scale, pm.patch_size, pm.num_iterations, pm.num_samples, pm.min_views,
            pm.depth_min, pm.depth_max) == (0.25, 11, 3, 8, 3, 0.1, 100.0)
    assert np.allclose(pm.K_scaled, [[25, 0, 12.5], [0, 25, 10], [0, 0, 1]])
    ds = DenseStereoReconstructor(cam)
    assert (ds.scale, ds.num_depths, ds.patch_size, ds.min_views, ds.consistency_thresh) == (0.25, 64, 5, 3, 0.8)
    assert np.allclose(ds.K_scaled, [[25, 0, 12.5], [0, 25, 10], [0, 0, 1]])
    pose = amvs.CameraPose(R=np.eye(3), t=np.array([1.0, 2.0, 3.0]))
    assert np.allclose(pose.center, [-1, -2, -3])
    assert np.allclose(pose.transform_points(np.array([[1.0, 1, 1]])), [[2, 3, 4]])


def test_prepare_view_identity_scale_and_gray_formula():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 16, 3), dtype=np.uint8)
    out = prepare_view(img, 1.0)
    assert out["shape"] == (12, 16) and np.array_equal(out["color"], img)
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    gray8 = (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15       # OpenCV >= 4.x RGB2Gray<uchar>
    assert np.array_equal(out["gray"], gray8.astype(np.float32) / np.float32(255.0))
    half = prepare_view(img, 0.5)
    assert half["shape"] == (6, 8) and half["gray"].dtype == np.float32
    # an exact 2x reduction equals the 2x2 area average cv.resize routes it to
    a = img.astype(np.int64)
    area = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    assert np.array_equal(half["color"], area)
    # cv.resize's fixed-point linear interpolation, one destination pixel by hand (scale 0.3: 12x16 -> 3x4)
    small = prepare_view(img, 0.3)["color"]
    assert small.shape == (3, 4, 3)

    def taps(d, n_dst, n_src):
        f = np.float32((d + 0.5) * (1.0 / (n_dst / n_src)) - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        return s, int(np.rint((np.float32(1) - f) * np.float32(2048))), int(np.rint(f * np.float32(2048)))
    for dy, dx in ((0, 0), (1, 2), (2, 3)):
        sx, a0, a1 = taps(dx, 4, 16)
        sy, b0, b1 = taps(dy, 3, 12)
        for c in range(3):
            d0 = int(img[sy, sx, c]) * a0 + int(img[sy, sx + 1, c]) * a1
            d1 = int(img[sy + 1, sx, c]) * a0 + int(img[sy + 1, sx + 1, c]) * a1
            assert small[dy, dx, c] == ((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2)
    # all 2^24 colours through the 32-bit gray formula, and the thread pool keeps the order
    from amvs.core.imageprep import prepare_views
    cube = np.stack(np.meshgrid(*[np.arange(256, dtype=np.uint8)] * 3, indexing="ij"), axis=-1).reshape(4096, 4096, 3)
    b, g, r = (cube[..., i].astype(np.int64) for i in range(3))
    assert np.array_equal(prepare_view(cube, 1.0)["gray"],
                          (((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.float32) / np.float32(255.0)))
    imgs = [rng.integers(0, 256, (10, 14, 3), dtype=np.uint8) for _ in range(9)]
    pooled, serial = prepare_views(imgs, 0.5), [prepare_view(im, 0.5) for im in imgs]
    assert all(np.array_equal(p["gray"], q["gray"]) and np.array_equal(p["color"], q["color"])
               for p, q in zip(pooled, serial))


def test_save_ply_matches_the_reference_golden(tmp_path):
    """amvs_write_ply against the bytes the reference's own utils.save_ply (utils.py:8-37) wrote for the
    same cloud (g18_ply, captured by tests/golden/make_golden_r3.py): header, `%.6f` rounding incl.
    negative zero, denormals and round-up at the sixth decimal, the empty cloud."""
    from conftest import load_golden
    from amvs.core.utils import save_ply
    g = load_golden("g18_ply")
    out = tmp_path / "sub" / "cloud.ply"
    save_ply(g["points"], g["colors"], str(out))
    assert out.read_bytes() == g["ply_bytes"].tobytes()
    save_ply(np.zeros((0, 3)), np.zeros((0, 3), np.uint8), str(tmp_path / "empty.ply"))
    assert (tmp_path / "empty.ply").read_bytes() == g["ply_bytes_empty"].tobytes()


def test_image_preparation_equals_cv2_when_present():
    """wherever OpenCV is importable, the restated 8-bit resize / BGR2GRAY (core/imageprep.py,
    and with it the bit-identical device path amvs_set_view_bgr8) must equal cv2 itself for several
    scales, incl. the 0.5 / 0.25 reductions the CLI uses.  Skipped in the build container (no cv2: image
    preparation stays parity-UNPINNED there)."""
    cv2 = pytest.importorskip("cv2")
    from amvs.core import imageprep
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (96, 128, 3), dtype=np.uint8)
    for scale in (1.0, 0.75, 0.5, 0.37, 0.25):
        h, w = int(96 * scale), int(128 * scale)
        want_c = cv2.resize(img, (w, h))
        want_g = cv2.cvtColor(want_c, cv2.COLOR_BGR2GRAY)
        got_c = imageprep._resize_linear_u8(img, w, h)
        assert np.array_equal(got_c, want_c), f"resize differs from cv2 at scale {scale}"
        assert np.array_equal(imageprep._bgr_to_gray_u8(got_c), want_g), f"gray differs from cv2 at scale {scale}"


def test_save_ply_writes_the_reference_bytes(tmp_path):
    """The native writer against a straight restatement of utils.save_ply (utils.py:20-35)."""
    from amvs.core.utils import save_ply
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.normal(0, 3, (5000, 3)), rng.normal(0, 1e-7, (50, 3)), rng.normal(0, 1e6, (50, 3)),
                          np.array([[0.0, -0.0, 0.5], [1e-7, -4.9999995e-7, 123456.7890125]])])
    cols = rng.integers(0, 256, (len(pts), 3), dtype=np.uint8)
    want = ["ply", "format ascii 1.0", f"element vertex {len(pts)}", "property float x", "property float y",
            "property float z", "property uchar red", "property uchar green", "property uchar blue", "end_header"]
    for i in range(len(pts)):
        x, y, z = pts[i]
        r, g, b = cols[i].astype(int)
        want.append(f"{x:.6f} {y:.6f} {z:.6f} {r} {g} {b}")
    out = tmp_path / "sub" / "cloud.ply"
    save_ply(pts, cols, str(out))
    assert out.read_text() == "\n".join(want) + "\n"
    save_ply(np.zeros((0, 3)), np.zeros((0, 3)), str(tmp_path / "empty.ply"))
    assert (tmp_path / "empty.ply").read_text().count("\n") == 10


def test_save_ply_fast_formatter_equals_printf_everywhere(tmp_path):
    """Round 4: the writer formats "%.6f" itself (scaled integer + FMA residual; printf only inside 1e-9 of a
    rounding tie and outside |x| < 1e9).  Against Python's own correctly rounded "%.6f" on values built to sit on
    and around every decision of that code: exact ties (k + 1/2 millionths that are binary fractions), their
    neighbours one ulp away, the range limits, powers of ten, denormals, negative zero, non-finite values, huge
    magnitudes -- and 300 000 random doubles over 30 decades."""
    from amvs.core.utils import save_ply
    rng = np.random.default_rng(11)
    # x * 1e6 = n + 1/2 exactly needs x = (2n + 1) / (2^7 5^6) to be a binary fraction: odd multiples of 5^6 / 2^7 millionths
    exact_ties = np.array([(2 * k + 1) * 0.0078125 for k in range(-40, 40)])            # 0.0078125 * 1e6 = 7812.5
    near = np.concatenate([np.nextafter(exact_ties, np.inf), np.nextafter(exact_ties, -np.inf)])
    half_ulps = np.array([0.5e-6, 1.5e-6, 2.5e-6, -0.5e-6, 1234.5678905, 1234.5678915, 0.9999995, 0.9999994999999999,
                          999999999.9999995, 1e9, -1e9, np.nextafter(1e9, 0), 1e15, 1e22, -3.7e300, 1e-320, -1e-320, 5e-324,
                          0.0, -0.0, np.inf, -np.inf, np.nan, 0.1, 0.2, 0.3, 1e-7, 4.9999995e-7, 123456.7890125])
    decades = rng.standard_normal(300_000) * 10.0 ** rng.integers(-12, 18, 300_000)
    millis = rng.integers(-10 ** 12, 10 ** 12, 60_000) / 1e6                              # values ON the 6-decimal grid
    halves = (rng.integers(-10 ** 9, 10 ** 9, 60_000) + 0.5) / 1e6                        # ... and half way between
    vals = np.concatenate([exact_ties, near, half_ulps, decades, millis, halves])
    vals = np.resize(vals, (len(vals) + 2) // 3 * 3).reshape(-1, 3)
    cols = rng.integers(-3, 300, (len(vals), 3))                                           # (the writer takes int64 colours)
    out = tmp_path / "fmt.ply"
    save_ply(vals, cols, str(out))
    got = out.read_text().split("\n")[10:-1]
    assert len(got) == len(vals)
    for i, line in enumerate(got):
        x, y, z = vals[i]
        want = "%.6f %.6f %.6f %d %d %d" % (x, y, z, cols[i][0], cols[i][1], cols[i][2])
        assert line == want, (i, line, want)


def test_bench_labels_and_evidence_helpers():
    """bench.py's host-side helpers: a BASELINE config is named only when the run IS that config; the
    traffic figure is served only for the exact workload key and the kernel sources it was measured on."""
    import argparse
    import bench
    a = argparse.Namespace(patch=7, iters=8, samples=8, fusion=False)
    assert bench.baseline_label(a, 16, 1920, 1080, 1).startswith("BASELINE config 3")
    assert bench.baseline_label(a, 32, 1920, 1080, 4).startswith("BASELINE config 4")
    assert bench.baseline_label(a, 8, 1920, 1080, 2).startswith("custom workload")          # the 8-view rehearsal scene
    assert bench.baseline_label(a, 16, 1280, 720, 1).startswith("custom workload")
    a.fusion = True
    assert bench.baseline_label(a, 64, 3840, 2160, 8).startswith("BASELINE config 5")
    a.fusion, a.patch = False, 11
    assert bench.baseline_label(a, 16, 1920, 1080, 1).startswith("custom workload")
    h = bench.kernel_source_hash("pm_step_fast_kernel<7,4>")
    assert len(h) == 40 and h != bench.kernel_source_hash("pm_step_kernel<7,4>")
    assert bench.profiled_traffic("pm_step_fast_kernel<7,4>", "no such workload") is None
    assert bench.profiled_traffic("no such kernel<1,1>", "4x1920x1080") is None
    t = bench.profiled_traffic("pm_step_fast_kernel<7,4>", "4x1920x1080")
    assert t is None or t > 0                      # None as soon as the kernel sources differ from the profiled ones
    r = bench.valu_issue_roofline("plane_sweep_fast_kernel<5,6>", "8x1280x720", 6.6e8, 6.7)
    assert r is None or (r["bound"] == "valu-issue" and 0.3 < r["frac"] < 1.2)


def test_extended_restatement_is_sane_on_the_cpu():
    """oracle/xpm_oracle.py on its own (no GPU): at the true depth with fronto-parallel planes the window
    cost is low, at a wrong depth high; the view candidates land near the true depth; a half sweep from a
    perturbed map moves its colour towards the truth and leaves the other colour alone."""
    from oracle import oracle, xpm_oracle
    from amvs.synthetic import make_scene
    sc = make_scene(5, 48, 64, seed=17)
    codes = [np.round(g * 255).clip(0, 255).astype(np.uint8) for g in sc.grays]
    K = sc.camera.K.astype(np.float32)
    poses = [(sc.poses[i].R, sc.poses[i].t) for i in range(5)]
    v = xpm_oracle.View(K, np.linalg.inv(K), codes, poses, 2, [1, 3, 0, 4], 7, 2)
    n = np.zeros((48, 64, 3), np.float32)
    n[..., 2] = -1
    gt = sc.depths[2].astype(np.float32)
    c_gt, c_bad = v.cost_map(gt, n), v.cost_map(gt * np.float32(1.2), n)
    fin = np.isfinite(c_gt) & np.isfinite(c_bad)
    assert fin.mean() > 0.5 and np.median(c_gt[fin]) < 0.25 < np.median(c_bad[fin])
    d_all = np.stack([sc.depths[i].astype(np.float32) for i in range(5)])
    cd, cn = v.view_candidates(d_all, np.stack([n] * 5), 0, sc.depth_min, sc.depth_max)
    m = cd > 0
    assert m.mean() > 0.8 and np.median(np.abs(cd[m] - gt[m]) / gt[m]) < 5e-3
    start = gt * np.float32(1.05)
    D, N, C = v.half_sweep(start, n, np.full((48, 64), np.inf, np.float32), cd, cn, 0, 0, 11, oracle.rng_fill,
                           sc.depth_min, sc.depth_max)
    yy, xx = np.mgrid[0:48, 0:64]
    swept = ((xx + yy) & 1) == 0
    assert np.array_equal(D[~swept], start[~swept]) and np.isinf(C[~swept]).all()
    got = swept & np.isfinite(C)
    assert got.mean() > 0.25 and np.median(np.abs(D[got] - gt[got]) / gt[got]) < 0.01


def test_native_exchange_without_rccl_is_unsupported_not_fatal():
    """ADVICE (round 3): when librccl cannot be opened, amvs_comm_unique_id / amvs_comm_init return
    AMVS_EUNSUPPORTED with a message, as include/amvs.h documents (the first version built the message from
    two dlerror() calls, the second of which returns NULL).  AMVS_RCCL_LIB names the library to open; a
    fresh process, because the lookup happens once per process."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); import amvs; from amvs import _lib; L = _lib.load();"
            "buf = (C.c_uint8 * 128)(); rc = L.amvs_comm_unique_id(buf); print(rc); print(L.amvs_last_error(None).decode());"
            "rc2 = L.amvs_comm_unique_id(buf); print(rc2)") % ROOT
    env = dict(os.environ, AMVS_RCCL_LIB="/nonexistent/librccl.so.1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0] == "-3" and lines[-1] == "-3", out.stdout          # AMVS_EUNSUPPORTED, twice
    assert "RCCL not found" in lines[1] and "/nonexistent/librccl.so.1" in lines[1]


def test_stereo_subsample_draw_equals_numpy_choice():
    """DenseStereoReconstructor's draw for clouds above 500 000 points is np.random.choice(total, size, replace=False)
    (dense_stereo.py:449-451) index for index, and leaves NumPy's global generator in the same state -- made in kept
    buffers (core/dense_stereo.py: _draw_without_replacement)."""
    from amvs.core.dense_stereo import DenseStereoReconstructor
    ds = DenseStereoReconstructor.__new__(DenseStereoReconstructor)
    for total, size in ((603638, 500000), (500001, 500000), (1300003, 500000), (700000, 500000), (20, 7)):
        np.random.seed(total % 1000)
        want = np.random.choice(total, size, replace=False)
        after_want = np.random.rand()
        np.random.seed(total % 1000)
        got = ds._draw_without_replacement(total, size)
        after_got = np.random.rand()
        assert got.dtype == want.dtype and np.array_equal(got, want) and after_got == after_want


def test_committed_bench_line_keeps_the_contract():
    """The driver-shaped line of the final build (profiles/r04_logs/bench_default_line.json, written by
    `python bench.py --steps 20 --warmup 5` on an MI355X) carries every field the bench contract names, the figures
    agree with each other, and every replayed traffic figure names its profile."""
    import json
    path = os.path.join(ROOT, "profiles", "r04_logs", "bench_default_line.json")
    lines = [ln for ln in open(path).read().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f32"
    assert "workload" in r["config"] and "model" not in r["config"] and "config 3" in r["config"]["workload"]
    assert r["data"].startswith("synthetic")
    # value = pixel-hypotheses per step / time per step
    assert abs(r["value"] - r["config"]["pixel_hypotheses_per_step"] / r["ms_per_step"] / 1e3) < 0.01 * r["value"]
    records = [("main", r), ("planesweep", r["planesweep"]), ("exact.patchmatch", r["exact"]["patchmatch"]),
               ("exact.planesweep", r["exact"]["planesweep"]), ("cli_defaults", r["cli_defaults"])]
    for name, rec in records:
        roof = rec["roofline"]
        for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert key in roof, (name, key)
        assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s"
        assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
        assert abs(roof["achieved"] - roof["algorithmic_bytes_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e9) < 0.01 * roof["achieved"]
        assert roof["traffic"] is not None and roof["traffic_source"].startswith("profiles/r04_"), name
        assert roof["traffic"] < 3 * roof["algorithmic_bytes_per_launch"]
    cpu = r["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["unit"] == r["unit"] and "sample" in cpu
    cli = r["cli_defaults"]
    assert len(cli["timed_calls_s"]) == 5 and min(cli["timed_calls_s"]) <= cli["end_to_end_s"] <= max(cli["timed_calls_s"])
    assert len(cli["stereo"]["timed_calls_s"]) == 5
