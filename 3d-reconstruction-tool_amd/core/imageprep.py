"""Image preparation in front of the sweep (reference mvs_patchmatch.py:167-191,
dense_stereo.py:156-176): resize by `scale`, BGR -> gray, /255.

The reference does this with OpenCV (cv.resize INTER_LINEAR, cv.cvtColor BGR2GRAY).
When cv2 is importable it is used, so results are identical to the reference's;
otherwise a NumPy restatement of OpenCV's documented formulas runs (parity with cv2
unpinned: OpenCV is absent from the build container).  At scale 1.0 the resize is the
identity in both.
"""
import numpy as np

try:  # pragma: no cover - cv2 is not available in the build container
    import cv2 as _cv
except Exception:  # noqa: BLE001
    _cv = None


def _resize_linear_u8(img, new_w, new_h):
    """cv.resize(..., INTER_LINEAR) restated: half-pixel centres, clamped taps, 11-bit
    fixed-point weights, rounding as OpenCV's 8-bit path."""
    h, w = img.shape[:2]
    if (new_w, new_h) == (w, h):
        return img.copy()

    def axis(n_dst, n_src):
        pos = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
        i0 = np.floor(pos).astype(np.int64)
        frac = pos - i0
        frac[i0 < 0] = 0.0
        i0 = np.clip(i0, 0, n_src - 1)
        i1 = np.clip(i0 + 1, 0, n_src - 1)
        w1 = np.rint(frac * 2048).astype(np.int64)
        return i0, i1, 2048 - w1, w1

    x0, x1, wx0, wx1 = axis(new_w, w)
    y0, y1, wy0, wy1 = axis(new_h, h)
    src = img.astype(np.int64)
    if src.ndim == 2:
        src = src[:, :, None]
    rows = src[:, x0, :] * wx0[None, :, None] + src[:, x1, :] * wx1[None, :, None]
    out = rows[y0] * wy0[:, None, None] + rows[y1] * wy1[:, None, None]
    out = (out + (1 << 21)) >> 22
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def _bgr_to_gray_u8(img):
    """cv.cvtColor(BGR2GRAY) for 8-bit: (B*1868 + G*9617 + R*4899 + 8192) >> 14."""
    # (32-bit arithmetic: the sum is below 2^22)
    acc = img[:, :, 0].astype(np.uint32) * np.uint32(1868)
    acc += img[:, :, 1].astype(np.uint32) * np.uint32(9617)
    acc += img[:, :, 2].astype(np.uint32) * np.uint32(4899)
    acc += np.uint32(8192)
    acc >>= np.uint32(14)
    return acc.astype(np.uint8)


def prepare_view(image_bgr_u8: np.ndarray, scale: float) -> dict:
    """-> {'color': u8 (H,W,3) BGR, 'gray': f32 (H,W) in [0,1], 'shape': (H,W)}"""
    h, w = image_bgr_u8.shape[:2]
    new_h, new_w = int(h * scale), int(w * scale)
    if _cv is not None:
        scaled = _cv.resize(image_bgr_u8, (new_w, new_h))
        gray8 = _cv.cvtColor(scaled, _cv.COLOR_BGR2GRAY)
    else:
        scaled = _resize_linear_u8(np.ascontiguousarray(image_bgr_u8), new_w, new_h)
        gray8 = _bgr_to_gray_u8(scaled)
    gray = gray8.astype(np.float32)
    gray /= np.float32(255.0)
    return {"color": scaled, "gray": gray, "shape": (new_h, new_w)}


def prepare_views(images_bgr_u8, scale: float, workers: int = 8) -> list:
    """prepare_view for a list of images on a small thread pool (the NumPy / OpenCV kernels
    release the GIL); results in input order."""
    if len(images_bgr_u8) <= 1 or workers <= 1:
        return [prepare_view(im, scale) for im in images_bgr_u8]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(workers, len(images_bgr_u8))) as pool:
        return list(pool.map(lambda im: prepare_view(im, scale), images_bgr_u8))
