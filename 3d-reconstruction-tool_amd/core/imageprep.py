"""Image preparation in front of the sweep (reference mvs_patchmatch.py:167-191,
dense_stereo.py:156-176): resize by `scale`, BGR -> gray, /255.

The reference does this with OpenCV (cv.resize INTER_LINEAR, cv.cvtColor BGR2GRAY; requirements.txt
pins opencv-python>=4.5.0, which is NOT vendored and absent from the build container).  When cv2 is
importable it is used, so results are the reference's by construction; otherwise the NumPy
restatement below runs -- the published algorithm of OpenCV 4.x for 8-bit images:

  resize (imgproc/resize.cpp, resizeGeneric_ with HResizeLinear / VResizeLinear<uchar>):
    per destination column dx: fx = (float)((dx + 0.5) * (src_w / dst_w) - 0.5); sx = floor(fx);
    fx -= sx; taps clamped into the image with fx = 0; weights cvRound((1 - fx) * 2048) and
    cvRound(fx * 2048), each rounded on its own (float32 arithmetic, round half to even); rows the
    same, except that row indices are clamped and the weights kept;
    horizontal pass D = S[sx] * a0 + S[sx+1] * a1 (int32);
    vertical pass   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
    (an exact 2x reduction is routed to INTER_AREA by cv.resize, which gives the same bytes);
  gray (imgproc/color_rgb.simd.hpp, RGB2Gray<uchar>, the 15-bit coefficients of OpenCV >= 4.x):
    (B * 3735 + G * 19235 + R * 9798 + 16384) >> 15.

PARITY WITH cv2 IS UNPINNED (no OpenCV here to compare with): the stated expectation is bit
equality for every scale; a build of OpenCV that dispatches 8-bit resize to IPP / another HAL, or
the older 14-bit gray coefficients (4899, 9617, 1868, >> 14; OpenCV 3.x), can differ by one gray
code on isolated pixels.  The device path (amvs_set_view_bgr8, csrc/amvs_prep.hip) is bit-identical to
this file.  At scale 1.0 the resize is a copy in all of them.
"""
import numpy as np

try:  # pragma: no cover - cv2 is not available in the build container
    import cv2 as _cv
except Exception:  # noqa: BLE001
    _cv = None


def _axis_tables(n_dst, n_src, clamp_taps):
    """(offsets, w0, w1) of OpenCV's linear resize for one axis (float32 arithmetic as there)."""
    scale = 1.0 / (n_dst / n_src)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if clamp_taps:
        lo, hi = s < 0, s >= n_src - 1
        f[lo | hi] = np.float32(0.0)
        s[lo] = 0
        s[hi] = n_src - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int32)
    w1 = np.rint(f * np.float32(2048.0)).astype(np.int32)
    return s, w0, w1


def _resize_linear_u8(img, new_w, new_h):
    """cv.resize(img, (new_w, new_h)) for 8-bit images, INTER_LINEAR (see the module docstring)."""
    h, w = img.shape[:2]
    if (new_w, new_h) == (w, h):
        return img.copy()
    x0, a0, a1 = _axis_tables(new_w, w, True)
    y0, b0, b1 = _axis_tables(new_h, h, False)
    x1 = np.minimum(x0 + 1, w - 1)
    src = img.astype(np.int32)
    if src.ndim == 2:
        src = src[:, :, None]
    rows = src[:, x0, :] * a0[None, :, None] + src[:, x1, :] * a1[None, :, None]
    r0 = rows[np.clip(y0, 0, h - 1)] >> 4
    r1 = rows[np.clip(y0 + 1, 0, h - 1)] >> 4
    out = (((b0[:, None, None] * r0) >> 16) + ((b1[:, None, None] * r1) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def _bgr_to_gray_u8(img):
    """cv.cvtColor(BGR2GRAY) for 8-bit: (B*3735 + G*19235 + R*9798 + 16384) >> 15."""
    acc = img[:, :, 0].astype(np.uint32) * np.uint32(3735)
    acc += img[:, :, 1].astype(np.uint32) * np.uint32(19235)
    acc += img[:, :, 2].astype(np.uint32) * np.uint32(9798)
    acc += np.uint32(16384)
    acc >>= np.uint32(15)
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
