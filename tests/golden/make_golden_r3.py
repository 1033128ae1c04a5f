"""Round-3 golden capture (run ONLY in the build container; /root/reference never travels).

g18_ply: the bytes the reference's own utils.save_ply (utils.py:8-37) writes for a small cloud that
covers the formatting corner cases of `%.6f` (negative zero, values that round up at the sixth
decimal, magnitudes from 1e-7 to 1e6) -- input points / colours and the file's bytes.  utils.py needs
neither cv2 nor torch; it is imported through the same synthetic package as the other captures.
"""
import importlib
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_loader  # noqa: E402


def main():
    ref_loader.load()                      # registers the synthetic `refcore` package
    utils = importlib.import_module("refcore.utils")
    rng = np.random.default_rng(18)
    pts = np.concatenate([rng.normal(0, 3, (400, 3)), rng.normal(0, 1e-7, (20, 3)), rng.normal(0, 1e6, (20, 3)),
                          np.array([[0.0, -0.0, 0.5], [1e-7, -4.9999995e-7, 123456.7890125],
                                    [0.9999995, -0.9999995, 2.5000005], [1e-320, -1e-320, 1.0]])])
    cols = rng.integers(0, 256, (len(pts), 3), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "sub", "cloud.ply")
        utils.save_ply(pts, cols, path)
        data = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
        empty = os.path.join(d, "empty.ply")
        utils.save_ply(np.zeros((0, 3)), np.zeros((0, 3), np.uint8), empty)
        data_empty = np.frombuffer(open(empty, "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "g18_ply.npz"), points=pts, colors=cols, ply_bytes=data,
                        ply_bytes_empty=data_empty)
    print("g18_ply:", len(pts), "points,", data.size, "bytes")


if __name__ == "__main__":
    main()
