import sys, time
sys.path.insert(0, ".")
import numpy as np
import amvs
rng = np.random.default_rng(1)
xy = rng.uniform(-2, 2, (61000, 2))
pts = np.column_stack([xy, 0.15 * np.sin(1.3 * xy[:, 0]) * np.cos(1.7 * xy[:, 1]) + rng.normal(0, 0.004, 61000)])
K = np.eye(3, dtype=np.float32)
with amvs.Engine(8, 8, 1, K) as eng:
    for n in (61000, 61000, 20000, 61000):
        t = time.time(); m = eng.knn_mean_distance(pts[:n], 20); print(n, "knn_mean_distance", round((time.time() - t) * 1e3, 2), "ms")
big = np.vstack([pts + [4.0 * i, 0, 0] for i in range(8)])
with amvs.Engine(8, 8, 1, K) as eng:
    for _ in range(2):
        t = time.time(); m = eng.knn_mean_distance(big, 20); print(len(big), "knn_mean_distance", round((time.time() - t) * 1e3, 2), "ms")
out = np.vstack([pts, rng.uniform(-30, 30, (300, 3))])
with amvs.Engine(8, 8, 1, K) as eng:
    for _ in range(2):
        t = time.time(); m = eng.knn_mean_distance(out, 20); print(len(out), "with 300 far outliers", round((time.time() - t) * 1e3, 2), "ms")
    from sklearn.neighbors import NearestNeighbors
    d, _ = NearestNeighbors(n_neighbors=20).fit(out).kneighbors(out)
    print("bit-exact with outliers:", np.array_equal(m, np.mean(d[:, 1:], axis=1)))
