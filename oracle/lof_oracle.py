"""CPU restatement of the Local Outlier Factor as the reference uses it -- TEST INFRASTRUCTURE ONLY
(imported by tests/ alone; the product path is csrc/knn.hip behind knn.local_outlier_factor).

The reference calls scikit-learn (an unpinned dependency of its environment; 1.7.2 in this image):
/root/reference/gs_init_compare/point_cloud_postprocess/postprocess.py:16-22
`LocalOutlierFactor(n_neighbors=config.lof_num_neighbors, n_jobs=-1).fit_predict(pts) == -1`.
Restated from scikit-learn's published algorithm (sklearn/neighbors/_lof.py: `fit`,
`_local_reachability_density`; Breunig et al. 2000) with brute-force neighbours in float64:
  n_neighbors_ = max(1, min(n_neighbors, n_samples - 1))
  dist, ind    = the n_neighbors_ nearest OTHER samples of every sample, ascending
  lrd(i)       = 1 / (mean_j max(dist[i, j], dist[ind[i, j], -1]) + 1e-10)
  nof(i)       = -mean_j (lrd(ind[i, j]) / lrd(i));   outlier <=> nof(i) < -1.5  (contamination="auto")
Pinned by tests/golden/lof_golden.npz, recorded from the reference's function itself
(tests/golden/make_lof_golden.py)."""
import numpy as np


def lof(points: np.ndarray, n_neighbors: int = 40, offset: float = -1.5, chunk: int = 1024):
    x = np.asarray(points, dtype=np.float64)
    n = x.shape[0]
    k = max(1, min(int(n_neighbors), n - 1))
    dist = np.empty((n, k))
    ind = np.empty((n, k), dtype=np.int64)
    for a in range(0, n, chunk):
        d2 = ((x[a:a + chunk, None, :] - x[None, :, :]) ** 2).sum(-1)
        d2[np.arange(d2.shape[0]), np.arange(a, a + d2.shape[0])] = np.inf        # not its own neighbour
        part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        pd = np.take_along_axis(d2, part, 1)
        order = np.argsort(pd, axis=1, kind="stable")
        ind[a:a + chunk] = np.take_along_axis(part, order, 1)
        dist[a:a + chunk] = np.sqrt(np.take_along_axis(pd, order, 1))
    reach = np.maximum(dist, dist[ind, k - 1])
    lrd = 1.0 / (reach.mean(1) + 1e-10)
    nof = -(lrd[ind] / lrd[:, None]).mean(1)
    return nof < offset, nof
