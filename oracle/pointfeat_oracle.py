"""oracle/pointfeat_oracle.py -- TEST INFRASTRUCTURE ONLY (not product code).

CPU restatement of the RING++ point-feature front-end:
  * build_neighbors_NN + covariation_eigenvalue: LoopDetection/src/RING_ros/util.py:123-170
    (sklearn kd-tree kNN -> exact kNN via scipy cKDTree here; torch.linalg.eigvalsh on float32)
  * calculate_features: LoopDetection/generate_bev_pointfeat_cython/src/kernel.cu:16-104, whose numpy
    twin is generate_bev_pointfeat_cython/test.py:68-98
The reference pins this only by a printed (not asserted) comparison that needs a GPU + knn_cuda
(test.py:233): parity unpinned; cross-checked against the direct numpy formulas in the tests.
"""
import numpy as np
import torch
from scipy.spatial import cKDTree


def knn_indices(pc, k):
    """util.py:166-167: kneighbors on the fitted set (the point itself is its first neighbour)."""
    pc = np.asarray(pc, dtype=np.float32)
    _, idx = cKDTree(pc).query(pc, k=k)
    return idx.astype(np.int32)


def covariation_eigenvalue(pc, idx):
    """util.py:123-160: covariance P^T P / (k-1) in float32, eigvalsh, descending; 3-D then 2-D."""
    pc = np.asarray(pc, dtype=np.float32)
    nb = pc[idx]
    ex = np.average(nb, axis=1)
    p = nb - ex[:, None, :]
    cov = np.matmul(p.transpose((0, 2, 1)), p) / (nb.shape[1] - 1)
    e3 = torch.linalg.eigvalsh(torch.from_numpy(cov)).numpy()[:, ::-1]
    e2 = torch.linalg.eigvalsh(torch.from_numpy(np.ascontiguousarray(cov[:, :2, :2]))).numpy()[:, ::-1]
    return np.concatenate([e3, e2], axis=1).astype(np.float32)


def calculate_features(pc, idx, eig):
    """kernel.cu:16-104 (float32 like the kernel).  Returns [n,13]."""
    pc = np.asarray(pc, dtype=np.float32)
    e = np.asarray(eig, dtype=np.float32)
    k = idx.shape[1]
    e0, e1, e2 = e[:, 0], e[:, 1], e[:, 2]
    s = e0 + e1 + e2
    prod = e0 * e1 * e2
    with np.errstate(all="ignore"):
        C = e2 / s
        O = np.power((prod / (s ** 3)).astype(np.float64), 1.0 / 3.0).astype(np.float32)
        L = (e0 - e1) / e0
        E = -((e0 / s) * np.log(e0 / s) + (e1 / s) * np.log(e1 / s) + (e2 / s) * np.log(e2 / s))
        P = (e1 - e2) / e0
        S = e2 / e0
        A = (e0 - e2) / e0
        X = s
        D = (3 * k / (4 * np.pi * prod.astype(np.float64))).astype(np.float32)
        S2 = e[:, 3] + e[:, 4]
        L2 = e[:, 4] / e[:, 3]
    nz = pc[idx][:, :, 2]
    mean = nz.sum(1, dtype=np.float32) / np.float32(k)
    dz = (nz - nz.min(1, keepdims=True)).max(1)
    vz = (np.abs(nz - mean[:, None]) ** 2).sum(1, dtype=np.float32) / np.float32(k)
    return np.stack([C, O, L, E, P, S, A, X, D, S2, L2, dz, vz], axis=1).astype(np.float32)
