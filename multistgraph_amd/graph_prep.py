"""One-off host graph preparation for the model constructor (numpy; not a kernel).

Restates what the reference constructor does before any tensor touches the device
(libcity/model/traffic_flow_prediction/MultiATGCN.py:15-56, 238-283): the OD, distance and
similarity adjacencies and their scaled Laplacians.  With lambda_max fixed at 2 and directed
graphs, ``2L/lambda_max - I`` collapses to ``-D^-1/2 A^T D^-1/2`` (D = row sums), so no sparse
algebra is needed.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np

EARTH_RADIUS_KM = 6371.0


def od_normalise(adj_mx: np.ndarray) -> np.ndarray:
    """adj / diag(adj) with torch broadcasting semantics: entry (i, j) is divided by adj[j, j];
    values above 1 are clamped (reference :238-241)."""
    a = np.asarray(adj_mx, dtype=np.float32)
    a = a / np.diagonal(a)[None, :]
    return np.minimum(a, np.float32(1.0)).astype(np.float32)


def lonlat_table(coordinate) -> np.ndarray:
    """(N, 2) array of [lon, lat], rows ordered by geo_id (the reference pivots on geo_id, :256-260)."""
    ids = np.asarray(coordinate["geo_id"])
    pts = np.empty((len(ids), 2), dtype=np.float64)
    for row, text in enumerate(coordinate["coordinates"]):
        lon, lat = text.replace("[", " ").replace("]", " ").split(",")[:2]
        pts[row] = (float(lon), float(lat))
    return pts[np.argsort(ids, kind="stable")]


def haversine_km(lonlat: np.ndarray) -> np.ndarray:
    """All-pairs great-circle distance (reference haversine_array, :41-48)."""
    lam = np.radians(lonlat[:, 0])
    phi = np.radians(lonlat[:, 1])
    half_dphi = 0.5 * (phi[None, :] - phi[:, None])
    half_dlam = 0.5 * (lam[None, :] - lam[:, None])
    inner = np.sin(half_dphi) ** 2 + np.cos(phi)[:, None] * np.cos(phi)[None, :] * np.sin(half_dlam) ** 2
    return 2.0 * EARTH_RADIUS_KM * np.arcsin(np.sqrt(inner))


def gaussian_kernel(dist: np.ndarray, eps: float = 0.1) -> np.ndarray:
    """exp(-(d / std)^2), thresholded at eps (reference calculate_adjacency_matrix_dist, :51-56)."""
    finite = dist[~np.isinf(dist)]
    w = np.exp(-np.square(dist / finite.std()))
    w[w < eps] = 0.0
    return w.astype(np.float32)


def inverse_euclid(static: Optional[np.ndarray], n: int) -> np.ndarray:
    """Similarity adjacency: 1/||s_i - s_j|| (0 -> 1), or the identity without static features (:244-250)."""
    if static is None:
        return np.eye(n, dtype=np.float32)
    s = np.asarray(static, dtype=np.float64)
    dist = np.empty((s.shape[0], s.shape[0]), dtype=np.float64)
    for r0 in range(0, s.shape[0], 64):     # direct differences like scipy's cdist (exact zeros on the diagonal and for
        diff = s[r0:r0 + 64, None, :] - s[None, :, :]                      # duplicate rows), a block of rows at a time
        dist[r0:r0 + 64] = np.sqrt((diff * diff).sum(-1))
    dist[dist == 0] = 1.0
    return (1.0 / dist).astype(np.float32)


def chebyshev_first_order(adj: np.ndarray) -> np.ndarray:
    """Scaled Laplacian with lambda_max = 2 of a directed graph: -D^-1/2 A^T D^-1/2 (:15-38)."""
    a = np.asarray(adj, dtype=np.float32)
    deg = a.sum(axis=1)
    scale = np.zeros_like(deg)
    nz = deg > 0
    scale[nz] = deg[nz] ** np.float32(-0.5)
    inner = (a * scale[None, :]).T * scale[None, :]
    n = a.shape[0]
    eye = np.eye(n)
    return ((eye - inner.astype(np.float64)) - eye).astype(np.float32)


def build_static_supports(adj_mx, coordinate, static, adjtype: str) -> List[np.ndarray]:
    """First-order static supports in stack order for ``adjtype`` (:264-283)."""
    n = int(np.asarray(adj_mx).shape[0])
    if adjtype == "identity":
        return [np.eye(n, dtype=np.float32)]
    if adjtype == "od":
        return [chebyshev_first_order(od_normalise(adj_mx))]
    if adjtype == "dist":
        return [chebyshev_first_order(gaussian_kernel(haversine_km(lonlat_table(coordinate))))]
    if adjtype == "cosine":
        return [chebyshev_first_order(inverse_euclid(static, n))]
    if adjtype == "multi":
        return [chebyshev_first_order(od_normalise(adj_mx)),
                chebyshev_first_order(gaussian_kernel(haversine_km(lonlat_table(coordinate)))),
                chebyshev_first_order(inverse_euclid(static, n))]
    raise ValueError("adjtype must be one of multi/od/dist/cosine/identity, got %r" % (adjtype,))
