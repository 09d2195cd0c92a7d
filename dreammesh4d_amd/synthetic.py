"""Seeded synthetic inputs for the hot path.

No image files, no checkpoints: everything is generated on the host from
``numpy.random.default_rng(seed)`` so tests, bench.py and the CPU baseline see
identical inputs.

* random-splat scene  (BASELINE.json configs[0] and the headline 200k/512^2 raster number)
* mesh-bound scene    (configs[1..4]): UV-sphere re-tessellated to F faces, 6 Gaussians/face,
  M deformation-graph nodes, K-NN skin weights ``(1 - d_k/d_{K+1})^2`` row-normalised
  (reference: custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:845,859-861)
* orbit cameras following threestudio's random-camera convention
  (reference: custom/threestudio-dreammesh4d/data/uncond.py, threestudio/utils/ops.py:359-413)
"""
import math
from dataclasses import dataclass

import numpy as np

THICKNESS = 3.8 / 1_000_000  # spatial_extent / 1e6  (geometry/sugar.py:191, configs/sugar_dynamic_dg.yaml:80)


# ----------------------------------------------------------------------------- cameras
def orbit_c2w(elev_deg: float, azim_deg: float, dist: float) -> np.ndarray:
    """threestudio camera: position on a sphere, looking at the origin, world up = +z,
    OpenGL camera axes (right, up, -lookat)."""
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    pos = np.array([dist * math.cos(el) * math.cos(az), dist * math.cos(el) * math.sin(az), dist * math.sin(el)])
    lookat = -pos / np.linalg.norm(pos)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(lookat, up)
    right /= np.linalg.norm(right)
    up = np.cross(right, lookat)
    up /= np.linalg.norm(up)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, up, -lookat, pos
    return c2w


def gaussian_camera(c2w: np.ndarray, fovy: float, znear: float = 0.1, zfar: float = 100.0):
    """(world_view_transform, full_proj_transform, camera_center) in the row-vector
    ("transposed") convention the rasterizer expects.  Same math as
    threestudio/utils/ops.py:398-413 (get_cam_info_gaussian) -- fovx == fovy there
    (renderer/gaussian_batch_renderer.py:24-26)."""
    c2w = np.asarray(c2w, np.float64) @ np.diag([1.0, -1.0, -1.0, 1.0])
    wvt = np.linalg.inv(c2w).T.astype(np.float32)
    t = math.tan(fovy / 2)
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (2.0 * t * znear)
    P[1, 1] = 2.0 * znear / (2.0 * t * znear)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    full = (wvt @ P.T).astype(np.float32)
    cam = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return wvt, full, cam


@dataclass
class Camera:
    H: int
    W: int
    fovy: float
    tanfov: float
    c2w: np.ndarray
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    campos: np.ndarray


def make_camera(H: int, W: int, elev_deg=15.0, azim_deg=0.0, dist=3.8, fovy_deg=20.0) -> Camera:
    fovy = math.radians(fovy_deg)
    c2w = orbit_c2w(elev_deg, azim_deg, dist)
    v, p, c = gaussian_camera(c2w, fovy)
    return Camera(H, W, fovy, math.tan(fovy * 0.5), c2w, v, p, c)


# ----------------------------------------------------------------------------- random splats
def random_splat_scene(n: int, seed: int = 0, radius: float = 0.6, log_scale_mean: float = math.log(0.004),
                       log_scale_std: float = 0.35):
    """means ~ U(ball r), scales = (thickness, e^a, e^b), rotations = normalised N(0,I),
    opacity = sigmoid(N(2,1)), rgb ~ U(0,1).  float32 numpy arrays."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = radius * rng.random(n) ** (1.0 / 3.0)
    means = (d * r[:, None]).astype(np.float32)
    ab = rng.normal(log_scale_mean, log_scale_std, size=(n, 2))
    scales = np.concatenate([np.full((n, 1), THICKNESS), np.exp(ab)], axis=1).astype(np.float32)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(2.0, 1.0, size=n)))).astype(np.float32)
    rgb = rng.random((n, 3)).astype(np.float32)
    return {"means3D": means, "scales": scales, "rotations": q.astype(np.float32), "opacities": opac, "colors": rgb}


# ----------------------------------------------------------------------------- mesh-bound scene
def uv_sphere(n_faces: int, radius: float = 0.6):
    """Closed UV-sphere with EXACTLY n_faces triangles when n_faces is even and >= 8
    (2*nlon*(nlat-1) faces); otherwise the nearest count from below is produced and the
    caller reads the real count from the arrays."""
    best = None
    target = max(8, n_faces)
    nlat0 = max(3, int(round(math.sqrt(target / 4.0))))
    for nlat in range(max(3, nlat0 - 40), nlat0 + 41):
        nlon = target // (2 * (nlat - 1))
        if nlon < 3:
            continue
        f = 2 * nlon * (nlat - 1)
        score = (target - f, abs(nlon - 2 * nlat))
        if f <= target and (best is None or score < best[0]):
            best = (score, nlat, nlon)
    _, nlat, nlon = best
    verts = [(0.0, 0.0, radius)]
    for i in range(1, nlat):
        th = math.pi * i / nlat
        for j in range(nlon):
            ph = 2 * math.pi * j / nlon
            verts.append((radius * math.sin(th) * math.cos(ph), radius * math.sin(th) * math.sin(ph), radius * math.cos(th)))
    verts.append((0.0, 0.0, -radius))
    faces = []
    ring = lambda i, j: 1 + (i - 1) * nlon + (j % nlon)
    for j in range(nlon):
        faces.append((0, ring(1, j), ring(1, j + 1)))
    for i in range(1, nlat - 1):
        for j in range(nlon):
            a, b, c, d = ring(i, j), ring(i, j + 1), ring(i + 1, j), ring(i + 1, j + 1)
            faces.append((a, c, b))
            faces.append((b, c, d))
    south = len(verts) - 1
    for j in range(nlon):
        faces.append((south, ring(nlat - 1, j + 1), ring(nlat - 1, j)))
    return np.asarray(verts, np.float32), np.asarray(faces, np.int64)


def knn_skin_weights(verts: np.ndarray, nodes: np.ndarray, k: int):
    """K Euclidean-nearest nodes per vertex and weights (1 - d_k/d_{K+1})^2, row-normalised
    (dynamic_sugar.py:845,859-861).  Returns (idx[V,K] int64, w[V,K] float32)."""
    from scipy.spatial import cKDTree

    tree = cKDTree(nodes.astype(np.float64))
    d, idx = tree.query(verts.astype(np.float64), k=k + 1)
    w = (1.0 - d[:, :k] / np.maximum(d[:, k:k + 1], 1e-12)) ** 2
    w = w / np.maximum(w.sum(axis=1, keepdims=True), 1e-20)
    return idx[:, :k].astype(np.int64), w.astype(np.float32)


def mesh_bound_scene(n_faces: int, n_nodes: int = 1000, k: int = 4, seed: int = 0, radius: float = 0.6):
    """Static SuGaR parameters + deformation graph for a sphere mesh."""
    rng = np.random.default_rng(seed)
    verts, faces = uv_sphere(n_faces, radius)
    F = faces.shape[0]
    N = 6 * F
    # nodes: seeded uniform surface samples
    d = rng.normal(size=(n_nodes, 3))
    nodes = (radius * d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    idx, w = knn_skin_weights(verts, nodes, k)
    # SuGaR init scales: min edge length * circle radius (sugar.py:266,309-313)
    fv = verts[faces]
    edge = np.linalg.norm(fv - fv[:, [1, 2, 0]], axis=-1).min(axis=-1)
    circ = 1.0 / (4.0 + 2.0 * math.sqrt(3.0)) * 1.7
    sc = np.maximum(edge * circ, 1e-7)
    log_scales = np.log(np.repeat(sc[:, None], 6, axis=0).repeat(2, axis=1)).astype(np.float32)
    log_scales += rng.normal(0, 0.1, size=log_scales.shape).astype(np.float32)
    cplx = np.zeros((N, 2), np.float32)
    ang = rng.uniform(-0.3, 0.3, size=N)
    cplx[:, 0], cplx[:, 1] = np.cos(ang), np.sin(ang)
    dens = rng.normal(2.0, 1.0, size=(N, 1)).astype(np.float32)
    sh_dc = ((rng.random((N, 1, 3)) - 0.5) / 0.28209479177387814).astype(np.float32)
    return {"verts": verts, "faces": faces, "nodes": nodes, "nbr_idx": idx, "nbr_w": w,
            "log_scales": log_scales, "complex": cplx, "densities": dens, "sh_dc": sh_dc, "n_gaussians": N}


def node_motion(n_nodes: int, n_frames: int, seed: int = 0, max_rot: float = 0.3, max_trans: float = 0.1,
                max_strain: float = 0.05):
    """Seeded smooth per-node motion standing in for the HexPlane output:
    d_rot (xyzw delta, added to identity then normalised by the caller), translation,
    strain 6-vector, opacity logit -- one set per frame.  timestamps = linspace(0,1,L+2)[1:-1]
    (data/temporal_image.py:155-158)."""
    rng = np.random.default_rng(seed + 7)
    ts = np.linspace(0.0, 1.0, n_frames + 2)[1:-1].astype(np.float32)
    axis = rng.normal(size=(n_nodes, 3))
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    ang = rng.uniform(0, max_rot, size=(n_nodes, 1))
    tr = rng.uniform(-1, 1, size=(n_nodes, 3)) * max_trans / math.sqrt(3)
    st = rng.uniform(-1, 1, size=(n_nodes, 6)) * max_strain
    op = rng.normal(0, 1, size=(n_nodes, 1))
    out = []
    for t in ts:
        half = 0.5 * ang * t
        q = np.concatenate([axis * np.sin(half), np.cos(half)], axis=1)  # xyzw
        dr = q - np.array([0, 0, 0, 1.0])
        out.append({"trans": (tr * t).astype(np.float32), "d_rot": dr.astype(np.float32),
                    "strain": (st * t).astype(np.float32), "d_opacity": op.astype(np.float32)})
    return ts, out
