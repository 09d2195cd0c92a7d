"""Host-side (PyTorch, any device) mirror of the static SuGaR properties the hot path consumes.

These are cheap elementwise / gather expressions evaluated once per iteration (static stage) or
once at set-up (dynamic stage, where the static parameters are frozen:
custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:77-87), so they stay PyTorch ops; the
per-view work is in the HIP kernels (ops.py / rasterizer).

  bary_coords          geometry/sugar.py:235-276
  points               geometry/sugar.py:439-455
  scaling              geometry/sugar.py:478-487
  strengths            geometry/sugar.py:470-472
  quaternions          geometry/sugar.py:489-518  (pytorch3d matrix_to_quaternion restated)
  points_rgb           geometry/sugar.py:640-661  (sh_levels == 1), gaussian_base.py:30-40 (SH2RGB)
  face_normals         geometry/sugar.py:520-526
"""
import torch
import torch.nn.functional as F

C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5


_BARY = {
    1: [[1 / 3, 1 / 3, 1 / 3]],
    3: [[1 / 2, 1 / 4, 1 / 4], [1 / 4, 1 / 2, 1 / 4], [1 / 4, 1 / 4, 1 / 2]],
    4: [[1 / 3, 1 / 3, 1 / 3], [2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]],
    6: [[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3],
        [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]],
}


def bary_coords(n_per_face, device=None, dtype=torch.float32):
    """[G, 3, 1] like SuGaRModel.surface_triangle_bary_coords."""
    return torch.tensor(_BARY[int(n_per_face)], dtype=dtype, device=device)[..., None]


def circle_radius(n_per_face, init_gs_scales_s=1.0):
    """``surface_triangle_circle_radius`` (sugar.py:236-267): radius of the circles packed into a unit triangle."""
    import math

    r = {1: 1.0 / 2.0 / math.sqrt(3.0), 3: 1.0 / 2.0 / (math.sqrt(3.0) + 1.0), 4: 1.0 / (4.0 * math.sqrt(3.0)),
         6: 1.0 / (4.0 + 2.0 * math.sqrt(3.0))}[int(n_per_face)]
    return r * init_gs_scales_s


def face_verts(verts, faces):
    """verts[faces] -> [F,3,3] as ONE gather whose backward is one index_add launch (advanced indexing differentiates through a
    sort-based index_put: ~140 us per use at 16.7k faces, four uses per evaluation of the static geometry).  The additions of a
    vertex's corners then come in the order the hardware schedules them (float atomics) unless
    torch.use_deterministic_algorithms(True) is set."""
    return verts.index_select(0, faces.reshape(-1)).view(faces.shape[0], 3, verts.shape[-1])


def points(verts, faces, bary, fv=None):
    fv = face_verts(verts, faces) if fv is None else fv  # [F,3,3]
    return (fv[:, None] * bary[None]).sum(dim=-2).reshape(-1, 3)


def scaling(log_scales, thickness):
    return torch.cat([thickness * torch.ones(len(log_scales), 1, device=log_scales.device, dtype=log_scales.dtype),
                      torch.exp(log_scales)], dim=-1)


def strengths(densities):
    return torch.sigmoid(densities.view(-1, 1))


def points_rgb(sh_dc):
    return SH2RGB(sh_dc).view(-1, 3)


def face_normals(verts, faces, fv=None):
    fv = face_verts(verts, faces) if fv is None else fv
    return F.normalize(torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1), dim=-1)


def matrix_to_quaternion(R):
    """Rotation matrices [...,3,3] -> quaternions (w,x,y,z): largest-component construction of
    pytorch3d.transforms.matrix_to_quaternion, standardised to w >= 0."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    x = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
    # pytorch3d's _sqrt_positive_part: sqrt with a ZERO subgradient where x <= 0 (sqrt(clamp(x, 0)) back-propagates
    # inf * 0 = NaN there -- it broke the static stage, where the vertices are learnt, within three iterations)
    pos = x > 0
    q_abs = torch.where(pos, torch.sqrt(torch.where(pos, x, torch.ones_like(x))), torch.zeros_like(x))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(dim=-1)
    out = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return torch.where(out[..., :1] < 0, -out, out)


def quaternions(verts, faces, complex_numbers, n_per_face, fv=None, normals=None):
    """Static Gaussian orientations: first axis = face normal, second = first triangle side rotated in the
    face plane by the learnt 2-D rotation.  Returns [N,4] (w,x,y,z), unit.  `fv` / `normals`: face_verts / face_normals of the
    same mesh when the caller already has them."""
    fv = face_verts(verts, faces) if fv is None else fv
    R0 = face_normals(verts, faces, fv=fv) if normals is None else normals
    b1 = F.normalize(fv[:, 0] - fv[:, 1], dim=-1)
    b2 = F.normalize(torch.linalg.cross(R0, b1, dim=-1), dim=-1)
    c = F.normalize(complex_numbers, dim=-1).view(len(faces), n_per_face, 2)
    R1 = c[..., 0:1] * b1[:, None] + c[..., 1:2] * b2[:, None]
    R2 = -c[..., 1:2] * b1[:, None] + c[..., 0:1] * b2[:, None]
    R = torch.stack([R0[:, None].expand(-1, n_per_face, -1), R1, R2], dim=-1).view(-1, 3, 3)
    return F.normalize(matrix_to_quaternion(R), dim=-1)
