"""oracle/skinning.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

PyTorch-CPU restatement (plain tensor ops, differentiable) of DreamMesh4D's
sparse-control skinning and face->Gaussian transform:

  node_attributes      custom/threestudio-dreammesh4d/geometry/dynamic_sugar.py:29-39,408-465
  skin_vertices        .../dynamic_sugar.py:487-613  (LBS / DQS / hybrid + Exp(sum w Log q))
                       .../utils/dual_quaternions.py:94-131,184-197,224-231
  static_quaternions   .../geometry/sugar.py:489-518 (+ pytorch3d matrix_to_quaternion)
  face_gaussians       .../dynamic_sugar.py:657-676,726-743,877-889 (means, fused rotations)
  face_normals         .../dynamic_sugar.py:330-364 (pytorch3d face normals, repeat x6)
  static attributes    .../geometry/sugar.py:471-487,640-661 (strengths, scaling, get_points_rgb)

The reference delegates quaternion algebra to `pypose==0.6.7` (requirements.txt:45) and face
normals / matrix_to_quaternion to `pytorch3d` (requirements.txt:46).  Neither package is in
/root/reference nor installed here, so their published algorithms are restated:
  * SO3 Log :  2*atan(|v|/w)/|v| * v   (series 2/w - 2|v|^2/(3 w^3) for tiny |v|)
  * so3 Exp :  (sin(t/2)/t * x, cos(t/2)), t = |x| (series for tiny t)
  * SO3 * SO3: Hamilton product, (x, y, z, w) storage, no re-normalisation
  * matrix(): R p = p + 2 w (v x p) + 2 v x (v x p)
PARITY UNPINNED for those conventions; pinned here (tests/test_oracle_skinning.py) by
closed-form identities: identity deformation == static geometry, single rigid motion
=> LBS == DQS == the rigid transform, Log/Exp round trip, R(q) orthonormal.

GRADIENTS, two conventions (``grad_mode``):
  "exact"   autograd of this file's tensor ops: the exact (Euclidean) gradient of the forward function;
  "pypose"  what the reference's autograd returns.  pypose's LieTensor operations have hand-written backward rules
            (pypose/lietensor/operation.py, release 0.6.7) that hand back LEFT-PERTURBATION tangent gradients zero-padded
            into the quaternion storage and read the first three components of an incoming storage gradient as such a
            tangent gradient:
               SO3_Log.backward   (g Jl^-1(out), 0)                  so3_Exp.backward   g[:3] Jl(x)
               SO3_Act.backward   X: (g (-hat(out)), 0), p: g R(X)    SO3_Mul.backward   X: (g[:3], 0), Y: (g[:3] R(X), 0)
            with Jl the left Jacobian of SO(3).  The custom autograd Functions below restate exactly these rules (forward
            values are the same functions as in "exact" mode), incl. the two SO3 products of the dual-quaternion algebra of
            the DQS branch (q_d = (t / 2) * q_r, translation = (2 q_d) * conj(q_r): pp.SO3 LieTensors of NON-unit quaternions,
            whose SO3_Mul backward is the same rule with SO3_Adj(X) evaluated on X's components as they are).  The torch ops
            around them (F.normalize, .tensor(), divisions by the norm, weighted sums) stay Euclidean.
            pypose is not in the tree: PARITY UNPINNED.
"""
import math

import torch

EPS32 = 1.1920928955078125e-07  # torch.finfo(float32).eps, pypose's branch threshold

BARY6 = [[2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3],
         [1 / 6, 5 / 12, 5 / 12], [5 / 12, 1 / 6, 5 / 12], [5 / 12, 5 / 12, 1 / 6]]  # sugar.py:265-276
SH_C0 = 0.28209479177387814


def bary_table(n=6, dtype=torch.float32):
    if n == 6:
        return torch.tensor(BARY6, dtype=dtype)
    if n == 1:
        return torch.tensor([[1 / 3, 1 / 3, 1 / 3]], dtype=dtype)
    if n == 3:
        return torch.tensor([[1 / 2, 1 / 4, 1 / 4], [1 / 4, 1 / 2, 1 / 4], [1 / 4, 1 / 4, 1 / 2]], dtype=dtype)
    if n == 4:
        return torch.tensor([[1 / 3, 1 / 3, 1 / 3], [2 / 3, 1 / 6, 1 / 6], [1 / 6, 2 / 3, 1 / 6], [1 / 6, 1 / 6, 2 / 3]],
                            dtype=dtype)
    raise ValueError(n)


# ----------------------------------------------------------------------------- quaternion algebra (x,y,z,w)
def so3_log(q):
    v, w = q[..., :3], q[..., 3:]
    vn = v.norm(dim=-1, keepdim=True)
    small = vn < EPS32
    vn_safe = torch.where(small, torch.ones_like(vn), vn)
    generic = 2.0 * torch.atan(vn_safe / w) / vn_safe
    series = 2.0 / w - (2.0 / 3.0) * vn * vn / (w * w * w)
    return torch.where(small, series, generic) * v


def so3_exp(x):
    t = x.norm(dim=-1, keepdim=True)
    small = t < EPS32
    t_safe = torch.where(small, torch.ones_like(t), t)
    imag = torch.where(small, 0.5 - t * t / 48.0 + t ** 4 / 3840.0, torch.sin(0.5 * t_safe) / t_safe)
    real = torch.where(small, 1.0 - t * t / 8.0 + t ** 4 / 384.0, torch.cos(0.5 * t))
    return torch.cat([imag * x, real], dim=-1)


def quat_mul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz], dim=-1)


def quat_conj(q):
    return torch.cat([-q[..., :3], q[..., 3:]], dim=-1)


def quat_act(q, p):
    v, w = q[..., :3], q[..., 3:]
    uv = 2.0 * torch.linalg.cross(v, p, dim=-1)
    return p + w * uv + torch.linalg.cross(v, uv, dim=-1)


def quat_matrix(q):
    eye = torch.eye(3, dtype=q.dtype)
    cols = [quat_act(q, eye[j].expand(q.shape[:-1] + (3,))) for j in range(3)]
    return torch.stack(cols, dim=-1)  # [..., 3, 3], column j = R e_j


# ----------------------------------------------------------------------------- pypose's backward rules
def _hat_row(g, x):
    """g K for K = hat(x): the row vector g x x."""
    return torch.linalg.cross(g, x, dim=-1)


def _row_times_Jl(x, g):
    t2 = (x * x).sum(-1, keepdim=True)
    t = t2.sqrt()
    small = t < 1e-6
    ts = torch.where(small, torch.ones_like(t), t)
    c1 = torch.where(small, 0.5 - t2 / 24.0, (1.0 - torch.cos(ts)) / (ts * ts))
    c2 = torch.where(small, 1.0 / 6.0 - t2 / 120.0, (ts - torch.sin(ts)) / (ts ** 3))
    gk = _hat_row(g, x)
    return g + c1 * gk + c2 * _hat_row(gk, x)


def _row_times_Jl_inv(x, g):
    t2 = (x * x).sum(-1, keepdim=True)
    t = t2.sqrt()
    small = t < 1e-6
    ts = torch.where(small, torch.ones_like(t), t)
    c2 = torch.where(small, 1.0 / 12.0 + t2 / 720.0, (1.0 - 0.5 * ts * (1.0 + torch.cos(ts)) / torch.sin(ts)) / (ts * ts))
    gk = _hat_row(g, x)
    return g - 0.5 * gk + c2 * _hat_row(gk, x)


def _pad0(t):
    return torch.cat([t, torch.zeros_like(t[..., :1])], dim=-1)


class _LogPP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q):
        out = so3_log(q)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        return _pad0(_row_times_Jl_inv(out, g))


class _ExpPP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return so3_exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return _row_times_Jl(x, g[..., :3])


class _ActPP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, p):
        out = quat_act(q, p)
        ctx.save_for_backward(q, out)
        return out

    @staticmethod
    def backward(ctx, g):
        q, out = ctx.saved_tensors
        gq = _pad0(torch.linalg.cross(out, g, dim=-1))                 # g (-hat(out)) = out x g
        gp = quat_act(quat_conj(q), g)                                   # g R(X) = R^T g for a unit quaternion
        return gq, gp


class _MulPP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a)
        return quat_mul(a, b)

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        g3 = g[..., :3]
        return _pad0(g3), _pad0(quat_act(quat_conj(a), g3))


def _ops(grad_mode):
    if grad_mode == "pypose":
        return _LogPP.apply, _ExpPP.apply, _ActPP.apply, _MulPP.apply
    if grad_mode != "exact":
        raise ValueError(grad_mode)
    return so3_log, so3_exp, quat_act, quat_mul


# ----------------------------------------------------------------------------- A2 node attributes
def strain_to_matrix(s):
    """I + sym(ds): diag = ds[0:3], (01)=ds[3], (02)=ds[4], (12)=ds[5]  (dynamic_sugar.py:29-39)."""
    one = torch.ones_like(s[..., 0])
    return torch.stack([one + s[..., 0], s[..., 3], s[..., 4],
                        s[..., 3], one + s[..., 1], s[..., 5],
                        s[..., 4], s[..., 5], one + s[..., 2]], dim=-1).reshape(s.shape[:-1] + (3, 3))


def node_attributes(dx, dr, ds=None, do=None):
    ident = torch.zeros_like(dr)
    ident[..., 3] = 1.0
    rot = torch.nn.functional.normalize(dr + ident, dim=-1)
    S = strain_to_matrix(ds) if ds is not None else None
    op = torch.sigmoid(do) if do is not None else None
    return dx, rot, S, op


# ----------------------------------------------------------------------------- A3 vertex skinning
def skin_vertices(verts, nbr_idx, nbr_w, trans, rot, S=None, opacity=None, method="hybrid", grad_mode="exact"):
    """verts [V,3]; nbr_idx [V,K] long; nbr_w [V,K]; node tables trans [M,3], rot [M,4] (unit, xyzw),
    S [M,3,3], opacity [M,1].  Returns (xyz [V,3], vrot [V,4] xyzw)."""
    log_, exp_, act_, mul_ = _ops(grad_mode)
    t_k = trans[nbr_idx]          # [V,K,3]
    q_k = rot[nbr_idx]            # [V,K,4]
    w = nbr_w[..., None]
    x_lbs = x_dqs = None
    if method in ("lbs", "hybrid"):
        S_k = S[nbr_idx]                                        # [V,K,3,3]
        sv = (S_k @ verts[:, None, :, None]).squeeze(-1)        # S_k v
        x_k = act_(q_k, sv) + t_k                               # rots.matrix() @ (S v): the matrix IS Act on the basis
        x_lbs = (w * x_k).sum(dim=1)
    if method in ("dqs", "hybrid"):
        q_r = q_k / q_k.norm(dim=-1, keepdim=True)
        t4 = torch.cat([t_k, torch.zeros_like(t_k[..., :1])], dim=-1)
        q_d = mul_(0.5 * t4, q_r)                               # pp.SO3(0.5 * pp.SO3(t)) * q_r: SO3_Mul (dual_quaternions.py:124-130)
        br = (q_r * w).sum(dim=1)
        bd = (q_d * w).sum(dim=1)
        nrm = br.norm(dim=-1, keepdim=True)
        br, bd = br / nrm, bd / nrm
        tr = mul_(2.0 * bd, quat_conj(br))[..., :3]             # .translation: pp.SO3(2 q_d) * conj(q_r), SO3_Mul (:224-231)
        x_dqs = act_(br, verts) + tr                            # transform_point_simple: q_r.matrix() @ p + translation
    if method == "lbs":
        xyz = x_lbs
    elif method == "dqs":
        xyz = x_dqs
    elif method == "hybrid":
        eta = (w * opacity[nbr_idx]).sum(dim=1)
        eta = torch.clamp(eta + 0.4, max=1.0)
        xyz = eta * x_lbs + (1 - eta) * x_dqs
    else:
        raise ValueError(method)
    vrot = exp_((w * log_(q_k)).sum(dim=1))
    return xyz, vrot


# ----------------------------------------------------------------------------- A4 static quaternions
def matrix_to_quaternion_wxyz(R):
    """pytorch3d.transforms.matrix_to_quaternion (branch-free, largest-component variant), real part first,
    standardised to w >= 0."""
    m00, m01, m02 = R[..., 0, 0], R[..., 0, 1], R[..., 0, 2]
    m10, m11, m12 = R[..., 1, 0], R[..., 1, 1], R[..., 1, 2]
    m20, m21, m22 = R[..., 2, 0], R[..., 2, 1], R[..., 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                                1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1), min=0.0))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(dim=-1)
    out = torch.gather(cand, -2, best[..., None, None].expand(best.shape + (1, 4))).squeeze(-2)
    return torch.where(out[..., :1] < 0, -out, out)


def face_normals(verts, faces):
    fv = verts[faces]
    n = torch.linalg.cross(fv[:, 1] - fv[:, 0], fv[:, 2] - fv[:, 0], dim=-1)
    return torch.nn.functional.normalize(n, dim=-1)


def static_quaternions(verts, faces, cplx, n_per_face=6):
    """sugar.py:489-518.  Returns [N,4] wxyz unit quaternions (R = [n, R1, R2] columns)."""
    R0 = face_normals(verts, faces)
    fv = verts[faces]
    b1 = torch.nn.functional.normalize(fv[:, 0] - fv[:, 1], dim=-1)
    b2 = torch.nn.functional.normalize(torch.linalg.cross(R0, b1, dim=-1), dim=-1)
    c = torch.nn.functional.normalize(cplx, dim=-1).view(len(faces), n_per_face, 2)
    R1 = c[..., 0:1] * b1[:, None] + c[..., 1:2] * b2[:, None]
    R2 = -c[..., 1:2] * b1[:, None] + c[..., 0:1] * b2[:, None]
    R = torch.stack([R0[:, None].expand(-1, n_per_face, -1), R1, R2], dim=-1).view(-1, 3, 3)
    return torch.nn.functional.normalize(matrix_to_quaternion_wxyz(R), dim=-1)


# ----------------------------------------------------------------------------- A4/A5 face -> Gaussians
def face_gaussians(vxyz, vrot, faces, q_static_wxyz, n_per_face=6, grad_mode="exact"):
    """vxyz [V,3], vrot [V,4] xyzw (deformed).  Returns means [N,3], rotations [N,4] wxyz (unit),
    normals [N,3] (deformed-mesh unit face normals repeated per Gaussian)."""
    bary = bary_table(n_per_face, vxyz.dtype)                   # [G,3]
    fv = vxyz[faces]                                            # [F,3,3]
    means = (fv[:, None] * bary[None, :, :, None]).sum(dim=-2).reshape(-1, 3)
    log_, exp_, _, mul_ = _ops(grad_mode)
    logs = log_(vrot[faces])                                    # [F,3,3]
    r = (logs[:, None] * bary[None, :, :, None]).sum(dim=-2).reshape(-1, 3)
    q_def = exp_(r)                                             # xyzw
    q_st = q_static_wxyz[:, [1, 2, 3, 0]]
    q = mul_(q_def, q_st)[:, [3, 0, 1, 2]]
    q = torch.nn.functional.normalize(q, dim=-1)
    normals = face_normals(vxyz, faces).repeat_interleave(n_per_face, dim=0)
    return means, q, normals


def static_attributes(log_scales, densities, sh_dc, thickness):
    """scaling (sugar.py:479-487), strengths (:471-472), get_points_rgb (:640-661, sh_levels == 1)."""
    scales = torch.cat([thickness * torch.ones_like(log_scales[:, :1]), torch.exp(log_scales)], dim=-1)
    opac = torch.sigmoid(densities.view(-1, 1))
    rgb = (sh_dc * SH_C0 + 0.5).view(-1, 3)
    return scales, opac, rgb


# ----------------------------------------------------------------------------- d_scale branch
def vertex_scales(nbr_idx, nbr_w, S, opacity=None, method="hybrid"):
    """dynamic_sugar.py:593-611: per-vertex scale matrices [V,3,3] from the node strain matrices S [M,3,3] (and, hybrid, the
    node opacities [M,1], with the position blend's lbs weight of :572-578)."""
    Sn = S[nbr_idx]                                          # [V,K,3,3]
    if method == "lbs":
        return (nbr_w[..., None, None] * Sn).sum(dim=-3)
    if method != "hybrid":
        raise ValueError("the reference defines vertex scales for lbs and hybrid only")
    on = opacity[nbr_idx]                                    # [V,K,1]
    lbs_w = torch.clamp((nbr_w[..., None] * on).sum(dim=-2) + 0.4, max=1.0)      # [V,1]
    out = (nbr_w[..., None, None] * on[..., None] * Sn).sum(dim=-3)
    return out + (1.0 - lbs_w)[..., None] * torch.eye(3, dtype=S.dtype)


def gaussian_scales(faces, n_per_face, vertex_scale, scaling):
    """dynamic_sugar.py:697-704: gs_timed_dscale = sum_c bary_c * vert_scale[corner c]; scales = dscale @ scaling.
    faces [F,3], vertex_scale [V,3,3], scaling [N,3] -> [N,3]."""
    bary = bary_table(int(n_per_face), vertex_scale.dtype)                       # [G,3]
    conn = faces.repeat_interleave(int(n_per_face), dim=0)                       # _gs_vert_connections [N,3]
    w = bary.repeat(faces.shape[0], 1)                                           # _gs_bary_weights     [N,3]
    d = (w[..., None, None] * vertex_scale[conn]).sum(dim=-3)                    # [N,3,3]
    return torch.einsum("pij,pj->pi", d, scaling)
