"""Independent dense fp64 PyTorch restatement of the splatting math, with autograd.

Used only to pin the C oracle (tests/test_oracle_raster.py): it shares no code
with oracle/raster_oracle.c -- per-pixel blending over ALL Gaussians sorted by
(depth, index), masked by each Gaussian's tile rectangle (which is integer state
taken from the oracle), straight-through handling of the alpha clamp and of the
1.3*tanfov clamp exactly as the published CUDA backward treats them.
"""
import torch


def quat_to_R(q):
    r, x, y, z = q.unbind(-1)
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(*q.shape[:-1], 3, 3)


def dense_rasterize(means3D, means2D, opac, colors, scales, rots, view, proj, bg, tanfovx, tanfovy, H, W,
                    scale_mod, radii, rect_min, rect_max, depth32, cov3D_precomp=None):
    """All float64 tensors.  radii/rect_*/depth32 come from the oracle (integer / ordering state).
    Returns color[3,H,W], depth[H,W], alpha[H,W]."""
    dd = dict(dtype=torch.float64)
    N = means3D.shape[0]
    V = view.reshape(4, 4)
    P = proj.reshape(4, 4)
    pv = means3D @ V[:3, :3] + V[3, :3]
    ph = torch.cat([means3D, torch.ones(N, 1, **dd)], 1) @ P
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None] + means2D[:, :2]
    pix = ((ndc + 1.0) * torch.tensor([W, H], **dd) - 1.0) * 0.5

    if cov3D_precomp is None:
        R = quat_to_R(rots)
        M = R * (scale_mod * scales)[:, None, :]
        Sigma = M @ M.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], -1).reshape(N, 3, 3)
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    tz = pv[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    xcl = (txtz < -limx) | (txtz > limx)
    ycl = (tytz < -limy) | (tytz > limy)
    # upstream backward treats the clamped t.x as a constant (x_grad_mul = 0 and no tz dependence)
    tcx = torch.where(xcl, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    tcy = torch.where(ycl, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    z0 = torch.zeros_like(tz)
    J = torch.stack([fx / tz, z0, -(fx * tcx) / (tz * tz), z0, fy / tz, -(fy * tcy) / (tz * tz)], -1).reshape(N, 2, 3)
    Wm = V[:3, :3].T
    T = J @ Wm
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    cA, cB, cC = c / det, -b / det, a / det

    ys, xs = torch.meshgrid(torch.arange(H, **dd), torch.arange(W, **dd), indexing="ij")
    tyi, txi = (ys // 16).long(), (xs // 16).long()
    order = sorted([i for i in range(N) if radii[i] > 0], key=lambda i: (float(depth32[i]), i))
    Tt = torch.ones(H, W, **dd)
    done = torch.zeros(H, W, dtype=torch.bool)
    C = torch.zeros(3, H, W, **dd)
    D = torch.zeros(H, W, **dd)
    A = torch.zeros(H, W, **dd)
    for g in order:
        inrect = (txi >= rect_min[g][0]) & (txi < rect_max[g][0]) & (tyi >= rect_min[g][1]) & (tyi < rect_max[g][1])
        dx = pix[g, 0] - xs
        dy = pix[g, 1] - ys
        power = -0.5 * (cA[g] * dx * dx + cC[g] * dy * dy) - cB[g] * dx * dy
        G = torch.exp(power)
        araw = opac[g] * G
        alpha = araw + (araw.clamp(max=0.99) - araw).detach()
        valid = inrect & (power <= 0) & (alpha >= 1.0 / 255.0) & ~done
        test_T = Tt * (1 - alpha)
        stop = valid & (test_T < 1e-4)
        done = done | stop
        contrib = valid & ~stop
        w = torch.where(contrib, alpha * Tt, torch.zeros_like(Tt))
        C = C + colors[g][:, None, None] * w
        D = D + pv[g, 2] * w
        A = A + w
        Tt = torch.where(contrib, test_T, Tt)
    C = C + Tt * bg[:, None, None]
    return C, D, A
