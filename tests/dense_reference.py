"""Independent dense fp64 PyTorch restatement of the splatting math, with autograd.

Used only to pin the C oracle (tests/test_oracle_raster.py): it shares no code
with oracle/raster_oracle.c -- per-pixel blending over ALL Gaussians sorted by
(depth, index), masked by each Gaussian's tile rectangle (integer state computed by
preprocess_fp64 below, independently of the oracle), straight-through handling of the alpha clamp and of the
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
    """All float64 tensors.  radii/rect_*/depth32: the integer / ordering state (preprocess_fp64).
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


def preprocess_fp64(means3D, scales, rots, view, proj, tanfovx, tanfovy, H, W, scale_mod=1.0):
    """Independent float64 numpy statement of the per-Gaussian DECISIONS of the published 3DGS preprocess
    (cull, radius, tile rect, depth): nothing here is taken from oracle/raster_oracle.c.

    Returns a dict with radii, rect_min, rect_max, tiles_touched, depth, xy and `margin` [N]: the distance of the
    closest decision of each Gaussian to its rounding boundary (near-plane test, ceil of 3 sigma, the four integer
    truncations of the rect) in units where float32 rounding of the inputs is ~1e-5.  A float32 implementation may
    legitimately differ from this one only where margin is tiny; tests count those cases."""
    import numpy as np

    m = np.asarray(means3D, np.float64)
    N = m.shape[0]
    V = np.asarray(view, np.float64).reshape(4, 4)          # row-vector convention: p' = p @ V
    P = np.asarray(proj, np.float64).reshape(4, 4)
    ph = np.concatenate([m, np.ones((N, 1))], 1)
    pv = ph @ V
    pp = ph @ P
    pw = 1.0 / (pp[:, 3] + 1e-7)
    ndc = pp[:, :2] * pw[:, None]
    pix = ((ndc + 1.0) * np.array([W, H], np.float64) - 1.0) * 0.5
    q = np.asarray(rots, np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]           # (w, x, y, z), NOT normalised by the rasterizer
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                  2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(N, 3, 3)
    Mx = R * (scale_mod * np.asarray(scales, np.float64))[:, None, :]
    Sigma = Mx @ Mx.transpose(0, 2, 1)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = pv[:, 2]
    safe = np.where(np.abs(tz) > 1e-12, tz, 1e-12)
    tx = np.clip(pv[:, 0] / safe, -1.3 * tanfovx, 1.3 * tanfovx) * tz
    ty = np.clip(pv[:, 1] / safe, -1.3 * tanfovy, 1.3 * tanfovy) * tz
    J = np.zeros((N, 2, 3))
    J[:, 0, 0] = fx / safe
    J[:, 0, 2] = -fx * tx / (safe * safe)
    J[:, 1, 1] = fy / safe
    J[:, 1, 2] = -fy * ty / (safe * safe)
    T = J @ V[:3, :3].T
    cov = T @ Sigma @ T.transpose(0, 2, 1)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    mid = 0.5 * (a + c)
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - det))
    three_sigma = 3.0 * np.sqrt(lam)
    radius = np.ceil(three_sigma)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    lo = np.stack([(pix[:, 0] - radius) / 16.0, (pix[:, 1] - radius) / 16.0], 1)
    hi = np.stack([(pix[:, 0] + radius + 15.0) / 16.0, (pix[:, 1] + radius + 15.0) / 16.0], 1)
    g = np.array([gx, gy])
    rect_min = np.clip(np.trunc(lo), 0, g).astype(np.int64)      # (int) truncates toward zero
    rect_max = np.clip(np.trunc(hi), 0, g).astype(np.int64)
    area = (rect_max[:, 0] - rect_min[:, 0]) * (rect_max[:, 1] - rect_min[:, 1])
    visible = (tz > 0.2) & (det != 0.0) & (area > 0)
    frac = lambda v: np.abs(v - np.round(v))
    # only truncations that are not already decided by the clamp to [0, grid] matter
    def trunc_margin(v, gmax):
        inside = (v > -1.0) & (v < gmax + 1.0)
        return np.where(inside, frac(v) * 16.0, np.inf)
    margin = np.minimum.reduce([np.abs(tz - 0.2) * 1e3, frac(three_sigma) , trunc_margin(lo[:, 0], gx), trunc_margin(lo[:, 1], gy),
                                trunc_margin(hi[:, 0], gx), trunc_margin(hi[:, 1], gy)])
    return {"radii": np.where(visible, radius, 0).astype(np.int64), "rect_min": rect_min, "rect_max": rect_max,
            "tiles_touched": np.where(visible, area, 0).astype(np.int64), "depth": tz, "xy": pix, "margin": margin,
            "visible": visible}
