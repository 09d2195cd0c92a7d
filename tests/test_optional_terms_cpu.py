"""Value tests of the dynamic stage's OPTIONAL loss terms (dreammesh4d_amd/dynamic_stage.py::_optional_terms; the reference writes them
at custom/threestudio-dreammesh4d/system/sugar_4dgen.py:181-300) against independent float64 compositions in numpy on a small synthetic
batch -- round 5 only checked that they run and are finite.  Plain torch operators: runs on CPU.
The scene is a PLANE seen by three cameras, which pins the geometry terms in closed form: the normals derived from the depth image equal
the plane's normal away from the image border, so normal_depth_consistency is 0 there."""
import types

import numpy as np
import torch

from dreammesh4d_amd import renderer as R
from dreammesh4d_amd.dynamic_stage import DynamicStage

H = W = 24
TANFOV = float(np.tan(np.deg2rad(10.0)))


def _look_at(eye):
    """camera-to-world of an OpenGL camera (x right, y up, looking down -z) at `eye`, looking at the origin."""
    eye = np.asarray(eye, np.float64)
    f = -eye / np.linalg.norm(eye)
    r = np.cross(f, [0.0, 0.0, 1.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, u, -f, eye
    return c2w


def _batch(seed=0):
    rng = np.random.default_rng(seed)
    eyes = [(3.8, 0.3, 0.5), (3.5, -1.2, 0.9), (3.0, 1.5, 1.4)]
    c2w = np.stack([_look_at(e) for e in eyes])
    n_pl = np.array([1.0, 0.15, 0.25]); n_pl /= np.linalg.norm(n_pl)          # the plane n . x = 0.1, facing the cameras
    d0 = 0.1
    dirs = R.ray_directions(H, W, 0.5 * H / TANFOV).double().numpy()
    rd = np.einsum("hwj,bij->bhwi", dirs, c2w[:, :3, :3])
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)
    ro = np.broadcast_to(c2w[:, None, None, :3, 3], rd.shape)
    t = (d0 - ro @ n_pl) / (rd @ n_pl)                                          # distance along the (unit) ray to the plane
    B = len(eyes)
    depth = t[:, None].astype(np.float32)
    alpha = np.ones((B, 1, H, W), np.float32)
    rgb = rng.random((B, 3, H, W)).astype(np.float32) * 1.2 - 0.1              # some values outside [0, 1]: comp_rgb clamps
    return c2w, n_pl, rd, ro, depth, alpha, rgb


def _depth_normal64(ro, rd, depth):
    """Depth2Normal in float64: xyz = o + t d, zero-padded central differences, -cross(d/dx, d/dy), normalised (eps 1e-12)."""
    xyz = ro + depth[..., None] * rd                                             # [B,H,W,3]
    p = np.pad(xyz, ((0, 0), (1, 1), (1, 1), (0, 0)))
    ddx = p[:, 1:-1, 2:] - p[:, 1:-1, :-2]
    ddy = p[:, 2:, 1:-1] - p[:, :-2, 1:-1]
    n = -np.cross(ddx, ddy)
    return n / np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)


def _tv64(x):
    """threestudio/utils/loss.py:8-16 on [B,H,W,C] in float64."""
    b, h, w, c = x.shape
    h_tv = ((x[:, 1:] - x[:, :-1]) ** 2).sum()
    w_tv = ((x[:, :, 1:] - x[:, :, :-1]) ** 2).sum()
    return 2 * (h_tv / (c * (h - 1) * w) + w_tv / (c * h * (w - 1))) / b


def _stage(lam, ref_masks, ref_depths, ref_normals):
    st = types.SimpleNamespace(lam=lam, dev=torch.device("cpu"), r=types.SimpleNamespace(H=H, W=W), ref_camera=types.SimpleNamespace(tanfov=TANFOV),
                               ref_masks=ref_masks, ref_depths=ref_depths, ref_normals=ref_normals)
    st._weight_is_set = lambda name: DynamicStage._weight_is_set(st, name)
    return st


def test_optional_terms_against_float64_compositions():
    c2w, n_pl, rd, ro, depth, alpha, rgb = _batch()
    B = depth.shape[0]
    nd64 = _depth_normal64(ro, rd, depth[:, 0].astype(np.float64))
    # the blended normal image = the orientation the depth image implies (interior pixels agree with +-n_pl; the sign is the formula's)
    sign = np.sign(nd64[0, H // 2, W // 2] @ n_pl)
    assert abs(abs(nd64[0, H // 2, W // 2] @ n_pl) - 1.0) < 1e-6                 # a plane's depth normals ARE its normal
    nrm = np.broadcast_to((sign * n_pl)[None, :, None, None], (B, 3, H, W)).astype(np.float32) * 0.7      # un-normalised: compose normalises
    rng = np.random.default_rng(1)
    nrm_noisy = nrm + 0.2 * rng.standard_normal(nrm.shape).astype(np.float32)
    T = torch.tensor
    vxyz = T(rng.standard_normal((2, 50, 3)).astype(np.float32))
    ref_masks = T((rng.random((4, H, W, 1)) > 0.3).astype(np.float32))
    ref_depths = T((0.5 + rng.random((4, H, W, 1))).astype(np.float32))
    ref_normals = T(rng.random((4, H, W, 3)).astype(np.float32))
    lam = {k: 1.0 for k in ("depth", "depth_rel", "normal", "normal_smooth", "rgb_tv", "depth_tv", "normal_tv", "normal_depth_consistency", "obj_centric")}
    lam["normal_tv"] = [0, 0.0, 0.0, 100, 2.0, 200]                            # piecewise C(): off at first, on later -- still a set weight
    st = _stage(lam, ref_masks, ref_depths, ref_normals)
    b = {"c2w": T(c2w.astype(np.float32)), "ref_idx": T([0]), "rnd_idx": T([1, 2]), "n_ref": 1, "fidx_ref": T([2])}

    def run(normal_img):
        out = {"color": T(np.concatenate([rgb, normal_img], 1)), "depth": T(depth), "alpha": T(alpha), "vxyz": vxyz}
        return {k: float(v) for k, v in DynamicStage._optional_terms(st, out, b, 0).items()}

    res = run(nrm_noisy)
    # ---- independent float64 compositions ----
    n64 = nrm_noisy.astype(np.float64).transpose(0, 2, 3, 1)
    n64 = n64 / np.maximum(np.linalg.norm(n64, axis=-1, keepdims=True), 1e-12)
    comp_normal = n64 * 0.5 + 0.5                                                # alpha = 1
    comp_rgb = np.clip(rgb.astype(np.float64), 0, 1).transpose(0, 2, 3, 1)
    comp_depth = depth.astype(np.float64).transpose(0, 2, 3, 1)
    comp_nd = nd64 * 0.5 + 0.5
    want = {}
    for name, idx in (("ref", [0]), ("zero123", [1, 2])):
        cn = comp_normal[idx]
        want[f"normal_smooth/{name}"] = ((cn[:, 1:] - cn[:, :-1]) ** 2).mean() + ((cn[:, :, 1:] - cn[:, :, :-1]) ** 2).mean()
        want[f"rgb_tv/{name}"], want[f"depth_tv/{name}"], want[f"normal_tv/{name}"] = _tv64(comp_rgb[idx]), _tv64(comp_depth[idx]), _tv64(cn)
        want[f"normal_depth_consistency/{name}"] = (1 - ((2 * cn - 1) * (2 * comp_nd[idx] - 1)).sum(-1)).mean()
        v = vxyz.double().numpy()
        want[f"obj_centric/{name}"] = abs(v[..., 0].mean()) + abs(v[..., 1].mean())
    m = ref_masks[2].numpy()[..., 0] > 0.5
    gt, pred = ref_depths[2].double().numpy()[..., 0][m], comp_depth[0, ..., 0][m]
    A = np.stack([gt, np.ones_like(gt)], -1)
    sol = np.linalg.lstsq(A, pred, rcond=None)[0]                                # scale / shift of the ground truth onto the prediction
    want["depth/ref"] = ((A @ sol - pred) ** 2).mean()
    want["depth_rel/ref"] = 1 - np.corrcoef(pred, gt)[0, 1]
    gtn = 1 - 2 * ref_normals[2].double().numpy()[m]
    prn = 2 * comp_normal[0][m] - 1
    cos = (gtn * prn).sum(-1) / (np.maximum(np.linalg.norm(gtn, axis=-1), 1e-8) * np.maximum(np.linalg.norm(prn, axis=-1), 1e-8))
    want["normal/ref"] = 1 - cos.mean()
    assert set(res) == set(want), sorted(set(res) ^ set(want))
    for k, w_ in want.items():
        assert abs(res[k] - w_) <= 2e-5 * max(1.0, abs(w_)) + 2e-6, (k, res[k], w_)

    # ---- the plane itself: the rendered normals equal the depth image's normals away from the border => consistency 0 there ----
    out = {"color": T(np.concatenate([rgb, nrm], 1)), "depth": T(depth), "alpha": T(alpha)}
    rays_o, rays_d = R.rays(R.ray_directions(H, W, 0.5 * H / TANFOV), b["c2w"])
    img = R.compose_outputs(out["color"], out["depth"], out["alpha"], rays_o, rays_d)
    dot = ((img["comp_normal"] * 2 - 1) * (img["comp_normal_from_dist"] * 2 - 1)).sum(-1)
    assert float((1 - dot[:, 1:-1, 1:-1]).abs().max()) < 2e-3                  # float32 differences of ~4-unit depths over a 0.03-unit pixel
    assert float((1 - dot).mean()) > 0.0                                         # (the zero-padded border rows are NOT consistent: the reference's padding)
    border = 1.0 - (H - 2) * (W - 2) / (H * W)                                   # only the border ring contributes (each pixel at most 2)
    assert run(nrm)["normal_depth_consistency/ref"] <= 2.0 * border + 2e-3


def test_weight_is_set_reads_every_value_of_a_schedule():
    """a weight that C() switches on later (3-entry, 4-entry and piecewise lists) is a set weight."""
    st = types.SimpleNamespace(lam={"a": 0, "b": 0.0, "c": None, "d": [0, 0.0, 0.0, 100], "e": [0.0, 1.0, 100], "f": [0, 0.0, 0.0, 100, 2.0, 200],
                                    "g": [0, 0.0, 0.0, 100, 0.0, 200], "h": [1, 2], "i": 3.0, "j": [10, 0.0, 0.5, 100]})
    got = {k: DynamicStage._weight_is_set(st, k) for k in st.lam}
    assert got == {"a": False, "b": False, "c": False, "d": False, "e": True, "f": True, "g": False, "h": True, "i": True, "j": True}
