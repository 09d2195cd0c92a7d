#!/usr/bin/env python
"""Generates the golden fixtures under tests/golden/ from the REFERENCE's own code.

Run in the authoring container only (needs /root/reference, which does not exist on the GPU box):
    python tests/golden/make_golden.py
Nothing from /root/reference is copied: the fixtures are data (seeded inputs, a small random
state, and the outputs / gradients the reference computes for them).

  deformation_small.npz   custom/threestudio-dreammesh4d/geometry/deformation.py (imports as-is:
                          torch only) -- DeformationNetwork with a reduced HexPlane
                          (resolution [8,8,8,5], multires [1,2]), heads perturbed away from their
                          zero init; forward_dynamic_delta outputs and parameter gradients.
  deformation_nodes.npz   the same reference network (state of deformation_small.npz) queried the way the hot path
                          queries it (dynamic_sugar.py:420-431): M static nodes x B timestamps, t = 2*ts - 1, outputs
                          [B,M,*] and parameter gradients -- what the -m gpu test compares the HIP kernels with.
  strain_matrix.npz       strain_tensor_to_matrix (dynamic_sugar.py:29-39), extracted by AST (the
                          enclosing module needs pypose/pytorch3d) and executed.
  schedule_C.npz          C() (threestudio/utils/misc.py:66-101), extracted by AST.
  arap_small.npz          ARAPCoach (custom/threestudio-dreammesh4d/utils/arap_utils.py:17-224), imported with stubs for
                          threestudio.utils.typing / open3d: energy with GIVEN vertex rotations (the dynamic stage's
                          use, system/sugar_4dgen.py:374-385) of a deformed 120-face sphere + autograd gradients.
  zero123_small.npz       extern/ldm_zero123 UNetModel (openaimodel.py:429-842) and AutoencoderKL Encoder
                          (modules/diffusionmodules/model.py) at reduced width, imported with no-op stubs for the
                          unused heavy imports; weights come from a name-seeded recipe (seeded_fill) the test
                          repeats, so only inputs / outputs / key lists are stored.
"""
import ast
import importlib.util
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
C_DIR = os.path.join(REF, "custom", "threestudio-dreammesh4d")
OUT = os.path.dirname(os.path.abspath(__file__))


def extract_function(path, name, extra_globals):
    tree = ast.parse(open(path).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == name:
            node.returns = None
            for a in node.args.args + node.args.kwonlyargs:
                a.annotation = None
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(extra_globals)
            exec(compile(mod, path, "exec"), ns)
            return ns[name]
    raise KeyError(name)


def deformation():
    spec = importlib.util.spec_from_file_location("ref_deformation", os.path.join(C_DIR, "geometry", "deformation.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.manual_seed(1234)
    args = ref.ModelHiddenParams(None)
    args.kplanes_config = dict(args.kplanes_config, resolution=[8, 8, 8, 5])
    args.multires = [1, 2]
    args.no_ds, args.no_dr, args.no_do = False, False, False
    net = ref.DeformationNetwork(args)
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():   # heads are zero-initialised: perturb so the fixture is informative
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_(0.05 * torch.randn(p.shape, generator=g))
    M = 37
    pts = (torch.rand(M, 3, generator=g) * 1.6 - 0.8)
    pts[0] = torch.tensor([1.3, -1.2, 0.0])          # outside the aabb: border padding
    t = torch.rand(M, 1, generator=g) * 2 - 1
    dx, dr, ds, do = net.forward_dynamic_delta(pts, t)
    w = [torch.randn(x.shape, generator=g) for x in (dx, dr, ds, do)]
    loss = sum((a * b).sum() for a, b in zip((dx, dr, ds, do), w))
    loss.backward()
    out = {"pts": pts.numpy(), "t": t.numpy(), "dx": dx.detach().numpy(), "dr": dr.detach().numpy(),
           "ds": ds.detach().numpy(), "do": do.detach().numpy(), "loss": np.float64(loss.item()),
           "resolution": np.array([8, 8, 8, 5]), "multires": np.array([1, 2]), "bounds": np.float64(args.bounds),
           "net_width": np.int64(args.net_width), "defor_depth": np.int64(args.defor_depth)}
    for i, x in enumerate(w):
        out[f"w{i}"] = x.numpy()
    for k, v in net.state_dict().items():
        out["state/" + k] = v.numpy()
    for k, p in net.named_parameters():
        out["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    out["n_params"] = np.int64(sum(p.numel() for p in net.parameters()))
    out["mlp_param_names"] = np.array([n for n, _ in net.named_parameters() if "grid" not in n])
    np.savez_compressed(os.path.join(OUT, "deformation_small.npz"), **out)
    # full-size parameter count of the shipped configuration (35,755,892)
    full = ref.DeformationNetwork(ref.ModelHiddenParams(None))
    print("deformation_small.npz written; full-size params:", sum(p.numel() for p in full.parameters()))


def deformation_nodes():
    spec = importlib.util.spec_from_file_location("ref_deformation", os.path.join(C_DIR, "geometry", "deformation.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    z = np.load(os.path.join(OUT, "deformation_small.npz"))
    args = ref.ModelHiddenParams(None)
    args.kplanes_config = dict(args.kplanes_config, resolution=[8, 8, 8, 5])
    args.multires = [1, 2]
    args.no_ds, args.no_dr, args.no_do = False, False, False
    net = ref.DeformationNetwork(args)
    net.load_state_dict({k[len("state/"):]: torch.tensor(z[k]) for k in z.files if k.startswith("state/")}, strict=True)
    g = torch.Generator().manual_seed(11)
    M, B = 41, 3
    nodes = torch.rand(M, 3, generator=g) * 1.6 - 0.8
    nodes[0] = torch.tensor([1.3, -1.2, 0.0])          # outside the aabb: border padding
    nodes[1] = torch.tensor([-1.0, 1.0, 0.999])        # on / next to the upper border
    ts = torch.tensor([0.0, 0.37, 0.96875])            # t = -1 exactly (lower border of the time axis) and two interior
    pts = nodes.unsqueeze(0).expand(B, M, 3).reshape(-1, 3)
    t = ts.view(B, 1, 1).expand(B, M, 1).reshape(-1, 1) * 2.0 - 1.0
    dx, dr, ds, do = net.forward_dynamic_delta(pts, t)
    w = [torch.randn(x.shape, generator=g) for x in (dx, dr, ds, do)]
    loss = sum((a * b).sum() for a, b in zip((dx, dr, ds, do), w))
    loss.backward()
    out = {"nodes": nodes.numpy(), "ts": ts.numpy(), "dx": dx.detach().view(B, M, 3).numpy(), "dr": dr.detach().view(B, M, 4).numpy(),
           "ds": ds.detach().view(B, M, 6).numpy(), "do": do.detach().view(B, M).numpy(), "loss": np.float64(loss.item())}
    for i, (x, k) in enumerate(zip(w, (3, 4, 6, 1))):
        out[f"w{i}"] = x.view(B, M, k).numpy()
    for k, p in net.named_parameters():
        out["grad/" + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(OUT, "deformation_nodes.npz"), **out)
    print("deformation_nodes.npz written; loss", float(loss))


def strain():
    f = extract_function(os.path.join(C_DIR, "geometry", "dynamic_sugar.py"), "strain_tensor_to_matrix", {"torch": torch})
    g = torch.Generator().manual_seed(3)
    s = torch.randn(5, 11, 6, generator=g)
    np.savez_compressed(os.path.join(OUT, "strain_matrix.npz"), s=s.numpy(), m=f(s).numpy())
    print("strain_matrix.npz written")


def schedule():
    f = extract_function(os.path.join(REF, "threestudio", "utils", "misc.py"), "C",
                         {"math": math, "config_to_primitive": lambda v: list(v), "Any": object})
    cases = [([200, 500.0, 5000.0, 1000], "linear"), ([0, 0.01, 0.0001, 2000], "exp"), ([0.5, 0.1, 300], "linear"),
             ([0, 1.0, 2.0, 100, 3.0, 200, 0.5, 400], "linear")]
    steps = [0, 1, 99, 100, 150, 199, 200, 250, 399, 400, 600, 999, 1000, 1500, 2000, 5000]
    vals = np.array([[f(v, 0, s, interpolation=i) for s in steps] for v, i in cases])
    np.savez_compressed(os.path.join(OUT, "schedule_C.npz"), steps=np.array(steps), values=vals,
                        scalar=np.float64(f(0.37, 0, 10)))
    print("schedule_C.npz written")


def seeded_fill(module, base=1000, scale=0.05):
    """Name-ordered deterministic weights shared by the generator and the test: tensor i (sorted by
    state-dict key) = randn(seed = base + i) * scale, +1 for normalisation gains."""
    sd = module.state_dict()
    with torch.no_grad():
        for i, k in enumerate(sorted(sd.keys())):
            t = sd[k]
            if not t.dtype.is_floating_point:
                continue
            g = torch.Generator().manual_seed(base + i)
            v = torch.randn(t.shape, generator=g) * scale
            if ("norm" in k or k.endswith("in_layers.0.weight") or k.endswith("out_layers.0.weight") or k == "out.0.weight") and k.endswith("weight") and t.dim() == 1:
                v = v + 1.0
            t.copy_(v)
    return sorted(sd.keys())


def zero123_small():
    import types
    for name in ["cv2", "torchvision", "torchvision.transforms", "omegaconf", "omegaconf.listconfig", "pytorch_lightning",
                 "kornia", "clip", "taming", "taming.modules", "taming.modules.vqvae", "taming.modules.vqvae.quantize"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["omegaconf.listconfig"].ListConfig = list
    sys.modules["omegaconf"].ListConfig = list
    sys.path.insert(0, REF)
    from extern.ldm_zero123.modules.diffusionmodules.openaimodel import UNetModel
    from extern.ldm_zero123.modules.diffusionmodules.model import Encoder
    unet = UNetModel(image_size=32, in_channels=8, out_channels=4, model_channels=32, attention_resolutions=[4, 2, 1],
                     num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                     transformer_depth=1, context_dim=48, use_checkpoint=False, legacy=False).eval()
    ukeys = seeded_fill(unet, base=1000)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 16, 16, generator=g)
    t = torch.tensor([7, 613])
    ctx = torch.randn(2, 1, 48, generator=g)
    with torch.no_grad():
        y = unet(x, t, context=ctx)
    enc = Encoder(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, attn_resolutions=[], dropout=0.0,
                  in_channels=3, resolution=64, z_channels=4, double_z=True).eval()
    ekeys = seeded_fill(enc, base=5000)
    img = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        m = enc(img)
    np.savez_compressed(os.path.join(OUT, "zero123_small.npz"), x=x.numpy(), t=t.numpy(), ctx=ctx.numpy(), y=y.numpy(),
                        img=img.numpy(), moments=m.numpy(), unet_keys=np.array(ukeys), enc_keys=np.array(ekeys),
                        unet_params=np.int64(sum(p.numel() for p in unet.parameters())))
    print("zero123_small.npz written; reduced UNet params:", sum(p.numel() for p in unet.parameters()), "| y std", float(y.std()))


def arap():
    import types
    import typing

    ty = types.ModuleType("threestudio.utils.typing")
    for n in dir(typing):
        if not n.startswith("_"):
            setattr(ty, n, getattr(typing, n))

    class _SubMeta(type):
        def __getitem__(cls, k):
            return cls

    class _Sub(metaclass=_SubMeta):
        pass

    ty.Float = ty.Int = ty.Num = ty.Bool = _Sub
    ty.Tensor = torch.Tensor
    sys.modules.update({"threestudio": types.ModuleType("threestudio"), "threestudio.utils": types.ModuleType("threestudio.utils"),
                        "threestudio.utils.typing": ty, "open3d": types.ModuleType("open3d")})
    spec = importlib.util.spec_from_file_location("ref_arap", os.path.join(C_DIR, "utils", "arap_utils.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from dreammesh4d_amd import synthetic as syn

    verts_np, faces_np = syn.uv_sphere(120, radius=0.6)
    verts = torch.tensor(verts_np, dtype=torch.float32)
    coach = ref.ARAPCoach(verts, np.asarray(faces_np), torch.device("cpu"))
    g = torch.Generator().manual_seed(21)
    xyz = (verts + 0.05 * torch.randn(verts.shape, generator=g)).requires_grad_(True)
    q = torch.nn.functional.normalize(torch.cat([0.15 * torch.randn(len(verts), 3, generator=g), torch.ones(len(verts), 1)], -1), dim=-1)
    x, y, z, w = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3).requires_grad_(True)
    E = coach.compute_arap_energy(xyz_prime=xyz, vert_rotations=R)
    E.backward()
    np.savez_compressed(os.path.join(OUT, "arap_small.npz"), verts=verts_np.astype(np.float32), faces=np.asarray(faces_np, np.int64),
                        xyz_prime=xyz.detach().numpy(), rotations=R.detach().numpy(), energy=np.float64(E.item()),
                        g_xyz=xyz.grad.numpy(), g_rot=R.grad.numpy(),
                        energy_rigid=np.float64(coach.compute_arap_energy(verts + 0.3, torch.eye(3)[None].repeat(len(verts), 1, 1)).item()))
    print("arap_small.npz: V", len(verts), "F", len(faces_np), "E", float(E))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    deformation()
    deformation_nodes()
    arap()
    strain()
    schedule()
    zero123_small()
