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
  strain_matrix.npz       strain_tensor_to_matrix (dynamic_sugar.py:29-39), extracted by AST (the
                          enclosing module needs pypose/pytorch3d) and executed.
  schedule_C.npz          C() (threestudio/utils/misc.py:66-101), extracted by AST.
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


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (authoring container only)")
    deformation()
    strain()
    schedule()
