"""Stage hand-off artefacts (dreammesh4d_amd/wire_formats.py): PLY round trips in the layouts the reference's tools
produce, and the Lightning checkpoint key scheme (threestudio/utils/misc.py:33-63)."""
import numpy as np
import pytest
import torch

from dreammesh4d_amd import synthetic as syn
from dreammesh4d_amd import wire_formats as wf


def _mesh():
    sc = syn.mesh_bound_scene(300, n_nodes=10, k=4, seed=0)
    v, f = np.asarray(sc["verts"], np.float64), np.asarray(sc["faces"], np.int64)
    c = np.random.default_rng(0).random((len(v), 3))
    return v, f, c


def test_binary_ply_round_trip_with_colours_and_normals(tmp_path):
    v, f, c = _mesh()
    p = tmp_path / "exported_mesh_step10.ply"
    wf.write_ply(p, v, f, colors=c, normals=wf.vertex_normals(v, f))
    head = open(p, "rb").read(400).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0") and "property double x" in head
    assert "property uchar red" in head and "property list uchar uint vertex_indices" in head
    m = wf.read_ply(p)
    assert np.array_equal(m["verts"], v) and np.array_equal(m["faces"], f)
    assert np.abs(m["colors"] - c).max() <= 0.5 / 255 + 1e-12                   # 8-bit colours
    assert np.abs(np.linalg.norm(m["normals"], axis=1) - 1).max() < 1e-12
    # outward on a sphere centred at the origin
    assert (np.einsum("ij,ij->i", m["normals"], v) > 0).mean() > 0.99


def test_ascii_and_big_endian_float_variants(tmp_path):
    v, f, c = _mesh()
    # ascii, float positions, a quad face (fan-triangulated), float colours in [0, 1]
    p = tmp_path / "a.ply"
    with open(p, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment hand written\nelement vertex 4\nproperty float x\nproperty float y\n"
                 "property float z\nproperty float red\nproperty float green\nproperty float blue\nelement face 1\n"
                 "property list uchar int vertex_index\nend_header\n"
                 "0 0 0 1 0 0\n1 0 0 0 1 0\n1 1 0 0 0 1\n0 1 0 0.5 0.5 0.5\n4 0 1 2 3\n")
    m = wf.read_ply(p)
    assert m["verts"].shape == (4, 3) and m["faces"].tolist() == [[0, 1, 2], [0, 2, 3]]
    assert np.allclose(m["colors"][3], 0.5) and m["normals"] is None
    # binary big endian with float32 positions and int faces
    p2 = tmp_path / "b.ply"
    with open(p2, "wb") as fh:
        fh.write(f"ply\nformat binary_big_endian 1.0\nelement vertex {len(v)}\nproperty float x\nproperty float y\nproperty float z\n"
                 f"element face {len(f)}\nproperty list uchar int vertex_indices\nend_header\n".encode())
        fh.write(v.astype(">f4").tobytes())
        for t in f:
            fh.write(bytes([3]) + np.asarray(t, ">i4").tobytes())
    m2 = wf.read_ply(p2)
    assert np.array_equal(m2["verts"], v.astype(np.float32).astype(np.float64)) and np.array_equal(m2["faces"], f)
    assert m2["colors"] is None
    with pytest.raises(ValueError):
        (tmp_path / "c.ply").write_text("not a ply\n")
        wf.read_ply(tmp_path / "c.ply")


def test_checkpoint_key_scheme_round_trip(tmp_path):
    """``geometry.<name>`` keys, epoch / global_step, module_name / ignore_modules selection."""
    torch.manual_seed(0)
    geo = torch.nn.Module()
    geo._points = torch.nn.Parameter(torch.randn(5, 3))
    geo._deformation = torch.nn.Linear(4, 2)
    plane = torch.randn(1, 8, 3, 5).contiguous(memory_format=torch.channels_last)
    geo.plane = torch.nn.Parameter(plane)
    bg = torch.nn.Linear(3, 3)
    p = tmp_path / "last.ckpt"
    wf.save_checkpoint(p, {"geometry": geo, "background": bg}, epoch=2, global_step=1500)
    sd, epoch, step = wf.load_module_weights(p, module_name="geometry")
    assert (epoch, step) == (2, 1500)
    assert set(sd) == {"_points", "_deformation.weight", "_deformation.bias", "plane"}
    assert sd["plane"].is_contiguous() and torch.equal(sd["plane"], plane)      # stored as a plain [1,C,H,W] tensor
    rest, _, _ = wf.load_module_weights(p, ignore_modules=["geometry"])
    assert set(rest) == {"background.weight", "background.bias"}
    with pytest.raises(ValueError):
        wf.load_module_weights(p, module_name="geometry", ignore_modules=["background"])
    geo2 = torch.nn.Module()
    geo2._points = torch.nn.Parameter(torch.zeros(5, 3))
    geo2._deformation = torch.nn.Linear(4, 2)
    geo2.plane = torch.nn.Parameter(torch.zeros(1, 8, 3, 5).contiguous(memory_format=torch.channels_last))
    missing, unexpected, _, _ = wf.load_geometry(geo2, p, strict=True)
    assert not missing and not unexpected
    assert torch.equal(geo2._points, geo._points) and torch.equal(geo2.plane, plane)
    assert geo2.plane.is_contiguous(memory_format=torch.channels_last)          # the parameter keeps its own storage
