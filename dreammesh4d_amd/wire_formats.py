"""Stage hand-off artefacts either side of the hot path.

* The refined surface mesh the static stage exports and the dynamic stage binds to
  (``surface_mesh_to_bind_path``, README.md:88): a PLY with per-vertex colours written by
  ``BaseSuGaRSystem.export_mesh`` (custom/threestudio-dreammesh4d/system/base.py:49-69, through
  ``o3d.io.write_triangle_mesh(..., write_vertex_colors=True, write_vertex_normals=True)``) and read back by
  ``SuGaRModel`` (geometry/sugar.py:174-212: vertices, triangles, vertex colours).  open3d is not installed here, so
  the reader follows the PLY format itself (ascii / binary_little_endian / binary_big_endian, any scalar property
  types, list-typed faces) and the writer emits the layout open3d uses (double positions and normals, uchar colours,
  ``list uchar uint vertex_indices``); byte-for-byte agreement with open3d's files is unpinned.
* The Lightning checkpoint of a stage (``system.weights``): ``{"state_dict", "epoch", "global_step"}`` with the
  geometry's entries under ``geometry.`` -- ``load_module_weights`` of threestudio/utils/misc.py:33-63.
  ``DynamicSuGaR`` keeps the reference's parameter names, so the geometry entries load by name.
"""
import re
import struct

import numpy as np

_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def read_ply(path):
    """-> dict(verts [V,3] float64, faces [F,3] int64, colors [V,3] float64 in [0,1] or None, normals [V,3] or None)."""
    with open(path, "rb") as fh:
        data = fh.read()
    end = data.index(b"end_header")
    end = data.index(b"\n", end) + 1
    header = data[:end].decode("ascii", "replace").splitlines()
    if not header or header[0].strip() != "ply":
        raise ValueError(f"{path}: not a PLY file")
    fmt, elements = None, []
    for line in header[1:]:
        tok = line.split()
        if not tok or tok[0] == "comment" or tok[0] == "obj_info":
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
        elif tok[0] == "property":
            if tok[1] == "list":
                elements[-1]["props"].append(("list", tok[2], tok[3], tok[4]))
            else:
                elements[-1]["props"].append(("scalar", tok[1], tok[2]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"{path}: unsupported PLY format {fmt!r}")
    out = {}
    if fmt == "ascii":
        tokens = iter(data[end:].split())
        for el in elements:
            rows = []
            for _ in range(el["count"]):
                row = {}
                for p in el["props"]:
                    if p[0] == "scalar":
                        row[p[2]] = float(next(tokens))
                    else:
                        n = int(float(next(tokens)))
                        row[p[3]] = [int(float(next(tokens))) for _ in range(n)]
                rows.append(row)
            out[el["name"]] = rows
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        pos = end
        for el in elements:
            if all(p[0] == "scalar" for p in el["props"]):
                dt = np.dtype([(p[2], bo + _PLY_TYPES[p[1]]) for p in el["props"]])
                arr = np.frombuffer(data, dt, el["count"], pos)
                pos += dt.itemsize * el["count"]
                out[el["name"]] = arr
            else:
                rows = []
                for _ in range(el["count"]):
                    row = {}
                    for p in el["props"]:
                        if p[0] == "scalar":
                            t = np.dtype(bo + _PLY_TYPES[p[1]])
                            row[p[2]] = np.frombuffer(data, t, 1, pos)[0]
                            pos += t.itemsize
                        else:
                            tc, ti = np.dtype(bo + _PLY_TYPES[p[1]]), np.dtype(bo + _PLY_TYPES[p[2]])
                            n = int(np.frombuffer(data, tc, 1, pos)[0])
                            pos += tc.itemsize
                            row[p[3]] = np.frombuffer(data, ti, n, pos).tolist()
                            pos += ti.itemsize * n
                    rows.append(row)
                out[el["name"]] = rows

    def column(el, name):
        if isinstance(el, np.ndarray):
            return el[name].astype(np.float64) if name in el.dtype.names else None
        return np.asarray([r[name] for r in el], np.float64) if el and name in el[0] else None

    v = out.get("vertex")
    if v is None:
        raise ValueError(f"{path}: no vertex element")
    xyz = [column(v, k) for k in "xyz"]
    if any(c is None for c in xyz):
        raise ValueError(f"{path}: vertices lack x / y / z")
    res = {"verts": np.stack(xyz, 1), "colors": None, "normals": None}
    rgb = [column(v, k) for k in ("red", "green", "blue")]
    if all(c is not None for c in rgb):
        integer = isinstance(v, np.ndarray) and v.dtype["red"].kind in "ui"
        if not isinstance(v, np.ndarray):
            decl = {p[2]: p[1] for el in elements if el["name"] == "vertex" for p in el["props"] if p[0] == "scalar"}
            integer = _PLY_TYPES[decl["red"]][0] in "ui"
        res["colors"] = np.stack(rgb, 1) / (255.0 if integer else 1.0)       # open3d: uchar -> [0, 1]
    nrm = [column(v, k) for k in ("nx", "ny", "nz")]
    if all(c is not None for c in nrm):
        res["normals"] = np.stack(nrm, 1)
    faces = []
    for r in out.get("face", []):
        idx = r.get("vertex_indices", r.get("vertex_index"))
        if idx is None:
            continue
        for k in range(1, len(idx) - 1):                                     # fan-triangulate polygons, as open3d does
            faces.append((idx[0], idx[k], idx[k + 1]))
    res["faces"] = np.asarray(faces, np.int64).reshape(-1, 3)
    return res


def write_ply(path, verts, faces, colors=None, normals=None):
    """Binary little-endian PLY in the layout open3d's writer uses: double x y z [nx ny nz] [uchar red green blue],
    faces ``list uchar uint vertex_indices``.  colors in [0, 1]."""
    v = np.asarray(verts, np.float64).reshape(-1, 3)
    f = np.asarray(faces, np.int64).reshape(-1, 3)
    fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")]
    head = ["ply", "format binary_little_endian 1.0", "comment Created by dreammesh4d_amd (open3d layout)",
            f"element vertex {len(v)}", "property double x", "property double y", "property double z"]
    if normals is not None:
        head += ["property double nx", "property double ny", "property double nz"]
        fields += [("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8")]
    if colors is not None:
        head += ["property uchar red", "property uchar green", "property uchar blue"]
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    head += [f"element face {len(f)}", "property list uchar uint vertex_indices", "end_header"]
    arr = np.zeros(len(v), np.dtype(fields))
    arr["x"], arr["y"], arr["z"] = v[:, 0], v[:, 1], v[:, 2]
    if normals is not None:
        n = np.asarray(normals, np.float64).reshape(-1, 3)
        arr["nx"], arr["ny"], arr["nz"] = n[:, 0], n[:, 1], n[:, 2]
    if colors is not None:
        c = np.clip(np.asarray(colors, np.float64).reshape(-1, 3), 0.0, 1.0)
        c8 = np.minimum(255, np.floor(c * 255.0 + 0.5 - 1e-12)).astype(np.uint8)     # round to nearest
        arr["red"], arr["green"], arr["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    farr = np.zeros(len(f), np.dtype([("n", "u1"), ("a", "<u4"), ("b", "<u4"), ("c", "<u4")]))
    farr["n"], farr["a"], farr["b"], farr["c"] = 3, f[:, 0], f[:, 1], f[:, 2]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        fh.write(arr.tobytes())
        fh.write(farr.tobytes())


def vertex_normals(verts, faces):
    """Area-weighted vertex normals (what ``mesh.compute_vertex_normals()`` writes, system/base.py:60)."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    fn = np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]])
    n = np.zeros_like(v)
    for k in range(3):
        np.add.at(n, f[:, k], fn)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.where(ln > 0, ln, 1.0)


def export_mesh(path, geometry, vertex_colors):
    """``BaseSuGaRSystem.export_mesh`` (system/base.py:49-69) for a geometry with ``get_xyz_verts`` / ``get_faces``."""
    v = geometry.get_xyz_verts.detach().cpu().numpy()
    f = geometry.get_faces.detach().cpu().numpy()
    c = vertex_colors.detach().cpu().numpy() if hasattr(vertex_colors, "detach") else np.asarray(vertex_colors)
    write_ply(path, v, f, colors=c, normals=vertex_normals(v, f))


def load_module_weights(path, module_name=None, ignore_modules=None, map_location="cpu"):
    """threestudio/utils/misc.py:33-63: the entries of a Lightning checkpoint's ``state_dict`` under ``module_name.``
    (prefix stripped), or everything except ``ignore_modules`` -> (state_dict, epoch, global_step)."""
    import torch

    if module_name is not None and ignore_modules is not None:
        raise ValueError("module_name and ignore_modules cannot be both set")
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt["state_dict"]
    if ignore_modules is not None:
        sd = {k: v for k, v in sd.items() if not any(k.startswith(m + ".") for m in ignore_modules)}
    if module_name is not None:
        pat = re.compile(rf"^{re.escape(module_name)}\.(.*)$")
        sd = {pat.match(k).group(1): v for k, v in sd.items() if pat.match(k)}
    return sd, ckpt["epoch"], ckpt["global_step"]


def save_checkpoint(path, modules, epoch=0, global_step=0, optimizer_states=None):
    """The inverse: ``{"state_dict": {"<name>.<key>": tensor}, "epoch", "global_step"}`` for ``modules`` =
    {"geometry": nn.Module, ...}, so the reference's ``system.weights`` can load what this package trained.
    ``optimizer_states`` (optional): a list like Lightning's checkpoint key of that name -- here the stages'
    ``optimizer_state_dict()`` (the state of the optimiser that actually steps), restored by ``load_optimizer_states``."""
    import torch

    sd = {}
    for name, m in modules.items():
        for k, v in m.state_dict().items():
            sd[f"{name}.{k}"] = v.detach().cpu().contiguous()      # plain contiguous tensors (the planes are channels_last here)
    ckpt = {"state_dict": sd, "epoch": int(epoch), "global_step": int(global_step)}
    if optimizer_states is not None:
        to_cpu = lambda o: o.detach().cpu() if torch.is_tensor(o) else ({k: to_cpu(v) for k, v in o.items()} if isinstance(o, dict) else
                                                                      ([to_cpu(v) for v in o] if isinstance(o, (list, tuple)) else o))
        ckpt["optimizer_states"] = [to_cpu(o) for o in optimizer_states]
    torch.save(ckpt, path)


def load_optimizer_states(path, map_location="cpu"):
    """The ``optimizer_states`` list of a checkpoint written by ``save_checkpoint`` ([] when it has none)."""
    import torch

    return list(torch.load(path, map_location=map_location, weights_only=False).get("optimizer_states", []))


def load_geometry(geometry, path, strict=False):
    """Load the ``geometry.*`` entries of a checkpoint into a ``sugar.DynamicSuGaR`` (names follow the reference:
    geometry/sugar.py:108,188-233,314-325; deformation.py module paths).  Returns (missing, unexpected, epoch, step)."""
    sd, epoch, step = load_module_weights(path, module_name="geometry")
    res = geometry.load_state_dict(sd, strict=strict)
    return list(res.missing_keys), list(res.unexpected_keys), epoch, step
