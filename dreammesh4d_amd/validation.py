"""Validation / test sweep on the fast path: every frame of the video from a handful of
azimuths, forward only -- ``SuGaR4DGen.validation_step`` / ``test_step``
(custom/threestudio-dreammesh4d/system/sugar_4dgen.py:481-504,559-587) render the L frames one ``self(batch)`` call
at a time per azimuth (32 frames x 5 azimuths at 512^2 in the shipped config).  Here the sweep is batched across
frames: one deformation query per chunk of timestamps and ONE ``render_views`` call (one launch per kernel) for the
chunk's (frame, azimuth) units, under ``torch.no_grad``.  Image files / mp4 encoding are the caller's (out of scope:
image IO)."""
import torch

from . import synthetic as syn
from .views import render_views


@torch.no_grad()
def sweep(renderer, net, nodes, static, timestamps, azimuths_deg=(0.0, 72.0, 144.0, 216.0, 288.0), elevation_deg=0.0,
          distance=3.8, frames_per_call=4, background=(0.0, 0.0, 0.0), on_chunk=None):
    """Renders every (timestamp, azimuth) unit.  Returns a dict of tensors indexed [frame, azimuth]:
    comp_rgb [L,A,H,W,3] (clamped to [0,1]), comp_normal [L,A,H,W,3], depth [L,A,H,W,1], opacity [L,A,H,W,1] -- or, when
    `on_chunk(frame_indices, chunk_dict)` is given, streams the chunks to it and returns None (a 32 x 5 x 512^2 sweep
    is 1.3 GB of float32 colour + normal).  The evaluation background is black (inverted training background,
    renderer/diff_sugar_rasterizer_temporal.py:96-103)."""
    dev = nodes.device
    H, W = renderer.H, renderer.W
    L, A = int(timestamps.shape[0]), len(azimuths_deg)
    cams = [syn.make_camera(H, W, elev_deg=elevation_deg, azim_deg=float(a), dist=distance) for a in azimuths_deg]
    vm = torch.stack([torch.tensor(c.viewmatrix, device=dev) for c in cams])
    pm = torch.stack([torch.tensor(c.projmatrix, device=dev) for c in cams])
    bg6 = torch.tensor(list(background) + [0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
    keep = None if on_chunk is not None else {k: [] for k in ("comp_rgb", "comp_normal", "depth", "opacity")}
    for f0 in range(0, L, frames_per_call):
        fr = list(range(f0, min(L, f0 + frames_per_call)))
        dx, dr, ds, do = net.node_outputs(nodes, timestamps[fr])
        u = torch.arange(len(fr), device=dev, dtype=torch.int32).repeat_interleave(A)          # unit -> frame of the chunk
        out = render_views(renderer, dx, dr, ds, do, static["q_static"], static["scales"], static["opacities"], static["rgb"],
                           vm.repeat(len(fr), 1, 1), pm.repeat(len(fr), 1, 1), bg6, frame_index=u)
        alpha = out["alpha"].permute(0, 2, 3, 1)
        n = torch.nn.functional.normalize(out["color"][:, 3:].permute(0, 2, 3, 1), dim=-1)
        chunk = {"comp_rgb": out["color"][:, :3].clamp(0, 1).permute(0, 2, 3, 1).reshape(len(fr), A, H, W, 3),
                 "comp_normal": (n * 0.5 * alpha + 0.5).reshape(len(fr), A, H, W, 3),      # …temporal.py:212-216
                 "depth": out["depth"].permute(0, 2, 3, 1).reshape(len(fr), A, H, W, 1),
                 "opacity": alpha.reshape(len(fr), A, H, W, 1)}
        if on_chunk is not None:
            on_chunk(fr, chunk)
        else:
            for k in keep:
                keep[k].append(chunk[k])
    renderer.check()
    return None if keep is None else {k: torch.cat(v) for k, v in keep.items()}
