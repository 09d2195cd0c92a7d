"""Worker of tests/test_dynamic_stage_gpu.py::test_two_rank_rehearsal_*: launched by torch.distributed.run with 2 ranks that SHARE
cuda:0 and exchange over gloo (a functional rehearsal of the N > 1 control flow on a 1-GPU box; never a performance number).
Runs 2 dynamic-stage iterations (frames sharded by rank, one gradient exchange, AdamW) and checks that the replicas hold
bit-identical parameters afterwards; argv[1] = "replicated" | "sharded"."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    mode = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from dreammesh4d_amd import distributed as D, geometry as geo, ops, synthetic as syn, views
    from dreammesh4d_amd.deformation import DeformationNetwork
    from dreammesh4d_amd.dynamic_stage import DynamicStage

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    H = W = 96
    M, L = 200, 16
    sc = syn.mesh_bound_scene(3000, n_nodes=M, k=4, seed=0)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    static = {"q_static": geo.quaternions(verts, faces, T(sc["complex"]), 6), "scales": geo.scaling(T(sc["log_scales"]), syn.THICKNESS),
              "opacities": geo.strengths(T(sc["densities"])), "rgb": geo.points_rgb(T(sc["sh_dc"]))}
    cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)
    r = views.ViewRenderer(graph, topo, H, W, cam.tanfov, method="hybrid")
    torch.manual_seed(0)                                   # identical initial replicas (DDP would broadcast rank 0's)
    net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2), no_ds=False, no_dr=False, no_do=False).to(dev)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if "_deform" in name:
                p.add_(0.01 * torch.randn_like(p))
    g = torch.Generator().manual_seed(1)
    ref_img = torch.rand(L, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(L, H, W, 1, generator=g) > 0.5).float().to(dev)
    stage = DynamicStage(r, net, T(sc["nodes"]), static, torch.linspace(0, 1, L + 2, device=dev)[1:-1], ref_img, ref_mask, cam, guidance=None,
                         frames_per_step=4, random_views_per_frame=1, sharded_optimizer=(mode == "sharded"))
    frames = []
    for _ in range(2):
        out = stage.iteration()
        frames.append(D.shard_frames(L, rank, world, 4, stage.global_step - 1))
        assert torch.isfinite(out["loss"])
    assert stage._sparse_reducer and stage.reducer.nbytes < 4 * stage.reducer.dense_elements       # the structured-sparse message
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    other_frames = [None] * world
    dist.all_gather_object(other_frames, frames)
    if rank == 0:
        assert torch.equal(both[0], both[1]), "replicas diverged: max |diff| %g" % float((both[0] - both[1]).abs().max())
        assert set(other_frames[0][0]).isdisjoint(other_frames[1][0]), "ranks rendered the same frames"
        moved = float((both[0] != 0).float().mean())
        print(f"DP_REHEARSAL_OK mode={mode} message_bytes={stage.reducer.nbytes} dense_bytes={4 * stage.reducer.dense_elements} nonzero_params={moved:.3f}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
