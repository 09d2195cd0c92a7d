"""Worker of tests/test_dynamic_stage_gpu.py::test_two_rank_rehearsal_*: launched by torch.distributed.run with 2 ranks that SHARE
cuda:0 and exchange over gloo (a functional rehearsal of the N > 1 control flow on a 1-GPU box; never a performance number).
Runs 2 dynamic-stage iterations (frames sharded by rank, one gradient exchange, AdamW) and checks that the replicas hold
bit-identical parameters afterwards; argv[1] = "replicated" | "sharded" | "compare" (3 iterations through BOTH optimisers: the
sharded one must reproduce the replicated parameters)."""
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
    M, L = 200, int(os.environ.get("DM4D_REHEARSAL_FRAMES", "16"))      # (8 ranks: 32 frames = cfg 4's exact partition, 4 frames per rank)
    sc = syn.mesh_bound_scene(3000, n_nodes=M, k=4, seed=0)
    T = lambda a: torch.tensor(a, device=dev)
    graph = ops.DeformGraph(sc["verts"], sc["nbr_idx"], sc["nbr_w"], M, dev)
    topo = ops.MeshTopology(sc["faces"], len(sc["verts"]), 6, dev)
    verts, faces = T(sc["verts"]), T(sc["faces"])
    static = {"q_static": geo.quaternions(verts, faces, T(sc["complex"]), 6), "scales": geo.scaling(T(sc["log_scales"]), syn.THICKNESS),
              "opacities": geo.strengths(T(sc["densities"])), "rgb": geo.points_rgb(T(sc["sh_dc"]))}
    cam = syn.make_camera(H, W, elev_deg=5.0, azim_deg=0.0)
    g = torch.Generator().manual_seed(1)
    ref_img = torch.rand(L, H, W, 3, generator=g).to(dev)
    ref_mask = (torch.rand(L, H, W, 1, generator=g) > 0.5).float().to(dev)

    def build(sharded):
        torch.manual_seed(0)                               # identical initial replicas (DDP would broadcast rank 0's)
        net = DeformationNetwork(resolution=(16, 16, 16, 9), multires=(1, 2), no_ds=False, no_dr=False, no_do=False).to(dev)
        with torch.no_grad():
            for name, p in net.named_parameters():
                if "_deform" in name:
                    p.add_(0.01 * torch.randn_like(p))
        r = views.ViewRenderer(graph, topo, H, W, cam.tanfov, method="hybrid")
        return DynamicStage(r, net, T(sc["nodes"]), static, torch.linspace(0, 1, L + 2, device=dev)[1:-1], ref_img, ref_mask, cam,
                            guidance=None, frames_per_step=4, random_views_per_frame=1, sharded_optimizer=sharded), net

    def run(stage, n):
        frames = []
        for _ in range(n):
            out = stage.iteration()
            frames.append(D.shard_frames(L, rank, world, 4, stage.global_step - 1))
            assert torch.isfinite(out["loss"])
        return frames

    def flat_params(stage, net):
        stage.state_for_checkpoint()                       # (the sharded optimiser's deferred decay of the untouched texels)
        return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()

    if mode == "compare":
        # sharded == replicated: the same 3 iterations (same seeds, same frames, same cameras) through both optimisers
        (st_r, net_r), (st_s, net_s) = build(False), build(True)
        assert st_s.sharded is not None and st_s.sharded.reducer is st_s.reducer            # created in the constructor, never rebuilt
        run(st_r, 3)
        run(st_s, 3)
        a, b = flat_params(st_r, net_r), flat_params(st_s, net_s)
        err = float((a - b).abs().max() / a.abs().max())
        if rank == 0 and os.environ.get("DM4D_REHEARSAL_DEBUG"):
            k = int((a - b).abs().argmax())
            off = 0
            for (name, pr), ps in zip(net_r.named_parameters(), net_s.parameters()):
                if off <= k < off + pr.numel():
                    j = k - off
                    print(f"REHEARSAL_DEBUG max diff in {name}[{j}] of {pr.numel()}: replicated {float(pr.detach().reshape(-1)[j]):.9g} sharded "
                          f"{float(ps.detach().reshape(-1)[j]):.9g}; grad r {None if pr.grad is None else float(pr.grad.reshape(-1)[j]):} "
                          f"s {None if ps.grad is None else float(ps.grad.reshape(-1)[j])}; n elements with |diff| > 1e-6 max|a|: "
                          f"{int(((a - b).abs() > 1e-6 * a.abs().max()).sum())} of {a.numel()}", flush=True)
                off += pr.numel()
        # (the replicated side is torch's FUSED multi-tensor AdamW, whose operation order differs from the single-tensor
        #  formula the sharded step follows; with eps = 1e-15 the update is ~lr * sign(g), so last-bit differences of the
        #  moments show at 1e-6 of the parameter scale)
        # World 2: 4.4e-6.  World 8 (64 renders per iteration): 1.3e-4 on 1.5 % of the elements -- the two optimisers' parameters differ
        # in the last bits after the first step (operation order), and a rasterizer is DISCONTINUOUS in its inputs: a 1e-7 change flips
        # an alpha >= 1/255 or tile-membership decision in a few of the 64 renders, which moves the gradient of the texels behind
        # those Gaussians by a visible fraction (the element of the largest difference had gradients -1.5e-4 / -8.7e-4 in the two
        # runs).  Hence a bound on the bulk and a loose one (a small fraction of one learning-rate step) on the rest.
        big = float(((a - b).abs() > 2e-5 * a.abs().max()).float().mean())
        assert (err < 2e-5) if world <= 2 else (err < 1e-3 and big < 0.01), f"sharded optimiser diverged from the replicated one: max {err}, {big:.4f} of the elements beyond 2e-5"
        both = [torch.empty_like(b) for _ in range(world)]
        dist.all_gather(both, b)
        if rank == 0:
            assert all(torch.equal(both[0], x) for x in both[1:])
            print(f"DP_REHEARSAL_OK mode=compare world={world} max_rel_diff={err:.2e} moment_elems_per_rank={st_s.sharded.exp_avg.numel()} "
                  f"message_elems={st_s.reducer.flat.numel()}", flush=True)
        dist.destroy_process_group()
        return
    stage, net = build(mode == "sharded")
    frames = run(stage, 2)
    assert stage.reducer.nbytes < 4 * stage.reducer.dense_elements       # the structured-sparse message, from the first step
    flat = flat_params(stage, net)
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    other_frames = [None] * world
    dist.all_gather_object(other_frames, frames)
    if rank == 0:
        for x in both[1:]:
            assert torch.equal(both[0], x), "replicas diverged: max |diff| %g" % float((both[0] - x).abs().max())
        for it in range(len(frames)):
            seen = [f for r_ in range(world) for f in other_frames[r_][it]]
            assert len(seen) == len(set(seen)), f"ranks rendered the same frames in iteration {it}: {other_frames}"
            if L == 4 * world:          # cfg 4's partition: rank r owns frames 4r .. 4r + 3 (rotated per iteration): the timeline is covered
                assert sorted(seen) == list(range(L)), seen
        if L == 4 * world:
            assert other_frames[0][0] == [0, 1, 2, 3] and other_frames[world - 1][0] == [L - 4, L - 3, L - 2, L - 1]
        moved = float((both[0] != 0).float().mean())
        print(f"DP_REHEARSAL_OK mode={mode} world={world} message_bytes={stage.reducer.nbytes} dense_bytes={4 * stage.reducer.dense_elements} nonzero_params={moved:.3f}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
