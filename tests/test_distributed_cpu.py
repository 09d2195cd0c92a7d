"""world_size-2 gloo tests (CPU) of the N > 1 path: frame sharding and the single gradient all-reduce."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dreammesh4d_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.ones(5))            # trainable but never used (like the reference's timenet)
    params = list(net.parameters()) + [unused]
    red = D.GradAllReducer(params)
    x = torch.full((2, 4), float(rank + 1))
    net(x).sum().backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    red()
    out[rank] = {"grads": [p.grad.clone() for p in params], "local": local, "nbytes": red.nbytes,
                 "frames": D.shard_frames(32, rank, world, 4, iteration=1)}
    dist.destroy_process_group()


def test_grad_allreduce_and_frame_sharding_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for ga, gb, la, lb in zip(a["grads"], b["grads"], a["local"], b["local"]):
        assert torch.equal(ga, gb)                                        # identical after the exchange
        la = torch.zeros_like(ga) if la is None else la
        lb = torch.zeros_like(ga) if lb is None else lb
        assert torch.allclose(ga, 0.5 * (la + lb))                        # mean over ranks; missing grads count as zeros
    assert not a["grads"][-1].any()                                       # the unused parameter got a zero gradient
    assert a["nbytes"] == sum(g.numel() for g in a["grads"]) * 4
    assert set(a["frames"]).isdisjoint(b["frames"]) and len(a["frames"]) == 4


def _worker_sparse(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    grid = torch.nn.Parameter(torch.zeros(1, 4, 5, 6))           # a "plane": gradient only at the touched elements
    dense = torch.nn.Parameter(torch.zeros(7))
    idx = torch.tensor([3, 17, 18, 64, 119])
    g = torch.zeros(grid.numel())
    g[idx] = torch.arange(1.0, 6.0) * (rank + 1)
    grid.grad = g.view_as(grid).clone()
    dense.grad = torch.full((7,), float(rank))
    red = D.GradAllReducer([grid, dense], touched={grid: idx})
    red()
    out[rank] = {"grid": grid.grad.clone(), "dense": dense.grad.clone(), "nbytes": red.nbytes}
    dist.destroy_process_group()


def test_structured_sparse_allreduce_world2():
    """Only the touched elements of a grid travel; the result equals the dense mean."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sparse, args=(2, _free_port(), out), nprocs=2, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["grid"], b["grid"]) and torch.equal(a["dense"], b["dense"])
    want = torch.zeros(120)
    want[torch.tensor([3, 17, 18, 64, 119])] = torch.arange(1.0, 6.0) * 1.5
    assert torch.equal(a["grid"].view(-1), want)
    assert torch.equal(a["dense"], torch.full((7,), 0.5))
    assert a["nbytes"] == (5 + 7) * 4


def test_shard_frames_covers_the_timeline():
    seen = set()
    for r in range(8):
        f = D.shard_frames(32, r, 8, 4, iteration=0)
        assert len(f) == 4 and seen.isdisjoint(f)
        seen.update(f)
    assert seen == set(range(32))
    assert D.shard_frames(32, 0, 1, 4, iteration=3) == [12, 13, 14, 15]
    assert D.world() == 1 and D.rank() == 0


def _worker_sharded(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def build():
        torch.manual_seed(0)
        grid = torch.nn.Parameter(torch.randn(1, 4, 5, 6))          # a "plane": gradient only at the touched elements
        mlp = torch.nn.Linear(6, 3)
        unused = torch.nn.Parameter(torch.ones(5))
        return grid, mlp, unused

    idx = torch.tensor([3, 17, 18, 64, 119])
    results = {}
    for mode in ("replicated", "sharded"):
        grid, mlp, unused = build()
        groups = [{"params": list(mlp.parameters()) + [unused], "lr": 3.2e-4, "name": "deformation"}, {"params": [grid], "lr": 3.2e-3, "name": "grid"}]
        params = list(mlp.parameters()) + [unused, grid]
        red = D.GradAllReducer(params, touched={grid: idx})
        opt = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, foreach=False) if mode == "replicated" else \
            D.ShardedAdamW(groups, red, betas=(0.9, 0.99), eps=1e-15)
        for it in range(4):
            opt.zero_grad(set_to_none=True)
            x = torch.full((2, 6), float(rank + 1 + it))
            loss = mlp(x).pow(2).sum() + (grid.view(-1)[idx] * float(rank + 2)).pow(2).sum()      # unused gets no gradient
            loss.backward()
            for g in (opt.param_groups):
                g["lr"] = g["lr"] * 0.9                                 # a schedule, as update_learning_rate applies
            overflow = it == 1                                          # a forward that overflowed its workspaces: the step is skipped
            if mode == "replicated":
                red()
                if not overflow:
                    opt.step()
            else:
                opt.step(found_inf=torch.tensor(1.0 if overflow else 0.0))     # masked on the device; the collectives still run
        if mode == "sharded":
            opt.materialize()        # the deferred weight decay of the elements outside the message
        results[mode] = [p.detach().clone() for p in params]
        if mode == "sharded":
            results["state_elems"] = opt.exp_avg.numel()
            results["message"] = red.flat.numel()
    out[rank] = results
    dist.destroy_process_group()


def test_sharded_adamw_matches_the_replicated_step_world2():
    """reduce-scatter -> AdamW on 1/world of the message -> all-gather == all-reduce -> replicated torch AdamW: identical
    parameters on both ranks after 4 iterations with changing learning rates -- one of them skipped on an overflow flag
    (found_inf: parameters, moments and the bias-correction step count must not move) --, incl. a parameter that never
    gets a gradient and the untouched elements of a structured-sparse plane (weight decay only, applied lazily)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_sharded, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for pa, pb in zip(a["sharded"], b["sharded"]):
        assert torch.equal(pa, pb)                                            # replicas stay identical
    for ps, pr in zip(a["sharded"], a["replicated"]):
        assert torch.allclose(ps, pr, rtol=1e-6, atol=1e-7), (ps - pr).abs().max()
    assert not torch.equal(a["replicated"][-1], torch.zeros_like(a["replicated"][-1]))
    assert a["state_elems"] == (a["message"] + 1) // 2                        # each rank holds half of the moments


def _worker_sharded_mixed(rank, world, port, out):
    """As _worker_sharded, with the reference's EFFECTIVE mix of hyperparameters: the group training_setup created runs betas
    (0.9, 0.999) without decay (distributed.REFERENCE_GEOMETRY_GROUP), the appended group AdamW's (0.9, 0.99) / 0.01."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = torch.tensor([3, 17, 18, 64, 119])
    results = {}
    for mode in ("replicated", "sharded"):
        torch.manual_seed(0)
        grid = torch.nn.Parameter(torch.randn(1, 4, 5, 6))
        mlp = torch.nn.Linear(6, 3)
        groups = [{"params": list(mlp.parameters()), "lr": 3.2e-4, "name": "deformation", **D.REFERENCE_GEOMETRY_GROUP},
                  {"params": [grid], "lr": 3.2e-3, "name": "appended", "weight_decay": 0.05, "eps": 1e-8}]
        params = list(mlp.parameters()) + [grid]
        red = D.GradAllReducer(params, touched={grid: idx})
        opt = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, foreach=False) if mode == "replicated" else \
            D.ShardedAdamW(groups, red, betas=(0.9, 0.99), eps=1e-15)
        for it in range(5):
            opt.zero_grad(set_to_none=True)
            x = torch.full((2, 6), float(rank + 1 + it))
            (mlp(x).pow(2).sum() + (grid.view(-1)[idx] * float(rank + 2)).pow(2).sum()).backward()
            if mode == "replicated":
                red()
            opt.step()
        if mode == "sharded":
            assert opt.param_groups[0]["betas"] == (0.9, 0.999) and opt.param_groups[0]["weight_decay"] == 0.0
            opt.materialize()
        results[mode] = [p.detach().clone() for p in params]
    out[rank] = results
    dist.destroy_process_group()


def test_sharded_adamw_per_group_hyperparameters_world2():
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_sharded_mixed, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    for pa, pb in zip(a["sharded"], b["sharded"]):
        assert torch.equal(pa, pb)
    for ps, pr in zip(a["sharded"], a["replicated"]):
        assert torch.allclose(ps, pr, rtol=1e-6, atol=1e-7), (ps - pr).abs().max()


def test_parameter_without_gradient_is_skipped_like_torch_skips_it():
    """One process: a parameter whose .grad is None takes NO step in torch.optim (no weight decay, no moment decay, its step counter
    stays) -- round 4's message-space step treated it as a zero gradient.  Three steps, the second without a gradient on `b`."""
    runs = {}
    for mode in ("torch", "message"):
        torch.manual_seed(1)
        a, b = torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(5))
        groups = [{"params": [a], "lr": 1e-2, **D.REFERENCE_GEOMETRY_GROUP}, {"params": [b], "lr": 2e-2, "weight_decay": 0.1}]
        opt = torch.optim.AdamW(groups, lr=0.0, betas=(0.9, 0.99), eps=1e-15, foreach=False) if mode == "torch" else \
            D.ShardedAdamW(groups, D.GradAllReducer([a, b]), betas=(0.9, 0.99), eps=1e-15)
        g = torch.Generator().manual_seed(5)
        for it in range(3):
            a.grad = torch.randn(7, generator=g)
            gb = torch.randn(5, generator=g)
            b.grad = None if it == 1 else gb
            opt.step()
        if mode == "message":
            assert opt.step_t.tolist() == [3.0, 2.0]
            opt.materialize()
        runs[mode] = (a.detach().clone(), b.detach().clone())
    for x, y in zip(runs["torch"], runs["message"]):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7), (x - y).abs().max()


def test_merge_optimizer_keeps_the_reference_effective_hyperparameters():
    """geometry/sugar.py:382,406-416: Adam(l, lr=0, eps=1e-15) fills the group dicts in place, AdamW(l, betas=[0.9, 0.99], eps=1e-15) only
    fills what is missing -- the geometry groups run betas (0.9, 0.999) WITHOUT weight decay, appended groups (0.9, 0.99) / 0.01.
    (Round 4 built fresh dicts and so trained the geometry with beta2 = 0.99 and weight_decay = 0.01.)"""
    import numpy as np

    from dreammesh4d_amd import sugar, synthetic as syn

    sc = syn.mesh_bound_scene(60, n_nodes=8, k=4, seed=0)
    g = sugar.SuGaR(sc["verts"], sc["faces"], vertex_colors=np.random.default_rng(0).random((len(sc["verts"]), 3)), device=torch.device("cpu"))
    extra = torch.nn.Linear(3, 3)
    opt = g.merge_optimizer(torch.optim.SGD([{"params": list(extra.parameters()), "lr": 0.01}], lr=0.0))
    named = [q for q in opt.param_groups if "name" in q]
    assert {q["name"] for q in named} >= {"points", "f_dc", "all_densities", "scales", "quaternions"}
    for q in named:
        assert tuple(q["betas"]) == (0.9, 0.999) and q["weight_decay"] == 0 and q["eps"] == 1e-15, q["name"]
    tail = [q for q in opt.param_groups if "name" not in q]
    assert len(tail) == 1 and tuple(tail[0]["betas"]) == (0.9, 0.99) and tail[0]["weight_decay"] == 0.01 and tail[0]["eps"] == 1e-15
    # the in-place behaviour itself, as torch has it (what the reference relies on without saying so)
    p = torch.nn.Parameter(torch.zeros(2))
    l = [{"params": [p], "lr": 0.1, "name": "x"}]
    torch.optim.Adam(l, lr=0.0, eps=1e-15)
    o = torch.optim.AdamW(l, lr=0.0, betas=[0.9, 0.99], eps=1e-15)
    assert tuple(o.param_groups[0]["betas"]) == (0.9, 0.999) and o.param_groups[0]["weight_decay"] == 0
    assert D.REFERENCE_GEOMETRY_GROUP == {"betas": (0.9, 0.999), "eps": 1e-15, "weight_decay": 0.0}


def _worker_resume(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = torch.tensor([3, 17, 18, 64, 119])

    def build():
        torch.manual_seed(0)
        grid = torch.nn.Parameter(torch.randn(1, 4, 5, 6))
        mlp = torch.nn.Linear(6, 3)
        unused = torch.nn.Parameter(torch.ones(5))
        groups = [{"params": list(mlp.parameters()) + [unused], "lr": 3.2e-4, "name": "deformation"}, {"params": [grid], "lr": 3.2e-3, "name": "grid"}]
        params = list(mlp.parameters()) + [unused, grid]
        red = D.GradAllReducer(params, touched={grid: idx})
        return grid, mlp, params, D.ShardedAdamW(groups, red, betas=(0.9, 0.99), eps=1e-15), torch.Generator().manual_seed(100 + rank)

    def iterate(grid, mlp, opt, gen, n):
        for _ in range(n):
            opt.zero_grad(set_to_none=True)
            x = torch.rand(2, 6, generator=gen) + float(rank)           # this rank's own batch: drawn from ITS generator
            (mlp(x).pow(2).sum() + (grid.view(-1)[idx] * float(rank + 2)).pow(2).sum()).backward()
            opt.step()

    # the uninterrupted run: 3 iterations, checkpoint, 2 more
    grid, mlp, params, opt, gen = build()
    iterate(grid, mlp, opt, gen, 3)
    saved_params = [p.detach().clone() for p in params]
    sd = D.stage_optimizer_state(opt, None, 3, gen)                      # collective; the "file" is what RANK 0 holds
    shard0 = [opt.state_dict()]
    box = [sd]
    dist.broadcast_object_list(box, src=0)
    dist.broadcast_object_list(shard0, src=0)
    sd0 = box[0]
    iterate(grid, mlp, opt, gen, 2)
    want = [p.detach().clone() for p in params]
    # the resumed run: a fresh stage, the checkpoint's parameters, RANK 0's optimiser entry on every rank
    grid2, mlp2, params2, opt2, gen2 = build()
    with torch.no_grad():
        for p, v in zip(params2, saved_params):
            p.copy_(v)
    res = {"full_moments": int(sd0["state"]["exp_avg"].numel()), "message": int(opt2.reducer.flat.numel()), "layout": sd0["state"]["layout"]}
    try:
        opt2.load_state_dict(shard0[0])                                 # rank 0's SHARD: fine on rank 0, refused on rank 1
        res["shard_of_rank0"] = "loaded"
    except ValueError:
        res["shard_of_rank0"] = "refused"
    step = D.load_stage_optimizer_state(sd0, opt2, None, gen2)
    res["step"] = step
    res["gen_is_mine"] = bool(torch.equal(gen2.get_state(), sd0["rng_states"][rank]))
    res["gens_differ"] = not torch.equal(sd0["rng_states"][0], sd0["rng_states"][1])
    iterate(grid2, mlp2, opt2, gen2, 2)
    res["equal"] = all(torch.equal(a.detach(), b) for a, b in zip(params2, want))
    res["moved"] = any(not torch.equal(a, b) for a, b in zip(saved_params, want))
    try:            # a single-process checkpoint's sampler state must not make every rank draw the same batches
        D.load_stage_optimizer_state({"kind": "dm4d.ShardedAdamW", "state": sd0["state"], "rng_state": gen.get_state()}, opt2, None, gen2)
        res["single_rng"] = "loaded"
    except ValueError:
        res["single_rng"] = "refused"
    out[rank] = res
    dist.destroy_process_group()


def test_sharded_adamw_checkpoint_resumes_every_rank_world2():
    """a checkpoint written by rank 0 must resume EVERY rank -- the optimiser entry of a stage holds the moments of the
    whole message (all-gathered: each rank loads its own slice) and every rank's sampler state; the resumed 2-rank run is bit-identical to
    the uninterrupted one; another rank's shard is refused instead of silently applied to the wrong slice."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_resume, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert o["layout"] == "full" and o["full_moments"] == o["message"]
        assert o["equal"] and o["moved"] and o["step"] == 3
        assert o["gen_is_mine"] and o["gens_differ"]
        assert o["single_rng"] == "refused"
    assert out[0]["shard_of_rank0"] == "loaded" and out[1]["shard_of_rank0"] == "refused"
