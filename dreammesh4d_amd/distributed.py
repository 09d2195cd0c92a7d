"""Multi-GPU layer of the hot path: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference relies on implicit Lightning DDP (launch.py:114,228-235: every rank seeds differently,
samples its own 4 frames x views, parameters replicated) and would in fact fail on its
trainable-but-unused parameters.  Here the two pieces are explicit:

* `shard_frames`   -- (frame, view) units are independent: rank r takes frames {4r .. 4r+3} of the
                      timeline each iteration; no data-path collective.
* `GradAllReducer` -- the ONE exchange step: a single all-reduce (sum, x 1/world) of one flat
                      float32 buffer holding every trainable gradient; parameters that received no
                      gradient on a rank contribute zeros (DDP would raise on them).  At the shipped
                      dynamic-stage configuration the buffer is 143 MB (35.76 M floats), i.e. one large
                      message -- what a point-to-point xGMI ring wants -- instead of DDP's 25 MB buckets.
"""
from typing import Iterable, List

import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_frames(n_frames: int, rank_: int, world_: int, frames_per_rank: int = 4, iteration: int = 0) -> List[int]:
    """Frames rendered by `rank_` in `iteration`: consecutive blocks of `frames_per_rank`, rotated by
    the iteration so every rank visits the whole timeline (cfg 4: 32 frames / 8 ranks = 4 each)."""
    start = (rank_ + iteration * world_) * frames_per_rank
    return [(start + i) % n_frames for i in range(frames_per_rank)]


class GradAllReducer:
    """One all-reduce per step over one flat float32 buffer.

    `touched` (optional): {parameter: LongTensor of element indices in STORAGE order (`storage_flat`)}.  For those parameters only the listed
    elements are exchanged; every rank must pass the SAME index sets and the gradient must be zero elsewhere on
    every rank.  That is the case for the HexPlane grids: the graph nodes are static and identical on all ranks, so
    the spatial planes only ever receive gradient at the texels the nodes touch (`touched_from_plan`) -- 1.0 M of
    their 33.4 M elements at the shipped configuration -- which shrinks the message from 143 MB to ~13 MB (spatial
    texels + the dense time planes + the MLP): the exchange stops being the per-link-bound term of a step
    (one xGMI ring moves 143 MB in ~1 ms; a step is ~2 ms).  The result is identical to the
    dense all-reduce (mean of zeros is zero)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], touched=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, torch.float32
        touched = touched or {}
        self.index = [touched.get(p) for p in self.params]
        self.offsets, n = [], 0
        for p, ix in zip(self.params, self.index):
            self.offsets.append(n)
            n += p.numel() if ix is None else int(ix.numel())
            if ix is not None and (ix.dtype != torch.long or ix.device != p.device):
                raise ValueError("touched indices must be int64 tensors on the parameter's device")
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self.dense_elements = sum(p.numel() for p in self.params)

    @property
    def nbytes(self):
        return self.flat.numel() * 4

    def __call__(self):
        """all-reduce (mean over ranks) of every gradient, in place.  No-op for a single process."""
        w = world()
        if w == 1:
            return
        self.pack()
        if self.flat.is_cuda and dist.get_backend() == "gloo":
            host = self.flat.cpu()      # functional rehearsals over gloo: stage through the host
            dist.all_reduce(host)
            self.flat.copy_(host)
        else:
            dist.all_reduce(self.flat)  # one message; RCCL picks ring / direct on the xGMI mesh
        self.unpack(1.0 / w)

    # On a HIP device packing and unpacking are ONE launch each (csrc/gradpack.hip, C ABI dm4d_grad_pack / _unpack)
    # instead of ~80 copy / index kernels; on CPU tensors (the gloo tests) the same thing with torch ops.
    def _segments(self, for_unpack):
        import ctypes as C

        from . import _lib

        if len(self.params) > _lib.MAX_GRAD_SEGMENTS:
            raise ValueError(f"more than {_lib.MAX_GRAD_SEGMENTS} gradient tensors")
        seg = _lib.GradSegments()
        seg.n_segments = len(self.params)
        for k, (p, o, ix) in enumerate(zip(self.params, self.offsets, self.index)):
            if p.grad is None and for_unpack:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or g.stride() != p.stride()):
                raise ValueError("gradients must be float32 with the parameter's strides")
            storage_flat(p)                                     # dense (contiguous or channels_last), else raises
            seg.grad[k] = None if g is None else g.data_ptr()
            seg.index[k] = None if ix is None else ix.data_ptr()
            seg.count[k] = p.numel() if ix is None else ix.numel()
            seg.offset[k] = o
        return seg

    def pack(self):
        flat = self.flat
        if flat.is_cuda:
            import ctypes as C

            from . import _lib

            seg = self._segments(False)
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().dm4d_grad_pack(C.byref(seg), flat.data_ptr(),
                                                     torch.cuda.current_stream(flat.device).cuda_stream), "dm4d_grad_pack")
            return
        for p, o, ix in zip(self.params, self.offsets, self.index):
            n = p.numel() if ix is None else ix.numel()
            seg = flat[o:o + n]
            if p.grad is None:
                seg.zero_()
            elif ix is None:
                seg.copy_(storage_flat(p.grad))
            else:
                torch.index_select(storage_flat(p.grad), 0, ix, out=seg)

    def unpack(self, scale):
        flat = self.flat
        if flat.is_cuda:
            import ctypes as C

            from . import _lib

            seg = self._segments(True)
            with torch.cuda.device(flat.device):
                _lib.check(_lib.lib().dm4d_grad_unpack(C.byref(seg), flat.data_ptr(), float(scale),
                                                       torch.cuda.current_stream(flat.device).cuda_stream), "dm4d_grad_unpack")
            return
        flat.mul_(scale)
        for p, o, ix in zip(self.params, self.offsets, self.index):
            n = p.numel() if ix is None else ix.numel()
            seg = flat[o:o + n]
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif p.grad.stride() != p.stride():
                raise ValueError("gradients must have the parameter's strides")
            if ix is None:
                storage_flat(p.grad).copy_(seg)
            else:
                storage_flat(p.grad).index_copy_(0, ix, seg)


def storage_flat(t):
    """1-D view of a dense tensor in STORAGE order (contiguous and channels_last tensors alike)."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.as_strided((t.numel(),), (1,), t.storage_offset())
    raise ValueError("parameters / gradients must be dense (contiguous or channels_last)")


def touched_from_plan(field, plan):
    """{spatial plane parameter: STORAGE indices of the elements that can receive gradient} from a `hexplane.HexPlan`
    (its `sp` lists are the touched texels per (scale, plane); element = channel * H * W + texel for a contiguous
    plane, texel * C + channel for a channels_last one).  The time planes are exchanged densely (which rows a step
    touches depends on the rank's timestamps)."""
    sc, pl, tx = (plan.sp[k].to(torch.long) for k in ("scale", "plane", "texel"))
    out = {}
    for s, planes in enumerate(field.grids):
        for p, par in enumerate(planes):
            m = (sc == s) & (pl == p)
            if not bool(m.any()):
                continue
            C, HW = int(par.shape[1]), int(par.shape[2]) * int(par.shape[3])
            t = tx[m]
            ch = torch.arange(C, device=t.device, dtype=torch.long)
            if par.is_contiguous():
                out[par] = (ch[:, None] * HW + t[None, :]).reshape(-1)
            else:
                storage_flat(par)                               # raises unless channels_last
                out[par] = (t[:, None] * C + ch[None, :]).reshape(-1)
    return out


def all_reduce_max(t):
    """In-place MAX over the ranks of a small tensor (the overflow flag of a step: if ANY rank's forward overflowed, every
    rank skips the optimiser step, so the replicas stay identical).  No-op for a single process."""
    if world() == 1:
        return t
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.MAX)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


# The reference's EFFECTIVE hyperparameters (verified with torch: `Adam(l, lr=0, eps=1e-15)` fills the dicts of `l` in place,
# `AdamW(l, betas=[0.9, 0.99], eps=1e-15)` afterwards only setdefault()s): the groups training_setup / training_setup_dynamic
# create -- points, f_dc, f_rest, all_densities, scales, quaternions, deformation, grid -- run betas (0.9, 0.999) and NO weight decay;
# groups appended by merge_optimizer from `net_optimizer` run AdamW's (0.9, 0.99) / 0.01
# (C/geometry/sugar.py:382,406-416, C/geometry/dynamic_sugar.py:231-235).
REFERENCE_GEOMETRY_GROUP = {"betas": (0.9, 0.999), "eps": 1e-15, "weight_decay": 0.0}
REFERENCE_MERGED_GROUP = {"betas": (0.9, 0.99), "eps": 1e-15, "weight_decay": 0.01}


def stage_optimizer_state(sharded, opt, global_step, gen):
    """The ``optimizer_states`` entry of a checkpoint for a training stage (DynamicStage / StaticStage ``optimizer_state_dict``): the
    optimiser that actually STEPS.  With the message-space optimiser and world > 1 this is COLLECTIVE -- every rank calls it; the
    moments are all-gathered into ``ShardedAdamW.full_state_dict()`` and the batch samplers' generator states (seeded per rank: the
    ranks draw different frames and cameras) are gathered into ``rng_states`` -- so the one file rank 0 writes resumes every rank with
    ITS moments slice and ITS generator (round 5 saved rank 0's shard and rank 0's generator: loaded on every rank without an error)."""
    if sharded is None:
        return {"kind": "torch.optim.AdamW", "state": opt.state_dict(), "global_step": int(global_step), "rng_state": gen.get_state()}
    w = world()
    sd = {"kind": "dm4d.ShardedAdamW", "state": sharded.full_state_dict() if w > 1 else sharded.state_dict(), "global_step": int(global_step),
          "rng_state": gen.get_state()}
    if w > 1:
        states = [None] * w
        dist.all_gather_object(states, gen.get_state())
        sd["rng_states"], sd["world"] = states, w
    return sd


def load_stage_optimizer_state(sd, sharded, opt, gen):
    """Inverse of ``stage_optimizer_state``; returns the checkpoint's global step (None if it has none)."""
    kind = "dm4d.ShardedAdamW" if sharded is not None else "torch.optim.AdamW"
    if sd.get("kind") != kind:
        raise ValueError(f"the checkpoint's optimiser state is a {sd.get('kind')}, this stage steps a {kind}")
    (sharded if sharded is not None else opt).load_state_dict(sd["state"])
    rs = sd.get("rng_states")
    if rs is not None and len(rs) == world():
        gen.set_state(rs[rank()].cpu())          # this rank's own sampler: the resumed run draws the frames / cameras the uninterrupted one would
    elif rs is not None:
        raise ValueError(f"the checkpoint holds the sampler states of {len(rs)} ranks, this run has {world()}: the frames / cameras drawn after "
                         "the resume cannot continue the saved run's")
    elif sd.get("rng_state") is not None:
        if world() > 1:
            raise ValueError("a single-process checkpoint's sampler state would make every rank draw the same frames and cameras")
        gen.set_state(sd["rng_state"].cpu())
    return sd.get("global_step")


class ShardedAdamW:
    """The AdamW step of the data-parallel loop with the optimiser state SHARDED over the ranks (SURVEY.md section 8e:
    "reduce-scatter -> sharded AdamW -> all-gather params"), in the MESSAGE space of a ``GradAllReducer``:

        gradients --pack--> flat message --reduce-scatter (sum, x 1/world)--> this rank's 1/world slice
        slice: AdamW (moments live only here); the parameters are read and written IN their storages through the message's index lists
        updated slices --all-gather--> flat message --unpack--> parameters

    Every rank ends with the same parameters as the replicated ``torch.optim.AdamW`` step after an all-reduce (same
    per-element arithmetic, same order of operations as torch's single-tensor implementation), but holds and updates
    1/world of the two moment buffers: at the shipped dynamic-stage configuration the replicated step streams
    35.76 M x (param + grad + 2 moments) = 572 MB per rank per iteration, the sharded one 1/world of the 13.5 MB message.
    Elements outside the message (HexPlane texels no node touches) have zero gradient on every rank, hence zero moments:
    their update is the weight decay alone, applied locally (and lazily: ``materialize``).

    ``groups``: an optimiser's ``param_groups`` -- [{"params": [...], "lr": float, "betas": (b1, b2), "eps": e, "weight_decay": w,
    "name": ...}]; lr / betas / eps / weight_decay are read PER GROUP at every step (so schedules and the reference's mixed
    hyperparameters work: ``REFERENCE_GEOMETRY_GROUP``), the constructor's values are the defaults of groups that lack a key.
    On a HIP device the step is the kernel pair of csrc/gradpack.hip (``dm4d_adamw_step``) for ANY world size (round 5; round 4
    ran the slice through torch operators when world > 1); on CPU tensors -- the gloo tests -- the same arithmetic in torch
    operators.  Works over RCCL (reduce_scatter_tensor / all_gather_into_tensor) and over gloo (all_reduce + slice / all_gather).
    A parameter whose ``.grad`` is None on a single process is skipped like torch.optim skips it (no decay, no moment decay, its
    step counter not advanced); with world > 1 a missing gradient is this rank's zeros in the sum, as in ``GradAllReducer``."""

    def __init__(self, groups, reducer: GradAllReducer, betas=(0.9, 0.99), eps=1e-15, weight_decay=0.01):
        self.param_groups = [dict(g) for g in groups]
        if not 1 <= len(self.param_groups) <= 8:
            raise ValueError("ShardedAdamW: 1..8 parameter groups")
        self.reducer, self.betas, self.eps, self.weight_decay = reducer, tuple(betas), eps, weight_decay
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize"):
                raise NotImplementedError("ShardedAdamW: amsgrad / maximize groups")
        w, r = world(), rank()
        n = reducer.flat.numel()
        self.chunk = (n + w - 1) // w
        dev = reducer.flat.device
        self.padded = torch.zeros(self.chunk * w, dtype=torch.float32, device=dev)
        self.lo, self.hi = r * self.chunk, (r + 1) * self.chunk
        self.exp_avg = torch.zeros(self.chunk, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.chunk, dtype=torch.float32, device=dev)
        self.step_count = 0
        nseg = len(reducer.params)
        self.step_t = torch.zeros(nseg, dtype=torch.float64, device=dev)          # steps APPLIED per segment (a step masked by found_inf, or a segment without a gradient, does not count)
        self.pending_decay = torch.ones(nseg, dtype=torch.float64, device=dev)    # per segment: product of the decay factors not yet applied to its elements OUTSIDE the message
        self._decay_pending = False                                               # (host: False while every step so far had weight_decay == 0 everywhere)
        gid_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        for p in reducer.params:
            if id(p) not in gid_of:
                raise ValueError("every reduced parameter must belong to a group")
        self._seg_group = [gid_of[id(p)] for p in reducer.params]
        # segment id of every message element of the local slice (the torch-operator path gathers its per-element scalars with it)
        sid = torch.zeros(self.chunk * w, dtype=torch.long, device=dev)
        for k, (p, o, ix) in enumerate(zip(reducer.params, reducer.offsets, reducer.index)):
            sid[o:o + (p.numel() if ix is None else ix.numel())] = k
        self.sid = sid[self.lo:self.hi].clone()
        self._scal = None

    fused = True      # HIP device: the step as two launches of csrc/gradpack.hip (dm4d_adamw_step); False: the torch-operator form (tests)

    def zero_grad(self, set_to_none=True):
        for p in self.reducer.params:
            p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)

    def _hyper(self):
        """Per group (lr, beta1, beta2, eps, weight_decay) as the groups hold them NOW."""
        out = []
        for g in self.param_groups:
            b = g.get("betas", self.betas)
            out.append((float(g["lr"]), float(b[0]), float(b[1]), float(g.get("eps", self.eps)), float(g.get("weight_decay", self.weight_decay))))
        return out

    def _pack_params(self):
        """parameter VALUES in message layout (the reducer's pack, applied to .data instead of .grad)."""
        red, flat = self.reducer, self.padded
        if flat.is_cuda:
            self._kernel_pack("dm4d_grad_pack")
            return
        for p, o, ix in zip(red.params, red.offsets, red.index):
            cnt = p.numel() if ix is None else ix.numel()
            src = storage_flat(p.data)
            flat[o:o + cnt].copy_(src if ix is None else src.index_select(0, ix))

    def _kernel_pack(self, fn):
        """dm4d_grad_pack / dm4d_grad_unpack between `padded` and the PARAMETER storages (one launch)."""
        import ctypes as C

        from . import _lib

        red, dev = self.reducer, self.padded.device
        seg = _lib.GradSegments()
        seg.n_segments = len(red.params)
        for k, (p, o, ix) in enumerate(zip(red.params, red.offsets, red.index)):
            seg.grad[k] = storage_flat(p.data).data_ptr()
            seg.index[k] = None if ix is None else ix.data_ptr()
            seg.count[k] = p.numel() if ix is None else ix.numel()
            seg.offset[k] = o
        st = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            if fn == "dm4d_grad_pack":
                _lib.check(_lib.lib().dm4d_grad_pack(C.byref(seg), self.padded.data_ptr(), st), fn)
            else:
                _lib.check(_lib.lib().dm4d_grad_unpack(C.byref(seg), self.padded.data_ptr(), 1.0, st), fn)

    @torch.no_grad()
    def step(self, found_inf=None):
        """found_inf (optional, scalar float tensor on the device, 1.0 = skip): the step is then masked ON THE DEVICE the way a
        fused torch optimiser skips on `found_inf` (no host sync): parameters, both moments and the step counters keep
        their values.  The collectives still run, so the ranks stay in lockstep; every rank must pass the same flag."""
        dev = self.padded.device
        hyper = self._hyper()
        self._decay_pending = self._decay_pending or any(h[4] != 0.0 for h in hyper)
        self.step_count += 1            # steps ATTEMPTED (host-side bookkeeping only)
        if dev.type == "cuda" and self.fused:
            return self._step_kernel(found_inf, hyper)
        return self._step_ops(found_inf, hyper)

    # ------------------------------------------------------------------ HIP: csrc/gradpack.hip
    def _step_kernel(self, found_inf, hyper):
        import ctypes as C

        from . import _lib

        red, dev, w = self.reducer, self.padded.device, world()
        n = red.flat.numel()
        if self._scal is None:
            self._scal = torch.zeros(1 + 2 * _lib.MAX_GRAD_SEGMENTS, dtype=torch.float32, device=dev)
            self._p_slice = torch.zeros(self.chunk, dtype=torch.float32, device=dev) if w > 1 else None
            self._g_slice = torch.zeros(self.chunk, dtype=torch.float32, device=dev) if w > 1 else None
        a = _lib.AdamwStepArgs()
        a.n_groups = len(hyper)
        for gi, (lr, b1, b2, eps, wd) in enumerate(hyper):
            a.lr[gi], a.beta1[gi], a.beta2[gi], a.eps[gi], a.weight_decay[gi] = lr, b1, b2, eps, wd
        a.exp_avg, a.exp_avg_sq = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        a.step, a.pending_decay = self.step_t.data_ptr(), self.pending_decay.data_ptr()
        fi = None if found_inf is None else found_inf.reshape(()).to(dev, torch.float32)
        a.found_inf = None if fi is None else fi.data_ptr()
        a.scratch = self._scal.data_ptr()
        if w == 1:
            # one process: gradients and parameters are read from / written to their storages through the message's index lists,
            # the moments live in message layout -- no pack, no unpack
            seg = red._segments(False)
            for k, p in enumerate(red.params):
                a.group[k] = self._seg_group[k]
                a.param[k] = storage_flat(p.data).data_ptr()
                a.skip[k] = 1 if p.grad is None else 0
            scale = 1.0
        else:
            # pack -> reduce-scatter -> the kernel on this rank's slice -> all-gather -> unpack into the parameters
            seg0 = red._segments(False)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().dm4d_grad_pack(C.byref(seg0), self.padded.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dm4d_grad_pack")
            g_slice = self._g_slice
            if dist.get_backend() == "gloo":
                # (functional rehearsal on a shared device: the SAME all-reduce over the SAME n elements as the replicated path, then this
                #  rank's slice -- a ring's order of additions depends on the element's position in the message, and AdamW with eps = 1e-15
                #  turns a last-bit difference of a gradient that is rounding noise into a full +-lr step)
                host = self.padded.cpu()
                dist.all_reduce(host[:n])
                g_slice.copy_(host[self.lo:self.hi])
            else:
                dist.reduce_scatter_tensor(g_slice, self.padded)
            seg = _lib.GradSegments()
            seg.n_segments = len(red.params)
            for k, (p, o, ix) in enumerate(zip(red.params, red.offsets, red.index)):
                cnt = p.numel() if ix is None else ix.numel()
                lo, hi = max(o, self.lo), min(o + cnt, self.hi)
                a.group[k] = self._seg_group[k]
                base = storage_flat(p.data).data_ptr()
                if hi <= lo:
                    seg.count[k], seg.offset[k], seg.grad[k], seg.index[k], a.param[k] = 0, 0, None, None, base
                    continue
                seg.count[k], seg.offset[k] = hi - lo, lo - self.lo
                seg.grad[k] = g_slice.data_ptr() + 4 * (lo - self.lo)
                a.grad_in_message[k] = 1
                a.param_out[k] = self._p_slice.data_ptr() + 4 * (lo - self.lo)
                if ix is None:
                    seg.index[k], a.param[k] = None, base + 4 * (lo - o)
                else:
                    seg.index[k], a.param[k] = ix.data_ptr() + 8 * (lo - o), base
            scale = 1.0 / w
            # (a step skipped by found_inf still fills the send slice: the kernel copies the CURRENT parameter values into param_out)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().dm4d_adamw_step(C.byref(seg), C.byref(a), scale, torch.cuda.current_stream(dev).cuda_stream), "dm4d_adamw_step")
        if w > 1:
            if dist.get_backend() == "gloo":
                parts = [torch.empty(self.chunk, dtype=torch.float32) for _ in range(w)]
                dist.all_gather(parts, self._p_slice.cpu())
                self.padded.copy_(torch.cat(parts))
            else:
                dist.all_gather_into_tensor(self.padded, self._p_slice)
            self._kernel_pack("dm4d_grad_unpack")

    # ------------------------------------------------------------------ torch operators (CPU tensors: the gloo tests; `fused = False`)
    def _step_ops(self, found_inf, hyper):
        red, w = self.reducer, world()
        n = red.flat.numel()
        dev = self.padded.device
        keep = None if found_inf is None else (found_inf.reshape(()).to(dev) == 0)          # bool scalar: True = apply
        has_grad = torch.tensor([1.0 if (p.grad is not None or w > 1) else 0.0 for p in red.params], dtype=torch.float64, device=dev)
        red.pack()
        self.padded[:n].copy_(red.flat)
        g_shard = torch.empty(self.chunk, dtype=torch.float32, device=dev)
        gloo = w > 1 and dist.get_backend() == "gloo"
        if w == 1:
            g_shard.copy_(self.padded)
        elif gloo:
            host = (self.padded.cpu() if self.padded.is_cuda else self.padded.clone())
            dist.all_reduce(host[:n])
            g_shard.copy_(host[self.lo:self.hi])
        else:
            dist.reduce_scatter_tensor(g_shard, self.padded)
        g_shard.mul_(1.0 / w)
        # ---- AdamW on the slice: torch/optim/adamw.py::_single_tensor_adamw, operation for operation (the step counts live
        #      on the device so that a skipped step does not advance the bias corrections)
        self._pack_params()
        p_old = self.padded[self.lo:self.hi].clone()
        apply_seg = has_grad if keep is None else has_grad * keep.to(torch.float64)          # [n_seg] 1 = this segment steps
        step_new = self.step_t + apply_seg
        sg = torch.tensor(self._seg_group, dtype=torch.long, device=dev)
        col = lambda j, dt: torch.tensor([h[j] for h in hyper], dtype=dt, device=dev)[sg]   # per SEGMENT
        lr_s, b1_s, b2_s, eps_s, wd_s = (col(j, torch.float32) for j in range(5))
        bc1_s = (1.0 - torch.pow(col(1, torch.float64), step_new)).to(torch.float32)
        bc2s_s = torch.sqrt(1.0 - torch.pow(col(2, torch.float64), step_new)).to(torch.float32)
        sid = self.sid
        lr, b2 = lr_s[sid], b2_s[sid]
        w1, w2 = (1.0 - b1_s)[sid], (1.0 - b2_s)[sid]
        p = p_old * (1.0 - lr * wd_s[sid])
        exp_avg = self.exp_avg + w1 * (g_shard - self.exp_avg)                   # torch.lerp(exp_avg, grad, 1 - beta1), weight < 0.5
        exp_avg_sq = self.exp_avg_sq * b2 + (w2 * g_shard) * g_shard
        denom = (exp_avg_sq.sqrt() / bc2s_s[sid]).add_(eps_s[sid])
        p = p + (exp_avg * (-(lr / bc1_s[sid]))) / denom
        decay_s = (1.0 - lr_s.double() * wd_s.double())                          # this step's decay factor per segment
        on = apply_seg[sid] != 0
        p = torch.where(on, p, p_old)
        exp_avg = torch.where(on, exp_avg, self.exp_avg)
        exp_avg_sq = torch.where(on, exp_avg_sq, self.exp_avg_sq)
        decay_s = torch.where(apply_seg != 0, decay_s, torch.ones_like(decay_s))
        self.exp_avg, self.exp_avg_sq, self.step_t = exp_avg, exp_avg_sq, step_new
        # ---- everyone gets every slice
        if w == 1:
            self.padded.copy_(p)
        elif gloo:
            parts = [torch.empty(self.chunk, dtype=torch.float32) for _ in range(w)]
            dist.all_gather(parts, p.cpu() if p.is_cuda else p)
            self.padded.copy_(torch.cat(parts))
        else:
            dist.all_gather_into_tensor(self.padded, p)
        # ---- back into the parameters.  Elements outside the message (HexPlane texels no node touches) have zero gradient
        #      and zero moments on every rank: their whole update is the weight decay -- and no forward ever READS them (the
        #      nodes are static).  Multiplying 134 MB of grids by the decay every step is therefore deferred: the product of the
        #      steps' decay factors is kept per segment on the device and applied by materialize() (checkpointing, state_dict, tests).
        self.pending_decay = self.pending_decay * decay_s
        self._unpack_message()

    def state_dict(self):
        """This RANK's shard of the optimiser state: its slice [lo, hi) of the two moments (message layout), the steps applied and the
        pending decay factors per segment, the groups' hyperparameters -- and whose slice it is (rank, lo, hi): ``load_state_dict``
        refuses another rank's shard.  With world > 1 a checkpoint therefore holds either one shard PER RANK or ``full_state_dict()``
        (what the stages' ``optimizer_state_dict()`` write).  (The message layout is a function of the reducer's parameters and
        touched-index sets: a checkpoint resumes into a stage constructed the same way.)"""
        return {"layout": "shard", "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "step": self.step_t.clone(),
                "pending_decay": self.pending_decay.clone(), "decay_pending": bool(self._decay_pending), "step_count": int(self.step_count),
                "hyper": self._hyper(), "world": world(), "rank": rank(), "lo": int(self.lo), "hi": int(self.hi),
                "elements": int(self.reducer.flat.numel()), "segments": len(self.reducer.params)}

    def full_state_dict(self):
        """COLLECTIVE (every rank calls it, every rank gets the result): the moments of the WHOLE message, all-gathered from the ranks'
        slices -- a state any rank of any world size can load (``load_state_dict`` takes its own slice), so one file written by
        rank 0 resumes every rank.  13.5 MB x 2 at the shipped dynamic-stage configuration."""
        sd = self.state_dict()
        w, n = world(), int(self.reducer.flat.numel())
        for k in ("exp_avg", "exp_avg_sq"):
            mine = getattr(self, k)
            if w == 1:
                full = mine
            elif dist.get_backend() == "gloo" or not mine.is_cuda:
                parts = [torch.empty(self.chunk, dtype=torch.float32) for _ in range(w)]
                dist.all_gather(parts, mine.detach().cpu().contiguous())
                full = torch.cat(parts).to(mine.device)
            else:
                full = torch.empty(self.chunk * w, dtype=torch.float32, device=mine.device)
                dist.all_gather_into_tensor(full, mine.contiguous())
            sd[k] = full[:n].clone()
        sd["layout"] = "full"
        for k in ("rank", "lo", "hi"):
            sd.pop(k)
        return sd

    def load_state_dict(self, sd):
        """A state of ``full_state_dict()`` (any writer world size: this rank takes its slice) or this rank's own shard of
        ``state_dict()`` -- another rank's shard, another world size's shard or another message layout raises."""
        n = int(self.reducer.flat.numel())
        if int(sd["elements"]) != n or int(sd.get("segments", -1)) != len(self.reducer.params):
            raise ValueError("ShardedAdamW.load_state_dict: the state belongs to another message layout")
        dev = self.exp_avg.device
        # (a state without "layout" was written by round 5: a shard that did not say whose -- only a single-process run can trust it)
        layout = sd.get("layout", "shard")
        if layout == "full":
            if tuple(sd["exp_avg"].shape) != (n,) or tuple(sd["exp_avg_sq"].shape) != (n,):
                raise ValueError("ShardedAdamW.load_state_dict: a full state holds the moments of the whole message")
            take = lambda t: torch.nn.functional.pad(t.to(dev, torch.float32), (0, self.padded.numel() - n))[self.lo:self.hi].clone()
            ea, es = take(sd["exp_avg"]), take(sd["exp_avg_sq"])
        else:
            mine = (int(sd.get("rank", 0 if world() == 1 else -1)), int(sd.get("lo", self.lo if world() == 1 else -1)), int(sd.get("hi", self.hi if world() == 1 else -1)))
            if int(sd["world"]) != world() or tuple(sd["exp_avg"].shape) != tuple(self.exp_avg.shape) or mine != (rank(), int(self.lo), int(self.hi)):
                raise ValueError(f"ShardedAdamW.load_state_dict: this is the shard of rank {sd.get('rank', '?')} of {sd['world']} (elements "
                                 f"[{sd.get('lo', '?')}, {sd.get('hi', '?')})), not of rank {rank()} of {world()} ([{self.lo}, {self.hi})): save "
                                 "full_state_dict() (collective) or one state_dict() per rank")
            ea, es = sd["exp_avg"].to(dev, torch.float32).clone(), sd["exp_avg_sq"].to(dev, torch.float32).clone()
        self.exp_avg, self.exp_avg_sq = ea, es
        self.step_t = sd["step"].to(dev, torch.float64).clone()
        self.pending_decay = sd["pending_decay"].to(dev, torch.float64).clone()
        self._decay_pending = bool(sd.get("decay_pending", True))
        self.step_count = int(sd.get("step_count", 0))
        for g, (lr, b1, b2, eps, wd) in zip(self.param_groups, sd["hyper"]):
            g["lr"], g["betas"], g["eps"], g["weight_decay"] = lr, (b1, b2), eps, wd

    def _unpack_message(self):
        red = self.reducer
        if self.padded.is_cuda:
            self._kernel_pack("dm4d_grad_unpack")
            return
        for q, o, ix in zip(red.params, red.offsets, red.index):
            cnt = q.numel() if ix is None else ix.numel()
            seg = self.padded[o:o + cnt]
            if ix is None:
                storage_flat(q.data).copy_(seg)
            else:
                storage_flat(q.data).index_copy_(0, ix, seg)

    @torch.no_grad()
    def materialize(self):
        """Apply the deferred weight decay of the elements outside the message (see step()); afterwards every parameter
        element equals what the replicated AdamW would hold.  Call before reading the parameters as a whole (checkpoints).
        Nothing to do while no group has a weight decay (the reference's effective configuration of the geometry groups)."""
        if not self._decay_pending:
            return
        red = self.reducer
        ks = [k for k, ix in enumerate(red.index) if ix is not None]
        if ks:
            # the touched elements carry their own (exact) values: taken from the PARAMETERS as they are now (a load_state_dict
            # since the last step must not be overwritten by the stale message buffer), decayed with the rest, written back
            self._pack_params()
            factors = self.pending_decay.to(torch.float32)
            torch._foreach_mul_([red.params[k].data for k in ks], [factors[k] for k in ks])       # one multi-tensor launch
            self._unpack_message()
        self.pending_decay = torch.ones_like(self.pending_decay)
        self._decay_pending = False
