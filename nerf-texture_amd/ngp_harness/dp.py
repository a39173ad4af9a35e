"""Data parallelism for the hot path: rays are sharded across one-process-per-GPU ranks, every rank keeps a
full replica of the hash table + MLP weights, and ONE all-reduce per optimizer step sums the flat gradient.

The reference only wraps its model in torch DDP and never enables it (nerf/utils.py:439-441, world_size=1); DDP
would bucket the 48 MiB table gradient into many NCCL ring all-reduces.  Here the gradient of every parameter is
a VIEW into one contiguous fp32 buffer, so the exchange is a single RCCL all-reduce with no pack / unpack copies
(xGMI is point-to-point: one large collective per step is the cheap shape).  State that must stay identical on
all ranks besides the weights: the occupancy bitfield (same seed / same analytic scene on every rank here) and
`mean_count` (all-reduce MAX so every rank sizes its sample buffers alike).

backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test rig for boxes with ONE GPU: NERFTEX_DP_SHARE_GPU=1 puts every rank on cuda:0 and exchanges through gloo (RCCL refuses
    # two ranks on one device), so the multi-process step -- two graphs per step, eager all-reduce of fp16 gradients, mean_count
    # agreement -- runs end to end; not a performance configuration
    if os.environ.get("NERFTEX_DP_SHARE_GPU") == "1":
        local, backend = 0, "gloo"
    if world > 1 and not dist.is_initialized():
        import datetime
        import sys

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # the host driver only supports dmabuf IPC: without this RCCL's peer mappings fail with `hipIpcGetMemHandle: invalid argument`
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        # a rendezvous or a first collective that hangs (a rank that died, a GPU another process holds, a wrong MASTER_PORT) must end in a
        # message, not in a silent stall of the whole node: NERFTEX_DP_TIMEOUT_S seconds (default 180) for the group and every collective
        timeout = datetime.timedelta(seconds=float(os.environ.get("NERFTEX_DP_TIMEOUT_S", "180")))
        try:
            if backend == "nccl":
                if local >= torch.cuda.device_count():
                    raise RuntimeError(f"LOCAL_RANK {local} but only {torch.cuda.device_count()} visible GPU(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?)")
                torch.cuda.set_device(local)
                dist.init_process_group(backend=backend, device_id=torch.device("cuda", local), timeout=timeout)
                # first contact: one tiny all-reduce here, where a failure can still be explained (RCCL sets up its xGMI rings lazily, in the
                # first collective -- without this the first symptom would be a hang inside the first training step)
                probe = torch.ones(1, device=torch.device("cuda", local))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError(f"first all-reduce returned {probe.item()} on rank {rank}, expected {world}")
            else:
                dist.init_process_group(backend=backend, timeout=timeout)
        except Exception as e:  # noqa: BLE001 -- re-raised below with what a person at an 8-GPU node needs to know
            print(f"[dp] rank {rank}/{world} (local {local}): process group '{backend}' failed to come up within {timeout.total_seconds():.0f} s: "
                  f"{type(e).__name__}: {e}\n[dp] rendezvous {os.environ['MASTER_ADDR']}:{os.environ['MASTER_PORT']}, HSA_ENABLE_IPC_MODE_LEGACY="
                  f"{os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')}, visible GPUs {torch.cuda.device_count() if torch.cuda.is_available() else 0}; "
                  f"NCCL_DEBUG=INFO shows RCCL's own log, NERFTEX_DP_TIMEOUT_S changes the limit", file=sys.stderr, flush=True)
            raise
    return rank, world, local


def preflight(expected_world, dev=None):
    """What a multi-GPU run checks BEFORE anything is timed (the first time an 8-GPU node appears the run must either measure or
    say exactly why not): the process group has `expected_world` ranks, one small all-reduce per wire type returns the known sum on every rank
    (fp16 and fp32 SUM -- the gradient exchange -- and an int32 MAX -- the mean_count agreement), a barrier completes.  Returns a dict for the
    bench line's `config.collective` (backend, world size, the collective library's version, device names); raises RuntimeError with the rank and
    the failing check otherwise.  Runs on gloo / CPU tensors as well (tests/test_dp_gloo.py at world 8).  The reference's only precedent for any of
    this is its dormant DistributedDataParallel wrap, nerf/utils.py:439-441."""
    if not (dist.is_available() and dist.is_initialized()):
        if expected_world != 1:
            raise RuntimeError(f"preflight: no process group, but {expected_world} ranks were asked for")
        return {"world_size": 1, "backend": None}
    world, rank, backend = dist.get_world_size(), dist.get_rank(), dist.get_backend()
    if world != expected_world:
        raise RuntimeError(f"preflight: the process group has {world} ranks, {expected_world} were asked for (rank {rank})")
    if dev is None:
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    want_sum = world * (world + 1) // 2
    for dt in (torch.float16, torch.float32):
        if backend == "gloo" and dt == torch.float16 and dev.type == "cpu":
            t = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)  # (gloo has no fp16 sum on CPU tensors)
        else:
            t = torch.full((1024,), float(rank + 1), dtype=dt, device=dev)
        dist.all_reduce(t)
        if not bool((t == want_sum).all()):
            raise RuntimeError(f"preflight: {dt} all-reduce returned {float(t[0])} on rank {rank}, expected {want_sum}")
    m = torch.tensor([rank + 7], dtype=torch.int32, device=dev)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    if int(m.item()) != world + 6:
        raise RuntimeError(f"preflight: int32 MAX all-reduce returned {int(m.item())} on rank {rank}, expected {world + 6}")
    dist.barrier()
    info = {"world_size": world, "backend": backend + (" (RCCL)" if backend == "nccl" else ""), "preflight": "world size, fp16 / fp32 SUM and int32 MAX all-reduce, barrier: ok"}
    if backend == "nccl":
        try:
            info["collective_library_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:  # noqa: BLE001
            info["collective_library_version"] = f"unknown ({type(e).__name__})"
        names = [None] * world
        dist.all_gather_object(names, torch.cuda.get_device_name(dev))
        info["devices"] = names
    return info


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard(n_global, rank, world):
    """Contiguous shard [lo, hi) of a global ray batch (rank r gets rays [r*n/w, (r+1)*n/w))."""
    per = n_global // world
    lo = rank * per
    hi = n_global if rank == world - 1 else lo + per
    return lo, hi


def sync_occupancy(renderer, src=0, check_only=False):
    """The must-sync state of SURVEY.md 8(e) besides the weights: density_grid, density_bitfield, mean_density, iter_density.
    Ranks that run `update_extra_state_device` with the same seed already agree (the jitter and the cell picks are a function of (seed,
    row), the field replicas are identical), so the default use is `check_only=True` -- one MIN/MAX all-reduce of a checksum, returns
    whether all ranks hold the same grid -- and a broadcast from `src` (4.5 MiB for cascade 2) is the repair / the mode for a rank-0-only
    update."""
    if world_size() == 1:
        return True
    grid, bits = renderer.density_grid, renderer.density_bitfield
    if check_only:
        sums = torch.stack([grid.double().sum(), grid.double().abs().sum(), bits.double().sum(),
                            torch.as_tensor(float(renderer.mean_density), dtype=torch.float64, device=grid.device)])
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))
    dist.broadcast(grid, src)
    dist.broadcast(bits, src)
    meta = torch.tensor([float(renderer.mean_density), float(renderer.iter_density)], dtype=torch.float64, device=grid.device)
    dist.broadcast(meta, src)
    renderer.mean_density, renderer.iter_density = float(meta[0]), int(meta[1])
    return True


def broadcast(tensors, src=0):
    """Rank `src`'s values to every rank (replica initialisation)."""
    if world_size() > 1:
        for t in tensors:
            dist.broadcast(t, src)


class FlatGradAllReduce:
    """Gradient exchange of data-parallel training: one collective for everything small, one per big tensor.

    Small parameters (the MLP weights) get `.grad` views into ONE flat fp32 buffer, zeroed per step and all-reduced in one call.
    A big parameter (the hash table: 12.6 M floats) is left to autograd instead: its `.grad` is dropped every step, so the backward
    pass hands its freshly computed gradient tensor over without the 3 x 48 MB read-add-write into a persistent buffer and
    without the 48 MB zero fill; that tensor is all-reduced in place."""

    def __init__(self, params, average=True, big_numel=1 << 20, big_comm_dtype=None):
        params = [p for p in params if p.requires_grad]
        assert params, "no trainable parameters"
        self.big = [p for p in params if p.numel() >= big_numel]
        self.small = [p for p in params if p.numel() < big_numel]
        self.params = params
        dev = params[0].device
        total = sum(p.numel() for p in self.small)
        self.flat = torch.zeros(max(total, 1), dtype=torch.float32, device=dev)
        self.average = average
        # big_comm_dtype=torch.float16: a big gradient travels as fp16.  Under autocast the hash-table gradient IS fp16-valued (the
        # kernel accumulates into an fp16 tensor, autograd widens it), so the narrowing is lossless locally; only the cross-rank sum
        # rounds, and a sum that overflows becomes inf, which GradScaler treats like any other overflow (skip + rescale).
        # Halves the bytes on xGMI, which at 2 GPUs (one link) is most of the all-reduce time.
        self.big_comm_dtype = big_comm_dtype
        off = 0
        for p in self.small:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero_grad(self):
        if self.small:
            self.flat.zero_()
        for p in self.big:
            p.grad = None

    def broadcast_parameters(self, src=0):
        if world_size() > 1:
            for p in self.params:
                dist.broadcast(p.data, src)

    def big_grads(self):
        """The gradient tensors of the big parameters as they stand (a graph capture keeps them: their addresses are baked into it)."""
        return [p.grad for p in self.big]

    def all_reduce_start(self, grads=None):
        """Launch the gradient exchange without waiting for it (the collectives run on the backend's own stream / thread): whatever
        the caller enqueues next on the current stream overlaps with it.  Returns a handle for `all_reduce_finish`."""
        if world_size() == 1:
            return None
        works, post = [], []
        off = 0
        for p in self.small:  # a caller's optimizer.zero_grad(set_to_none=True) drops the views into the flat buffer: put them back
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n
        for p, g in zip(self.big, grads if grads is not None else self.big_grads()):
            if g is None:
                continue
            if self.big_comm_dtype is not None and g.dtype != self.big_comm_dtype:
                wire = g.to(self.big_comm_dtype)
                works.append(dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True))
                post.append((g, wire))
            else:
                works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True))
                post.append((g, None))
        if self.small:
            works.append(dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True))
        return works, post

    def start_tensor(self, t):
        """Asynchronous sum of ONE tensor (a chunk of a big gradient) across the ranks, on the wire dtype of the big gradients; returns a
        handle for `finish_tensors`."""
        if world_size() == 1:
            return None
        if self.big_comm_dtype is not None and t.dtype != self.big_comm_dtype:
            wire = t.to(self.big_comm_dtype)
            return (dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True), t, wire)
        return (dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t, None)

    def finish_tensors(self, handles):
        w = world_size()
        for h in handles:
            if h is None:
                continue
            work, t, wire = h
            work.wait()
            if wire is not None:
                t.copy_(wire)
            if self.average:
                t.div_(w)

    def all_reduce_finish(self, handle):
        """Make the current stream wait for the exchange started by `all_reduce_start`, then the local epilogue (widen / average)."""
        if handle is None:
            return
        works, post = handle
        for wk in works:
            wk.wait()
        w = world_size()
        for g, wire in post:
            if wire is not None:
                g.copy_(wire)
            if self.average:
                g.div_(w)
        if self.small and self.average:
            self.flat.div_(w)

    def all_reduce(self, extra=None, grads=None):
        """Sum (then average) the gradients: biggest tensors first, then the flat buffer of the small ones; `extra` (e.g. a
        loss or found-inf flag) is a further tiny all-reduce.  grads: the big parameters' gradient tensors to use instead of the
        current `.grad` (replayed graphs write to the tensors that existed when they were captured)."""
        if world_size() == 1:
            return extra
        self.all_reduce_finish(self.all_reduce_start(grads))
        if extra is not None:
            dist.all_reduce(extra, op=dist.ReduceOp.SUM)
        return extra


class TableGradChunks:
    """The hash table's gradient exchanged level group by level group: the large-batch backward first bins every level's
    contributions (one kernel), then sums them tile by tile, level by level -- so the rows of the first levels are final long before the
    last ones, and their all-reduce can be on the wire while the rest is still being summed, instead of everything starting after the
    backward pass (the table gradient is the LAST thing the backward produces and the first the optimizer needs: nothing else of the step
    can hide the exchange).

        chunks = TableGradChunks(field.encoder, k)      # attaches itself: the fused field's backward now only bins (ngp_harness/fused.py)
        loss.backward()
        for i in range(len(chunks)):
            chunks.sum_chunk(i)                          # finishes rows chunks.rows[i] of the table gradient (current stream)
            handles.append(reducer.start_tensor(chunks.view(i, table_leaf.grad)))
        ...; reducer.finish_tensors(handles)

    Level groups hold about equal numbers of rows, in level order.  Same gradient bits as the one-call backward (the sums are exact)."""

    def __init__(self, encoder, chunks):
        off = encoder.offsets.detach().cpu().tolist()
        L = len(off) - 1
        chunks = max(1, min(int(chunks), L))
        bounds, lo = [], 0
        for c in range(chunks):
            if c == chunks - 1:
                hi = L
            else:
                want = off[-1] * (c + 1) / chunks
                hi = lo + 1
                while hi < L - (chunks - 1 - c) and off[hi] < want:
                    hi += 1
            bounds.append((lo, hi))
            lo = hi
        self.levels = bounds
        self.rows = [(off[a], off[b]) for a, b in bounds]
        self.grad_ptr, self._run, self._keep, self._summed = None, None, None, None
        self.encoder = encoder
        encoder.grad_chunker = self

    def __len__(self):
        return len(self.levels)

    def begin(self, grad, run, keep):
        """Called by the backward: `grad` the table-gradient tensor being produced (only its ADDRESS is kept: a second reference would make
        autograd copy the gradient instead of handing the tensor itself to `.grad`), run(level_lo, level_hi) finishes a level range (None:
        the gradient is complete already), keep: what must stay alive until the last range has run."""
        leaf = getattr(self.encoder, "half_leaf", None)
        if run is not None and leaf is not None and leaf.grad is not None:
            raise RuntimeError("TableGradChunks: the table leaf already has a .grad -- autograd would ACCUMULATE the still unfinished gradient into it and the "
                               "level groups summed afterwards would never reach it; clear it first (leaf.grad = None / zero_grad(set_to_none=True))")
        self.grad_ptr, self._run, self._keep = grad.data_ptr(), run, keep
        self._summed = set() if run is not None else None

    def take(self):
        """(address of the gradient, run, keep) of the backward that just ran, for a caller that records the per-group work itself (graph capture)."""
        return self.grad_ptr, self._run, self._keep

    def sum_chunk(self, i, state=None):
        run = (state or self.take())[1]
        if run is not None:
            run(*self.levels[i])
            if state is None and self._summed is not None:
                self._summed.add(i)

    def complete(self):
        """True once every level group of the last backward has been summed (eager use; a caller that replays recorded sums keeps its own
        books): an optimizer step before that would read rows the backward left uninitialised."""
        return self._summed is None or len(self._summed) == len(self.levels)

    def view(self, i, grad, state=None):
        """Rows of level group i of `grad` -- the tensor autograd put into `.grad`, which must be the one the backward produced."""
        assert grad.data_ptr() == (state or self.take())[0], "the table gradient in .grad is not the tensor the backward wrote (was it copied or accumulated?)"
        a, b = self.rows[i]
        return grad[a:b]


def all_reduce_max_int(value, device):
    if world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def barrier():
    if world_size() > 1:
        dist.barrier()
