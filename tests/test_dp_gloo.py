"""N>1 path on CPU: world_size-2 gloo processes exercise the ray sharding + the gradient exchange (one flat all-reduce for the
small parameters, an in-place all-reduce of autograd's own tensor for each big one)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, comm_dtype=None):
    sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ngp_harness import dp

    r, w, _ = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)  # different init per rank: broadcast must fix it
    model = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
    # threshold 100: the first layer (128 weights) takes the "big tensor" route, the second (48) the flat-buffer route
    red = dp.FlatGradAllReduce(model.parameters(), big_numel=100, big_comm_dtype=comm_dtype)
    assert len(red.big) == 1 and len(red.small) == 1
    red.broadcast_parameters()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15)

    torch.manual_seed(123)
    x_global = torch.randn(64, 8)
    y_global = torch.randn(64, 3)
    lo, hi = dp.shard(64, rank, world)
    assert hi - lo == 64 // world
    for _ in range(3):
        red.zero_grad()
        loss = torch.nn.functional.mse_loss(model(x_global[lo:hi]), y_global[lo:hi])
        loss.backward()
        assert model[2].weight.grad.data_ptr() == red.flat.data_ptr(), "a small .grad must stay a view of the flat buffer"
        assert model[0].weight.grad is not None and model[0].weight.grad.data_ptr() != red.flat.data_ptr()
        if _ == 1:  # the replayed-graph form: the caller names the gradient tensors (same objects here)
            red.all_reduce(grads=red.big_grads())
        elif _ == 2:  # the overlapped form: start, do unrelated work, finish
            handle = red.all_reduce_start(red.big_grads())
            _unrelated = torch.randn(64, 64) @ torch.randn(64, 64)
            red.all_reduce_finish(handle)
        else:
            red.all_reduce()
        opt.step()
    # must-sync state besides the weights: the occupancy grid.  Different grids are detected, a broadcast repairs them.
    import types

    gen = torch.Generator().manual_seed(50 + rank)
    occ = types.SimpleNamespace(density_grid=torch.rand(2, 512, generator=gen), density_bitfield=torch.randint(0, 256, (128,), generator=gen, dtype=torch.uint8),
                                mean_density=1.0 + rank, iter_density=3 + rank)
    differs = not dp.sync_occupancy(occ, check_only=True)
    dp.sync_occupancy(occ, src=0)
    agree = dp.sync_occupancy(occ, check_only=True)
    ref_gen = torch.Generator().manual_seed(50)
    occ_ok = differs and agree and torch.equal(occ.density_grid, torch.rand(2, 512, generator=ref_gen)) and occ.mean_density == 1.0 and occ.iter_density == 3
    # mean_count agreement (every rank sizes its sample buffers by the MAX): the largest value sits on a middle rank
    mc = dp.all_reduce_max_int(100 + (7 * rank) % world, torch.device("cpu")) if occ_ok else -1
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    q.put((rank, flat.numpy().copy(), mc))  # by value: torch tensors travel as shared-memory fds that die with the worker
    dp.barrier()
    dist.destroy_process_group()


def _single_process_reference():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16, bias=False), torch.nn.ReLU(), torch.nn.Linear(16, 3, bias=False))
    opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15)
    torch.manual_seed(123)
    x = torch.randn(64, 8)
    y = torch.randn(64, 3)
    for _ in range(3):
        opt.zero_grad()
        # mean over the global batch == average of the two equal-size shard means
        torch.nn.functional.mse_loss(model(x), y).backward()
        opt.step()
    return torch.cat([p.detach().reshape(-1) for p in model.parameters()])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8], ids=["world2", "world8"])
@pytest.mark.parametrize("comm_dtype", [None, torch.float16], ids=["fp32-wire", "fp16-wire"])
def test_n_rank_gloo_matches_single_process(comm_dtype, world):
    """shard / FlatGradAllReduce (flat route, big-tensor route, the named-tensor and the overlapped forms) / sync_occupancy / all_reduce_max_int
    at world 2 and at world 8 -- configs[4]'s rank count (8 ranks x 8 of the 64 samples)."""
    port = 29600 + os.getpid() % 300 + (17 if comm_dtype is not None else 0) + 40 * (world == 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, comm_dtype)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort(key=lambda t: t[0])
    w0 = torch.from_numpy(results[0][1])
    for r in range(1, world):
        assert torch.equal(w0, torch.from_numpy(results[r][1])), "replicas must stay bit-identical"
    assert all(res[2] == 100 + world - 1 for res in results)
    ref = _single_process_reference()
    # fp16 on the wire rounds the big tensor's summed gradient to 11 bits; Adam's normalised step keeps the effect at ~1e-3 of lr
    assert torch.allclose(w0, ref, atol=1e-6 if comm_dtype is None else 2e-3), "N-rank DP == single-process training on the global batch"


# ---------------------------------------------------------------------------------------------------------------------------
# The step the benchmark actually runs at N > 1: fp16 LEAVES (HalfLeafAdam) whose fp16 gradients are exchanged in place, fp16 or
# fp32 on the wire, the loss scaler (FusedAmp) deciding skip / back-off from the EXCHANGED gradients -- so every rank decides alike.
# The two HIP launches are replaced by torch stand-ins (tests/cpu_half_adam.py); everything around them is the product code.
class _Owner(torch.nn.Module):
    def __init__(self, shape, seed):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.weights = torch.nn.Parameter(torch.randn(shape, generator=gen) * 0.3)


def _half_leaf_setup(seed_shift, second_leaf=torch.float16):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_half_adam import CpuFusedAmp, CpuHalfLeafAdam

    a, b = _Owner((16, 8), 1 + seed_shift), _Owner((3, 16), 2 + seed_shift)
    # (second_leaf bf16: the mixed set of the bf16 fused field -- fp16 table leaf, bf16 MLP leaves -- accelerate.py)
    opt = CpuHalfLeafAdam([(a, "weights"), (b, "weights", second_leaf)], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    amp = CpuFusedAmp(opt, init_scale=2.0 ** 14, growth_interval=2)
    return a, b, opt, amp


def _half_leaf_steps(a, b, opt, amp, x, y, n_steps, loss_mul, exchange, poison_step=None, poison=False):
    for k in range(n_steps):
        for leaf in opt.leaves:
            leaf.grad = None
        h = torch.relu(x @ a.half_leaf.float().t()) @ b.half_leaf.float().t()
        loss = torch.nn.functional.mse_loss(h, y) * loss_mul
        amp.scale_loss(loss).backward()
        assert all(leaf.grad.dtype == leaf.dtype and leaf.dtype in (torch.float16, torch.bfloat16) for leaf in opt.leaves)
        if poison and k == poison_step:
            opt.leaves[0].grad[0, 0] = float("inf")  # ONE rank overflows: after the exchange every rank must see it and skip
        exchange()
        amp.step()


def _half_leaf_worker(rank, world, port, q, wire, second_leaf=torch.float16):
    sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ngp_harness import dp

    dp.init_from_env(backend="gloo")
    a, b, opt, amp = _half_leaf_setup(10 * rank, second_leaf)  # different init per rank: the broadcast must fix it
    dp.broadcast([a.weights.data, b.weights.data])
    opt.resync()
    red = dp.FlatGradAllReduce(opt.trainable(), average=False, big_numel=0, big_comm_dtype=wire)  # bench.py's construction for the fused optimizer
    assert len(red.big) == 2 and not red.small
    torch.manual_seed(123)
    x, y = torch.randn(64, 8), torch.randn(64, 3)
    lo, hi = dp.shard(64, rank, world)
    _half_leaf_steps(a, b, opt, amp, x[lo:hi], y[lo:hi], 5, 1.0 / world, red.all_reduce, poison_step=2, poison=rank == 1)
    q.put((rank, torch.cat([a.weights.detach().reshape(-1), b.weights.detach().reshape(-1)]).numpy().copy(),
           torch.cat([l.detach().float().reshape(-1) for l in opt.leaves]).numpy().copy(), amp.get_scale(), float(opt.step_count)))
    dp.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8], ids=["world2", "world8"])
@pytest.mark.parametrize("wire", [torch.float32, None], ids=["fp32-wire", "fp16-wire"])
def test_n_rank_half_leaf_adam_matches_single_process(wire, world):
    port = 29950 + os.getpid() % 300 + (23 if wire is None else 0) + 50 * (world == 8)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_half_leaf_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    _, m0, l0, s0, c0 = results[0]
    for _, m1, l1, s1, c1 in results[1:]:
        assert (m0 == m1).all() and (l0 == l1).all(), "replicas stay bit-identical (masters and fp16 leaves)"
        assert s0 == s1 and c0 == c1 == 4.0, "one of five steps overflowed on ONE rank: every rank skipped it and backed the scale off alike"
    # single process on the global batch, same overflow step
    sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
    a, b, opt, amp = _half_leaf_setup(seed_shift=0)
    torch.manual_seed(123)
    x, y = torch.randn(64, 8), torch.randn(64, 3)
    _half_leaf_steps(a, b, opt, amp, x, y, 5, 1.0, lambda: None, poison_step=2, poison=True)
    ref = torch.cat([a.weights.detach().reshape(-1), b.weights.detach().reshape(-1)]).numpy()
    assert amp.get_scale() == s0 and float(opt.step_count) == 4.0
    # fp32 wire: the two shard gradients (each rounded to fp16) are summed in fp32 and rounded once -- within 1.5 fp16 ulps of the
    # single-process gradient, i.e. 2^-10 relative on Adam's normalised step: 4 steps x lr x 2e-3.  fp16 wire: one more rounding.
    import numpy as np

    # (world 8: eight shard gradients, each rounded to fp16, and on the fp16 wire a chain of seven more roundings)
    bar = 4 * 1e-2 * (2e-3 if wire is not None else 4e-3) * (1 if world == 2 else 2)
    assert np.abs(m0 - ref).max() <= bar, (float(np.abs(m0 - ref).max()), bar)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("wire", [torch.float32, None], ids=["fp32-wire", "leaf-dtype-wire"])
def test_two_rank_mixed_fp16_bf16_leaves_match_single_process(wire):
    """The leaf set of the bf16 fused field (one fp16 leaf, one bf16 leaf in ONE HalfLeafAdam; optim.py bf16_mask) under gloo: replicas stay
    bit-identical, the overflow seen by one rank skips the step on both, and the result is the single-process one to bf16 rounding."""
    import numpy as np

    world = 2
    port = 29950 + os.getpid() % 300 + 411 + (17 if wire is None else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_half_leaf_worker, args=(r, world, port, q, wire, torch.bfloat16)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, m0, l0, s0, c0), (_, m1, l1, s1, c1) = results
    assert (m0 == m1).all() and (l0 == l1).all() and s0 == s1 and c0 == c1 == 4.0
    a, b, opt, amp = _half_leaf_setup(0, torch.bfloat16)
    assert opt.leaves[0].dtype == torch.float16 and opt.leaves[1].dtype == torch.bfloat16 and opt.bf16_mask == 0b10
    torch.manual_seed(123)
    x, y = torch.randn(64, 8), torch.randn(64, 3)
    _half_leaf_steps(a, b, opt, amp, x, y, 5, 1.0, lambda: None, poison_step=2, poison=True)
    ref = torch.cat([a.weights.detach().reshape(-1), b.weights.detach().reshape(-1)]).numpy()
    assert amp.get_scale() == s0 and float(opt.step_count) == 4.0
    # the bf16 leaf's shard gradients are rounded to 8 bits before the sum: 2^-8 relative on Adam's normalised step, 4 steps x lr, and the
    # forward of later steps reads a bf16 weight -- 4 x 1e-2 x 1.6e-2
    assert np.abs(m0 - ref).max() <= 4 * 1e-2 * 1.6e-2, float(np.abs(m0 - ref).max())


def test_shard_covers_batch():
    sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
    from ngp_harness import dp

    for n, w in ((65536, 8), (4096, 2), (1000, 3)):
        spans = [dp.shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))


# ------------------------------------------------------------------------------------------------- the multi-GPU pre-flight (round 6)
def _preflight_worker(rank, world, port, q, expected):
    import os

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ngp_harness import dp

    dp.init_from_env(backend="gloo")
    try:
        q.put((rank, "ok", dp.preflight(expected, torch.device("cpu"))))
    except RuntimeError as e:
        q.put((rank, "error", str(e)))


@pytest.mark.parametrize("world, expected", [(8, 8), (2, 4)], ids=["world8_ok", "wrong_world_size_is_refused"])
def test_preflight_checks_the_group_before_anything_is_timed(world, expected):
    """dp.preflight -- what `bench.py --gpus N` runs before its first timed step: the group has N ranks, one all-reduce per wire type returns the
    known sum on every rank, a barrier completes; its report goes into the bench line (config.collective).  At world 8, configs[4]'s rank
    count, on gloo; and a group of the wrong size is refused on every rank with a message that names both sizes."""
    import multiprocessing as mp

    port = 29350 + os.getpid() % 200 + 7 * world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_preflight_worker, args=(r, world, port, q, expected)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted([q.get(timeout=150) for _ in range(world)], key=lambda t: t[0])
    [p.join(30) for p in procs]
    if world == expected:
        assert all(r[1] == "ok" for r in results), results
        info = results[0][2]
        assert info["world_size"] == 8 and info["backend"] == "gloo" and "ok" in info["preflight"]
    else:
        assert all(r[1] == "error" and "has 2 ranks, 4 were asked for" in r[2] for r in results), results


def test_bench_reports_why_it_fell_back():
    """The two opt-in exchange structures never fail the run: a backend that cannot capture the collective, or a configuration without the fused
    optimizer path, falls back to the eager exchange and the REASON travels in the JSON line (config.collective.fallbacks).  Checked on the source:
    every assignment of a fallback is next to the stderr message it mirrors, and the line carries the list."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")).read()
    assert src.count('ar_state["fallback"] = ') == 2 and src.count('ar_state["chunk_fallback"] = ') == 1
    assert '"fallbacks": [v for v in (ar_state.get("fallback"), ar_state.get("chunk_fallback")) if v]' in src
    assert "pre = dp.preflight(world, dev) if world > 1 else None" in src and 'sys.exit(3)' in src
