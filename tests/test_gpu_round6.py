"""GPU: round 6.

  * nerftex_grid_encode_backward_adam + nerftex_adam_mixed_step_amp_db (the hash-grid backward's summing kernel applies Adam to the tiles it owns,
    over double-buffered optimizer state) against nerftex_grid_encode_backward_amp + nerftex_adam_mixed_step_amp: the same fp32 masters, moments
    and fp16 table, bit for bit, over steps that include an inf in the incoming gradient, a row that overflows fp16 as a SUM of finite shares
    (the case no tile can see coming) and a scale that is not a power of two;
  * accelerate(fused_table_update=True) against accelerate(fused_table_update=False): same losses and parameters through priming, warm-up,
    replayed graphs and a forced overflow.
The optimizer being restated is the reference's: main_nerf.py:128 `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` under the GradScaler of
nerf/utils.py:1003-1009 (tests/test_gpu_trainstep.py pins the streaming kernel to torch's fused Adam; this file pins the tile form to that kernel).
"""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _bits(t):
    return t.view(torch.int32 if t.dtype == torch.float32 else torch.int16)


@pytest.mark.parametrize("B", [65536, 4096, 459264], ids=["two_shared_levels", "no_shared_level", "bench_size"])
def test_backward_adam_equals_backward_then_adam(dev, oracle, knobs, B):
    knobs(grid_bwd=2)  # the two-launch side on the binned path at every size (small batches would take the fp16-atomics path: another rounding order)
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, TableAdam, check, lib, ptr, stream

    off_np, rows = oracle.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
    S = float(np.log2(1.447269))
    gen = torch.Generator(device=dev).manual_seed(B)
    x = torch.rand(B, 3, device=dev, generator=gen) * 4 - 2
    n_w = 7168  # a second, small tensor in the closing launch (an MLP's weights)

    def fresh():
        g = torch.Generator(device=dev).manual_seed(5)
        p = (torch.rand(rows, 2, device=dev, generator=g) - 0.5) * 1e-2
        st = dict(p=p, m=torch.zeros_like(p), v=torch.zeros_like(p), h=p.half(), wp=torch.rand(n_w, device=dev, generator=g) - 0.5)
        st.update(wm=torch.zeros(n_w, device=dev), wv=torch.zeros(n_w, device=dev), wh=st["wp"].half())
        st.update(step=torch.zeros((), device=dev), scale=torch.full((), 65536.0, device=dev), tracker=torch.zeros((), dtype=torch.int32, device=dev),
                  found=torch.zeros((), device=dev), ticket=torch.zeros((), dtype=torch.int32, device=dev))
        return st

    hyper = (1e-2, 0.9, 0.99, 1e-15)
    amp_consts = (2.0, 0.5, 3)  # growth every 3 clean steps: the scale moves during the test

    def arr(ts):
        return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])

    def grads_for(step):
        g = torch.Generator(device=dev).manual_seed(1000 + step)
        gx = (torch.randn(B, 32, device=dev, generator=g) * 3e-2).half()
        gw = (torch.randn(n_w, device=dev, generator=g) * 1e-1).half()
        xs = x
        if step == 2:
            gx[B // 3, 9] = float("inf")
        if step == 4:  # 48 samples at one position, 60000 each on one level: a row overflows as a sum of finite shares
            xs = x.clone()
            xs[B // 2 - 24:B // 2 + 24] = x[B // 2].clone()
            gx[B // 2 - 24:B // 2 + 24, 2 * 11] = 60000.0
        return xs, gx, gw

    # ---- the two-launch path: backward writes the whole fp16 gradient, one Adam launch reads it
    ref = fresh()
    ref_trace = []
    for step in range(8):
        xs, gx, gw = grads_for(step)
        if step == 6:
            ref["scale"].fill_(3000.0)  # not a power of two: the division path of the unscale
        gt = torch.empty(rows, 2, dtype=torch.float16, device=dev)
        check(lib.nerftex_grid_encode_backward_amp(ptr(gx), ptr(xs), None, ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, None, None, 0, 1, F16,
                                                   LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, ptr(ref["found"]), stream()))
        n = (ctypes.c_uint64 * 2)(rows * 2, n_w)
        check(lib.nerftex_adam_mixed_step_amp(2, arr([ref["p"], ref["wp"]]), arr([ref["m"], ref["wm"]]), arr([ref["v"], ref["wv"]]), arr([gt, gw]),
                                              arr([ref["h"], ref["wh"]]), n, 0, ptr(ref["step"]), *hyper, ptr(ref["scale"]), ptr(ref["tracker"]), ptr(ref["found"]),
                                              ptr(ref["ticket"]), *amp_consts, stream()))
        ref_trace.append((float(ref["step"]), float(ref["scale"]), ref["h"].clone(), gt))
    assert [t[0] for t in ref_trace] == [1, 2, 2, 3, 3, 4, 5, 6], "steps 2 (inf) and 4 (row-sum overflow) are skipped"

    # ---- the fused path over double-buffered state
    st = fresh()
    sets = {k: [st[k], torch.full_like(st[k], float("nan"))] for k in ("p", "m", "v", "wp", "wm", "wv")}
    live = torch.zeros((), dtype=torch.int32, device=dev)
    ta = TableAdam()
    for k in range(2):
        ta.param[k], ta.exp_avg[k], ta.exp_avg_sq[k] = sets["p"][k].data_ptr(), sets["m"][k].data_ptr(), sets["v"][k].data_ptr()
    ta.param_half, ta.live, ta.step, ta.grad_scale, ta.found_inf = st["h"].data_ptr(), live.data_ptr(), st["step"].data_ptr(), st["scale"].data_ptr(), st["found"].data_ptr()
    ta.lr, ta.beta1, ta.beta2, ta.eps = hyper
    lives = []
    for step in range(8):
        xs, gx, gw = grads_for(step)
        if step == 6:
            st["scale"].fill_(3000.0)
        gt = torch.full((rows, 2), float("nan"), dtype=torch.float16, device=dev)
        first = ctypes.c_uint32(12345)
        check(lib.nerftex_grid_encode_backward_adam(ptr(gx), ptr(xs), ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, 1, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25,
                                                    ctypes.byref(ta), ctypes.byref(first), stream()))
        f = int(first.value)
        assert f == 0, "every row is updated by the owner of its final gradient (sole tile owners, or the combine kernel for shared tiles)"
        assert torch.isnan(gt).all(), "no gradient row is written"
        n = (ctypes.c_uint64 * 2)(f * 2, n_w)
        cut = lambda t: t[:f]  # noqa: E731
        check(lib.nerftex_adam_mixed_step_amp_db(
            2, arr([cut(sets["p"][0]), sets["wp"][0]]), arr([cut(sets["m"][0]), sets["wm"][0]]), arr([cut(sets["v"][0]), sets["wv"][0]]),
            arr([cut(sets["p"][1]), sets["wp"][1]]), arr([cut(sets["m"][1]), sets["wm"][1]]), arr([cut(sets["v"][1]), sets["wv"][1]]),
            arr([cut(gt), gw]), arr([cut(st["h"]), st["wh"]]), n, 0, ptr(st["step"]), *hyper, ptr(st["scale"]), ptr(st["tracker"]), ptr(st["found"]), ptr(st["ticket"]),
            *amp_consts, ptr(live), ptr(st["h"][f:]), ptr(sets["p"][0][f:]), ptr(sets["p"][1][f:]), (rows - f) * 2, stream()))
        lives.append(int(live.item()))
        assert (float(st["step"]), float(st["scale"])) == ref_trace[step][:2], step
        assert torch.equal(_bits(st["h"]), _bits(ref_trace[step][2])), f"fp16 table after step {step}"
    assert lives == [1, 0, 0, 1, 1, 0, 1, 0], "the state sets flip on applied steps only"
    k = lives[-1]
    for name, rname in (("p", "p"), ("m", "m"), ("v", "v"), ("wp", "wp"), ("wm", "wm"), ("wv", "wv")):
        assert torch.equal(_bits(sets[name][k]), _bits(ref[rname])), name
    assert torch.equal(_bits(st["wh"]), _bits(ref["wh"]))


def test_backward_adam_refuses_what_it_cannot_do(dev, oracle):
    from nerftex_hip import F16, F32, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, TableAdam, lib, ptr, stream

    off_np, rows = oracle.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data)
    B = 4096
    x = torch.rand(B, 3, device=dev)
    gx = torch.zeros(B, 32, dtype=torch.float16, device=dev)
    gt = torch.empty(rows, 2, dtype=torch.float16, device=dev)
    bufs = [torch.zeros(rows, 2, device=dev) for _ in range(6)]
    h = torch.zeros(rows, 2, dtype=torch.float16, device=dev)
    words = torch.zeros(4, device=dev)
    live = torch.zeros((), dtype=torch.int32, device=dev)
    ta = TableAdam()
    for k in range(2):
        ta.param[k], ta.exp_avg[k], ta.exp_avg_sq[k] = bufs[3 * k].data_ptr(), bufs[3 * k + 1].data_ptr(), bufs[3 * k + 2].data_ptr()
    ta.param_half, ta.live, ta.step, ta.grad_scale, ta.found_inf = h.data_ptr(), live.data_ptr(), words[0:].data_ptr(), 0, words[1:].data_ptr()
    ta.lr, ta.beta1, ta.beta2, ta.eps = 1e-2, 0.9, 0.99, 1e-15
    first = ctypes.c_uint32(0)

    def call(dtype=F16, layout=LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, C=2, adam=ta):
        return lib.nerftex_grid_encode_backward_adam(ptr(gx), ptr(x), ptr(off), ptr(gt), B, 3, C, 16, float(np.log2(1.447269)), 16, 0, 1, dtype, layout, 0.0, 1.0,
                                                     ctypes.byref(adam) if adam is not None else None, ctypes.byref(first), stream())

    assert call() == 0
    assert call(dtype=F32) != 0 and b"fp16" in lib.nerftex_last_error()
    assert call(layout=LAYOUT_BLC) != 0
    assert call(C=4) != 0
    assert call(adam=None) != 0
    ta.live = 0
    assert call() != 0 and b"NULL" in lib.nerftex_last_error()
    torch.cuda.synchronize()


def _trainer_pair(dev, fused_table_update, k):
    from ngp_harness import scene
    from ngp_harness.accelerate import accelerate
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    return field, accelerate(r, dt_gamma=1 / 128, steps_per_call=k, fused_table_update=fused_table_update)


def test_fused_table_update_trains_like_the_two_launch_step(dev):
    """accelerate(..., fused_table_update=True) -- the hashed levels updated from the summing kernel's tiles, double-buffered state -- against
    fused_table_update=False (gradient tensor + one streaming Adam launch): the same losses and the same parameters, bit for bit, after
    16 + 4 + 40 steps through `step_group` (priming on full-size buffers, warm-up, replayed graphs, marches ahead), with the loss scale forced
    to overflow inside the replayed part (GradScaler skips that step whole: the tiles updated before the overflow was seen must leave no
    trace) and one more overflow right after it."""
    from ngp_harness import scene

    n, n_pool, k = 4096, 8, 4
    pool = []
    for j in range(n_pool):
        o, d = scene.train_batch(n, seed=500 + j, n_views=2)
        pool.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    gt = torch.rand(n_pool, n, 3, generator=torch.Generator().manual_seed(23)).to(dev)
    po = [torch.stack([pool[c * 4 + i][0] for i in range(4)]).contiguous() for c in range(2)]
    pd = [torch.stack([pool[c * 4 + i][1] for i in range(4)]).contiguous() for c in range(2)]
    pt = [gt[c * 4:(c + 1) * 4].contiguous() for c in range(2)]
    total = 16 + 4 + 40
    out = {}
    for fuse in (False, True):
        field, tr = _trainer_pair(dev, fuse, k)
        assert tr.fused_table_update == fuse
        losses, scales = [], []
        for c in range(total // k):
            if c == 9:
                tr.amp.scale.fill_(2.0 ** 31)  # the next steps overflow until the scale has backed off far enough
            nxt = (po[(c + 1) % 2], pd[(c + 1) % 2]) if c >= 7 and c % 2 == 1 else None
            tr.step_group(po[c % 2], pd[c % 2], pt[c % 2], next_rays=nxt)
            losses.append(tr.loss.clone())
            scales.append(tr.amp.scale.clone())
        torch.cuda.synchronize()
        assert tr._groups is not None, "the grouped graphs were recorded"
        steps = float(tr.opt.step_count)
        tr.sync()
        out[fuse] = (losses, scales, steps, {n_: p.detach().clone() for n_, p in field.named_parameters()},
                     [t.clone() for t in tr.opt.exp_avg + tr.opt.exp_avg_sq], [leaf.detach().clone() for leaf in tr.opt.leaves],
                     {n_: v.clone() for n_, v in field.state_dict().items()})
    a, b = out[False], out[True]
    assert a[2] == b[2] and a[2] < total, "some steps were skipped, the same ones"
    for i, (la, lb) in enumerate(zip(a[0], b[0])):
        assert torch.equal(la, lb), f"loss of call {i}"
    assert all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    for name in a[3]:
        assert torch.equal(a[3][name], b[3][name]), name
        assert torch.equal(a[6][name], b[6][name]), f"state_dict()['{name}']"
    for x, y in zip(a[4] + a[5], b[4] + b[5]):
        assert torch.equal(x, y)


def test_fused_table_update_checkpoint_round_trip(dev, tmp_path):
    """A checkpoint written from the double-buffered optimizer loads into a fresh trainer (and the reverse direction is the ordinary path):
    training continues on the same parameters."""
    from ngp_harness import checkpoint, scene

    n = 4096
    o, d = scene.train_batch(n, seed=77, n_views=2)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    tg = torch.rand(n, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    f1, t1 = _trainer_pair(dev, True, 1)
    for _ in range(5):
        t1.step(ro, rd, tg)
    path = str(tmp_path / "ck.pth")
    checkpoint.save_checkpoint(path, t1.renderer, optimizer=t1.opt, scaler=t1.amp)
    f2, t2 = _trainer_pair(dev, True, 1)
    checkpoint.load_checkpoint(path, t2.renderer, optimizer=t2.opt, scaler=t2.amp)
    f3, t3 = _trainer_pair(dev, False, 1)
    checkpoint.load_checkpoint(path, t3.renderer, optimizer=t3.opt, scaler=t3.amp)
    for t in (t1, t2, t3):
        t.renderer.local_step = 0
        t.renderer.step_counter.zero_()
    for _ in range(3):
        for t in (t1, t2, t3):
            t.step(ro, rd, tg)
    for t in (t1, t2, t3):
        t.sync()
    for (n1, p1), (_, p2), (_, p3) in zip(f1.named_parameters(), f2.named_parameters(), f3.named_parameters()):
        assert torch.equal(p1, p2) and torch.equal(p1, p3), n1


# ------------------------------------------------------------------------------------------------- dead-sample skip (step flags)
def _opaque_case(dev, n_rays=2048, density_scale=400.0):
    """A renderer whose field is dense enough that most samples sit behind the point where their ray's transmittance has underflowed."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-2, 1e-2)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    r.density_scale = density_scale
    o, d = scene.train_batch(n_rays, seed=31, n_views=2)
    return field, r, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)


def test_compositing_backward_flags_exactly_the_steps_that_carry_a_gradient(dev):
    """nerftex_render_tail_forward_live clears, nerftex_composite_tail_backward_live sets: word s != 0 <=> some sample of rows [32 s, 32 s + 32) has a
    non-zero grad_sigma or grad_rgb.  The gradients themselves are the plain launch's (the reference's arithmetic, raymarching.cu:843-870)."""
    from nerftex_hip import check, lib, ptr, stream

    field, r, ro, rd = _opaque_case(dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=300000)
        nears, fars, xyzs, dirs, deltas, rays = marched
        sigmas, rgbs, _ = field(xyzs, dirs)
    sigmas = (sigmas.float() * r.density_scale).contiguous()
    rgbs = rgbs.float().contiguous()
    M, N = sigmas.shape[0], rays.shape[0]
    tgt = torch.rand(N, 3, device=dev)
    per_ray = torch.empty(9, N, device=dev)
    ws, depth, depth_out, image, image_out = per_ray[0], per_ray[1], per_ray[2], per_ray[3:6].view(N, 3), per_ray[6:9].view(N, 3)
    losses = torch.empty(2, device=dev)
    ticket, partial = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1024, device=dev)
    one = torch.ones((), device=dev)
    flags = torch.full(((M + 31) // 32,), 7, dtype=torch.int32, device=dev)
    check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(ws), ptr(depth), ptr(image), stream()))
    check(lib.nerftex_render_tail_forward_live(ptr(ws), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(tgt), 1.0, 1.0, N, ptr(image_out), ptr(depth_out), ptr(partial),
                                               ptr(ticket), ptr(losses), None, losses.data_ptr() + 4, ptr(flags), flags.numel(), stream()))
    assert int(flags.abs().sum()) == 0, "cleared by the render tail"
    g = torch.full((4 * M,), float("nan"), device=dev)
    g_plain = torch.full((4 * M,), float("nan"), device=dev)
    args = (ptr(one), None, 1.0, ptr(image_out), ptr(tgt), 1.0, ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(ws), ptr(image), M, N)
    check(lib.nerftex_composite_tail_backward_live(*args, ptr(g[:M]), ptr(g[M:]), ptr(flags), stream()))
    check(lib.nerftex_composite_tail_backward(*args, ptr(g_plain[:M]), ptr(g_plain[M:]), stream()))
    assert torch.equal(g.view(torch.int32), g_plain.view(torch.int32))
    live = (g[:M] != 0) | (g[M:].view(M, 3) != 0).any(-1)
    pad = (-M) % 32
    want = torch.nn.functional.pad(live, (0, pad)).view(-1, 32).any(-1)
    assert torch.equal(flags != 0, want)
    frac_dead = 1.0 - float(want.float().mean())
    assert 0.3 < frac_dead < 0.999, f"the case must have dead and live steps ({frac_dead:.3f} dead)"


@pytest.mark.parametrize("mlp_dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_field_backward_over_the_live_steps_equals_the_plain_backward(dev, oracle, mlp_dtype):
    """nerftex_field_backward_live against nerftex_field_backward_amp / _bf16 (each followed by nerftex_grid_encode_backward_amp) on gradients that are
    exactly zero on the dead steps: weight gradients, grad_x and the table gradient bit-identical; grad_cin identical on the live steps and
    UNTOUCHED on the dead ones (nobody reads it there)."""
    from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, LAYOUT_LBC, check, lib, ptr, stream

    bf16 = mlp_dtype == torch.bfloat16
    B = 128 * 300
    off_np, rows = oracle.grid_offsets(3, 16, 1.447269, 16, 19, True)
    off = torch.from_numpy(off_np).to(dev)
    check(lib.nerftex_grid_register_offsets(ptr(off), 16, off_np.ctypes.data))
    S = float(np.log2(1.447269))
    g = torch.Generator(device=dev).manual_seed(9)
    table = ((torch.rand(rows, 2, device=dev, generator=g) - 0.5) * 2).half()
    x = torch.rand(B, 3, device=dev, generator=g) * 3.8 - 1.9
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1)
    ws = ((torch.rand(64 * (32 + 64 + 16), device=dev, generator=g) - 0.5) * 0.3).to(mlp_dtype)
    wc = ((torch.rand(64 * (32 + 128 + 16), device=dev, generator=g) - 0.5) * 0.3).to(mlp_dtype)
    feats = torch.empty(16, B, 2, dtype=torch.float16, device=dev)
    dummy = torch.empty(1, dtype=torch.float16, device=dev)
    check(lib.nerftex_grid_encode_forward_affine(ptr(x), ptr(table), ptr(off), ptr(feats), B, 3, 2, 16, S, 16, 0, ptr(dummy), 0, 1, F16, LAYOUT_LBC, 2.0, 0.25, stream()))
    sigma, rgbs = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
    x_rows, h, cin = torch.empty(B, 32, dtype=mlp_dtype, device=dev), torch.empty(B, 16, dtype=mlp_dtype, device=dev), torch.empty(B, 32, dtype=mlp_dtype, device=dev)
    fwd = lib.nerftex_field_forward_bf16 if bf16 else lib.nerftex_field_forward
    check(fwd(ptr(feats), ptr(dirs), ptr(ws), ptr(wc), B, ptr(sigma), ptr(rgbs), ptr(x_rows), ptr(h), ptr(cin), None, stream()))
    # live steps: a random third of the 32-row steps, plus whole stretches of dead ones (a wave's 64-step ballot comes back empty somewhere)
    n_steps = B // 32
    live = torch.rand(n_steps, device=dev, generator=g) < 0.33
    live[200:500] = False
    live[0] = True
    live[-1] = True
    rows_live = live.repeat_interleave(32)
    gs = torch.randn(B, device=dev, generator=g) * 1e-2 * rows_live
    gc = torch.randn(B, 3, device=dev, generator=g) * 1e-2 * rows_live.unsqueeze(-1)
    gs[64] = 0.0  # (a dead SAMPLE inside a live step is just a zero row)
    flags = live.to(torch.int32)

    def run(use_flags):
        grad_cin = torch.full((B, 32), float("nan"), dtype=mlp_dtype, device=dev)
        grad_x = torch.full((B, 32), float("nan"), dtype=torch.float16, device=dev)
        gws, gwc = torch.empty_like(ws), torch.empty_like(wc)
        found = torch.zeros((), device=dev)
        common = (ptr(gs), ptr(gc), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), B, ptr(grad_cin), ptr(grad_x), ptr(gws), ptr(gwc))
        gt = torch.full((rows, 2), float("nan"), dtype=torch.float16, device=dev)
        if use_flags:
            check((lib.nerftex_field_backward_live_bf16 if bf16 else lib.nerftex_field_backward_live)(*common, ptr(flags), ptr(found), stream()))
        else:
            check((lib.nerftex_field_backward_bf16 if bf16 else lib.nerftex_field_backward_amp)(*common, ptr(found), stream()))
        check(lib.nerftex_grid_encode_backward_amp(ptr(grad_x), ptr(x), None, ptr(off), ptr(gt), B, 3, 2, 16, S, 16, 0, None, None, 0, 1, F16,
                                                   LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, 2.0, 0.25, ptr(found), stream()))
        return grad_cin, grad_x, gws, gwc, gt, float(found)

    a, b = run(False), run(True)
    assert a[5] == b[5] == 0.0
    for i, name in ((2, "sigma weight gradient"), (3, "colour weight gradient"), (4, "table gradient")):
        assert torch.equal(_bits(a[i]), _bits(b[i])), name
    assert torch.equal(_bits(a[0][rows_live]), _bits(b[0][rows_live])), "grad_cin on the live steps"
    assert torch.isnan(b[0][~rows_live].float()).all(), "grad_cin: the dead steps' rows are not written"
    assert torch.equal(_bits(a[1][rows_live]), _bits(b[1][rows_live])), "grad_x on the live steps"
    assert float(a[1][~rows_live].float().abs().max()) == 0.0 and float(b[1][~rows_live].float().abs().max()) == 0.0, "grad_x: zeros on the dead steps, both ways"
    assert float(a[4].float().abs().sum()) > 0


def test_skip_dead_samples_trains_like_the_full_backward(dev):
    """accelerate(skip_dead_samples=True) against skip_dead_samples=False on a dense field (most samples dead), replayed graphs included: the same
    losses and parameters bit for bit; and the flags really are mostly zero there."""
    from ngp_harness.accelerate import accelerate

    out = {}
    for skip in (False, True):
        field, r, ro, rd = _opaque_case(dev, n_rays=4096, density_scale=300.0)
        tgt = torch.rand(4096, 3, generator=torch.Generator().manual_seed(8)).to(dev)
        tr = accelerate(r, dt_gamma=1 / 128, skip_dead_samples=skip)
        assert r.skip_dead_samples == skip
        losses = []
        for _ in range(16 + 2 + 14):
            tr.step(ro, rd, tgt)
            losses.append(tr.loss.clone())
        torch.cuda.synchronize()
        assert tr._graphs is not None
        tr.sync()
        out[skip] = (losses, {n_: p.detach().clone() for n_, p in field.named_parameters()})
    for i, (la, lb) in enumerate(zip(out[False][0], out[True][0])):
        assert torch.equal(la, lb), f"loss of step {i}"
    for name in out[False][1]:
        assert torch.equal(out[False][1][name], out[True][1][name]), name


# ------------------------------------------------------------------------------------------------- graphed inference: the loop's step budget
def test_graphed_inference_stops_at_the_step_budget_like_the_reference_loop(dev):
    """nerf/renderer.py:459-483 runs `while step < max_steps: ... step += n_step`.  The graphed loop's kernels derive n_step (F .. 8 F) on the device, so
    the budget has to be kept there too (nerftex_compact_rays_budget_dev): an all-occupied grid and a thin medium -- no ray ever saturates, every ray
    outlives a small max_steps -- must give the image and depth of the reference loop and of the host-launched loop, bit for bit, and the
    device-side step word must end at the same count the host loop's `step` reaches."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-0.3, 0.3)
    field.eval()
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.full((r.cascade, 128 ** 3), 100.0, device=dev))  # every cell occupied
    r.density_scale = 1e-2  # thin: transmittance stays far above the 1e-4 cut
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(5))[0]
    o, d = scene.get_rays(pose, scene.intrinsics(96, 80), 96, 80)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    for max_steps in (24, 40):
        with torch.autocast("cuda", dtype=torch.float16):
            img_ref, dep_ref, n_ref = r.render_infer(ro, rd, dt_gamma=0.0, max_steps=max_steps, slots_per_ray=4)
            img_p, dep_p, _ = r.render_infer_pipelined(ro, rd, dt_gamma=0.0, max_steps=max_steps, slots_per_ray=4, parts=2)
            img_g, dep_g, _ = r.render_infer_graphed(ro, rd, dt_gamma=0.0, max_steps=max_steps, slots_per_ray=4, parts=2, block=2)
        assert float(img_ref.std()) > 1e-4 and float(dep_ref.max()) > 0
        assert torch.equal(img_p, img_ref) and torch.equal(dep_p, dep_ref), f"host-launched loop, max_steps {max_steps}"
        assert torch.equal(img_g, img_ref) and torch.equal(dep_g, dep_ref), f"graphed loop, max_steps {max_steps}"
        for job in r._infer_graphs["jobs"]:
            done = int(job.steps_done.item())
            assert max_steps <= done < max_steps + 8 * 4, (done, max_steps)  # the loop ran until the budget was used up, and not past it


def test_compaction_mirrors_its_count_into_pinned_host_memory(dev):
    """nerftex_compact_rays_budget_mirror_dev = nerftex_compact_rays_budget_dev (nerf/renderer.py:459-483's bookkeeping: the survivors of
    raymarching.cu:1093-1106's compaction, in order) + the survivor count stored by the kernel into pinned host memory: same arrays, same device
    counter, same step word as the plain call, and after an event behind the launch the host word holds the count -- also 0 once the step budget is
    used up, and also for an empty survivor set.  A NULL mirror is refused."""
    from nerftex_hip import check, lib, ptr, stream

    N = 70001  # ragged last workgroup
    gen = torch.Generator(device=dev).manual_seed(3)
    for frac_dead, done0, max_steps in ((0.37, 0, 1024), (0.0, 0, 1024), (1.0, 0, 1024), (0.5, 1024, 1024)):
        t_old = torch.rand(N, device=dev, generator=gen) + 0.1
        t_old[torch.rand(N, device=dev, generator=gen) < frac_dead] = -1.0
        alive_old = torch.randperm(N, device=dev, generator=gen).int()
        n_dev = torch.tensor([N - 5], dtype=torch.int32, device=dev)  # the true count lives on the device (launch sized by a bound)
        outs = []
        for mirror in (False, True):
            ra, rt = torch.full((N,), -7, dtype=torch.int32, device=dev), torch.full((N,), -7.0, device=dev)
            cnt, steps = torch.tensor([123], dtype=torch.int32, device=dev), torch.tensor([done0], dtype=torch.int32, device=dev)
            host = torch.full((2,), -1, dtype=torch.int32).pin_memory()
            if mirror:
                check(lib.nerftex_compact_rays_budget_mirror_dev(N, ptr(n_dev), ptr(ra), ptr(alive_old), ptr(rt), ptr(t_old), ptr(cnt), ptr(steps), max_steps, 4,
                                                                 host.data_ptr() + 4, stream()))
            else:
                check(lib.nerftex_compact_rays_budget_dev(N, ptr(n_dev), ptr(ra), ptr(alive_old), ptr(rt), ptr(t_old), ptr(cnt), ptr(steps), max_steps, 4, stream()))
            ev = torch.cuda.Event()
            ev.record()
            ev.synchronize()
            outs.append((ra, rt, cnt, steps, host.clone()))
        (ra0, rt0, c0, s0, h0), (ra1, rt1, c1, s1, h1) = outs
        assert torch.equal(ra0, ra1) and torch.equal(rt0, rt1) and torch.equal(c0, c1) and torch.equal(s0, s1)
        keep = (t_old[:N - 5] >= 0)
        expect = 0 if done0 >= max_steps else int(keep.sum())
        assert int(c1) == expect and h1.tolist() == [-1, expect] and h0.tolist() == [-1, -1], (frac_dead, done0, h1.tolist(), expect)
        if expect:
            assert torch.equal(ra1[:expect], alive_old[:N - 5][keep])
    rc = lib.nerftex_compact_rays_budget_mirror_dev(N, ptr(n_dev), ptr(ra), ptr(alive_old), ptr(rt), ptr(t_old), ptr(cnt), ptr(steps), 1024, 4, None, stream())
    assert rc != 0


def test_graphed_inference_with_the_mirrored_count_equals_the_copy_node_form(dev):
    """Renderer.render_infer_graphed reads the alive count from the word the compaction kernel mirrors into pinned memory (possibly a LATER block's
    count: a newer upper bound); the form with a 4-byte copy node behind every block is kept for an A/B (_InferGraphPart.COUNT_MIRROR = False).
    Same image, depth and iteration count as that form and as the reference loop (nerf/renderer.py:436-487), bit for bit."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer, _InferGraphPart

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-0.3, 0.3)
    field.eval()
    r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(11))[0]
    o, d = scene.get_rays(pose, scene.intrinsics(200, 160), 200, 160)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    res = {}
    try:
        with torch.autocast("cuda", dtype=torch.float16):
            img_ref, dep_ref, _ = r.render_infer(ro, rd, dt_gamma=1 / 128, slots_per_ray=4)
            for mirror in (True, False):
                _InferGraphPart.COUNT_MIRROR = mirror
                for _ in range(3):  # (the second and third frame replay the graphs over a word that still holds the last frame's final count)
                    img, dep, _ = r.render_infer_graphed(ro, rd, dt_gamma=1 / 128, slots_per_ray=4, parts=3, block=2)
                assert all(j.mirror == mirror for j in r._infer_graphs["jobs"])
                res[mirror] = (img.clone(), dep.clone(), r.last_iters)
    finally:
        _InferGraphPart.COUNT_MIRROR = True
    assert float(img_ref.std()) > 1e-3
    for mirror in (True, False):
        assert torch.equal(res[mirror][0], img_ref) and torch.equal(res[mirror][1], dep_ref), mirror
    assert abs(res[True][2] - res[False][2]) <= 2  # (a newer count can end the loop one block earlier, never later)


# ------------------------------------------------------------------------------------------------- curved field: the forward as one graph (item 8)
def test_curved_field_forward_as_one_graph_equals_the_eager_forward(dev):
    """CurvedField.forward_graphed: neighbour search, projector, lookup, the two FFMLPs and the framework ops between them replayed as one HIP graph --
    the values of forward() (tools/map.py:414-433, 620-641 through the harness), bit for bit, on a second batch too (the graph's static inputs are
    overwritten), and a changed parameter re-records."""
    from ngp_harness.curved import CurvedField

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, p = np.load(os.path.join(golden, "ref_python_curvedfield.npz")), np.load(os.path.join(golden, "ref_python_projector.npz"))
    field = CurvedField(p["vertices"], p["faces"], bound=1.0, h_threshold=float(p["h_threshold"]), vertex_normals=p["vertex_normals"], tbn=p["tbn"])
    gen = torch.Generator().manual_seed(int(g["table_seed"]))
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.rand(field.encoder.embeddings.shape, generator=gen) - 0.5)
        field.sigma_net.weights.copy_(torch.from_numpy(g["w_sigma"]))
        field.color_net.weights.copy_(torch.from_numpy(g["w_color"]))
    field = field.to(dev).eval()
    v = torch.from_numpy(p["vertices"]).float()
    n = 128 * 64
    for k in range(2):
        base = v[torch.randint(0, v.shape[0], (n,), generator=gen)]
        x = (base * (1 + (torch.rand(n, 1, generator=gen) - 0.5) * 0.1)).to(dev)
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=-1).to(dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            want_s, want_c, _ = field(x, d)
            got_s, got_c, _ = field.forward_graphed(x, d)
            assert torch.equal(got_s, want_s) and torch.equal(got_c, want_c), k
        assert float(want_s.abs().sum()) > 0
    graph = field._fwd_graph["graph"]
    with torch.no_grad():
        field.encoder.embeddings.mul_(1.5)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        want_s, _, _ = field(x, d)
        got_s, _, _ = field.forward_graphed(x, d)
    assert field._fwd_graph["graph"] is not graph and torch.equal(got_s, want_s)


def test_curved_field_glue_kernels_equal_the_framework_ops(dev):
    """CurvedField.forward with the three glue launches (nerftex_curved_pack_inputs / _mid_forward / _out_forward) against the same forward on framework ops
    (fused_glue = False: network_curvedfield.py:283-306 + tools/map.py:620-641 op by op): sigma, colour and the parameter gradients, eval and training."""
    from ngp_harness.curved import CurvedField

    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g, p = np.load(os.path.join(golden, "ref_python_curvedfield.npz")), np.load(os.path.join(golden, "ref_python_projector.npz"))
    field = CurvedField(p["vertices"], p["faces"], bound=1.0, h_threshold=float(p["h_threshold"]), vertex_normals=p["vertex_normals"], tbn=p["tbn"])
    gen = torch.Generator().manual_seed(int(g["table_seed"]))
    with torch.no_grad():
        field.encoder.embeddings.copy_(torch.rand(field.encoder.embeddings.shape, generator=gen) - 0.5)
        field.sigma_net.weights.copy_(torch.from_numpy(g["w_sigma"]))
        field.color_net.weights.copy_(torch.from_numpy(g["w_color"]))
    field = field.to(dev)
    x, d = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["dirs"]).to(dev)
    gs = torch.randn(x.shape[0], generator=gen).to(dev) * 1e-2
    gc = torch.randn(x.shape[0], 3, generator=gen).to(dev) * 1e-2
    out = {}
    for mode in ("eval", "train"):
        field.train(mode == "train")
        for fused in (False, True):
            field.fused_glue = fused
            for q in field.parameters():
                q.grad = None
            with torch.autocast("cuda", dtype=torch.float16):
                sigma, color, _ = field(x, d)
                if mode == "train":
                    torch.autograd.backward([sigma, color], [gs.to(sigma.dtype), gc.to(color.dtype)])
            out[(mode, fused)] = (sigma.detach().float(), color.detach().float(), [q.grad.detach().float().clone() for q in field.parameters() if q.grad is not None])
        a, b = out[(mode, False)], out[(mode, True)]
        assert a[0].dtype == b[0].dtype and float(a[0].abs().max()) > 0
        # one half-ulp: a normalisation's sum order, the fp32 exp that trunc_exp's derivative is taken from
        assert float(((a[0] - b[0]).abs() / (a[0].abs() + 1e-3)).max()) <= 2e-3, mode
        assert float((a[1] - b[1]).abs().max()) <= 2e-3, mode
        assert torch.equal(a[0] == 0, b[0] == 0), "the same samples are masked"
        for ga, gb in zip(a[2], b[2]):
            assert float((ga - gb).abs().max()) <= 2e-2 * float(ga.abs().max()) + 1e-7, mode
    assert len(out[("train", True)][2]) == 3


# ------------------------------------------------------------------------------------------------- the compositing of a training step as one launch
@pytest.mark.parametrize("case", ["opaque", "thin", "budget_cut"])
def test_composite_step_equals_the_three_launches(dev, knobs, case):
    """nerftex_composite_step against nerftex_composite_rays_train_forward + nerftex_render_tail_forward_live + nerftex_composite_tail_backward_live
    (raymarching.cu:739-767 / :843-880 around the blend and the MSE): every output, the loss, the gradients and the step flags bit for bit -- for every
    number of 64-sample chunks a wave keeps in registers (rays longer than that reload), with and without a loss scale, with rows no ray covers and
    with rays the sample budget cut off."""
    from nerftex_hip import check, lib, ptr, stream

    field, r, ro, rd = _opaque_case(dev, density_scale={"opaque": 400.0, "thin": 3.0, "budget_cut": 30.0}[case])
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128 if case != "thin" else 0.0, perturb=True, mean_count={"opaque": 300000, "thin": 700000, "budget_cut": 60000}[case])
        nears, fars, xyzs, dirs, deltas, rays = marched
        sigmas, rgbs, _ = field(xyzs, dirs)
    sigmas = (sigmas.float() * r.density_scale).contiguous()
    rgbs = rgbs.float().contiguous()
    M, N = sigmas.shape[0], rays.shape[0]
    counts = rays[:, 2]
    total = int(rays[-1, 1] + rays[-1, 2])
    if case == "budget_cut":
        assert total >= M, "some rays must have been cut off by the budget"
    else:
        assert total < M and int(counts.max()) > 128, "rows no ray covers, and rays longer than two chunks"
    tgt = torch.rand(N, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    one = torch.ones((), device=dev)
    words = (M + 31) // 32
    for scale in (None, torch.full((), 1024.0, device=dev)):
        # the three launches
        per_ray = torch.empty(9, N, device=dev)
        ws, depth, depth_out, image, image_out = per_ray[0], per_ray[1], per_ray[2], per_ray[3:6].view(N, 3), per_ray[6:9].view(N, 3)
        losses = torch.empty(2, device=dev)
        ticket, partial = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1024, device=dev)
        flags = torch.full((words,), 7, dtype=torch.int32, device=dev)
        check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(ws), ptr(depth), ptr(image), stream()))
        check(lib.nerftex_render_tail_forward_live(ptr(ws), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(tgt), 1.0, 0.5, N, ptr(image_out), ptr(depth_out),
                                                   ptr(partial), ptr(ticket), ptr(losses), ptr(scale), losses.data_ptr() + 4, ptr(flags), words, stream()))
        g = torch.full((4 * M,), float("nan"), device=dev)
        check(lib.nerftex_composite_tail_backward_live(ptr(one), ptr(scale), 0.5, ptr(image_out), ptr(tgt), 1.0, ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), ptr(ws),
                                                       ptr(image), M, N, ptr(g[:M]), ptr(g[M:]), ptr(flags), stream()))
        assert 0 < int((flags != 0).sum()) and float(g[:M].abs().max()) > 0
        for keep in (0, 1, 3, 4):
            knobs(composite_keep=keep)
            per_ray2 = torch.full((9, N), float("nan"), device=dev)
            ws2, depth2, depth_out2, image2, image_out2 = per_ray2[0], per_ray2[1], per_ray2[2], per_ray2[3:6].view(N, 3), per_ray2[6:9].view(N, 3)
            losses2 = torch.full((2,), float("nan"), device=dev)
            err = torch.empty(N, device=dev)
            flags2 = torch.zeros(words, dtype=torch.int32, device=dev)
            g2 = torch.full((4 * M,), float("nan"), device=dev)
            check(lib.nerftex_composite_step(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(nears), ptr(fars), ptr(tgt), 1.0, 0.5, ptr(scale), ptr(ws2),
                                             ptr(depth2), ptr(image2), ptr(image_out2), ptr(depth_out2), ptr(err), ptr(losses2), losses2.data_ptr() + 4, ptr(g2[:M]),
                                             ptr(g2[M:]), ptr(flags2), stream()))
            assert torch.equal(_bits(per_ray), _bits(per_ray2)), f"per-ray outputs (keep {keep})"
            assert torch.equal(_bits(losses), _bits(losses2)), f"loss, scaled loss (keep {keep}): {losses.tolist()} {losses2.tolist()}"
            assert torch.equal(_bits(g), _bits(g2)), f"gradients (keep {keep})"
            assert torch.equal(flags != 0, flags2 != 0), f"step flags (keep {keep})"


def test_field_backward_consume_leaves_the_step_flags_zero(dev):
    """nerftex_field_backward_live_consume = nerftex_field_backward_live + the flags zeroed by its last launch + (with a nerftex_step_loss) the loss
    nerftex_composite_step would have formed from the same squared errors with its second launch, bit for bit."""
    import ctypes

    from nerftex_hip import StepLoss, check, lib, ptr, stream

    B = 128 * 64
    g = torch.Generator(device=dev).manual_seed(2)
    ws = ((torch.rand(64 * (32 + 64 + 16), device=dev, generator=g) - 0.5) * 0.3).half()
    wc = ((torch.rand(64 * (32 + 128 + 16), device=dev, generator=g) - 0.5) * 0.3).half()
    feats = ((torch.rand(16, B, 2, device=dev, generator=g) - 0.5)).half()
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1)
    sigma, rgbs = torch.empty(B, device=dev), torch.empty(B, 3, device=dev)
    x_rows, h, cin = (torch.empty(B, 32, dtype=torch.float16, device=dev), torch.empty(B, 16, dtype=torch.float16, device=dev),
                      torch.empty(B, 32, dtype=torch.float16, device=dev))
    check(lib.nerftex_field_forward(ptr(feats), ptr(dirs), ptr(ws), ptr(wc), B, ptr(sigma), ptr(rgbs), ptr(x_rows), ptr(h), ptr(cin), None, stream()))
    live = torch.rand(B // 32, device=dev, generator=g) < 0.5
    rows_live = live.repeat_interleave(32)
    gs = torch.randn(B, device=dev, generator=g) * 1e-2 * rows_live
    gc = torch.randn(B, 3, device=dev, generator=g) * 1e-2 * rows_live.unsqueeze(-1)
    out = []
    for consume in (False, True):
        flags = torch.cat([live.to(torch.int32), torch.full((5,), 9, dtype=torch.int32, device=dev)])  # (words past B / 32 are not this call's)
        grad_cin, grad_x = torch.zeros(B, 32, dtype=torch.float16, device=dev), torch.empty(B, 32, dtype=torch.float16, device=dev)
        gws, gwc = torch.empty_like(ws), torch.empty_like(wc)
        common = (ptr(gs), ptr(gc), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), B, ptr(grad_cin), ptr(grad_x), ptr(gws), ptr(gwc), ptr(flags))
        if consume:
            check(lib.nerftex_field_backward_live_consume(*common, None, None, stream()))
        else:
            check(lib.nerftex_field_backward_live(*common, None, stream()))
        out.append((grad_x, gws, gwc))
        if consume:
            assert int(flags[:B // 32].abs().sum()) == 0 and (flags[B // 32:] == 9).all()
        else:
            assert torch.equal(flags[:B // 32] != 0, live)
    for a, b in zip(*out):
        assert torch.equal(_bits(a), _bits(b))
    # the loss job: a real step's squared errors (ray counts that are and are not multiples of 64 / 256)
    field, r, ro, rd = _opaque_case(dev, n_rays=2048 + 77, density_scale=30.0)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        marched, _ = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=300000)
        nears, fars, xyzs, dirs_, deltas, rays = marched
        sg, rgb, _ = field(xyzs, dirs_)
    sg, rgb = (sg.float() * r.density_scale).contiguous(), rgb.float().contiguous()
    M, N = sg.shape[0], rays.shape[0]
    tgt = torch.rand(N, 3, device=dev, generator=g)
    scale = torch.full((), 512.0, device=dev)
    per_ray, err, grads = torch.empty(9, N, device=dev), torch.empty(N, device=dev), torch.empty(4 * M, device=dev)
    for n_rays in (N, 2048, 300):  # (the job's own ray count: a prefix of the errors)
        want, got = torch.empty(2, device=dev), torch.full((2,), float("nan"), device=dev)
        args = (ptr(sg), ptr(rgb), ptr(deltas), ptr(rays), M, N, ptr(nears), ptr(fars), ptr(tgt), 1.0, 0.25, ptr(scale), ptr(per_ray[0]), ptr(per_ray[1]), ptr(per_ray[3:6]),
                ptr(per_ray[6:9]), ptr(per_ray[2]), ptr(err))
        if n_rays == N:
            check(lib.nerftex_composite_step(*args, ptr(want), want.data_ptr() + 4, ptr(grads[:M]), ptr(grads[M:]), None, stream()))
        check(lib.nerftex_composite_step(*args, None, None, ptr(grads[:M]), ptr(grads[M:]), None, stream()))  # loss NULL: left to the job
        job = StepLoss(ptr(err), n_rays, 0.25, ptr(scale), ptr(got), got.data_ptr() + 4)
        flags = live.to(torch.int32)
        check(lib.nerftex_field_backward_live_consume(*common[:-1], ptr(flags), ctypes.byref(job), None, stream()))
        if n_rays == N:
            assert torch.equal(_bits(want), _bits(got)), (want.tolist(), got.tolist())
        else:
            ref = float(err[:n_rays].double().sum() / (n_rays * 3) * 0.25)
            assert abs(float(got[0]) - ref) <= 1e-5 * ref and float(got[1]) == float(got[0]) * 512.0
        assert torch.equal(_bits(out[0][1]), _bits(gws)) and int(flags.abs().sum()) == 0
    bad = StepLoss(ptr(err), 0, 1.0, None, ptr(got), None)
    assert lib.nerftex_field_backward_live_consume(*common[:-1], ptr(flags), ctypes.byref(bad), None, stream()) != 0
    # the same call in two: the backward kernels now, the trailer (reduction, flags, loss) as a launch of its own later -- same bits; an empty trailer is refused
    from nerftex_hip import StepTrailer

    got2, flags = torch.full((2,), float("nan"), device=dev), live.to(torch.int32)
    gws2, gwc2 = torch.full_like(ws, float("nan")), torch.full_like(wc, float("nan"))
    job = StepLoss(ptr(err), 300, 0.25, ptr(scale), ptr(got2), got2.data_ptr() + 4)
    trailer = StepTrailer()
    assert lib.nerftex_step_trailer_run(ctypes.byref(trailer), stream()) != 0, "empty"
    check(lib.nerftex_field_backward_live_deferred(*common[:-3], ptr(gws2), ptr(gwc2), ptr(flags), ctypes.byref(job), None, ctypes.byref(trailer), stream()))
    torch.cuda.synchronize()
    assert torch.isnan(gws2.float()).all() and torch.equal(flags != 0, live), "nothing of the trailer has run yet"
    check(lib.nerftex_step_trailer_run(ctypes.byref(trailer), stream()))
    assert torch.equal(_bits(out[0][1]), _bits(gws2)) and torch.equal(_bits(out[0][2]), _bits(gwc2)) and int(flags.abs().sum()) == 0
    assert torch.equal(_bits(got), _bits(got2)), "the loss of the last job above (300 rays)"


def test_fused_composite_step_trains_like_the_three_launch_step(dev):
    """accelerate(fused_composite_step=True) against False, replayed graphs included, on a field dense enough to have dead steps: the same losses and
    parameters bit for bit, and the flag buffer is clean between steps -- with the field backward's trailer (weight-gradient reduction, flags, loss) on
    the first workgroups of the hash-grid backward's fill launch (nerftex_field_backward_live_deferred -> nerftex_grid_encode_backward_adam_trailer, the
    default) and as a launch of its own (fused.STEP_TRAILER = False)."""
    import nerftex_hip
    from ngp_harness import fused
    from ngp_harness.accelerate import accelerate

    out = {}
    try:
        for form in ("three_launches", "one_launch", "one_launch_own_trailer_launch"):
            fused.STEP_TRAILER = form != "one_launch_own_trailer_launch"
            fused_step = form != "three_launches"
            field, r, ro, rd = _opaque_case(dev, n_rays=4096, density_scale=300.0)
            tgt = torch.rand(4096, 3, generator=torch.Generator().manual_seed(8)).to(dev)
            tr = accelerate(r, dt_gamma=1 / 128, fused_composite_step=fused_step)
            assert (r.root_one is not None) == fused_step and r.skip_dead_samples and tr.fused_table_update
            losses = []
            for i in range(16 + 2 + 14):
                if i == 17:
                    nerftex_hip.kernel_profile(reset=True)
                    nerftex_hip.kernel_profile(True)
                tr.step(ro, rd, tgt)
                if i == 17:
                    torch.cuda.synchronize()
                    names = set(nerftex_hip.kernel_profile())
                    nerftex_hip.kernel_profile(False)
                    assert ("ffmlp_wgrad_reduce2_kernel" in names) == (form != "one_launch"), (form, sorted(names))
                losses.append(tr.loss.clone())
            torch.cuda.synchronize()
            assert tr._graphs is not None
            if fused_step:
                assert int(r._live_words.abs().sum()) == 0, "consumed by the field's backward"
            tr.sync()
            out[form] = (losses, {n_: p.detach().clone() for n_, p in field.named_parameters()})
    finally:
        fused.STEP_TRAILER = True
    for form in ("one_launch", "one_launch_own_trailer_launch"):
        for i, (la, lb) in enumerate(zip(out["three_launches"][0], out[form][0])):
            assert torch.equal(la, lb), f"{form}: loss of step {i}"
        for name in out["three_launches"][1]:
            assert torch.equal(out["three_launches"][1][name], out[form][1][name]), (form, name)
