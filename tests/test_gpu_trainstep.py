"""The fused ends of the training step (csrc/trainstep.hip) against the framework ops they replace: torch's fused Adam under a
GradScaler for the hash table, and the background blend / depth normalisation / MSE of the training render."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [8 * 4096, 100003])
@pytest.mark.parametrize("scaled", [False, True])
def test_table_adam_matches_torch_fused_adam(dev, n, scaled):
    from nerftex_hip import check, lib, ptr, stream

    torch.manual_seed(0)
    p0 = (torch.rand(n, device=dev) * 2 - 1) * 1e-4
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
    scale = torch.full((), 1024.0 if scaled else 1.0, device=dev)
    found = torch.zeros((), device=dev)

    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    p16 = torch.empty(n, dtype=torch.half, device=dev)
    step = torch.zeros((), device=dev)
    for it in range(6):
        g16 = (torch.randn(n, device=dev) * (1e-3 * (it + 1)) * scale).half()
        g16[::7] = 0  # untouched table rows: zero gradient, moments still decay
        ref.grad = g16.float()
        if scaled:
            opt.grad_scale, opt.found_inf = scale, found
        opt.step()
        step += 1
        check(lib.nerftex_table_adam_step(ptr(p), ptr(m), ptr(v), ptr(g16), ptr(p16), n, ptr(step), 1e-2, 0.9, 0.99, 1e-15,
                                          ptr(scale) if scaled else None, ptr(found) if scaled else None, stream()))
        st = opt.state[ref]
        # same operations in the same precision as the framework's kernel, down to the fused multiply-adds: identical bits
        assert torch.equal(st["exp_avg"], m) and torch.equal(st["exp_avg_sq"], v), it
        assert torch.equal(ref.data, p), it
        assert torch.equal(p16, p.half())


def test_table_adam_skips_on_found_inf(dev):
    from nerftex_hip import check, lib, ptr, stream

    n = 4096
    p = torch.randn(n, device=dev)
    m, v = torch.rand(n, device=dev), torch.rand(n, device=dev)
    p16 = p.half()
    keep = [t.clone() for t in (p, m, v, p16)]
    g16 = torch.full((n,), float("inf"), dtype=torch.half, device=dev)
    step, scale, found = torch.ones((), device=dev), torch.full((), 65536.0, device=dev), torch.ones((), device=dev)
    check(lib.nerftex_table_adam_step(ptr(p), ptr(m), ptr(v), ptr(g16), ptr(p16), n, ptr(step), 1e-2, 0.9, 0.99, 1e-15, ptr(scale), ptr(found),
                                      stream()))
    for a, b in zip(keep, (p, m, v, p16)):
        assert torch.equal(a, b)


def test_table_adam_rejects_misaligned(dev):
    from nerftex_hip import lib, ptr, stream

    n = 64
    buf = torch.zeros(n + 1, device=dev)
    z, h = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.half, device=dev)
    step = torch.ones((), device=dev)
    rc = lib.nerftex_table_adam_step(ptr(buf[1:]), ptr(z), ptr(z.clone()), ptr(h), ptr(h.clone()), n, ptr(step), 1e-2, 0.9, 0.99, 1e-15, None, None,
                                     stream())
    assert rc != 0


@pytest.mark.parametrize("N", [1, 257, 8192, 70001])
def test_render_tail_matches_framework_ops(dev, N):
    from ngp_harness import fused

    torch.manual_seed(N)
    ws = torch.rand(N, device=dev)
    depth = torch.rand(N, device=dev) * 3
    image = torch.rand(N, 3, device=dev)
    nears = torch.rand(N, device=dev) + 0.2
    fars = nears + torch.rand(N, device=dev) * 3 + 0.1
    target = torch.rand(N, 3, device=dev)
    bg, mul = 1.0, 0.5
    gl = torch.full((), 128.0, device=dev)

    ws1, im1 = ws.clone().requires_grad_(True), image.clone().requires_grad_(True)
    img_ref = im1 + (1 - ws1).unsqueeze(-1) * bg
    depth_ref = torch.clamp(depth - nears, min=0) / (fars - nears)
    loss_ref = torch.nn.functional.mse_loss(img_ref, target) * mul
    loss_ref.backward(gl)

    ws2, im2 = ws.clone().requires_grad_(True), image.clone().requires_grad_(True)
    img, dep, loss, scaled = fused.render_tail(ws2, depth, im2, nears, fars, target, bg, mul)
    assert scaled.item() == loss.item()
    scaled.backward(gl)
    assert torch.equal(img, img_ref.detach()) and torch.equal(dep, depth_ref)
    assert abs(loss.item() - loss_ref.item()) <= 2e-6 * abs(loss_ref.item())  # a mean: summation order
    assert torch.equal(im2.grad, im1.grad)
    torch.testing.assert_close(ws2.grad, ws1.grad, rtol=1e-5, atol=1e-7)  # 3-term sum with cancellation: order of the adds
    # the ticket is back at zero: a second call gives the same loss
    _, _, loss2, _ = fused.render_tail(ws, depth, image, nears, fars, target, bg, mul)
    assert loss2.item() == loss.item()
    # with a loss scaler's device scalar: the scaled loss and the gradient both carry it
    ws3, im3 = ws.clone().requires_grad_(True), image.clone().requires_grad_(True)
    scale = torch.full((), 128.0, device=dev)
    _, _, loss3, scaled3 = fused.render_tail(ws3, depth, im3, nears, fars, target, bg, mul, scale)
    assert loss3.item() == loss.item() and scaled3.item() == loss.item() * 128.0
    scaled3.backward(torch.ones((), device=dev))
    assert torch.equal(im3.grad, im2.grad) and torch.equal(ws3.grad, ws2.grad)


@pytest.mark.parametrize("N", [3, 1000, 8192])
def test_composite_tail_equals_compositing_then_render_tail(dev, N):
    """fused.composite_tail (compositing + background blend + depth normalisation + MSE as one autograd node whose backward is ONE
    launch, nerftex_composite_tail_backward) against raymarching.composite_rays_train followed by fused.render_tail: images, depths,
    loss and BOTH gradients identical, bit for bit.  Ragged rays, an empty ray, a ray past the buffer's end."""
    import raymarching
    from ngp_harness import fused

    g = torch.Generator(device="cpu").manual_seed(N)
    counts = torch.randint(0, 150, (N,), generator=g)
    counts[0] = 0
    offsets = torch.cumsum(counts, 0) - counts
    M = int(counts.sum()) + 8
    if N > 2:
        counts[-1] = counts[-1] + 9  # runs past the end of the buffers: dropped (raymarching.cu:727-733)
    rays = torch.stack([torch.arange(N), offsets, counts], dim=1).to(torch.int32).to(dev)
    sigmas = (torch.rand(M, generator=g) * 30).to(dev)
    rgbs = torch.rand(M, 3, generator=g).to(dev)
    deltas = torch.stack([torch.rand(M, generator=g) * 0.02 + 0.003, torch.rand(M, generator=g) * 0.03 + 0.003], dim=1).to(dev)
    nears = (torch.rand(N, generator=g) + 0.2).to(dev)
    fars = nears + (torch.rand(N, generator=g) * 3 + 0.1).to(dev)
    target = torch.rand(N, 3, generator=g).to(dev)
    scale = torch.full((), 1024.0, device=dev)
    bg, mul = 1.0, 0.5

    s1, c1 = sigmas.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
    ws, depth, image = raymarching.composite_rays_train(s1, c1, deltas, rays)
    img1, dep1, loss1, scaled1 = fused.render_tail(ws, depth, image, nears, fars, target, bg, mul, scale)
    scaled1.backward()
    s2, c2 = sigmas.clone().requires_grad_(True), rgbs.clone().requires_grad_(True)
    img2, dep2, loss2, scaled2 = fused.composite_tail(s2, c2, deltas, rays, nears, fars, target, bg, mul, scale)
    junk = torch.full((4 * M,), float("nan"), device=dev)  # the one-launch backward gets an UNINITIALISED gradient buffer (most likely this block,
    del junk                                               # recycled by the caching allocator) and must zero the rows no ray covers itself
    scaled2.backward()
    assert torch.equal(img2, img1) and torch.equal(dep2, dep1)
    assert loss2.item() == loss1.item() and scaled2.item() == loss2.item() * 1024.0
    assert torch.equal(s2.grad, s1.grad) and torch.equal(c2.grad, c1.grad)
    assert float(s1.grad.abs().max()) > 0
    _, _, loss3, _ = fused.composite_tail(sigmas, rgbs, deltas, rays, nears, fars, target, bg, mul, scale)  # the ticket is back at zero
    assert loss3.item() == loss2.item()


@pytest.mark.parametrize("amp", ["gradscaler", "fused"])
def test_half_leaf_adam_trains_like_torch_adam(dev, knobs, amp):
    """A small hash grid + FFMLP trained for a few steps under autocast with loss scaling: HalfLeafAdam (fp16 leaves, fp16 gradients
    consumed as produced; driven by torch's GradScaler or by FusedAmp) against torch.optim.Adam(fused=True) + GradScaler on the fp32
    parameters.  With the encoder backward on its order-independent path (the large-batch one, forced here for 4096 points) all see
    identical gradients, so the tables, the weights and the loss scale stay identical -- through an overflowing first step, the
    skips and the scale growth (growth interval 3 here)."""
    from ffmlp import FFMLP
    from gridencoder import GridEncoder
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    knobs(grid_bwd=2)
    gs = dict(init_scale=2.0 ** 30, growth_interval=3)

    def run(mode):
        torch.manual_seed(3)
        enc = GridEncoder(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=128).to(dev)
        net = FFMLP(input_dim=16, output_dim=3, hidden_dim=64, num_layers=2).to(dev)
        enc.train(), net.train()
        fused_amp = None
        if mode == "torch":
            opt = torch.optim.Adam(list(enc.parameters()) + list(net.parameters()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
        else:
            opt = HalfLeafAdam([(enc, "embeddings"), (net, "weights")], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
            if mode == "fused":
                fused_amp = FusedAmp(opt, **gs)
        scaler = None if fused_amp else torch.amp.GradScaler("cuda", **gs)
        x = torch.rand(4096, 3, device=dev) * 2 - 1
        y = torch.rand(4096, 3, device=dev)
        losses, scales = [], []
        for _ in range(10):
            if mode == "torch":
                opt.zero_grad(set_to_none=True)
            else:
                for t in opt.trainable():
                    t.grad = None
            with torch.autocast("cuda", dtype=torch.float16):
                loss = torch.nn.functional.mse_loss(net(enc(x, bound=1)).float(), y)
            if fused_amp:
                fused_amp.scale_loss(loss).backward()
                fused_amp.step()
                scales.append(fused_amp.get_scale())
            else:
                scaler.scale(loss).backward()
                scaler.step(opt)
                scaler.update()
                scales.append(scaler.get_scale())
            losses.append(loss.item())
        return enc.embeddings.detach().clone(), net.weights.detach().clone(), losses, scales

    e1, w1, l1, s1 = run("torch")
    e2, w2, l2, s2 = run(amp)
    assert s1 == s2 and s1[0] < 2.0 ** 30 and max(s1[1:]) > min(s1)  # backed off first, grew later
    assert torch.equal(e1, e2) and torch.equal(w1, w2)
    assert l1 == l2


def test_amp_check_and_update_follow_gradscaler_rules(dev):
    import ctypes

    from nerftex_hip import check, lib, ptr, stream

    g = [torch.randn(4096 * 8 + 3, device=dev).half(), torch.randn(64, device=dev).half()]
    found = torch.zeros((), device=dev)
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])  # noqa: E731
    n = lambda ts: (ctypes.c_uint64 * len(ts))(*[t.numel() for t in ts])  # noqa: E731
    check(lib.nerftex_amp_check_half(2, arr(g), n(g), ptr(found), stream()))
    assert found.item() == 0.0
    for bad, where in ((float("inf"), (0, 4096 * 8 + 2)), (float("nan"), (1, 5)), (-float("inf"), (0, 777))):
        found.zero_()
        g2 = [t.clone() for t in g]
        g2[where[0]][where[1]] = bad
        check(lib.nerftex_amp_check_half(2, arr(g2), n(g2), ptr(found), stream()))
        assert found.item() == 1.0, (bad, where)
    g2 = [t.clone() for t in g]
    g2[0][11] = 65504.0  # the largest finite half is not an overflow
    found.zero_()
    check(lib.nerftex_amp_check_half(2, arr(g2), n(g2), ptr(found), stream()))
    assert found.item() == 0.0

    scale = torch.full((), 65536.0, device=dev)
    tracker = torch.zeros((), dtype=torch.int32, device=dev)
    step = torch.zeros((), device=dev)
    ref = torch.amp.GradScaler("cuda", init_scale=65536.0, growth_interval=4)
    ref._lazy_init_scale_growth_tracker(dev)
    pattern = [0, 0, 1, 0, 0, 0, 0, 1, 1, 0, 0, 0, 0, 0]
    good = 0
    for f in pattern:
        found.fill_(float(f))
        check(lib.nerftex_amp_update(ptr(scale), ptr(tracker), ptr(found), ptr(step), 2.0, 0.5, 4, stream()))
        torch._amp_update_scale_(ref._scale, ref._growth_tracker, torch.full((), float(f), device=dev), 2.0, 0.5, 4)
        good += 1 - f
        assert found.item() == 0.0
        assert scale.item() == ref._scale.item() and tracker.item() == ref._growth_tracker.item()
        assert step.item() == good


def test_half_leaf_adam_checkpoint_resume_is_bit_identical(dev, knobs):
    """optimizer.state_dict() / load_state_dict() of HalfLeafAdam + FusedAmp (what the reference trainer saves as 'optimizer' and
    'scaler', nerf/utils.py:1505-1507): 6 steps == 3 steps, save, fresh objects, load (masters through load_state_dict on the modules,
    moments / step / scale through the optimizer and scaler), 3 more steps.  The state has torch.optim.Adam's layout and loads into it."""
    from ffmlp import FFMLP
    from gridencoder import GridEncoder
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    knobs(grid_bwd=2)
    torch.manual_seed(5)
    x = torch.rand(4096, 3, device=dev) * 2 - 1
    y = torch.rand(4096, 3, device=dev)

    def make():
        torch.manual_seed(3)
        enc = GridEncoder(input_dim=3, num_levels=8, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=128).to(dev)
        net = FFMLP(input_dim=16, output_dim=3, hidden_dim=64, num_layers=2).to(dev)
        enc.train(), net.train()
        opt = HalfLeafAdam([(enc, "embeddings"), (net, "weights")], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        return enc, net, opt, FusedAmp(opt, init_scale=2.0 ** 20, growth_interval=2)

    def steps(enc, net, opt, amp, n):
        for _ in range(n):
            for t in opt.trainable():
                t.grad = None
            with torch.autocast("cuda", dtype=torch.float16):
                loss = torch.nn.functional.mse_loss(net(enc(x, bound=1)).float(), y)
            amp.scale_loss(loss).backward()
            amp.step()

    a = make()
    steps(*a, 6)
    b = make()
    steps(*b, 3)
    saved = {"enc": b[0].state_dict(), "net": b[1].state_dict(), "optimizer": b[2].state_dict(), "scaler": b[3].state_dict()}
    assert sorted(saved["optimizer"]["state"][0]) == ["exp_avg", "exp_avg_sq", "step"] and float(saved["optimizer"]["state"][0]["step"]) > 0
    c = make()
    c[0].load_state_dict(saved["enc"])
    c[1].load_state_dict(saved["net"])
    c[2].load_state_dict(saved["optimizer"])  # also re-derives the fp16 leaves the kernels read from the loaded masters
    c[3].load_state_dict(saved["scaler"])
    assert torch.equal(c[2].leaves[0], c[0].embeddings.detach().half())
    steps(*c, 3)
    assert torch.equal(a[0].embeddings, c[0].embeddings) and torch.equal(a[1].weights, c[1].weights)
    assert a[3].get_scale() == c[3].get_scale() and torch.equal(a[2].step_count, c[2].step_count)
    ref = torch.optim.Adam([b[0].embeddings, b[1].weights], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    ref.load_state_dict(saved["optimizer"])
    assert torch.equal(ref.state[b[0].embeddings]["exp_avg"], saved["optimizer"]["state"][0]["exp_avg"])
    # ... and the reference's optimizer can STEP on it: Adam.step() reads weight_decay / amsgrad / maximize / ... from the loaded param_group
    # (Adam.__setstate__ fills in every default but weight_decay)
    assert {"weight_decay", "amsgrad", "maximize", "foreach", "capturable", "differentiable", "fused"} <= set(saved["optimizer"]["param_groups"][0])
    for p in (b[0].embeddings, b[1].weights):
        p.grad = torch.zeros_like(p)
    before = float(ref.state[b[0].embeddings]["step"])
    ref.step()
    assert float(ref.state[b[0].embeddings]["step"]) == before + 1
    bad = {"state": saved["optimizer"]["state"], "param_groups": [dict(saved["optimizer"]["param_groups"][0], weight_decay=1e-2)]}
    with pytest.raises(ValueError, match="weight_decay"):
        c[2].load_state_dict(bad)
