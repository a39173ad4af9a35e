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
    img, dep, loss = fused.render_tail(ws2, depth, im2, nears, fars, target, bg, mul)
    loss.backward(gl)
    assert torch.equal(img, img_ref.detach()) and torch.equal(dep, depth_ref)
    assert abs(loss.item() - loss_ref.item()) <= 2e-6 * abs(loss_ref.item())  # a mean: summation order
    assert torch.equal(im2.grad, im1.grad)
    torch.testing.assert_close(ws2.grad, ws1.grad, rtol=1e-5, atol=1e-7)  # 3-term sum with cancellation: order of the adds
    # the ticket is back at zero: a second call gives the same loss
    _, _, loss2 = fused.render_tail(ws, depth, image, nears, fars, target, bg, mul)
    assert loss2.item() == loss.item()


def test_table_adam_optimizer_trains_like_torch_adam(dev, monkeypatch):
    """A small hash grid trained for a few steps under autocast + GradScaler: TableAdam (fp16 leaf, fp16 gradient consumed as produced)
    against torch.optim.Adam(fused=True) on the fp32 parameter.  With the encoder backward on its order-independent path (the
    large-batch one, forced here for 4096 points) both see identical gradients and the tables stay identical."""
    monkeypatch.setenv("NERFTEX_GRID_BWD", "owner")
    from gridencoder import GridEncoder
    from ngp_harness.optim import TableAdam

    def run(fused):
        torch.manual_seed(3)
        enc = GridEncoder(input_dim=3, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=12, desired_resolution=128).to(dev)
        lin = torch.nn.Linear(8, 3, bias=False).to(dev)
        if fused:
            opt = TableAdam(enc, lin.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
        else:
            opt = torch.optim.Adam(list(enc.parameters()) + list(lin.parameters()), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
        scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 30)  # the first step overflows: the skip path is exercised too
        x = torch.rand(4096, 3, device=dev) * 2 - 1
        y = torch.rand(4096, 3, device=dev)
        losses = []
        for _ in range(8):
            if fused:
                for t in opt.trainable():
                    t.grad = None
            else:
                opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = torch.nn.functional.mse_loss(lin(enc(x, bound=1)).float(), y)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(loss.item())
        return enc.embeddings.detach().clone(), lin.weight.detach().clone(), losses, scaler.get_scale()

    e1, w1, l1, s1 = run(False)
    e2, w2, l2, s2 = run(True)
    assert s1 == s2 and s1 < 2.0 ** 30
    assert (e1 != 0).any() and not torch.equal(e1, torch.zeros_like(e1))
    assert torch.equal(e1, e2)
    torch.testing.assert_close(w1, w2, rtol=1e-5, atol=1e-7)
    assert l1 == pytest.approx(l2, rel=1e-5)
