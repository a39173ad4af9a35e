"""FFMLP on MFMA vs the oracle (exact-accumulation restatement of the reference's data flow).

fp16 storage on both sides; the HIP kernels accumulate in fp32 on the matrix cores, the oracle in double, the
reference in fp16 -- so parity is a tolerance: a couple of half ulps on activations / outputs, ~1e-3 relative on
weight gradients (sums over the batch).  Tolerances are stated per assert.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import nerftex_hip  # noqa: F401

    return torch.device("cuda:0")


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


CASES = [
    # in, hidden, num_layers, act, B
    (32, 64, 2, 0, 1024),   # sigma net of the ngp field
    (32, 64, 3, 0, 1152),   # colour net
    (16, 16, 2, 0, 256),
    (48, 32, 2, 3, 384),    # sigmoid, in % 32 == 16
    (64, 128, 2, 0, 256),
    (32, 64, 4, 4, 256),    # squareplus, deeper
    (32, 256, 2, 0, 128),
    (32, 64, 2, 6, 128),    # none
    (32, 32, 3, 5, 256),    # softplus
    (16, 64, 2, 1, 128),    # exponential
    # round 5: networks whose weight fragments exceed one CU's LDS -- one layer's fragments staged at a time (the reference's width dispatch takes
    # any num_layers, ffmlp/src/ffmlp.cu:652-658, because its threadblock_layer reads each layer's weights from global memory, :47-129)
    (32, 256, 3, 0, 256),   # 280 KB of fragments
    (64, 128, 6, 0, 256),   # 180 KB
]


def _setup(case, seed):
    IN, H, NL, act, B = case
    rng = np.random.default_rng(seed)
    P = H * (IN + H * (NL - 1) + 16)
    bound = np.sqrt(3.0 / H)
    w = rng.uniform(-bound, bound, size=P).astype(np.float16)
    scale = 0.05 if act in (1,) else 1.0  # keep exp() activations in range
    x = (rng.uniform(-1, 1, size=(B, IN)) * scale).astype(np.float16)
    return IN, H, NL, act, B, w, x


def _hip_forward(dev, IN, H, NL, act, B, w, x, inference):
    from nerftex_hip import check, lib, ptr, stream

    xt, wt = t(x, dev), t(w, dev)
    out = torch.full((B, 16), 9.0, dtype=torch.float16, device=dev)
    fb = torch.full((NL, B, H), 9.0, dtype=torch.float16, device=dev)
    if inference:
        check(lib.nerftex_ffmlp_inference(ptr(xt), ptr(wt), B, IN, 16, H, NL, act, 6, None, ptr(out), stream()))
    else:
        check(lib.nerftex_ffmlp_forward(ptr(xt), ptr(wt), B, IN, 16, H, NL, act, 6, ptr(fb), ptr(out), stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy(), fb.cpu().numpy()


def _close_half(got, want, ulps=2.0, floor=1e-3):
    got = got.astype(np.float32)
    want = want.astype(np.float32)
    tol = ulps * 2.0 ** -10 * np.maximum(np.abs(want), floor)  # half has a 10-bit mantissa
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} beyond {ulps} half-ulps; worst {np.abs(got - want).max()} at |want| {np.abs(want)[bad].max() if bad.any() else 0}"


@pytest.mark.parametrize("case", CASES, ids=[f"in{c[0]}_h{c[1]}_L{c[2]}_act{c[3]}" for c in CASES])
def test_ffmlp_forward_and_inference(oracle, dev, case):
    IN, H, NL, act, B, w, x = _setup(case, 31)
    want_out, want_fb = oracle.ffmlp_forward(x, w, IN, 16, H, NL, act, 6)
    got_out, got_fb = _hip_forward(dev, IN, H, NL, act, B, w, x, False)
    # layer by layer: later layers inherit 1-ulp differences of earlier ones through the next matmul
    _close_half(got_fb[0], want_fb[0], ulps=1.01)
    for l in range(1, NL):
        _close_half(got_fb[l], want_fb[l], ulps=4.0, floor=2e-2)
    _close_half(got_out, want_out, ulps=6.0, floor=5e-2)
    inf_out, _ = _hip_forward(dev, IN, H, NL, act, B, w, x, True)
    assert np.array_equal(inf_out, got_out), "inference kernel must equal the training forward bit for bit"


@pytest.mark.parametrize("mode", ["fused", "split"])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[3], CASES[4], CASES[5], CASES[10], CASES[11]], ids=lambda c: f"in{c[0]}_h{c[1]}_L{c[2]}_act{c[3]}")
def test_ffmlp_backward(oracle, dev, case, mode, knobs):
    """mode "fused" (default): activation + weight gradients in one kernel, backward_buffer untouched (hidden 64, 2-4 layers,
    input <= 64; other shapes fall through to the split kernels).  mode "split": dgrad kernel -> backward_buffer -> wgrad kernel."""
    from nerftex_hip import check, lib, ptr, stream

    knobs(ffmlp_bwd_split=int(mode == "split"))
    IN, H, NL, act, B, w, x = _setup(case, 32)
    fused = mode == "fused" and H == 64 and 2 <= NL <= 4 and IN <= 64
    _, fb = oracle.ffmlp_forward(x, w, IN, 16, H, NL, act, 6)  # same forward activations on both sides
    rng = np.random.default_rng(33)
    grad = (rng.standard_normal((B, 16)) * 1e-2).astype(np.float16)
    want_gw, want_gi, want_bb = oracle.ffmlp_backward(grad, x, w, fb, IN, 16, H, NL, act, True)

    gt, xt, wt, ft = t(grad, dev), t(x, dev), t(w, dev), t(fb, dev)
    bb = torch.zeros(NL, B, H, dtype=torch.float16, device=dev)
    gi = torch.zeros(B, IN, dtype=torch.float16, device=dev)
    gw = torch.zeros_like(wt)
    check(lib.nerftex_ffmlp_backward(ptr(gt), ptr(xt), ptr(wt), ptr(ft), B, IN, 16, H, NL, act, 6, 1, ptr(bb), ptr(gi), ptr(gw), stream()))
    torch.cuda.synchronize()
    bb, gi, gw = bb.cpu().numpy(), gi.cpu().numpy(), gw.cpu().numpy()
    gscale = float(np.abs(want_bb.astype(np.float32)).max())
    if fused:
        assert not bb.any(), "the fused backward must not touch backward_buffer"
        bb = want_bb  # the isolated dW0 check below then runs against the oracle's dPre
    else:
        _close_half(bb[0], want_bb[0], ulps=1.5, floor=1e-3 * gscale)
        for j in range(1, NL):
            _close_half(bb[j], want_bb[j], ulps=6.0, floor=2e-2 * gscale)
    _close_half(gi, want_gi, ulps=8.0, floor=0.1 * float(np.abs(want_gi.astype(np.float32)).max()))
    # weight gradients: the oracle uses ITS OWN bb; feed differences are <= a few half-ulps per element and average out
    wscale = float(np.abs(want_gw.astype(np.float32)).max())
    assert wscale > 0
    np.testing.assert_allclose(gw.astype(np.float32), want_gw.astype(np.float32), rtol=0, atol=4e-3 * wscale)
    # exact check of the wgrad kernel in isolation: recompute from the HIP bb in float64
    P0 = H * IN
    dW0 = bb[NL - 1].astype(np.float64).T @ x.astype(np.float64)
    np.testing.assert_allclose(gw[:P0].astype(np.float64).reshape(H, IN), dW0, rtol=2e-3, atol=(4e-3 if fused else 2e-3) * np.abs(dW0).max())
    dWo = grad.astype(np.float64).T @ fb[NL - 1].astype(np.float64)
    np.testing.assert_allclose(gw[-16 * H:].astype(np.float64).reshape(16, H), dWo, rtol=2e-3, atol=2e-3 * np.abs(dWo).max())

    # without grad_inputs: same weight gradients, grad_inputs untouched
    gi2 = torch.full((B, IN), 5.0, dtype=torch.float16, device=dev)
    gw2 = torch.zeros_like(wt)
    bb2 = torch.zeros(NL, B, H, dtype=torch.float16, device=dev)
    check(lib.nerftex_ffmlp_backward(ptr(gt), ptr(xt), ptr(wt), ptr(ft), B, IN, 16, H, NL, act, 6, 0, ptr(bb2), ptr(gi2), ptr(gw2), stream()))
    torch.cuda.synchronize()
    assert np.array_equal(gw2.cpu().numpy(), gw) and (gi2 == 5.0).all()


def test_ffmlp_module_training_step(oracle, dev):
    """The FFMLP module end to end under autocast: padding to 128, output slicing, autograd into .weights."""
    import ffmlp

    torch.manual_seed(0)
    net = ffmlp.FFMLP(32, 3, 64, 3).to(dev)
    x = (torch.rand(1000, 32, device=dev) * 2 - 1).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        y = net(x)
    assert y.shape == (1000, 3) and y.dtype == torch.float16
    (y.float() ** 2).sum().backward()
    assert net.weights.grad is not None and net.weights.grad.shape == net.weights.shape and torch.isfinite(net.weights.grad).all()
    assert x.grad is not None and x.grad.shape == x.shape and x.grad.abs().sum() > 0

    w = net.weights.detach().half().cpu().numpy()
    xp = np.zeros((1024, 32), np.float16)
    xp[:1000] = x.detach().half().cpu().numpy()
    want, _ = oracle.ffmlp_forward(xp, w, 32, 16, 64, 3, 0, 6)
    _close_half(y.detach().cpu().numpy(), want[:1000, :3], ulps=6.0, floor=5e-2)
    net.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y2 = net(x)
    assert torch.equal(y2, y.detach())


def test_ffmlp_errors(dev):
    from nerftex_hip import lib, ptr, stream

    x = torch.zeros(128, 32, dtype=torch.float16, device=dev)
    w = torch.zeros(64 * (32 + 64 + 16), dtype=torch.float16, device=dev)
    out = torch.zeros(128, 16, dtype=torch.float16, device=dev)
    fb = torch.zeros(2, 128, 64, dtype=torch.float16, device=dev)
    assert lib.nerftex_ffmlp_forward(ptr(x), ptr(w), 128, 32, 16, 48, 2, 0, 6, ptr(fb), ptr(out), stream()) != 0
    assert lib.nerftex_last_error().decode() == "hidden_dim should in [16, 32, 64, 128, 256]"
    assert lib.nerftex_ffmlp_forward(ptr(x), ptr(w), 100, 32, 16, 64, 2, 0, 6, ptr(fb), ptr(out), stream()) != 0
    assert "128" in lib.nerftex_last_error().decode()
    # (round 5: hidden 256 with 4 layers -- 536 KB of fragments -- is no longer refused: the streaming kernels take it)
    w4 = torch.zeros(256 * (32 + 256 * 3 + 16), dtype=torch.float16, device=dev)
    fb4 = torch.zeros(4, 128, 256, dtype=torch.float16, device=dev)
    assert lib.nerftex_ffmlp_forward(ptr(x), ptr(w4), 128, 32, 16, 256, 4, 0, 6, ptr(fb4), ptr(out), stream()) == 0


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[5], CASES[7], CASES[9]], ids=lambda c: f"in{c[0]}_h{c[1]}_L{c[2]}_act{c[3]}")
def test_ffmlp_backward_recompute_equals_stored_activations(dev, case):
    """forward_buffer = NULL: the fused backward rebuilds the activations from the inputs with the forward kernel's chain.
    Same halfs, so the gradients must be bit-identical to the run that reads the stored forward_buffer."""
    from nerftex_hip import check, lib, ptr, stream

    IN, H, NL, act, B, w, x = _setup(case, 41)
    rng = np.random.default_rng(42)
    grad = (rng.standard_normal((B, 16)) * 1e-2).astype(np.float16)
    gt, xt, wt = t(grad, dev), t(x, dev), t(w, dev)
    fb = torch.empty(NL, B, H, dtype=torch.float16, device=dev)
    out = torch.empty(B, 16, dtype=torch.float16, device=dev)
    check(lib.nerftex_ffmlp_forward(ptr(xt), ptr(wt), B, IN, 16, H, NL, act, 6, ptr(fb), ptr(out), stream()))
    res = []
    for fwd in (fb, None):
        gi = torch.zeros(B, IN, dtype=torch.float16, device=dev)
        gw = torch.zeros_like(wt)
        bb = torch.zeros(NL, B, H, dtype=torch.float16, device=dev)
        check(lib.nerftex_ffmlp_backward(ptr(gt), ptr(xt), ptr(wt), ptr(fwd), B, IN, 16, H, NL, act, 6, 1, ptr(bb), ptr(gi), ptr(gw), stream()))
        torch.cuda.synchronize()
        res.append((gi.cpu().numpy(), gw.cpu().numpy()))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.count_nonzero(res[1][1]) > 0


# ------------------------------------------------------------------------------------------------------------------ bf16 storage
BF16_CASES = [CASES[0], CASES[1], CASES[3], CASES[4], CASES[5]]


def _close_bf16(got, want, ulps, floor):
    tol = ulps * 2.0 ** -7 * np.maximum(np.abs(want), floor)  # bfloat16 has a 7-bit mantissa
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} beyond {ulps} bf16-ulps; worst {np.abs(got - want).max()}"


@pytest.mark.parametrize("case", BF16_CASES, ids=[f"in{c[0]}_h{c[1]}_L{c[2]}_act{c[3]}" for c in BF16_CASES])
def test_ffmlp_bf16_forward_and_backward(oracle, dev, case):
    """The bf16 instantiation of the same kernels (BASELINE.json configs[2]) against the oracle run in its bf16 storage mode: forward with
    and without forward_buffer, inference == training forward bit for bit, fused recomputing backward (where instantiated) == stored-
    activation backward, gradients vs the exact-accumulation oracle."""
    from nerftex_hip import check, lib, ptr, stream

    IN, H, NL, act, B, w, x = _setup(case, 41)
    wb, xb = oracle.to_bf16(w.astype(np.float32)), oracle.to_bf16(x.astype(np.float32))
    rng = np.random.default_rng(42)
    gb = oracle.to_bf16((rng.standard_normal((B, 16)) * 0.05).astype(np.float32))
    with oracle.ffmlp_bf16():
        want_out, want_fb = oracle.ffmlp_forward(xb, wb, IN, 16, H, NL, act, 6)
        want_gw, want_gx, _ = oracle.ffmlp_backward(gb, xb, wb, want_fb, IN, 16, H, NL, act, True)
    view = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(dev).view(torch.bfloat16)  # noqa: E731
    xt, wt, gt = view(xb), view(wb), view(gb)
    out = torch.full((B, 16), 9.0, dtype=torch.bfloat16, device=dev)
    fb = torch.full((NL, B, H), 9.0, dtype=torch.bfloat16, device=dev)
    check(lib.nerftex_ffmlp_forward_bf16(ptr(xt), ptr(wt), B, IN, 16, H, NL, act, 6, ptr(fb), ptr(out), stream()))
    got_fb = fb.float().cpu().numpy()
    _close_bf16(got_fb[0], oracle.from_bf16(want_fb[0]), 1.01, 1e-2)
    for l in range(1, NL):
        _close_bf16(got_fb[l], oracle.from_bf16(want_fb[l]), 4.0, 5e-2)
    _close_bf16(out.float().cpu().numpy(), oracle.from_bf16(want_out), 6.0, 1e-1)
    inf = torch.empty_like(out)
    check(lib.nerftex_ffmlp_inference_bf16(ptr(xt), ptr(wt), B, IN, 16, H, NL, act, 6, None, ptr(inf), stream()))
    assert torch.equal(inf, out)

    def backward(fwd_buffer):
        gw = torch.full_like(wt, 7.0)
        gx = torch.full_like(xt, 7.0)
        bb = torch.zeros_like(fb)
        check(lib.nerftex_ffmlp_backward_bf16(ptr(gt), ptr(xt), ptr(wt), ptr(fwd_buffer), B, IN, 16, H, NL, act, 6, 1, ptr(bb), ptr(gx), ptr(gw), stream()))
        torch.cuda.synchronize()
        return gw.float().cpu().numpy(), gx.float().cpu().numpy()

    gw, gx = backward(fb)
    wgw, wgx = oracle.from_bf16(want_gw), oracle.from_bf16(want_gx)
    np.testing.assert_allclose(gw, wgw, rtol=0, atol=2.5e-2 * np.abs(wgw).max())  # bf16: 2^-8 per rounding, sums over the batch
    np.testing.assert_allclose(gx, wgx, rtol=0, atol=2.5e-2 * np.abs(wgx).max())
    if H == 64 and 2 <= NL <= 4 and IN <= 64:  # the fused kernel can rebuild the activations from the inputs
        gw2, gx2 = backward(None)
        assert np.array_equal(gw2, gw) and np.array_equal(gx2, gx)


def test_ffmlp_module_bf16_trains(dev):
    """FFMLP(dtype=torch.bfloat16) under bf16 autocast: outputs / gradients are bf16-valued, close to the fp16 module's, and a few Adam
    steps reduce the loss (no loss scaling: bf16 has fp32's exponent range)."""
    from ffmlp import FFMLP

    torch.manual_seed(0)
    a = FFMLP(32, 16, 64, 2).to(dev)
    b = FFMLP(32, 16, 64, 2, dtype=torch.bfloat16).to(dev)
    x = torch.rand(1000, 32, device=dev) * 2 - 1
    y = torch.rand(1000, 16, device=dev)
    with torch.autocast("cuda", dtype=torch.float16):
        ya = a(x)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yb = b(x)
    assert ya.dtype == torch.float16 and yb.dtype == torch.bfloat16
    torch.testing.assert_close(yb.float(), ya.float(), rtol=0, atol=3e-2)
    opt = torch.optim.Adam(b.parameters(), lr=1e-2)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.mse_loss(b(x).float(), y)
        loss.backward()
        assert b.weights.grad.dtype == torch.float32 and torch.isfinite(b.weights.grad).all()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.7 * losses[0]
