"""The fused field glue (nerftex_field_*) against the framework ops it replaces (nerf/network_ff.py:60-110 as restated in
ngp_harness/model.py): same values, same roundings."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _unfused_mid(h, dirs):
    from ngp_harness.model import trunc_exp
    from shencoder import SHEncoder

    with torch.autocast("cuda", dtype=torch.float16):
        sigma = trunc_exp(h[..., 0])
        geo = h[..., 1:]
        d = SHEncoder(input_dim=3, degree=4)(dirs)
        cin = torch.cat([d.to(geo.dtype), geo, torch.zeros_like(geo[..., :1])], dim=-1)
    return sigma, cin


def test_mid_forward_backward_match_framework_ops(dev):
    from ngp_harness import fused

    torch.manual_seed(0)
    B = 4096
    h = (torch.randn(B, 16, device=dev) * 3).half()
    h[:64, 0] = torch.linspace(-20, 20, 64, device=dev).half()  # beyond the +-15 clamp of the backward
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    gs = torch.randn(B, device=dev)
    gc = (torch.randn(B, 32, device=dev) * 1e-2).half()

    h1 = h.clone().requires_grad_(True)
    s1, c1 = fused.sigma_geo_dir(h1, dirs)
    torch.autograd.backward([s1, c1], [gs, gc])
    h2 = h.clone().requires_grad_(True)
    s2, c2 = _unfused_mid(h2, dirs)
    torch.autograd.backward([s2, c2], [gs, gc])
    assert s1.dtype == torch.float32 and c1.dtype == torch.float16 and c1.shape == (B, 32)
    assert torch.equal(s1, s2.float())
    assert torch.equal(c1, c2)
    assert torch.equal(h1.grad, h2.grad)


def test_out_forward_backward_match_framework_ops(dev):
    from ngp_harness import fused

    torch.manual_seed(1)
    B = 4096
    hc = (torch.randn(B, 16, device=dev) * 4).half()
    g = torch.randn(B, 3, device=dev) * 1e-2
    a = hc.clone().requires_grad_(True)
    r1 = fused.color_out(a)
    r1.backward(g)
    b = hc.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        r2 = torch.sigmoid(b[:, :3]).float()
    r2.backward(g)
    assert r1.dtype == torch.float32 and torch.equal(r1, r2)
    assert torch.equal(a.grad, b.grad)


def test_field_fused_equals_unfused(dev):
    """Whole field, fused glue vs the reference's op sequence: identical outputs; parameter gradients up to summation order."""
    from ngp_harness.model import NGPField

    torch.manual_seed(2)
    f1 = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    f2 = NGPField(bound=2.0, mlp="ffmlp", fused_glue=False).to(dev)
    f1.encoder.embeddings.data.uniform_(-1e-1, 1e-1)
    f2.load_state_dict(f1.state_dict())
    B = 2048
    x = (torch.rand(B, 3, device=dev) * 2 - 1) * 1.9
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    outs = []
    for f in (f1, f2):
        f.train()
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color, _ = f(x, d)
        loss = (sigma.float().clamp(max=50) * 1e-2).sum() + color.float().sum()
        loss.backward()
        outs.append((sigma.float(), color.float()))
    assert f1.fused_glue and not f2.fused_glue
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    for (n1, p1), (n2, p2) in zip(f1.named_parameters(), f2.named_parameters()):
        g1, g2 = p1.grad.float(), p2.grad.float()
        scale = float(g2.abs().max())
        assert scale > 0, n1
        np.testing.assert_allclose(g1.cpu().numpy(), g2.cpu().numpy(), rtol=0, atol=2e-2 * scale, err_msg=n1)
    f1.eval()
    f2.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s1, c1, _ = f1(x, d)
        s2, c2, _ = f2(x, d)
    assert torch.equal(s1.float(), s2.float()) and torch.equal(c1.float(), c2.float())


@pytest.mark.parametrize("B", [2048, 20480], ids=["small_batch_gather", "xcd_pinned_gather"])
def test_one_kernel_field_equals_the_glue_kernel_sequence_bit_for_bit(dev, B):
    """nerftex_field_forward (both MLPs, trunc_exp, SH, concat, sigmoid behind a level-major gather) vs the sequence it replaces (rows ->
    FFMLP -> field_mid -> FFMLP -> field_out): outputs AND every parameter gradient identical -- the backward runs the same kernels on the
    side outputs the fused forward left (x_rows, h, cin), which must therefore equal the tensors of the unfused sequence."""
    from ngp_harness.model import NGPField

    torch.manual_seed(3)
    f1 = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True, fused_field=True).to(dev)
    f2 = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True, fused_field=False).to(dev)
    f1.encoder.embeddings.data.uniform_(-0.5, 0.5)
    f2.load_state_dict(f1.state_dict())
    assert f1.fused_field and not f2.fused_field
    x = (torch.rand(B, 3, device=dev) * 2 - 1) * 1.99
    x[:7] = 2.5  # outside the table's domain: zero features
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
    gs, gc = torch.randn(B, device=dev) * 1e-2, torch.randn(B, 3, device=dev)
    for f in (f1, f2):
        f.train()
        with torch.autocast("cuda", dtype=torch.float16):
            sigma, color, _ = f(x, d)
        torch.autograd.backward([sigma, color], [gs, gc])
        f.out = (sigma.detach().clone(), color.detach().clone())
    assert f1.out[0].dtype == torch.float32 and f1.out[1].dtype == torch.float32
    assert torch.equal(f1.out[0], f2.out[0].float()) and torch.equal(f1.out[1], f2.out[1].float())
    for (n1, p1), (_, p2) in zip(f1.named_parameters(), f2.named_parameters()):
        assert p1.grad is not None
        if n1 == "encoder.embeddings" and B < 16384:  # the small-batch table gradient uses fp16 atomics: same addends, order-dependent rounding
            torch.testing.assert_close(p1.grad, p2.grad, rtol=0, atol=2e-2 * float(p2.grad.abs().max()))
        else:
            assert torch.equal(p1.grad, p2.grad), n1
    f1.eval()
    f2.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s1, c1, _ = f1(x, d)
        s2, c2, _ = f2(x, d)
    assert torch.equal(s1, s2.float()) and torch.equal(c1, c2.float())
    assert torch.equal(s1, f1.out[0]) and torch.equal(c1, f1.out[1]), "inference variant == training variant"


def test_field_backward_entry_equals_glue_plus_mlp_backward(dev):
    """nerftex_field_backward (the glue kernels folded into the load stage of the two recomputing MLP backward kernels, one reduction
    launch for both networks) against nerftex_field_out_backward + nerftex_ffmlp_backward + nerftex_field_mid_backward +
    nerftex_ffmlp_backward: every output half
    identical -- including logits outside the trunc_exp clamp, zero / huge / non-finite incoming gradients and colours at 0 and 1."""
    from nerftex_hip import check, lib, ptr, stream

    torch.manual_seed(11)
    B = 4096
    half = dict(dtype=torch.float16, device=dev)
    wc = ((torch.rand(64 * (32 + 128 + 16), device=dev) * 2 - 1) * 0.2).half()
    ws = ((torch.rand(64 * (32 + 64 + 16), device=dev) * 2 - 1) * 0.2).half()
    cin = torch.randn(B, 32, device=dev).half()
    x_rows = torch.randn(B, 32, device=dev).half()
    rgbs = torch.sigmoid(torch.randn(B, 3, device=dev) * 3).half().float()
    rgbs[:16] = 0.0
    rgbs[16:32] = 1.0
    grad_rgbs = torch.randn(B, 3, device=dev) * 10
    grad_rgbs[32:48] = 0.0
    grad_rgbs[48:56] = 1e6      # overflows half: inf, as the glue kernel gives it
    grad_rgbs[56:60] = float("nan")
    h = (torch.randn(B, 16, device=dev) * 6).half()
    h[:8, 0] = 30.0             # beyond the clamp of trunc_exp's backward
    h[8:16, 0] = -30.0
    grad_sigma = torch.randn(B, device=dev) * 1e-2
    grad_sigma[64:72] = 0.0
    grad_sigma[72:76] = 1e30

    def run(fused):
        grad_cin, grad_wc = torch.empty(B, 32, **half), torch.empty_like(wc)
        grad_x, grad_ws = torch.empty(B, 32, **half), torch.empty_like(ws)
        if fused:
            check(lib.nerftex_field_backward(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), B, ptr(grad_cin),
                                             ptr(grad_x), ptr(grad_ws), ptr(grad_wc), stream()))
        else:
            grad_hc, grad_h = torch.empty(B, 16, **half), torch.empty(B, 16, **half)
            check(lib.nerftex_field_out_backward(ptr(grad_rgbs), ptr(rgbs), B, ptr(grad_hc), stream()))
            check(lib.nerftex_ffmlp_backward(ptr(grad_hc), ptr(cin), ptr(wc), None, B, 32, 16, 64, 3, 0, 6, 1, None, ptr(grad_cin), ptr(grad_wc), stream()))
            check(lib.nerftex_field_mid_backward(ptr(grad_sigma), ptr(grad_cin), ptr(h), B, ptr(grad_h), stream()))
            check(lib.nerftex_ffmlp_backward(ptr(grad_h), ptr(x_rows), ptr(ws), None, B, 32, 16, 64, 2, 0, 6, 1, None, ptr(grad_x), ptr(grad_ws), stream()))
        torch.cuda.synchronize()
        return grad_cin, grad_wc, grad_x, grad_ws

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert lib.nerftex_field_backward(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws), ptr(wc), 100, ptr(cin), ptr(cin),
                                      ptr(ws), ptr(wc), stream()) != 0  # B % 128
