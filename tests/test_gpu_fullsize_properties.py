"""Size-independent properties at BASELINE.json's FULL sizes -- the 8192-ray training batch of configs[2] (~460 k samples, the fox table
L = 16, F = 2, T = 2^19) and the 640 000 rays of an 800 x 800 frame -- where the scalar oracle would take minutes: things that must hold
whatever the inputs are, checked on the HIP path's own outputs.

  G1 / G2   adjointness  <G1(x; T), g> = <T, G2(x; g)>  (the backward IS the transpose of the forward), fp32 and fp16;
            linearity of G2 in g (fp32); exact scaling G2(x; 2 g) = 2 G2(x; g) on the fp16 fixed-point path
  FFMLP     the backward is linear in the output gradient: doubling it doubles dL/dinputs and dL/dweights EXACTLY (a power of two)
  march     ray records are an exclusive prefix sum that adds up to the counter; every sample lies inside the box; steps are positive;
            the same call twice gives the same bits (no ordering atomics)
  composite weights_sum in [0, 1]; colours in [0, 1] composite to [0, 1]; zero density composites to zero; the one-launch training step equals the
            three launches bit for bit and its gradient scales exactly with a power-of-two loss scale
  compact   the survivors keep their order (an ordered compaction of a sorted list is sorted) and their number is the counter
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def batch(dev):
    """The bench's workload: 8192 rays of the fox-style scene marched through its occupancy grid (perturbed), ~460 k samples."""
    import raymarching
    from ngp_harness import scene

    sc = scene.Scene(bound=2.0, seed=0)
    _, _, bits = sc.bitfield()
    o, d = scene.train_batch(8192, seed=100, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)
    bits_t = torch.from_numpy(bits).to(dev)
    nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
    counter = torch.zeros(2, dtype=torch.int32, device=dev)
    xyzs, dirs, deltas, rays = raymarching.march_rays_train(ro, rd, 2.0, bits_t, sc.cascade, 128, nears, fars, counter, -1, True, 128, False, 1 / 128, 1024)
    return dict(ro=ro, rd=rd, bits=bits_t, cascade=sc.cascade, nears=nears, fars=fars, xyzs=xyzs, dirs=dirs, deltas=deltas, rays=rays, counter=counter,
                total=int(counter[0]))


def _fox_encoder(dev):
    from gridencoder import GridEncoder

    torch.manual_seed(3)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * 2).to(dev)
    enc.embeddings.data.uniform_(-1.0, 1.0)
    return enc


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_grid_backward_is_the_adjoint_of_the_forward_at_full_size(dev, batch, half):
    enc = _fox_encoder(dev)
    x = batch["xyzs"][: batch["total"] // 128 * 128].detach()
    assert x.shape[0] > 400000
    g = torch.randn(x.shape[0], 32, generator=torch.Generator().manual_seed(5)).to(dev) * 1e-2
    with torch.autocast("cuda", dtype=torch.float16, enabled=half):
        out = enc(x, bound=2.0)
    assert out.dtype == (torch.float16 if half else torch.float32)
    gq = g.to(out.dtype)
    (grad_t,) = torch.autograd.grad(out, enc.embeddings, gq)
    table = enc.embeddings.detach().half().double() if half else enc.embeddings.detach().double()
    lhs = float((out.double() * gq.double()).sum())
    rhs = float((table * grad_t.double()).sum())
    scale = float((out.double().abs() * gq.double().abs()).sum())
    # fp32: both sides are sums of the same 59 M products in different association; fp16: each output / each table-gradient row is rounded to half
    # once (2^-11 of its own magnitude, signs random)
    assert abs(lhs - rhs) <= (3e-6 if not half else 2e-4) * scale, (lhs, rhs, scale)


def test_grid_backward_is_linear_and_scales_exactly_at_full_size(dev, batch):
    from nerftex_hip import F16, F32, check, lib, ptr, stream

    enc = _fox_encoder(dev)
    B = batch["total"] // 128 * 128
    x = ((batch["xyzs"][:B] + 2.0) / 4.0).contiguous()
    gen = torch.Generator().manual_seed(6)
    g1 = (torch.randn(B, 32, generator=gen) * 1e-2).to(dev)
    g2 = (torch.randn(B, 32, generator=gen) * 1e-2).to(dev)
    S, L = float(np.log2(enc.per_level_scale)), 16
    rows = enc.embeddings.shape[0]

    def backward(g, tag):
        g = g.contiguous()
        out = torch.zeros(rows, 2, dtype=g.dtype, device=dev)
        dummy = torch.zeros(1, dtype=g.dtype, device=dev)
        check(lib.nerftex_grid_encode_backward(ptr(g), ptr(x), None, ptr(enc.offsets), ptr(out), B, 3, 2, L, S, 16, 0, ptr(dummy), ptr(dummy), 0, 0, tag, 1, stream()))
        return out

    from gridencoder.grid import register_offsets

    register_offsets(enc.offsets, L)
    a, b, ab = backward(g1, F32), backward(g2, F32), backward(g1 + g2, F32)
    err = (ab.double() - (a.double() + b.double())).abs().max()
    assert float(err) <= 2e-5 * float(ab.abs().max()), float(err)  # fp32 atomics / sums in another order
    # fp16, the benchmarked path: exact fixed-point sums of half shares -- doubling every gradient doubles every NORMAL share, every sum and
    # every rounded result exactly; a share in half's subnormal range (a corner weight below 2^-14 / |g|) is rounded on an absolute grid of
    # 2^-24 and does not scale: those few leave a row off by a grain or two
    h = (g1 * 64).half()
    one, two = backward(h, F16), backward(h * 2, F16)
    # (and a pair's split between its two rows is rounded on the 2^-24 grid as well: b = round(fixed(g') p / 2^16) -- a grain that can tip the
    # row's final rounding: at most one ulp of the result, on few rows)
    diff = (two.double() - 2 * one.double()).abs()
    assert bool((diff <= 2.0 ** -10 * two.double().abs() + 2.0 ** -20).all()), float(diff.max())
    assert float((diff > 0).double().mean()) < 2e-2, float((diff > 0).double().mean())
    assert float(one.float().abs().max()) > 0


def test_ffmlp_backward_scales_exactly_with_the_output_gradient_at_full_size(dev, batch):
    from ffmlp import FFMLP

    B = batch["total"] // 128 * 128
    torch.manual_seed(8)
    net = FFMLP(32, 16, 64, 3).to(dev)
    x = (torch.randn(B, 32, generator=torch.Generator().manual_seed(9)) * 0.5).to(dev).half().requires_grad_(True)
    g = torch.randn(B, 16, generator=torch.Generator().manual_seed(10)).to(dev).half()
    res = []
    for k in (1.0, 2.0, 0.5):
        with torch.autocast("cuda", dtype=torch.float16):
            y = net(x)
        gx, gw = torch.autograd.grad(y, (x, net.weights), g * k)
        res.append((gx.clone(), gw.clone()))
    # every product and sum of the chain scales by the power of two; the fp32 weight-gradient sums too (same order, same roundings); the only
    # exception would be halves that leave the normal range, which these magnitudes do not
    tiny = 2.0 ** -12  # (twice the smallest normal half: below it a result, or the intermediate it came from, may sit on the subnormal grid)
    for (gx, gw), k in zip(res[1:], (2.0, 0.5)):
        ok = (gx == res[0][0] * k) | (res[0][0].abs().float() < tiny / min(k, 1.0))
        assert float(ok.double().mean()) > 0.999, float(ok.double().mean())
        rel = (gx.double() - res[0][0].double() * k).abs().max() / (res[0][0].double().abs().max() * k)
        assert float(rel) < 1e-3
        # fp32 sums of products that scale exactly wherever the hidden gradients stay in half's normal range, rounded to half once: the few that
        # do not can tip that rounding by one ulp
        dw = (gw.double() - res[0][1].double() * k).abs()
        assert bool((dw <= 2.0 ** -10 * gw.double().abs() + 1e-6).all()), float(dw.max())
        assert float((dw > 0).double().mean()) < 1e-2
    assert float(res[0][1].float().abs().max()) > 0 and float(res[0][0].float().abs().max()) > 0


def test_training_march_invariants_at_full_size(dev, batch):
    import raymarching

    rays, total = batch["rays"].cpu().numpy().astype(np.int64), batch["total"]
    N = rays.shape[0]
    assert np.array_equal(rays[:, 0], np.arange(N)), "record n is ray n"
    assert np.array_equal(rays[:, 1], np.cumsum(rays[:, 2]) - rays[:, 2]), "offsets are the exclusive prefix sum of the counts"
    assert int(rays[:, 2].sum()) == total and int(batch["counter"][1]) == N
    xyzs, deltas = batch["xyzs"][:total], batch["deltas"][:total]
    assert float(xyzs.abs().max()) <= 2.0
    assert float(deltas.min()) > 0 and bool((deltas[:, 1] >= deltas[:, 0] - 1e-6).all()), "dt > 0; the distance to the previous step end is at least the step"
    assert float(batch["xyzs"][total:].abs().sum()) == 0 and float(batch["deltas"][total:].abs().sum()) == 0, "rows past the total stay zero"
    # every sample of a ray lies on the ray, in order
    k = np.flatnonzero(rays[:, 2] > 4)[:256]
    for n in k[:64]:
        o, c = rays[n, 1], rays[n, 2]
        p = batch["xyzs"][o:o + c].double()
        t = ((p - batch["ro"][n].double()) * batch["rd"][n].double()).sum(-1) / (batch["rd"][n].double() ** 2).sum()
        assert bool((t[1:] > t[:-1]).all()) and float(t[0]) >= float(batch["nears"][n]) - 1e-4 and float(t[-1]) <= float(batch["fars"][n]) + 1e-4
    # no ordering atomics: the same call again gives the same bits
    c2 = torch.zeros(2, dtype=torch.int32, device=dev)
    again = raymarching.march_rays_train(batch["ro"], batch["rd"], 2.0, batch["bits"], batch["cascade"], 128, batch["nears"], batch["fars"], c2, -1, True, 128,
                                         False, 1 / 128, 1024)
    for a, b in zip(again, (batch["xyzs"], batch["dirs"], batch["deltas"], batch["rays"])):
        assert torch.equal(a, b)
    assert torch.equal(c2, batch["counter"])


def test_compositing_bounds_at_full_size(dev, batch):
    import raymarching

    M = batch["xyzs"].shape[0]
    gen = torch.Generator().manual_seed(11)
    sigmas = (torch.rand(M, generator=gen) * 40).to(dev)
    rgbs = torch.rand(M, 3, generator=gen).to(dev)
    ws, depth, image = raymarching.composite_rays_train(sigmas, rgbs, batch["deltas"], batch["rays"])
    assert float(ws.min()) >= 0 and float(ws.max()) <= 1 + 1e-5
    assert float(image.min()) >= 0 and float(image.max()) <= 1 + 1e-5, "a convex combination of colours in [0, 1], weights summing to at most 1"
    assert bool((image.max(dim=-1).values <= ws + 1e-5).all())
    ws0, d0, im0 = raymarching.composite_rays_train(torch.zeros_like(sigmas), rgbs, batch["deltas"], batch["rays"])
    assert float(ws0.abs().max()) == 0 and float(im0.abs().max()) == 0 and float(d0.abs().max()) == 0


def test_composite_step_at_full_size(dev, batch):
    """nerftex_composite_step (round 6: compositing forward + render tail + backward as one launch) on the bench's batch in full-size buffers
    (8192 x 1024 rows, ~95 % of them covered by no ray): every output, the gradients and the step flags equal the three launches' bit for bit; the
    gradient scales EXACTLY with a power-of-two loss scale; rows no ray covers get exact zeros and dead flags."""
    from nerftex_hip import check, lib, ptr, stream

    M, N, total = batch["xyzs"].shape[0], batch["rays"].shape[0], batch["total"]
    gen = torch.Generator().manual_seed(12)
    sigmas = (torch.rand(M, generator=gen) * 40).to(dev)
    rgbs = torch.rand(M, 3, generator=gen).to(dev)
    tgt = torch.rand(N, 3, generator=gen).to(dev)
    deltas, rays, nears, fars = batch["deltas"], batch["rays"], batch["nears"], batch["fars"]
    words = (M + 31) // 32
    one = torch.ones((), device=dev)

    def three(scale):
        per_ray, losses = torch.empty(9, N, device=dev), torch.empty(2, device=dev)
        ticket, partial = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(1024, device=dev)
        flags = torch.full((words,), 3, dtype=torch.int32, device=dev)
        g = torch.full((4 * M,), float("nan"), device=dev)
        check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(per_ray[0]), ptr(per_ray[1]), ptr(per_ray[3:6]), stream()))
        check(lib.nerftex_render_tail_forward_live(ptr(per_ray[0]), ptr(per_ray[1]), ptr(per_ray[3:6]), ptr(nears), ptr(fars), ptr(tgt), 1.0, 1.0, N, ptr(per_ray[6:9]),
                                                   ptr(per_ray[2]), ptr(partial), ptr(ticket), ptr(losses), ptr(scale), losses.data_ptr() + 4, ptr(flags), words, stream()))
        check(lib.nerftex_composite_tail_backward_live(ptr(one), ptr(scale), 1.0, ptr(per_ray[6:9]), ptr(tgt), 1.0, ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays),
                                                       ptr(per_ray[0]), ptr(per_ray[3:6]), M, N, ptr(g[:M]), ptr(g[M:]), ptr(flags), stream()))
        return per_ray, losses, g, flags

    def step(scale):
        per_ray, losses = torch.full((9, N), float("nan"), device=dev), torch.full((2,), float("nan"), device=dev)
        err, flags = torch.empty(N, device=dev), torch.zeros(words, dtype=torch.int32, device=dev)
        g = torch.full((4 * M,), float("nan"), device=dev)
        check(lib.nerftex_composite_step(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(nears), ptr(fars), ptr(tgt), 1.0, 1.0, ptr(scale), ptr(per_ray[0]),
                                         ptr(per_ray[1]), ptr(per_ray[3:6]), ptr(per_ray[6:9]), ptr(per_ray[2]), ptr(err), ptr(losses), losses.data_ptr() + 4, ptr(g[:M]),
                                         ptr(g[M:]), ptr(flags), stream()))
        return per_ray, losses, g, flags

    s1, s2 = torch.full((), 1024.0, device=dev), torch.full((), 2048.0, device=dev)
    a, b = three(s1), step(s1)
    for x, y, name in zip(a, b, ("per-ray outputs", "loss", "gradients", "flags")):
        if name == "flags":
            assert torch.equal(x != 0, y != 0), name
        else:
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), name
    g1, g2 = b[2], step(s2)[2]
    big = g1.abs() > 1e-30  # (below that a doubled subnormal and a flushed one may differ)
    assert torch.equal((2 * g1)[big], g2[big]), "the gradient is linear in the loss scale, exactly for a power of two"
    assert float(g1[total:M].abs().max()) == 0 and float(g1[M + 3 * total:].abs().max()) == 0, "rows no ray covers"
    assert int(b[3][(total + 31) // 32:].abs().sum()) == 0 and 0 < int((b[3] != 0).sum())
    assert float(b[1][0]) > 0 and float(b[1][1]) == float(b[1][0]) * 1024.0


def test_compaction_keeps_order_on_a_full_frame(dev):
    import raymarching

    N = 640000  # an 800 x 800 frame
    gen = torch.Generator().manual_seed(12)
    alive = torch.arange(N, dtype=torch.int32, device=dev)
    t = torch.rand(N, generator=gen).to(dev)
    t[torch.rand(N, generator=gen).to(dev) < 0.37] = -1.0  # dead rays carry a negative t (raymarching.cu:1131)
    out_alive = torch.zeros(N, dtype=torch.int32, device=dev)
    out_t = torch.zeros(N, dtype=torch.float32, device=dev)
    counter = torch.zeros(1, dtype=torch.int32, device=dev)
    raymarching.compact_rays(N, out_alive, alive, out_t, t, counter)
    n = int(counter[0])
    assert n == int((t >= 0).sum())
    kept = out_alive[:n].long()
    assert bool((kept[1:] > kept[:-1]).all()), "order-preserving: a sorted list stays sorted"
    assert torch.equal(kept, torch.nonzero(t >= 0).flatten()) and torch.equal(out_t[:n], t[t >= 0])
