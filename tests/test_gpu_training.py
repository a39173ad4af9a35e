"""End-to-end sanity of the whole hot path as a training loop: the field must actually learn a simple target through every fused /
binned / recomputing kernel of the default configuration (fp16 autocast, FFMLP, fused glue), eagerly and from a replayed graph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _setup(dev, rays=4096):
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    o, d = scene.train_batch(rays, seed=5, n_views=4)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    target = (rd * 0.5 + 0.5).clamp(0, 1)  # a smooth, view-dependent colour: learnable by the SH + MLP head alone
    return field, r, ro, rd, target


def test_training_reduces_the_loss(dev):
    field, r, ro, rd, target = _setup(dev)
    opt = torch.optim.Adam(field.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True)
    scaler = torch.amp.GradScaler("cuda")
    field.train()
    losses = []
    for it in range(120):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            image, _, _ = r.render_train(ro, rd, dt_gamma=1 / 128, bg_color=1, perturb=True)
            loss = torch.nn.functional.mse_loss(image, target)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss.detach()))
        if r.local_step == 16:
            r.update_mean_count()
    assert all(np.isfinite(losses))
    # rays that miss every blob render the white background whatever the field does: part of the loss is irreducible
    assert losses[-1] < 0.45 * losses[0] and min(losses[-10:]) < min(losses[:10]), (losses[0], losses[-1])
    for p in field.parameters():
        assert torch.isfinite(p).all()


def test_graph_replay_trains_like_eager(dev):
    """The same steps from one captured HIP graph: the loss trajectory must follow the eager one (same kernels, same order;
    only the sample buffers are sized a little larger)."""
    traj = []
    for use_graph in (False, True):
        field, r, ro, rd, target = _setup(dev)
        opt = torch.optim.Adam(field.parameters(), lr=1e-2, betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=True)
        scaler = torch.amp.GradScaler("cuda")
        field.train()
        with torch.autocast("cuda", dtype=torch.float16):
            r.render_train(ro, rd, dt_gamma=1 / 128)  # first step sizes the buffers from the counter (one read-back)
        r.update_mean_count()
        M = (r.mean_count + 4095) // 4096 * 4096 + 4096
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        loss_buf = torch.zeros((), device=dev)

        def body():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                image, _, _ = r.render_train(ro, rd, dt_gamma=1 / 128, bg_color=1, perturb=True, counter=counter, mean_count=M)
                loss = torch.nn.functional.mse_loss(image, target)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            loss_buf.copy_(loss.detach())

        losses = []
        if use_graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    body()
                    losses.append(float(loss_buf))
            torch.cuda.current_stream().wait_stream(side)
            from ngp_harness.streams import capture_section

            g = torch.cuda.CUDAGraph()
            with capture_section(), torch.cuda.graph(g):
                body()
            for _ in range(37):
                g.replay()
                losses.append(float(loss_buf))
        else:
            for _ in range(40):
                body()
                losses.append(float(loss_buf))
        traj.append(losses)
    eager, graph = np.array(traj[0]), np.array(traj[1])
    assert np.isfinite(graph).all() and graph[-1] < 0.6 * graph[0]
    np.testing.assert_allclose(graph, eager, rtol=0.15, atol=2e-3)


def test_pipelined_inference_loop_renders_the_same_image(dev):
    """Renderer.render_infer_pipelined (launches sized by the previous iteration's alive count, true count read on the device: no
    per-iteration host stall) against Renderer.render_infer (the reference loop with its alive_counter.item()): the same image and
    depth -- per ray the arithmetic is identical, only the batching of the loop differs."""
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer

    torch.manual_seed(0)
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
    field.encoder.embeddings.data.uniform_(-0.5, 0.5)
    r = Renderer(field, bound=2.0, min_near=0.2, density_thresh=10.0).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(3))[0]
    o, d = scene.get_rays(pose, scene.intrinsics(160, 160), 160, 160)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    with torch.autocast("cuda", dtype=torch.float16):
        img_a, dep_a, _ = r.render_infer(ro, rd, dt_gamma=1 / 128)
        iters_a = r.last_iters
        img_b, dep_b, _ = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128)
    assert r.last_iters <= iters_a + 8  # a bound that lags one iteration gives a smaller n_step now and then: a few more, smaller iterations
    torch.testing.assert_close(img_b, img_a, rtol=0, atol=1e-5)
    torch.testing.assert_close(dep_b, dep_a, rtol=0, atol=1e-5)
    assert float((img_a - 1).abs().max()) > 0.05, "the view shows the scene"
    # Bigger chunks (F N sample slots per iteration instead of the reference's N), the rays cut into ranges that run on their own streams,
    # rows past the device-side alive count skipped by the field kernels: a ray's samples and the order they are composited in are
    # untouched by any of it -- the SAME image, bit for bit, in a third of the iterations.
    with torch.autocast("cuda", dtype=torch.float16):
        for F, parts in ((4, 1), (3, 2), (4, 3)):
            img_c, dep_c, _ = r.render_infer_pipelined(ro, rd, dt_gamma=1 / 128, slots_per_ray=F, parts=parts)
            assert torch.equal(img_c, img_b) and torch.equal(dep_c, dep_b), (F, parts)
            assert r.last_iters < iters_a // 2
        img_d, dep_d, _ = r.render_infer(ro, rd, dt_gamma=1 / 128, slots_per_ray=4)  # the reference loop itself with the bigger chunks
        assert torch.equal(img_d, img_a) and torch.equal(dep_d, dep_a)


def test_field_kernels_skip_rows_past_the_device_count(dev):
    """nerftex_grid_encode_forward_rows + nerftex_field_forward_rows: the first count * rows_per_unit rows equal the full evaluation, the
    rest of the outputs is left alone."""
    from ngp_harness import fused
    from ngp_harness.model import NGPField

    torch.manual_seed(1)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).eval()
    field.encoder.embeddings.data.uniform_(-0.5, 0.5)
    B = 16384
    x = (torch.rand(B, 3, device=dev) * 4 - 2).contiguous()
    d = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1).contiguous()
    count = torch.tensor([1500], dtype=torch.int32, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        s_full, c_full = field.infer(x, d)
        s_full, c_full = s_full.clone(), c_full.clone()
        s_part, c_part = field.infer(x, d, live=(count, 4))
    live = 1500 * 4
    assert torch.equal(s_part[:live], s_full[:live]) and torch.equal(c_part[:live], c_full[:live])
    # whole 32-row steps past the live rows are not computed at all: with the output buffer pre-filled, the tail keeps the fill
    flat = torch.full((4 * B,), -7.0, dtype=torch.float32, device=dev)
    from nerftex_hip import F16, LAYOUT_LBC, check, lib, ptr, stream
    enc = field.encoder
    feats = torch.zeros(16, B, 2, dtype=torch.float16, device=dev)
    with torch.autocast("cuda", dtype=torch.float16):
        table = enc._table() if enc._table().dtype == torch.float16 else enc.embeddings.detach().half()
    check(lib.nerftex_grid_encode_forward_rows(ptr(x), ptr(table), ptr(enc.offsets), ptr(feats), B, 3, 2, 16, float(np.log2(enc.per_level_scale)),
                                               int(enc.base_resolution), int(enc.gridtype_id), int(bool(enc.align_corners)), F16, LAYOUT_LBC, 2.0, 0.25,
                                               ptr(count), 4, stream()))
    assert float(feats[:, live + 256:].float().abs().max()) == 0.0 and float(feats[:, :live].float().abs().max()) > 0
    ws, wc = field.sigma_net.weights.detach().half(), field.color_net.weights.detach().half()
    check(lib.nerftex_field_forward_rows(ptr(feats), ptr(d), ptr(ws), ptr(wc), B, ptr(flat[:B]), ptr(flat[B:]), ptr(count), 4, stream()))
    torch.cuda.synchronize()
    assert torch.equal(flat[:live], s_full[:live])
    assert float((flat[live + 128:B] + 7.0).abs().max()) == 0.0


def test_release_workspaces_then_keep_working(dev):
    """nerftex_release_workspaces frees the library's scratch (march log, binning records, ...); the next calls allocate again and give
    the same results."""
    import raymarching
    from nerftex_hip import check, lib
    from ngp_harness import scene

    sc = scene.Scene(bound=2.0, seed=0)
    _, _, bits = sc.bitfield()
    o, d = scene.train_batch(2048, seed=5, n_views=2)
    ro, rd, bt = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), torch.from_numpy(bits).to(dev)
    aabb = torch.tensor([-2, -2, -2, 2, 2, 2.0], device=dev)

    def march():
        nears, fars = raymarching.near_far_from_aabb(ro, rd, aabb, 0.2)
        counter = torch.zeros(2, dtype=torch.int32, device=dev)
        return raymarching.march_rays_train(ro, rd, 2.0, bt, sc.cascade, 128, nears, fars, counter, -1, False, 128, False, 1 / 128, 1024)

    before = march()
    free0 = torch.cuda.mem_get_info()[0]
    check(lib.nerftex_release_workspaces())
    assert torch.cuda.mem_get_info()[0] > free0, "the march's accepted-t log (2048 x 1024 floats) went back to the driver"
    after = march()
    for a, b in zip(before, after):
        assert torch.equal(a, b)
