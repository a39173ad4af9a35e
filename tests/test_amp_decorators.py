"""nerftex_hip.amp.custom_fwd / custom_bwd (the leaner decorators the drop-in packages use) against torch.amp's: same casts, same
autocast state inside forward and backward, same ctx attributes -- checked on the CPU autocast device, where both can run here."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))


def _make(custom_fwd, custom_bwd, cast_inputs, log):
    class F(torch.autograd.Function):
        @staticmethod
        @custom_fwd(device_type="cpu", cast_inputs=cast_inputs)
        def forward(ctx, a, b, k, opt=None, pair=None):
            log.append(("fwd", a.dtype, b.dtype, None if opt is None else opt.dtype, None if pair is None else pair[0].dtype, torch.is_autocast_enabled("cpu"),
                        ctx._fwd_used_autocast, ctx._dtype))
            ctx.save_for_backward(a, b)
            out = torch.mm(a, b.to(a.dtype)) * k  # mm: autocast would run it in bf16 if it were on
            log.append(("fwd_out", out.dtype))
            return out

        @staticmethod
        @custom_bwd(device_type="cpu")
        def backward(ctx, g):
            a, b = ctx.saved_tensors
            log.append(("bwd", g.dtype, torch.is_autocast_enabled("cpu")))
            r = torch.mm(g, b.to(g.dtype).t())
            log.append(("bwd_out", r.dtype))
            return r.to(a.dtype), None, None, None, None

    return F


@pytest.mark.parametrize("cast_inputs", [None, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("autocast_on", [False, True])
def test_lean_decorators_behave_like_torch_amp(cast_inputs, autocast_on):
    from nerftex_hip import amp as lean

    logs = []
    for impl in (torch.amp, lean):
        log = []
        F = _make(impl.custom_fwd, impl.custom_bwd, cast_inputs, log)
        torch.manual_seed(0)
        a = torch.randn(4, 5, requires_grad=True)
        b = torch.randn(5, 3, dtype=torch.float64)  # float64 is never cast
        opt = torch.randn(2).to(torch.bfloat16)
        pair = (torch.randn(2), 3)                   # a container among the arguments
        idx = torch.arange(3)                        # integer tensors are left alone
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast_on):
            out = F.apply(a, b, 2.0, opt, pair)
            log.append(("after", torch.is_autocast_enabled("cpu")))
            out2 = F.apply(a, b, 2.0)                 # optional arguments left out
        (out.float().sum() + out2.float().sum()).backward()
        log.append(("grad", a.grad.dtype, a.grad.clone()))
        del idx
        logs.append(log)
    ref, got = logs
    assert len(ref) == len(got)
    for r, g in zip(ref, got):
        assert r[0] == g[0]
        if r[0] == "grad":
            assert r[1] == g[1] and torch.equal(r[2], g[2])
        else:
            assert r == g, (r, g)


def test_lean_custom_fwd_keeps_the_signature():
    import inspect

    from nerftex_hip import amp as lean

    def forward(ctx, x, y, flag=False):
        return x

    wrapped = lean.custom_fwd(device_type="cuda", cast_inputs=torch.float32)(forward)
    assert str(inspect.signature(wrapped)) == str(inspect.signature(forward))
    with pytest.raises(ValueError):
        lean.custom_fwd(device_type=0)


def test_capture_section_collects_first_and_keeps_the_collector_off():
    """ngp_harness.streams.capture_section (around every graph recording of the package): dead cycles are collected on entry, the collector is
    off inside and back to what it was afterwards -- also when the body raises, and when it was off to begin with."""
    import gc
    import weakref

    from ngp_harness.streams import capture_section

    class Node:
        pass

    a = Node()
    a.me = a
    alive = weakref.ref(a)
    was = gc.isenabled()
    gc.disable()
    del a
    gc.enable()
    try:
        with capture_section():
            assert alive() is None and not gc.isenabled()
        assert gc.isenabled()
        try:
            with capture_section():
                raise KeyError("x")
        except KeyError:
            pass
        assert gc.isenabled()
        gc.disable()
        with capture_section():
            assert not gc.isenabled()
        assert not gc.isenabled()
    finally:
        gc.enable() if was else gc.disable()


def test_stream_pool_module_is_inert_without_a_gpu():
    """ngp_harness.streams imports and reports nothing measured until a caller on a GPU asks for the pool (the pool itself: tests/test_gpu_streams.py)."""
    from ngp_harness import streams

    assert streams.pool_report("cuda:0") is None and streams.POOL_PARTS == 3
    assert not streams._SIDE and not streams._PARTS.get(0)
