"""Fused elementwise glue of the ngp field (first step): autograd wrappers over nerftex_field_*.

`sigma_geo_dir(h, dirs)` replaces, between the two FFMLPs of nerf/network_ff.py:60-110, the slice / cast / trunc_exp of the
density logit, the SH(4) direction encoding, its narrowing to fp16, the zero pad column and the concatenation to the colour
net's 32 inputs; `color_out(hc)` replaces the slice / sigmoid / cast to fp32 after the colour net.  Same arithmetic and
roundings as the framework ops (tests/test_gpu_field_glue.py compares against them).
"""
import torch
from torch.autograd import Function

from nerftex_hip import check, lib, ptr, stream


class _sigma_geo_dir(Function):
    @staticmethod
    def forward(ctx, h, dirs):
        h = h.contiguous()
        dirs = dirs.contiguous().float()
        assert h.dtype == torch.float16 and h.shape[1] == 16 and dirs.shape == (h.shape[0], 3)
        B = h.shape[0]
        sigma = torch.empty(B, dtype=torch.float32, device=h.device)
        cin = torch.empty(B, 32, dtype=torch.float16, device=h.device)
        check(lib.nerftex_field_mid_forward(ptr(h), ptr(dirs), B, ptr(sigma), ptr(cin), stream()))
        ctx.save_for_backward(h)
        ctx.set_materialize_grads(False)
        return sigma, cin

    @staticmethod
    def backward(ctx, grad_sigma, grad_cin):
        (h,) = ctx.saved_tensors
        B = h.shape[0]
        grad_sigma = (torch.zeros(B, dtype=torch.float32, device=h.device) if grad_sigma is None else grad_sigma.contiguous().float())
        grad_cin = (torch.zeros(B, 32, dtype=torch.float16, device=h.device) if grad_cin is None else grad_cin.contiguous().half())
        grad_h = torch.empty_like(h)
        check(lib.nerftex_field_mid_backward(ptr(grad_sigma), ptr(grad_cin), ptr(h), B, ptr(grad_h), stream()))
        return grad_h, None


class _color_out(Function):
    @staticmethod
    def forward(ctx, hc):
        hc = hc.contiguous()
        assert hc.dtype == torch.float16 and hc.shape[1] == 16
        B = hc.shape[0]
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=hc.device)
        check(lib.nerftex_field_out_forward(ptr(hc), B, ptr(rgbs), stream()))
        ctx.save_for_backward(rgbs)
        return rgbs

    @staticmethod
    def backward(ctx, grad_rgbs):
        (rgbs,) = ctx.saved_tensors
        B = rgbs.shape[0]
        grad_rgbs = grad_rgbs.contiguous().float()
        grad_hc = torch.empty(B, 16, dtype=torch.float16, device=rgbs.device)
        check(lib.nerftex_field_out_backward(ptr(grad_rgbs), ptr(rgbs), B, ptr(grad_hc), stream()))
        return grad_hc


sigma_geo_dir = _sigma_geo_dir.apply
color_out = _color_out.apply


import os

# False (NERFTEX_FIELD_BACKWARD=split): the six launches (glue, MLP + reduce, glue, MLP + reduce) nerftex_field_backward's three replace --
# same gradients, for A/B
STEP_TRAILER = True  # the field backward's reduction launch rides on the hash-grid backward's fill launch (round 6; False: its own launch)
FIELD_BACKWARD_FUSED = os.environ.get("NERFTEX_FIELD_BACKWARD", "fused") != "split"
DEBUG_TAP = None  # debugging aid (tools/determinism_probe.py): a callable that is shown the field backward's intermediate gradients


class _ngp_field(Function):
    """The whole --ff field (nerf/network_ff.py:85-101) as two launches forward: the hash-grid gather writing level-major features and
    nerftex_field_forward (both MLPs, trunc_exp, SH, concat, sigmoid in one kernel; no transpose, no intermediate read-backs).  The
    backward is the chain of the existing entry points on the side outputs the forward kernel left (bit-identical to the unfused field:
    tests/test_gpu_field_glue.py)."""

    @staticmethod
    def forward(ctx, x, dirs, table, offsets, ws, wc, enc, training, live=None, mlp_dtype=torch.float16):
        import numpy as np

        from nerftex_hip import F16, LAYOUT_LBC

        from gridencoder.grid import register_offsets

        x = x.contiguous().float()
        dirs = dirs.contiguous().float()
        B = x.shape[0]
        L, C, D = offsets.shape[0] - 1, table.shape[1], x.shape[1]
        assert (L, C, D) == (16, 2, 3) and B % 128 == 0
        register_offsets(offsets, L)
        table_h = table if table.dtype == torch.float16 else table.to(torch.float16)
        # mlp_dtype bf16 (round 5): the two networks, their saved rows and their weight gradients are bf16 (nerftex_field_*_bf16); the table, the
        # gathered features and dL/dfeatures stay fp16 (the table is fp16 under any autocast, gridencoder/grid.py:38-41)
        bf16 = mlp_dtype == torch.bfloat16
        assert mlp_dtype in (torch.float16, torch.bfloat16)
        ws_h = ws if ws.dtype == mlp_dtype else ws.to(mlp_dtype)
        wc_h = wc if wc.dtype == mlp_dtype else wc.to(mlp_dtype)
        field_forward = lib.nerftex_field_forward_bf16 if bf16 else lib.nerftex_field_forward
        field_forward_rows = lib.nerftex_field_forward_rows_bf16 if bf16 else lib.nerftex_field_forward_rows
        S, H, gridtype, align, bound = float(np.log2(enc.per_level_scale)), int(enc.base_resolution), int(enc.gridtype_id), int(bool(enc.align_corners)), enc_bound(enc)
        dev = x.device
        feats = torch.empty(L, B, C, dtype=torch.float16, device=dev)
        dummy = torch.empty(1, dtype=torch.float16, device=dev)
        affine = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))
        sigma = torch.empty(B, dtype=torch.float32, device=dev)
        rgbs = torch.empty(B, 3, dtype=torch.float32, device=dev)
        if live is not None and not training:  # (device count of units, rows per unit): rows past count * rows carry nothing
            units, rows_per_unit = live
            assert units.dtype == torch.int32 and units.device == dev
            check(lib.nerftex_grid_encode_forward_rows(ptr(x), ptr(table_h), ptr(offsets), ptr(feats), B, D, C, L, S, H, gridtype, align, F16, LAYOUT_LBC,
                                                       affine[0], affine[1], ptr(units), int(rows_per_unit), stream()))
            check(field_forward_rows(ptr(feats), ptr(dirs), ptr(ws_h), ptr(wc_h), B, ptr(sigma), ptr(rgbs), ptr(units), int(rows_per_unit), stream()))
            ctx.set_materialize_grads(False)
            return sigma, rgbs
        check(lib.nerftex_grid_encode_forward_affine(ptr(x), ptr(table_h), ptr(offsets), ptr(feats), B, D, C, L, S, H, 0, ptr(dummy), gridtype, align, F16,
                                                     LAYOUT_LBC, affine[0], affine[1], stream()))
        if training:
            x_rows = torch.empty(B, 32, dtype=mlp_dtype, device=dev)
            h = torch.empty(B, 16, dtype=mlp_dtype, device=dev)
            cin = torch.empty(B, 32, dtype=mlp_dtype, device=dev)
            check(field_forward(ptr(feats), ptr(dirs), ptr(ws_h), ptr(wc_h), B, ptr(sigma), ptr(rgbs), ptr(x_rows), ptr(h), ptr(cin), None, stream()))
            ctx.save_for_backward(x, table_h, offsets, ws_h, wc_h, x_rows, h, cin, rgbs)
            ctx.meta = (S, H, gridtype, align, affine, table.dtype, ws.dtype, wc.dtype)
            ctx.mlp_dtype = mlp_dtype
            ctx.amp_sink = getattr(enc, "amp_sink", None)  # optim.FusedAmp.attach: the backward's kernels raise found_inf themselves
            ctx.grad_chunker = getattr(enc, "grad_chunker", None)  # dp.TableGradChunks: the table gradient is finished level group by level group
            ctx.table_adam = getattr(enc, "table_adam", None)  # optim.FusedAmp.fuse_table_update: the summing kernel applies Adam to the tiles it owns
            ctx.live_holder = getattr(enc, "step_live_holder", None)  # Renderer.shade_train(skip_dead_samples): where composite_tail's backward leaves its step flags
            if ctx.live_holder is not None and (FIELD_BACKWARD_FUSED or mlp_dtype == torch.bfloat16) and ws_h.dtype == wc_h.dtype == mlp_dtype:
                ctx.live_holder["field_consumes"] = True  # this node's backward is nerftex_field_backward_live_consume: it also finishes the step's loss
        else:
            check(field_forward(ptr(feats), ptr(dirs), ptr(ws_h), ptr(wc_h), B, ptr(sigma), ptr(rgbs), None, None, None, None, stream()))
        ctx.set_materialize_grads(False)
        return sigma, rgbs

    @staticmethod
    def backward(ctx, grad_sigma, grad_rgbs):
        from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE

        x, table_h, offsets, ws_h, wc_h, x_rows, h, cin, rgbs = ctx.saved_tensors
        S, H, gridtype, align, affine, t_dtype, ws_dtype, wc_dtype = ctx.meta
        B, dev = x.shape[0], x.device
        half = dict(dtype=torch.float16, device=dev)
        mlp_dtype = ctx.mlp_dtype
        bf16 = mlp_dtype == torch.bfloat16
        grad_sigma = torch.zeros(B, dtype=torch.float32, device=dev) if grad_sigma is None else grad_sigma.contiguous().float()
        grad_rgbs = torch.zeros(B, 3, dtype=torch.float32, device=dev) if grad_rgbs is None else grad_rgbs.contiguous().float()
        grad_cin, grad_wc = torch.empty(B, 32, dtype=mlp_dtype, device=dev), torch.empty_like(wc_h)
        grad_x, grad_ws = torch.empty(B, 32, **half), torch.empty_like(ws_h)  # (grad_x feeds the hash-grid backward: the TABLE's type, fp16)
        sink = ctx.amp_sink if ((FIELD_BACKWARD_FUSED or bf16) and t_dtype == torch.float16 and ws_dtype == wc_dtype == mlp_dtype) else None
        found = ptr(sink.found_inf) if sink is not None else None
        # round 6, dead-sample skip: one word per 32 samples, 0 = the compositing backward gave all 32 exactly zero gradients (composite_tail's
        # backward, which has run just before this one, left them in the holder).  Both MLP backward kernels walk the live steps only: dead steps
        # issue no loads and no MFMAs; their rows of grad_x are written as zeros (what the plain kernels compute: the hash-grid backward drops them).
        holder = ctx.live_holder
        flags = holder.pop("flags", None) if holder is not None else None
        # consume (composite_tail's one-launch form set the flags in the FORWARD, in a buffer that lives across steps): this call leaves them zero again
        consume = holder.pop("consume", False) if holder is not None else False
        loss_job = holder.pop("loss_job", None) if holder is not None else None  # (composite_tail's one-launch form left the loss for this call to finish)
        if flags is not None and not (flags.numel() * 32 >= B and (FIELD_BACKWARD_FUSED or bf16) and ws_dtype == wc_dtype == mlp_dtype):
            if consume:
                flags.zero_()
            flags = None
        trailer = None
        if consume or loss_job is not None:
            assert (FIELD_BACKWARD_FUSED or bf16) and ws_dtype == wc_dtype == mlp_dtype, "announced in the forward (field_consumes)"
            import ctypes

            job = None
            if loss_job is not None:
                from nerftex_hip import StepLoss

                err, n_rays, loss_mul, scale, losses = loss_job
                job = ctypes.byref(StepLoss(ptr(err), n_rays, loss_mul, ptr(scale), ptr(losses), losses.data_ptr() + 4))
            args = (ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws_h), ptr(wc_h), B, ptr(grad_cin), ptr(grad_x), ptr(grad_ws),
                    ptr(grad_wc), ptr(flags), job, found)
            # the reduction launch of this call (weight gradients + found_inf, the flags' clearing, the loss) is small, latency-bound and feeds nothing before
            # the optimizer: when the hash-grid backward below is the tile-owner form, its fill launch runs it on its first workgroups (STEP_TRAILER)
            if STEP_TRAILER and B > 0 and ctx.table_adam is not None and ctx.table_adam is sink and ctx.grad_chunker is None and t_dtype == torch.float16:
                from nerftex_hip import StepTrailer

                trailer = StepTrailer()
                check((lib.nerftex_field_backward_live_deferred_bf16 if bf16 else lib.nerftex_field_backward_live_deferred)(*args, ctypes.byref(trailer), stream()))
            else:
                check((lib.nerftex_field_backward_live_consume_bf16 if bf16 else lib.nerftex_field_backward_live_consume)(*args, stream()))
        elif flags is not None:
            field_backward = lib.nerftex_field_backward_live_bf16 if bf16 else lib.nerftex_field_backward_live
            check(field_backward(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws_h), ptr(wc_h), B, ptr(grad_cin), ptr(grad_x),
                                 ptr(grad_ws), ptr(grad_wc), ptr(flags), found, stream()))
        elif bf16:  # one entry point, found_inf optional (no split / unfused form of the bf16 field backward exists)
            check(lib.nerftex_field_backward_bf16(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws_h), ptr(wc_h), B,
                                                  ptr(grad_cin), ptr(grad_x), ptr(grad_ws), ptr(grad_wc), found, stream()))
        elif sink is not None:  # GradScaler's non-finite scan rides on the stores of the three gradients (no amp_check launch this step)
            check(lib.nerftex_field_backward_amp(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws_h), ptr(wc_h), B,
                                                 ptr(grad_cin), ptr(grad_x), ptr(grad_ws), ptr(grad_wc), found, stream()))
        elif FIELD_BACKWARD_FUSED:  # the two glue kernels ride on the MLP backward kernels' load stage, one reduction for both networks
            check(lib.nerftex_field_backward(ptr(grad_sigma), ptr(grad_rgbs), ptr(rgbs), ptr(h), ptr(cin), ptr(x_rows), ptr(ws_h), ptr(wc_h), B, ptr(grad_cin),
                                             ptr(grad_x), ptr(grad_ws), ptr(grad_wc), stream()))
        else:
            grad_hc = torch.empty(B, 16, **half)
            check(lib.nerftex_field_out_backward(ptr(grad_rgbs), ptr(rgbs), B, ptr(grad_hc), stream()))
            check(lib.nerftex_ffmlp_backward(ptr(grad_hc), ptr(cin), ptr(wc_h), None, B, 32, 16, 64, 3, 0, 6, 1, None, ptr(grad_cin), ptr(grad_wc), stream()))
            grad_h = torch.empty(B, 16, **half)
            check(lib.nerftex_field_mid_backward(ptr(grad_sigma), ptr(grad_cin), ptr(h), B, ptr(grad_h), stream()))
            check(lib.nerftex_ffmlp_backward(ptr(grad_h), ptr(x_rows), ptr(ws_h), None, B, 32, 16, 64, 2, 0, 6, 1, None, ptr(grad_x), ptr(grad_ws), stream()))
        if DEBUG_TAP is not None:
            DEBUG_TAP(grad_x=grad_x, x=x, grad_sigma=grad_sigma, grad_rgbs=grad_rgbs, grad_cin=grad_cin, meta=(S, H, gridtype, align, affine))
        grad_table = torch.empty_like(table_h)
        dummy = torch.empty(1, **half)
        # (the chunked form hands autograd a gradient that is FINISHED LATER, in place: only valid when `.grad` becomes this very tensor --
        # an fp16 leaf, so that `.to(t_dtype)` below is the identity, and no earlier `.grad` to accumulate into: TableGradChunks.begin checks)
        chunker = ctx.grad_chunker if ((sink is None or getattr(ctx.grad_chunker, "with_amp", False)) and t_dtype == torch.float16) else None
        if chunker is not None:
            # only BIN the contributions here; the caller sums the level groups one by one (chunker.sum_chunk) -- data parallelism: each group's
            # all-reduce starts while the next group is being summed; single GPU (round 5, with_amp): each group's Adam runs on a second stream
            # while the next group is being summed (the non-finite scan rides on each group's stores).  grad_table is complete once every group
            # has been summed.
            L = offsets.shape[0] - 1
            args = (ptr(grad_x), ptr(x), ptr(table_h), ptr(offsets), ptr(grad_table), B, 3, 2, L, S, H, gridtype, align, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE,
                    affine[0], affine[1])
            if sink is not None:
                phase = lambda ph, lo, hi: lib.nerftex_grid_encode_backward_phase_amp(*args, ph, lo, hi, found, stream())  # noqa: E731
            else:
                phase = lambda ph, lo, hi: lib.nerftex_grid_encode_backward_phase(*args, ph, lo, hi, stream())  # noqa: E731
            if phase(1, 0, L) == 0:
                chunker.begin(grad_table, lambda lo, hi: check(phase(2, lo, hi)), keep=(grad_x, x, table_h, offsets, dummy))
                if sink is not None:
                    sink.covered = (grad_table.data_ptr(), grad_ws.data_ptr(), grad_wc.data_ptr())
                return None, None, grad_table.to(t_dtype), None, grad_ws.to(ws_dtype), grad_wc.to(wc_dtype), None, None, None, None
            chunker.begin(grad_table, None, keep=None)  # (small batch / unknown table: the one-call backward below; the groups are complete already)
        fuse = ctx.table_adam.table_adam_for(table_h) if (ctx.table_adam is not None and ctx.table_adam is sink and chunker is None) else None
        if trailer is not None and fuse is not None:
            import ctypes

            first = ctypes.c_uint32(0)
            rc = lib.nerftex_grid_encode_backward_adam_trailer(ptr(grad_x), ptr(x), ptr(offsets), ptr(grad_table), B, 3, 2, offsets.shape[0] - 1, S, H, gridtype, align,
                                                               F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0], affine[1], ctypes.byref(fuse), ctypes.byref(first),
                                                               ctypes.byref(trailer), stream())
            if rc != 0:  # (refused: nothing was launched -- the trailer as a launch of its own, then the error)
                lib.nerftex_step_trailer_run(ctypes.byref(trailer), stream())
            check(rc)
            trailer = None
            sink.opt.fused_table = (sink.table_index, int(first.value))
            sink.covered = (grad_table.data_ptr(), grad_ws.data_ptr(), grad_wc.data_ptr())
            return None, None, grad_table.to(t_dtype), None, grad_ws.to(ws_dtype), grad_wc.to(wc_dtype), None, None, None, None
        if trailer is not None:  # (the hash-grid backward is not the tile-owner form after all)
            import ctypes

            check(lib.nerftex_step_trailer_run(ctypes.byref(trailer), stream()))
            trailer = None
        if fuse is not None:
            # round 6: the hashed levels' tiles never leave LDS as a gradient -- their owners run Adam on the rows (double-buffered state, so that a
            # step GradScaler skips leaves no trace); grad_table receives the coarse levels' rows [0, first) only, the rest stays uninitialised
            import ctypes

            first = ctypes.c_uint32(0)
            check(lib.nerftex_grid_encode_backward_adam(ptr(grad_x), ptr(x), ptr(offsets), ptr(grad_table), B, 3, 2, offsets.shape[0] - 1, S, H, gridtype, align,
                                                        F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0], affine[1], ctypes.byref(fuse), ctypes.byref(first),
                                                        stream()))
            sink.opt.fused_table = (sink.table_index, int(first.value))
            sink.covered = (grad_table.data_ptr(), grad_ws.data_ptr(), grad_wc.data_ptr())
        elif sink is not None:
            check(lib.nerftex_grid_encode_backward_amp(ptr(grad_x), ptr(x), ptr(table_h), ptr(offsets), ptr(grad_table), B, 3, 2, offsets.shape[0] - 1, S, H,
                                                       0, ptr(dummy), ptr(dummy), gridtype, align, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0],
                                                       affine[1], found, stream()))
            # the buffers whose scan is done (addresses, not references: a second reference would make autograd copy the gradient instead
            # of handing the tensor itself to `.grad`); FusedAmp.step checks `.grad` off against them
            sink.covered = (grad_table.data_ptr(), grad_ws.data_ptr(), grad_wc.data_ptr())
        else:
            check(lib.nerftex_grid_encode_backward_affine(ptr(grad_x), ptr(x), ptr(table_h), ptr(offsets), ptr(grad_table), B, 3, 2, offsets.shape[0] - 1, S,
                                                          H, 0, ptr(dummy), ptr(dummy), gridtype, align, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0],
                                                          affine[1], stream()))
        return None, None, grad_table.to(t_dtype), None, grad_ws.to(ws_dtype), grad_wc.to(wc_dtype), None, None, None, None


_INFER_CACHE = {}


def ngp_field_infer(x, dirs, encoder, sigma_net, color_net, bound, live=None, mlp_dtype=torch.float16):
    """The no-grad form of `ngp_field` for a loop that calls it a few dozen times per frame: two launches and nothing else -- the level
    constants and the fp16 (mlp_dtype bf16: bf16) copies of the MLP weights are kept between calls (re-made when a parameter changes), no
    autograd node.  x [B,3] fp32 contiguous, dirs [B,3] fp32 contiguous, B % 128 == 0; call under autocast(mlp_dtype)."""
    import numpy as np

    from nerftex_hip import F16, LAYOUT_LBC

    from gridencoder.grid import register_offsets

    B, dev = x.shape[0], x.device
    table_h = encoder._table()
    ws, wc = sigma_net._weights(), color_net._weights()
    key = (id(encoder), mlp_dtype)
    c = _INFER_CACHE.get(key)
    stamp = (ws._version, ws.data_ptr(), wc._version, wc.data_ptr(), float(bound), encoder.offsets.data_ptr())
    if c is None or c[0] != stamp:
        L = encoder.offsets.shape[0] - 1
        register_offsets(encoder.offsets, L)
        c = (stamp, L, float(np.log2(encoder.per_level_scale)), int(encoder.base_resolution), int(encoder.gridtype_id), int(bool(encoder.align_corners)),
             float(bound), float(np.float32(1.0) / np.float32(2 * bound)), ws.detach().to(mlp_dtype), wc.detach().to(mlp_dtype))
        _INFER_CACHE[key] = c
    _, L, S, H, gridtype, align, add, mul, ws_h, wc_h = c
    assert table_h.dtype == torch.float16 and (L, table_h.shape[1], x.shape[1]) == (16, 2, 3) and B % 128 == 0
    feats = torch.empty(L, B, 2, dtype=torch.float16, device=dev)
    out = torch.empty(4 * B, dtype=torch.float32, device=dev)  # [sigma | rgb] in one allocation
    sigma, rgbs = out[:B], out[B:].view(B, 3)
    units, rows_per_unit = (ptr(live[0]), int(live[1])) if live is not None else (None, 0)
    st = stream()
    check(lib.nerftex_grid_encode_forward_rows(ptr(x), ptr(table_h), ptr(encoder.offsets), ptr(feats), B, 3, 2, L, S, H, gridtype, align, F16, LAYOUT_LBC, add,
                                               mul, units, rows_per_unit, st))
    field_forward_rows = lib.nerftex_field_forward_rows_bf16 if mlp_dtype == torch.bfloat16 else lib.nerftex_field_forward_rows
    check(field_forward_rows(ptr(feats), ptr(dirs), ptr(ws_h), ptr(wc_h), B, ptr(sigma), ptr(rgbs), units, rows_per_unit, st))
    return sigma, rgbs


def ngp_density(x, encoder, sigma_net, bound, mlp_dtype=torch.float16):
    """sigma [B] fp32 of the --ff field's `density` (nerf/network_ff.py:103-117) for B % 128 == 0 points: the hash-grid gather (level-major
    output) and the sigma net + trunc_exp as ONE kernel behind it (nerftex_field_density) -- the occupancy-grid update's query
    (nerf/renderer.py:566-660: 2-4 M cell positions every 16 steps).  No autograd; call under autocast(mlp_dtype) -- float16, or bfloat16 for
    the bf16 field (nerftex_field_density_bf16)."""
    import numpy as np

    from nerftex_hip import F16, LAYOUT_LBC

    from gridencoder.grid import register_offsets

    B, dev = x.shape[0], x.device
    table_h = encoder._table()
    ws = sigma_net._weights()
    L = encoder.offsets.shape[0] - 1
    assert table_h.dtype == torch.float16 and (L, table_h.shape[1], x.shape[1]) == (16, 2, 3) and B % 128 == 0 and x.dtype == torch.float32
    register_offsets(encoder.offsets, L)
    ws_h = ws.detach() if ws.dtype == mlp_dtype else ws.detach().to(mlp_dtype)
    feats = torch.empty(L, B, 2, dtype=torch.float16, device=dev)
    sigma = torch.empty(B, dtype=torch.float32, device=dev)
    st = stream()
    check(lib.nerftex_grid_encode_forward_rows(ptr(x), ptr(table_h), ptr(encoder.offsets), ptr(feats), B, 3, 2, L, float(np.log2(encoder.per_level_scale)),
                                               int(encoder.base_resolution), int(encoder.gridtype_id), int(bool(encoder.align_corners)), F16, LAYOUT_LBC,
                                               float(bound), float(np.float32(1.0) / np.float32(2 * bound)), None, 0, st))
    check((lib.nerftex_field_density_bf16 if mlp_dtype == torch.bfloat16 else lib.nerftex_field_density)(ptr(feats), ptr(ws_h), B, ptr(sigma), st))
    return sigma


def enc_bound(enc):
    return getattr(enc, "_field_bound", 1.0)


def ngp_field(x, dirs, encoder, sigma_net, color_net, bound, training, live=None, mlp_dtype=torch.float16):
    """sigma [B] fp32, rgbs [B,3] fp32 of the --ff field for B % 128 == 0 points, 16-bit kernels (call under autocast): fp16 networks, or
    bf16 networks (mlp_dtype=torch.bfloat16, round 5) over the fp16 hash table.
    live = (int32 device tensor, rows per unit), fp16 inference only: just the first live[0][0] * live[1] points are evaluated."""
    encoder._field_bound = float(bound)
    return _ngp_field.apply(x, dirs, encoder._table(), encoder.offsets, sigma_net._weights(), color_net._weights(), encoder, bool(training), live, mlp_dtype)


class _render_tail(Function):
    """image + (1 - weights_sum) * bg, depth normalisation and mean squared error against `target` in one launch
    (nerf/renderer.py:417-425 + the MSE of nerf/utils.py:602-640); returns (image_out, depth_out, loss * loss_mul, scaled loss).
    scale: device scalar of a loss scaler or None; the last output is loss * scale (GradScaler.scale(loss)) and is the one to call
    backward on -- it carries the gradient to `image` and `weights_sum`; the first three are plain outputs."""

    @staticmethod
    def forward(ctx, weights_sum, depth, image, nears, fars, target, bg, loss_mul, scale):
        args = [t.contiguous().float() for t in (weights_sum, depth, image, nears, fars, target)]
        weights_sum, depth, image, nears, fars, target = args
        N, dev = weights_sum.shape[0], weights_sum.device
        assert image.shape == (N, 3) and target.shape == (N, 3) and depth.shape == (N,)
        assert scale is None or (scale.dtype == torch.float32 and scale.numel() == 1 and scale.device == dev)
        image_out = torch.empty_like(image)
        depth_out = torch.empty_like(depth)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        scratch = _tail_scratch(dev, (N + 255) // 256)
        check(lib.nerftex_render_tail_forward(ptr(weights_sum), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(target), float(bg),
                                              float(loss_mul), N, ptr(image_out), ptr(depth_out), ptr(scratch[1]), ptr(scratch[0]), ptr(losses),
                                              ptr(scale), losses.data_ptr() + 4, stream()))
        ctx.save_for_backward(image_out, target, scale)
        ctx.consts = (float(bg), float(loss_mul))
        loss, scaled = losses[0], losses[1]
        ctx.mark_non_differentiable(image_out, depth_out, loss)
        ctx.set_materialize_grads(False)
        return image_out, depth_out, loss, scaled

    @staticmethod
    def backward(ctx, _gi, _gd, _gl, grad_scaled):
        image_out, target, scale = ctx.saved_tensors
        if grad_scaled is None:
            return (None,) * 9
        bg, loss_mul = ctx.consts
        N = image_out.shape[0]
        grad_scaled = grad_scaled.contiguous().float()
        grad_image = torch.empty_like(image_out)
        grad_ws = torch.empty(N, dtype=torch.float32, device=image_out.device)
        check(lib.nerftex_render_tail_backward(ptr(grad_scaled), ptr(scale), loss_mul, ptr(image_out), ptr(target), bg, N, ptr(grad_image),
                                               ptr(grad_ws), stream()))
        return grad_ws, None, grad_image, None, None, None, None, None, None


class _composite_tail(Function):
    """raymarching.composite_rays_train + render_tail as ONE autograd node: two launches forward (the two kernels as they are), one
    launch backward (nerftex_composite_tail_backward: the render tail's backward rides on the compositing backward).
    -> (image_out, depth_out, loss * loss_mul, that times `scale`); backward through the last one reaches sigmas and rgbs.

    one (round 6): the tensor the caller will hand to `scaled.backward(one)` -- a device float holding 1.0.  With it (and a gradient wanted) the
    forward is nerftex_composite_step: ONE launch computes the outputs AND the gradients for a root gradient of one (+ a one-workgroup launch for
    the loss); the backward returns them when the root gradient is that very tensor, and runs the backward launch as before for any other."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, deltas, rays, nears, fars, target, bg, loss_mul, scale, live_holder=None, one=None):
        sigmas, rgbs, deltas = sigmas.contiguous().float(), rgbs.contiguous().float(), deltas.contiguous().float()
        nears, fars, target = nears.contiguous().float(), fars.contiguous().float(), target.contiguous().float()
        rays = rays.contiguous()
        M, N, dev = sigmas.shape[0], rays.shape[0], sigmas.device
        assert target.shape == (N, 3) and rays.dtype == torch.int32
        assert scale is None or (scale.dtype == torch.float32 and scale.numel() == 1 and scale.device == dev)
        per_ray = torch.empty(9, N, dtype=torch.float32, device=dev)  # weights_sum, depth, depth_out | image [N,3] | image_out [N,3]
        weights_sum, depth, depth_out = per_ray[0], per_ray[1], per_ray[2]
        image, image_out = per_ray[3:6].view(N, 3), per_ray[6:9].view(N, 3)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        ctx.step_grads = None
        if one is not None and 0 < N <= 262144 and M > 0 and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
            assert one.dtype == torch.float32 and one.numel() == 1 and one.device == dev
            grads = torch.empty(4 * M, dtype=torch.float32, device=dev)
            err = torch.empty(N, dtype=torch.float32, device=dev)
            # the step flags: zero on entry, set by this launch, zeroed again by the field's backward (nerftex_field_backward_live_consume) -- so the
            # buffer lives across steps (holder["buffer"], Renderer.shade_train); without one: a zero fill
            flags = None
            if live_holder is not None:
                buf = live_holder.get("buffer")
                words = (M + 31) // 32
                flags = buf[:words] if buf is not None and buf.numel() >= words else torch.zeros(words, dtype=torch.int32, device=dev)
            # the loss: the field's backward finishes it (an extra workgroup of its weight-gradient reduction: `loss` and `scaled` are COMPLETE AFTER THE
            # BACKWARD, which is when a training step reads them) when it has announced that it will; else a one-workgroup launch of this call's
            defer = live_holder is not None and live_holder.get("field_consumes", False) and live_holder.get("defer_loss", False)
            check(lib.nerftex_composite_step(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(nears), ptr(fars), ptr(target), float(bg),
                                             float(loss_mul), ptr(scale), ptr(weights_sum), ptr(depth), ptr(image), ptr(image_out), ptr(depth_out), ptr(err),
                                             None if defer else ptr(losses), losses.data_ptr() + 4, ptr(grads), grads.data_ptr() + 4 * M, ptr(flags), stream()))
            if defer:
                live_holder["loss_job"] = (err, N, float(loss_mul), scale, losses)
            ctx.step_grads, ctx.one_ptr = (grads[:M], grads[M:].view(M, 3)), one.data_ptr()
            ctx.live_holder, ctx.step_live = live_holder, flags
            if flags is not None:
                live_holder["flags"], live_holder["consume"] = flags, True
                live_holder["last"] = flags.clone() if live_holder.get("keep_last") else None
            ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image, image_out, target, scale)
            ctx.consts = (float(bg), float(loss_mul))
            loss, scaled = losses[0], losses[1]
            ctx.mark_non_differentiable(image_out, depth_out, loss)
            ctx.set_materialize_grads(False)
            return image_out, depth_out, loss, scaled
        scratch = _tail_scratch(dev, (N + 255) // 256)
        check(lib.nerftex_composite_rays_train_forward(ptr(sigmas), ptr(rgbs), ptr(deltas), ptr(rays), M, N, ptr(weights_sum), ptr(depth), ptr(image), stream()))
        # live_holder (a dict the field's backward shares: Renderer.shade_train): this node's backward leaves one flag per 32 samples in it -- 0 = all 32
        # got exactly zero gradients -- for the MLP and hash-grid backward to skip; the flags are cleared by the render tail's launch
        ctx.live_holder, ctx.step_live = live_holder, None
        if live_holder is not None and M > 0:
            ctx.step_live = torch.empty((M + 31) // 32, dtype=torch.int32, device=dev)
        check(lib.nerftex_render_tail_forward_live(ptr(weights_sum), ptr(depth), ptr(image), ptr(nears), ptr(fars), ptr(target), float(bg), float(loss_mul), N,
                                                   ptr(image_out), ptr(depth_out), ptr(scratch[1]), ptr(scratch[0]), ptr(losses), ptr(scale),
                                                   losses.data_ptr() + 4, ptr(ctx.step_live), 0 if ctx.step_live is None else ctx.step_live.numel(), stream()))
        ctx.save_for_backward(sigmas, rgbs, deltas, rays, weights_sum, image, image_out, target, scale)
        ctx.consts = (float(bg), float(loss_mul))
        loss, scaled = losses[0], losses[1]
        ctx.mark_non_differentiable(image_out, depth_out, loss)
        ctx.set_materialize_grads(False)
        return image_out, depth_out, loss, scaled

    @staticmethod
    def backward(ctx, _gi, _gd, _gl, grad_scaled):
        sigmas, rgbs, deltas, rays, weights_sum, image, image_out, target, scale = ctx.saved_tensors
        if grad_scaled is None:
            return (None,) * 12
        if ctx.step_grads is not None and grad_scaled.data_ptr() == ctx.one_ptr and grad_scaled.numel() == 1:
            return (*ctx.step_grads, None, None, None, None, None, None, None, None, None, None)  # computed by the forward's launch
        bg, loss_mul = ctx.consts
        M, N = sigmas.shape[0], rays.shape[0]
        if N == 0 or M == 0:
            return torch.zeros_like(sigmas), torch.zeros_like(rgbs), None, None, None, None, None, None, None, None, None, None
        grad_scaled = grad_scaled.contiguous().float()
        # PRECONDITION of the uninitialised gradient buffers below: `rays` are the records of THIS library's march with the counter at zero
        # on entry (march_rays_train / march_rays_train_fresh: record n = ray n, offsets an exclusive prefix sum from 0), so that the rows past
        # rays[N-1].offset + count are exactly the rows no ray covers -- the kernel zeroes those.  Records from anywhere else (a non-zero
        # starting counter, another order): use raymarching.composite_rays_train + render_tail, whose backward takes zero-filled buffers.
        # rows the rays do not cover (the tail of a buffer sized by the mean count) get no gradient: zeros, like the reference's buffers --
        grads = torch.empty(4 * M, dtype=torch.float32, device=sigmas.device)  # (zeroed where no ray writes by the launch itself)
        grad_sigmas, grad_rgbs = grads[:M], grads[M:].view(M, 3)
        check(lib.nerftex_composite_tail_backward_live(ptr(grad_scaled), ptr(scale), loss_mul, ptr(image_out), ptr(target), bg, ptr(sigmas), ptr(rgbs), ptr(deltas),
                                                       ptr(rays), ptr(weights_sum), ptr(image), M, N, ptr(grad_sigmas), ptr(grad_rgbs), ptr(ctx.step_live), stream()))
        if ctx.step_live is not None and ctx.step_grads is None:
            ctx.live_holder["flags"] = ctx.live_holder["last"] = ctx.step_live  # ("last" stays for whoever wants to look: bench.py's dead-step fraction)
        return grad_sigmas, grad_rgbs, None, None, None, None, None, None, None, None, None, None


def composite_tail(sigmas, rgbs, deltas, rays, nears, fars, target, bg=1.0, loss_mul=1.0, scale=None, live_holder=None, one=None):
    """-> (image_out, depth_out, loss, scaled_loss): compositing, background blend, depth normalisation and MSE; one backward launch.
    live_holder: a dict shared with the fused field's backward (Renderer.shade_train, skip_dead_samples): the backward leaves its step flags there.
    one: the root-gradient tensor of the coming `scaled_loss.backward(one)` (a device 1.0): forward + backward become one launch."""
    return _composite_tail.apply(sigmas, rgbs, deltas, rays, nears, fars, target, bg, loss_mul, scale, live_holder, one)


_SCRATCH = {}


def _tail_scratch(dev, blocks):
    """(ticket uint32 [1] kept at zero by the kernel, partial sums float [blocks]) per device."""
    s = _SCRATCH.get(dev)
    if s is None or s[1].numel() < blocks:
        s = (torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(max(blocks, 1024), dtype=torch.float32, device=dev))
        _SCRATCH[dev] = s
    return s


def render_tail(weights_sum, depth, image, nears, fars, target, bg=1.0, loss_mul=1.0, scale=None):
    """-> (image_out, depth_out, loss, scaled_loss); call backward on scaled_loss (== loss when scale is None)."""
    return _render_tail.apply(weights_sum, depth, image, nears, fars, target, bg, loss_mul, scale)
