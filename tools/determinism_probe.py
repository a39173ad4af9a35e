#!/usr/bin/env python3
"""Debug probe (GPU): exact checksums after every stage of a few eager training steps, and -- round 4 -- the instruments that found out why
two-rank runs on one GPU differed from run to run.  Plain use: run several copies concurrently (or under torchrun with
NERFTEX_DP_SHARE_GPU=1: the two-ranks-on-one-GPU rig) and diff gpurun_out/detprobe/run_<tag>.txt.  Environment switches:

  STEPS=n            steps (default 10); the first two size their buffers by the count, later ones use the fresh-ray march
  RECHECK=1          after every backward, recompute the table gradient three times from the tapped inputs (ngp_harness.fused.DEBUG_TAP) and
                     say how many elements of autograd's result differ; then a second SUM over the scratch autograd's launch left
                     (phase 2 only: same wrong bits => the fill kernel wrote wrong records); PHASE_REPS / AGAIN_REPS repeat the
                     bin-once-sum-twice and the one-call forms; FRESH_OUT=1|2 gives every repetition freshly allocated output memory
  SIDE_AGAIN=1       the recomputations run on another stream (their own scratch set): when autograd's launch is the odd one out its
                     scratch is still intact and is compared with a good launch's -- directory words, then the record regions as multisets,
                     then which samples of the chunk the odd records belong to (nerftex_debug_workspace)
  SNAPSHOT=1         the same comparison from copies taken right after the backward (perturbs the timing: the failure went away)
  DUMP=prefix        torch.save the table gradient and the tapped inputs of step DUMP_STEP; DUMP_FAIL=1: of the first failing steps
  TIMESYNC=seconds   processes without a collective between them start their steps on a common wall-clock grid
  USE_STREAM=1       everything on an explicit stream instead of the legacy null stream; EXTRA_STREAMS=1: staged copies on a pool stream
  READY_FILE=path    touch it at step 2 (tools/*_concurrency_probe.py use this script as the neighbour that keeps the GPU busy)
  LOSS_MUL, NO_ATTACH, SYNC_AFTER, RELEASE: single switches of the data-parallel step, for bisecting"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))
import torch  # noqa: E402


def cs(t):
    t = t.detach()
    if t.dtype in (torch.float16, torch.bfloat16):
        v = t.view(torch.int16).to(torch.int64)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32).to(torch.int64)
    else:
        v = t.to(torch.int64)
    v = v.reshape(-1)
    w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
    return int((v * w).sum().item())


def main():
    from ngp_harness import scene
    from ngp_harness.model import NGPField, Renderer
    from ngp_harness.optim import FusedAmp, HalfLeafAdam

    tag = sys.argv[1] if len(sys.argv) > 1 else "x"
    dev = torch.device("cuda:0")
    from ngp_harness import dp

    rank, world, _ = dp.init_from_env()  # (torchrun + NERFTEX_DP_SHARE_GPU=1: the two-ranks-on-one-GPU rig)
    tag = f"{tag}_r{rank}" if world > 1 else tag
    sc = scene.Scene(bound=2.0, seed=0)
    grid, _, _ = sc.bitfield()
    torch.manual_seed(0)
    field = NGPField(bound=2.0, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    r = Renderer(field, bound=2.0, min_near=0.2).to(dev)
    r.set_occupancy(torch.from_numpy(grid).to(dev))
    opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")], lr=1e-2, betas=(0.9, 0.99), eps=1e-15)
    amp = FusedAmp(opt)
    if world == 1 and not os.environ.get("NO_ATTACH"):
        amp.attach(field.encoder)
    reducer = dp.FlatGradAllReduce(opt.trainable(), average=False, big_comm_dtype=torch.float16, big_numel=0)
    one = torch.ones((), device=dev)
    pool = []
    n_global = 8192 * world
    lo, hi = dp.shard(n_global, rank, world)
    for k in range(4):
        o, d = scene.train_batch(n_global, seed=100 + k, n_views=4)
        pool.append((torch.from_numpy(o[lo:hi]).to(dev), torch.from_numpy(d[lo:hi]).to(dev)))
    gt = torch.rand(4, n_global, 3, generator=torch.Generator().manual_seed(4321))[:, lo:hi].contiguous().to(dev)
    lines = []
    tapped = {}
    dumped = [0]
    if os.environ.get("DUMP") or os.environ.get("RECHECK"):
        from ngp_harness import fused

        fused.DEBUG_TAP = lambda **kw: tapped.update({k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in kw.items()})
    if os.environ.get("USE_STREAM"):  # everything on an explicit stream instead of the legacy null stream
        work_stream = torch.cuda.Stream()
        work_stream.wait_stream(torch.cuda.current_stream())
        torch.cuda.set_stream(work_stream)
    side_again = torch.cuda.Stream() if os.environ.get("SIDE_AGAIN") else None
    side = torch.cuda.Stream() if os.environ.get("EXTRA_STREAMS") else None  # what gloo's CUDA all-reduce does, without gloo: staged copies on a pool stream
    pinned = torch.empty(1 << 24, dtype=torch.float16).pin_memory() if side is not None else None
    for step in range(int(os.environ.get("STEPS", "10"))):
        ro, rd = pool[step % 4]
        for leaf in opt.leaves:
            leaf.grad = None
        if step == 2 and os.environ.get("READY_FILE"):
            torch.cuda.synchronize()
            open(os.environ["READY_FILE"], "w").write("running\n")
        if os.environ.get("TIMESYNC"):
            import time

            torch.cuda.synchronize()
            q = float(os.environ["TIMESYNC"])
            target = (int(time.time() / q) + 1) * q
            while time.time() < target:
                pass
        with torch.autocast("cuda", dtype=torch.float16):
            marched, counter = r.march_train(ro, rd, dt_gamma=1 / 128, perturb=True, mean_count=462848 if step >= 2 else None)
            nears, fars, xyzs, dirs, deltas, rays = marched
            lines.append(f"{step} march xyzs {cs(xyzs)} deltas {cs(deltas)} rays {cs(rays)} counter {counter.tolist()}")
            image, depth, loss, scaled = r.shade_train(marched, 1, target=gt[step % 4], loss_mul=float(os.environ.get("LOSS_MUL", 1.0 / world)), scale=amp.scale)
            lines.append(f"{step} fwd image {cs(image)} loss {cs(loss.reshape(1))} scaled {cs(scaled.reshape(1))}")
        scaled.backward(one)
        g = [leaf.grad for leaf in opt.leaves]
        lines.append(f"{step} bwd table {cs(g[0])} sigma {cs(g[1])} color {cs(g[2])} found {float(amp.found_inf)}")
        if os.environ.get("RECHECK") and tapped:
            import nerftex_hip
            from nerftex_hip import F16, LAYOUT_BLC, LAYOUT_GRAD_OVERWRITE, check, lib

            def again(prezero=False):
                if os.environ.get("SIDE_AGAIN"):  # on another stream: its own scratch set -- the scratch autograd's launch left stays as it is
                    torch.cuda.synchronize()
                    with torch.cuda.stream(side_again):
                        out_ = again_here(prezero)
                    return out_
                return again_here(prezero)

            def again_here(prezero=False):
                enc = field.encoder
                out = torch.empty_like(g[0])
                if prezero:
                    out.zero_()
                    torch.cuda.synchronize()
                dummy = torch.empty(1, dtype=torch.float16, device=dev)
                S, H, gridtype, align, affine = tapped["meta"]
                gx, x = tapped["grad_x"], tapped["x"]
                check(lib.nerftex_grid_encode_backward_affine(gx.data_ptr(), x.data_ptr(), enc.embeddings.data_ptr(), enc.offsets.data_ptr(), out.data_ptr(), x.shape[0],
                                                              3, 2, 16, S, H, 0, dummy.data_ptr(), dummy.data_ptr(), gridtype, align, F16,
                                                              LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0], affine[1], nerftex_hip.stream()))
                torch.cuda.synchronize()
                return out

            def tick():  # TIMESYNC: processes without a collective between them start their launches together (wall-clock grid)
                if os.environ.get("TIMESYNC"):
                    import time

                    torch.cuda.synchronize()
                    q = float(os.environ["TIMESYNC"])
                    target = (int(time.time() / q) + 1) * q
                    while time.time() < target:
                        pass

            def phases():  # bin once, sum twice: do two sums over the SAME records agree?
                enc = field.encoder
                S, H, gridtype, align, affine = tapped["meta"]
                gx, x = tapped["grad_x"], tapped["x"]
                outs = [torch.empty_like(g[0]) for _ in range(2)]
                head = (gx.data_ptr(), x.data_ptr(), enc.embeddings.data_ptr(), enc.offsets.data_ptr())
                tail = (x.shape[0], 3, 2, 16, S, H, gridtype, align, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0], affine[1])
                tick()
                check(lib.nerftex_grid_encode_backward_phase(*head, outs[0].data_ptr(), *tail, 1, 0, 16, nerftex_hip.stream()))
                for o in outs:
                    check(lib.nerftex_grid_encode_backward_phase(*head, o.data_ptr(), *tail, 2, 0, 16, nerftex_hip.stream()))
                torch.cuda.synchronize()
                return tuple(int((o.view(torch.int16) != good.view(torch.int16)).sum()) for o in outs)

            def snapshot(handle=None):  # copies of the directory and the record regions the last hash-grid backward left (csrc/gridencoder_binned.hip's layout)
                import ctypes as C

                torch.cuda.synchronize()
                ptr_, bytes_ = C.c_void_p(), C.c_size_t()
                check(lib.nerftex_debug_workspace(5, nerftex_hip.stream() if handle is None else handle, C.byref(ptr_), C.byref(bytes_)))
                Bn = tapped["x"].shape[0]
                nch = (Bn + 1023) // 1024
                offs = field.encoder.offsets.tolist()
                part_tiles = 0
                for l in range(16):
                    nt = (offs[l + 1] - offs[l] + 4095) // 4096
                    sl = min(max((Bn * 4 // max(nt, 1) + 32767) // 32768, 1), nch)
                    part_tiles += nt * sl if sl > 1 else 0
                dir_bytes = (4 * 16 * 128 * nch + 255) // 256 * 256
                rec_off = dir_bytes + part_tiles * 65536
                rec_bytes = 8 * 16 * nch * 8704
                assert rec_off + rec_bytes <= bytes_.value, (rec_off, rec_bytes, bytes_.value)

                class Raw:
                    def __init__(self, p, n):
                        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "|u1", "data": (p, False), "version": 2}

                raw = torch.as_tensor(Raw(ptr_.value, rec_off + rec_bytes), device=dev)
                d_ = raw[: 4 * 16 * 128 * nch].view(torch.int32).reshape(16, 128, nch).clone()
                r_ = raw[rec_off: rec_off + rec_bytes].view(torch.int64).reshape(16 * nch, 8704).clone()
                return d_, r_, nch

            def compare_scratch(A, G):
                (da, ra, nch), (dg, rg, _) = A, G
                offs = field.encoder.offsets.tolist()
                msgs = []
                for l in range(16):
                    nt = (offs[l + 1] - offs[l] + 4095) // 4096
                    dd = (da[l, :nt] != dg[l, :nt])
                    if dd.any():
                        idx = dd.nonzero()
                        msgs.append(f"level {l}: {int(dd.sum())} directory words differ, first (tile, chunk) {idx[:6].tolist()} a {da[l, :nt][dd][:4].tolist()} g {dg[l, :nt][dd][:4].tolist()}")
                # record regions as multisets: slots past the region's block (directory: last tile's offset + count) are not part of it
                ends = ((dg >> 16) & 0xffff) + (dg & 0xffff)  # [16, 128, nch]
                total = ends.amax(1).reshape(-1)  # [16 * nch]
                slot = torch.arange(8704, device=dev)[None, :]
                live = slot < total[:, None]
                big = torch.iinfo(torch.int64).max
                sa = torch.where(live, ra, big).sort(1).values
                sg = torch.where(live, rg, big).sort(1).values
                bad_regions = (sa != sg).any(1).nonzero().reshape(-1)
                msgs.append(f"record regions whose multiset of records differs: {bad_regions.numel()} of {total.numel()}")
                for rgn in [r_ for r_ in bad_regions.tolist() if r_ // nch >= 5][:8]:
                    a_, g_ = ra[rgn][: int(total[rgn])], rg[rgn][: int(total[rgn])]
                    # slots of A whose record does not occur in G
                    missing = ~torch.isin(a_, g_)
                    pos = missing.nonzero().reshape(-1)
                    if rgn // nch >= 5 and len(msgs) < 40:
                        extra = (~torch.isin(g_, a_)).nonzero().reshape(-1)

                        def dec(v):
                            v = int(v)
                            w, gb = v & 0xffffffff, (v >> 32) & 0xffffffff
                            import numpy as np

                            h = np.array([gb & 0xffff, gb >> 16], dtype=np.uint16).view(np.float16)
                            return f"(row {w & 4095} code {(w >> 12) & 15} p16 {w >> 16} g {float(h[0]):.3e} {float(h[1]):.3e})"

                        # which samples of the chunk do the zeroed records belong to?  (pair records carry the x fraction: match (local row, p16))
                        import numpy as np

                        lvl, chk_ = rgn // nch, rgn % nch
                        S_, H_, _, al_, af_ = tapped["meta"]
                        xs_ = tapped["x"][chk_ * 1024:(chk_ + 1) * 1024].cpu().numpy().astype(np.float32)
                        gx_ = tapped["grad_x"][chk_ * 1024:(chk_ + 1) * 1024, 2 * lvl:2 * lvl + 2].float().cpu().numpy()
                        u_ = ((xs_ + np.float32(af_[0])) * np.float32(af_[1])).astype(np.float32)
                        scale_ = np.float32(np.exp2(np.float32(lvl * S_)) * H_ - 1.0)
                        pos_ = u_ * scale_ + np.float32(0.0 if al_ else 0.5)
                        pg_ = np.floor(pos_).astype(np.int64)
                        fr_ = (pos_ - pg_).astype(np.float32)
                        p16_ = np.minimum(65535, (fr_[:, 0] * np.float32(65536.0) + np.float32(0.5)).astype(np.int64))
                        size_ = offs[lvl + 1] - offs[lvl]
                        owners = {}
                        for q in range(4):
                            cy, cz = pg_[:, 1] + (q & 1), pg_[:, 2] + (q >> 1)
                            ra_ = ((pg_[:, 0].astype(np.uint32)) ^ (cy.astype(np.uint32) * np.uint32(2654435761)) ^ (cz.astype(np.uint32) * np.uint32(805459861))).astype(np.int64) % size_
                            for i_ in range(1024):
                                owners.setdefault((int(ra_[i_] % 4096), int(p16_[i_])), []).append(i_)
                        found = []
                        for v in a_[pos].tolist():
                            w_ = int(v) & 0xffffffff
                            if ((w_ >> 12) & 15) != 15:
                                found.extend(owners.get((w_ & 4095, w_ >> 16), [-1]))
                        found = sorted(set(found))
                        msgs.append(f"      samples (index inside the chunk) the zeroed PAIR records belong to: {found}; their gradient on this level in the tapped copy: "
                                    f"{[gx_[i_].tolist() for i_ in found if i_ >= 0][:6]}")
                        msgs.append("      wrong launch has: " + " ".join(dec(v) for v in a_[pos][:16].sort().values.tolist()))
                        msgs.append("      good launch has:  " + " ".join(dec(v) for v in g_[extra][:16].sort().values.tolist()))
                    msgs.append(f"   region level {rgn // nch} chunk {rgn % nch}: block of {int(total[rgn])} records, {pos.numel()} slots hold records the good launch does not have: "
                                f"slots {pos[:20].tolist()}{' ...' if pos.numel() > 20 else ''} last {int(pos[-1]) if pos.numel() else -1}")
                return msgs

            snap_auto = snapshot() if os.environ.get("SNAPSHOT") else None
            # before anything else runs the hash-grid backward again: a second SUM over the scratch autograd's launch left
            resum = torch.empty_like(g[0])
            if not os.environ.get("NO_RESUM"):
                encr = field.encoder
                Sr, Hr, gtr, alr, afr = tapped["meta"]
                check(lib.nerftex_grid_encode_backward_phase(tapped["grad_x"].data_ptr(), tapped["x"].data_ptr(), encr.embeddings.data_ptr(), encr.offsets.data_ptr(),
                                                             resum.data_ptr(), tapped["x"].shape[0], 3, 2, 16, Sr, Hr, gtr, alr, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE,
                                                             afr[0], afr[1], 2, 0, 16, nerftex_hip.stream()))
                torch.cuda.synchronize()
            tick()
            outs3 = [again() for _ in range(3)]
            a = [int((o.view(torch.int16) != g[0].view(torch.int16)).sum()) for o in outs3]
            # the reference for the phase checks: the majority of (autograd's, recomputed) results
            good = g[0] if a.count(0) >= 2 else (outs3[0] if torch.equal(outs3[0], outs3[1]) or torch.equal(outs3[0], outs3[2]) else outs3[1])
            if os.environ.get("DUMP_FAIL") and a.count(0) < 2 and dumped[0] < 2:  # autograd's launch is the odd one out: keep what an offline look needs
                dumped[0] += 1
                rows = (g[0].view(torch.int16) != good.view(torch.int16)).any(1).nonzero().reshape(-1)
                os.makedirs(os.path.join(ROOT, "gpurun_out", "detprobe"), exist_ok=True)
                torch.save({"rows": rows.cpu(), "wrong": g[0][rows].cpu(), "good": good[rows].cpu(), "x": tapped["x"].cpu(), "meta": tapped["meta"],
                            "grad_x_absmax": tapped["grad_x"].float().abs().amax(1).cpu().half(), "offsets": field.encoder.offsets.cpu()},
                           os.path.join(ROOT, "gpurun_out", "detprobe", f"fail_{tag}_{step}.pt"))
            if os.environ.get("SIDE_AGAIN") and a.count(0) < 2 and dumped[0] < 3:
                dumped[0] += 1
                chk = again()
                if torch.equal(chk, good):
                    lines.append(f"{step} scratch of autograd's (wrong) launch vs scratch of a good launch on the same inputs (another stream's scratch):")
                    lines.extend("      " + m for m in compare_scratch(snapshot(), snapshot(side_again.cuda_stream)))
            if snap_auto is not None and a.count(0) < 2 and dumped[0] < 3:
                dumped[0] += 1
                chk = again()
                if torch.equal(chk, good):
                    lines.append(f"{step} scratch of autograd's (wrong) launch vs scratch of a good launch on the same inputs:")
                    lines.extend("      " + m for m in compare_scratch(snap_auto, snapshot()))
            snap_auto = None
            if not os.environ.get("NO_RESUM"):
                r_good = int((resum.view(torch.int16) != good.view(torch.int16)).sum())
                r_auto = int((resum.view(torch.int16) != g[0].view(torch.int16)).sum())
                if r_good or r_auto:
                    lines.append(f"{step} resum: second sum over autograd's scratch differs from the good gradient in {r_good} elements, from autograd's in {r_auto}")
            del outs3
            ph = [phases() for _ in range(int(os.environ.get("PHASE_REPS", "6")))]
            # a failing one-call backward: is a second SUM over the scratch it left right (the records in memory are fine, the sum kernel read
            # or wrote something stale) or wrong in the same way (the fill kernel left wrong records)?
            enc = field.encoder
            S, H, gridtype, align, affine = tapped["meta"]
            gx, x = tapped["grad_x"], tapped["x"]
            head = (gx.data_ptr(), x.data_ptr(), enc.embeddings.data_ptr(), enc.offsets.data_ptr())
            tail = (x.shape[0], 3, 2, 16, S, H, gridtype, align, F16, LAYOUT_BLC | LAYOUT_GRAD_OVERWRITE, affine[0], affine[1])
            verdicts = []
            fresh_mode = int(os.environ.get("FRESH_OUT", "0"))  # 1: the output is freshly hipMalloc'ed memory; 2: ... written once before the backward
            for rep in range(int(os.environ.get("AGAIN_REPS", "24"))):
                if fresh_mode:
                    out = None
                    torch.cuda.empty_cache()
                if os.environ.get("TICK_EVERY", "1") == "1" or rep % 3 == 0:
                    tick()
                out = again(prezero=fresh_mode == 2)
                bad = int((out.view(torch.int16) != good.view(torch.int16)).sum())
                if bad:
                    re = torch.empty_like(out)
                    check(lib.nerftex_grid_encode_backward_phase(*head, re.data_ptr(), *tail, 2, 0, 16, nerftex_hip.stream()))
                    torch.cuda.synchronize()
                    verdicts.append((bad, int((re.view(torch.int16) != good.view(torch.int16)).sum()), int((re.view(torch.int16) != out.view(torch.int16)).sum())))
            if verdicts:
                lines.append(f"{step} failing one-call backwards (wrong elements, wrong elements of a second sum over the same scratch, second sum vs first): {verdicts}")
            lines.append(f"{step} recheck: elements differing from autograd's table gradient: again {a} phases (bin once, sum twice) {ph}")
        if os.environ.get("DUMP") and step == int(os.environ.get("DUMP_STEP", "1")):
            torch.save({"g": g[0].cpu(), "offsets": field.encoder.offsets.cpu(), "table": field.encoder.embeddings.detach().cpu(),
                        **{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in tapped.items()}}, f"{os.environ['DUMP']}_{tag}.pt")
        if side is not None:
            flat = g[0].reshape(-1)[: pinned.numel()]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                pinned.copy_(flat, non_blocking=True)
                side.synchronize()
                flat.copy_(pinned, non_blocking=True)
            torch.cuda.current_stream().wait_stream(side)
        if world > 1:
            reducer.all_reduce()
            if os.environ.get("SYNC_AFTER"):
                torch.cuda.synchronize()
                torch.distributed.barrier()
                torch.cuda.synchronize()
            g = [leaf.grad for leaf in opt.leaves]
            lines.append(f"{step} exchanged table {cs(g[0])} sigma {cs(g[1])} color {cs(g[2])}")
        amp.step()
        lines.append(f"{step} opt table {cs(opt.masters[0])} sigma {cs(opt.masters[1])} color {cs(opt.masters[2])} scale {float(amp.scale)}")
        if step == 1 and os.environ.get("RELEASE"):
            import nerftex_hip

            torch.cuda.synchronize()
            nerftex_hip.check(nerftex_hip.lib.nerftex_release_workspaces())
        if r.local_step == 16:
            r.update_mean_count()
    torch.cuda.synchronize()
    out = os.path.join(ROOT, "gpurun_out", "detprobe")
    os.makedirs(out, exist_ok=True)
    open(os.path.join(out, f"run_{tag}.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
