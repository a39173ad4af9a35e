#!/usr/bin/env python3
"""bench.py -- ray-samples/s (train) of the NeRF-Texture rendering hot path on MI355X, + rendered Mpix/s.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full pass of the hot path over one batch of synthetic rays, exactly the sequence of
nerf/utils.py:1011-1022 + nerf/renderer.py:338-425 of the reference:
    near_far_from_aabb -> march_rays_train -> hash-grid encode -> sigma MLP -> SH encode -> colour MLP ->
    composite_rays_train -> MSE loss -> backward through the same chain -> (N>1: one RCCL all-reduce of the flat
    gradient) -> GradScaler.step(Adam) -> every 16 steps the `mean_count` read-back of update_extra_state.
Default workload = BASELINE.json configs[2] at N=1 and configs[4] at N=8 (8192 rays per GPU -> 65536 rays/step): fox-style
scene, L=16 F=2 hash grid (T=2^19, desired 2048*bound, bound 2), all four native components on the HIP path
(gridencoder, raymarching, shencoder, ffmlp 2x64 + 3x64 on MFMA), 800x800 random-pose pixels, fp16 autocast like the
reference's `-O`.  `--mlp torch --rays 4096` is configs[1] ("MLP still PyTorch-ROCm"); at N=1 the other configuration
is also run briefly and reported under "other_config".  Inputs are resident in HBM before the timed region; data =
synthetic, weights = seeded random init.

Prints ONE JSON line (rank 0).  `value` = ray samples actually marched (sum over steps and ranks of the
march_rays_train counter) / max-over-ranks wall time of the K timed steps.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # CPU-baseline leg: idle OpenMP workers sleep instead of spinning (set before libgomp loads)

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "nerf-texture_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

L2_PEAK_BS = 34.5e12  # aggregate L2 bandwidth of the 8 XCDs (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured streaming ceiling
# Memory-side bytes per launch of the default workload's hash-grid ops (fp16 table, ~459 k samples), from separate rocprofv3 PMC
# passes (FETCH_SIZE, WRITE_SIZE; profiles/r05_pmc_grid.txt: the bench's own children over the replayed step).  Backward: 2 x FETCH_SIZE (the gfx950 correction of
# MI355X_MICROARCH.md for wide coalesced streams: here the 8-B record stream) + WRITE_SIZE.  Forward: FETCH_SIZE as reported
# (4-B gathers are uncalibrated, and gathers served by L2 never reach the counter) + WRITE_SIZE.  None = not collected.
TRAFFIC_BYTES_PER_LAUNCH = {"grid_encode_forward": 85.3e6, "grid_encode_backward": 800.8e6}  # (the FALLBACK only: the run measures them itself, rocprof_traffic)
TRAFFIC_PROFILE = "profiles/r06_pmc_grid.txt"
COMMITTED_STATS = "profiles/r06_kernel_stats.csv"  # rocprofv3 --kernel-trace --stats of this script: the fallback when no live profile can be taken
# device kernels behind each hash-grid C-ABI call (whichever of them ran)
GRID_KERNELS = {
    "grid_encode_forward": ("grid_forward_level_kernel", "level_major_to_rows_kernel", "grid_forward_kernel"),
    "grid_encode_backward": ("bin_fill_dir_kernel", "sum_tiles_dir_kernel", "sum_tiles_adam_kernel", "combine_tiles_kernel", "grad_to_level_major_kernel", "bin_count_kernel", "scan_tiles_kernel",
                             "scan_global_kernel", "bin_fill_kernel", "sum_tiles_kernel", "grid_backward_owner_kernel", "grid_backward_kernel",
                             "grid_input_backward_kernel"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=208, help="timed steps (default 208 = 13 rings of the 16-entry step-counter ring: ~0.12 s of device time)")
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--rays", type=int, default=8192, help="rays per batch PER GPU (weak scaling; 8 GPUs x 8192 = configs[4]'s 65536)")
    ap.add_argument("--mlp", choices=["torch", "ffmlp"], default="ffmlp")
    ap.add_argument("--no-other", action="store_true", help="skip the short run of the other single-GPU configuration")
    ap.add_argument("--no-fused-glue", action="store_true", help="run the ops between/after the two FFMLPs as framework ops (reference structure)")
    ap.add_argument("--no-fused-tail", action="store_true", help="background blend / depth / MSE as framework ops instead of one kernel per direction")
    ap.add_argument("--no-fused-opt", action="store_true", help="torch.optim.Adam(fused=True) on an fp32 table instead of ngp_harness.optim.TableAdam")
    ap.add_argument("--no-fused-amp", action="store_true", help="torch.amp.GradScaler instead of ngp_harness.optim.FusedAmp")
    ap.add_argument("--graph-split", action="store_true", help="1 GPU: use the two-graph form of the multi-GPU path (for testing it)")
    ap.add_argument("--graph-allreduce", action="store_true", help="N > 1: capture the gradient all-reduce INSIDE the step's graph (march | shade + backward + "
                    "all-reduce + optimizer) instead of launching it eagerly between two graphs; falls back to the default structure if the "
                    "backend cannot be captured.  Also NERFTEX_BENCH_GRAPH_ALLREDUCE=1.  Untested on RCCL from the 1-GPU boxes this was written on")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a captured HIP graph (1 GPU)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not record per-kernel hipEvent pairs in the timed region (no roofline)")
    ap.add_argument("--dtype", choices=["fp16", "bf16", "fp32"], default="fp16", help="fp16 = autocast like the reference's --fp16/-O; bf16 = "
                    "bf16 autocast with the FFMLPs on their bf16 kernels (BASELINE.json configs[2] names bf16)")
    ap.add_argument("--wire", choices=["fp16", "fp32"], default="fp16", help="N > 1: dtype of the table gradient on xGMI. fp16 = the gradient's own dtype "
                    "under autocast (half the bytes); fp32 = SURVEY 8(e)'s parity form: the cross-rank sum is formed in fp32 and rounded once")
    ap.add_argument("--no-perturb", action="store_true", help="march without the per-ray start jitter (it is seeded by the ray's index in the LOCAL batch, so "
                    "a sharded run and a single-rank run of the same global batch only see the same samples without it)")
    ap.add_argument("--bound", type=float, default=2.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--steps-per-graph", type=int, default=4, help="1 GPU, march-ahead: consecutive steps recorded into ONE graph (1, 2, 4, 8 or 16; the hand-over "
                    "between two graph launches idles the device for ~10 us).  Falls back to 1 when --steps is not a multiple of it")
    ap.add_argument("--no-lean-march", action="store_true", help="march-ahead: the count pass at its normal 88 registers instead of the 64-register build "
                    "(`march_lean` knob) that leaves more of the CU to the step's kernels it runs beside")
    ap.add_argument("--no-march-ahead", action="store_true", help="1 GPU: march inside the step's one graph instead of one step ahead on a second stream")
    ap.add_argument("--baked-pool", action="store_true", help="1 GPU: report the loop whose pool of ray batches is baked into the graphs as the headline (rounds 1-3) "
                    "instead of the fresh-ray loop through ngp_harness.accelerate")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--infer-slots", type=int, default=3, help="sample slots per iteration of the rendered frame, in units of N rays (reference: 1; 4 until the per-block "
                    "copy node of the graphed loop was replaced by a store of the compaction kernel: with cheaper block boundaries 3 renders faster, 90.5 vs 88 Mpix/s)")
    ap.add_argument("--infer-parts", type=int, default=3, help="ray ranges of the rendered frame, each on its own stream")
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays of the bounded CPU-baseline sample")
    ap.add_argument("--warm-seconds", type=float, default=0.5, help="untimed steps (beyond --warmup) until this much wall time has passed under load, right before "
                    "the timed region: the device's clocks after an idle phase")
    ap.add_argument("--no-replay-profile", action="store_true", help="skip the per-kernel timing of the REPLAYED step (a child run of this script under "
                    "rocprofv3 --kernel-trace --stats, ~40 s); roofline.avg_launch_ms then comes from the committed profile or from eager launches")
    ap.add_argument("--no-traffic-profile", action="store_true", help="skip the two PMC child runs (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, ~30 s each) "
                    "that measure roofline.traffic; the committed constant is then used and labelled")
    ap.add_argument("--pipeline-adam", type=int, default=0, help="1 GPU, the fresh-ray headline loop (accelerate): sum the table gradient in this many level groups "
                    "and run each group's Adam on a second stream while the next group is being summed (A/B, off by default: DESIGN.md 4.5)")
    ap.add_argument("--no-fused-table-update", action="store_true", help="1 GPU, the fresh-ray headline loop (accelerate): write the whole table gradient and update "
                    "the table with the streaming Adam launch (rounds 1-5) instead of applying Adam from the summing kernel's tiles (round 6)")
    ap.add_argument("--no-skip-dead-samples", action="store_true", help="walk every 32-sample step in the MLP / hash-grid backward instead of the steps the compositing "
                    "backward flagged as carrying a gradient (round 6; no difference on the headline's young field, see other_config 'trained state')")
    ap.add_argument("--no-fused-composite-step", action="store_true", help="compositing forward, render tail and their backward as the three launches of rounds 3-5 "
                    "instead of one (round 6: the forward's launch forms the loss gradient itself; same outputs and gradients bit for bit)")
    ap.add_argument("--trained-steps", type=int, default=1008, help="training steps against rendered targets of the analytic scene before the 'trained state' entry "
                    "of other_config is timed (0 = skip it)")
    ap.add_argument("--no-occupancy-timing", action="store_true", help="skip timing the every-16-steps occupancy-grid update (reported separately, SURVEY 8(d))")
    ap.add_argument("--allreduce-chunks", type=int, default=1, help="N > 1: exchange the table gradient as this many level-group chunks, each started as soon "
                    "as the backward has produced its rows (default 1 = one all-reduce after the backward)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, bits, n_rays):
    """The reference's path on the host's cores, as SURVEY.md 8(d) defines it (oracle/cpu_path.py, pinned against the reference's own
    Python by tests/test_reference_python_cpu.py), each leg on a BOUNDED sample of the workload:

      value            one training step of the --cuda_ray path (R1 + R6 march, G1, MLPs, S1, R8, loss, R9, MLP backward, G2; no optimizer):
                       oracle C (OpenMP over samples / rays / levels) for every native op + torch-CPU nn.Linear MLPs, ALL host threads;
      one_thread       the same on one thread;
      run_path         BASELINE.json configs[0], "the reference's pure-PyTorch CPU path": NeRFRenderer.run semantics (512 uniform samples per
                       ray, exp / cumprod compositing, main_nerf.py:29-30 defaults) on rays of a 400 x 400 view, inference, all threads."""
    from ngp_harness import scene
    from oracle import cpu_path
    from oracle import oracle as orc

    cores = os.cpu_count() or 1
    b = args.bound
    field = cpu_path.Field(bound=b, mlp="linear")  # nerf/network.py: bias-free nn.Linear MLPs, fp32 ("MLP still PyTorch")
    r = cpu_path.Renderer(field, bound=b, min_near=0.2)
    r.density_bitfield = torch.from_numpy(np.ascontiguousarray(bits))
    o, d = scene.train_batch(n_rays, seed=1234)
    ro, rd = torch.from_numpy(o), torch.from_numpy(d)

    def train_step():
        field.zero_grad(set_to_none=True)
        image, _, counter = r.run_cuda_train(ro, rd, dt_gamma=1 / 128, perturb=True)
        torch.nn.functional.mse_loss(image, torch.full_like(image, 0.5)).backward()
        return int(counter[0])

    def timed(fn, budget_s, max_reps):
        fn()  # warm caches / page in the table
        t0, total, reps = time.perf_counter(), 0, 0
        while time.perf_counter() - t0 < budget_s and reps < max_reps:
            total += fn()
            reps += 1
        return total / (time.perf_counter() - t0), total // max(reps, 1), reps

    def use(threads):
        torch.set_num_threads(threads)
        orc.set_threads(threads)

    # "all cores" = the thread count that is fastest on this host: a 70 k-sample step does not feed 256 threads (barrier and wake-up
    # costs grow with the team); one probe step per candidate, the best one is timed
    probe = {}
    for threads in sorted({cores, max(1, cores // 2), max(1, cores // 4), min(cores, 64), min(cores, 32), min(cores, 16)}, reverse=True):
        use(threads)
        train_step()
        t0 = time.perf_counter()
        train_step()
        probe[threads] = time.perf_counter() - t0
    best = min(probe, key=probe.get)
    legs = {}
    for name, threads, budget in (("all", best, 6.0), ("one", 1, 5.0)):
        use(threads)
        legs[name] = timed(train_step, budget, 50)
    use(best)
    field.eval()
    pose = scene.rand_poses(1, 2.0, np.random.default_rng(7))[0]
    o4, d4 = scene.get_rays(pose, scene.intrinsics(400, 400), 400, 400)
    n_run = 4096  # rows 190..200 of the 400 x 400 view: rays through the middle of the scene
    sl = slice(190 * 400, 190 * 400 + n_run)
    r4o, r4d = torch.from_numpy(np.ascontiguousarray(o4[sl])), torch.from_numpy(np.ascontiguousarray(d4[sl]))

    def run_frame_part():
        with torch.no_grad():
            return r.run(r4o, r4d, num_steps=512, upsample_steps=0)[2]

    run_rate, run_samples, run_reps = timed(run_frame_part, 8.0, 20)
    torch.set_num_threads(max(1, cores // 2))
    return {
        "value": legs["all"][0], "unit": "ray-samples/s", "cores": best, "kind": "port",
        "sample": f"{legs['all'][2]} training steps (forward + backward, no optimizer) of {n_rays} rays / ~{legs['all'][1]} marched samples each; "
                  f"oracle C with OpenMP + torch-CPU nn.Linear MLPs on {best} threads -- the fastest of "
                  f"{ {k: round(v, 3) for k, v in probe.items()} } (threads: seconds per step) on a host with {cores} hardware threads",
        "one_thread": {"value": legs["one"][0], "cores": 1, "sample": f"{legs['one'][2]} of the same steps on one thread"},
        "run_path": {"value": run_rate, "unit": "ray-samples/s", "cores": best, "workload": "BASELINE.json configs[0]: NeRFRenderer.run semantics, "
                     "512 uniform samples per ray, nn.Linear MLPs, 400x400 view, inference", "sample": f"{run_reps} x {n_run} rays of the view "
                     f"({run_samples} samples each); a whole 400 x 400 frame is {160000 * 512} samples",
                     "s_per_400x400_frame": 160000 * 512 / run_rate},
    }


# ----------------------------------------------------------------------------------------------------- training leg
def measure_training(args, mlp, rays, steps, warmup, dev, rank, world, sc, grid, bits, time_grid_kernels, graph=False, dtype=None, dropin_only=False):
    """K timed training steps of one configuration. Returns (result dict, field, renderer).
    dtype: "fp16" | "bf16" | "fp32" (default args.dtype).  bf16 = bf16 autocast with the FFMLPs on their bf16 kernels (the hash table is
    narrowed to fp16 under any autocast, gridencoder/grid.py:41), GradScaler (the table gradient is fp16), torch's fused Adam.
    dropin_only: the reference's callers unchanged -- eager launches, no field glue / render tail kernels, torch.optim.Adam + GradScaler."""
    dtype = dtype or args.dtype
    no_ext = dropin_only
    import nerftex_hip
    from ngp_harness import dp, scene
    from ngp_harness.model import NGPField, Renderer

    torch.manual_seed(0)
    # (bf16: NGPField has no bf16 glue kernels -- fused_glue then selects the one-kernel bf16 field, nerftex_field_*_bf16, round 5)
    field = NGPField(bound=args.bound, mlp=mlp, fused_glue=not (args.no_fused_glue or no_ext),
                     mlp_dtype=torch.bfloat16 if dtype == "bf16" else torch.float16).to(dev)
    torch.manual_seed(1)  # FFMLP.reset_parameters reseeds with 42; give the table its own stream
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    renderer = Renderer(field, bound=args.bound, min_near=0.2, density_thresh=10.0).to(dev)
    renderer.set_occupancy(torch.from_numpy(grid).to(dev))
    assert np.array_equal(renderer.density_bitfield.cpu().numpy(), bits), "packbits parity (HIP vs numpy)"

    # inputs resident in HBM before anything is timed: a pool of ray batches (this rank's shard) + target colours
    n_pool = 8
    n_global = rays * world
    pool = []
    for k in range(n_pool):
        o, d = scene.train_batch(n_global, seed=100 + k, n_views=4)
        lo, hi = dp.shard(n_global, rank, world)
        pool.append((torch.from_numpy(o[lo:hi]).to(dev), torch.from_numpy(d[lo:hi]).to(dev)))
    lo, hi = dp.shard(n_global, rank, world)  # the targets belong to the rays: a global tensor, sharded like them
    gt = torch.rand(n_pool, n_global, 3, generator=torch.Generator().manual_seed(4321))[:, lo:hi].contiguous().to(dev)

    # same optimizer as main_nerf.py:128 (Adam, betas (0.9, 0.99), eps 1e-15); fused=True keeps GradScaler.step free of its
    # per-step found_inf .item() read-back (the unscale / skip-on-inf logic runs inside the fused kernel instead)
    # 1 GPU: the whole step is one graph.  N GPUs: RCCL inside a captured graph is not something this round could test, so the step
    # is two graphs (forward + backward | optimizer) with the gradient all-reduce launched eagerly between them.
    use_graph = graph
    split_graph = use_graph and (world > 1 or args.graph_split)
    # opt-in A/B for the 8-GPU node: the all-reduce as nodes of the step's graph (one graph per step, nothing launched from the host between
    # backward and optimizer).  `capture` falls back to the split structure when the backend refuses to be captured.
    ar_state = {"in_graph": bool(split_graph and world > 1 and (args.graph_allreduce or os.environ.get("NERFTEX_BENCH_GRAPH_ALLREDUCE") == "1"))}
    if ar_state["in_graph"]:
        import torch.distributed as dist

        if dist.get_backend() != "nccl":  # gloo moves device tensors through the host with stream synchronisations (and aborts the process in a capture)
            print(f"[bench] --graph-allreduce needs the nccl (RCCL) backend, not {dist.get_backend()}: using the split structure", file=sys.stderr)
            ar_state["in_graph"] = False
            ar_state["fallback"] = f"--graph-allreduce needs the nccl (RCCL) backend, this group is {dist.get_backend()}: eager exchange between two graphs"
    # 1 GPU: the march of step k+1 needs nothing from step k (rays and the occupancy grid, not the weights): it is its own graph, replayed
    # on a second stream while step k shades, goes backward and updates -- latency-bound work on otherwise idle issue slots
    march_ahead = use_graph and not split_graph and not args.no_march_ahead
    # steps per replayed graph (march-ahead mode): the marches of a group run on the second stream under the PREVIOUS group
    group = 1
    if march_ahead and args.steps_per_graph in (1, 2, 4, 8, 16):
        group = args.steps_per_graph
        while steps % group:  # the timed region must end on a graph boundary
            group //= 2
    use_amp = dtype in ("fp16", "bf16")
    amp_dtype = torch.bfloat16 if dtype == "bf16" else torch.float16
    fused_opt = dtype in ("fp16", "bf16") and mlp == "ffmlp" and not (args.no_fused_opt or no_ext)
    fused_tail = not (args.no_fused_tail or no_ext)
    fused_amp = fused_opt and not args.no_fused_amp
    dp.broadcast([p.data for p in field.parameters()])
    if fused_opt:  # same Adam; every parameter's fp16 copy is the autograd leaf, its fp16 gradient consumed as produced (ngp_harness/optim.py)
        from ngp_harness.optim import FusedAmp, HalfLeafAdam

        mlp_dt = torch.bfloat16 if dtype == "bf16" else torch.float16  # (the table's copy and gradient are fp16 either way)
        opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights", mlp_dt), (field.color_net, "weights", mlp_dt)], lr=1e-2,
                           betas=(0.9, 0.99), eps=1e-15)
        trainable = opt.trainable()
    else:
        opt = torch.optim.Adam(field.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15, fused=True, capturable=use_graph)
        trainable = list(field.parameters())
    # the hash-table gradient crosses xGMI as fp16 (it is fp16-valued under autocast): half the all-reduce bytes
    # (the 1/world of the gradient average is folded into the loss below, so the exchange is a plain sum: no division pass over 48 MB)
    wire_dtype = torch.float32 if args.wire == "fp32" else (torch.float16 if dtype == "fp16" else None)
    reducer = dp.FlatGradAllReduce(trainable, average=False, big_comm_dtype=wire_dtype,
                                   big_numel=0 if fused_opt else 1 << 20)
    # N > 1, --allreduce-chunks k > 1: the table gradient is finished and exchanged level group by level group (dp.TableGradChunks)
    chunker = None
    if world > 1 and args.allreduce_chunks > 1 and fused_opt and field.fused_field:
        chunker = dp.TableGradChunks(field.encoder, args.allreduce_chunks)
    elif world > 1 and args.allreduce_chunks > 1:  # (said in the JSON line, not only here: config.collective.fallbacks)
        ar_state["chunk_fallback"] = ("--allreduce-chunks needs the fused optimizer path and the fused fp16 field (FFMLP, --dtype fp16 or bf16, no --no-fused-opt): "
                                      "one exchange of the whole table gradient per step")

    def exchange(grads=None, state=None):
        """One step's gradient exchange, start to finish (grads / state: the tensors and the per-group work of a captured backward)."""
        if chunker is None:
            reducer.all_reduce(grads=grads)
            return
        g = list(grads) if grads is not None else reducer.big_grads()
        hs = []
        for i in range(len(chunker)):
            chunker.sum_chunk(i, state)  # rows of level group i are final ...
            hs.append(reducer.start_tensor(chunker.view(i, g[0], state)))  # ... and go on the wire while the next group is summed
        h2 = reducer.all_reduce_start([None] + g[1:])  # the two MLPs' gradients
        reducer.finish_tensors(hs)
        reducer.all_reduce_finish(h2)
    inv_world = 1.0 / world
    # loss scaling: GradScaler's rules either way; with the fused optimizer its device side is three launches (optim.FusedAmp)
    amp = FusedAmp(opt) if fused_amp else None
    renderer.skip_dead_samples = bool(fused_opt and getattr(field, "fused_field", False) and not args.no_skip_dead_samples)  # (round 6; as accelerate() does)
    fused_table_update = False
    if amp is not None and world == 1 and field.fused_field:
        amp.attach(field.encoder)  # the non-finite scan rides on the kernels that write the gradients (N > 1: the scan must see the cross-rank sum)
        if not args.no_fused_table_update:  # round 6: the table's rows updated by the owners of their final gradient (double-buffered state), as accelerate() does
            amp.fuse_table_update(field.encoder)
            fused_table_update = True
    # bf16 keeps the loss scaler: the hash table -- and so its gradient -- is fp16 under ANY autocast (gridencoder/grid.py:41), and unscaled
    # gradients of ~1e-6 sit in fp16's subnormal range (measured: 38 % L1 error of the table gradient without scaling, tools/precision_table.py)
    scaler = None if fused_amp else torch.amp.GradScaler("cuda", enabled=dtype in ("fp16", "bf16"))
    one = torch.ones((), dtype=torch.float32, device=dev)  # root gradient, so that autograd does not fill one per step
    renderer.root_one = one if (fused_amp and not args.no_fused_composite_step) else None  # (round 6; as accelerate() does: compositing + tail + backward = one launch)
    renderer.defer_step_loss = renderer.root_one is not None  # (... and the loss finished by the field's backward: nobody reads it before)

    def march(ro, rd, **kw):
        with torch.autocast("cuda", dtype=amp_dtype, enabled=use_amp):
            return renderer.march_train(ro, rd, dt_gamma=dt_gamma, perturb=not args.no_perturb, max_steps=1024, **kw)

    def forward_backward(ro, rd, tgt, marched=None, **kw):
        """One training render + loss + backward; marched = (sample tensors, counter) of an earlier `march` of the same rays, or None."""
        marched, counter = march(ro, rd, **kw) if marched is None else marched
        with torch.autocast("cuda", dtype=amp_dtype, enabled=use_amp):
            if fused_tail:
                image, depth, loss, scaled = renderer.shade_train(marched, 1, target=tgt, loss_mul=inv_world, scale=amp.scale if amp else None)
            else:
                image, depth = renderer.shade_train(marched, 1)
                scaled = torch.nn.functional.mse_loss(image, tgt)
                if world > 1:
                    scaled = scaled * inv_world
                if amp:
                    scaled = amp.scale_loss(scaled)
        if amp:
            scaled.backward(one)
        else:
            scaler.scale(scaled).backward()
        return counter

    def optimizer_step():
        if amp:
            amp.step()
        else:
            scaler.step(opt)
            scaler.update()
    # samples marched in the timed region: the step-counter ring already holds every step's count and is read back every 16 steps (the reference's
    # own mean_count read-back): the ring sums are added up on the host there -- no counting kernel inside the step
    counted = {"rings": 0}

    def ring_end():
        renderer.update_mean_count()
        counted["rings"] += renderer.last_ring_samples
        renderer.mean_count = dp.all_reduce_max_int(renderer.mean_count, dev)

    def partial_ring():  # the executed steps of the ring in progress (a read-back: outside the timed region only)
        n = renderer.local_step
        return int(renderer.step_counter[:n, 0].sum().item()) if n else 0
    dt_gamma = 1 / 128
    field.train()

    def train_step(k, count=True):
        ro, rd = pool[k % n_pool]
        reducer.zero_grad()
        counter = forward_backward(ro, rd, gt[k % n_pool])
        exchange()
        optimizer_step()
        if renderer.local_step == 16:  # update_extra_state cadence (nerf/utils.py:1011): mean_count read-back
            ring_end()

    # ---- the same step as ONE replayed HIP graph: the eager loop is bound by the host's launch rate (the same kernels took 1.60 or
    # 1.83 ms per step depending on the box's host); a graph takes the host out of it.  Sixteen graphs, one per slot of the
    # step-counter ring, each with its ray batch and its counter slot baked in; the sample buffers are sized by a fixed count (the
    # ring's mean rounded up to 4096 + 4096); all graphs share one memory pool (they never run concurrently).
    RING = 16  # = the renderer's step-counter ring: graph g is step g of a 16-step cycle
    gstate = {"graphs": None, "M": 0, "marched": -1}
    from ngp_harness.streams import side_stream as shared_side_stream

    side_stream = shared_side_stream(dev) if march_ahead else None  # high priority; ONE per process (ngp_harness/streams.py: a later-created one may share a hardware queue)

    def body_march(g):
        renderer.local_step = g  # the step's counter is ring slot g, exactly as in the eager loop
        return march(*pool[g % n_pool], mean_count=gstate["M"])

    def body_fb(g, marched=None):
        reducer.zero_grad()
        forward_backward(*pool[g % n_pool], gt[g % n_pool], marched=body_march(g) if marched is None else marched)

    def body_opt():
        optimizer_step()

    def capture():
        from ngp_harness.streams import capture_section

        with capture_section():  # (no collector run in the middle of a recording: streams.py)
            capture_body()

    def capture_body():
        gstate["M"] = (renderer.mean_count + 4095) // 4096 * 4096 + 4096
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for g in range(3):  # allocator / library workspaces at this size, outside the capture
                body_fb(g)
                exchange()
                body_opt()
        torch.cuda.current_stream().wait_stream(side)
        graphs, mem, marches = [], None, [None] * RING
        # thread_local: only this thread's calls can invalidate a capture (an RCCL watchdog thread may query events meanwhile)
        if split_graph or march_ahead:  # the march of a step is its own graph: replayed while the PREVIOUS step's gradients are on the wire
            mem_m = None                # (N > 1) or while the previous step runs (1 GPU: own memory pool, the two graphs run concurrently)
            for g in range(RING):
                gm = torch.cuda.CUDAGraph()
                # (march_lean: the count pass compiled for 64 registers -- slower alone, but the step's kernels keep twice the occupancy beside it)
                with nerftex_hip.tune(march_lean=int(march_ahead and not args.no_lean_march)):
                    with torch.cuda.graph(gm, pool=mem_m if march_ahead else mem, capture_error_mode="thread_local"):
                        out = body_march(g)
                if march_ahead:
                    mem_m = gm.pool()
                else:
                    mem = gm.pool()
                marches[g] = (gm, out)  # the sample tensors stay alive: graph A of the slot reads them
        if march_ahead and group > 1:  # `group` consecutive steps per graph (their marches are the per-slot graphs above)
            groups = []
            for g0 in range(0, RING, group):
                gg = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gg, pool=mem, capture_error_mode="thread_local"):
                    for g in range(g0, g0 + group):
                        body_fb(g, marches[g][1])
                        body_opt()
                mem = gg.pool()
                groups.append(gg)
            gstate["graphs"] = [(marches[g][0], None, None, None) for g in range(RING)]
            gstate["groups"] = groups
            gstate["marches"], gstate["pool"] = marches, mem
            gstate["marched"] = -1
            renderer.local_step = 0
            return
        for g in range(RING):  # one graph per ring slot: static ray batch, static counter slot -> nothing to select or copy per step
            ga = torch.cuda.CUDAGraph()
            if ar_state["in_graph"]:
                try:
                    with torch.cuda.graph(ga, pool=mem, capture_error_mode="thread_local"):
                        body_fb(g, marches[g][1])
                        exchange()
                        body_opt()
                    mem = ga.pool()
                    graphs.append((marches[g][0], ga, "in_graph", None))
                    continue
                except Exception as e:  # noqa: BLE001 -- this backend's collectives cannot be captured: the default structure
                    print(f"[bench] all-reduce inside the graph failed ({type(e).__name__}: {e}); using the split structure", file=sys.stderr)
                    ar_state["in_graph"] = False
                    ar_state["fallback"] = f"--graph-allreduce: the collective could not be captured ({type(e).__name__}: {str(e)[:200]}): eager exchange between two graphs"
                    ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga, pool=mem, capture_error_mode="thread_local"):
                body_fb(g, marches[g][1] if (split_graph or march_ahead) else None)
                if not split_graph:
                    body_opt()
            mem = ga.pool()
            gm, gb, grads = None, None, None
            if march_ahead:
                gm = marches[g][0]
            if split_graph:
                gm = marches[g][0]
                grads = reducer.big_grads()  # this graph's gradient tensors: A writes them, the all-reduce and B read them -- kept alive
                if chunker is not None:  # A only binned the table's contributions: one small graph per level group finishes its rows
                    state = chunker.take()
                    sums = []
                    for i in range(len(chunker)):
                        gs_ = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gs_, pool=mem, capture_error_mode="thread_local"):
                            chunker.sum_chunk(i, state)
                        mem = gs_.pool()
                        sums.append(gs_)
                    reducer.all_reduce(grads=grads)  # (as below: the collective sees these buffers once before the timed region; their contents are not used)
                    grads = (grads, state, sums)
                else:
                    reducer.all_reduce(grads=grads)
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb, pool=mem, capture_error_mode="thread_local"):
                    body_opt()
            graphs.append((gm, ga, gb, grads))
        gstate["graphs"] = graphs
        gstate["marched"] = -1
        renderer.local_step = 0

    def graph_step(k):
        g = renderer.local_step
        gm, ga, gb, grads = gstate["graphs"][g]
        if march_ahead and group > 1:  # steps g .. g + group - 1 as one graph; the next group's marches on the side stream beside it
            if g % group == 0:
                main = torch.cuda.current_stream()
                if gstate["marched"] < g + group - 1:  # (first group after a capture: nothing marched ahead)
                    side_stream.wait_stream(main)
                    with torch.cuda.stream(side_stream):
                        for s_ in range(max(gstate["marched"] + 1, g), g + group):
                            gstate["graphs"][s_][0].replay()
                    gstate["marched"] = g + group - 1
                main.wait_stream(side_stream)
                if g + group < RING:  # not across the ring's end: the mean_count read-back comes first there
                    with torch.cuda.stream(side_stream):
                        for s_ in range(g + group, g + 2 * group):
                            gstate["graphs"][s_][0].replay()
                    gstate["marched"] = g + 2 * group - 1
                gstate["groups"][g // group].replay()
        elif march_ahead:  # march(g + 1) on the side stream under shade + backward + optimizer(g) on this one
            main = torch.cuda.current_stream()
            if gstate["marched"] != g:  # first step of a ring: nothing marched ahead
                gm.replay()
                side_stream.wait_stream(main)  # the marches share scratch: in order
            else:
                main.wait_stream(side_stream)
            if g + 1 < RING:  # not across the ring's end: the mean_count read-back (and a possible re-capture) comes first there
                with torch.cuda.stream(side_stream):
                    gstate["graphs"][g + 1][0].replay()
                gstate["marched"] = g + 1
            ga.replay()
        elif gb is None:
            ga.replay()
        elif gb == "in_graph":  # march(g) | shade + backward + all-reduce + optimizer(g); the next march is replayed behind it
            if gstate["marched"] != g:
                gm.replay()
            ga.replay()
            if g + 1 < RING:
                gstate["graphs"][g + 1][0].replay()
                gstate["marched"] = g + 1
        else:  # march(g) | shade + backward(g) | all-reduce(g) overlapped with march(g + 1) | optimizer(g)
            if gstate["marched"] != g:
                gm.replay()
            ga.replay()
            if chunker is not None:
                g_list, state, sums = grads
                hs = []
                for i, gs_ in enumerate(sums):
                    gs_.replay()  # rows of level group i are final ...
                    hs.append(reducer.start_tensor(chunker.view(i, g_list[0], state)))  # ... and on the wire while group i + 1 is summed
                handle = reducer.all_reduce_start([None] + list(g_list[1:]))
            else:
                handle = reducer.all_reduce_start(grads)
            if g + 1 < RING:  # not across the ring's end: the mean_count read-back (and a possible re-capture) comes first there
                gstate["graphs"][g + 1][0].replay()
                gstate["marched"] = g + 1
            if chunker is not None:
                reducer.finish_tensors(hs)
            reducer.all_reduce_finish(handle)
            gb.replay()
        renderer.local_step = g + 1
        if renderer.local_step == RING:
            if march_ahead:
                # Every march of the ring has run by now (the last one under step RING - 2), so the ring's counters are final while step RING - 1
                # is still on the main stream: the read-back waits for the SIDE stream only, and the next ring's first march goes out at once --
                # under the tail of this step -- instead of running inline after the main stream has drained (same 16 counters, same mean)
                with torch.cuda.stream(side_stream):
                    ring_end()
                    if not (renderer.mean_count + 128 > gstate["M"] or renderer.mean_count < 0.8 * gstate["M"]):
                        for s_ in range(group):
                            gstate["graphs"][s_][0].replay()
                        gstate["marched"] = group - 1
            else:
                ring_end()
            if renderer.mean_count + 128 > gstate["M"] or renderer.mean_count < 0.8 * gstate["M"]:
                capture()  # the sample count left the captured buffer size (does not happen on a static scene)

    # priming = the reference's first "epoch-0" steps: full-size buffers until a mean sample count exists
    for k in range(2):
        train_step(k, count=False)
    renderer.update_mean_count()
    renderer.mean_count = dp.all_reduce_max_int(renderer.mean_count, dev)
    # the two priming steps ran on full-size buffers (8192 x 1024 rows): the library's grow-only scratch followed (gigabytes of binning records);
    # hand it back before the steady state sizes it again
    torch.cuda.synchronize()
    nerftex_hip.check(nerftex_hip.lib.nerftex_release_workspaces())
    for k in range(warmup):
        train_step(k, count=False)
    if use_graph:
        try:
            capture()
            for k in range(32):  # past the first two mean_count read-backs after the capture (the first one costs ~13 ms once)
                graph_step(k)
        except Exception as e:  # noqa: BLE001 -- fall back to eager launches, say so
            print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = split_graph = march_ahead = False
    # clocks: a device that idled (process start-up, a graph capture, a profiler run in another process) needs a while under load before it
    # runs at speed -- the first ring after an idle phase measured 30 % slow, a loop started right after 30 idle seconds 2x slow for 0.3 s.
    # Untimed steps until --warm-seconds of wall time have passed (whole rings, so the ring bookkeeping below is unchanged)
    t_w = time.perf_counter()
    k_w = 0
    while time.perf_counter() - t_w < args.warm_seconds:
        for _ in range(RING):
            graph_step(k_w) if use_graph else train_step(k_w, count=False)
            k_w += 1
        torch.cuda.synchronize()
    torch.cuda.synchronize()
    counted["rings"] = -partial_ring()  # the steps of the ring in progress that ran before the timed region

    if time_grid_kernels and not use_graph:
        nerftex_hip.kernel_profile(2, reset=True)  # hipEvent pairs around the hash-grid kernels only (8 of ~90 launches per step)
    dp.barrier()
    torch.cuda.synchronize()
    ring_marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps // 16 + 1)]  # one event per 16 steps: the spread of the step time inside the region
    t0 = time.perf_counter()
    ring_marks[0].record()
    for k in range(steps):
        graph_step(k) if use_graph else train_step(k)
        if k % 16 == 15:
            ring_marks[k // 16 + 1].record()
    torch.cuda.synchronize()
    dp.barrier()
    t1 = time.perf_counter()
    per_ring_ms = [ring_marks[i].elapsed_time(ring_marks[i + 1]) / 16 for i in range(steps // 16)]
    spread = None
    if len(per_ring_ms) >= 3:
        q = sorted(per_ring_ms)
        spread = {"min": q[0], "median": q[len(q) // 2], "max": q[-1], "rings": len(q),
                  "note": "ms per step over each 16-step ring of the timed region (device events on the main stream)"}
    samples_timed = torch.tensor(counted["rings"] + partial_ring(), dtype=torch.int64, device=dev)
    replay_us = {}  # (filled by main(): the rocprofv3 child runs after every timed loop of this process -- the device idles ~30 s meanwhile)
    kernel_us, all_kernel_us = {}, {}
    if time_grid_kernels:
        if use_graph:  # event pairs cannot be read back from a replayed graph: the same step, launched eagerly, right after the timed region
            nerftex_hip.kernel_profile(2, reset=True)
            for k in range(16):
                train_step(k, count=False)
        nerftex_hip.kernel_profile(0)
        kernel_us = nerftex_hip.kernel_profile()
        # outside the timed region: 8 more steps with every library kernel bracketed, for the per-kernel table
        nerftex_hip.kernel_profile(1, reset=True)
        for k in range(8):
            train_step(k, count=False)
        nerftex_hip.kernel_profile(0)
        all_kernel_us = nerftex_hip.kernel_profile()
        nerftex_hip.kernel_profile(reset=True)
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    samples = samples_timed
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(samples, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    samples = int(samples.item())
    collective = None
    if world > 1:  # what the collective library reports, and the cost of one gradient exchange on its own (outside the timed region)
        import torch.distributed as dist

        grads = reducer.big_grads()
        torch.cuda.synchronize()
        dist.barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(10):
            reducer.all_reduce(grads=grads)
        ev1.record()
        torch.cuda.synchronize()
        chunk_us = None
        if chunker is not None:  # each level group's all-reduce on its own
            chunk_us = []
            for i in range(len(chunker)):
                a_, b_ = chunker.rows[i]
                view = grads[0][a_:b_]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                dist.barrier()
                e0.record()
                for _ in range(10):
                    reducer.finish_tensors([reducer.start_tensor(view)])
                e1.record()
                torch.cuda.synchronize()
                chunk_us.append({"levels": list(chunker.levels[i]), "rows": b_ - a_, "us": e0.elapsed_time(e1) * 100.0})
        collective = {"backend": dist.get_backend() + (" (RCCL over xGMI)" if dist.get_backend() == "nccl" else ""), "world_size": dist.get_world_size(),
                      "wire_dtype": str(wire_dtype).replace("torch.", "") if wire_dtype is not None else "the gradient's dtype",
                      "bytes_per_step": int(sum(g.numel() * (torch.finfo(wire_dtype).bits // 8 if wire_dtype else g.element_size()) for g in grads if g is not None)),
                      "allreduce_us_per_step": ev0.elapsed_time(ev1) * 100.0,
                      "allreduce_in_graph": bool(use_graph and ar_state["in_graph"]), "fallbacks": [v for v in (ar_state.get("fallback"), ar_state.get("chunk_fallback")) if v],
                      "table_gradient_chunks": len(chunker) if chunker is not None else 1, "per_chunk": chunk_us}
    if fused_opt:
        opt.sync()  # (double-buffered optimizer state: point the fp32 module parameters at the live set)
    param_l1 = float(sum(p.detach().double().abs().sum() for p in field.parameters()))
    replicas_identical = None
    if world > 1:  # data parallelism keeps full replicas: after the run every rank must hold the same bits (cheap: one checksum vector)
        torch.cuda.synchronize()
        sums = torch.stack([p.detach().double().sum() for p in field.parameters()] + [p.detach().double().abs().sum() for p in field.parameters()])
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))
        if not replicas_identical:
            print(f"[bench] rank {rank}: parameter replicas differ across ranks after training", file=sys.stderr)
    occupancy = None
    if time_grid_kernels and not args.no_occupancy_timing:
        occupancy = measure_occupancy_update(renderer, use_amp, amp_dtype, elapsed / steps * 1e3, samples / steps / world)
    res = dict(fused_table_update=fused_table_update, table_params=int(field.encoder.embeddings.numel()), replicas_identical=replicas_identical, collective=collective, param_l1=param_l1, value=samples / elapsed, ms_per_step=elapsed / steps * 1e3, samples_per_step_per_gpu=samples / steps / world,
               replay_us=replay_us, spread=spread, occupancy=occupancy, graph_used=bool(use_graph),
               mean_count=renderer.mean_count, kernel_us=kernel_us, all_kernel_us=all_kernel_us, use_amp=use_amp, fused_opt=fused_opt, dtype=dtype,
               graph=("three replayed HIP graphs per step (march | shade + backward | optimizer); the gradient all-reduce, launched eagerly after the backward, overlaps the next step's march" if split_graph else
                      (f"replayed HIP graphs: shade + backward + optimizer of {group} consecutive steps per graph, and on a second stream the marches of the next {group} steps (a march needs the rays and the occupancy grid, not the weights)") if march_ahead else
                      "one replayed HIP graph per step") if use_graph else "", dt_gamma=dt_gamma, n_global=n_global)
    return res, field, renderer


def _short_kernel_name(name):
    """rocprofv3 kernel name (mangled or demangled) -> the library's kernel name."""
    import re

    m = re.search(r"\d+([a-z_0-9]+_kernel)", name) if name.startswith("_Z") else re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:60]


def read_kernel_stats(path):
    """A rocprofv3 `*_kernel_stats.csv` -> {kernel: {"calls", "avg_us", "total_us"}} (template instantiations of one kernel added up)."""
    import csv

    out = {}
    with open(path) as fh:
        for row in csv.DictReader(fh):
            k = _short_kernel_name(row["Name"])
            e = out.setdefault(k, {"calls": 0, "total_us": 0.0})
            e["calls"] += int(row["Calls"])
            e["total_us"] += float(row["TotalDurationNs"]) * 1e-3
    for e in out.values():
        e["avg_us"] = e["total_us"] / max(e["calls"], 1)
    return out


def _replay_child_cmd(args, mlp, rays, dtype, steps):
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(steps), "--warmup", str(args.warmup), "--rays", str(rays), "--mlp", mlp,
           "--dtype", dtype, "--bound", str(args.bound), "--steps-per-graph", str(args.steps_per_graph), "--no-cpu-baseline", "--no-other", "--no-infer",
           "--no-kernel-timing", "--baked-pool", "--no-occupancy-timing"]
    for flag in ("no_fused_glue", "no_fused_tail", "no_fused_opt", "no_fused_amp", "no_lean_march", "no_march_ahead", "no_perturb", "no_fused_table_update",
                 "no_skip_dead_samples", "no_fused_composite_step"):
        if getattr(args, flag, False):
            cmd.append("--" + flag.replace("_", "-"))
    return cmd


def rocprof_replay(args, mlp, rays, dtype, steps=416):
    """This benchmark's training leg again, as a child process under rocprofv3 (kernel trace + stats): the per-kernel device durations of the
    replayed step, with whatever runs beside each kernel in the replay.  416 replayed steps against ~20 eager ones (priming, warm-up: they are
    in the averages too, < 5 %).  {} when rocprofv3 is missing or the child fails (the caller falls back and says so)."""
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        print("[bench] rocprofv3 not found: no per-kernel timing of the replayed step", file=sys.stderr)
        return {}
    tmp = tempfile.mkdtemp(prefix="nerftex_prof_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "--"] + _replay_child_cmd(args, mlp, rays, dtype, steps)
        env = dict(os.environ, TMPDIR="/tmp")
        env.pop("WORLD_SIZE", None)
        run = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        files = glob.glob(os.path.join(tmp, "**", "*kernel_stats.csv"), recursive=True)
        if run.returncode != 0 or not files:
            print(f"[bench] rocprofv3 child failed (rc {run.returncode}): {run.stderr[-400:]}", file=sys.stderr)
            return {}
        out = read_kernel_stats(files[0])
        child = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
        if child:
            out["_child"] = {"ms_per_step": json.loads(child[-1])["ms_per_step"], "steps": steps}
        if os.environ.get("NERFTEX_KEEP_STATS"):  # (tools/gpu_*.sh: the same table under profiles/)
            shutil.copy(files[0], os.environ["NERFTEX_KEEP_STATS"])
        return out
    except Exception as e:  # noqa: BLE001 -- a side measurement
        print(f"[bench] rocprofv3 child failed ({type(e).__name__}: {e})", file=sys.stderr)
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def rocprof_traffic(args, mlp, rays, dtype, steps=48):
    """Memory-side bytes per launch of every kernel of the step, MEASURED: two more child runs of this script under `rocprofv3 --pmc FETCH_SIZE`
    and `--pmc WRITE_SIZE` (separate passes: the TCC block cannot hold both, MI355X_MICROARCH.md; kernel trace only beside them -- no other
    trace domain).  Counter collection serialises the dispatches, so the marches no longer run BESIDE the step here: these are each kernel's
    own bytes, which is what `traffic` is.  -> {kernel: {"fetch_kib": avg FETCH_SIZE, "write_kib": avg WRITE_SIZE, "launches": n}} or {}."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    out = collections.defaultdict(dict)
    for counter, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib"), ("TCP_TCC_READ_REQ_sum", "l2_read_req")):
        tmp = tempfile.mkdtemp(prefix="nerftex_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--"] + _replay_child_cmd(args, mlp, rays, dtype, steps)
            env = dict(os.environ, TMPDIR="/tmp")
            env.pop("WORLD_SIZE", None)
            run = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
            files = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
            if run.returncode != 0 or not files:
                print(f"[bench] rocprofv3 --pmc {counter} child failed (rc {run.returncode}): {run.stderr[-300:]}", file=sys.stderr)
                return {}
            vals = collections.defaultdict(list)
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter:
                            vals[_short_kernel_name(row["Kernel_Name"])].append(float(row["Counter_Value"]))
            for k, v in vals.items():
                # the large-sample launches only: the priming steps run the same kernels on full-size buffers and tiny eager launches share names
                v.sort()
                mid = v[len(v) // 4: max(len(v) // 4 + 1, 3 * len(v) // 4)]
                out[k][key] = sum(mid) / len(mid)
                out[k]["launches"] = len(v)
            if os.environ.get("NERFTEX_KEEP_PMC"):
                with open(os.environ["NERFTEX_KEEP_PMC"], "a") as fh:
                    for k, v in sorted(vals.items()):
                        unit = "MB/launch" if counter.endswith("_SIZE") else "M requests/launch (x 128 B = L2 -> L1 bytes)"
                        fh.write(f"{counter:21s} {k:36s} launches={len(v):4d}  interquartile mean {out[k][key] * (1024 if counter.endswith('_SIZE') else 1) / 1e6:10.3f} {unit}\n")
        except Exception as e:  # noqa: BLE001 -- a side measurement
            print(f"[bench] rocprofv3 --pmc {counter} child failed ({type(e).__name__}: {e})", file=sys.stderr)
            return {}
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return dict(out)


def measure_occupancy_update(renderer, use_amp, amp_dtype, ms_per_step, samples_per_step, reps=4):
    """The every-16-steps occupancy-grid update of the trainer (nerf/utils.py:1011 -> nerf/renderer.py:566-660), which the metric excludes
    and SURVEY 8(d) wants reported separately: Renderer.update_extra_state_device, the full sweep (cascade x 128^3 density queries: the
    first 16 updates of a run) and the partial update (cascade x 2 x 128^3 / 4: every later one), device time from events.  The grid, the
    bitfield and the ring are restored afterwards (the rendered frame below uses the analytic occupancy)."""
    saved = (renderer.density_grid.clone(), renderer.density_bitfield.clone(), renderer.mean_density, renderer.iter_density, renderer.mean_count,
             renderer.local_step, renderer.step_counter.clone())
    out = {}
    try:
        for name, it, strat in (("full", 0, True), ("partial", 16, True), ("partial_iid_draws", 16, False)):
            ms = []
            for rep in range(reps + 1):
                renderer.iter_density = it
                renderer.density_grid.copy_(saved[0])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                with torch.autocast("cuda", dtype=amp_dtype, enabled=use_amp):
                    renderer.update_extra_state_device(seed=rep, stratified=strat)
                e1.record()
                torch.cuda.synchronize()
                if rep:  # (first repetition: allocator / workspace growth)
                    ms.append(e0.elapsed_time(e1))
            out[f"ms_{name}"] = sum(ms) / len(ms)
    finally:
        renderer.density_grid.copy_(saved[0])
        renderer.density_bitfield.copy_(saved[1])
        renderer.mean_density, renderer.iter_density, renderer.mean_count, renderer.local_step = saved[2:6]
        renderer.step_counter.copy_(saved[6])
    cells = renderer.cascade * renderer.grid_size ** 3
    out["points_full"], out["points_partial"] = cells, cells // 2
    out["amortised_us_per_step"] = out["ms_partial"] * 1e3 / 16
    out["ms_per_step_including_update"] = ms_per_step + out["ms_partial"] / 16
    out["value_including_occupancy_update_per_gpu"] = samples_per_step / (out["ms_per_step_including_update"] * 1e-3)
    out["note"] = ("update_extra_state every 16 steps (nerf/utils.py:1011), excluded from `value` as SURVEY 8(d) defines the metric; steady state = the partial "
                   "update; device time of Renderer.update_extra_state_device (occupancy kernels + hash-grid gather + sigma net over the sampled cells).  "
                   "ms_partial: the stratified draw (round 5: one cell per run of 4 consecutive Morton indices + one entry per slice of the occupied list -- rows in "
                   "Morton order, the gather sees the full sweep's locality); ms_partial_iid_draws: the reference's N iid draws with replacement (renderer.py:611-621)")
    return out


def measure_accelerated(args, mlp, rays, steps, dev, grid, group=1, dtype="fp16", randint_rays=False, fused_table_update=None, n_views=4):
    """A training loop that feeds FRESH rays every step through ngp_harness.accelerate (the one call a trainer adds to the drop-in
    packages: graph replay + fused field + HalfLeafAdam / FusedAmp, or torch's capturable Adam + GradScaler for nn.Linear MLPs).
    group = k > 1: `step_group` -- the loop has the batches of k consecutive steps at a time ([k, N, 3] tensors, copied into the graphs'
    static buffers every call) and hands the NEXT k batches over one call early: one replayed graph per k steps, their marches ahead on the
    second stream -- the structure of the baked-pool loop of measure_training, with rays the graphs have never seen."""
    from ngp_harness import scene
    from ngp_harness.accelerate import accelerate
    from ngp_harness.model import NGPField, Renderer

    torch.manual_seed(0)
    amp_dt = torch.bfloat16 if dtype == "bf16" else torch.float16
    field = NGPField(bound=args.bound, mlp=mlp, fused_glue=True, mlp_dtype=amp_dt).to(dev)
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    renderer = Renderer(field, bound=args.bound, min_near=0.2, density_thresh=10.0).to(dev)
    renderer.set_occupancy(torch.from_numpy(grid).to(dev))
    n_pool = max(8, group)
    pool = []
    for k in range(n_pool):
        o, d = scene.train_batch(rays, seed=100 + k, n_views=n_views)
        pool.append((torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)))
    gt = torch.rand(n_pool, rays, 3, generator=torch.Generator().manual_seed(4321)).to(dev)
    field.train()
    # (march_across_ring_end: the loop makes no occupancy update inside the timed region -- the metric excludes it -- so the next ring's first
    # marches may start behind the ring's read-back, as they do in the baked-pool loop)
    trainer = accelerate(renderer, dt_gamma=1 / 128, steps_per_call=group, march_across_ring_end=group > 1, amp_dtype=amp_dt,
                         pipeline_adam=getattr(args, "pipeline_adam", 0),
                         fused_table_update=False if getattr(args, "no_fused_table_update", False) else fused_table_update,
                         skip_dead_samples=False if getattr(args, "no_skip_dead_samples", False) else None,
                         fused_composite_step=False if getattr(args, "no_fused_composite_step", False) else None)
    if group > 1:
        assert n_pool % group == 0 and steps % group == 0
        po = [torch.stack([pool[c * group + i][0] for i in range(group)]).contiguous() for c in range(n_pool // group)]
        pd = [torch.stack([pool[c * group + i][1] for i in range(group)]).contiguous() for c in range(n_pool // group)]
        pt = [gt[c * group:(c + 1) * group].contiguous() for c in range(n_pool // group)]
        n_calls = len(po)

        def call(c):
            trainer.step_group(po[c % n_calls], pd[c % n_calls], pt[c % n_calls], next_rays=(po[(c + 1) % n_calls], pd[(c + 1) % n_calls]))
    else:
        def call(c):
            trainer.step(*pool[c % n_pool], gt[c % n_pool], next_rays=pool[(c + 1) % n_pool])
    if randint_rays:
        # NO batch ever repeats (the default loop cycles 8 pre-built batches, so the table's access pattern repeats every 8 steps):
        # every call draws its pixels with torch.randint on the device -- 4 of 32 orbit views per batch, as scene.train_batch does on the host --
        # and builds the rays and fresh random target colours there, INSIDE the timed loop (the next call's rays, one call early: next_rays)
        assert group > 1
        n_views, per_view, HW = 4, rays // 4, 800
        rng_np = np.random.default_rng(99)
        poses = torch.from_numpy(scene.rand_poses(32, 2.0, rng_np)).to(dev)
        fx, fy, cx, cy = [float(v) for v in scene.intrinsics(HW, HW)]
        gen = torch.Generator(device=dev).manual_seed(5)
        state = {}

        def draw():
            views = torch.randint(0, 32, (group, n_views), device=dev, generator=gen)
            inds = torch.randint(0, HW * HW, (group, n_views, per_view), device=dev, generator=gen)
            i = (inds % HW).float() + 0.5
            j = (inds // HW).float() + 0.5
            dcam = torch.stack([(i - cx) / fx, (j - cy) / fy, torch.ones_like(i)], -1)
            dcam = dcam / dcam.norm(dim=-1, keepdim=True)
            R, t = poses[views][..., :3, :3], poses[views][..., :3, 3]
            rd = torch.einsum("gvnj,gvij->gvni", dcam, R).reshape(group, rays, 3).contiguous()
            ro = t[:, :, None, :].expand(group, n_views, per_view, 3).reshape(group, rays, 3).contiguous()
            return ro, rd, torch.rand(group, rays, 3, device=dev, generator=gen)

        state["cur"] = draw()

        def call(c):  # noqa: F811
            nxt = draw()
            ro_, rd_, tg_ = state["cur"]
            trainer.step_group(ro_, rd_, tg_, next_rays=(nxt[0], nxt[1]))
            state["cur"] = nxt
    per_call = max(group, 1)
    for c in range((16 + 16 + 48) // per_call):  # priming (full-size buffers, mean count), warm steps, capture, past the first mean_count read-backs
        call(c)
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < args.warm_seconds or renderer.local_step != 0:  # clocks (see measure_training); then on to a ring boundary
        call(c := c + 1)                                                               # (the sample count is read from whole rings)
        if renderer.local_step == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    base = c + 1
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps // 16 + 1)]
    rings = []
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps // per_call):
        slot0 = renderer.local_step
        call(base + i)
        done = (i + 1) * per_call
        if done % 16 == 0:
            marks[done // 16].record()
            rings.append(renderer.last_ring_samples)  # (set by the ring's mean_count read-back, a host number)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if steps % 16 == 0:
        samples = int(sum(rings))
    else:  # short runs: read the counters of the steps that ran (outside the timed region)
        samples = int(sum(rings)) + int(renderer.step_counter[:renderer.local_step, 0].sum().item())
    per_ring_ms = sorted(marks[i].elapsed_time(marks[i + 1]) / 16 for i in range(steps // 16))
    spread = {"min": per_ring_ms[0], "median": per_ring_ms[len(per_ring_ms) // 2], "max": per_ring_ms[-1], "rings": len(per_ring_ms)} if len(per_ring_ms) >= 3 else None
    return {"value": samples / (t1 - t0), "ms_per_step": (t1 - t0) / steps * 1e3, "loss": float(trainer.loss), "spread": spread, "samples": samples, "steps": steps,
            "steps_per_call": per_call}


def measure_trained_state(args, dev, sc, grid, rays=8192, train_steps=1008, timed_steps=208):
    """the same scene TRAINED against rendered targets of the analytic scene (not noise) -- the field turns opaque, and the
    compositing backward hands exactly zero gradients to every sample behind the point where its ray's transmittance has underflowed
    (raymarching.cu:843-870) -- then timed with and without the dead-sample skip (accelerate(skip_dead_samples=...)): ms per step, the dead fraction,
    and the backward kernels' device time from the library's own timers over eager steps."""
    import nerftex_hip
    from ngp_harness import scene
    from ngp_harness.accelerate import accelerate
    from ngp_harness.model import NGPField, Renderer

    torch.manual_seed(0)
    field = NGPField(bound=args.bound, mlp="ffmlp", fused_glue=True).to(dev).train()
    torch.manual_seed(1)
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    renderer = Renderer(field, bound=args.bound, min_near=0.2, density_thresh=10.0).to(dev)
    renderer.set_occupancy(torch.from_numpy(grid).to(dev))
    k, n_pool = 4, 8
    pool = []
    for j in range(n_pool):
        o, d = scene.train_batch(rays, seed=100 + j, n_views=4)
        ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        pool.append((ro, rd, torch.cat([scene.render_targets(sc, ro[i:i + 2048], rd[i:i + 2048]) for i in range(0, rays, 2048)])))
    po = [torch.stack([pool[c * k + i][0] for i in range(k)]).contiguous() for c in range(n_pool // k)]
    pd = [torch.stack([pool[c * k + i][1] for i in range(k)]).contiguous() for c in range(n_pool // k)]
    pt = [torch.stack([pool[c * k + i][2] for i in range(k)]).contiguous() for c in range(n_pool // k)]
    trainer = accelerate(renderer, dt_gamma=1 / 128, steps_per_call=k, march_across_ring_end=True)
    n_calls = len(po)

    def call(c, ahead=True):
        trainer.step_group(po[c % n_calls], pd[c % n_calls], pt[c % n_calls], next_rays=(po[(c + 1) % n_calls], pd[(c + 1) % n_calls]) if ahead else None)

    c = 0
    loss0 = None
    while c * k < train_steps:  # the trainer's loop: the occupancy grid follows the field every 16 steps (nerf/utils.py:1011)
        call(c, ahead=(c * k + k) % 16 != 0)
        c += 1
        if loss0 is None and c * k >= 32:
            loss0 = float(trainer.loss)
        if (c * k) % 16 == 0:
            with torch.autocast("cuda", dtype=torch.float16):
                renderer.update_extra_state_device()

    def timed(skip):
        nonlocal c
        if trainer._ahead is not None:  # a march started for the next call: that call, without another one behind it
            call(c, ahead=False)
            c += 1
        renderer.skip_dead_samples = trainer.skip_dead_samples = skip
        trainer._graphs, trainer._groups, trainer._warm = None, None, 0  # record the graphs again with / without the flags
        for _ in range(8 + 16 // k * 2):
            call(c, ahead=False)
            c += 1
        while renderer.local_step != 0:
            call(c, ahead=False)
            c += 1
        torch.cuda.synchronize()
        rings = []
        t0 = time.perf_counter()
        for i in range(timed_steps // k):
            call(c)
            c += 1
            if (i + 1) * k % 16 == 0:
                rings.append(renderer.last_ring_samples)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # the backward kernels' own durations: a few eager steps under the library's timers
        if trainer._ahead is not None:
            call(c, ahead=False)
            c += 1
        trainer_eager = trainer.use_graph
        trainer.use_graph = False
        nerftex_hip.lib.nerftex_profile_reset()
        nerftex_hip.lib.nerftex_profile_enable(1)
        renderer.keep_step_live = True  # (the one-launch compositing hands its flags to the backward that zeroes them: keep a copy of the last step's to count)
        for _ in range(8 // k):
            call(c, ahead=False)
            c += 1
        torch.cuda.synchronize()
        renderer.keep_step_live = False
        prof = nerftex_hip.kernel_profile()
        nerftex_hip.lib.nerftex_profile_enable(0)
        trainer.use_graph = trainer_eager
        want = ("field_color_backward_kernel", "field_sigma_backward_kernel", "bin_fill_dir_kernel", "sum_tiles_adam_kernel", "sum_tiles_dir_kernel", "combine_tiles_kernel",
                "composite_tail_bwd_kernel", "composite_step_kernel", "composite_step_loss_kernel", "grid_forward_level_kernel", "field_forward_train_kernel")
        kern = {n: round(prof[n]["avg_us"], 1) for n in want if n in prof}
        mlp_bwd = sum(v for n, v in kern.items() if n.startswith("field_") and "backward" in n)
        return {"ms_per_step": dt / timed_steps * 1e3, "value": sum(rings) / dt if rings else None, "samples_per_step": (sum(rings) / (len(rings) * 16)) if rings else None,
                "mlp_backward_us_eager": round(mlp_bwd, 1), "hash_grid_backward_us_eager": round(sum(v for n, v in kern.items() if n.split("_")[0] in ("bin", "sum", "combine")), 1),
                "kernels_avg_us_eager": kern}

    with_skip = timed(True)
    holder = getattr(renderer, "last_step_live", None) or {}
    flags = holder.get("last")
    dead_steps = float((flags == 0).float().mean()) if flags is not None else None
    without = timed(False)
    return {"workload": f"configs[2], TRAINED STATE: {c * k} steps of accelerate(steps_per_call={k}).step_group against rendered targets of the analytic scene (ngp_harness.scene."
                        "render_targets: the blobs' density composited with a smooth colour field), the occupancy grid updated every 16 steps; then timed with the "
                        "dead-sample skip (the compositing backward flags the 32-sample steps that carry a gradient; the MLP backward and the hash-grid record builder "
                        "walk those only) and without it.  Same parameters either way, bit for bit (tests/test_gpu_round6.py)",
            "rays_per_batch": rays, "dtype": "fp16", "loss_after_32_steps": loss0, "loss_now": float(trainer.loss), "dead_32_sample_steps_fraction": dead_steps,
            "with_skip": with_skip, "without_skip": without, "value": with_skip["value"], "unit": "ray-samples/s", "ms_per_step": with_skip["ms_per_step"]}


def measure_curved(dev, n_points=262144, reps=10):
    """BASELINE.json configs[3], the curved-field texture on a star_flower-shaped synthetic mesh (~20 k faces, query points
    within the height threshold of the surface and beyond it).  Per batch of sample points: the neighbour search (K = 8 nearest mesh
    vertices, csrc/knn.hip -- the role of the reference's frnn), the projector (normal from those neighbours, two BVH closest-hit traces
    along +-normal, select, mask, frame, FreqEncoder: one kernel), and the curved field's hash lookup (GridEncoder_clustering L = 8,
    512 -> 1024, tools/map.py:563) forward + table-gradient backward.  Device time from events; points/s = points / (search + projector +
    lookup forward + backward).  `field_*`: the whole CurvedField (projector -> table ++ FreqEncoder -> FFMLP 48-32-16 -> reflection SH
    -> FFMLP 32-64-64-3, masked), forward and forward + backward."""
    from ngp_harness.curved import CurvedField, star_flower_mesh

    v, f = star_flower_mesh()
    torch.manual_seed(0)
    field = CurvedField(v, f, bound=1.0, h_threshold=0.05).to(dev)
    proj = field.projector
    g = torch.Generator().manual_seed(7)
    vt = torch.as_tensor(v, dtype=torch.float32)
    # round 5: the points in the order a renderer hands them to the field -- 64 consecutive samples along each of n_points / 64 rays that hit the surface
    # (rounds 3-4 measured points around randomly drawn vertices in RANDOM order: no coherence at all, which no caller of the reference produces --
    # its MeshFeatureField is fed by march_rays_train; that order is still measured below, as `random_order`: profiles/r05_pmc_curved.txt)
    n_rays = n_points // 64
    ro = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=-1) * 2.5
    hit = vt[torch.randint(0, vt.shape[0], (n_rays,), generator=g)]
    rd = torch.nn.functional.normalize(hit - ro, dim=-1)
    ts = (hit - ro).norm(dim=-1, keepdim=True) + (torch.arange(64).float().reshape(1, 64) - 32) * (0.12 / 64) + torch.rand(n_rays, 1, generator=g) * 1e-3
    xyz = (ro[:, None] + rd[:, None] * ts[..., None]).reshape(-1, 3).contiguous().to(dev)
    base = vt[torch.randint(0, vt.shape[0], (n_points,), generator=g)]
    xyz_random = (base * (1 + (torch.rand(n_points, 1, generator=g) - 0.5) * 0.12) + (torch.rand(n_points, 3, generator=g) - 0.5) * 0.01).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n_points, 3, generator=g), dim=-1).to(dev)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, out

    t_knn, neighbours = timed(lambda: proj.knn(xyz))
    t_proj, out = timed(lambda: proj.project_fused(xyz, neighbours=neighbours))
    p_sur, mask = out[0], out[2]
    field.encoder.embeddings.data.uniform_(-1e-4, 1e-4)
    field.train()
    with torch.autocast("cuda", dtype=torch.float16):
        t_fwd, feat = timed(lambda: field.encoder(p_sur, bound=1.0))
        go = torch.randn_like(feat) * 1e-3

        def fb():
            field.encoder.embeddings.grad = None
            field.encoder(p_sur, bound=1.0).backward(go)
        t_fb, _ = timed(fb)
        t_field_fwd, (sigma, color, _) = timed(lambda: field(xyz, dirs))
        t_field_graphed, graphed_equal = None, None
        try:  # the same forward as one replayed HIP graph (round 6): what the ~40 launches of a call cost the host is gone
            with torch.no_grad():
                want_s, want_c, _ = field(xyz, dirs)
                t_field_graphed, (gs_, gc_, _) = timed(lambda: field.forward_graphed(xyz, dirs))
                graphed_equal = bool(torch.equal(gs_, want_s) and torch.equal(gc_, want_c))
        except Exception as e:  # noqa: BLE001 -- a side measurement
            print(f"[bench] graphed curved-field forward failed ({type(e).__name__}: {e})", file=sys.stderr)
        gs, gc = torch.randn_like(sigma) * 1e-3, torch.randn_like(color) * 1e-3

        def field_fb():
            for p in field.parameters():
                p.grad = None
            s_, c_, _ = field(xyz, dirs)
            torch.autograd.backward([s_, c_], [gs, gc])
        t_field_fb, _ = timed(field_fb)
    total = t_knn + t_proj + t_fb
    t_knn_r, nb_r = timed(lambda: proj.knn(xyz_random))
    t_proj_r, _ = timed(lambda: proj.project_fused(xyz_random, neighbours=nb_r))
    return {"workload": "configs[3]: curved-field texture lookup on a star_flower-shaped synthetic mesh (%d faces), sample points in a RENDERER'S ORDER (64 consecutive samples per ray -- what march_rays_train feeds the reference's MeshFeatureField; the random-order workload of rounds 3-4 is `random_order` below and is the round-over-round comparable): K = 8 neighbour search (uniform vertex grid, "
                        "exact) + projector (K-neighbour normal + two BVH traces + select + frame + FreqEncoder, one kernel) + GridEncoder_clustering L=8 hash "
                        "lookup forward and table-gradient backward, 1 GPU" % len(f),
            "points_per_batch": n_points, "inside_height_threshold": float(mask.float().mean()), "value": n_points / (total * 1e-6), "unit": "sample points/s",
            "random_order": {"neighbour_search_us": t_knn_r, "projector_us": t_proj_r, "value": n_points / ((t_knn_r + t_proj_r + t_fb) * 1e-6),
                             "note": "the same number of points around randomly drawn vertices in random order (the workload of rounds 3-4): divergence-bound search, profiles/r05_pmc_curved.txt"},
            "neighbour_search_us": t_knn, "projector_us": t_proj, "lookup_forward_us": t_fwd, "lookup_forward_backward_us": t_fb,
            "field_forward_us": t_field_fwd, "field_forward_graphed_us": t_field_graphed, "field_forward_graphed_equals_eager": graphed_equal,
            "field_forward_backward_us": t_field_fb, "field_points_per_s_forward_backward": n_points / (t_field_fb * 1e-6),
            "dtype": "f32 geometry, f16 table and MLPs under autocast"}


WORKLOADS = {
    "ffmlp": "configs[2] (1 GPU) / configs[4] (8 GPUs, 65536 rays/step): fox-style scene, hashgrid L=16 F=2 T=2^19, FFMLP 2x64 + 3x64 on MFMA, "
             "HIP gridencoder+raymarching+shencoder+ffmlp, 800x800 random-pose pixels",
    "torch": "configs[1]: fox-style scene, hashgrid L=16 F=2 T=2^19, nn.Linear 2x64 + 3x64 MLPs on PyTorch-ROCm, "
             "HIP gridencoder+raymarching+shencoder, 800x800 random-pose pixels",
}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves (torch.distributed.run, one process per GPU,
    rendezvous on 127.0.0.1 and a free port) and hand its exit code back.  Under torchrun (WORLD_SIZE set) this is never reached."""
    import socket
    import subprocess

    share = os.environ.get("NERFTEX_DP_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not share:
        print(f"[bench] --gpus {args.gpus} but this node exposes {have} GPU(s) (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES?); "
              f"NERFTEX_DP_SHARE_GPU=1 runs all ranks on cuda:0 over gloo (a test rig, not a measurement)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer mappings fail with the legacy mode on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr)
    return subprocess.call(cmd, env=env)


# ----------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    from ngp_harness import dp, scene

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank, world, local = dp.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (HIP path only; there is no CPU fallback)"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    from ngp_harness.streams import ensure_pool, pool_report

    ensure_pool(dev)  # (the package's streams get their hardware queues before anything else does: ngp_harness/streams.py for what the order costs otherwise)
    # N > 1: nothing is timed before the group has proven itself (world size, one all-reduce per wire type, barrier): dp.preflight raises with
    # the rank and the failing check; its report goes into config.collective
    pre = dp.preflight(world, dev) if world > 1 else None

    sc = scene.Scene(bound=args.bound, seed=0)
    grid, thresh, bits = sc.bitfield()
    res, field, renderer = measure_training(args, args.mlp, args.rays, args.steps, args.warmup, dev, rank, world, sc, grid, bits,
                                            not args.no_kernel_timing, graph=not args.no_graph)
    use_amp, dt_gamma = res["use_amp"], res["dt_gamma"]
    # ---- the headline at N = 1: the same step fed with FRESH rays through the one call a trainer makes (ngp_harness.accelerate, step_group): the
    # loop above bakes its pool of ray batches into the graphs -- a trainer cannot -- so it is reported under other_config and the number a
    # training loop gets is `value`.  Same kernels, same graphs-per-steps structure; the rays are copied into static buffers every call.
    fresh = None
    plain = not (args.no_graph or args.no_fused_glue or args.no_fused_tail or args.no_fused_opt or args.no_fused_amp or args.no_march_ahead or args.graph_split
                 or args.no_perturb or args.no_lean_march or args.baked_pool)
    if world == 1 and args.mlp == "ffmlp" and args.dtype == "fp16" and plain and args.steps_per_graph in (1, 2, 4, 8, 16) and args.steps % args.steps_per_graph == 0:
        try:
            # (what accelerate() gives a trainer, nothing else: until round 5 this loop ran with the march_lean knob set; on round 6's step the default march is
            # the better neighbour by a hair -- 0.508 vs 0.511 - 0.516 ms per step over two pairs of runs on one box -- and it is what a caller gets)
            fresh = measure_accelerated(args, "ffmlp", args.rays, args.steps, dev, grid, group=args.steps_per_graph)
        except Exception as e:  # noqa: BLE001 -- fall back to the baked-pool loop as the headline, and say so
            print(f"[bench] fresh-ray loop failed ({type(e).__name__}: {e}); headline = the baked-pool loop", file=sys.stderr)
            fresh = None

    # ---- rendered Mpix/s: one 800x800 frame through the reference's inference loop (nerf/renderer.py:436-487)
    mpix = None
    if not args.no_infer and rank == 0:
        field.eval()
        rng = np.random.default_rng(7)
        pose = scene.rand_poses(1, 2.0, rng)[0]
        o, d = scene.get_rays(pose, scene.intrinsics(800, 800), 800, 800)
        ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        def frames(render, n_frames=3):
            with torch.autocast("cuda", dtype=torch.bfloat16 if res["dtype"] == "bf16" else torch.float16, enabled=use_amp):
                render(ro, rd, dt_gamma=dt_gamma)  # warm-up frame
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(n_frames):
                    img, _, n_inf = render(ro, rd, dt_gamma=dt_gamma)
                torch.cuda.synchronize()
                return (time.perf_counter() - t2) / n_frames, img, int(n_inf), renderer.last_iters

        t_ref, img_ref, n_ref, it_ref = frames(renderer.render_infer)
        t_pipe, img_pipe, n_pipe, it_pipe = frames(renderer.render_infer_pipelined)
        F, P = args.infer_slots, args.infer_parts
        t_big, img_big, n_big, it_big = frames(lambda ro, rd, dt_gamma: renderer.render_infer_pipelined(ro, rd, dt_gamma=dt_gamma, slots_per_ray=F, parts=P))
        # round 5: the same loop with the host taken out (Renderer.render_infer_graphed: per ray range one graph per block of iterations, n_step derived
        # on the device); per-frame times over 10 frames, and the host's own share (time spent enqueueing, before the final synchronise)
        graphed = None
        if use_amp and ((res["dtype"] == "fp16" and getattr(field, "fused_field", False)) or (res["dtype"] == "bf16" and getattr(field, "fused_field_bf16", False))):
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16 if res["dtype"] == "bf16" else torch.float16):
                    render_g = lambda: renderer.render_infer_graphed(ro, rd, dt_gamma=dt_gamma, slots_per_ray=F, parts=P)  # noqa: E731
                    img_g, _, n_g = render_g()  # records the graphs
                    render_g()
                    torch.cuda.synchronize()
                    per_frame, enqueue = [], []
                    for _ in range(10):
                        t2 = time.perf_counter()
                        img_g, _, n_g = render_g()
                        t3 = time.perf_counter()
                        torch.cuda.synchronize()
                        per_frame.append(time.perf_counter() - t2)
                        enqueue.append(t3 - t2)
                per_frame.sort()
                t_g = sum(per_frame) / len(per_frame)
                graphed = {"mpix_per_s": 0.64 / t_g, "ms_per_frame": t_g * 1e3, "ms_per_frame_min_median_max": [per_frame[0] * 1e3, per_frame[5] * 1e3, per_frame[-1] * 1e3],
                           # (mostly WAITING for the block-old alive counts, not enqueueing: ~30 graph replays per frame)
                           "host_ms_inside_the_call": sum(enqueue) / len(enqueue) * 1e3, "iterations_launched": renderer.last_iters, "sample_slots_launched": int(n_g),
                           "max_abs_image_difference_vs_reference_loop": float((img_g - img_ref).abs().max()),
                           "loop": f"the loop below as HIP graphs: per ray range one graph resets it and one runs 2 iterations (replayed until no ray is left; the host "
                                   f"learns the alive count one block late from a 4-byte copy and picks the recorded launch size that covers it: all rays, 1/2, 1/4, 1/8, 1/32, ...), n_step = clamp({F} N / alive, {F}, {8 * F}) "
                                   f"derived by the kernels from the alive count on the device (NERFTEX_ROWS_AUTO); {P} ranges side by side; same image bit for bit"}
            except Exception as e:  # noqa: BLE001 -- a side measurement
                print(f"[bench] graphed inference failed ({type(e).__name__}: {e})", file=sys.stderr)
        # how many of the sample SLOTS a frame shades hold a sample at all (delta > 0: march_rays pads a ray's row of n_step slots with zeros once it
        # has left the volume): counted in the reference-shaped loop with the reference's schedule (1 slot-unit per ray) and with the fast loops'
        # (F units: the same n_step rule, so the same slots as the graphed / host-launched loops) -- outside every timed frame
        real = {}
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16 if res["dtype"] == "bf16" else torch.float16, enabled=use_amp):
                for name, f_ in (("reference_schedule", 1), ("fast_loops_schedule", F)):
                    renderer.count_real_samples, renderer.real_samples = True, 0
                    _, _, slots = renderer.render_infer(ro, rd, dt_gamma=dt_gamma, slots_per_ray=f_)
                    real[name] = {"sample_slots": int(slots), "real_samples": int(renderer.real_samples), "slots_per_ray_unit": f_}
        except Exception as e:  # noqa: BLE001 -- a side measurement
            print(f"[bench] real-sample count failed ({type(e).__name__}: {e})", file=sys.stderr)
        finally:
            renderer.count_real_samples = False
        host_form = {"mpix_per_s": 0.64 / t_big, "ms_per_frame": t_big * 1e3}
        use_graphed = graphed is not None and graphed["mpix_per_s"] >= host_form["mpix_per_s"]
        head_inf = graphed if use_graphed else host_form  # (the faster of the two forms of the same loop in THIS run; both are reported)
        mpix = {"mpix_per_s": head_inf["mpix_per_s"], "ms_per_frame": head_inf["ms_per_frame"], "mpix_per_s_is": "graphed" if use_graphed else "host_launched", "graphed": graphed,
                "real_samples_per_frame": real,
                "host_launched": {"mpix_per_s": 0.64 / t_big, "ms_per_frame": t_big * 1e3, "samples_per_frame": n_big, "iterations": it_big},
                "samples_per_frame": n_big, "iterations": it_big,
                "loop": f"run_cuda's inference loop, no per-iteration host stall (launches sized by an earlier alive count, the true count read on the "
                        f"device: nerftex_*_rays_dev / *_rows), {F} N sample slots per iteration instead of N (n_step = clamp({F} N // n_alive, {F}, {8 * F})), "
                        f"the rays in {P} ranges on their own streams; same image as the reference loop, bit for bit"
                        + ("; `graphed` is the same loop replayed as HIP graphs (round 5), `host_launched` enqueued launch by launch; `mpix_per_s` = the faster of the two in this run (`mpix_per_s_is`)" if graphed else ""),
                "reference_schedule_no_stall": {"mpix_per_s": 0.64 / t_pipe, "ms_per_frame": t_pipe * 1e3, "samples_per_frame": n_pipe, "iterations": it_pipe,
                                                "loop": "the reference's schedule (N slots per iteration), only the host stall removed"},
                "reference_loop": {"mpix_per_s": 0.64 / t_ref, "ms_per_frame": t_ref * 1e3, "samples_per_frame": n_ref, "iterations": it_ref,
                                   "loop": "nerf/renderer.py:436-487 as written, alive_counter.item() every iteration"},
                "max_abs_image_difference": max(float((img_pipe - img_ref).abs().max()), float((img_big - img_ref).abs().max()))}
        field.train()
    dp.barrier()

    # ---- the other single-GPU configurations, short eager runs, for the record (rank 0, N = 1 only)
    other = None
    if rank == 0 and world == 1 and not args.no_other:
        other = []
        o_mlp, o_rays = ("torch", 4096) if args.mlp == "ffmlp" else ("ffmlp", 8192)
        runs = [("configs[1]" if o_mlp == "torch" else "configs[2]", o_mlp, o_rays, args.dtype, False, False),
                # the step the UNMODIFIED reference callers would run: drop-in packages only, eager launches, torch.optim.Adam + GradScaler
                ("configs[2] through the drop-in API only (reference callers unchanged: no graph, no field-glue / render-tail kernels, torch Adam + GradScaler)",
                 "ffmlp", 8192, "fp16", True, False),
                ("configs[2] with bf16 networks (BASELINE's dtype for this config): bf16 autocast, the one-kernel bf16 field over the fp16 table, HalfLeafAdam with bf16 MLP leaves + FusedAmp "
                 "(the loss scaler stays: the table gradient is fp16), one replayed HIP graph per step, baked ray pool", "ffmlp", 8192, "bf16", False, True)]
        for label, mlp_k, rays_k, dt_k, dropin, graph_k in runs:
            r2, _, _ = measure_training(args, mlp_k, rays_k, 16, 16, dev, rank, world, sc, grid, bits, False, graph=graph_k, dtype=dt_k, dropin_only=dropin)
            other.append({"workload": label if "configs[2]" in label and len(label) > 12 else WORKLOADS[mlp_k], "rays_per_batch": rays_k, "dtype": dt_k,
                          "value": r2["value"], "unit": "ray-samples/s", "ms_per_step": r2["ms_per_step"], "steps": 16,
                          "launch": r2["graph"] if r2["graph"] else "eager launches"})
        for label, mlp_k, rays_k in (("configs[2] through ngp_harness.accelerate(renderer): a training loop that feeds fresh rays every step (and has the next batch's rays one step early: next_rays); replayed HIP graphs "
                                      "per step, fused field, HalfLeafAdam + FusedAmp", "ffmlp", 8192),
                                     ("configs[1] through ngp_harness.accelerate(renderer): nn.Linear MLPs on PyTorch-ROCm, torch's capturable fused Adam + GradScaler "
                                      "inside one replayed HIP graph per step", "torch", 4096)):
            try:
                r3 = measure_accelerated(args, mlp_k, rays_k, 32, dev, grid)
                other.append({"workload": label, "rays_per_batch": rays_k, "dtype": "fp16", "value": r3["value"], "unit": "ray-samples/s",
                              "ms_per_step": r3["ms_per_step"], "steps": 32, "launch": "two replayed HIP graphs per step (inputs copied into static buffers): the step, and on a second stream the march of the batch handed over as next_rays",
                              "loss_after_run": r3["loss"]})
            except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
                print(f"[bench] accelerate() measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        try:  # the headline's loop with rays that never repeat, drawn on the device inside the timed loop
            r5 = measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4, randint_rays=True)
            other.append({"workload": "configs[2], the headline's loop with NEVER-REPEATING rays: every call draws its 4 x 8192 pixels with torch.randint on the device (4 of 32 orbit "
                                      "views per batch) and builds rays and random target colours there, inside the timed loop (the headline cycles 8 pre-built batches)",
                          "rays_per_batch": 8192, "dtype": "fp16", "value": r5["value"], "unit": "ray-samples/s", "ms_per_step": r5["ms_per_step"], "steps": 208,
                          "ms_per_step_spread": r5["spread"], "launch": "one replayed HIP graph per 4 steps + their marches ahead on the second stream + ~25 framework launches per call for the ray generation",
                          "loss_after_run": r5["loss"]})
        except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
            print(f"[bench] randint-ray measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        try:  # the headline's loop with the batch composition of the reference's own loader: all rays of a batch from ONE image
            r6 = measure_accelerated(args, "ffmlp", 8192, 208, dev, grid, group=4, n_views=1)
            other.append({"workload": "configs[2], the headline's loop with every batch drawn from ONE view (what the reference's loader hands its trainer: nerf/provider.py:326-366, "
                                      "`B = len(index) # always 1`, num_rays random pixels of that image); the headline mixes 4 views per batch, which is the less coherent sample stream",
                          "rays_per_batch": 8192, "dtype": "fp16", "value": r6["value"], "unit": "ray-samples/s", "ms_per_step": r6["ms_per_step"], "steps": 208,
                          "ms_per_step_spread": r6["spread"], "samples_per_step": r6["samples"] / r6["steps"]})
        except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
            print(f"[bench] one-view measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        if args.trained_steps > 0:
            try:  # the scene trained against rendered targets, then timed with and without the dead-sample skip
                other.append(measure_trained_state(args, dev, sc, grid, train_steps=args.trained_steps))
            except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
                print(f"[bench] trained-state measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        try:  # the headline's loop (fresh rays, 4 steps per call) with bf16 networks
            r4 = measure_accelerated(args, "ffmlp", 8192, 64, dev, grid, group=4, dtype="bf16")
            other.append({"workload": "configs[2] in bf16 through ngp_harness.accelerate(renderer, steps_per_call=4, amp_dtype=torch.bfloat16).step_group: the headline's loop "
                                      "(fresh rays every call) with bf16 networks over the fp16 table", "rays_per_batch": 8192, "dtype": "bf16", "value": r4["value"],
                          "unit": "ray-samples/s", "ms_per_step": r4["ms_per_step"], "steps": 64, "launch": "one replayed HIP graph per 4 steps + their marches ahead on the second stream",
                          "loss_after_run": r4["loss"]})
        except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
            print(f"[bench] bf16 accelerate() measurement failed ({type(e).__name__}: {e})", file=sys.stderr)
        try:
            other.append(measure_curved(dev))
        except Exception as e:  # noqa: BLE001 -- a side measurement must not take the headline line down with it
            print(f"[bench] configs[3] measurement failed ({type(e).__name__}: {e})", file=sys.stderr)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, bits, args.cpu_rays)

    # ---- per-kernel durations INSIDE the replayed step.  hipEvent pairs cannot be read back from a replayed graph (and external event-record
    # nodes fail to capture on this stack), so this very script runs once more as a child under `rocprofv3 --kernel-trace --stats`: same
    # workload, same graphs, the marches of the next steps on the second stream -- the device durations rocprofv3 reports are the replay's own.
    # LAST of everything this process measures: the device idles for ~30 s while rocprofv3 writes its tables, and a loop timed right after
    # that runs on cold clocks (2x slow for a third of a second: DESIGN.md 6).
    if rank == 0 and not args.no_kernel_timing and res.get("graph_used") and world == 1 and not args.no_replay_profile:
        res["replay_us"] = rocprof_replay(args, args.mlp, args.rays, res["dtype"])
        if not args.no_traffic_profile:
            res["traffic_kib"] = rocprof_traffic(args, args.mlp, args.rays, res["dtype"])

    # ---- roofline of the dominant hash-grid op, from the library's own per-kernel hipEvent pairs over the timed region
    M_launch = renderer.mean_count + 128 - renderer.mean_count % 128  # rows per launch (padded like raymarching.py:198-201)
    s_bytes = 2 if use_amp else 4
    bytes_fwd = 12 + 8 * 16 * 2 * s_bytes + 16 * 2 * s_bytes  # 588 B (fp16) / 1164 B (fp32) per point
    bytes_bwd = 12 + 16 * 2 * s_bytes + 8 * 16 * 2 * s_bytes
    def per_op(prof):
        kern = {}
        for name, bpp in (("grid_encode_forward", bytes_fwd), ("grid_encode_backward", bytes_bwd)):
            parts = {k: prof[k] for k in GRID_KERNELS[name] if k in prof}
            if parts:
                calls = max(v["calls"] for v in parts.values())
                ms = sum(v["total_us"] for v in parts.values()) / calls * 1e-3
                # round 6, fused table update: the backward's summing / combine kernels ALSO run the optimizer on the table (fp32 master + two moments
                # read, the same three + the fp16 copy written: 26 B per parameter, no gradient tensor in between) -- those algorithmic bytes belong
                # to the launch; the figure on the backward's own 588 B per point is kept beside it
                opt_bytes = 26 * res["table_params"] if (name == "grid_encode_backward" and res.get("fused_table_update")) else 0
                kern[name] = {"ms": ms, "gbs": (bpp * M_launch + opt_bytes) / (ms * 1e-3) / 1e9, "bytes_per_point": bpp,
                              "kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in parts.items()}}
                if opt_bytes:
                    kern[name].update(optimizer_bytes_per_launch=opt_bytes, gbs_backward_bytes_only=bpp * M_launch / (ms * 1e-3) / 1e9,
                                      includes="Adam on the 12.6 M table parameters, applied by the owners of each row's final gradient (sum_tiles / combine_tiles); the "
                                               "step's adam_half_kernel is left with the MLP weights and the loss scaler's update")
        return kern

    # durations: from INSIDE the replayed step when they could be taken there (external event-record nodes in a copy of the step's graph, the
    # next steps' marches running beside it as in the timed region), else from eager launches of the same step after the timed region
    replay_us = dict(res.get("replay_us") or {})
    child = replay_us.pop("_child", None)
    source = "rocprofv3 --kernel-trace --stats over a child run of this script (same workload, %d replayed steps, child ms_per_step %.4f)" % (
        child["steps"], child["ms_per_step"]) if child else None
    if not replay_us and res["graph"] and os.path.exists(os.path.join(ROOT, COMMITTED_STATS)):  # no live profile: the committed one, labelled
        replay_us = read_kernel_stats(os.path.join(ROOT, COMMITTED_STATS))
        source = "CONSTANT from " + COMMITTED_STATS + " (rocprofv3 of the same command, committed): no live rocprofv3 run in this invocation"
    res["replay_us"] = replay_us
    in_replay = bool(replay_us)
    kern = per_op(res["replay_us"] if in_replay else res["kernel_us"])
    kern_eager = per_op(res["kernel_us"]) if in_replay else {}
    # what bounds each op, by the counters: the gather is served by the L2s (94 % hits, 128-B lines for 8-B rows): it sits on the
    # L2 -> L1 line rate, not on HBM; the backward's two kernels sit on VALU issue (profiles/r05_pmc_sq_grid.txt) -- its algorithmic bytes are
    # still priced against HBM, the nearest roof the contract names
    BOUND = {"grid_encode_forward": ("l2_line", "the gather is bound by the L2 -> L1 line bandwidth (9.4x line amplification: 128 B moved per 8 B used, 94 % L2 hits; "
                                                "profiles/r05_pmc_l2.txt), not by HBM; `frac` is algorithmic bytes over the HBM peak all the same"),
             "grid_encode_backward": ("hbm", "priced against HBM as the contract asks; by the SQ counters both kernels sit on VALU issue (DESIGN.md 4.1)")}
    dominant = max(kern, key=lambda n: kern[n]["ms"]) if kern else None
    roofline = None
    # memory-side bytes per launch: MEASURED by the two PMC child runs when they could be taken (rocprof_traffic), else the committed constant.
    # Correction as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE counts wide coalesced streaming reads at half their bytes -- the
    # backward's record stream (K4d reads 16 B per lane) is doubled; the gather's 4-8 B reads are uncalibrated and taken as reported.
    tk = res.get("traffic_kib") or {}
    traffic, traffic_source = {}, None
    for name in GRID_KERNELS:
        parts = {k: tk[k] for k in GRID_KERNELS[name] if k in tk and "fetch_kib" in tk[k] and "write_kib" in tk[k]}
        if parts:
            f = 2.0 if name == "grid_encode_backward" else 1.0
            per = {k: round((f * v["fetch_kib"] + v["write_kib"]) * 1024 / 1e6, 2) for k, v in parts.items()}
            traffic[name] = {"bytes": sum(per.values()) * 1e6, "per_kernel_MB": per}
    if traffic:
        traffic_source = ("MEASURED in this run: two child runs of this script under rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, kernel trace "
                          "only), interquartile mean per kernel over the replayed steps; backward = 2 x FETCH_SIZE + WRITE_SIZE (the guide's gfx950 correction "
                          "for wide coalesced streams), forward = FETCH_SIZE + WRITE_SIZE as reported")
    else:
        traffic = {k: {"bytes": v} for k, v in TRAFFIC_BYTES_PER_LAUNCH.items()}
        traffic_source = ("CONSTANT, not measured in this run (no rocprofv3, a failed PMC child, or --no-traffic-profile): " + TRAFFIC_PROFILE +
                          "; 2 x FETCH_SIZE + WRITE_SIZE as the guide's gfx950 correction prescribes for coalesced streams")
    if dominant:
        for k_, v_ in kern.items():
            v_["bound"], v_["bound_note"] = BOUND[k_]
            v_["frac_of_hbm_peak"] = v_["gbs"] / HBM_PEAK_GBS
            v_["traffic"] = traffic.get(k_, {}).get("bytes")
            # L2 -> L1 side, measured in this run when the third PMC pass could be taken: TCP_TCC_READ_REQ_sum x 128 B over the op's duration,
            # against the guide's 34.5 TB/s aggregate L2 bandwidth (what `l2_line` means as a number)
            reqs = [tk[k]["l2_read_req"] for k in GRID_KERNELS[k_] if k in tk and "l2_read_req" in tk[k]]
            if reqs:
                v_["l2_to_l1_bytes"] = sum(reqs) * 128.0
                v_["frac_of_l2_peak"] = v_["l2_to_l1_bytes"] / (v_["ms"] * 1e-3) / L2_PEAK_BS
        roofline = {
            "bound": BOUND[dominant][0], "kernel": dominant, "achieved": kern[dominant]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": kern[dominant]["gbs"] / HBM_PEAK_GBS, "traffic": traffic.get(dominant, {}).get("bytes"),
            "traffic_source": traffic_source, "traffic_per_kernel_MB": traffic.get(dominant, {}).get("per_kernel_MB"),
            "traffic_over_algorithmic": (traffic[dominant]["bytes"] / (kern[dominant]["bytes_per_point"] * M_launch + kern[dominant].get("optimizer_bytes_per_launch", 0)))
            if traffic.get(dominant, {}).get("bytes") else None,
            "fused_table_update": bool(kern[dominant].get("optimizer_bytes_per_launch")), "optimizer_bytes_per_launch": kern[dominant].get("optimizer_bytes_per_launch"),
            "frac_backward_bytes_only": (kern[dominant]["gbs_backward_bytes_only"] / HBM_PEAK_GBS) if kern[dominant].get("gbs_backward_bytes_only") else None,
            "includes": kern[dominant].get("includes"),
            "avg_launch_ms": kern[dominant]["ms"], "points_per_launch": M_launch, "algorithmic_bytes_per_point": kern[dominant]["bytes_per_point"],
            "kernels_avg_us": kern[dominant]["kernels_avg_us"],
            "durations_from": ("the replayed step: " + source if in_replay else
                               "eager launches of the same step after the timed region (hipEvent pairs)" if res["graph"] else "the timed region itself (hipEvent pairs)"),
            "eager_avg_launch_ms": kern_eager.get(dominant, {}).get("ms"),
            "other": {k: v for k, v in kern.items() if k != dominant},
            # (spin_kernel = torch.cuda._sleep: the ~15 sleep kernels of ngp_harness.streams.ensure_pool's one-off measurements at the child's start, not part of a step)
            "all_kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in sorted((res["replay_us"] if in_replay else res["all_kernel_us"]).items(),
                                                                              key=lambda kv: -kv[1]["total_us"]) if k != "spin_kernel"},
            "all_kernels_avg_us_eager": {k: round(v["avg_us"], 2) for k, v in sorted(res["all_kernel_us"].items(), key=lambda kv: -kv[1]["total_us"])} if in_replay else None,
            "note": "avg_launch_ms = sum of the device durations of the kernels one C-ABI call launches (hipEvent pairs recorded by the library on the "
                    "launch stream, names = rocprofv3 kernel names); the 24 MiB table is Infinity-Cache resident, see DESIGN.md 4/6",
        }

    if rank == 0:
        head = fresh if fresh is not None else res
        baked = None
        if fresh is not None:
            baked = {"workload": "the same step with a pool of 8 ray batches BAKED into the graphs (the headline of rounds 1-3; a trainer cannot do this): " + res["graph"],
                     "rays_per_batch": args.rays, "dtype": "fp16", "value": res["value"], "unit": "ray-samples/s", "ms_per_step": res["ms_per_step"], "steps": args.steps,
                     "ms_per_step_spread": res.get("spread")}
            other = [baked] + (other or [])
        occ = res.get("occupancy")
        if occ and fresh is not None:  # amortise the update against the headline's step time
            occ = dict(occ)
            occ["ms_per_step_including_update"] = fresh["ms_per_step"] + occ["ms_partial"] / 16
            occ["value_including_occupancy_update_per_gpu"] = fresh["samples"] / fresh["steps"] / (occ["ms_per_step_including_update"] * 1e-3)
        out = {
            "metric": "ray-samples/s (train)",
            "value": head["value"],
            "unit": "ray-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "ms_per_step_spread": head.get("spread"),
            "occupancy_update": occ,
            "value_including_occupancy_update": (occ["value_including_occupancy_update_per_gpu"] * world) if occ else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp16": "f16", "bf16": "bf16", "fp32": "f32"}[res["dtype"]],
            "data": "synthetic",
            "config": {
                "workload": WORKLOADS[args.mlp],
                "rays_per_batch_per_gpu": args.rays, "global_rays": res["n_global"], "bound": args.bound, "dt_gamma": dt_gamma, "max_steps": 1024,
                "samples_per_step_per_gpu": res["samples_per_step_per_gpu"], "mean_count": res["mean_count"], "parallelism": f"dp{world}",
                "optimizer": ("Adam(eps=1e-15) + GradScaler rules, as HIP kernels on the fp16 gradients (fp32 masters + fp16 copies; bit-identical to torch fused Adam)"
                              if res.get("fused_opt") else "fused Adam(eps=1e-15)+GradScaler" if use_amp else "fused Adam(eps=1e-15)"),
                "replicas_identical_after_run": res.get("replicas_identical"), "collective": (dict(res["collective"], **pre) if (res.get("collective") and pre) else res.get("collective")), "param_l1_after_run": res.get("param_l1"),
                "launch": (f"ngp_harness.accelerate(renderer, steps_per_call={args.steps_per_graph}).step_group: FRESH rays every call ([{args.steps_per_graph}, N, 3] tensors "
                           f"copied into the graphs' static buffers), one replayed HIP graph per {args.steps_per_graph} steps (shade + backward + optimizer), the marches of "
                           f"the next {args.steps_per_graph} batches (handed over one call early) as graphs of their own on a second stream.  The calls cycle 8 PRE-BUILT "
                           "device batches: the graphs have never seen their contents, but the table's access pattern repeats every 8 steps and ray generation is outside "
                           "the step (SURVEY 8(d): synthetic, resident inputs); other_config holds the same loop with never-repeating rays drawn by torch.randint inside "
                           "the timed loop, and the baked-pool loop of rounds 1-3") if fresh is not None else (res["graph"] if res["graph"] else "eager launches"),
                "headline_loop": "fresh rays through ngp_harness.accelerate" if fresh is not None else "pool of ray batches baked into the graphs",
                # what ngp_harness.streams.ensure_pool measured when it placed the side stream and the range streams on hardware queues (streams.py: two
                # queues 4 apart cost the step 0.97 ms instead of 0.52, two ranges on one queue cost the frame 14 Mpix/s)
                "stream_pool": pool_report(dev),
                "precision_note": "fp16 autocast = the reference's -O/--fp16 (its ffmlp is fp16-only); BASELINE configs[2] says bf16: the same loop with bf16 networks is in other_config",
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "rendered": mpix,
            "other_config": other,
        }
        print(json.dumps(out))
    if res.get("replicas_identical") is False:  # (the line above says so too: config.replicas_identical_after_run) -- a multi-GPU number over diverged replicas is not a result
        sys.exit(3)


if __name__ == "__main__":
    main()
