"""`accelerate(renderer, ...)`: ONE call that gives a training loop the benchmarked path.

The reference trainer's step (nerf/utils.py:1011-1022, :559-620) is
    pred = model.render(rays_o, rays_d, staged=False, bg_color=..., perturb=True, force_all_rays=..., **opt)
    loss = criterion(pred['image'], gt);  scaler.scale(loss).backward();  scaler.step(optimizer);  scaler.update()
-- ~90 framework launches per step on the HIP kernels of the drop-in packages, bound by the host's launch rate (1.0-1.4 ms per
8192-ray step depending on the host, against 0.57 ms of kernels).  What the headline adds on top of the drop-in packages is host-side
only, and this module packages it behind one call for a loop that feeds FRESH rays every step (bench.py's headline bakes its ray pool
into the graphs; a trainer cannot):

    trainer = accelerate(renderer)                       # renderer: ngp_harness.model.Renderer over an NGPField
    loss = trainer.step(rays_o, rays_d, target_rgb)      # one training step; a device scalar, nothing is read back
    loss = trainer.step(rays_o, rays_d, target_rgb, next_rays=(o2, d2))   # ... and start marching the NEXT batch beside it
    trainer = accelerate(renderer, steps_per_call=4)     # the benchmarked structure for a loop that has 4 batches at a time:
    loss = trainer.step_group(o4, d4, t4, next_rays=(o4n, d4n))           # [4, N, 3] tensors: one graph for 4 steps, 4 marches ahead

  * the step replayed as HIP graphs, one set per slot of the renderer's 16-entry step-counter ring (renderer.py:656-660): the march of a
    batch (near / far, DDA, sample expansion: it needs the rays and the occupancy grid, NOT the weights) and the rest of the step (field,
    compositing + loss, backward, loss scaler, optimizer) are separate graphs, reading rays / targets from static buffers the call copies into;
  * `next_rays`: a trainer that has its next batch's rays when it calls `step` (one `get_rays` ahead: software pipelining) hands them over
    and their march runs on a second, high-priority stream UNDER this step's kernels -- latency-bound work on issue slots the step leaves
    idle -- instead of in front of the next step; the next call must then pass the same rays (checked).  It is ordered behind the
    production of those rays (an event recorded at the call), not behind this step.  Without `next_rays` the march runs inline;
  * `NGPField(fused_glue=True)`: everything behind the hash-grid gather as one kernel forward, the glue folded into the MLP backward;
  * `HalfLeafAdam` + `FusedAmp`: Adam on the fp16 gradients and GradScaler's device side as two launches (FFMLP fields); a field with
    nn.Linear MLPs (BASELINE configs[1]) gets torch's fused capturable Adam + GradScaler inside the same graph;
  * sample buffers of a FIXED size, the ring's mean count rounded up to 4096 + 4096 (the reference sizes every step's buffers by the
    mean itself and silently drops the rays that do not fit, raymarching.cu:419; with the margin none are dropped); every 16 steps the
    mean is read back (the reference does the same in update_extra_state) and, if it left the size, two eager steps at the new size
    and a new capture follow.
The occupancy update stays the caller's (`renderer.update_extra_state_device()` every 16 steps writes grid and bitfield in place: the
graphs keep reading the same tensors).  A march started by `next_rays` reads the grid as it is at that moment: hand the next rays over
AFTER the update when one is due (the ring's end, where nothing is marched ahead anyway: the read-back comes first there).
Values: the same kernels in the same order as the eager step -- tests/test_gpu_round3.py holds the replayed step to the eager loss
trajectory, with and without `next_rays`.
"""
import torch

RING = 16


class AcceleratedTrainer:
    def __init__(self, renderer, rays_per_batch=None, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, dt_gamma=1 / 128, bg_color=1, perturb=True, max_steps=1024,
                 amp_dtype=torch.float16, graph=True, steps_per_call=1, march_across_ring_end=False, pipeline_adam=0, skip_dead_samples=None,
                 fused_table_update=None, fused_composite_step=None):
        from .model import NGPField

        field = renderer.field
        assert isinstance(field, NGPField), "accelerate() knows the ngp field (hash grid + two MLPs)"
        self.renderer, self.field = renderer, field
        self.dev = next(field.parameters()).device
        self.n_rays = rays_per_batch
        self.dt_gamma, self.bg_color, self.perturb, self.max_steps = dt_gamma, bg_color, perturb, max_steps
        self.amp_dtype = amp_dtype
        self.use_graph = bool(graph)
        bf16 = amp_dtype == torch.bfloat16 and getattr(field, "fused_field_bf16", False)  # bf16 networks over the fp16 table (round 5)
        self.fused = field.mlp == "ffmlp" and ((field.fused_glue and amp_dtype == torch.float16) or bf16)
        if self.fused:
            from .optim import FusedAmp, HalfLeafAdam

            mlp_dt = torch.bfloat16 if bf16 else torch.float16
            self.opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights", mlp_dt), (field.color_net, "weights", mlp_dt)], lr=lr,
                                    betas=betas, eps=eps)
            # bf16 keeps the loss scaler: the table gradient is fp16 and unscaled gradients of ~1e-6 sit in its subnormals (profiles/r04_precision.json)
            self.amp, self.scaler = FusedAmp(self.opt), None
            if field.fused_field or bf16:
                self.amp.attach(field.encoder)  # found_inf raised by the kernels that write the gradients: no separate scan launch
        elif field.mlp == "torch" and amp_dtype == torch.float16 and self._split_k_layers(field):
            # configs[1] (nn.Linear MLPs on PyTorch-ROCm; round 5): the same optimizer path as the FFMLP field -- fp16 leaves for the table and the
            # five weight matrices (SplitKLinear hands its leaf to F.linear under autocast), one Adam launch, the loss scaler's device side in two
            # launches -- instead of torch's capturable fused Adam + GradScaler: no per-step cast of the 48 MB table, no widening of its gradient
            from .optim import FusedAmp, HalfLeafAdam

            self.fused = True
            self.opt = HalfLeafAdam([(field.encoder, "embeddings")] + [(layer, "weight") for layer in self._split_k_layers(field)], lr=lr, betas=betas, eps=eps)
            self.amp, self.scaler = FusedAmp(self.opt), None
        else:
            self.opt = torch.optim.Adam(field.get_params(lr), betas=betas, eps=eps, fused=True, capturable=self.use_graph)
            self.amp, self.scaler = None, torch.amp.GradScaler("cuda", enabled=amp_dtype in (torch.float16, torch.bfloat16))  # (the table gradient is fp16 either way)
        # pipeline_adam = k > 1 (fused path; round 5, an A/B -- OFF by default): the table gradient is summed in k level groups and the Adam update of
        # group g runs on a second stream while group g + 1 is being summed (VALU-bound sums beside an HBM-bound update).  CAVEAT, the reason it is
        # not the default: GradScaler skips the WHOLE step when any gradient element is non-finite; here group g's update has started before the
        # groups behind it have been scanned.  Non-finite incoming gradients are caught before (the MLP backward's scan, which every overflow of
        # dL/dfeatures passes through); what is not is a row of a LATER level group overflowing fp16 as a sum of finite contributions -- then the
        # earlier groups are already updated and the step is skipped for the rest.
        self.pipeline_adam = int(pipeline_adam) if (self.fused and int(pipeline_adam) > 1 and self.amp is not None and field.fused_field) else 0
        if self.pipeline_adam:
            from .dp import TableGradChunks

            self._chunks = TableGradChunks(field.encoder, self.pipeline_adam)
            self._chunks.with_amp = True
            self._adam_stream = torch.cuda.Stream(device=self.dev)
        # skip_dead_samples (round 6; None = on wherever it exists -- the fused FFMLP field): the compositing backward flags the 32-sample steps that
        # carry a gradient and both MLP backward kernels and the hash-grid backward's record builder walk the flagged steps only.  In a trained scene
        # most samples sit behind the point where their ray's transmittance has underflowed and get EXACTLY zero gradients (raymarching.cu:843-870:
        # 42 % of the samples after 100 steps of the bench's scene, 99.6 % after 1000): the backward then costs what the live samples cost.  Exact: the
        # same parameters bit for bit (tests/test_gpu_round6.py); a young field, where every sample carries a gradient, pays one ballot per 64 steps.
        can_skip = bool(self.fused and field.mlp == "ffmlp" and not self.pipeline_adam)
        self.skip_dead_samples = can_skip if skip_dead_samples is None else bool(skip_dead_samples)
        assert not self.skip_dead_samples or can_skip, "skip_dead_samples needs the fused FFMLP field (and no pipeline_adam)"
        renderer.skip_dead_samples = self.skip_dead_samples
        # fused_table_update (round 6; None = on wherever it exists): the hash-grid backward's summing kernel applies Adam to the hashed levels' rows
        # from its LDS tiles -- the record walk of some workgroups beside the parameter stream of others -- instead of writing their gradient for
        # the optimizer launch to read back (optim.FusedAmp.fuse_table_update).  Same parameters bit for bit, skipped overflow steps included; the
        # optimizer state is double-buffered: the fp32 module parameters are current after `trainer.sync()` (state_dict() calls it).
        can_fuse = bool(self.fused and self.amp is not None and hasattr(self.amp, "covered") and not self.pipeline_adam and field.mlp == "ffmlp")
        self.fused_table_update = can_fuse if fused_table_update is None else bool(fused_table_update)
        assert not self.fused_table_update or can_fuse, "fused_table_update needs the fused FFMLP field under FusedAmp (and no pipeline_adam)"
        if self.fused_table_update:
            self.amp.fuse_table_update(field.encoder)
        self._one = torch.ones((), dtype=torch.float32, device=self.dev)
        # fused_composite_step (round 6; None = on under the fused AMP step): compositing forward, render tail and their backward -- three adjacent,
        # latency-bound launches -- as ONE (fused.composite_tail's `one`: the step's root gradient is the tensor `_one`, so the forward's launch can
        # form the loss gradient itself).  Same outputs and gradients bit for bit (tests/test_gpu_round6.py).
        can_step = bool(self.fused and self.amp is not None)
        self.fused_composite_step = can_step if fused_composite_step is None else bool(fused_composite_step)
        assert not self.fused_composite_step or can_step, "fused_composite_step needs the fused AMP step (the root gradient must be known to be one)"
        renderer.root_one = self._one if self.fused_composite_step else None
        renderer.defer_step_loss = bool(self.fused_composite_step)  # (`_shade` reads the loss after the backward: the field's backward may finish it)
        self._graphs, self._M = None, 0
        # steps_per_call = k > 1: `step_group` takes the batches of k consecutive steps at once and replays ONE graph for their shade + backward +
        # optimizer (the hand-over between two graph launches idles the device ~10 us: bench.py's --steps-per-graph), their k marches being
        # graphs of their own that run ahead on the second stream when the caller hands the NEXT group's rays over
        # march_across_ring_end: at the ring's last call, `next_rays` are marched ahead as well (right behind the mean_count read-back) instead
        # of inline at the start of the next ring.  That march reads the occupancy grid while this call's steps still run: a trainer that
        # updates the grid every 16 steps (nerf/utils.py:1011) must then do so BEFORE the ring's last call, not after it.  Off by default.
        self.march_across_ring_end = bool(march_across_ring_end)
        self.group = int(steps_per_call)
        assert self.group in (1, 2, 4, 8, 16), "steps_per_call must divide the 16-entry step-counter ring"
        self._groups = None
        self._rays, self._targets = None, None  # static inputs per ring slot: rays (the march of slot g + 1 may run while slot g's is still read), targets
        self._primed, self._warm = 0, 0
        self._ahead = None  # (slot, data_ptr of rays_o, data_ptr of rays_d) of a march started by `next_rays`
        self._side = None
        if self.dev.type == "cuda":
            from .streams import ensure_pool

            ensure_pool(self.dev)  # (every stream of the package exists BEFORE this trainer's captures: streams.py)
        self.loss = torch.zeros((), dtype=torch.float32, device=self.dev)

    @staticmethod
    def _split_k_layers(field):
        """The nn.Linear field's layers when every one of them is a SplitKLinear (at most 7: HalfLeafAdam takes 8 tensors), else []."""
        from .model import SplitKLinear

        layers = list(field.sigma_net) + list(field.color_net)
        return layers if layers and len(layers) <= 7 and all(isinstance(m, SplitKLinear) for m in layers) else []

    # ---- the two halves of one eager step; mean_count None = the ring's (full-size buffers while it is unknown)
    def _march(self, ro, rd, mean_count=None):
        with torch.autocast("cuda", dtype=self.amp_dtype):
            return self.renderer.march_train(ro, rd, dt_gamma=self.dt_gamma, perturb=self.perturb, max_steps=self.max_steps, mean_count=mean_count)

    def _shade(self, marched, tgt):
        r = self.renderer
        if self.fused:
            for leaf in self.opt.leaves:
                leaf.grad = None
        else:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.amp_dtype):
            image, depth, loss, scaled = r.shade_train(marched, self.bg_color, target=tgt, scale=self.amp.scale if self.amp else None)
        if self.amp and self.pipeline_adam:
            scaled.backward(self._one)  # (the table gradient is only BINNED: TableGradChunks is attached)
            chunks, main, side = self._chunks, torch.cuda.current_stream(), self._adam_stream
            state = chunks.take()
            if state[1] is None:  # a batch the phased backward does not take (small): the gradient is complete, one update
                self.amp.step()
            else:
                for i in range(len(chunks)):
                    chunks.sum_chunk(i, state)
                    ev = torch.cuda.Event()
                    ev.record(main)
                    with torch.cuda.stream(side):
                        side.wait_event(ev)
                        a, b = chunks.rows[i]
                        self.opt.launch_rows(0, a, b, 1.0, self.amp.scale, self.amp.found_inf)
                main.wait_stream(side)
                self.amp.step(exclude=(0,))  # the two MLP weight vectors + the scale / step-counter update
        elif self.amp:
            scaled.backward(self._one)
            self.amp.step()
        else:
            self.scaler.scale(scaled).backward()  # (scale None: `scaled` is the loss itself, with its graph)
            self.scaler.step(self.opt)
            self.scaler.update()
        self.loss.copy_(loss.detach().reshape(()))

    def _capture(self):
        """Record the graphs.  Nothing is executed here: the two eager steps at this buffer size that `step` ran just before (real
        training steps) have sized the library's workspaces and done every lazy initialisation outside the capture."""
        from .streams import capture_section

        r = self.renderer
        with capture_section():
            keep_step = r.local_step
            graphs, pool, pool_m = [], None, None
            for g in range(RING):  # per ring slot: the step's counter is slot g, as in the eager loop
                r.local_step = g
                gm = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gm, pool=pool_m, capture_error_mode="thread_local"):  # (own memory pool: it may run beside the other graph)
                    marched, _ = self._march(*self._rays[g], mean_count=self._M)
                pool_m = gm.pool()
                ga = None
                if self.group == 1:
                    ga = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                        self._shade(marched, self._targets[g])
                    pool = ga.pool()
                graphs.append((gm, ga, marched))  # (the sample tensors stay alive: the second graph reads them)
            self._groups = None
            if self.group > 1:  # shade + backward + optimizer of `group` consecutive steps per graph
                self._groups = []
                for g0 in range(0, RING, self.group):
                    gg = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gg, pool=pool, capture_error_mode="thread_local"):
                        for g in range(g0, g0 + self.group):
                            self._shade(graphs[g][2], self._targets[g])
                    pool = gg.pool()
                    self._groups.append(gg)
            self._graphs = graphs
            r.local_step = keep_step % RING

    def _ensure_buffers(self, n_rays):
        if self._rays is None:
            self.n_rays = n_rays
            self._ray_o = torch.empty(RING, n_rays, 3, dtype=torch.float32, device=self.dev)
            self._ray_d = torch.empty(RING, n_rays, 3, dtype=torch.float32, device=self.dev)
            self._rays = [(self._ray_o[g], self._ray_d[g]) for g in range(RING)]
            self._targets = torch.empty(RING, n_rays, 3, dtype=torch.float32, device=self.dev)
        assert n_rays == self.n_rays, "a captured step has a fixed batch size"

    def _ring_end(self, ready):
        """One read-back per 16 steps, as in the reference.  Every march of the ring has run by now, so with a second stream at hand the
        read-back waits for the marches only, not for the last step's backward.  Nothing is marched ahead across the ring's end: that is
        where the trainer updates the occupancy grid (nerf/utils.py:1011, before the next step), and a march started here would read the grid
        while the update writes it."""
        r = self.renderer
        if ready is not None:
            with torch.cuda.stream(self._side_stream()):
                self._side.wait_event(ready)
                r.update_mean_count()
        else:
            r.update_mean_count()
        self._resize()

    def step_group(self, rays_o, rays_d, target, next_rays=None):
        """steps_per_call = k consecutive training steps in one call: rays_o / rays_d / target [k, N, 3] -- batch i is step i's (FRESH rays every
        call: they are copied into the graphs' static buffers).  next_rays = (rays_o, rays_d) [k, N, 3] of the NEXT call: their k marches start
        now, on the second stream, beside this group's kernels; the next call must pass those very tensors.  Same arithmetic as k calls of
        `step`: the same kernels in the same order on the same data (tests/test_gpu_round4.py).  Returns the last step's loss (device scalar)."""
        r, k = self.renderer, self.group
        assert k > 1 and rays_o.shape[0] == k and rays_o.dim() == 3, "step_group: [steps_per_call, N, 3] rays (steps_per_call > 1)"
        assert rays_o.is_contiguous() and rays_d.is_contiguous(), "step_group: rays_o / rays_d must be contiguous [k, N, 3] tensors"
        self._ensure_buffers(rays_o.shape[1])
        if not self.use_graph or self._primed < RING or self._warm < max(2, k):
            # the reference's first steps (full-size buffers until the ring holds a mean count), then `k` eager steps at the size the graphs
            # will be recorded with -- one by one, on the eager path; k of them so that the ring slot is a multiple of k when the graphs start
            assert self._ahead is None
            for i in range(k):
                self.step(rays_o[i], rays_d[i], target[i], _eager=True)
            return self.loss
        assert r.local_step % k == 0, "step_group and step must not be mixed once the graphs run (the ring slot must stay a multiple of steps_per_call)"
        if self._graphs is None:
            self._capture()
        g0 = r.local_step
        main = torch.cuda.current_stream()
        if self._ahead is not None and self._ahead == (g0, rays_o.data_ptr(), rays_d.data_ptr()):
            main.wait_stream(self._side)  # marched beside the previous group
        else:
            if self._ahead is not None:
                self._ahead = None
                main.wait_stream(self._side)
                raise AssertionError("next_rays of the previous call must be the rays of this call (same tensors)")
            self._ray_o[g0:g0 + k].copy_(rays_o, non_blocking=True), self._ray_d[g0:g0 + k].copy_(rays_d, non_blocking=True)
            for g in range(g0, g0 + k):
                self._graphs[g][0].replay()
        self._ahead = None
        self._targets[g0:g0 + k].copy_(target, non_blocking=True)
        last = g0 + k == RING
        ready = None
        if next_rays is not None:
            ready = torch.cuda.Event()
            ready.record(main)  # everything enqueued so far (the production of the next rays, an occupancy update) -- NOT this group's kernels
        self._groups[g0 // k].replay()
        r.local_step = g0 + k
        def march_ahead(slot0):
            no, nd = next_rays
            assert no.shape == rays_o.shape and no.is_contiguous() and nd.is_contiguous(), "next_rays: contiguous [k, N, 3] tensors"
            with torch.cuda.stream(self._side_stream()):
                self._side.wait_event(ready)
                self._ray_o[slot0:slot0 + k].copy_(no, non_blocking=True), self._ray_d[slot0:slot0 + k].copy_(nd, non_blocking=True)
                for g in range(slot0, slot0 + k):
                    self._graphs[g][0].replay()
            self._ahead = (slot0, no.data_ptr(), nd.data_ptr())

        if not last:
            if ready is not None:
                march_ahead(g0 + k)
        else:
            self._ring_end(ready)
            if ready is not None and self.march_across_ring_end and self._graphs is not None:  # (graphs dropped: the buffer size changed)
                march_ahead(0)
        return self.loss

    def step(self, rays_o, rays_d, target, next_rays=None, _eager=False):
        """One training step on a batch of rays [N,3], [N,3] and their target colours [N,3] (device tensors; N fixed after the first call).
        next_rays = (rays_o, rays_d) of the batch the NEXT call will pass: its march starts now, beside this step (module docstring).
        Returns the loss as a device scalar that the NEXT call overwrites."""
        r = self.renderer
        # contiguous views: the hand-over of `next_rays` is recognised by the address of these tensors (a reshape of a non-contiguous
        # tensor would be a fresh copy with a fresh address every call)
        rays_o, rays_d, target = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), target.reshape(-1, 3)
        assert rays_o.is_contiguous() and rays_d.is_contiguous(), "accelerate().step: rays_o / rays_d must be contiguous [N,3] tensors"
        self._ensure_buffers(rays_o.shape[0])
        main = torch.cuda.current_stream()
        if _eager or not self.use_graph or self._primed < RING or self._warm < 2:
            # the reference's first steps: full-size sample buffers until the ring holds a mean count (its update_extra_state cadence);
            # then -- and after every change of the buffer size -- two eager steps at the size the graphs will be recorded with
            sized = self._primed >= RING  # (graph=False keeps the same buffer-size policy, launched eagerly)
            ro, rd = self._rays[r.local_step % RING]
            tg = self._targets[r.local_step % RING]
            ro.copy_(rays_o, non_blocking=True), rd.copy_(rays_d, non_blocking=True), tg.copy_(target, non_blocking=True)
            marched, _ = self._march(ro, rd, mean_count=self._M if sized else None)
            self._shade(marched, tg)
            self._primed += 1
            self._warm += 1 if sized else 0
            if r.local_step == RING:
                r.update_mean_count()
                self._resize()
            return self.loss
        assert self.group == 1, "this trainer was built with steps_per_call > 1: call step_group"
        if self._graphs is None:
            self._capture()
        g = r.local_step
        gm, ga, _ = self._graphs[g]
        if self._ahead is not None and self._ahead == (g, rays_o.data_ptr(), rays_d.data_ptr()):
            main.wait_stream(self._side)  # marched beside the previous step
        else:
            if self._ahead is not None:  # the march that ran ahead was of other tensors: forget it (the next call starts clean), then refuse
                self._ahead = None
                main.wait_stream(self._side)
                raise AssertionError("next_rays of the previous call must be the rays of this call (same tensors)")
            self._rays[g][0].copy_(rays_o, non_blocking=True), self._rays[g][1].copy_(rays_d, non_blocking=True)
            gm.replay()
        self._ahead = None
        self._targets[g].copy_(target, non_blocking=True)
        last = g + 1 == RING
        ready = None
        if next_rays is not None:
            ready = torch.cuda.Event()
            ready.record(main)  # everything enqueued so far (the production of the next rays, an occupancy update, this batch's march) -- NOT the rest of this step
        ga.replay()
        r.local_step = g + 1
        if not last:
            if ready is not None:
                self._march_ahead(g + 1, next_rays, ready)
        else:
            self._ring_end(ready)
        return self.loss

    def sync(self):
        """Make the fp32 module parameters and the optimizer's moment tensors current (double-buffered optimizer state, `fused_table_update`): one
        4-byte read-back.  The 16-bit copies the kernels read are always current."""
        if self.fused:
            self.opt.sync()

    def _side_stream(self):
        if self._side is None:
            from .streams import side_stream

            self._side = side_stream(self.dev)  # the process-wide high-priority stream (streams.py: why it is shared)
        return self._side

    def _march_ahead(self, slot, next_rays, ready):
        no, nd = next_rays[0].reshape(-1, 3), next_rays[1].reshape(-1, 3)
        assert no.shape[0] == self.n_rays
        assert next_rays[0].is_contiguous() and next_rays[1].is_contiguous(), "next_rays must be contiguous (they are recognised by address in the next call)"
        with torch.cuda.stream(self._side_stream()):
            self._side.wait_event(ready)
            self._rays[slot][0].copy_(no, non_blocking=True), self._rays[slot][1].copy_(nd, non_blocking=True)
            self._graphs[slot][0].replay()
        self._ahead = (slot, no.data_ptr(), nd.data_ptr())

    def _resize(self):
        """After a mean_count read-back: (re)choose the sample-buffer size; a change drops the graphs (two eager steps, then a new capture)."""
        r = self.renderer
        if self._M == 0 or r.mean_count + 128 > self._M or r.mean_count < 0.8 * self._M:
            if self._M == 0:
                # the first ring ran on full-size buffers (N * max_steps rows, as the reference's first steps do): the library's scratch grew to
                # match -- gigabytes of binning records -- and would stay that size; give it back once, now that a sample count exists
                from nerftex_hip import check, lib

                torch.cuda.synchronize()
                check(lib.nerftex_release_workspaces())
            self._M = (r.mean_count + 4095) // 4096 * 4096 + 4096
            self._graphs, self._groups, self._warm = None, None, 0


def accelerate(renderer, **kw):
    """See the module docstring.  Keyword arguments: rays_per_batch, lr, betas, eps, dt_gamma, bg_color, perturb, max_steps, amp_dtype, graph,
    steps_per_call (k > 1: `step_group` takes the batches of k consecutive steps and replays one graph for them), march_across_ring_end."""
    return AcceleratedTrainer(renderer, **kw)
