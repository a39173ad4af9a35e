"""`accelerate(renderer, ...)`: ONE call that gives a training loop the benchmarked path.

The reference trainer's step (nerf/utils.py:1011-1022, :559-620) is
    pred = model.render(rays_o, rays_d, staged=False, bg_color=..., perturb=True, force_all_rays=..., **opt)
    loss = criterion(pred['image'], gt);  scaler.scale(loss).backward();  scaler.step(optimizer);  scaler.update()
-- ~90 framework launches per step on the HIP kernels of the drop-in packages, bound by the host's launch rate (1.0-1.4 ms per
8192-ray step depending on the host, against 0.65 ms of kernels).  What the headline adds on top of the drop-in packages is host-side
only, and this module packages it behind one call for a loop that feeds FRESH rays every step (bench.py's headline bakes its ray pool
into the graphs; a trainer cannot):

    trainer = accelerate(renderer)                       # renderer: ngp_harness.model.Renderer over an NGPField
    loss = trainer.step(rays_o, rays_d, target_rgb)      # one training step; a device scalar, nothing is read back

  * the whole step -- march, field, compositing + loss, backward, loss scaler, optimizer -- replayed as ONE HIP graph per slot of the
    renderer's 16-entry step-counter ring (renderer.py:656-660), reading its rays / targets from static buffers the call copies into;
  * `NGPField(fused_glue=True)`: everything behind the hash-grid gather as one kernel forward, the glue folded into the MLP backward;
  * `HalfLeafAdam` + `FusedAmp`: Adam on the fp16 gradients and GradScaler's device side as two launches (FFMLP fields); a field with
    nn.Linear MLPs (BASELINE configs[1]) gets torch's fused capturable Adam + GradScaler inside the same graph;
  * sample buffers of a FIXED size, the ring's mean count rounded up to 4096 + 4096 (the reference sizes every step's buffers by the
    mean itself and silently drops the rays that do not fit, raymarching.cu:419; with the margin none are dropped); every 16 steps the
    mean is read back (the reference does the same in update_extra_state) and, if it left the size, two eager steps at the new size
    and a new capture follow.
The occupancy update stays the caller's (`renderer.update_extra_state_device()` every 16 steps writes grid and bitfield in place: the
graphs keep reading the same tensors).  Values: the same kernels in the same order as the eager step -- tests/test_gpu_training.py holds
the replayed step to the eager loss trajectory.
"""
import torch

RING = 16


class AcceleratedTrainer:
    def __init__(self, renderer, rays_per_batch=None, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, dt_gamma=1 / 128, bg_color=1, perturb=True, max_steps=1024,
                 amp_dtype=torch.float16, graph=True):
        from .model import NGPField

        field = renderer.field
        assert isinstance(field, NGPField), "accelerate() knows the ngp field (hash grid + two MLPs)"
        self.renderer, self.field = renderer, field
        self.dev = next(field.parameters()).device
        self.n_rays = rays_per_batch
        self.dt_gamma, self.bg_color, self.perturb, self.max_steps = dt_gamma, bg_color, perturb, max_steps
        self.amp_dtype = amp_dtype
        self.use_graph = bool(graph)
        self.fused = field.mlp == "ffmlp" and field.fused_glue and amp_dtype == torch.float16
        if self.fused:
            from .optim import FusedAmp, HalfLeafAdam

            self.opt = HalfLeafAdam([(field.encoder, "embeddings"), (field.sigma_net, "weights"), (field.color_net, "weights")], lr=lr, betas=betas, eps=eps)
            self.amp, self.scaler = FusedAmp(self.opt), None
        else:
            self.opt = torch.optim.Adam(field.get_params(lr), betas=betas, eps=eps, fused=True, capturable=self.use_graph)
            self.amp, self.scaler = None, torch.amp.GradScaler("cuda", enabled=amp_dtype == torch.float16)
        self._one = torch.ones((), dtype=torch.float32, device=self.dev)
        self._graphs, self._M, self._static = None, 0, None
        self._primed, self._warm = 0, 0
        self.loss = torch.zeros((), dtype=torch.float32, device=self.dev)

    # ---- one eager step on (ro, rd, tgt); mean_count None = the ring's (full-size buffers while it is unknown)
    def _body(self, ro, rd, tgt, mean_count=None):
        r = self.renderer
        if self.fused:
            for leaf in self.opt.leaves:
                leaf.grad = None
        else:
            self.opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=self.amp_dtype):
            marched, counter = r.march_train(ro, rd, dt_gamma=self.dt_gamma, perturb=self.perturb, max_steps=self.max_steps, mean_count=mean_count)
            image, depth, loss, scaled = r.shade_train(marched, self.bg_color, target=tgt, scale=self.amp.scale if self.amp else None)
        if self.amp:
            scaled.backward(self._one)
            self.amp.step()
        else:
            self.scaler.scale(scaled).backward()  # (scale None: `scaled` is the loss itself, with its graph)
            self.scaler.step(self.opt)
            self.scaler.update()
        self.loss.copy_(loss.detach().reshape(()))
        return counter

    def _capture(self):
        """Record the 16 graphs.  Nothing is executed here: the two eager steps at this buffer size that `step` ran just before (real
        training steps) have sized the library's workspaces and done every lazy initialisation outside the capture."""
        r = self.renderer
        ro, rd, tgt = self._static
        keep_step = r.local_step
        graphs, pool = [], None
        for g in range(RING):  # one graph per ring slot: the step's counter is slot g, as in the eager loop
            r.local_step = g
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=pool, capture_error_mode="thread_local"):
                self._body(ro, rd, tgt, mean_count=self._M)
            pool = graph.pool()
            graphs.append(graph)
        self._graphs = graphs
        r.local_step = keep_step % RING

    def step(self, rays_o, rays_d, target):
        """One training step on a batch of rays [N,3], [N,3] and their target colours [N,3] (device tensors; N fixed after the first call).
        Returns the loss as a device scalar that the NEXT call overwrites."""
        r = self.renderer
        rays_o, rays_d, target = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), target.reshape(-1, 3)
        if self._static is None:
            self.n_rays = rays_o.shape[0]
            self._static = (torch.empty_like(rays_o, dtype=torch.float32), torch.empty_like(rays_d, dtype=torch.float32),
                            torch.empty_like(target, dtype=torch.float32))
        assert rays_o.shape[0] == self.n_rays, "a captured step has a fixed batch size"
        for dst, src in zip(self._static, (rays_o, rays_d, target)):
            dst.copy_(src, non_blocking=True)
        if not self.use_graph or self._primed < RING or self._warm < 2:
            # the reference's first steps: full-size sample buffers until the ring holds a mean count (its update_extra_state cadence);
            # then -- and after every change of the buffer size -- two eager steps at the size the graphs will be recorded with
            sized = self._primed >= RING  # (graph=False keeps the same buffer-size policy, launched eagerly)
            self._body(*self._static, mean_count=self._M if sized else None)
            self._primed += 1
            self._warm += 1 if sized else 0
            if r.local_step == RING:
                r.update_mean_count()
                self._resize()
            return self.loss
        if self._graphs is None:
            self._capture()
        g = r.local_step
        self._graphs[g].replay()
        r.local_step = g + 1
        if r.local_step == RING:
            r.update_mean_count()  # one read-back per 16 steps, as in the reference
            self._resize()
        return self.loss

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
            self._graphs, self._warm = None, 0


def accelerate(renderer, **kw):
    """See the module docstring.  Keyword arguments: rays_per_batch, lr, betas, eps, dt_gamma, bg_color, perturb, max_steps, amp_dtype, graph."""
    return AcceleratedTrainer(renderer, **kw)
