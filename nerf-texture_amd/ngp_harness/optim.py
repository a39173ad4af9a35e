"""Optimizer + loss scaling of the ngp field for fp16 (autocast) training: `HalfLeafAdam`, `FusedAmp`.

main_nerf.py:128 trains everything with `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` under a GradScaler
(nerf/utils.py:360, :1003-1009).  For the hash table that costs, per step, a widening copy of the fp16 gradient the encoder backward
produced, a non-finite scan, the fused Adam kernel and a narrowing copy of the fp32 table for the next forward: ~650 MB of traffic
for a 12.6 M-parameter table, plus ~20 small launches of scaler bookkeeping.

`HalfLeafAdam` keeps the same update (arithmetic restated from torch's fused kernel, bit-identical: csrc/trainstep.hip,
tests/test_gpu_trainstep.py) but makes the fp16 copy of every parameter the autograd leaf: the kernels read it, the backward's fp16
gradient lands in its `.grad` as is, and one launch updates the fp32 masters + moments and rewrites the fp16 leaves (28 B per
parameter).  It is a `torch.optim.Optimizer` a GradScaler can drive (`_step_supports_amp_scaling`, like `Adam(fused=True)`).

`FusedAmp` is GradScaler's device side (non-finite check, skip, scale back-off / growth: same constants, same formulas) as two
launches per step: the non-finite check, then Adam with the scale update done by its last block.
"""
import ctypes

import torch

from nerftex_hip import check, lib, ptr, stream


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def _poll_deferred_error():
    """A hash-grid backward whose level table no longer matched its registration wrote NO gradient and left a deferred error
    (include/nerftex_hip.h, nerftex_deferred_error): raise it HERE, before the update consumes that gradient tensor (a host read of one
    pinned word; an eager step sees a launch of an earlier step at the latest -- the launches of this step may still be in flight)."""
    check(lib.nerftex_deferred_error())


_ADAM_GROUP_DEFAULTS = dict(weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                            decoupled_weight_decay=False)


class HalfLeafAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True
    _needs_device = True  # the update is a HIP kernel; tests of the host-side / wire logic subclass this with a torch stand-in for _launch

    def __init__(self, owners, lr=1e-2, betas=(0.9, 0.99), eps=1e-15):
        """owners: [(module, attribute name)] or [(module, attribute name, 16-bit dtype)] of fp32 parameters, e.g. (encoder, "embeddings"),
        (sigma_net, "weights", torch.bfloat16); each module gets a `half_leaf` attribute that GridEncoder._table / FFMLP._weights hand to the
        kernels under autocast.  The 16-bit type is per owner (round 5): a bf16 field keeps its hash table in fp16 -- gridencoder/grid.py:38-41
        casts it to half under ANY autocast -- and its MLP weights in bf16; one launch updates all of them."""
        assert 1 <= len(owners) <= 8
        self.masters, self.leaves, self._owners = [], [], []
        self.bf16_mask = 0
        for i, owner in enumerate(owners):
            mod, name = owner[0], owner[1]
            dt = owner[2] if len(owner) > 2 else torch.half
            assert dt in (torch.half, torch.bfloat16)
            self.bf16_mask |= (1 << i) if dt == torch.bfloat16 else 0
            master = getattr(mod, name)
            assert (master.is_cuda or not self._needs_device) and master.dtype == torch.float32 and master.is_contiguous()
            leaf = master.detach().to(dt).requires_grad_(True)
            mod.half_leaf = leaf
            self._owners.append(mod)
            self.masters.append(master)
            self.leaves.append(leaf)
        super().__init__([{"params": list(self.leaves)}], dict(lr=lr, betas=betas, eps=eps))
        self.exp_avg = [torch.zeros_like(m.data) for m in self.masters]
        self.exp_avg_sq = [torch.zeros_like(m.data) for m in self.masters]
        self.step_count = torch.zeros((), dtype=torch.float32, device=self.masters[0].device)  # completed steps; device-side for graph replay
        self.live = None  # double-buffered form (enable_double_buffer): device word, which of the two state sets holds the current state
        self.fused_table = None  # (leaf index, first row) of a table whose rows from there on this step's backward has updated already

    # ---- double-buffered state (round 6): what lets the hash-grid backward apply the update of the hashed levels from its LDS tiles -----------
    def enable_double_buffer(self):
        """Give every parameter a SECOND set of (fp32 master, exp_avg, exp_avg_sq) and a device word `live` that says which set is current.  Every
        update then reads the live set and writes the other one, and the word flips -- on the device, by the step's last launch -- only when the
        step is applied: a step GradScaler has to skip leaves no trace, whichever kernels of it had already written (csrc/gridencoder_binned.hip
        TileAdam, nerftex_adam_mixed_step_amp_db).  +12 B per parameter (151 MB for the fox table).
        The module parameters (`encoder.embeddings`, ...) and `exp_avg` / `exp_avg_sq` are Python references to ONE of the two sets: call `sync()`
        (one 4-byte read-back) before looking at them -- `state_dict()`, `load_state_dict()`, `resync()` and the owners' `state_dict()` do."""
        if self.live is not None:
            return self
        dev = self.masters[0].device
        self._p = [[m.data for m in self.masters], [torch.empty_like(m.data) for m in self.masters]]
        self._m = [list(self.exp_avg), [torch.empty_like(t) for t in self.exp_avg]]
        self._v = [list(self.exp_avg_sq), [torch.empty_like(t) for t in self.exp_avg_sq]]
        self.live = torch.zeros((), dtype=torch.int32, device=dev)
        # reading OR loading an owner's state dict must see / land in the live set (a checkpoint loaded into the stale set would be lost at the next sync)
        self._hooks = [mod.register_state_dict_pre_hook(lambda *_a, **_k: self.sync()) for mod in self._owners]
        self._hooks += [mod.register_load_state_dict_pre_hook(lambda *_a, **_k: self.sync()) for mod in self._owners]
        return self

    def sync(self):
        """Double-buffered form: point the module parameters and `exp_avg` / `exp_avg_sq` at the live state set (reads one device word: blocks
        until the steps launched so far are done).  Replayed graphs are not affected: they hold both sets' addresses."""
        if self.live is None:
            return
        live = int(self.live.item()) & 1
        for i, master in enumerate(self.masters):
            master.data = self._p[live][i]
            self.exp_avg[i] = self._m[live][i]
            self.exp_avg_sq[i] = self._v[live][i]

    def table_adam(self, i, amp):
        """nerftex_table_adam for leaf i (the hash table) under the loss scaler `amp`: what nerftex_grid_encode_backward_adam needs to update the
        table's hashed rows itself."""
        from nerftex_hip import TableAdam

        assert self.live is not None, "enable_double_buffer() first"
        grp = self.param_groups[0]
        t = TableAdam()
        for k in range(2):
            t.param[k], t.exp_avg[k], t.exp_avg_sq[k] = self._p[k][i].data_ptr(), self._m[k][i].data_ptr(), self._v[k][i].data_ptr()
        t.param_half, t.live, t.step = self.leaves[i].data_ptr(), self.live.data_ptr(), self.step_count.data_ptr()
        t.grad_scale, t.found_inf = amp.scale.data_ptr(), amp.found_inf.data_ptr()
        t.lr, t.beta1, t.beta2, t.eps = float(grp["lr"]), grp["betas"][0], grp["betas"][1], grp["eps"]
        return t

    def trainable(self):
        """The tensors whose `.grad` the backward pass fills (what a gradient all-reduce has to cover)."""
        return list(self.leaves)

    @torch.no_grad()
    def resync(self):
        """Re-derive the fp16 leaves from the fp32 masters: after `load_state_dict` on the owning modules (a checkpoint) the kernels
        would otherwise keep reading the old copies."""
        self.sync()
        for master, leaf in zip(self.masters, self.leaves):
            leaf.copy_(master)

    def state_dict(self):
        """torch.optim.Adam's layout (state: {index: {step, exp_avg, exp_avg_sq}}, param_groups), so that a checkpoint written here
        loads into `torch.optim.Adam` over the fp32 parameters and vice versa (nerf/utils.py:1505, 1581-1586)."""
        self.sync()
        step = self.step_count.detach().clone()
        # the full key set of torch.optim.Adam's param_group: Adam.__setstate__ fills in amsgrad / maximize / ... when they are missing but
        # NOT weight_decay, and its step() reads every one of them
        group = dict(_ADAM_GROUP_DEFAULTS)
        group.update({k: v for k, v in self.param_groups[0].items() if k != "params"})
        group["params"] = list(range(len(self.masters)))
        return {"state": {i: {"step": step.clone(), "exp_avg": self.exp_avg[i].detach().clone(), "exp_avg_sq": self.exp_avg_sq[i].detach().clone()}
                          for i in range(len(self.masters))},
                "param_groups": [group]}

    @torch.no_grad()
    def load_state_dict(self, sd):
        self.sync()
        state = sd["state"]
        grp = sd["param_groups"][0]
        if grp.get("weight_decay", 0) or grp.get("amsgrad", False) or grp.get("maximize", False):
            raise ValueError("HalfLeafAdam implements plain Adam: weight_decay / amsgrad / maximize of the loaded state are not supported")
        if len(state) not in (0, len(self.masters)):
            raise ValueError(f"loaded state has {len(state)} parameters, this optimizer {len(self.masters)}")
        for i in range(len(self.masters)):
            st = state.get(i, state.get(str(i))) if state else None
            if st is None:
                self.exp_avg[i].zero_()
                self.exp_avg_sq[i].zero_()
                continue
            self.exp_avg[i].copy_(st["exp_avg"])
            self.exp_avg_sq[i].copy_(st["exp_avg_sq"])
            self.step_count.fill_(float(st["step"]))
        if not state:
            self.step_count.zero_()
        for k, v in grp.items():
            if k in ("lr", "betas", "eps", "initial_lr"):
                self.param_groups[0][k] = v
        self.resync()

    def launch_rows(self, i, a, b, step_offset, grad_scale, found_inf):
        """Adam over rows [a, b) of leaf i alone (current stream): the table updated level group by level group, each group as soon as its rows
        of the gradient are final (accelerate(pipeline_adam=True)).  The step number is *step_count + step_offset; step_count is not advanced."""
        g = self.leaves[i].grad
        assert g is not None and g.dtype == self.leaves[i].dtype and g.is_contiguous()
        if b <= a:
            return
        parts = [t[a:b] for t in (self.masters[i].data, self.exp_avg[i], self.exp_avg_sq[i], g, self.leaves[i].data)]
        grp = self.param_groups[0]
        n = (ctypes.c_uint64 * 1)(parts[0].numel())
        check(lib.nerftex_adam_mixed_step(1, *[_ptr_array([t]) for t in parts], n, (self.bf16_mask >> i) & 1, ptr(self.step_count), float(step_offset),
                                          float(grp["lr"]), grp["betas"][0], grp["betas"][1], grp["eps"], ptr(grad_scale), ptr(found_inf), stream()))

    def _launch(self, step_offset, grad_scale, found_inf, amp=None, exclude=()):
        """One launch over every leaf that has a gradient (but `exclude`: leaves already updated by launch_rows).  amp = (scale, growth_tracker,
        found_inf, ticket, growth, backoff, interval): the loss scaler's update rides along (step number *step_count + 1; the launch advances
        step_count itself)."""
        idx = [i for i, leaf in enumerate(self.leaves) if leaf.grad is not None and i not in exclude]
        if self.live is not None:
            return self._launch_double_buffered(idx, amp, exclude)
        if not idx and amp is None:
            return
        grads = [self.leaves[i].grad for i in idx]
        for i, g in zip(idx, grads):
            assert g.dtype == self.leaves[i].dtype and g.is_contiguous() and g.shape == self.masters[i].shape
        mask = sum(1 << k for k, i in enumerate(idx) if (self.bf16_mask >> i) & 1)
        grp = self.param_groups[0]
        n = (ctypes.c_uint64 * max(len(idx), 1))(*[self.masters[i].numel() for i in idx])
        arrays = (_ptr_array([self.masters[i] for i in idx]), _ptr_array([self.exp_avg[i] for i in idx]),
                  _ptr_array([self.exp_avg_sq[i] for i in idx]), _ptr_array(grads), _ptr_array([self.leaves[i] for i in idx]), n)
        hyper = (float(grp["lr"]), grp["betas"][0], grp["betas"][1], grp["eps"])
        if amp is None:
            check(lib.nerftex_adam_mixed_step(len(idx), *arrays, mask, ptr(self.step_count), float(step_offset), *hyper, ptr(grad_scale), ptr(found_inf),
                                              stream()))
        else:
            scale, tracker, found, ticket, growth, backoff, interval = amp
            check(lib.nerftex_adam_mixed_step_amp(len(idx), *arrays, mask, ptr(self.step_count), *hyper, ptr(scale), ptr(tracker), ptr(found), ptr(ticket),
                                                  growth, backoff, interval, stream()))
        for i in list(idx) + list(exclude):
            torch.autograd.graph.increment_version(self.masters[i])

    def _launch_double_buffered(self, idx, amp, exclude):
        """The step's last launch over double-buffered state (nerftex_adam_mixed_step_amp_db): reads the live set, writes the other one, flips
        `live` iff the step is applied.  A table whose hashed rows the backward has updated already (`fused_table`) is passed from row 0 to the
        first updated row only; the rows behind are the launch's REPAIR range (their fp16 copy is re-derived from the live set on a skipped step)."""
        assert amp is not None and not exclude, "the double-buffered optimizer runs under FusedAmp, every leaf in one launch"
        assert idx == list(range(len(self.leaves))), "double-buffered state: every parameter must be written every step (a leaf without a gradient would go stale when the sets flip)"
        fused, self.fused_table = self.fused_table, None
        cut = {}
        repair = (None, None, None, 0)
        if fused is not None:
            i, first_row = fused
            cut[i] = int(first_row)
            leaf = self.leaves[i]
            if first_row < leaf.shape[0]:
                repair = (leaf.data[first_row:], self._p[0][i][first_row:], self._p[1][i][first_row:], leaf.data[first_row:].numel())
        view = lambda t, i: t[:cut[i]] if i in cut else t  # noqa: E731
        grads = [view(self.leaves[i].grad, i) for i in idx]
        for i, g in zip(idx, grads):
            assert g.dtype == self.leaves[i].dtype and g.is_contiguous()
        mask = sum(1 << k for k, i in enumerate(idx) if (self.bf16_mask >> i) & 1)
        grp = self.param_groups[0]
        n = (ctypes.c_uint64 * max(len(idx), 1))(*[g.numel() for g in grads])
        sets = [_ptr_array([view(s[k][i], i) for i in idx]) for k in range(2) for s in (self._p, self._m, self._v)]  # p0 m0 v0 p1 m1 v1
        scale, tracker, found, ticket, growth, backoff, interval = amp
        check(lib.nerftex_adam_mixed_step_amp_db(len(idx), *sets, _ptr_array(grads), _ptr_array([view(self.leaves[i].data, i) for i in idx]), n, mask,
                                                 ptr(self.step_count), float(grp["lr"]), grp["betas"][0], grp["betas"][1], grp["eps"], ptr(scale), ptr(tracker),
                                                 ptr(found), ptr(ticket), growth, backoff, interval, ptr(self.live), ptr(repair[0]), ptr(repair[1]), ptr(repair[2]),
                                                 int(repair[3]), stream()))
        for i in idx:
            torch.autograd.graph.increment_version(self.masters[i])

    @torch.no_grad()
    def step(self, closure=None):
        """torch.optim protocol (plain, or driven by torch.amp.GradScaler through grad_scale / found_inf)."""
        assert closure is None
        assert self.live is None, "the double-buffered optimizer is driven by FusedAmp.step()"
        _poll_deferred_error()
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        self.step_count += 1
        self._launch(0.0, grad_scale, found_inf)
        if found_inf is not None:  # a skipped step does not count (torch's _fused_adam does the same)
            self.step_count -= found_inf.reshape(())
        return None


TableAdam = HalfLeafAdam


class FusedAmp:
    """Loss scaling with torch.amp.GradScaler's rules (init 65536, x2 after 2000 clean steps, x0.5 on overflow, overflowing steps
    skipped) around a HalfLeafAdam, all on the device: `scale` multiplies the loss (or is handed to fused.render_tail), `step()` =
    GradScaler.step(optimizer) + GradScaler.update()."""

    def __init__(self, optimizer, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        assert isinstance(optimizer, HalfLeafAdam)
        dev = optimizer.masters[0].device
        self.opt = optimizer
        self.scale = torch.full((), float(init_scale), dtype=torch.float32, device=dev)
        self.growth_tracker = torch.zeros((), dtype=torch.int32, device=dev)
        self.found_inf = torch.zeros((), dtype=torch.float32, device=dev)
        self.ticket = torch.zeros((), dtype=torch.int32, device=dev)  # "last block to finish" counter of the Adam launch, self-resetting
        self.consts = (float(growth_factor), float(backoff_factor), int(growth_interval))

    def attach(self, encoder):
        """Let the fused field's backward (ngp_harness/fused.py, over this encoder) raise `found_inf` from the kernels that write the three
        gradients (nerftex_field_backward_amp, nerftex_grid_encode_backward_amp): `step()` then skips its scan of every gradient tensor that
        backward produced (checked by data pointer: a gradient that went through anything else -- an accumulation, a cross-rank sum in fp16
        that can overflow on its own -- is still scanned).  Single-process training; under data parallelism the scan must see the SUMMED
        gradient, so bench.py does not attach there."""
        self.covered = None
        encoder.amp_sink = self
        return self

    def fuse_table_update(self, encoder):
        """Round 6: let the fused field's backward over this encoder apply Adam to the hash table's hashed levels FROM THE TILES of its summing
        kernel (nerftex_grid_encode_backward_adam) instead of writing their gradient for the optimizer launch to read back: the VALU-bound record
        walk of some workgroups then overlaps the HBM-bound parameter stream of others inside one kernel.  Same parameters, bit for bit, overflow
        steps included (tests/test_gpu_round6.py); the optimizer state becomes double-buffered (HalfLeafAdam.enable_double_buffer: read `sync()`
        there).  After such a step the table's `.grad` holds the gradient of the coarse levels' rows only -- the rest of that tensor is
        uninitialised memory.  Single-process training (a gradient all-reduce needs the whole gradient); implies attach()."""
        self.attach(encoder)
        self.opt.enable_double_buffer()
        tables = [i for i, mod in enumerate(self.opt._owners) if mod is encoder]
        assert tables and not (self.opt.bf16_mask >> tables[0]) & 1, "the encoder's table must be one of the optimizer's fp16 leaves"
        self.table_index = tables[0]
        encoder.table_adam = self
        return self

    def table_adam_for(self, table_h):
        """-> nerftex_table_adam when `table_h` is the optimizer's own fp16 leaf (else None: the ordinary backward)."""
        i = getattr(self, "table_index", None)
        if i is None or table_h.data_ptr() != self.opt.leaves[i].data_ptr():
            return None
        return self.opt.table_adam(i, self)

    def scale_loss(self, loss):
        return loss * self.scale

    def state_dict(self):
        """torch.amp.GradScaler.state_dict()'s keys (what the reference trainer saves as 'scaler', nerf/utils.py:1507)."""
        return {"scale": float(self.scale.item()), "growth_factor": self.consts[0], "backoff_factor": self.consts[1], "growth_interval": self.consts[2],
                "_growth_tracker": int(self.growth_tracker.item())}

    def load_state_dict(self, sd):
        self.scale.fill_(float(sd["scale"]))
        self.growth_tracker.fill_(int(sd["_growth_tracker"]))
        self.consts = (float(sd["growth_factor"]), float(sd["backoff_factor"]), int(sd["growth_interval"]))

    def get_scale(self):
        return float(self.scale.item())

    def _check(self, grads):
        n = (ctypes.c_uint64 * len(grads))(*[g.numel() for g in grads])
        mask = sum(1 << k for k, g in enumerate(grads) if g.dtype == torch.bfloat16)
        check(lib.nerftex_amp_check_mixed(len(grads), _ptr_array(grads), n, mask, ptr(self.found_inf), stream()))

    @torch.no_grad()
    def step(self, exclude=()):
        """exclude: indices of leaves whose Adam update has been launched already (HalfLeafAdam.launch_rows, reading this object's scale and
        found_inf): they are neither scanned nor updated here; the scale / step-counter update still is."""
        _poll_deferred_error()
        grads = [leaf.grad for i, leaf in enumerate(self.opt.leaves) if leaf.grad is not None and i not in exclude]
        covered = getattr(self, "covered", None)
        if covered:  # gradients whose producing kernels already raised found_inf (attach): the very tensors, untouched since
            grads = [g for g in grads if g.data_ptr() not in covered]
            self.covered = None
        if self.opt.fused_table is not None:  # most of that tensor is uninitialised memory: it must not be scanned (nor be anything but the backward's own buffer)
            g = self.opt.leaves[self.opt.fused_table[0]].grad
            assert g is not None and all(g.data_ptr() != o.data_ptr() for o in grads), "the table gradient of a fused update must be the backward's own tensor"
        if grads:
            self._check(grads)
        # Adam (skipped on overflow) and the scale / step-counter update in one launch
        self.opt._launch(1.0, self.scale, self.found_inf, (self.scale, self.growth_tracker, self.found_inf, self.ticket, *self.consts), exclude=exclude)
