"""Optimizer of the ngp field for fp16 (autocast) training: `TableAdam`.

main_nerf.py:128 trains everything with `torch.optim.Adam(betas=(0.9, 0.99), eps=1e-15)` under a GradScaler
(nerf/utils.py:360, :1003-1009).  For the hash table that costs, per step, a widening copy of the fp16 gradient the encoder backward
produced, a non-finite scan, the fused Adam kernel and a narrowing copy of the fp32 table for the next forward: ~650 MB of traffic
for a 12.6 M-parameter table.  `TableAdam` keeps the same update (arithmetic restated from torch's fused kernel, see
csrc/trainstep.hip) but makes the fp16 table the autograd leaf: the encoder reads it, the backward's fp16 gradient lands in its
`.grad` as is, and one kernel updates the fp32 master + moments and rewrites the fp16 leaf (353 MB).  The MLP weights (18 K
parameters) go through `torch._fused_adam_` unchanged.  GradScaler sees an optimizer that unscales and skips by itself
(`_step_supports_amp_scaling`), exactly like `Adam(fused=True)`; its non-finite scan runs over the fp16 gradient (25 MB).
"""
import torch

from nerftex_hip import check, lib, ptr, stream


class TableAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, encoder, small_params, lr=1e-2, betas=(0.9, 0.99), eps=1e-15):
        master = encoder.embeddings
        assert master.is_cuda and master.dtype == torch.float32 and master.is_contiguous() and encoder.level_dim % 2 == 0
        leaf = master.detach().to(torch.half).requires_grad_(True)
        encoder.half_leaf = leaf  # GridEncoder._table hands this to grid_encode under autocast
        small = [p for p in small_params if p.requires_grad]
        super().__init__([{"params": [leaf]}, {"params": small}], dict(lr=lr, betas=betas, eps=eps))
        self.master = master
        self.leaf = leaf
        self.exp_avg = torch.zeros_like(master.data)
        self.exp_avg_sq = torch.zeros_like(master.data)
        self.small_avg = [torch.zeros_like(p.data) for p in small]
        self.small_avg_sq = [torch.zeros_like(p.data) for p in small]
        self.step_count = torch.zeros((), dtype=torch.float32, device=master.device)  # device-side: graph replay advances it

    def trainable(self):
        """The tensors whose `.grad` the backward pass fills (what a gradient all-reduce has to cover)."""
        return [self.leaf] + list(self.param_groups[1]["params"])

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        grad_scale = getattr(self, "grad_scale", None)
        found_inf = getattr(self, "found_inf", None)
        self.step_count += 1
        g = self.leaf.grad
        if g is not None:
            grp = self.param_groups[0]
            assert g.dtype == torch.half and g.is_contiguous() and g.shape == self.master.shape
            check(lib.nerftex_table_adam_step(ptr(self.master), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(g), ptr(self.leaf),
                                              self.master.numel(), ptr(self.step_count), float(grp["lr"]), grp["betas"][0], grp["betas"][1],
                                              grp["eps"], ptr(grad_scale), ptr(found_inf), stream()))
            torch.autograd.graph.increment_version(self.master)
        grp = self.param_groups[1]
        idx = [i for i, p in enumerate(grp["params"]) if p.grad is not None]
        if idx:
            torch._fused_adam_([grp["params"][i] for i in idx], [grp["params"][i].grad for i in idx], [self.small_avg[i] for i in idx],
                               [self.small_avg_sq[i] for i in idx], [], [self.step_count] * len(idx), lr=float(grp["lr"]),
                               beta1=grp["betas"][0], beta2=grp["betas"][1], weight_decay=0.0, eps=grp["eps"], amsgrad=False,
                               maximize=False, grad_scale=grad_scale, found_inf=found_inf)
        if found_inf is not None:  # a skipped step does not count (torch's _fused_adam does the same)
            self.step_count -= found_inf.reshape(())
        return None
