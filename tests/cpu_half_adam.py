"""TEST INFRASTRUCTURE: torch stand-ins for the two HIP launches of ngp_harness.optim (HalfLeafAdam._launch, FusedAmp._check), so that
the HOST-side logic around them -- fp16 leaves as autograd leaves, their gradients on a real process group, the loss scaler's skip /
back-off agreeing across ranks, state dicts -- runs in the CPU suite under gloo.  Same formulas as csrc/trainstep.hip
(adam_half_kernel, amp_update, amp_check_half_kernel); not bit-identical to them and not meant to be (tests/test_gpu_trainstep.py
holds the kernels to torch's fused Adam bit for bit)."""
import torch

from ngp_harness.optim import FusedAmp, HalfLeafAdam


class CpuHalfLeafAdam(HalfLeafAdam):
    _needs_device = False

    def _launch(self, step_offset, grad_scale, found_inf, amp=None, exclude=()):
        idx = [i for i, leaf in enumerate(self.leaves) if leaf.grad is not None and i not in exclude]
        grp = self.param_groups[0]
        lr, (b1, b2), eps = float(grp["lr"]), grp["betas"], grp["eps"]
        if amp is not None:
            scale, tracker, found, _ticket, growth, backoff, interval = amp
            grad_scale, found_inf, step_offset = scale, found, 1.0
        skip = found_inf is not None and float(found_inf) == 1.0
        if not skip:
            steps = float(self.step_count) + step_offset
            bc1, bc2_sqrt = 1 - b1 ** steps, (1 - b2 ** steps) ** 0.5
            for i in idx:
                g = self.leaves[i].grad.float()
                if grad_scale is not None:
                    g = g / float(grad_scale)
                self.exp_avg[i].lerp_(g, 1 - b1)
                self.exp_avg_sq[i].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.exp_avg_sq[i].sqrt() / bc2_sqrt).add_(eps)
                self.masters[i].data.addcdiv_(self.exp_avg[i], denom, value=-lr / bc1)
                self.leaves[i].data.copy_(self.masters[i].data)
        if amp is not None:  # amp_update: back off / grow the scale; a skipped step does not count
            if float(found) != 0.0:
                scale.mul_(backoff)
                tracker.zero_()
            else:
                ok = int(tracker) + 1
                if ok == interval:
                    grown = float(scale) * growth
                    if grown != float("inf"):
                        scale.fill_(grown)
                    tracker.zero_()
                else:
                    tracker.fill_(ok)
                self.step_count += 1
            found.zero_()


class CpuFusedAmp(FusedAmp):
    def _check(self, grads):
        if any(not torch.isfinite(g).all() for g in grads):
            self.found_inf.fill_(1.0)
