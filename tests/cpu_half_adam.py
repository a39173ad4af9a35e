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
        if self.live is not None:
            return self._launch_double_buffered(idx, amp, exclude)
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


    # ---- double-buffered form (round 6): the stand-in of nerftex_adam_mixed_step_amp_db -- reads the live set, writes the other one, flips `live` iff the
    # step is applied; a table whose rows from `first_row` on an earlier kernel of the step has updated already is passed up to that row only, and on a
    # skipped step the 16-bit copy of the rows behind it is re-derived from the live set (the repair)
    def _launch_double_buffered(self, idx, amp, exclude):
        assert amp is not None and not exclude and idx == list(range(len(self.leaves)))
        fused, self.fused_table = self.fused_table, None
        grp = self.param_groups[0]
        lr, (b1, b2), eps = float(grp["lr"]), grp["betas"], grp["eps"]
        scale, tracker, found, _ticket, growth, backoff, interval = amp
        live = int(self.live) & 1
        skip = float(found) == 1.0
        if not skip:
            steps = float(self.step_count) + 1.0
            bc1, bc2_sqrt = 1 - b1 ** steps, (1 - b2 ** steps) ** 0.5
            for i in idx:
                rows = slice(0, fused[1]) if (fused is not None and fused[0] == i) else slice(None)
                g = self.leaves[i].grad.float()[rows] / float(scale)
                m = self._m[live][i][rows].clone().lerp_(g, 1 - b1)  # (the single-buffered stand-in's operations, on copies)
                v = self._v[live][i][rows].clone().mul_(b2).addcmul_(g, g, value=1 - b2)
                p = self._p[live][i][rows].clone().addcdiv_(m, (v.sqrt() / bc2_sqrt).add_(eps), value=-lr / bc1)
                self._m[live ^ 1][i][rows], self._v[live ^ 1][i][rows], self._p[live ^ 1][i][rows] = m, v, p
                self.leaves[i].data[rows] = p.to(self.leaves[i].dtype)
        elif fused is not None:
            i, first = fused
            self.leaves[i].data[first:] = self._p[live][i][first:].to(self.leaves[i].dtype)
        if float(found) != 0.0:
            scale.mul_(backoff)
            tracker.zero_()
        else:
            ok = int(tracker) + 1
            if ok == interval:
                scale.mul_(growth)
                tracker.zero_()
            else:
                tracker.fill_(ok)
            self.step_count += 1
            self.live ^= 1
        found.zero_()


class CpuFusedAmp(FusedAmp):
    def _check(self, grads):
        if any(not torch.isfinite(g).all() for g in grads):
            self.found_inf.fill_(1.0)
