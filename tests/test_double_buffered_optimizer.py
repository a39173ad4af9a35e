"""CPU: the HOST side of the double-buffered optimizer (ngp_harness/optim.py, round 6) on the torch stand-ins of its two launches (tests/cpu_half_adam.py):
which state set the module parameters point at, what a skipped step leaves behind, state dicts in both directions, and the bookkeeping of a table whose
hashed rows the backward has already updated (`fused_table`: the cut of the closing launch and the repair of the 16-bit copy on a skipped step).  The
kernels themselves are held to each other bit for bit on the GPU (tests/test_gpu_round6.py)."""
import copy

import torch

from cpu_half_adam import CpuFusedAmp, CpuHalfLeafAdam


class _Owner(torch.nn.Module):
    def __init__(self, n, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w = torch.nn.Parameter(torch.rand(n, 2, generator=g) - 0.5)


def _pair(db):
    a, b = _Owner(64, 1), _Owner(16, 2)
    opt = CpuHalfLeafAdam([(a, "w"), (b, "w")], lr=1e-2)
    amp = CpuFusedAmp(opt, init_scale=8.0, growth_interval=3)
    if db:
        opt.enable_double_buffer()
    return a, b, opt, amp


def _grads(step, opt, poison=False):
    g = torch.Generator().manual_seed(100 + step)
    for leaf in opt.leaves:
        leaf.grad = (torch.randn(leaf.shape, generator=g) * 0.1 * 8.0).to(leaf.dtype)
    if poison:
        opt.leaves[0].grad[3, 1] = float("inf")


def test_double_buffered_state_follows_the_single_buffered_optimizer():
    ref, dbl = _pair(False), _pair(True)
    for step in range(9):
        for a, b, opt, amp in (ref, dbl):
            _grads(step, opt, poison=step in (2, 5))
            amp.step()
        assert float(ref[2].step_count) == float(dbl[2].step_count) and float(ref[3].scale) == float(dbl[3].scale)
        for la, lb in zip(ref[2].leaves, dbl[2].leaves):
            assert torch.equal(la.data, lb.data), step  # the 16-bit copies the kernels read are always current
    assert int(dbl[2].live) == 7 % 2, "seven applied steps of nine"
    # the module parameters may point at the stale set until sync() ...
    dbl[2].sync()
    for (na, pa), (_, pb) in zip(list(ref[0].named_parameters()) + list(ref[1].named_parameters()), list(dbl[0].named_parameters()) + list(dbl[1].named_parameters())):
        assert torch.equal(pa, pb), na
    for x, y in zip(ref[2].exp_avg + ref[2].exp_avg_sq, dbl[2].exp_avg + dbl[2].exp_avg_sq):
        assert torch.equal(x, y)


def test_state_dicts_of_the_owners_and_of_the_optimizer_see_the_live_set():
    ref, dbl = _pair(False), _pair(True)
    for step in range(3):  # an odd number of applied steps: the live set is set 1
        for a, b, opt, amp in (ref, dbl):
            _grads(step, opt)
            amp.step()
    assert int(dbl[2].live) == 1
    assert torch.equal(dbl[0].state_dict()["w"], ref[0].state_dict()["w"]), "the owner's state_dict() pre-hook syncs"
    sd_ref, sd = ref[2].state_dict(), dbl[2].state_dict()
    for i in range(2):
        assert torch.equal(sd["state"][i]["exp_avg"], sd_ref["state"][i]["exp_avg"]) and torch.equal(sd["state"][i]["exp_avg_sq"], sd_ref["state"][i]["exp_avg_sq"])
    # ... and a state loaded INTO a double-buffered optimizer lands in its live set: training continues identically
    fresh = _pair(True)
    for step in range(1):
        _grads(50, fresh[2])
        fresh[3].step()  # (live = 1 when the state arrives)
    fresh[0].load_state_dict(copy.deepcopy(ref[0].state_dict())), fresh[1].load_state_dict(copy.deepcopy(ref[1].state_dict()))
    fresh[2].load_state_dict(copy.deepcopy(sd_ref))
    fresh[3].load_state_dict(ref[3].state_dict())
    for step in range(3, 6):
        for a, b, opt, amp in (ref, fresh):
            _grads(step, opt)
            amp.step()
    fresh[2].sync()
    assert torch.equal(fresh[0].w, ref[0].w) and torch.equal(fresh[1].w, ref[1].w)


def test_rows_the_backward_updated_are_cut_from_the_closing_launch_and_repaired_on_a_skip():
    """`fused_table = (leaf, first_row)`: rows from first_row on were written (other state set + the 16-bit copy, in place) by an earlier kernel of the
    step.  Applied step: the closing launch must leave them alone.  Skipped step: nothing flips and the 16-bit copy of those rows comes back."""
    a, b, opt, amp = _pair(True)
    first = 40
    for step, poison in ((0, False), (1, True), (2, False)):
        _grads(step, opt, poison=False)
        live = int(opt.live)
        before = opt.leaves[0].data.clone()
        want_tail = torch.full_like(opt._p[live ^ 1][0][first:], 0.25 + step)  # what "the backward's tiles" wrote
        opt._p[live ^ 1][0][first:] = want_tail
        opt.leaves[0].data[first:] = want_tail.to(opt.leaves[0].dtype)
        opt.fused_table = (0, first)
        amp.covered = (opt.leaves[0].grad.data_ptr(),)  # (the backward marks its own gradient buffer as scanned: most of it is uninitialised memory)
        if poison:
            amp.found_inf.fill_(1.0)  # (raised late, by a tile: the closing launch only sees the flag)
        amp.step()
        assert opt.fused_table is None
        if poison:
            assert int(opt.live) == live and torch.equal(opt.leaves[0].data, before), "a skipped step leaves no trace, the rewritten rows included"
        else:
            assert int(opt.live) == live ^ 1
            assert torch.equal(opt._p[live ^ 1][0][first:], want_tail) and torch.equal(opt.leaves[0].data[first:], want_tail.to(opt.leaves[0].dtype))
            assert not torch.equal(opt.leaves[0].data[:first], before[:first]), "the rows in front were updated by the closing launch"
