"""`torch.amp.custom_fwd` / `custom_bwd` with the same meaning and less host time.

The reference decorates every wrapper with them (raymarching/raymarching.py:22 ..., ffmlp/ffmlp.py:18, shencoder/sphere_harmonics.py:16,
gridencoder/grid.py:21).  torch's versions walk the arguments with a recursive, container-aware `_cast` and enter / leave a full
`autocast` context per call: ~25 us of Python per op, a dozen ops per eager training step.  These do what those do -- cast the eligible
floating-point tensors to `cast_inputs` and run `forward` with autocast off when it was on; run `backward` under the forward's autocast
state -- with a flat pass over the positional arguments and by flipping the thread's autocast flag.  Anything unusual (keyword arguments,
containers among the arguments, a backward whose autocast state has to change) takes torch's own path.
"""
import functools

import torch
from torch.amp import autocast
from torch.amp.autocast_mode import _cast

_PLAIN = (int, float, bool, str, bytes, type(None))


def custom_fwd(fwd=None, *, device_type, cast_inputs=None):
    if not isinstance(device_type, str):
        raise ValueError(f"Expected `device_type` of type `str`, got: `{type(device_type)}`")
    if fwd is None:
        return functools.partial(custom_fwd, device_type=device_type, cast_inputs=cast_inputs)

    @functools.wraps(fwd)
    def decorate_fwd(*args, **kwargs):
        ctx = args[0]
        ctx._dtype = torch.get_autocast_dtype(device_type)
        enabled = torch.is_autocast_enabled(device_type)
        if cast_inputs is None:
            ctx._fwd_used_autocast = enabled
            return fwd(*args, **kwargs)
        ctx._fwd_used_autocast = False
        if not enabled:
            return fwd(*args, **kwargs)
        if kwargs:
            with autocast(device_type=device_type, enabled=False):
                return fwd(*_cast(args, device_type, cast_inputs), **_cast(kwargs, device_type, cast_inputs))
        cast = [ctx]
        for a in args[1:]:
            if isinstance(a, torch.Tensor):
                if a.dtype is not cast_inputs and a.dtype is not torch.float64 and a.is_floating_point() and a.device.type == device_type:
                    a = a.to(cast_inputs)
            elif not isinstance(a, _PLAIN):
                a = _cast(a, device_type, cast_inputs)
            cast.append(a)
        torch.set_autocast_enabled(device_type, False)
        try:
            return fwd(*cast)
        finally:
            torch.set_autocast_enabled(device_type, True)

    return decorate_fwd


def custom_bwd(bwd=None, *, device_type):
    if not isinstance(device_type, str):
        raise ValueError(f"Expected `device_type` of type `str`, got: `{type(device_type)}`")
    if bwd is None:
        return functools.partial(custom_bwd, device_type=device_type)

    @functools.wraps(bwd)
    def decorate_bwd(*args, **kwargs):
        ctx = args[0]
        want = ctx._fwd_used_autocast
        if want == torch.is_autocast_enabled(device_type) and (not want or torch.get_autocast_dtype(device_type) == ctx._dtype):
            return bwd(*args, **kwargs)  # already in the forward's autocast state
        with autocast(device_type=device_type, enabled=want, dtype=ctx._dtype):
            return bwd(*args, **kwargs)

    return decorate_bwd
