"""Checkpoint wire format of the reference trainer (nerf/utils.py:1485-1602), for the --ff network served by this hot path.

A reference `.pth` is `{'epoch', 'global_step', 'stats', 'mean_count', 'mean_density', 'model': state_dict [, 'optimizer', 'lr_scheduler',
'scaler', 'ema']}`; its network IS its renderer (class NeRFNetwork(NeRFRenderer)), so 'model' holds the renderer's buffers
(aabb_train, aabb_infer, density_grid, density_bitfield, step_counter: nerf/renderer.py:107-122) and the field's tensors
(encoder.embeddings, encoder.offsets, sigma_net.weights, color_net.weights) under flat names.  Here the two are separate objects
(`ngp_harness.model.Renderer` holds an `NGPField`): `model_state` / `load_model_state` translate.
"""
import torch

RENDERER_BUFFERS = ("aabb_train", "aabb_infer", "density_grid", "density_bitfield", "step_counter")


def model_state(renderer):
    """The reference's `model.state_dict()` for this (renderer, field) pair: same names, shapes and dtypes."""
    out = {k: getattr(renderer, k).detach().clone() for k in RENDERER_BUFFERS}
    out.update({k: v.detach().clone() for k, v in renderer.field.state_dict().items()})
    return out


def save_checkpoint(path, renderer, epoch=0, global_step=0, stats=None, optimizer=None, scaler=None):
    """Write what Trainer.save_checkpoint(full=optimizer is not None) writes (nerf/utils.py:1485-1523)."""
    state = {"epoch": epoch, "global_step": global_step, "stats": stats or {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None},
             # plain Python numbers, as nerf/utils.py:1496-1497 stores them (the device-side occupancy update keeps mean_density as a view of
             # a buffer every later update overwrites: never pickle that)
             "mean_count": int(renderer.mean_count), "mean_density": float(renderer.mean_density), "model": model_state(renderer)}
    if optimizer is not None:
        state["optimizer"] = optimizer.state_dict()
    if scaler is not None:
        state["scaler"] = scaler.state_dict()
    torch.save(state, path)
    return state


def load_model_state(renderer, model_sd, strict=True):
    """`model.load_state_dict(checkpoint['model'], strict=...)` (nerf/utils.py:1561) for the split objects; returns (missing, unexpected).
    Optimizers that keep their own copy of the parameters (ngp_harness.optim.HalfLeafAdam) must be told: call their resync()."""
    sd = dict(model_sd)
    missing, unexpected = [], []
    for k in RENDERER_BUFFERS:
        if k in sd:
            buf = getattr(renderer, k)
            v = sd.pop(k).to(buf.device)
            if v.shape != buf.shape or v.dtype != buf.dtype:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} {v.dtype}, model {tuple(buf.shape)} {buf.dtype}")
            buf.copy_(v)
        else:
            missing.append(k)
    res = renderer.field.load_state_dict(sd, strict=False)
    missing += list(res.missing_keys)
    unexpected += list(res.unexpected_keys)
    if strict and (missing or unexpected):
        raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
    return missing, unexpected


def load_checkpoint(path, renderer, optimizer=None, scaler=None, model_only=False, map_location=None):
    """Trainer.load_checkpoint (nerf/utils.py:1537-1602): model (non-strict, like the reference), mean_count / mean_density, then -- unless
    model_only -- optimizer and scaler state.  Returns the checkpoint dict (epoch, global_step, stats are the caller's)."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    if "model" not in ckpt:
        load_model_state(renderer, ckpt, strict=True)
        if optimizer is not None and hasattr(optimizer, "resync"):
            optimizer.resync()  # the masters changed under it
        return ckpt
    load_model_state(renderer, ckpt["model"], strict=False)
    if "mean_count" in ckpt:
        renderer.mean_count = int(ckpt["mean_count"])
    if "mean_density" in ckpt:
        renderer.mean_density = float(ckpt["mean_density"])
    if optimizer is not None and hasattr(optimizer, "resync"):
        optimizer.resync()  # the masters changed under it
    if model_only:
        return ckpt
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if scaler is not None and "scaler" in ckpt:
        scaler.load_state_dict(ckpt["scaler"])
    return ckpt
