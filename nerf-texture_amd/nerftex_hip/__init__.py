"""ctypes binding of libnerftex_hip.so (C ABI: include/nerftex_hip.h).

This is plumbing: device memory, streams and autograd come from PyTorch-ROCm, every kernel comes from
the hand-written HIP library.  There is NO fallback: if the library is missing or a GPU op is invoked
without it, importing / calling fails loudly (a CPU or eager-PyTorch substitute would void parity).

    from nerftex_hip import lib, ptr, stream, check
    check(lib.nerftex_packbits(ptr(grid), N, thresh, ptr(bitfield), stream()))
"""
import ctypes as C
import os

from ._build import CSRC, LIB_PATH, PKG_ROOT  # noqa: F401

F32, F16 = 0, 1
LAYOUT_LBC, LAYOUT_BLC = 0, 1
LAYOUT_GRAD_OVERWRITE = 0x100  # backward: grad_embeddings is uninitialised and gets overwritten


def rows_auto(n_rays, slots_per_ray):
    """NERFTEX_ROWS_AUTO(N, F): the n_step / rows_per_unit code that makes the kernels derive n_step from the alive count on the device."""
    assert 0 < n_rays < (1 << 24) and 0 < slots_per_ray <= 127
    return 0x80000000 | (int(slots_per_ray) << 24) | int(n_rays)


from ._build import build  # noqa: E402,F401


_u32, _f32, _i, _vp, _sz = C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_size_t
_u64, _f64 = C.c_uint64, C.c_double

# name -> argument ctypes, in the order of include/nerftex_hip.h
_SIGNATURES = {
    "nerftex_field_forward": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_amp": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_grid_encode_backward_amp": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _i, _vp, _vp, _u32, _i, _i, _i, _f32, _f32, _vp, _vp],
    "nerftex_field_density": [_vp, _vp, _u32, _vp, _vp],
    "nerftex_grid_encode_backward_phase": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i, _i, _i, _f32, _f32, _i, _u32, _u32, _vp],
    "nerftex_grid_encode_backward_phase_amp": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i, _i, _i, _f32, _f32, _i, _u32, _u32, _vp, _vp],
    "nerftex_release_workspaces": [],
    "nerftex_workspace_capture_set": [_i],
    "nerftex_field_forward_rows": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _vp],
    "nerftex_grid_encode_forward_rows": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i, _i, _i, _f32, _f32, _vp, _u32, _vp],
    "nerftex_field_mid_forward": [_vp, _vp, _u32, _vp, _vp, _vp],
    "nerftex_curved_pack_inputs": [_vp, _vp, _u32, _vp, _vp],
    "nerftex_curved_mid_forward": [_vp, _vp, _vp, _u32, _f32, _i, _vp, _vp, _vp],
    "nerftex_curved_out_forward": [_vp, _u32, _vp, _vp, _u32, _vp, _vp, _vp],
    "nerftex_field_mid_backward": [_vp, _vp, _vp, _u32, _vp, _vp],
    "nerftex_field_out_forward": [_vp, _u32, _vp, _vp],
    "nerftex_field_out_backward": [_vp, _vp, _u32, _vp, _vp],
    "nerftex_render_tail_forward": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_render_tail_backward": [_vp, _vp, _f32, _vp, _vp, _f32, _u32, _vp, _vp, _vp],
    "nerftex_composite_tail_backward": [_vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "nerftex_adam_half_step": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _f64, _f64, _f64, _f64, _vp, _vp, _vp],
    "nerftex_amp_check_half": [_i, _vp, _vp, _vp, _vp],
    "nerftex_adam_half_step_amp": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _f64, _f64, _i, _vp],
    "nerftex_amp_update": [_vp, _vp, _vp, _vp, _f64, _f64, _i, _vp],
    "nerftex_adam_mixed_step": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _f32, _f64, _f64, _f64, _f64, _vp, _vp, _vp],
    "nerftex_adam_mixed_step_amp": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _f64, _f64, _i, _vp],
    "nerftex_amp_check_mixed": [_i, _vp, _vp, _u32, _vp, _vp],
    "nerftex_adam_mixed_step_amp_db": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _f64, _f64, _i,
                                       _vp, _vp, _vp, _vp, _u64, _vp],
    "nerftex_field_backward_live": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_live_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_live_consume": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_live_consume_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_live_deferred": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_backward_live_deferred_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_step_trailer_run": [_vp, _vp],
    "nerftex_grid_encode_backward_adam_trailer": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i, _i, _i, _f32, _f32, _vp, _vp, _vp, _vp],
    "nerftex_composite_step": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_render_tail_forward_live": [_vp, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_composite_tail_backward_live": [_vp, _vp, _f32, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "nerftex_grid_encode_backward_adam": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _i, _i, _i, _f32, _f32, _vp, _vp, _vp],
    "nerftex_field_forward_bf16": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_field_density_bf16": [_vp, _vp, _u32, _vp, _vp],
    "nerftex_field_forward_rows_bf16": [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _u32, _vp],
    "nerftex_field_backward_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_table_adam_step": [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _f64, _f64, _f64, _f64, _vp, _vp, _vp],
    "nerftex_tune_set": [C.c_char_p, C.c_long],
    "nerftex_profile_enable": [_i],
    "nerftex_profile_reset": [],
    "nerftex_profile_report": [C.c_char_p, _sz],
    "nerftex_grid_encode_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _i, _vp, _u32, _i, _i, _i, _vp],
    "nerftex_grid_encode_backward": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _i, _vp, _vp, _u32, _i, _i, _i, _vp],
    "nerftex_grid_register_offsets": [_vp, _u32, _vp],
    "nerftex_deferred_error": [],
    "nerftex_grid_encode_forward_affine": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _i, _vp, _u32, _i, _i, _i, _f32, _f32, _vp],
    "nerftex_grid_encode_backward_affine": [_vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _f32, _u32, _i, _vp, _vp, _u32, _i, _i, _i, _f32, _f32, _vp],
    "nerftex_sh_encode_forward": [_vp, _vp, _u32, _u32, _u32, _i, _vp, _vp],
    "nerftex_sh_encode_backward": [_vp, _vp, _u32, _u32, _u32, _vp, _vp, _vp],
    "nerftex_near_far_from_aabb": [_vp, _vp, _vp, _u32, _f32, _vp, _vp, _vp],
    "nerftex_polar_from_ray": [_vp, _vp, _f32, _u32, _vp, _vp],
    "nerftex_morton3D": [_vp, _u32, _vp, _vp],
    "nerftex_morton3D_invert": [_vp, _u32, _vp, _vp],
    "nerftex_packbits": [_vp, _u32, _f32, _vp, _vp],
    "nerftex_march_rays_train": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_march_rays_train_fresh": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_march_rays_train_differentiable": [_vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_composite_rays_train_forward": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "nerftex_composite_rays_train_backward": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "nerftex_march_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_composite_rays": [_u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_compact_rays": [_u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_march_rays_dev": [_u32, _vp, _u32, _vp, _vp, _vp, _vp, _f32, _f32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "nerftex_composite_rays_dev": [_u32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_compact_rays_dev": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_compact_rays_budget_dev": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp],
    "nerftex_compact_rays_budget_mirror_dev": [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp],
    "nerftex_occupancy_sample_full": [_vp, _u32, _u32, _f32, _vp, _u64, _vp],
    "nerftex_occupancy_sample_partial": [_vp, _u32, _u32, _f32, _u32, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp],
    "nerftex_occupancy_sample_partial_ordered": [_vp, _u32, _u32, _f32, _u32, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _i, _vp],
    "nerftex_occupancy_update": [_vp, _vp, _vp, _u32, _u32, _u32, _f32, _i, _f32, _vp, _vp, _vp],
    "nerftex_ffmlp_forward": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "nerftex_ffmlp_inference": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "nerftex_ffmlp_backward": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i, _vp, _vp, _vp, _vp],
    "nerftex_ffmlp_forward_bf16": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "nerftex_ffmlp_inference_bf16": [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp],
    "nerftex_ffmlp_backward_bf16": [_vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _i, _vp, _vp, _vp, _vp],
    "nerftex_ffmlp_allocate_splitk": [_sz],
    "nerftex_ffmlp_free_splitk": [],
    "nerftex_create_raytracer": [_vp, _u32, _vp, _u32, C.POINTER(_vp)],
    "nerftex_destroy_raytracer": [_vp],
    "nerftex_curved_project": [_vp, _vp, _vp, _vp, _u32, _u32, _vp, _vp, _u32, _f32, _f32, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "nerftex_knn_create": [_vp, _u32, C.POINTER(_vp)],
    "nerftex_knn_destroy": [_vp],
    "nerftex_knn_query": [_vp, _vp, _u32, _u32, _vp, _vp, _vp],
    "nerftex_debug_workspace": [C.c_int, _vp, C.POINTER(_vp), C.POINTER(C.c_size_t)],
    "nerftex_raytracer_trace": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
}
class TableAdam(C.Structure):
    """nerftex_table_adam of include/nerftex_hip.h, field for field."""
    _fields_ = [("param", _vp * 2), ("exp_avg", _vp * 2), ("exp_avg_sq", _vp * 2), ("param_half", _vp), ("live", _vp), ("step", _vp),
                ("grad_scale", _vp), ("found_inf", _vp), ("lr", _f64), ("beta1", _f64), ("beta2", _f64), ("eps", _f64)]


class StepTrailer(C.Structure):
    """nerftex_step_trailer of include/nerftex_hip.h: opaque, filled by nerftex_field_backward_live_deferred."""
    _fields_ = [("opaque", C.c_uint64 * 16)]


class StepLoss(C.Structure):
    """nerftex_step_loss of include/nerftex_hip.h, field for field."""
    _fields_ = [("err", _vp), ("n_rays", _u32), ("loss_mul", _f32), ("scale", _vp), ("loss", _vp), ("scaled_loss", _vp)]


EXPORTS = ["nerftex_last_error", "nerftex_version", "nerftex_tune_get", "nerftex_workspace_slots_touched"] + list(_SIGNATURES)


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU / eager fallback."
        )
    lib = C.CDLL(LIB_PATH)
    lib.nerftex_last_error.restype = C.c_char_p
    lib.nerftex_version.restype = C.c_char_p
    lib.nerftex_tune_get.argtypes = [C.c_char_p]
    lib.nerftex_tune_get.restype = C.c_long
    lib.nerftex_workspace_slots_touched.argtypes = []
    lib.nerftex_workspace_slots_touched.restype = C.c_uint
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == a symbol the header declares is not exported
        fn.argtypes = args
        fn.restype = C.c_int
    return lib


lib = _load()


def check(rc):
    """Raise RuntimeError with the library's message (mirrors the reference's TORCH_CHECK / runtime_error texts)."""
    if rc != 0:
        msg = lib.nerftex_last_error().decode() or f"nerftex_hip call failed with status {rc}"
        raise RuntimeError(msg)


class tune:
    """Set tuning knobs / A-B switches of the kernels (include/nerftex_hip.h: nerftex_tune_set); usable as a context manager:
    `with tune(grid_bwd=1): ...` restores the previous values on exit."""

    def __init__(self, **knobs):
        self._old = {k: lib.nerftex_tune_get(k.encode()) for k in knobs}
        for k, v in knobs.items():
            check(lib.nerftex_tune_set(k.encode(), int(v)))

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for k, v in self._old.items():
            check(lib.nerftex_tune_set(k.encode(), v))
        return False


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    """torch's current HIP stream as the void* the C ABI expects (the raw handle: `torch.cuda.current_stream()` builds a Stream object,
    ~11 us a call, seven calls per eager training step)."""
    import torch

    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def require_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


# ---- optional per-op device timing (bench.py turns it on to measure the roofline kernel live) ----
class _Timer:
    def __init__(self):
        self.enabled = False
        self.only = None  # optional set of op names to time (keeps the event overhead off everything else)
        self.records = {}

    def start(self, name):
        if not self.enabled or (self.only is not None and name not in self.only):
            return None
        import torch

        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        return (name, a, b)

    def stop(self, tok):
        if tok is None:
            return
        name, a, b = tok
        b.record()
        self.records.setdefault(name, []).append((a, b))

    def summary(self, reset=True):
        import torch

        torch.cuda.synchronize()
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in self.records.items()}
        if reset:
            self.records = {}
        return out


timer = _Timer()


def kernel_profile(enable=None, reset=False):
    """Per-kernel device timing of the library itself (hipEvent pairs on the launch stream).  kernel_profile(True) starts,
    kernel_profile() returns {"kernel": {"calls", "avg_us", "total_us"}} so far, kernel_profile(False) stops."""
    import json

    if reset:
        check(lib.nerftex_profile_reset())
    if enable is not None:
        check(lib.nerftex_profile_enable(int(enable)))  # 0 off, 1 / True every kernel, 2 hash-grid kernels only
        return None
    buf = C.create_string_buffer(1 << 16)
    check(lib.nerftex_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode() or "{}")
