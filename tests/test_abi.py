"""C-ABI surface (CPU only): the library loads and exports every symbol include/nerftex_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "nerftex_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nerftex_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    import nerftex_hip

    names = _declared()
    assert len(names) >= 25, names
    lib = ctypes.CDLL(nerftex_hip.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in include/nerftex_hip.h but not exported: {missing}"
    # the python binding table covers the header one to one
    assert sorted(nerftex_hip.EXPORTS) == names


def test_version_and_error_channel():
    import nerftex_hip

    assert nerftex_hip.lib.nerftex_version().decode().endswith("gfx950")
    assert nerftex_hip.lib.nerftex_last_error() is not None


def test_reference_api_surface():
    """Names / signatures the reference's callers use; no GPU work."""
    import inspect

    import ffmlp
    import gridencoder
    import raymarching
    import shencoder
    import RayTracer

    for fn in ("near_far_from_aabb", "polar_from_ray", "morton3D", "morton3D_invert", "packbits", "march_rays_train",
               "march_rays_train_differentiable", "composite_rays_train", "march_rays", "composite_rays", "compact_rays"):
        assert callable(getattr(raymarching, fn)), fn
    sig = inspect.signature(raymarching.raymarching._march_rays_train.forward)
    assert list(sig.parameters)[1:] == ["rays_o", "rays_d", "bound", "density_bitfield", "C", "H", "nears", "fars", "step_counter",
                                        "mean_count", "perturb", "align", "force_all_rays", "dt_gamma", "max_steps"]
    sig = inspect.signature(raymarching.raymarching._march_rays.forward)
    assert list(sig.parameters)[1:] == ["n_alive", "n_step", "rays_alive", "rays_t", "rays_o", "rays_d", "bound", "density_bitfield", "C",
                                        "H", "near", "far", "align", "perturb", "dt_gamma", "max_steps"]
    enc = gridencoder.GridEncoder(desired_resolution=4096)
    assert enc.output_dim == 32 and enc.embeddings.shape == (6328848, 2) and enc.offsets.dtype.is_floating_point is False
    assert list(inspect.signature(gridencoder.GridEncoder.__init__).parameters)[1:] == [
        "input_dim", "num_levels", "level_dim", "per_level_scale", "base_resolution", "log2_hashmap_size", "desired_resolution", "gridtype",
        "align_corners"]
    c = gridencoder.GridEncoder_clustering(num_levels=2, log2_hashmap_size=8)
    assert len(c.cluster_layers) == 2 and float(c.clustering_loss(pick_level=False)) == float(c.clustering_loss(pick_level=False))
    assert shencoder.SHEncoder(degree=4).output_dim == 16
    m = ffmlp.FFMLP(32, 16, 64, 2)
    assert m.num_parameters == 7168 and ffmlp.FFMLP(32, 3, 64, 3).num_parameters == 11264
    assert hasattr(RayTracer.RayTracer, "trace")


def test_ffmlp_init_matches_reference_golden():
    import json

    import ffmlp

    rows = json.load(open(os.path.join(ROOT, "tests", "golden", "ffmlp_params.json")))
    for r in rows:
        m = ffmlp.FFMLP(**r["kwargs"])
        assert m.num_parameters == r["num_parameters"] and m.padded_output_dim == r["padded_output_dim"]
        w = m.weights.detach()
        assert [float(v) for v in w[:8].tolist()] == r["first8"], "same init stream as the reference (manual_seed(42) + uniform_)"


def test_grid_level_table_matches_reference_golden():
    import json

    from gridencoder.grid import level_table

    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "grid_offsets.json"))):
        kw = c["kwargs"]
        off, total = level_table(kw.get("input_dim", 3), kw.get("num_levels", 16), c["per_level_scale"], kw.get("base_resolution", 16),
                                 kw.get("log2_hashmap_size", 19), kw.get("align_corners", False))
        assert off.tolist() == c["offsets"] and total == c["rows"], c["name"]


def test_header_compiles_as_c_and_its_constants_match_the_python_binding(tmp_path):
    """include/nerftex_hip.h is a C header (the boundary is a C ABI): gcc -std=c99 -Wall -Werror -pedantic takes it, and the constants a binding
    has to restate -- NERFTEX_ROWS_AUTO, the dtype / layout / error codes -- have the values nerftex_hip (the ctypes binding) uses."""
    import re
    import subprocess

    import nerftex_hip

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "nerftex_hip.h")).read()
    names = sorted(set(re.findall(r"#define\s+(NERFTEX_[A-Z0-9_]+)\s+[-0-9(]", header)))
    assert "NERFTEX_OK" in names and len(names) >= 6, names
    src = tmp_path / "consts.c"
    lines = ['#include <stdio.h>', '#include "nerftex_hip.h"', "int main(void) {"]
    lines += [f'    printf("{n} %ld\\n", (long)({n}));' for n in names]
    lines += ['    printf("ROWS_AUTO_A %lu\\n", (unsigned long)NERFTEX_ROWS_AUTO(640000, 4));',
              '    printf("ROWS_AUTO_B %lu\\n", (unsigned long)NERFTEX_ROWS_AUTO(16777215, 127));', "    return 0;", "}"]
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "consts"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    assert int(out["ROWS_AUTO_A"]) == nerftex_hip.rows_auto(640000, 4) and int(out["ROWS_AUTO_B"]) == nerftex_hip.rows_auto(16777215, 127)
    checked = 0
    for n in names:
        py = getattr(nerftex_hip, n[len("NERFTEX_"):], None)
        if isinstance(py, int):
            assert int(out[n]) == py, (n, out[n], py)
            checked += 1
    assert checked >= 4, (checked, names)  # F16 / F32 / LAYOUT_* / ... restated by the binding
