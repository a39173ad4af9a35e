#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
What can be executed / evaluated there without nvcc:

  sh_golden.npz       the 64 SH basis polynomials and their 192 partial derivatives, evaluated from
                      the reference's own expression text (shencoder/src/shencoder.cu:51-351) in
                      float64 on seeded inputs.  The .cu is read at run time and its right-hand
                      sides are evaluated with numpy; no reference text is stored in this repo.
  grid_offsets.json   GridEncoder.__init__ level tables (gridencoder/grid.py:93-131), obtained by
                      importing the reference class with a stubbed `_gridencoder` backend.
  ffmlp_params.json   FFMLP.__init__ parameter counts (ffmlp/ffmlp.py:99-144), same mechanism.

Import discipline (SURVEY.md incident): the reference wrappers fall back to a JIT build that
hipifies sources INTO /root/reference unless the `_X` backend modules are pre-seeded, so every
`_gridencoder/_raymarching/_shencoder/_ffmlp` (+ `turtle`) is stubbed and bytecode writing is off.
"""
import json
import os
import re
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np

REF = os.environ.get("NERFTEX_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def sh_golden():
    src = open(os.path.join(REF, "shencoder/src/shencoder.cu")).read()
    # assignments of the form  <lhs>[<idx>] = <expr> ;   with lhs in outputs/dx/dy/dz
    pat = re.compile(r"^\s*(outputs|dx|dy|dz)\[(\d+)\]\s*=\s*(.*?)\s*;", re.M)
    exprs = {"outputs": {}, "dx": {}, "dy": {}, "dz": {}}
    for lhs, idx, rhs in pat.findall(src):
        rhs = re.sub(r"(\d+\.\d*(?:[eE][-+]?\d+)?|\d+)f\b", r"\1", rhs)  # strip float suffix
        rhs = rhs.replace("pow(", "np.power(")
        exprs[lhs][int(idx)] = rhs
    for k in exprs:
        assert sorted(exprs[k]) == list(range(64)), (k, len(exprs[k]))

    rng = np.random.default_rng(20260926)
    B = 96
    pts = rng.uniform(-1.0, 1.0, size=(B, 3))
    unit = rng.normal(size=(B // 2, 3))
    unit /= np.linalg.norm(unit, axis=1, keepdims=True)
    pts[: B // 2] = unit  # half on the unit sphere, half raw points in the cube
    pts = pts.astype(np.float32).astype(np.float64)
    x, y, z = pts[:, 0], pts[:, 1], pts[:, 2]
    env = dict(np=np, x=x, y=y, z=z, xy=x * y, xz=x * z, yz=y * z, x2=x * x, y2=y * y, z2=z * z)
    env["xyz"] = env["xy"] * z
    for v in "xyz":
        env[v + "4"] = env[v + "2"] ** 2
        env[v + "6"] = env[v + "4"] * env[v + "2"]

    def ev(table):
        out = np.zeros((B, 64))
        for i in range(64):
            out[:, i] = eval(table[i], {"__builtins__": {}}, env) + np.zeros(B)
        return out

    np.savez_compressed(
        os.path.join(OUT, "sh_golden.npz"),
        inputs=pts.astype(np.float32),
        outputs=ev(exprs["outputs"]),
        dx=ev(exprs["dx"]),
        dy=ev(exprs["dy"]),
        dz=ev(exprs["dz"]),
    )
    print("sh_golden.npz: 64 basis + 3x64 derivative polynomials on", B, "points")


def module_goldens():
    import torch

    sys.path.insert(0, REF)
    _stub("_gridencoder")
    _stub("_shencoder")
    _stub("_raymarching")
    _stub("_ffmlp", allocate_splitk=lambda n: None, free_splitk=lambda: None)
    _stub("turtle", backward=None, forward=None, bgcolor=None)

    from gridencoder.grid import GridEncoder  # noqa: E402

    cases = [
        dict(name="fox_bound2", kw=dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096)),
        # tools/encoding.py:45 get_encoder defaults align_corners=True: what network_ff.py / network.py actually build
        dict(name="fox_bound2_align", kw=dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=4096, align_corners=True)),
        dict(name="bound1_align", kw=dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048, align_corners=True)),
        dict(name="bound1", kw=dict(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048)),
        dict(name="curved_L8", kw=dict(input_dim=3, num_levels=8, level_dim=2, base_resolution=512, log2_hashmap_size=19, desired_resolution=1024, align_corners=True)),
        dict(name="normal_L4", kw=dict(input_dim=3, num_levels=4, level_dim=2, base_resolution=16, log2_hashmap_size=19, per_level_scale=2 ** (1 / 3))),
        dict(name="default", kw=dict()),
        dict(name="2d_tiled", kw=dict(input_dim=2, num_levels=8, level_dim=4, base_resolution=16, log2_hashmap_size=12, per_level_scale=2, gridtype="tiled")),
        dict(name="small_C1", kw=dict(input_dim=3, num_levels=6, level_dim=1, base_resolution=4, log2_hashmap_size=10, per_level_scale=1.5, align_corners=True)),
    ]
    out = []
    for c in cases:
        enc = GridEncoder(**c["kw"])
        out.append(
            dict(
                name=c["name"],
                kwargs=c["kw"],
                per_level_scale=float(enc.per_level_scale),
                offsets=[int(v) for v in enc.offsets.tolist()],
                rows=int(enc.embeddings.shape[0]),
                output_dim=int(enc.output_dim),
            )
        )
    json.dump(out, open(os.path.join(OUT, "grid_offsets.json"), "w"), indent=1)
    print("grid_offsets.json:", [c["name"] for c in out])

    from ffmlp.ffmlp import FFMLP  # noqa: E402

    rows = []
    for kw in [
        dict(input_dim=32, output_dim=16, hidden_dim=64, num_layers=2),
        dict(input_dim=32, output_dim=3, hidden_dim=64, num_layers=3),
        dict(input_dim=16, output_dim=1, hidden_dim=32, num_layers=2),
        dict(input_dim=64, output_dim=16, hidden_dim=64, num_layers=4),
        dict(input_dim=48, output_dim=8, hidden_dim=128, num_layers=2),
    ]:
        m = FFMLP(**kw)
        w = m.weights.detach()
        rows.append(
            dict(
                kwargs=kw,
                num_parameters=int(m.num_parameters),
                padded_output_dim=int(m.padded_output_dim),
                init_bound=float(w.abs().max()),
                first8=[float(v) for v in w[:8].tolist()],
            )
        )
    json.dump(rows, open(os.path.join(OUT, "ffmlp_params.json"), "w"), indent=1)
    print("ffmlp_params.json:", len(rows), "modules")

    # checkpoint wire format: parameter / buffer names, shapes and dtypes of the --ff network
    # (nerf/network_ff.py:11-56 = NeRFRenderer buffers + encoder + two FFMLPs).  The network class itself cannot be imported here
    # (nerf/utils.py needs a dozen absent packages), so the entries come from the component classes it is built from, instantiated
    # exactly as network_ff.py / tools/encoding.py:45 do, plus the register_buffer names read from nerf/renderer.py.
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19, desired_resolution=2048 * 2,
                      align_corners=True)
    sig = FFMLP(input_dim=32, output_dim=16, hidden_dim=64, num_layers=2)
    col = FFMLP(input_dim=32, output_dim=3, hidden_dim=64, num_layers=3)
    entries = {}
    for prefix, mod in (("encoder", enc), ("sigma_net", sig), ("color_net", col)):
        for k, v in mod.state_dict().items():
            entries[f"{prefix}.{k}"] = [list(v.shape), str(v.dtype).replace("torch.", "")]
    src = open(os.path.join(REF, "nerf/renderer.py")).read()
    buffers = sorted(set(re.findall(r"register_buffer\(\s*['\"](\w+)['\"]", src)))
    json.dump(dict(bound=2, cascade=2, grid_size=128, entries=entries, renderer_buffers=buffers),
              open(os.path.join(OUT, "checkpoint_keys.json"), "w"), indent=1)
    print("checkpoint_keys.json:", sorted(entries), buffers)

    # guard: nothing may have been written into the reference tree
    import subprocess

    new = subprocess.run(
        ["find", REF, "-newer", os.path.join(OUT, "..", "..", "BASELINE.json"), "-type", "f"], capture_output=True, text=True
    ).stdout.strip()
    assert new == "", "files appeared under the reference tree:\n" + new


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    sh_golden()
    module_goldens()
