#!/usr/bin/env python3
"""Child of tools/make_golden.py's checkpoint cross-check: load a `.pth` the REFERENCE's classes wrote into the drop-in harness
(ngp_harness.checkpoint), verify every tensor, and write it back in the reference's format for the parent to load strictly.

    python tools/ckpt_roundtrip_child.py <reference.pth> <out.pth>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nerf-texture_amd"))

import torch  # noqa: E402

from ngp_harness import checkpoint  # noqa: E402
from ngp_harness.model import NGPField, Renderer  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
ref = torch.load(src, weights_only=False)
bound = float(ref["model"]["aabb_train"][3])
field = NGPField(bound=bound, mlp="ffmlp")
renderer = Renderer(field, bound=bound)
ckpt = checkpoint.load_checkpoint(src, renderer, model_only=True)
mine = checkpoint.model_state(renderer)
assert sorted(mine) == sorted(ref["model"]), (sorted(mine), sorted(ref["model"]))
for k, v in ref["model"].items():
    assert mine[k].dtype == v.dtype and mine[k].shape == v.shape and torch.equal(mine[k], v), k
assert renderer.mean_count == ref["mean_count"] and renderer.mean_density == ref["mean_density"]
checkpoint.save_checkpoint(dst, renderer, epoch=ckpt["epoch"], global_step=ckpt["global_step"], stats=ckpt["stats"])
print(json.dumps({"loaded_keys": sorted(mine), "mean_count": renderer.mean_count}))
