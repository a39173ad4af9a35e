"""The multi-process training step of bench.py (two replayed graphs per step, eager all-reduce of the fp16 gradients in between,
mean_count agreement) run end to end with TWO ranks on the ONE GPU of the test box: NERFTEX_DP_SHARE_GPU=1 puts both ranks on cuda:0
and exchanges through gloo (RCCL refuses two ranks on one device).  What it pins: the ranks finish with bit-identical replicas."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("extra", [[], ["--no-fused-opt"]], ids=["half-leaf-adam", "torch-adam"])
def test_two_ranks_on_one_gpu_keep_identical_replicas(extra):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(os.environ, NERFTEX_DP_SHARE_GPU="1")
    port = 29600 + os.getpid() % 200 + (7 if extra else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "4", "--rays", "2048", "--no-cpu-baseline",
           "--no-other", "--no-infer", "--no-kernel-timing"] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["config"]["replicas_identical_after_run"] is True
    assert res["value"] > 0
