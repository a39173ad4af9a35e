"""Build-level pins that need no GPU."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-texture_amd", "csrc")


@pytest.mark.parametrize("source, kernel", [("gridencoder_binned", "bin_fill_dir_kernel"), ("gridencoder", "grid_forward_level_kernel")])
def test_hash_grid_kernels_are_compiled_without_packed_fp32(source, kernel):
    """csrc/Makefile: the hash-grid sources must not contain v_pk_*_f32 instructions (round 4: with them the backward's record builder and the
    gather's input-gradient branch are not reproducible when other kernels share the GPU -- tests/test_gpu_dp_shared_gpu.py has the GPU side).  Compiles the file to assembly with the Makefile's own
    command line and looks."""
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    dry = subprocess.run(["make", "-C", CSRC, "-n", "-W", f"{source}.hip", f"../lib/obj/{source}.o"], capture_output=True, text=True, check=True).stdout
    cmd = [ln for ln in dry.splitlines() if f" {source}.hip" in ln and " -c " in ln][-1].split()
    assert "-fno-slp-vectorize" in cmd and "-disable-vector-combine" in cmd, cmd
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "binned.s")
        i = cmd.index("-c")
        cmd[i:i + 1] = ["--cuda-device-only", "-S"]
        cmd[cmd.index("-o") + 1] = asm
        subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True)
        text = open(asm).read()
    assert kernel in text
    assert not re.search(r"\bv_pk_[a-z]+_f32\b", text)
