"""Build-level pins that need no GPU."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nerf-texture_amd", "csrc")


def test_hash_grid_backward_is_compiled_without_packed_fp32():
    """csrc/Makefile: gridencoder_binned.hip must not contain v_pk_*_f32 instructions (round 4: with them the record builder is not reproducible
    when other kernels share the GPU -- tests/test_gpu_dp_shared_gpu.py has the GPU side).  Compiles the file to assembly with the Makefile's own
    command line and looks."""
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    dry = subprocess.run(["make", "-C", CSRC, "-n", "-W", "gridencoder_binned.hip", "../lib/obj/gridencoder_binned.o"], capture_output=True, text=True, check=True).stdout
    cmd = [ln for ln in dry.splitlines() if "gridencoder_binned.hip" in ln and " -c " in ln][-1].split()
    assert "-fno-slp-vectorize" in cmd, cmd
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "binned.s")
        i = cmd.index("-c")
        cmd[i:i + 1] = ["--cuda-device-only", "-S"]
        cmd[cmd.index("-o") + 1] = asm
        subprocess.run(cmd, cwd=CSRC, check=True, capture_output=True)
        text = open(asm).read()
    assert "bin_fill_dir_kernel" in text
    assert not re.search(r"\bv_pk_[a-z]+_f32\b", text)
