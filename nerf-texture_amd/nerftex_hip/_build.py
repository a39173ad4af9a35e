"""Compile every HIP source for gfx950 into lib/libnerftex_hip.so.  Importable without the library present."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(PKG_ROOT, "csrc")
# NERFTEX_HIP_LIB: another build of the same library (A/B experiments on compiler flags); the default is the in-tree build
LIB_PATH = os.environ.get("NERFTEX_HIP_LIB") or os.path.join(PKG_ROOT, "lib", "libnerftex_hip.so")


def build(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU; `make` only rebuilds what changed."""
    if shutil.which("hipcc") is None:
        raise RuntimeError("hipcc not found: cannot build libnerftex_hip.so")
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH
