"""pytest configuration: markers + import paths.

`-m "not gpu"`: oracle vs golden vectors / known answers, host logic, C-ABI symbol check, gloo DP.
`-m gpu`      : parity tests proper -- the HIP path (through the C ABI) against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nerf-texture_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(autouse=True)
def _collect_dead_cycles_before_a_gpu_test(request):
    """A dropped trainer or graphed renderer is a reference cycle that owns HIP graphs and pool memory; Python frees it whenever its collector next
    runs -- possibly inside a LATER test's graph recording, where it aborts the process (ngp_harness/streams.py capture_section is the product's
    guard; this keeps one test's garbage out of the next test)."""
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        gc.collect()
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.lib()
    return orc


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def knobs():
    """Set kernel A/B switches for one test (nerftex_tune_set); every knob touched is restored afterwards."""
    import nerftex_hip

    stack = []

    def set_knobs(**kw):
        stack.append(nerftex_hip.tune(**kw))

    yield set_knobs
    for t in reversed(stack):
        t.__exit__(None, None, None)
