#!/bin/bash
# The GPU parity tests (oracle comparisons, bit-exact or at their stated tolerances) run WHILE another process trains on the same GPU: any kernel
# whose result depends on co-scheduling shows up as a failing parity test.  (tests/test_gpu_streams.py is left out: it asserts on hand-over LATENCIES
# measured in child processes, which a neighbour that keeps the GPU busy moves -- [32.9, 174.3] us where an idle GPU gives [31, 31]; timing, not parity.)  Usage: bash tools/gpu_soak_beside_neighbour.sh [pytest args]
mkdir -p gpurun_out/soak
rm -f /tmp/ready
READY_FILE=/tmp/ready STEPS=400000 python tools/determinism_probe.py neighbour > /dev/null 2>&1 &
nb=$!
while [ ! -f /tmp/ready ]; do sleep 0.2; done
python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_gpu_dp_shared_gpu.py --deselect tests/test_gpu_streams.py "$@" 2>&1 | grep -v amdgpu | tail -40 > gpurun_out/soak/pytest_tail.txt
alive=$(kill -0 $nb 2>/dev/null && echo yes || echo no)
echo "neighbour still training when the tests ended: $alive" >> gpurun_out/soak/pytest_tail.txt
kill $nb 2>/dev/null; wait $nb 2>/dev/null
tail -25 gpurun_out/soak/pytest_tail.txt
