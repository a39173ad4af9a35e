#!/bin/bash
rm -f /tmp/ready
READY_FILE=/tmp/ready STEPS=400000 python tools/determinism_probe.py neighbour > /dev/null 2>&1 &
nb=$!
while [ ! -f /tmp/ready ]; do sleep 0.2; done
for lib in "" nerf-texture_amd/lib/ab/libnerftex_hip_gridnoslp.so; do
echo "== library: ${lib:-in-tree}"
for i in 1 2 3 4 5 6 7 8; do
NERFTEX_HIP_LIB=$lib python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "tight_per_row or folded_normalisation" 2>&1 | grep -E "entries off|passed|failed" | cut -c1-200
done
done
kill $nb 2>/dev/null; wait $nb 2>/dev/null
