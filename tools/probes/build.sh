#!/bin/bash
# Build the stand-alone GPU probes into tools/probes/_bin (git-ignored; travels to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.
# k3d_reduce is built TWICE: with the compiler's default flags (packed fp32 ON: reproduces the round-4 fault beside an MFMA neighbour) and with the
# packed-fp32 target feature off (the control = what csrc/Makefile ships).
set -e
cd "$(dirname "$0")"
mkdir -p _bin
F="-O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics"
build() {  # name extra-flags...
  local n=$1; shift
  local src=$1; shift
  if [ ! -e _bin/$n ] || [ $src -nt _bin/$n ] || [ ../../nerf-texture_amd/csrc/grid_record.hpp -nt _bin/$n ]; then
    hipcc $F "$@" $src -o _bin/$n 2> >(grep -v "not a recognized feature for this target\|argument unused" >&2)
  fi
}
build k3d_reduce k3d_reduce.hip
build k3d_reduce_nopk k3d_reduce.hip -Xclang -target-feature -Xclang -packed-fp32-ops
build pk_mfma_probe pk_mfma_probe.hip
for p in atomic_probe lds_atomic_probe; do build $p $p.hip; done
ls _bin
