#!/bin/bash
# do the RCCL kernels a data-parallel step runs beside this library's MFMA kernels contain the packed-fp32 chains of DESIGN 7.1?
# CPU only.  Extracts the gfx950 code object of the RCCL that torch loads (torch/lib/librccl.so: a compressed offload bundle in .hip_fatbin),
# disassembles it and counts v_pk_{add,mul,fma}_f32 per function.  ~2 min, ~600 MB under /tmp.   bash tools/rccl_packed_fp32_scan.sh > profiles/r06_rccl_packed_fp32_scan.txt
set -e
LLVM=/opt/rocm/lib/llvm/bin
LIB=${1:-$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so'))")}
W=$(mktemp -d /tmp/rccl_scan.XXXX)
$LLVM/llvm-objcopy --dump-section .hip_fatbin=$W/fatbin.bin $LIB
$LLVM/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$W/fatbin.bin --output=$W/gfx950.co
echo "library: $LIB ($(stat -c %s $LIB) bytes); gfx950 code object: $(stat -c %s $W/gfx950.co) bytes, $($LLVM/llvm-readelf --dyn-syms $W/gfx950.co | grep -c FUNC) functions"
$LLVM/llvm-objdump -d --no-show-raw-insn $W/gfx950.co | awk '
/^[0-9a-f]+ <.*>:$/ { sym=$2; next }
{ n++ }
/v_pk_(add|mul|fma)_f32/ { pk[sym]++; tot++; m[$1]++; if ($0 ~ /neg_lo|neg_hi|op_sel:/) mod++ }
/v_mfma/ { mf++ }
END { print "instructions", n, " packed fp32", tot, "(with neg / op_sel modifiers:", mod+0 ")  mfma", mf+0; for (k in m) print "  ", k, m[k]; print "per function:"; for (s in pk) print pk[s], s }' | sort -k1,1 -rn -s | sed 's/^\([a-z]\)/\1/'
rm -rf $W
