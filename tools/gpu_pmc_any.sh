#!/bin/bash
# SQ counter passes over an arbitrary python command; per-kernel per-dispatch averages -> gpurun_out/<tag>/summary_p*.txt
# usage: gpu_pmc_any.sh <tag> <kernel-name filter regex> -- <python args...>
tag=$1; filt=$2; shift 3
out=$PWD/gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F16"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
C="SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  ( cd /tmp && timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python "$@" > $out/p$i.log 2>&1 )
done
python - <<PY
import csv, glob, collections, re
for i in (1, 2, 3):
    fs = glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("no counter file for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]
        if not re.search(r"$filt", name): continue
        m = re.search(r"\d+([a-z_0-9]+_kernel)(I.*?)?Ev", name) if name.startswith("_Z") else re.search(r"(\w+_kernel)(<[^>]*>)?", name)
        k = (m.group(1) + (m.group(2) or "")) if m else name[:80]  # (mangled names keep their template arguments: Li3E = 3, Lb1E = true)
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    with open("$out/summary_p%d.txt" % i, "w") as f:
        for k in sorted(agg):
            f.write(k + " dispatches=%d\n" % len(n[k]))
            for c, v in agg[k].items():
                f.write("   %-28s %.5g per dispatch\n" % (c, v / len(n[k])))
PY
find $out -name "*.csv" -size +5M -delete
cat $out/summary_p*.txt | head -${LINES_OUT:-120}
