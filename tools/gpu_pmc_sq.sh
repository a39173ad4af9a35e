#!/bin/bash
# SQ counter passes over the fp16 hash-grid backward microbench (8192 rays); summaries -> gpurun_out/<tag>/
out=$PWD/gpurun_out/${1:-pmcsq}
mkdir -p $out
export TMPDIR=/tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_SMEM"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for set in "$A" "$B"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops ${2:-grid_bwd} --dtypes f16 --rays 8192 > $out/p$i.log 2>&1 )
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    fs = glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("no counter file for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    with open("$out/summary_p%d.txt" % i, "w") as f:
        for k in agg:
            f.write(k + " dispatches=%d\n" % len(n[k]))
            for c, v in agg[k].items():
                f.write("   %-24s %.4g per dispatch\n" % (c, v / len(n[k])))
PY
find $out -name "*.csv" -size +5M -delete
cat $out/summary_p*.txt | head -150
