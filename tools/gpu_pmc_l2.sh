#!/bin/bash
# L2 / L1 counter passes over the fp16 hash-grid microbench (8192 rays): hit rates and request counts of the gather kernels
out=$PWD/gpurun_out/${1:-pmcl2}
mkdir -p $out
export TMPDIR=/tmp
A="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
B="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"
i=0
for set in "$A" "$B"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops grid_fwd,grid_bwd --dtypes f16 --rays 8192 > $out/p$i.log 2>&1 )
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
    fs = glob.glob("$out/p%d/**/*counter_collection.csv" % i, recursive=True)
    if not fs:
        print("no counter file for pass", i); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])):
        name = r["Kernel_Name"]
        k = next((x for x in ("grid_forward_level_kernel", "level_major_to_rows_kernel", "bin_fill_dir_kernel", "sum_tiles_dir_kernel", "combine_tiles_kernel") if x in name), None)
        if k is None:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    with open("$out/summary_p%d.txt" % i, "w") as f:
        for k in agg:
            f.write(k + " dispatches=%d\n" % len(n[k]))
            for c, v in agg[k].items():
                f.write("   %-32s %.4g per dispatch\n" % (c, v / len(n[k])))
PY
find $out -name "*.csv" -size +5M -delete
cat $out/summary_p*.txt | head -80; tail -3 $out/p1.log
