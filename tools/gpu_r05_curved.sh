#!/bin/bash
# configs[3]: the three kernels' timings per point order, rocprofv3 kernel stats, SQ counters (VALU issue, occupancy, waits) -> gpurun_out/r05_curved
out=$PWD/gpurun_out/r05_curved
mkdir -p $out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
: > $out/orders.jsonl
for o in random cell rays; do timeout 120 python tools/bench_curved.py --order $o >> $out/orders.jsonl 2>> $out/err.txt; done
if [ "$1" = "lib" ]; then for o in random rays; do timeout 120 python tools/bench_curved.py --order $o --library-order >> $out/orders.jsonl 2>> $out/err.txt; done; fi
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python $R/tools/bench_curved.py --order random > $out/stats.log 2>&1 )
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT"
B="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU"
C="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for set in "$A" "$B" "$C"; do
  i=$((i+1))
  for o in random cell; do
    ( cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p${i}_$o -- python $R/tools/bench_curved.py --order $o --reps 3 > $out/p${i}_$o.log 2>&1 )
  done
done
python - <<PY
import csv, glob, collections
out = "$out"
with open(out + "/pmc_summary.txt", "w") as f:
    for o in ("random", "cell"):
        f.write("== point order: %s\n" % o)
        for i in (1, 2, 3):
            fs = glob.glob(out + "/p%d_%s/**/*counter_collection.csv" % (i, o), recursive=True)
            if not fs:
                f.write("no counter file for pass %d\n" % i); continue
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(fs[0])):
                k = next((x for x in ("knn_query_kernel", "curved_project_kernel", "raytrace_kernel") if x in r["Kernel_Name"]), None)
                if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k in agg:
                for c, v in agg[k].items():
                    f.write("%-24s %-30s %.5g per dispatch (%d dispatches)\n" % (k, c, sum(v) / len(v), len(v)))
print(open(out + "/pmc_summary.txt").read())
PY
find $out -name "*.csv" -size +2M -delete; find $out -name "*agent_info.csv" -delete
cat $out/orders.jsonl; grep -h "knn_query\|curved_project\|raytrace" $out/stats/*/*kernel_stats.csv 2>/dev/null | cut -c1-200
