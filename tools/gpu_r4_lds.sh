#!/bin/bash
# LDS staging of levels 0-1 in the hash-grid forward -- test, then time and count L1 accesses with and without
out=$PWD/gpurun_out/${1:-r4lds}
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -k lds_staged > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for n in 0 2; do
  NERFTEX_TUNE="grid_fwd_lds=$n" timeout 200 python tools/bench_kernels.py --rays 8192 --ops grid_fwd --dtypes f16 --kernels > $out/time_lds$n.json 2>> $out/err.log
  ( cd /tmp && NERFTEX_TUNE="grid_fwd_lds=$n" timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $out/pmc$n -- python $GRAFT_REPO_ROOT/tools/bench_kernels.py --ops grid_fwd --dtypes f16 --rays 8192 > $out/pmc$n.log 2>&1 )
done
python - <<PY
import csv, glob, collections, json
res = {}
for n in (0, 2):
    t = json.load(open("$out/time_lds%d.json" % n))
    res["grid_fwd_lds=%d" % n] = {"call_ms": t["grid_fwd_f16"]["ms"], "kernels_avg_us": {k: v for k, v in t["kernels_avg_us"].items() if "grid_forward" in k or "level_major" in k}}
    fs = glob.glob("$out/pmc%d/**/*counter_collection.csv" % n, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(fs[0])) if fs else []:
        k = next((x for x in ("grid_forward_level_kernel", "grid_forward_lds_kernel") if x in r["Kernel_Name"]), None)
        if k:
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    res["grid_fwd_lds=%d" % n]["pmc_per_dispatch"] = {k: {c: v / len(cnt[k]) for c, v in agg[k].items()} for k in agg}
print(json.dumps(res, indent=1))
json.dump(res, open("$out/lds_ab.json", "w"), indent=1)
PY
find $out -name "*.csv" -size +3M -delete
