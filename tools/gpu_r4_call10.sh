#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c10}
mkdir -p $out
export TMPDIR=/tmp
for v in occ noocc; do
  if [ $v = noocc ]; then export NO_OCC=1; else unset NO_OCC; fi
  ( cd /tmp && ORDER=normal timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$v -- python $GRAFT_REPO_ROOT/tools/fresh_probe3.py > $out/probe_$v.log 2>&1 )
  grep -v amdgpu.ids $out/probe_$v.log | tail -3
  f=$(find $out/prof_$v -name "*kernel_stats.csv" | head -1)
  python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:22]:
    print(r["Name"][:64].ljust(66), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"], r["MaxNs"])
PY
done
find $out -name "*.csv" -size +3M -delete
