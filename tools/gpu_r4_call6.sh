#!/bin/bash
out=$PWD/gpurun_out/${1:-r4c6}
mkdir -p $out
export TMPDIR=/tmp
timeout 300 python tools/fresh_probe.py > $out/probe.log 2>&1
( cd /tmp && GROUPS=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -- python $GRAFT_REPO_ROOT/tools/fresh_probe.py > $out/probe_prof.log 2>&1 )
cat $out/probe.log | grep -v amdgpu.ids
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:28]:
    print(r["Name"][:70].ljust(72), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(9), r["Percentage"])
PY
find $out -name "*.csv" -size +3M -delete
