"""Per-kernel averages of rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over tools/bench_kernels.py (what profiles/*pmc_grid.txt holds).

    python tools/pmc_summary.py gpurun_out/<tag>/pmc_FETCH_SIZE gpurun_out/<tag>/pmc_WRITE_SIZE
"""
import collections
import csv
import glob
import re
import sys

KERNELS = ("bin_fill_dir_kernel", "sum_tiles_dir_kernel", "combine_tiles_kernel", "grid_forward_level_kernel", "level_major_to_rows_kernel", "bin_fill_kernel",
           "sum_tiles_kernel", "bin_count_kernel", "grid_forward_kernel", "grid_backward_kernel")
rows = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = next((k for k in KERNELS if re.search(r"\d*" + k + r"(I|E|<|\()", name) or name.endswith(k)), None)
            if k is None:
                continue
            dt = "f16" if ("DF16_" in name or "_Float16" in name or "half" in name) else "f32"
            rows[(r["Counter_Name"], k, dt)].append(float(r["Counter_Value"]))
sums = collections.defaultdict(float)
for (c, k, dt), v in sorted(rows.items()):
    mb = sum(v) / len(v) * 1024 / 1e6  # rocprofv3 reports KiB
    print(f"{c:11s} {k:28s} {dt}   launches={len(v):3d}   {mb:9.2f} MB/launch")
    sums[(c, dt, "fwd" if "forward" in k or "rows" in k else "bwd")] += mb
print()
for key, v in sorted(sums.items()):
    print(f"sum {key}: {v:.1f} MB/launch")
