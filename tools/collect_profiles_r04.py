#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/gpu_profiles_r04.sh on the GPU box) -> profiles/r04_* (tracked): copies the summaries and adds the
derived per-kernel figures DESIGN.md quotes.   python tools/collect_profiles_r04.py gpurun_out/r04prof"""
import glob
import json
import os
import re
import shutil
import sys

src = sys.argv[1]
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def cp(a, b):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))


cp("bench.json", "r04_bench.json")
cp("bench_under_rocprof.json", "r04_bench_under_rocprof.json")
cp(os.path.relpath(glob.glob(os.path.join(src, "prof", "**", "*kernel_stats.csv"), recursive=True)[0], src), "r04_kernel_stats.csv")
cp("g2.json", "r04_g2_phase_ablation.json")
cp("kernels.json", "r04_kernel_microbench.json")
cp("infer.json", "r04_infer_frame.json")
cp("timeline/step.txt", "r04_step_timeline.txt")

cp("precision.json", "r04_precision.json")
cp("occupancy.json", "r04_occupancy_update.json")

with open(os.path.join(dst, "r04_pmc_grid.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over tools/bench_kernels.py --ops grid_fwd,grid_bwd\n"
            "# --rays 8192 --dtypes f16 (456 064 samples per launch); per-launch averages, KiB -> MB (tools/pmc_summary.py).\n"
            "# bench.py's roofline.traffic: backward 2 x FETCH (gfx950 correction for coalesced streams) + WRITE; forward FETCH + WRITE as reported.\n")
    f.write(open(os.path.join(src, "pmc_grid.txt")).read())


def parse(path):
    out, cur = {}, None
    for ln in open(path):
        if not ln.startswith(" "):
            cur = ln.split(" dispatches=")[0].strip()
            out[cur] = {}
        else:
            k, v = ln.split()[:2]
            out[cur][k] = float(v)
    return out


def merged(sub):
    m = {}
    for p in sorted(glob.glob(os.path.join(src, sub, "summary_p*.txt"))):
        for k, v in parse(p).items():
            m.setdefault(k, {}).update(v)
    return m


def sq_report(m, f, simds=1024):
    for k, c in m.items():
        f.write(k + "\n")
        for n, v in c.items():
            f.write("   %-30s %.5g\n" % (n, v))
        w = c.get("SQ_WAVES")
        if w and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
            f.write("   -> kernel length %.0f cycles (GRBM_GUI_ACTIVE / 8 XCDs); per wave: %.0f VALU, %.0f MFMA, %.0f LDS, %.0f SALU instructions\n"
                    % (cyc, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_MFMA", 0) / w, c.get("SQ_INSTS_LDS", 0) / w, c.get("SQ_INSTS_SALU", 0) / w))
            f.write("   -> per SIMD: VALU issue %.1f %% of the kernel's cycles (4 cycles per instruction), MFMA pipe busy %.1f %% (SQ_VALU_MFMA_BUSY_CYCLES / %d SIMDs)\n"
                    % (100 * c.get("SQ_INSTS_VALU", 0) * 4 / simds / cyc, 100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / simds / cyc, simds))


with open(os.path.join(dst, "r04_pmc_ffmlp.txt"), "w") as f:
    f.write("# SQ counters of the MLP kernels of the training step, three rocprofv3 --pmc passes (--kernel-trace only) over\n"
            "#   bench.py --no-graph --steps 16 --warmup 4 --no-kernel-timing --no-cpu-baseline --no-other --no-infer   (8192 rays, 459 k samples per launch)\n"
            "# per-dispatch averages; kernel names keep their template arguments: ffmlp_backward_fused_kernel<64, 3, 2, true, ReLU, FIELD=1> is the colour net's\n"
            "# backward (field_color_backward_kernel in bench.py's table), <64, 2, 2, true, ReLU, FIELD=2> the sigma net's; field_forward_kernel<true> = training.\n"
            "# Kernels run slower under the counters than in the bench (the cycles below are the profiled run's own).\n")
    sq_report(merged("sq_ffmlp"), f)
with open(os.path.join(dst, "r04_pmc_sq_grid.txt"), "w") as f:
    f.write("# SQ counters of the hash-grid and march kernels, three rocprofv3 --pmc passes over tools/bench_kernels.py --ops grid_fwd,grid_bwd,march --rays 8192 --dtypes f16\n")
    sq_report(merged("sq_grid"), f)
with open(os.path.join(dst, "r04_pmc_l2.txt"), "w") as f:
    f.write("# L2 (TCC) and L1 (TCP) counters of the hash-grid kernels, two rocprofv3 --pmc passes over tools/bench_kernels.py --ops grid_fwd,grid_bwd --dtypes f16 --rays 8192\n"
            "# (456 064 samples, fp16 table of 24 MiB); per-dispatch averages (tools/gpu_pmc_l2.sh)\n")
    m = merged("l2")
    for k, c in m.items():
        f.write(k + "\n")
        for n, v in c.items():
            f.write("   %-32s %.4g\n" % (n, v))
        if "TCC_HIT_sum" in c and "TCP_TOTAL_CACHE_ACCESSES_sum" in c:
            f.write("   -> L1: %.1f %% of %.3g accesses served without an L2 read; L2 hit rate %.1f %%; L2 -> L1 read traffic at 128 B per request: %.2f GB\n"
                    % (100 * (1 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]), c["TCP_TOTAL_CACHE_ACCESSES_sum"],
                       100 * c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1), c["TCP_TCC_READ_REQ_sum"] * 128 / 1e9))
print(sorted(x for x in os.listdir(dst) if x.startswith("r04_")))
