#!/usr/bin/env python3
"""gpurun_out/<tag>/ (written by tools/gpu_profiles_r06.sh on the GPU box) -> profiles/r06_* (tracked): copies the summaries and adds the
derived per-kernel figures DESIGN.md quotes.   python tools/collect_profiles_r06.py gpurun_out/r06prof"""
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


cp("bench.json", "r06_bench.json")
cp("kernel_stats.csv", "r06_kernel_stats.csv")            # the rocprofv3 child of the bench itself: the replayed step, 416 steps
cp("kernels.json", "r06_kernel_microbench.json")
cp("timeline/step.txt", "r06_step_timeline.txt")
cp("occupancy.json", "r06_occupancy_update.json")
cp("pytest.log", "r06_gpu_suite.txt")
for a, b in (("tile_adam.json", "r06_tile_adam_final.json"), ("table_update_ab.json", "r06_table_update_ab.json"), ("dead_skip.json", "r06_dead_skip_probe.json"),
             ("composite_step_ab.json", "r06_composite_step_ab.json"), ("composite_step_probe.json", "r06_composite_step_probe.json")):
    if os.path.exists(os.path.join(src, a)) and os.path.getsize(os.path.join(src, a)) > 10:
        cp(a, b)
cp("soak_pytest_tail.txt", "r06_gpu_suite_beside_a_training_neighbour.txt")

with open(os.path.join(dst, "r06_pmc_grid.txt"), "w") as f:
    f.write("# Memory-side bytes and L2 -> L1 requests per launch of EVERY kernel of the replayed training step, measured by bench.py itself (round 6): three\n"
            "# child runs of the bench under rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc TCP_TCC_READ_REQ_sum (separate passes, --kernel-trace only),\n"
            "# interquartile mean per kernel over the replayed steps.  roofline.traffic of the bench line = backward: 2 x FETCH (the guide's gfx950 correction for\n"
            "# wide coalesced streams) + WRITE over bin_fill + sum_tiles + combine_tiles; forward: FETCH + WRITE as reported.\n")
    f.write(open(os.path.join(src, "pmc_bench.txt")).read())


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


with open(os.path.join(dst, "r06_pmc_ffmlp.txt"), "w") as f:
    f.write("# SQ counters of the MLP kernels of the training step, three rocprofv3 --pmc passes (--kernel-trace only) over\n"
            "#   bench.py --no-graph --steps 16 --warmup 4 --no-kernel-timing --no-cpu-baseline --no-other --no-infer   (8192 rays, 459 k samples per launch)\n"
            "# per-dispatch averages; kernel names keep their template arguments: ffmlp_backward_fused_kernel<64, 3, 2, true, ReLU, FIELD=1> is the colour net's\n"
            "# backward (field_color_backward_kernel in bench.py's table), <64, 2, 2, true, ReLU, FIELD=2> the sigma net's; field_forward_kernel<true> = training.\n"
            "# Kernels run slower under the counters than in the bench (the cycles below are the profiled run's own).\n")
    sq_report(merged("sq_ffmlp"), f)
with open(os.path.join(dst, "r06_pmc_sq_grid.txt"), "w") as f:
    f.write("# SQ counters of the hash-grid and march kernels, three rocprofv3 --pmc passes over tools/bench_kernels.py --ops grid_fwd,grid_bwd,march --rays 8192 --dtypes f16\n")
    sq_report(merged("sq_grid"), f)
    f.write("\n# ... and of the table backward WITH the tile-owner Adam (round 6: sum_tiles_dir_kernel<half, true> = Lb1E, combine_tiles_kernel<true>) beside the plain\n"
            "# two-launch form (Lb0E + adam_half_kernel over the whole table), three passes over tools/tile_adam_probe.py (synthetic rays, 459 264 points)\n")
    sq_report(merged("sq_adam"), f)
with open(os.path.join(dst, "r06_pmc_l2.txt"), "w") as f:
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
print(sorted(x for x in os.listdir(dst) if x.startswith("r06_")))
