#!/usr/bin/env python3
"""Turn gpurun_out/profiles_<tag>/ into the committed summaries under profiles/."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "profiles_" + tag)
dst = os.path.join(root, "profiles")
KERNEL = "vit_pk_kernel"

lines = []
bench = None
with open(os.path.join(src, "bench.json")) as f:
    for line in f:
        if line.startswith("{"):
            bench = json.loads(line)
# (the header lines and profiles/<tag>_bench.json are written at the END: the bench run of a collection precedes its counter
#  passes, so the counters it attached from profiles/pmc_traffic.json are the previous collection's - they are replaced below
#  by the ones of THIS collection)

lines.append("\n# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-pipelined   (kernel_stats.csv, top rows; the bench line's `pipelined` leg is left out: its overlapping launches are not what roofline.kernel_ms prices)\n")
with open(os.path.join(src, "kernel_stats.csv")) as f:
    rows = list(csv.DictReader(f))
with open(os.path.join(src, "kernel_stats.csv")) as f, open(os.path.join(dst, "%s_kernel_stats_top.csv" % tag), "w") as g:
    g.writelines(f.readlines()[:9])  # header + the eight kernels with the most time: the same run the summary quotes
avg_ns = None
for r in rows[:6]:
    lines.append("%-60.60s calls=%s avg_ns=%s total_ns=%s pct=%s\n" % (r["Name"], r["Calls"], r["AverageNs"], r["TotalDurationNs"], r["Percentage"]))
    if KERNEL in r["Name"]:
        avg_ns = float(r["AverageNs"])

tot = {}
meta = {}
pmc_json = None
for p in ("p1", "p2", "p3", "p4"):
    fn = os.path.join(src, "pmc_%s.csv" % p)
    if not os.path.exists(fn):
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fn)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size") if k in r}
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
lines.append("\n# rocprofv3 --pmc ... (separate passes), mean per launch of %s; dispatch %s\n" % (KERNEL, json.dumps(meta)))
for k, v in sorted(tot.items()):
    lines.append("%-22s %.6g\n" % (k, v))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    # MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly 1/2 of a
    # coalesced streaming read -> double it.  Our read volume is known (the symbol buffer), which calibrates it.
    fetch = 2.0 * tot["FETCH_SIZE"] * 1024.0
    write = tot["WRITE_SIZE"] * 1024.0
    traffic = fetch + write
    lines.append("\nHBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = %.4g + %.4g = %.4g B; algorithmic %.4g B; ratio %.3f\n"
                 % (fetch, write, traffic, alg, traffic / alg))
    pmc_json = {"round": tag, "kernel": KERNEL,
                # the configuration these counters belong to: bench.py attaches them only to a run of the same one
                "config": {"frames_per_gpu": bench["config"].get("frames_per_gpu", 65536), "kernel": bench["config"].get("kernel", 0),
                           "mode": bench.get("mode", "shard"), "renorm_ge": bench["config"].get("renorm_ge", 0)},
                "source": "tools/collect_profiles.sh %s: rocprofv3 --pmc passes of `python bench.py --no-cpu`" % tag,
                "hbm_bytes_per_launch": int(traffic), "fetch_bytes_corrected": int(fetch),
                "write_bytes": int(write), "algorithmic_bytes_per_launch": alg,
                "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; KiB units; "
                          "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports 1/2 of a streaming read)"}
if avg_ns:
    lines.append("kernel avg (rocprof) %.1f us vs bench HIP-event kernel_ms %.4f ms\n" % (avg_ns / 1e3, bench["roofline"]["kernel_ms"]))
if "SQ_ACTIVE_INST_VALU" in tot and "GRBM_GUI_ACTIVE" in tot:
    # MI355X_MICROARCH.md: SQ_ACTIVE_INST_* count quad-cycles per wave; GRBM_GUI_ACTIVE is summed over the 8 XCDs
    kernel_cycles = tot["GRBM_GUI_ACTIVE"] / 8.0
    valu_busy = tot["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * kernel_cycles)
    lines.append("kernel cycles %.4g (clock %.2f GHz); VALU busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * cycles) = %.1f %%\n"
                 % (kernel_cycles, kernel_cycles / (avg_ns or 1) , 100.0 * valu_busy))
    if pmc_json is not None:
        pmc_json["valu_busy"] = round(valu_busy, 4)
if "SQ_LDS_BANK_CONFLICT" in tot and "SQ_LDS_IDX_ACTIVE" in tot and tot["SQ_LDS_IDX_ACTIVE"]:
    lines.append("LDS: bank-conflict cycles / index-active cycles = %.1f %%; LDS-instruction busy = SQ_ACTIVE_INST_LDS*4 / (256 CUs * cycles) = %.1f %%\n" % (
        100.0 * tot["SQ_LDS_BANK_CONFLICT"] / tot["SQ_LDS_IDX_ACTIVE"],
        100.0 * tot.get("SQ_ACTIVE_INST_LDS", 0) * 4.0 / (256.0 * (tot.get("GRBM_GUI_ACTIVE", 0) / 8.0 or 1))))
if "SQ_WAVES" in tot:
    w = tot["SQ_WAVES"]
    if pmc_json is not None and "SQ_INSTS_VALU" in tot:
        steps = bench["config"]["framebits"] + 6
        pmc_json["valu_insts_per_wave"] = round(tot["SQ_INSTS_VALU"] / w, 1)
        pmc_json["valu_insts_per_frame_step"] = round(tot["SQ_INSTS_VALU"] / w / steps / 4.0, 3)  # 4 frames per wave
    lines.append("per wave: VALU %.0f  SALU %.0f  LDS %.0f  wave-cycles(quad) %.0f\n" % (
        tot.get("SQ_INSTS_VALU", 0) / w, tot.get("SQ_INSTS_SALU", 0) / w, tot.get("SQ_INSTS_LDS", 0) / w, tot.get("SQ_WAVE_CYCLES", 0) / w))
# second stage (rs_kernel) of the same runs: duration from the kernel trace, HBM bytes from the FETCH/WRITE passes
ss = bench.get("second_stage") or {}
rs_rows = [r for r in rows if "rs_kernel" in r["Name"]]
if rs_rows and "ms" in ss:
    lines.append("\n# second stage: %s\n" % ss.get("kernel"))
    lines.append("rs_kernel avg (rocprof, incl. the untimed pre-conditioning launches) %.1f us, min %.1f us vs bench HIP-event ms %.4f\n"
                 % (float(rs_rows[0]["AverageNs"]) / 1e3, float(rs_rows[0]["MinNs"]) / 1e3, ss["ms"]))
rs_tot = {}
for pp in ("p1", "p2"):
    fn = os.path.join(src, "pmc_%s.csv" % pp)
    if os.path.exists(fn):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fn)):
            if "rs_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            rs_tot[k] = sum(v) / len(v)
if "FETCH_SIZE" in rs_tot and "WRITE_SIZE" in rs_tot and ss.get("roofline"):
    rs_alg = ss["roofline"]["algorithmic_bytes_per_launch"]
    rs_fetch, rs_write = 2.0 * rs_tot["FETCH_SIZE"] * 1024.0, rs_tot["WRITE_SIZE"] * 1024.0
    lines.append("rs_kernel HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = %.4g + %.4g = %.4g B; algorithmic %.4g B; ratio %.3f\n"
                 % (rs_fetch, rs_write, rs_fetch + rs_write, rs_alg, (rs_fetch + rs_write) / rs_alg))
    if pmc_json is not None:
        pmc_json["second_stage"] = {"kernel": "rs_kernel", "hbm_bytes_per_launch": int(rs_fetch + rs_write),
                                    "fetch_bytes_corrected": int(rs_fetch), "write_bytes": int(rs_write),
                                    "algorithmic_bytes_per_launch": rs_alg}
if pmc_json is not None:
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
        json.dump(pmc_json, f, indent=1)
    # what bench.py's _attach_cached_counters would attach from the file just written
    r = bench["roofline"]
    r["traffic"] = pmc_json["hbm_bytes_per_launch"]
    r["traffic_source"] = "cached from profiles/pmc_traffic.json (%s)" % pmc_json["source"]
    for k in ("valu_busy", "valu_insts_per_frame_step", "valu_insts_per_wave"):
        if k in pmc_json:
            r[k] = pmc_json[k]
    ss_r = (bench.get("second_stage") or {}).get("roofline")
    if ss_r and pmc_json.get("second_stage"):
        ss_r["traffic"] = pmc_json["second_stage"]["hbm_bytes_per_launch"]
        ss_r["traffic_source"] = r["traffic_source"]
with open(os.path.join(dst, "%s_bench.json" % tag), "w") as f:
    json.dump(bench, f, indent=1)
head = ["# %s — bench.py (default flags) on MI355X; counters in `roofline` = the passes summarised below\n" % tag,
        "value %.1f %s, ms_per_step %.4f, roofline %s\n" % (bench["value"], bench["unit"], bench["ms_per_step"], json.dumps(bench["roofline"])),
        "pipelined %s\n" % json.dumps(bench.get("pipelined")),
        "cpu_baseline %s\nparity %s\n" % (json.dumps(bench.get("cpu_baseline")), json.dumps(bench.get("parity")))]
lines = head + lines
vc = int(meta.get("VGPR_Count", 0) or 0)
lines.append("\n# note on the dispatch columns of rocprofv3's counter CSV: VGPR_Count %d = (granulated_workitem_vgpr_count + 1) * 4,\n"
             "# i.e. the kernel descriptor's %d granules priced at the pre-gfx90a granule of 4; gfx950 allocates in granules of 8, so\n"
             "# the wave holds %d * 8 = %d registers (the code object's .vgpr_count rounded up: 114 in rounds 1-2, 128 since the fast\n"
             "# traceback form; 512 / %d -> %d waves per SIMD).\n" % (vc, vc // 4, vc // 4, vc * 2, vc * 2, 512 // max(vc * 2, 1)))
lines.append("# LDS_Block_Size 0 is the STATIC group segment (.group_segment_fixed_size 0); the 10240 B per workgroup are DYNAMIC LDS\n"
             "# passed at launch (hipLaunchKernelGGL's sharedMemBytes), and they are what limits residency to 16 waves per CU.\n")
open(os.path.join(dst, "%s_rocprof_summary.txt" % tag), "w").writelines(lines)
print("".join(lines))
