#!/usr/bin/env python3
"""gpurun_out/configs_<tag>/ (tools/collect_config_profiles.sh) -> committed summaries under profiles/."""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "configs_" + tag)
dst = os.path.join(root, "profiles")
OURS = ("vit_pk", "vit_lat", "vit_wave", "vit_pack", "rs_kernel", "desc_")

with open(os.path.join(dst, "%s_other_configs.jsonl" % tag), "w") as f:
    for name in ("configs.jsonl", "rs.jsonl", "hostpaths.jsonl"):
        for line in open(os.path.join(src, name)):
            if line.startswith("{"):
                f.write(line)
with open(os.path.join(dst, "%s_vitbench.txt" % tag), "w") as f:
    f.write("# %s: tools/vitbench.cpp (tools/collect_config_profiles.sh) on one MI355X box: the five reference exports only (dlopen, no HIP in the\n"
            "# harness); every multi-thread result compared with the single-call result; 'ingest stage on (defaults)': window 50 us, min_callers 1,\n"
            "# depth 4, spin_cpus = CPU budget / 4\n" % tag)
    f.write(open(os.path.join(src, "vitbench.txt")).read())
if os.path.exists(os.path.join(src, "small_batch.jsonl")):
    with open(os.path.join(dst, "%s_small_batch.jsonl" % tag), "w") as f:
        f.writelines(l for l in open(os.path.join(src, "small_batch.jsonl")) if l.startswith("{"))
with open(os.path.join(dst, "%s_soak.txt" % tag), "w") as f:
    lines = [l for l in open(os.path.join(src, "soak.jsonl")) if l.startswith("{")]
    reps = 1 + max([__import__("json").loads(l).get("rep", 0) for l in lines] or [0])
    f.write("# tests/tools/soak.py %d on MI355X: GPU (auto kernel) vs CPU oracle, bit-exact compare of every output byte\n"
            "# decode (%d seed sets): lengths 768/288/1536/3072/6912/9216/776/784/770/9214 x {Eb/N0 3 dB, 0 dB, 12 dB, uniform random bytes, saturation/renormalisation stress patterns};\n"
            "# RS (%d seed sets): superframes with 0..8 symbol errors per column in three regimes (dense correctable, dense with failures, sparse), RSDims 24/12/5/1/37/256/300/8/3\n" % (reps, reps, reps))
    for line in open(os.path.join(src, "soak.jsonl")):
        if "total_frames" in line or "rs_superframes" in line or '"rsdims"' in line:
            f.write(line)

out = ["# %s - rocprofv3 --kernel-trace --stats -- python3 tests/tools/bench_configs.py  (this repo's kernels only)\n" % tag]
for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
    if any(k in r["Name"] for k in OURS):
        out.append("%-64.64s calls=%-4s avg_us=%9.1f min_us=%9.1f max_us=%9.1f\n" % (
            r["Name"].replace("(anonymous namespace)::", ""), r["Calls"], float(r["AverageNs"]) / 1e3,
            float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
out.append("\n# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of the same script, mean per launch, grouped by\n"
           "# kernel and grid size; HBM bytes = 2*FETCH_SIZE*1024 (gfx950 correction, MI355X_MICROARCH.md) and WRITE_SIZE*1024\n")
agg = collections.defaultdict(dict)
for p, cname in (("p1", "FETCH_SIZE"), ("p2", "WRITE_SIZE")):
    tmp = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(src, "pmc_%s.csv" % p))):
        if any(k in r["Kernel_Name"] for k in OURS) and r["Counter_Name"] == cname:
            tmp[(r["Kernel_Name"].replace("(anonymous namespace)::", "")[:48], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in tmp.items():
        agg[k][cname] = sum(v) / len(v)
for (name, grid), c in sorted(agg.items()):
    rd = 2.0 * c.get("FETCH_SIZE", 0) * 1024
    wr = c.get("WRITE_SIZE", 0) * 1024
    out.append("%-48s grid=%-9s read %9.1f MB  write %9.1f MB\n" % (name, grid, rd / 1e6, wr / 1e6))
out.append("\n# reading the table: launches are grouped by kernel and grid only, so several cases of bench_configs.py share a line (4096 persistent\n"
           "# workgroups = grid 262144: config 3, config 5 and the multi-round uniform cases).  Per-case traffic against the algorithmic bytes is in\n"
           "# %s_traffic_long.jsonl (tools/exp/pmc_traffic_long.sh).\n" % tag)
open(os.path.join(dst, "%s_config_kernels.txt" % tag), "w").writelines(out)
print("".join(out))
