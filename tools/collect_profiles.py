#!/usr/bin/env python3
"""Copy the summaries of a tools/profile_round.sh run (gpurun_out/prof_<tag>/) into profiles/ under the names
profiles/README.md lists: python tools/collect_profiles.py r02"""
import csv
import glob
import os
import shutil
import sys

def newest(pattern):
    """gpurun merges every call's output into gpurun_out/: keep the most recent run's file only"""
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1:]


tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
for d in sorted(glob.glob(os.path.join(src, "cfg*"))):
    c = os.path.basename(d)
    for name in ("bench.json", "bench_under_rocprof.json", "phase_cycles.txt", "phase_cycles_path.txt"):
        if os.path.exists(os.path.join(d, name)):
            shutil.copy(os.path.join(d, name), os.path.join(dst, f"{tag}_{c}_{name}"))
    for f in newest(os.path.join(d, "trace", "**", "*kernel_stats.csv")):
        shutil.copy(f, os.path.join(dst, f"{tag}_{c}_kernel_stats.csv"))
    for f in newest(os.path.join(d, "trace", "**", "*kernel_trace.csv")):
        rows = open(f).read().splitlines()
        keep = [rows[0]] + [r for r in rows[1:] if "solve_kernel" in r or "stage_kernel" in r][:3]
        open(os.path.join(dst, f"{tag}_{c}_kernel_trace_head.csv"), "w").write("\n".join(keep) + "\n")
    for p in ("fetch", "write", "sq1", "sq2", "flops"):
        for f in newest(os.path.join(d, "pmc_" + p, "**", "*counter_collection.csv")):
            rows = list(csv.reader(open(f)))
            keep = [rows[0]] + [r for r in rows[1:] if "solve_kernel" in r[8] or "stage_kernel" in r[8]]
            csv.writer(open(os.path.join(dst, f"{tag}_{c}_pmc_{p}.csv"), "w"), quoting=csv.QUOTE_NONNUMERIC).writerows(keep)
for f in newest(os.path.join(src, "lowlevel", "**", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_lowlevel_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "pmc_summary.json")):
    shutil.copy(os.path.join(src, "pmc_summary.json"), os.path.join(dst, "pmc_summary.json"))
print("copied", tag, "into", dst)
