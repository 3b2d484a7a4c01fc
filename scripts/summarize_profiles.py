#!/usr/bin/env python3
"""Copies the rocprofv3 summaries a gpurun call left under gpurun_out/ into profiles/ (tracked) and
derives the per-launch HBM traffic of the stage-1 kernel from the PMC passes.

Corrections (MI355X_MICROARCH.md, section HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a 16 B/lane coalesced streaming read -> doubled.
The factor is re-calibrated here on permute_rows_kernel, whose read volume is known exactly
(it copies every HLL and SuperMinHash row once with the same 16 B/lane loads).
usage: summarize_profiles.py <tag> <workload:algo> <stage1 kernel name substring> [known_permute_bytes]
"""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, key, kname = sys.argv[1], sys.argv[2], sys.argv[3]
known = float(sys.argv[4]) if len(sys.argv) > 4 else None
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
g = ROOT / "gpurun_out"
shutil.copy(g / f"prof_{tag}" / "trace_kernel_stats.csv", out / f"{tag}_kernel_stats.csv")
for f in (g / f"bench_{tag}.json",):
    if f.exists():
        shutil.copy(f, out / f.name)
summary = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(g / f"pmc_{c}_{tag}" / "pmc_counter_collection.csv")):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    summary[c] = {k: {"launches": len(v), "avg_KiB": sum(v) / len(v)} for k, v in agg.items()}
fetch_fix = 2.0
perm = [v for k, v in summary["FETCH_SIZE"].items() if "permute_rows_kernel" in k]
if perm and known:
    measured = perm[0]["avg_KiB"] * 1024 * perm[0]["launches"]
    fetch_fix = known / measured
    summary["fetch_calibration"] = {"known_bytes": known, "counter_bytes": measured, "factor": fetch_fix}
s1f = [v for k, v in summary["FETCH_SIZE"].items() if kname in k][0]["avg_KiB"] * 1024 * fetch_fix
s1w = [v for k, v in summary["WRITE_SIZE"].items() if kname in k][0]["avg_KiB"] * 1024
summary["stage1"] = {"kernel": kname, "hbm_read_bytes_per_launch": s1f, "hbm_write_bytes_per_launch": s1w,
                     "traffic_bytes_per_launch": s1f + s1w}
(out / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
tfile = out / "stage1_traffic.json"
t = json.loads(tfile.read_text()) if tfile.exists() else {}
t[key] = s1f + s1w
tfile.write_text(json.dumps(t, indent=1))
print(json.dumps(summary["stage1"]), summary.get("fetch_calibration"))
