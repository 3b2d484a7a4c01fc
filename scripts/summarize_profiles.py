#!/usr/bin/env python3
"""Copies the rocprofv3 summaries of one scripts/gpu_profile.sh run (gpurun_out/prof_<tag>/) into profiles/ (tracked) and derives
per-kernel figures from the PMC passes.

  usage: summarize_profiles.py <tag> <traffic key, e.g. cfg3:sig> <dominant stage-1 kernel name substring>

HBM traffic (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of
a 16 B/lane coalesced streaming read.  The factor is re-calibrated in the same run on permute_rows_kernel, whose read volume is
known exactly (synth_device copies every HLL / SuperMinHash / auxiliary row once with 16 B/lane loads): known bytes / counter bytes.
SQ counters: SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over all SIMDs; SQ_LDS_IDX_ACTIVE LDS-array cycles
summed over the CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (kernel cycles = / 8).  Kernels run slower under counter collection:
the fractions, not the cycle counts, are the evidence.
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag, key, kname = sys.argv[1], sys.argv[2], sys.argv[3]
g = ROOT / "gpurun_out" / f"prof_{tag}"
out = ROOT / "profiles"
out.mkdir(exist_ok=True)
shutil.copy(g / "trace" / "trace_kernel_stats.csv", out / f"{tag}_kernel_stats.csv")
bench = json.loads((g / "bench.json").read_text().strip().splitlines()[-1])
(out / f"{tag}_bench.json").write_text(json.dumps(bench) + "\n")
cfg = bench["config"]
n, m = cfg["n_genomes"], cfg["m"]
known = n * (16384 + m * 8) + (n * 256 if "hll_a" in cfg.get("criterion", "") else 0)


def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(g / f"pmc_{name}" / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0]


summary = {"command": "bench.py " + " ".join(f"{a}" for a in sys.argv[4:]), "workload": cfg["workload"], "kernels": {}}
fetch, write = pmc("FETCH_SIZE"), pmc("WRITE_SIZE")
fix = 2.0
perm = [v["FETCH_SIZE"] for k, v in fetch.items() if "permute_rows_kernel" in k]
if perm:
    counter_bytes = sum(perm[0]) * 1024
    fix = known / counter_bytes
    summary["fetch_calibration"] = {"kernel": "permute_rows_kernel", "known_bytes": known, "counter_bytes": counter_bytes, "factor": fix}
sq = [pmc("SQ_INSTS_VALU"), pmc("SQ_LDS_IDX_ACTIVE")]
names = set(fetch) | set(write) | set(sq[0]) | set(sq[1])
for k in sorted(names):
    if not any(s in k for s in ("join", "smh_stream", "hll_union_hist", "hll_bitslice", "verify16", "ertl_select", "sig_build", "aux_fused", "csr_", "stream_interleave", "small_pass")):
        continue
    e = {}
    if k in fetch:
        e["hbm_read_bytes_per_launch"] = sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"]) * 1024 * fix
    if k in write:
        e["hbm_write_bytes_per_launch"] = sum(write[k]["WRITE_SIZE"]) / len(write[k]["WRITE_SIZE"]) * 1024
    for s in sq:
        for c, v in s.get(k, {}).items():
            e[c] = sum(v) / len(v)
    if "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"] > 0:
        cyc = e["GRBM_GUI_ACTIVE"] / 8.0
        e["kernel_cycles"] = cyc
        if "SQ_ACTIVE_INST_VALU" in e:
            e["valu_busy_frac"] = e["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc)
        if "SQ_INSTS_VALU" in e:
            e["cycles_per_valu_instr_per_simd"] = 1024 * cyc / e["SQ_INSTS_VALU"]
        if "SQ_WAVE_CYCLES" in e:
            e["avg_waves_per_simd"] = e["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc)
        if "SQ_LDS_IDX_ACTIVE" in e:
            e["lds_array_busy_frac"] = e["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if "SQ_INSTS_SALU" in e:
            e["salu_issue_frac"] = e["SQ_INSTS_SALU"] / (256 * cyc)
    summary["kernels"][short(k)] = e
dom = [v for k, v in summary["kernels"].items() if kname in k]
if dom:
    traffic = dom[0].get("hbm_read_bytes_per_launch", 0) + dom[0].get("hbm_write_bytes_per_launch", 0)
    summary["stage1"] = {"kernel": kname, "traffic_bytes_per_launch": traffic}
    tfile = out / "stage1_traffic.json"
    t = json.loads(tfile.read_text()) if tfile.exists() else {}
    t[key] = traffic
    t["source"] = "profiles/*_pmc_summary.json"
    tfile.write_text(json.dumps(t, indent=1))
hk = [k for k in fetch if "hll_union_hist_runs" in k or "hll_union_hist_bs" in k]
sk = [k for k in fetch if "sig_build_kernel" in k or "cb_bounds_kernel" in k]       # one launch per pass
if hk and sk:
    # stage 2a: bytes fetched from beyond L2 per PASS (all windows and chunk lanes of a pass together)
    passes = sum(len(fetch[k]["FETCH_SIZE"]) for k in sk)
    per_pass = sum(sum(fetch[k]["FETCH_SIZE"]) for k in hk) * 1024 * fix / passes
    tfile = out / "stage2_traffic.json"
    t = json.loads(tfile.read_text()) if tfile.exists() else {}
    t[key.split(":")[0]] = per_pass
    t["source"] = "profiles/*_pmc_summary.json (FETCH_SIZE of the stage-2a histogram kernel, calibrated; bytes per pass)"
    tfile.write_text(json.dumps(t, indent=1))
    summary["stage2"] = {"kernel": short(hk[0]), "beyond_l2_bytes_per_pass": per_pass}
# instruction counts per STEP (all launches of a pass) of the two long kernels, for bench.py's model_vs_pmc fields
cfile = out / "pmc_counts.json"
cj = json.loads(cfile.read_text()) if cfile.exists() else {}
wk = key.split(":")[0]
passes = sum(len(sq[0][k]["SQ_INSTS_VALU"]) for k in sq[0] if "sig_build_kernel" in k or "cb_bounds_kernel" in k) or 1
ent = {}
for label, pat in (("join", ("sigl_join_kernel", "sig16_join_kernel", "sig_join_kernel")), ("hist", ("hll_union_hist_bs", "hll_union_hist_runs"))):
    ks = [k for k in sq[0] if any(p in k for p in pat) and "SQ_INSTS_VALU" in sq[0][k]]
    if ks:
        ent[label] = {"SQ_INSTS_VALU": sum(sum(sq[0][k]["SQ_INSTS_VALU"]) for k in ks) / passes, "kernel": short(ks[0]), "profile": f"profiles/{tag}_pmc_summary.json"}
if ent:
    cj[wk] = ent
    cj["source"] = "rocprofv3 --pmc SQ_INSTS_VALU, wave-instructions per step (scripts/gpu_profile.sh + scripts/summarize_profiles.py)"
    cfile.write_text(json.dumps(cj, indent=1))
(out / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
print(json.dumps(summary.get("stage1")), summary.get("fetch_calibration"))
for k, e in summary["kernels"].items():
    print(k[:40], {a: (round(b, 3) if b < 100 else f"{b:.4g}") for a, b in e.items() if a in ("valu_busy_frac", "cycles_per_valu_instr_per_simd", "avg_waves_per_simd", "lds_array_busy_frac", "salu_issue_frac", "hbm_read_bytes_per_launch")})
