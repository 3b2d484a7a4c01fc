#!/usr/bin/env python3
"""development probe: does the per-kernel time depend on how busy the GPU is kept (DVFS)?"""
import sys, os, time, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
cfg = pkg.SYNTH_CONFIGS["cfg3"]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "sclk" in l or "mclk" in l)[:300]
    except Exception as e:
        return str(e)
print("idle clocks:", smi())
for algo, name in ((pkg.ALGO_SIG, "sig"), (pkg.ALGO_STREAM, "stream")):
    for mode in ("sync-each", "back-to-back"):
        for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=algo, fetch=False)
        sel.timing(True)
        t0 = time.perf_counter()
        n = 60
        if mode == "sync-each":
            for _ in range(n): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=algo, fetch=False)
        else:
            for _ in range(n): sel.run_async(cfg.tau, pkg.MODE_SMH, r, b, algo=algo)
            sel.finish()
        dt = (time.perf_counter() - t0) / n
        print(name, mode, "wall/pass=%.1f us" % (dt * 1e6), " ".join("%s=%.1f" % (k, sel.kernel_ms(k) * 1e3) for k in ("sigbuild", "join", "verify", "stage1", "hist", "select", "total")))
        sel.timing(False)
print("clocks after load:", smi())
