#!/usr/bin/env python3
"""development sweep of the signature-join launch knobs (GPU box)"""
import itertools, os, subprocess, sys, json
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
code = r'''
import sys, os
sys.path.insert(0, %r)
import cuda_selection_criteria_amd as pkg
cfg = pkg.SYNTH_CONFIGS[os.environ.get("WL","cfg3")]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
sel.timing(True)
for _ in range(10): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
print("RES", " ".join("%%s=%%.1f" %% (k, sel.kernel_ms(k)*1e3) for k in ("sigbuild","join","verify","hist","select","total")), sel.stats())
''' % str(ROOT)
for qt, gpw, acc in itertools.product((32, 64, 128, 256, 512), (1,), (1,)):
    env = dict(os.environ, SELHIP_JOIN_QT=str(qt), SELHIP_JOIN_GPW=str(gpw), SELHIP_JOIN_ACC=str(acc))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("RES")]
    print(f"qt={qt} gpw={gpw} acc={acc}", line[0] if line else r.stderr[-300:], flush=True)
