#!/usr/bin/env python3
"""GPU box: a few passes of one signature-join variant (for rocprofv3 --pmc):  join_probe.py WORKLOAD Q WPB QT"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
wl = sys.argv[1]; q, wpb, qt = (int(x) for x in sys.argv[2:5])
cfg = pkg.SYNTH_CONFIGS[wl]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
sel.set_param("join_q", q); sel.set_param("join_wpb", wpb); sel.set_param("join_qt", qt)
for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
print(sel.stats())
sel.close()
