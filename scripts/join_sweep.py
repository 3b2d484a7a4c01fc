#!/usr/bin/env python3
"""development sweep of the signature-join tile height (GPU box): selhip_ctx_set_param("join_qt")"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
for wl in ("cfg3", "cfg4"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, _ = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    for qt in (16, 32, 48, 64, 96, 128, 192, 256):
        sel.set_param("join_qt", qt)
        for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
        sel.timing(True)
        for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
        print(wl, "qt=%d" % qt, "join=%.1f us total=%.1f us" % (sel.kernel_ms("join") * 1e3, sel.kernel_ms("total") * 1e3), flush=True)
        sel.timing(False)
    sel.close()
    del hll, aux, cards
