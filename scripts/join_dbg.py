#!/usr/bin/env python3
"""temporary: where does the per-tile overhead of the signature join go?  dbg bit0 = no candidate loads, bit1 = no compare loop"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
for wl in sys.argv[1:] or ("cfg4",):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, _ = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    for qt in (64, 128, 256):
        for wpb in (4, 8):
            for dbg in (0, 1, 2, 3):
                sel.set_param("join_q", 1); sel.set_param("join_wpb", wpb); sel.set_param("join_t", 1); sel.set_param("join_qt", qt); sel.set_param("join_dbg", dbg)
                for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
                sel.timing(True)
                for _ in range(4): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
                print(wl, "qt=%d wpb=%d dbg=%d join=%.1f us" % (qt, wpb, dbg, sel.kernel_ms("join") * 1e3), flush=True)
                sel.timing(False)
    sel.close()
