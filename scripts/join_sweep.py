#!/usr/bin/env python3
"""development sweep of the signature join (GPU box): selhip_ctx_set_param("join_bits" / "join_qt")"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
for wl in sys.argv[1:] or ("cfg3", "cfg4"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, _ = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    for bits, db in ((16, 4), (16, 1)):
        sel.set_param("join_bits", bits); sel.set_param("join_wpb", db)
        for qt in (64, 96, 128):
            sel.set_param("join_qt", qt)
            for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
            st = sel.stats()
            sel.timing(True)
            for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
            print(wl, "bits=%d db=%d qt=%d" % (bits, db, qt), "sigbuild=%.1f join=%.1f verify=%.1f total=%.1f us" % tuple(sel.kernel_ms(k) * 1e3 for k in ("sigbuild", "join", "verify", "total")), "stats", st, flush=True)
            sel.timing(False)
    sel.close()
    del hll, aux, cards
