#!/usr/bin/env python3
"""development sweep of stage 2a's task shape (GPU box): selhip_ctx_set_param("hist_run" / "hist_blocks")"""
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
    for blocks, run, dbg in [(b, r, g) for b in (2048, 8192, 10240, 16384) for r in (1,) for g in (1, 0)]:
        if True:
            sel.set_stage2_grouping(bool(dbg))
            sel.set_param("hist_run", run); sel.set_param("hist_blocks", blocks)
            for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
            sel.timing(True)
            for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
            print(wl, "blocks=%d run=%d grouping=%d" % (blocks, run, dbg), "hist=%.1f us total=%.1f us" % (sel.kernel_ms("hist") * 1e3, sel.kernel_ms("total") * 1e3), flush=True)
            sel.timing(False)
    sel.close()
    del hll, aux, cards
