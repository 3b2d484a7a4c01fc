#!/usr/bin/env python3
"""development sweep (GPU box) of stage 2a with the label-ordered grouping: one-wave blocks x pairs per task"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
for wl in sys.argv[1:] or ("cfg4", "cfg5"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, ah = pkg.synth_device(cfg)
    rows, bands = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    if cfg.p_aux:
        sel.attach_aux_hll(ah, cfg.p_aux); sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
    sel.set_pipeline(0)
    for blocks in (16384, 8192, 4096, 2560, 2048, 1280):
        for run in (1, 2, 4, 8, 16):
            sel.set_param("hist_blocks", blocks); sel.set_param("hist_run", run)
            for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, rows, bands, fetch=False)
            sel.timing(True)
            for _ in range(5): sel.run(cfg.tau, pkg.MODE_SMH, rows, bands, fetch=False)
            print(wl, "blocks=%d run=%d" % (blocks, run), "hist=%.1f us total=%.1f us" % (sel.kernel_ms("hist") * 1e3, sel.kernel_ms("total") * 1e3), flush=True)
            sel.timing(False)
    sel.close()
    del hll, aux, cards, ah
