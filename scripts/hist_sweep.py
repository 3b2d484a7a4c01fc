#!/usr/bin/env python3
"""development sweep of stage 2a (GPU box): grid size ("hist_blocks"), resident waves per CU ("hist_pad": extra LDS per one-wave
block: 0 -> 10 per CU, 4096 -> 8, 16384 -> 5, 24576 -> 4), task shape ("hist_run")"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg
for wl in sys.argv[1:] or ("cfg3", "cfg4"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, _ = pkg.synth_device(cfg)
    rows, bands = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    for blocks, pad, run in [(16384, 0, 1), (16384, 4096, 1), (16384, 10240, 1), (16384, 16384, 1), (16384, 24576, 1), (4096, 0, 1), (2048, 0, 1), (2560, 0, 1),
                             (1280, 16384, 1), (1024, 24576, 1), (16384, 0, 2), (8192, 16384, 1)]:
        sel.set_param("hist_run", run); sel.set_param("hist_blocks", blocks); sel.set_param("hist_pad", pad)
        for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, rows, bands, algo=pkg.ALGO_SIG, fetch=False)
        sel.timing(True)
        for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, rows, bands, algo=pkg.ALGO_SIG, fetch=False)
        print(wl, "blocks=%d pad=%d run=%d" % (blocks, pad, run), "hist=%.1f us total=%.1f us" % (sel.kernel_ms("hist") * 1e3, sel.kernel_ms("total") * 1e3), flush=True)
        sel.timing(False)
    sel.close()
    del hll, aux, cards
