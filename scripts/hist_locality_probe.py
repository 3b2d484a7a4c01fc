#!/usr/bin/env python3
"""development (GPU box): what does stage 2a cost when the pairs that share HLL rows are neighbours in the grouped list?
The same synthetic set is run (a) in ascending-cardinality order (clusters scattered over the rank order) and (b) in generation
order (every cluster contiguous) with constant stand-in cardinalities (ascending order is then trivially met; the Jaccard values are
meaningless, the kernels' work is the same)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import cuda_selection_criteria_amd as pkg
for wl in sys.argv[1:] or ("cfg3", "cfg4", "cfg5"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    r, b = pkg.banding(cfg.m, cfg.tau)
    for order in ("by cardinality", "clusters contiguous"):
        hll, aux, cards, _, ah = pkg.synth_device(cfg, sort=(order == "by cardinality"))
        if order != "by cardinality": cards = torch.full_like(cards, 1.0e5)
        sel = pkg.Selector(0); sel.attach(hll, aux, cards)
        if cfg.p_aux:
            sel.attach_aux_hll(ah, cfg.p_aux); sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
        sel.set_pipeline(0)
        for run in (1, 4):
            sel.set_param("hist_run", run)
            for _ in range(3): sel.run(0.0 if order != "by cardinality" else cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
            sel.timing(1)
            for _ in range(8): sel.run(0.0 if order != "by cardinality" else cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
            print(wl, order, "hist_run=%d" % run, sel.stats(), {k: round(sel.kernel_ms(k) * 1e3, 1) for k in ("join", "verify", "aux", "group", "hist", "select", "total") if sel.kernel_ms(k) > 0}, flush=True)
            sel.timing(0)
        sel.close()
        del hll, aux, cards, ah
