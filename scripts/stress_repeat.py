#!/usr/bin/env python3
"""development (GPU box): thousands of passes on the same inputs in every grouping / lane configuration: the counters must be identical
pass after pass and the fetched result must not change (looks for races between the counter sets, the lanes' streams, the label atomics)"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import cuda_selection_criteria_amd as pkg
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for wl, n in (("cfg3", None), ("cfg5", 12000)):
    cfg = pkg.SYNTH_CONFIGS[wl]
    if n: cfg = cfg.scaled(n)
    hll, aux, cards, _, ah = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    if cfg.p_aux:
        sel.attach_aux_hll(ah, cfg.p_aux); sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
    ref = None
    for chunks, label in ((0, 0), (2, 1), (3, 1), (0, 1), (2, 0)):
        sel.set_pipeline(chunks); sel.set_param("group_label", label)
        first = sel.run(cfg.tau, pkg.MODE_CB_SMH, r, b)
        st0 = sel.stats()
        if ref is None: ref = (first.copy(), st0)
        assert np.array_equal(first, ref[0]) and st0 == ref[1], (chunks, label)
        for it in range(passes):
            sel.run(cfg.tau, pkg.MODE_CB_SMH, r, b, fetch=False)
            assert sel.stats() == st0 and sel.last_attempts() == 1, (chunks, label, it, sel.stats(), st0)
        again = sel.run(cfg.tau, pkg.MODE_CB_SMH, r, b)
        assert np.array_equal(again, ref[0])
        print(wl, "chunks", chunks, "label", label, "ok:", passes, "passes,", st0, flush=True)
    sel.close(); del hll, aux, cards, ah
# the one-launch pass of a small set (barrier words put back by the kernel itself, barrier top word in the double-buffered counters),
# alternating with the regular chain and with the dense walk of stage 2a forced on
cfg = pkg.SYNTH_CONFIGS["cfg2"]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
ref = None
for small, dense, mode in ((1, 32, pkg.MODE_SMH), (0, 0, pkg.MODE_SMH), (1, 32, pkg.MODE_CB_SMH), (2, 32, pkg.MODE_SMH)):
    sel.set_param("small_pass", small); sel.set_param("hist_dense_degree", dense); sel.set_param("group_min_n", 0 if dense == 0 else 2048)
    first = sel.run(cfg.tau, mode, r, b)
    st0 = sel.stats()
    assert sel.get_param("small_pass_used") == (1 if small else 0)
    if ref is None: ref = first.copy()
    assert np.array_equal(first, ref)          # (cfg2 is the flat set: CB prunes nothing)
    for it in range(passes if small != 2 else passes // 10):
        if it % 7 == 3: sel.run(cfg.tau, mode, r, b, rows=(100, 900), fetch=False)     # another shape of pass in between
        sel.run(cfg.tau, mode, r, b, fetch=False)
        assert sel.stats() == st0 and sel.last_attempts() == 1, (small, dense, it, sel.stats(), st0)
    assert np.array_equal(sel.run(cfg.tau, mode, r, b), ref)
    print("cfg2 small_pass", small, "dense", dense, "ok:", passes, "passes,", st0, flush=True)
sel.close()
print("stress ok")
