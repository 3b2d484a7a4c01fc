#!/usr/bin/env python3
"""development sweep of the signature join (GPU box): query side (SGPR / DPP), waves per block, groups per wave, tile height.
   usage: join_sweep.py [workload ...]    prints one line per variant; every variant must report the same stats"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg

VARIANTS = [  # (join_bits, join_q: 1 = LDS query tile / 0 = DPP broadcast, join_wpb, join_qt)
    (16, 0, 1, 128), (16, 1, 4, 64), (16, 1, 4, 128), (15, 1, 4, 64), (15, 1, 8, 64), (15, 1, 4, 96), (15, 1, 4, 128), (15, 1, 8, 128), (15, 1, 4, 256),
]
for wl in sys.argv[1:] or ("cfg3", "cfg4"):
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, _ = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    ref = None
    for bits, q, wpb, qt in VARIANTS:
        sel.set_param("join_bits", bits); sel.set_param("join_q", q); sel.set_param("join_wpb", wpb); sel.set_param("join_qt", qt)
        for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
        st = sel.stats()
        st = {k: v for k, v in st.items() if k != "candidates"} if False else st
        if ref is None: ref = st
        sel.timing(True)
        for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
        print(wl, "bits=%d q=%d wpb=%d qt=%d" % (bits, q, wpb, qt), "sigbuild=%.1f join=%.1f verify=%.1f total=%.1f us" % tuple(sel.kernel_ms(k) * 1e3 for k in ("sigbuild", "join", "verify", "total")),
              "OK" if st == ref else "MISMATCH %s vs %s" % (st, ref), flush=True)
        sel.timing(False)
    sel.close()
    del hll, aux, cards
