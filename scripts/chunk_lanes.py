#!/usr/bin/env python3
"""development (GPU box): step time of a pass cut into row chunks that run as whole chains on two internal streams
(selhip_ctx_set_pipeline), against the plain pass -- whole workloads and one rank's interleaved share of a strong-scaled one.
   chunk_lanes.py [workload[:world] ...]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import cuda_selection_criteria_amd as pkg
for spec in sys.argv[1:] or ("cfg3", "cfg3:8", "cfg4", "cfg4:4", "cfg4:8", "cfg5", "cfg5:8"):
    wl, _, world = spec.partition(":")
    world = int(world or 1)
    cfg = pkg.SYNTH_CONFIGS[wl]
    hll, aux, cards, _, ah = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    sel = pkg.Selector(0); sel.attach(hll, aux, cards)
    if cfg.p_aux:
        sel.attach_aux_hll(ah, cfg.p_aux); sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
    if world > 1: sel.set_row_interleave(128, world, world // 2)
    ref = None
    out = []
    for chunks in (0, 2, 3, 4, 6, 8):
        sel.set_pipeline(chunks)
        for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
        cnt = sel.stats()
        if ref is None: ref = cnt
        assert cnt == ref, (cnt, ref)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
        torch.cuda.synchronize()
        out.append("%d: %.3f" % (chunks, (time.perf_counter() - t0) / 20 * 1e3))
    print(spec, "ms per pass by chunk count ->", "  ".join(out), flush=True)
    sel.close()
    del hll, aux, cards, ah
