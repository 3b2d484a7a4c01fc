#!/usr/bin/env python3
"""development (GPU box): hll_a / hll_an as FIRST criterion (the reference's `-c hll_a` / `-c hll_an`): the whole pair space goes through
enum_pairs_kernel + aux_fused_kernel; pairs/s and the kernel breakdown"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import cuda_selection_criteria_amd as pkg
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cfg = pkg.SYNTH_CONFIGS["cfg5"].scaled(n)
hll, aux, cards, _, ah = pkg.synth_device(cfg)
sel = pkg.Selector(0); sel.attach(hll, aux, cards); sel.attach_aux_hll(ah, cfg.p_aux)
for crit, name in ((pkg.CRIT_HLL_A, "hll_a"), (pkg.CRIT_HLL_AN, "hll_an")):
    sel.set_criterion(crit)
    for mode, mname in ((pkg.MODE_SMH, "all pairs"), (pkg.MODE_CB_SMH, "CB")):
        for _ in range(2): sel.run(cfg.tau, mode, 1, 1, fetch=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = 5
        for _ in range(k): sel.run(cfg.tau, mode, 1, 1, fetch=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
        st = sel.stats()
        sel.timing(1)
        for _ in range(3): sel.run(cfg.tau, mode, 1, 1, fetch=False)
        ks = {x: round(sel.kernel_ms(x), 3) for x in ("stage1", "aux", "group", "hist", "select", "total") if sel.kernel_ms(x) > 0}
        sel.timing(0)
        print(f"N={n} {name} {mname}: {dt*1e3:.3f} ms/pass, {st['evaluated']/dt:.4g} pairs/s, stats {st}, kernels ms {ks}", flush=True)
