#!/usr/bin/env python3
"""development (GPU box): rows per interleave block against the step and join time of one rank's share of the weak-scaled workload.
usage: il_block_sweep.py <world>"""
import sys, math, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import cuda_selection_criteria_amd as pkg
world = int(sys.argv[1])
base = pkg.SYNTH_CONFIGS["cfg3"]
n = int(round(base.n_genomes * math.sqrt(world) / base.cluster_size)) * base.cluster_size
cfg = base.scaled(n)
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
for blk in (32, 64, 128, 256, 512, 1024, 2048):
    worst, js = 0, []
    for part in sorted({0, world // 2, world - 1}):
        sel.set_row_interleave(blk, world, part)
        for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        sel.timing(1)
        for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
        js.append((round(sel.kernel_ms("join") * 1e3, 1), round(sel.kernel_ms("hist") * 1e3, 1), sel.stats()["evaluated"]))
        sel.timing(0)
        worst = max(worst, dt)
    print("world", world, "block", blk, "tile rows", sel.get_param("join_tile_rows"), "worst of 3 parts %.4f ms" % (worst * 1e3), "join/hist/pairs", js, flush=True)
