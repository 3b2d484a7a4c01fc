#!/usr/bin/env python3
"""development (GPU box): rows per interleave block (128 / 256 / 512) against the step time of EVERY part of the weak-scaled workload, twice.
usage: il_block_parts.py <world>"""
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
for rep in range(2):
    for blk in (128, 256, 512):
        ts = []
        for part in range(world):
            sel.set_row_interleave(blk, world, part)
            for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 30 * 1e3)
        print("world", world, "block", blk, "worst %.4f mean %.4f" % (max(ts), sum(ts) / len(ts)), " ".join("%.3f" % t for t in ts), flush=True)
