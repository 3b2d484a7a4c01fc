#!/usr/bin/env python3
"""development (GPU box, under rocprofv3 --pmc): a few passes of ONE interleaved share of the weak-scaled workload.  usage: share_passes.py <world> <part>"""
import sys, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import cuda_selection_criteria_amd as pkg
world, part = int(sys.argv[1]), int(sys.argv[2])
base = pkg.SYNTH_CONFIGS["cfg3"]
n = int(round(base.n_genomes * math.sqrt(world) / base.cluster_size)) * base.cluster_size
cfg = base.scaled(n)
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
if world > 1: sel.set_row_interleave(128, world, part)
for _ in range(6): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
print(sel.stats())
