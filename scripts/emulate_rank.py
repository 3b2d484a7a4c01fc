#!/usr/bin/env python3
"""development: per-rank step time of the 8-GPU weak-scaled workload, emulated on one GPU (each rank's share in turn)"""
import sys, time, math
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
import cuda_selection_criteria_amd as pkg
from cuda_selection_criteria_amd import distributed as D
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
base = pkg.SYNTH_CONFIGS["cfg3"]
n = int(round(base.n_genomes * math.sqrt(world) / base.cluster_size)) * base.cluster_size
cfg = base.scaled(n)
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
def timeit(rows, label):
    for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, rows=rows, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 20
    for _ in range(k): sel.run(cfg.tau, pkg.MODE_SMH, r, b, rows=rows, fetch=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    st = sel.stats()
    print(label, "ms/step=%.3f" % (dt * 1e3), "pairs=%d surv=%d" % (st["evaluated"], st["survivors"]), flush=True)
    return dt
print("N =", n, "world =", world)
worst = 0
for part in range(world):
    sel.set_row_interleave(128, world, part)
    worst = max(worst, timeit((0, n), f"interleaved part {part}"))
sel.set_row_interleave(0, 1, 0)
bounds = D.shard_rows(n, world)
worst_c = 0
for part in (0, world - 1):
    worst_c = max(worst_c, timeit((int(bounds[part]), int(bounds[part + 1])), f"contiguous part {part}"))
print("worst interleaved %.3f ms, worst contiguous %.3f ms" % (worst * 1e3, worst_c * 1e3))
