#!/usr/bin/env python3
"""development (GPU box): kernel breakdown of one rank's share of the weak-scaled workload -- first and last part of the row interleave
against the first and last contiguous equal-pair shard.  usage: il_breakdown.py <world>"""
import sys, math, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import cuda_selection_criteria_amd as pkg
from cuda_selection_criteria_amd import distributed as D
world = int(sys.argv[1])
base = pkg.SYNTH_CONFIGS["cfg3"]
n = int(round(base.n_genomes * math.sqrt(world) / base.cluster_size)) * base.cluster_size
cfg = base.scaled(n)
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
for kv in sys.argv[2:]:                                   # name=value ... -> selhip_ctx_set_param
    name, _, val = kv.partition("="); sel.set_param(name, int(val)); print("param", name, val)
def brk(rows, label):
    for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, rows=rows, fetch=False)
    sel.timing(1)
    for _ in range(5): sel.run(cfg.tau, pkg.MODE_SMH, r, b, rows=rows, fetch=False)
    print(label, {k: round(sel.kernel_ms(k) * 1e3, 1) for k in ("sigbuild", "join", "verify", "group", "hist", "select", "total") if sel.kernel_ms(k) > 0}, "tile rows", sel.get_param("join_tile_rows"), flush=True)
    sel.timing(0)
sel.set_row_interleave(128, world, 0); brk((0, n), "interleaved part 0")
sel.set_row_interleave(128, world, world - 1); brk((0, n), "interleaved last part")
sel.set_row_interleave(0, 1, 0)
bounds = D.shard_rows(n, world)
brk((int(bounds[0]), int(bounds[1])), "contiguous part 0")
brk((int(bounds[world - 1]), int(bounds[world])), "contiguous last part")
