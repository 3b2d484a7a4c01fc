#!/usr/bin/env python3
"""development: per-rank step time of a STRONG-scaled workload (cfg4 / cfg5 sharded over `world` ranks), emulated on one GPU by
running each rank's interleaved share in turn; prints the slowest rank and the implied scaling efficiency against the 1-rank step"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import cuda_selection_criteria_amd as pkg
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = [a for a in sys.argv[1:] if a.startswith("--")]
wl = args[0] if args else "cfg4"
worlds = [int(x) for x in args[1:]] or [2, 4, 8]
cfg = pkg.SYNTH_CONFIGS[wl]
hll, aux, cards, _, ah = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
if cfg.p_aux:
    sel.attach_aux_hll(ah, cfg.p_aux); sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
n = cfg.n_genomes
IL_BLOCK = 128
for f in flags:
    if f.startswith("--block="): IL_BLOCK = int(f.split("=")[1]); print(wl, "interleave block", IL_BLOCK, flush=True)
if "--sig-cache" in flags:
    sel.set_param("sig_cache", 1)          # every rank of a strong-scaled job runs many passes over the same replica: signatures built once
    print(wl, "signature cache ON (selhip_ctx_set_param sig_cache=1)", flush=True)
for f in flags:
    if f.startswith("--param="):
        name, _, val = f[len("--param="):].partition("=")
        sel.set_param(name, int(val)); print(wl, "param", name, "=", val, flush=True)
    if f.startswith("--pipeline="):
        sel.set_pipeline(int(f.split("=")[1])); print(wl, "pipeline", f.split("=")[1], flush=True)
def timeit(k=20):
    for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k
one = timeit()
print(wl, "1 rank: %.3f ms" % (one * 1e3), flush=True)
for world in worlds:
    worst = 0
    for part in range(world):
        sel.set_row_interleave(IL_BLOCK, world, part)
        worst = max(worst, timeit())
    sel.set_row_interleave(0, 1, 0)
    print(wl, "world %d: slowest rank %.3f ms -> speed-up %.2f (efficiency %.0f %%)" % (world, worst * 1e3, one / worst, 100 * one / worst / world), flush=True)
# kernel breakdown of one rank's share at the largest world size
world = worlds[-1]
sel.set_row_interleave(IL_BLOCK, world, world // 2)
for _ in range(2): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
sel.timing(1)
for _ in range(5): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
print(wl, "world %d part %d kernels (us):" % (world, world // 2), {k: round(sel.kernel_ms(k) * 1e3, 1) for k in ("sigbuild", "join", "verify", "aux", "group", "hist", "select", "total") if sel.kernel_ms(k) > 0}, flush=True)
sel.timing(0)
