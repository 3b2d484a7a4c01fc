#!/usr/bin/env python3
"""development (GPU box): how well do two half-passes overlap when they run on two streams?  A rank's share of a strong-scaled
workload (part p of `world`) is cut in two (parts p and p + world of 2*world); both halves are run back to back on ONE stream and
concurrently on TWO streams (two contexts).  Upper bound of what pipelining the stages of a pass over row chunks could give.
   overlap_probe.py [workload] [world]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
import cuda_selection_criteria_amd as pkg
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = pkg.SYNTH_CONFIGS[wl]
hll, aux, cards, _, ah = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sels = []
for st in (s1, s2):
    s = pkg.Selector(0, stream=st.cuda_stream); s.attach(hll, aux, cards)
    if cfg.p_aux:
        s.attach_aux_hll(ah, cfg.p_aux); s.set_criterion(pkg.CRIT_HLL_A_SMH_A)
    sels.append(s)
whole = pkg.Selector(0, stream=s1.cuda_stream); whole.attach(hll, aux, cards)
if cfg.p_aux:
    whole.attach_aux_hll(ah, cfg.p_aux); whole.set_criterion(pkg.CRIT_HLL_A_SMH_A)
p = world // 2
def run(fn, k=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
whole.set_row_interleave(128, world, p)
t_whole = run(lambda: whole.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False))
sels[0].set_row_interleave(128, 2 * world, p); sels[1].set_row_interleave(128, 2 * world, p + world)
def seq():
    sels[0].run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False); sels[1].run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
def conc():
    sels[0].run_async(cfg.tau, pkg.MODE_SMH, r, b); sels[1].run_async(cfg.tau, pkg.MODE_SMH, r, b)
    sels[0].finish(); sels[1].finish()
t_seq, t_conc = run(seq), run(conc)
print(f"{wl} world {world} part {p}: whole share {t_whole:.3f} ms; two halves in sequence {t_seq:.3f} ms; two halves on two streams {t_conc:.3f} ms")
