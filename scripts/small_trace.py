#!/usr/bin/env python3
"""development (GPU box; SELHIP_LIB = a library built with -DSELHIP_JOIN_TRACE): per-block phase stamps of ONE small_pass_kernel launch
(wall clock, 100 MHz): 0 start, 1 phase 0 done, 2 past the grid barrier, 3 join + verification done, 4 histograms of the first batch done,
5 estimator of the last batch done, 6 end"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import cuda_selection_criteria_amd as pkg
cfg = pkg.SYNTH_CONFIGS["cfg2"]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
lib = pkg.hip_lib()
fn = lib.selhip_debug_join_trace
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
assert sel.get_param("small_pass_used") == 1
assert fn(None, 1) == 0
sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
buf = np.zeros((1 << 17, 4), dtype=np.uint64)
assert fn(buf.ctypes.data, 0) == 0
st = buf[:768].reshape(256, 12).astype(np.float64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["start", "phase0 done", "past barrier", "join+verify done", "hist batch done", "estimator done", "end", "rows parked", "flagged (own)", "flagged (block)"]
for k, nm in enumerate(names):
    col = us[:, k]; col = col[st[:, k] > 0]
    if len(col): print("%-18s min %6.1f  median %6.1f  max %6.1f us  (%d blocks)" % (nm, col.min(), np.median(col), col.max(), len(col)))
sel.close()
