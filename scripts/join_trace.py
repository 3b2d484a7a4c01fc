#!/usr/bin/env python3
"""development (GPU box; library built with -DSELHIP_JOIN_TRACE and SELHIP_LIB pointing at it): timeline of ONE signature-join launch --
per wave {start, end} wall-clock ticks (100 MHz), rows compared, hardware id; prints busy waves over time, the ramp and the tail.
   SELHIP_LIB=scripts/microbench/libselhip_trace.so python scripts/join_trace.py cfg3 [name=value ...]"""
import ctypes as C, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import cuda_selection_criteria_amd as pkg
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = pkg.SYNTH_CONFIGS[wl]
hll, aux, cards, _, _ = pkg.synth_device(cfg)
r, b = pkg.banding(cfg.m, cfg.tau)
sel = pkg.Selector(0); sel.attach(hll, aux, cards)
for kv in sys.argv[2:]:
    k, _, v = kv.partition("="); sel.set_param(k, int(v))
lib = pkg.hip_lib()
fn = lib.selhip_debug_join_trace
fn.restype = C.c_int; fn.argtypes = [C.c_void_p, C.c_int]
for _ in range(3): sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
assert fn(None, 1) == 0
sel.timing(True)
sel.run(cfg.tau, pkg.MODE_SMH, r, b, algo=pkg.ALGO_SIG, fetch=False)
print("join %.1f us" % (sel.kernel_ms("join") * 1e3))
buf = np.zeros((1 << 17, 4), dtype=np.uint64)
assert fn(buf.ctypes.data, 0) == 0
w = buf[buf[:, 1] > 0]
t0 = w[:, 0].min()
start = (w[:, 0] - t0).astype(np.float64) / 100.0      # us (100 MHz wall clock)
end = (w[:, 1] - t0).astype(np.float64) / 100.0
rows = w[:, 2].astype(np.int64)
print("working waves %d, rows total %d, span %.1f us" % (len(w), rows.sum(), end.max()))
dur = end - start
print("wave duration us: min %.1f median %.1f max %.1f; per row ns: median %.0f" % (dur.min(), np.median(dur), dur.max(), np.median(dur / np.maximum(rows, 1)) * 1e3))
edges = np.arange(0, end.max() + 5, 5.0)
for a in edges[:-1]:
    live = ((start < a + 5) & (end > a)).sum()
    started = ((start >= a) & (start < a + 5)).sum()
    print("t=%5.0f..%5.0f us  live waves %5d  started %5d" % (a, a + 5, live, started))
hw = w[:, 3]
simd = ((hw >> 4) & 3).astype(int); cu = ((hw >> 8) & 15).astype(int); se = ((hw >> 13) & 7).astype(int)
print("distinct (se,cu,simd):", len(set(zip(se.tolist(), cu.tolist(), simd.tolist()))))
sel.close()
