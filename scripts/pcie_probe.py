#!/usr/bin/env python3
"""development: host -> device copy rates on the GPU box (pageable / pinned / registered), the inputs of the upload design"""
import time, ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
dev = torch.device("cuda", 0)
n = 164 * 1000 * 1000
h = np.random.default_rng(0).integers(0, 30, n, dtype=np.uint8)
d = torch.empty(n, dtype=torch.uint8, device=dev)
hip = ctypes.CDLL("libamdhip64.so")
def rate(label, fn, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("%-44s %7.2f ms  %6.1f GB/s" % (label, dt * 1e3, n / dt / 1e9), flush=True)
src = torch.from_numpy(h)
rate("pageable numpy -> device (hipMemcpy)", lambda: d.copy_(src))
t = time.perf_counter(); pin = src.pin_memory(); print("pin_memory() copy of 164 MB: %.2f ms" % ((time.perf_counter() - t) * 1e3))
rate("pinned -> device", lambda: d.copy_(pin, non_blocking=True))
t = time.perf_counter(); rc = hip.hipHostRegister(ctypes.c_void_p(h.ctypes.data), ctypes.c_size_t(n), 0); t1 = time.perf_counter() - t
print("hipHostRegister of 164 MB: rc=%d %.2f ms" % (rc, t1 * 1e3))
rate("registered numpy -> device", lambda: d.copy_(src, non_blocking=True))
t = time.perf_counter(); hip.hipHostUnregister(ctypes.c_void_p(h.ctypes.data)); print("hipHostUnregister: %.2f ms" % ((time.perf_counter() - t) * 1e3))
# chunked: 2 streams
for chunk_mb in (8, 32):
    c = chunk_mb * 1000 * 1000
    def chunked():
        for o in range(0, n, c):
            d[o:o + c].copy_(pin[o:o + c], non_blocking=True)
    rate(f"pinned -> device in {chunk_mb} MB chunks", chunked)
