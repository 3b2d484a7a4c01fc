#!/usr/bin/env python3
"""GPU box: the out-of-core driver (selhip_ooc_select: sketches in host memory, block pairs uploaded in turn) against the
in-core pass on the same set; prints one JSON line.  usage: bench_ooc.py [workload] [block_genomes ...]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import cuda_selection_criteria_amd as pkg  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
blocks = [int(x) for x in sys.argv[2:]] or [12500, 25000]
cfg = pkg.SYNTH_CONFIGS[wl]
hll_t, aux_t, cards_t, _, _ = pkg.synth_device(cfg)
hll, aux, cards = hll_t.cpu().numpy(), aux_t.cpu().numpy().view(np.uint64), cards_t.cpu().numpy()
del hll_t, aux_t, cards_t
r, b = pkg.banding(cfg.m, cfg.tau)
out = {"workload": wl, "n_genomes": cfg.n_genomes, "m": cfg.m, "host_bytes": int(hll.nbytes + aux.nbytes), "runs": []}
with pkg.Selector(0) as sel:
    t0 = time.perf_counter()
    sel.upload(hll, aux, cards)
    want = sel.run(cfg.tau, pkg.MODE_SMH, r, b)
    t_in = time.perf_counter() - t0
    t0 = time.perf_counter()
    sel.run(cfg.tau, pkg.MODE_SMH, r, b, fetch=False)
    t_pass = time.perf_counter() - t0
    st = sel.stats()
out["in_core"] = {"upload_plus_pass_s": t_in, "resident_pass_s": t_pass, "pairs": st["evaluated"], "selected": len(want)}
for blk in blocks:
    for streams in (1, 2):
        t0 = time.perf_counter()
        got, s = pkg.ooc_select(hll, aux, cards, cfg.tau, blk, pkg.MODE_SMH, r, b, n_streams=streams)
        dt = time.perf_counter() - t0
        same = len(got) == len(want) and np.array_equal(got["i"], want["i"]) and np.array_equal(got["k"], want["k"]) and \
            np.array_equal(got["jaccard"].view(np.uint64), want["jaccard"].view(np.uint64))
        nb = -(-cfg.n_genomes // blk)
        out["runs"].append({"block_genomes": blk, "blocks": nb, "tiles": nb * (nb + 1) // 2, "streams": streams, "seconds": dt,
                            "pairs_per_s": s["evaluated"] / dt, "identical_to_in_core": bool(same),
                            "device_resident_bytes": int(streams * 2 * blk * (16384 + 8 * cfg.m))})
print(json.dumps(out))
