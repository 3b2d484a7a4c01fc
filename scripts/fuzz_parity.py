#!/usr/bin/env python3
"""development (GPU box): randomized differential run of the product against the oracle -- random small sets (genome count, sketch width,
cluster shape, degenerate fraction, empty sketches, duplicates), random thresholds / band shapes / modes / row ranges / candidate starts,
and every internal route drawn at random (one-launch pass, grouping, label order, dense walk, chunk lanes, histogram kernel, join width,
interleave part).  Pairs, Jaccard bits and the evaluated / survivor counters must match.  usage: fuzz_parity.py [seconds] [seed]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import cuda_selection_criteria_amd as pkg
from cuda_selection_criteria_amd.synth import SynthConfig
import oracle_py

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = oracle_py.Oracle()
t_end = time.time() + budget
n_sets = n_runs = 0
while time.time() < t_end:
    n = int(rng.choice([2, 3, 17, 64, 65, 130, 257, 600, 1025, 2048, 2049, 2500]))
    m = int(rng.choice([64, 128, 256, 512]))
    with_aux = rng.random() < 0.35 and n <= 700           # (hll_a / hll_an as FIRST criterion run the estimator on every pair)
    cfg = SynthConfig("fuzz", n, m, 0.8, int(rng.integers(1, 1 << 30)), p_aux=8 if with_aux else 0, cluster_size=int(rng.choice([2, 5, 10, 40])),
                      mode=int(rng.integers(0, 2)), n_sh_lo=2000, n_sh_hi=int(rng.choice([2000, 20000])))
    hll, aux, aux_hll = pkg.synth_host(cfg)
    if rng.random() < 0.4: pkg.harden(aux, frac=float(rng.choice([0.1, 0.5, 1.0])) if n < 1000 else 0.1, seed=int(rng.integers(1, 1 << 30)))
    if rng.random() < 0.3 and n > 8: hll[: int(rng.integers(1, 4))] = 0
    if rng.random() < 0.3 and n > 8:
        j = int(rng.integers(1, n)); hll[j] = hll[j - 1]; aux[j] = aux[j - 1]
    cards = orc.cards(hll)
    perm = pkg.sort_by_card(cards)
    hll, aux, cards = hll[perm], aux[perm], cards[perm]
    if with_aux: aux_hll = aux_hll[perm]
    n_sets += 1
    with pkg.Selector(0) as sel:
        sel.set_param("init_cap", int(rng.choice([1024, 1 << 20])))
        sel.upload(hll, aux, cards)
        if with_aux: sel.upload_aux_hll(aux_hll, 8)
        for _ in range(6):
            tau = float(rng.choice([0.3, 0.5, 0.8, 0.9, 0.95]))
            if rng.random() < 0.6: r, b = pkg.banding(m, tau)
            else:
                r = int(rng.choice([2, 4, 8, 16, 32])); b = m // r
            use_cb = bool(rng.integers(0, 2))
            params = {"small_pass": int(rng.choice([-1, 0, 1])), "group_min_n": int(rng.choice([0, 2048])), "group_label": int(rng.choice([-1, 0, 1])),
                      "hist_dense_degree": int(rng.choice([-1, 0, 32])), "hist_run": int(rng.choice([0, 1, 4, 64, 100])),
                      "join_bits": int(rng.choice([16, 16, 15, 32])), "sig_cache": int(rng.integers(0, 2)), "hist_bs_blocks": int(rng.choice([8, 64, 2048]))}
            for k, v in params.items(): sel.set_param(k, v)
            sel.set_pipeline(int(rng.choice([-1, 0, 2, 3])))
            crit = int(rng.choice([pkg.CRIT_SMH_A, pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN, pkg.CRIT_HLL_A_SMH_A])) if with_aux else pkg.CRIT_SMH_A
            sel.set_criterion(crit)
            params["crit"] = crit
            want, st = orc.select(hll, aux, cards, tau, r, b, use_cb=use_cb, criterion=crit, aux_hll=aux_hll if with_aux else None, p_aux=8)
            mode = pkg.MODE_CB_SMH if use_cb else pkg.MODE_SMH
            kind = rng.integers(0, 4)
            ctx = (n, m, cfg.seed, tau, r, b, use_cb, params, int(kind))
            if kind == 0 or n < 8:
                got = sel.run(tau, mode, r, b)
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (ctx, s, st)
                w = want
            elif kind == 1:
                lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
                got = sel.run(tau, mode, r, b, rows=(lo, hi))
                w = want[(want["i"] >= lo) & (want["i"] < hi)]
            elif kind == 2:
                parts = int(rng.choice([2, 3, 8])); blk = int(rng.choice([32, 64, 128]))
                gots = []
                for part in range(parts):
                    sel.set_row_interleave(blk, parts, part)
                    gots.append(sel.run(tau, mode, r, b))
                sel.set_row_interleave(0, 1, 0)
                got = np.concatenate(gots); got = got[np.lexsort((got["k"], got["i"]))]
                w = want
            else:
                cb = int(rng.integers(1, n))
                sel.set_candidate_begin(cb)
                got = sel.run(tau, mode, r, b, rows=(0, cb))
                sel.set_candidate_begin(0)
                w = want[(want["i"] < cb) & (want["k"] >= cb)]
            assert len(got) == len(w) and np.array_equal(got["i"], w["i"]) and np.array_equal(got["k"], w["k"]), (ctx, len(got), len(w))
            assert np.array_equal(got["jaccard"].view(np.uint64), w["jacc"].view(np.uint64)), ctx
            n_runs += 1
    if n_sets % 10 == 0: print("sets", n_sets, "runs", n_runs, flush=True)
print("fuzz ok:", n_sets, "sets,", n_runs, "runs, seed", seed)
