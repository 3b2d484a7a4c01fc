"""Seeded random configurations (sizes around the 64-lane / 256-candidate / 16-query tile edges, every criterion,
both modes, every stage-1 algorithm that accepts the shape, random row sub-ranges, cards with zeros and ties):
the HIP path must equal the oracle bit for bit every time."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cuda_selection_criteria_amd as pkg  # noqa: E402
from cuda_selection_criteria_amd import ALGO_AUTO, ALGO_HASHJOIN, ALGO_SIG, ALGO_STREAM, MODE_CB_SMH, MODE_SMH, Selector  # noqa: E402
from cuda_selection_criteria_amd.synth import SynthConfig  # noqa: E402


def same(got, want):
    return (len(got) == len(want) and np.array_equal(got["i"], want["i"]) and np.array_equal(got["k"], want["k"])
            and np.array_equal(got["jaccard"].view(np.uint64), want["jacc"].view(np.uint64)))


@pytest.mark.parametrize("seed", range(24))
def test_random_configuration(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 129, 255, 256, 257, 300, 511, 513]))
    m = int(rng.choice([4, 16, 64, 128, 256, 512, 1024]))
    tau = float(rng.choice([0.3, 0.5, 0.8, 0.9, 0.95]))
    cluster = int(rng.choice([1, 2, 5, 10]))
    cfg = SynthConfig(f"rnd{seed}", n, m, tau, 7000 + seed, p_aux=int(rng.choice([4, 6, 8])), cluster_size=cluster,
                      mode=int(rng.integers(0, 2)), n_sh_lo=int(rng.choice([300, 2000, 8000])), n_sh_hi=40000)
    hll, aux, aux_hll = pkg.synth_host(cfg, threads=4)
    # duplicates (ties in cardinality, identical sketches) and empty genomes (cardinality 0)
    if n > 4:
        hll[1], aux[1], aux_hll[1] = hll[0], aux[0], aux_hll[0]
        if seed % 3 == 0:
            hll[2] = 0
            hll[3] = 0
    fp = int(seed % 2)
    oracle.set_fma(fp)
    try:
        cards = oracle.cards(hll)
        perm = pkg.sort_by_card(cards)
        hll, aux, aux_hll, cards = hll[perm], aux[perm], aux_hll[perm], cards[perm]
        r, b = pkg.banding(m, tau)
        with Selector(0, fp) as sel:
            sel.upload(hll, aux, cards)
            sel.upload_aux_hll(aux_hll, cfg.p_aux)
            for crit in (pkg.CRIT_SMH_A, pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN, pkg.CRIT_HLL_A_SMH_A):
                sel.set_criterion(crit)
                for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                    want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb, criterion=crit, aux_hll=aux_hll,
                                             p_aux=cfg.p_aux, threads=4)
                    algos = [ALGO_AUTO]
                    if crit in (pkg.CRIT_SMH_A, pkg.CRIT_HLL_A_SMH_A):
                        algos.append(ALGO_STREAM)
                        algos.append(ALGO_HASHJOIN)
                        if b in (8, 16, 32, 64, 128):
                            algos.append(ALGO_SIG)
                    for algo in algos:
                        got = sel.run(tau, mode, r, b, algo=algo)
                        assert same(got, want), (seed, n, m, tau, crit, mode, algo, len(got), len(want))
                        s = sel.stats()
                        assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (seed, crit, mode, algo, s, st)
                    # a random row sub-range returns exactly the rows it was asked for
                    lo = int(rng.integers(0, n))
                    hi_ = int(rng.integers(lo, n + 1))
                    part = sel.run(tau, mode, r, b, rows=(lo, hi_))
                    sub = want[(want["i"] >= lo) & (want["i"] < hi_)]
                    assert same(part, sub), (seed, crit, mode, lo, hi_)
    finally:
        oracle.set_fma(1)
