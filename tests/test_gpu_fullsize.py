"""BASELINE.json sizes on the GPU: bit-exact against the oracle where the oracle finishes in seconds (configs 2, 3),
and size-independent properties beyond that (two independent GPU algorithms agree, shards tile the pair space,
planted near-duplicates are found, runs are idempotent, two-stage = intersection of its stages on sampled rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import cuda_selection_criteria_amd as pkg  # noqa: E402
from cuda_selection_criteria_amd import ALGO_HASHJOIN, ALGO_SIG, ALGO_STREAM, MODE_CB_SMH, MODE_SMH, Selector  # noqa: E402


def fetch_np(hll_t, aux_t, cards_t):
    return hll_t.cpu().numpy(), aux_t.cpu().numpy().view(np.uint64), cards_t.cpu().numpy()


def same(got, want):
    return (len(got) == len(want) and np.array_equal(got["i"], want["i"]) and np.array_equal(got["k"], want["k"])
            and np.array_equal(got["jaccard"].view(np.uint64), want["jacc"].view(np.uint64)))


@pytest.mark.parametrize("key", ["cfg2", "cfg3", "cfg3-spread"])
def test_baseline_configs_bit_exact_vs_oracle(oracle, key):
    """configs[1] (1 000 genomes, m=256, tau=0.9) and configs[2] (10 000 genomes, m=512, tau=0.8) at FULL size"""
    cfg = pkg.SYNTH_CONFIGS[key]
    hll_t, aux_t, cards_t, _, _ = pkg.synth_device(cfg)
    hll, aux, cards = fetch_np(hll_t, aux_t, cards_t)
    assert np.array_equal(cards.view(np.uint64), oracle.cards(hll).view(np.uint64))       # device report() at full size
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        for mode, use_cb in ((MODE_SMH, False), (MODE_CB_SMH, True)):
            want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=use_cb, threads=16)
            for algo in (ALGO_SIG, ALGO_STREAM, ALGO_HASHJOIN):
                got = sel.run(cfg.tau, mode, r, b, algo=algo)
                assert same(got, want), (key, mode, algo, len(got), len(want))
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]
            if key == "cfg3" and not use_cb:
                assert st["evaluated"] == 49_995_000 and len(want) == 45_000               # 1000 clusters x C(10,2)
            if key == "cfg2":                    # configs[1] is the size the one-launch pass exists for (the suite's default switches it off)
                sel.set_param("small_pass", -1)
                got = sel.run(cfg.tau, mode, r, b)
                assert sel.get_param("small_pass_used") == 1 and same(got, want)
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]
                sel.set_param("small_pass", 0)


def test_dense_survivor_graph_full_size():
    """bench.py's harder workload at its size (configs[2] with a quarter of the genomes degenerate: ~730 000 stage-1 survivors among
    2 516 genomes, 290 partners per row): stage 2a walks such a list by candidate-row slice per XCD ("hist_dense_degree").  Too many
    union cardinalities for the oracle to finish in seconds, so: the dense walk == the list-order walk == the byte-row LDS kernel (an
    independent stage 2a: no bit planes, no slices), pairs, Jaccard bits and counters; and the same again on one of 8 interleaved row
    shares, where the threshold counts the share's rows"""
    cfg = pkg.SYNTH_CONFIGS["cfg3"]
    hll_t, aux_t, cards_t, _, _ = pkg.synth_device(cfg)
    n_deg = pkg.harden(aux_t)
    assert 2000 < n_deg < 3000
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        for shard in (None, (128, 8, 3)):
            if shard: sel.set_row_interleave(*shard)
            ref = None
            for dense, algo in ((32, 1), (-1, 1), (0, 1), (32, 0)):
                sel.set_param("hist_dense_degree", dense)
                if algo == 0 or ref is None:
                    sel.set_param("hist_algo", algo); sel.attach(hll_t, aux_t, cards_t)        # (takes effect at attach)
                got = sel.run(cfg.tau, MODE_SMH, r, b)
                st = sel.stats()
                if ref is None:
                    ref = (got.copy(), st)
                    assert st["survivors"] > (600_000 if not shard else 60_000) and 0 < len(got) < st["survivors"]
                assert st == ref[1], (shard, dense, algo, st, ref[1])
                assert np.array_equal(got["i"], ref[0]["i"]) and np.array_equal(got["k"], ref[0]["k"])
                assert np.array_equal(got["jaccard"].view(np.uint64), ref[0]["jaccard"].view(np.uint64))
            sel.set_param("hist_algo", 1)
        sel.set_row_interleave(0, 1, 0)


def test_grouping_scan_in_lds_above_64k(oracle):
    """17 000 genomes: the grouping's single-block kernels work on n ints of LDS with a 68 KB dynamic allocation (above the 64 KB
    default limit, below the 32 768-row switch to the rocPRIM scan) -- the sizes the weak-scaling bench reaches at 4 and 8 GPUs
    (20 000 / 28 280 genomes): the query-row order (csr_scan_fill_kernel, every block repeating the scan; the automatic choice here)
    and the label order (forced: group sums, rocPRIM scan, bucket starts, scatter); bit-exact against the oracle"""
    from cuda_selection_criteria_amd.synth import SynthConfig
    cfg = SynthConfig("n17000", 17_000, 128, 0.9, 0x5EED0044)
    hll_t, aux_t, cards_t, _, _ = pkg.synth_device(cfg)
    hll, aux, cards = fetch_np(hll_t, aux_t, cards_t)
    r, b = pkg.banding(cfg.m, cfg.tau)
    want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=False, threads=16)
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        for label in (-1, 0, 1):
            sel.set_param("group_label", label)
            got = sel.run(cfg.tau, MODE_SMH, r, b)
            assert same(got, want) and len(got) > 10_000
            s = sel.stats()
            assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]


def test_config4_scale_properties():
    """50 000 genomes, m=512 (configs[3]; 1.25e9 pairs -- beyond the oracle's reach in a test): properties only"""
    cfg = pkg.SYNTH_CONFIGS["cfg4"]
    hll_t, aux_t, cards_t, perm, _ = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    n = cfg.n_genomes
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        whole = sel.run(cfg.tau, MODE_SMH, r, b)
        st = sel.stats()
        assert st["evaluated"] == n * (n - 1) // 2
        # every in-cluster pair (expected Jaccard >= 0.83) and nothing else: 5000 clusters x 45 pairs
        cluster = perm // cfg.cluster_size
        assert (cluster[whole["i"]] == cluster[whole["k"]]).all()
        assert len(whole) == (n // cfg.cluster_size) * 45
        assert (whole["i"] < whole["k"]).all() and (whole["jaccard"] >= np.float32(cfg.tau)).all()
        key = whole["i"].astype(np.int64) * n + whole["k"]
        assert (np.diff(key) > 0).all()                                                    # sorted by (i,k), unique
        # idempotent
        again = sel.run(cfg.tau, MODE_SMH, r, b)
        assert np.array_equal(again, whole)
        # 8 equal-pair shards (the multi-GPU decomposition) tile the result
        from cuda_selection_criteria_amd import distributed as D
        bounds = D.shard_rows(n, 8)
        parts = [sel.run(cfg.tau, MODE_SMH, r, b, rows=(int(bounds[j]), int(bounds[j + 1]))) for j in range(8)]
        assert np.array_equal(np.concatenate(parts), whole)
        pc = D.pair_counts(n, bounds)
        assert pc.sum() == n * (n - 1) // 2 and pc.max() < 1.01 * pc.mean()
        # the literal stream kernel agrees on a band of rows (an independent implementation of stage 1)
        lo, hi = 20000, 20600
        a = sel.run(cfg.tau, MODE_SMH, r, b, rows=(lo, hi), algo=ALGO_STREAM)
        s = sel.run(cfg.tau, MODE_SMH, r, b, rows=(lo, hi), algo=ALGO_SIG)
        assert np.array_equal(a, s) and len(a) > 0
        # the sort-based join returns the same result for the whole 1.25e9-pair space
        hj = sel.run(cfg.tau, MODE_SMH, r, b, algo=ALGO_HASHJOIN)
        assert np.array_equal(hj, whole)


def test_config5_full_size_properties(oracle):
    """BASELINE configs[4] at ITS size: 100 000 genomes, hll_a (p=8) prefilter + smh_a m=1024, tau=0.9 (5.0e9 pairs -- beyond
    the oracle's reach in a test, so: properties, and the oracle pair by pair on a sampled band of rows).
      * smh_a alone: in-cluster pairs only (most of them), Jaccard >= tau, sorted, unique;
      * two-stage == the smh_a result filtered by the oracle's hll_a, checked pair by pair for the rows [40 000, 44 000);
        outside the band: a subset of the smh_a result;
      * the sort-based join gives the same two-stage result for the whole space;
      * the 8 interleaved row-block parts of the multi-GPU partition tile the result and the evaluated-pair count."""
    cfg = pkg.SYNTH_CONFIGS["cfg5"]
    n = cfg.n_genomes
    assert n == 100_000 and cfg.m == 1024 and cfg.p_aux == 8
    hll_t, aux_t, cards_t, perm, aux_hll_t = pkg.synth_device(cfg)
    r, b = pkg.banding(cfg.m, cfg.tau)
    assert (r, b) == (16, 64)
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        sel.attach_aux_hll(aux_hll_t, cfg.p_aux)
        smh = sel.run(cfg.tau, MODE_CB_SMH, r, b)
        st_smh = sel.stats()
        cluster = perm // cfg.cluster_size
        # in-cluster pairs only; at tau = 0.9 those with two large private sets (expected Jaccard 0.83) fall below the threshold
        assert (cluster[smh["i"]] == cluster[smh["k"]]).all() and 0.5 * (n // cfg.cluster_size) * 45 < len(smh) < (n // cfg.cluster_size) * 45
        assert (smh["jaccard"] >= np.float32(cfg.tau)).all() and (smh["i"] < smh["k"]).all()
        key = smh["i"].astype(np.int64) * n + smh["k"]
        assert (np.diff(key) > 0).all()
        sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
        two = sel.run(cfg.tau, MODE_CB_SMH, r, b)
        st = sel.stats()
        assert st["evaluated"] == st_smh["evaluated"] and sel.last_attempts() == 1
        hj = sel.run(cfg.tau, MODE_CB_SMH, r, b, algo=ALGO_HASHJOIN)
        assert np.array_equal(hj, two)
        parts, ev = [], 0
        for part in range(8):
            sel.set_row_interleave(128, 8, part)
            parts.append(sel.run(cfg.tau, MODE_CB_SMH, r, b))
            ev += sel.stats()["evaluated"]
        sel.set_row_interleave(0, 1, 0)
        allp = np.concatenate(parts)
        assert np.array_equal(allp[np.lexsort((allp["k"], allp["i"]))], two) and ev == st["evaluated"]
        sizes = np.array([len(p) for p in parts])
        assert sizes.max() < 1.1 * sizes.mean()                                            # the interleave balances stage 2 too
    key2 = two["i"].astype(np.int64) * n + two["k"]
    assert np.isin(key2, key).all() and 0 < len(two) <= len(smh)
    # the oracle's hll_a, pair by pair, on the band
    lo, hi = 40_000, 44_000
    cards = cards_t.cpu().numpy()
    ah = aux_hll_t.cpu().numpy()
    e = cards.astype(np.uint64)
    lib = oracle.lib
    import ctypes as C
    lib.orc_hll_a_from_union.argtypes = [C.c_double, C.c_size_t, C.c_size_t, C.c_double, C.c_int, C.c_float]
    tau64 = float(np.float32(cfg.tau))
    band = smh[(smh["i"] >= lo) & (smh["i"] < hi)]
    assert len(band) > 10_000
    keep = [bool(lib.orc_hll_a_from_union(tau64, int(e[rec["i"]]), int(e[rec["k"]]),
                                          oracle.union_size(ah[rec["i"]], ah[rec["k"]], cfg.p_aux), cfg.p_aux, C.c_float(1.96))) for rec in band]
    want = band[np.array(keep, dtype=bool)]
    assert np.array_equal(two[(two["i"] >= lo) & (two["i"] < hi)], want)
