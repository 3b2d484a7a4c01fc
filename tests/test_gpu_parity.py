"""Parity tests proper: the HIP path, called through the C ABI (include/selection_hip.h), against the
CPU oracle on the same seeded inputs, and against the golden outputs of the reference itself.
Integer / index work is compared bit-exact; Jaccard values are compared bit-exact too (both sides run
the same IEEE double sequence; flavour FMA = reference Makefile build, STRICT = -ffp-contract=off)."""
import ctypes as C
import sys

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

import cuda_selection_criteria_amd as pkg  # noqa: E402
from cuda_selection_criteria_amd import (ALGO_AUTO, ALGO_HASHJOIN, ALGO_SIG, ALGO_STREAM, FP_FMA, FP_STRICT, MODE_CB_SMH, MODE_SMH, Selector)  # noqa: E402
from cuda_selection_criteria_amd.synth import SynthConfig  # noqa: E402
from cuda_selection_criteria_amd import distributed as D  # noqa: E402

sys.path.insert(0, str(GOLDEN))
import make_golden  # noqa: E402

EXP = GOLDEN / "expected"


def sorted_set(cfg, oracle, fma=1):
    """host-generated set in rank order (cards by the oracle, sort by the product's std::sort mirror)"""
    hll, aux, aux_hll = pkg.synth_host(cfg)
    oracle.set_fma(fma)
    cards = oracle.cards(hll)
    oracle.set_fma(1)
    perm = pkg.sort_by_card(cards)
    return hll[perm], aux[perm], cards[perm], perm, (aux_hll[perm] if aux_hll.size else aux_hll)


def assert_same_pairs(got, want):
    assert got.shape[0] == want.shape[0], (got.shape[0], want.shape[0])
    assert np.array_equal(got["i"], want["i"]) and np.array_equal(got["k"], want["k"])
    # bit-exact doubles
    assert np.array_equal(got["jaccard"].view(np.uint64), want["jacc"].view(np.uint64))


def test_device_present():
    assert pkg.hip_lib().selhip_device_count() >= 1


@pytest.mark.parametrize("name", list(make_golden.GOLDEN_SYNTH))
@pytest.mark.parametrize("fp_mode", [FP_FMA, FP_STRICT])
def test_synthetic_vs_oracle_and_golden(oracle, name, fp_mode):
    cfg = make_golden.GOLDEN_SYNTH[name]
    hll, aux, cards, perm, _ = sorted_set(cfg, oracle, fp_mode)
    names = [f"g{g:06d}" for g in perm]
    flavour = "fma" if fp_mode == FP_FMA else "nofma"
    oracle.set_fma(fp_mode)
    try:
        with Selector(0, fp_mode) as sel:
            sel.upload(hll, aux, None)                      # cards computed on the device
            assert np.array_equal(sel.cards().view(np.uint64), cards.view(np.uint64))
            for tau in sorted({cfg.tau, 0.5}):
                r, b = pkg.banding(cfg.m, tau)
                assert (r, b) == oracle.banding(cfg.m, tau)
                for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                    want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb)
                    for algo in (ALGO_STREAM, ALGO_SIG, ALGO_AUTO, ALGO_HASHJOIN):
                        if algo == ALGO_SIG and b not in (8, 16, 32, 64, 128):
                            with pytest.raises(pkg.SelhipError):
                                sel.run(tau, mode, r, b, algo=algo)
                            continue
                        got = sel.run(tau, mode, r, b, algo=algo)
                        assert_same_pairs(got, want)
                        s = sel.stats()
                        assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (algo, s, st)
                        assert s["candidates"] >= s["survivors"]
                        if use_cb:
                            text = pkg.format_lines(names, got)
                            assert text == (EXP / f"{name}_smh_a_h{tau}.{flavour}.txt").read_text()
    finally:
        oracle.set_fma(1)


@pytest.mark.parametrize("a,h", [(32, 0.9), (32, 0.01), (128, 0.5), (512, 0.01), (512, 0.9), (2048, 0.9), (2048, 0.01),
                                 (4096, 0.8), (4096, 0.01), (8192, 0.9), (8192, 0.01)])
def test_influenza_filelist_text(a, h):
    """the drop-in flow of selection_cuda.cpp on the reference's own fixtures == reference CPU stdout"""
    import os
    cwd = os.getcwd()
    os.chdir(GOLDEN)
    try:
        for fp_mode, flavour in ((FP_FMA, "fma"), (FP_STRICT, "nofma")):
            got = pkg.select_from_filelist("influenza_filelist.txt", h, a, fp_mode=fp_mode)
            assert got == (EXP / f"influenza_smh_a_a{a}_h{h}.{flavour}.txt").read_text()
        if h == 0.9:
            want = (GOLDEN / "results_reference.txt").read_text().replace("datasets/test_influenzaA/", "influenza/")
            assert pkg.select_from_filelist("influenza_filelist.txt", h, a) == want
    finally:
        os.chdir(cwd)


@pytest.mark.parametrize("name", ["synth_spread_n600_m64", "synth_flat_n300_m512", "synth_flat_n200_m128"])
@pytest.mark.parametrize("fp_mode", [FP_FMA, FP_STRICT])
def test_aux_hll_criteria_vs_oracle_and_golden(oracle, name, fp_mode):
    """hll_a (selection.cpp:122-173), hll_an (:175-227) and the two-stage hll_a + smh_a of BASELINE config 5"""
    cfg = make_golden.GOLDEN_SYNTH[name]
    hll, aux, cards, perm, aux_hll = sorted_set(cfg, oracle, fp_mode)
    names = [f"g{g:06d}" for g in perm]
    flavour = "fma" if fp_mode == FP_FMA else "nofma"
    oracle.set_fma(fp_mode)
    try:
        with Selector(0, fp_mode) as sel:
            sel.upload(hll, aux, cards)
            sel.upload_aux_hll(aux_hll, 8)
            for tau in (cfg.tau, 0.5):
                r, b = pkg.banding(cfg.m, tau)
                for crit in (pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN, pkg.CRIT_HLL_A_SMH_A):
                    sel.set_criterion(crit)
                    for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                        want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb, criterion=crit, aux_hll=aux_hll, p_aux=8)
                        for algo in ((ALGO_STREAM, ALGO_AUTO) if crit == pkg.CRIT_HLL_A_SMH_A else (ALGO_AUTO,)):
                            got = sel.run(tau, mode, r, b, algo=algo)
                            assert_same_pairs(got, want)
                            s = sel.stats()
                            assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (crit, mode, s, st)
                        if crit == pkg.CRIT_HLL_A and use_cb and tau == cfg.tau:
                            assert pkg.format_lines(names, got) == (EXP / f"{name}_hll_a_h{cfg.tau}.{flavour}.txt").read_text()
            sel.set_criterion(pkg.CRIT_SMH_A)
            want, _ = oracle.select(hll, aux, cards, cfg.tau, *pkg.banding(cfg.m, cfg.tau))
            assert_same_pairs(sel.run(cfg.tau), want)
        # a context without auxiliary sketches refuses the criterion loudly
        with Selector(0, fp_mode) as sel:
            sel.upload(hll, aux, cards)
            sel.set_criterion(pkg.CRIT_HLL_A)
            with pytest.raises(pkg.SelhipError):
                sel.run(cfg.tau)
    finally:
        oracle.set_fma(1)


@pytest.mark.parametrize("crit", ["hll_a", "hll_an"])
@pytest.mark.parametrize("h", [0.9, 0.5, 0.01])
def test_influenza_hll_criteria_text(crit, h):
    import os
    cwd = os.getcwd()
    os.chdir(GOLDEN)
    try:
        for fp_mode, flavour in ((FP_FMA, "fma"), (FP_STRICT, "nofma")):
            got = pkg.select_from_filelist("influenza_filelist.txt", h, 256, fp_mode=fp_mode, criterion=crit)
            assert got == (EXP / f"influenza_{crit}_a256_h{h}.{flavour}.txt").read_text()
    finally:
        os.chdir(cwd)


def test_row_shards_union_equals_whole(oracle):
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n1000_m256"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        for algo in (ALGO_STREAM, ALGO_SIG, ALGO_HASHJOIN):
            whole = sel.run(cfg.tau, MODE_CB_SMH, r, b, algo=algo)
            parts = []
            bounds = [0, 137, 138, 500, 999, 1000]
            for lo, hi in zip(bounds[:-1], bounds[1:]):
                parts.append(sel.run(cfg.tau, MODE_CB_SMH, r, b, rows=(lo, hi), algo=algo))
            cat = np.concatenate(parts)
            assert np.array_equal(cat, whole)
            assert len(whole) > 1000


def test_chunk_lanes_equal_plain(oracle):
    """row chunks as whole chains on two internal streams (selhip_ctx_set_pipeline; automatic from 5e8 pairs per pass): same
    pairs, same counters -- smh_a and the two-stage criterion, whole range / sub-range / interleaved parts"""
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n1000_m256"]
    hll, aux, cards, _, aux_hll = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    want, st = oracle.select(hll, aux, cards, cfg.tau, r, b)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        sel.upload_aux_hll(aux_hll, 8)
        sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
        sel.set_pipeline(0)
        want2 = sel.run(cfg.tau, MODE_CB_SMH, r, b)
        st2 = sel.stats()
        assert 0 < len(want2) <= len(want)
        for chunks in (0, 2, 3, 4, 8):
            sel.set_pipeline(chunks)
            sel.set_param("group_label", chunks % 4 == 0)     # with and without the label order of the grouping
            sel.set_criterion(pkg.CRIT_SMH_A)
            for algo in (ALGO_SIG, ALGO_STREAM):
                for rows in (None, (100, 900)):
                    got = sel.run(cfg.tau, MODE_CB_SMH, r, b, algo=algo, rows=rows)
                    w = want if rows is None else want[(want["i"] >= rows[0]) & (want["i"] < rows[1])]
                    assert_same_pairs(got, w)
                    if rows is None:
                        s = sel.stats()
                        assert s["survivors"] == st["survivors"] and s["evaluated"] == st["evaluated"]
            parts = []
            for part in range(3):
                sel.set_row_interleave(32, 3, part)
                parts.append(sel.run(cfg.tau, MODE_CB_SMH, r, b))
            sel.set_row_interleave(0, 1, 0)
            cat = np.concatenate(parts)
            assert_same_pairs(cat[np.lexsort((cat["k"], cat["i"]))], want)
            sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
            got = sel.run(cfg.tau, MODE_CB_SMH, r, b)
            assert np.array_equal(got, want2)
            assert sel.stats() == st2
        # timers: a kernel's figure is the sum over its launches of a pass, "join_span" the time from the first start to the last end
        sel.set_pipeline(2)
        sel.set_criterion(pkg.CRIT_SMH_A)
        sel.timing(1)
        for _ in range(3):
            sel.run(cfg.tau, MODE_CB_SMH, r, b, algo=ALGO_SIG)
        assert sel.kernel_launches("join") == 2.0 and sel.kernel_launches("hist") == 2.0
        assert 0 < sel.kernel_ms("join_span") <= sel.kernel_ms("total")
        assert sel.kernel_ms("join_span") < sel.kernel_ms("join") * 1.5 + 0.05
        sel.timing(0)


def test_interleaved_row_blocks_tile_the_pair_space(oracle):
    """selhip_ctx_set_row_interleave: the parts' results are disjoint, their union is the whole, their evaluated-pair
    counts add up -- every stage-1 algorithm, CB and non-CB, and the explicit pair enumeration of hll_a"""
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n1000_m256"]
    hll, aux, cards, _, aux_hll = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        sel.upload_aux_hll(aux_hll, 8)
        for crit, algos in ((pkg.CRIT_SMH_A, (ALGO_SIG, ALGO_STREAM, ALGO_HASHJOIN)), (pkg.CRIT_HLL_A, (ALGO_AUTO,)),
                            (pkg.CRIT_HLL_A_SMH_A, (ALGO_AUTO,))):
            sel.set_criterion(crit)
            for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=use_cb, criterion=crit, aux_hll=aux_hll, p_aux=8)
                for algo in algos:
                    for parts, block in ((2, 96), (3, 32), (8, 64), (5, 128)):
                        got, ev = [], 0
                        for part in range(parts):
                            sel.set_row_interleave(block, parts, part)
                            res = sel.run(cfg.tau, mode, r, b, algo=algo)
                            assert (D.interleave_owner(res["i"], block, parts) == part).all()      # (blocks are dealt boustrophedon)
                            got.append(res)
                            ev += sel.stats()["evaluated"]
                        sel.set_row_interleave(0, 1, 0)
                        cat = np.concatenate(got)
                        cat = cat[np.lexsort((cat["k"], cat["i"]))]
                        assert_same_pairs(cat, want)
                        assert ev == st["evaluated"], (crit, mode, algo, parts, ev, st)
        # the multi-rank combination of ADVICE r2: row interleave + two chunk lanes + label-ordered grouping, every part (bit planes
        # and byte rows)
        for crit in (pkg.CRIT_SMH_A, pkg.CRIT_HLL_A_SMH_A):
            sel.set_criterion(crit)
            want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=False, criterion=crit, aux_hll=aux_hll, p_aux=8)
            for lanes, label in ((2, 1), (3, 1), (2, 0)):
                sel.set_pipeline(lanes); sel.set_param("group_label", label)
                for parts, block in ((2, 64), (8, 32)):
                    got, ev = [], 0
                    for part in range(parts):
                        sel.set_row_interleave(block, parts, part)
                        got.append(sel.run(cfg.tau, MODE_SMH, r, b))
                        ev += sel.stats()["evaluated"]
                    sel.set_row_interleave(0, 1, 0)
                    cat = np.concatenate(got)
                    cat = cat[np.lexsort((cat["k"], cat["i"]))]
                    assert_same_pairs(cat, want)
                    assert ev == st["evaluated"]
        sel.set_pipeline(-1); sel.set_param("group_label", -1)
        # a sub-range of rows combined with the interleave
        sel.set_criterion(pkg.CRIT_SMH_A)
        want, _ = oracle.select(hll, aux, cards, cfg.tau, r, b)
        got = []
        for part in range(4):
            sel.set_row_interleave(32, 4, part)
            got.append(sel.run(cfg.tau, MODE_CB_SMH, r, b, rows=(101, 877)))
        sel.set_row_interleave(0, 1, 0)
        cat = np.concatenate(got)
        cat = cat[np.lexsort((cat["k"], cat["i"]))]
        assert_same_pairs(cat, want[(want["i"] >= 101) & (want["i"] < 877)])
        with pytest.raises(pkg.SelhipError):
            sel.set_row_interleave(33, 2, 0)


def test_stage2_grouping_on_off(oracle):
    """survivors bucketed by query row (run-aware histogram kernel) vs the ungrouped kernel: identical results"""
    cfg = make_golden.GOLDEN_SYNTH["synth_spread_n600_m64"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.5)
    want, _ = oracle.select(hll, aux, cards, 0.5, r, b)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        for on in (True, False, True):
            sel.set_stage2_grouping(on)
            assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b), want)
            assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b, rows=(17, 333)), want[(want["i"] >= 17) & (want["i"] < 333)])


def test_join_and_histogram_variants(oracle):
    """every switch of the signature path gives the oracle's pairs: 16-bit packed join (default) vs 32-bit join, single- vs
    double-buffered query batches, the hash-collision fallback of the batched verification forced on every candidate,
    and stage 2a with other task shapes / block counts"""
    # (rows, bands): 2x32, 8x64, 4x128 (two 16-byte signature groups per lane), 8x16, 16x8 and 64x8 (bands longer than 16 rows)
    for name, tau, shape in (("synth_spread_n600_m64", 0.5, None), ("synth_flat_n300_m512", 0.8, None), ("synth_flat_n300_m512", 0.5, None),
                             ("synth_flat_n200_m128", 0.9, None), ("synth_flat_n200_m128", 0.5, (16, 8)), ("synth_flat_n300_m512", 0.5, (64, 8))):
        cfg = make_golden.GOLDEN_SYNTH[name]
        hll, aux, cards, _, _ = sorted_set(cfg, oracle)
        r, b = shape or pkg.banding(cfg.m, tau)
        assert b in (8, 16, 32, 64, 128)
        want, st = oracle.select(hll, aux, cards, tau, r, b)
        want_all, st_all = oracle.select(hll, aux, cards, tau, r, b, use_cb=False)
        with Selector(0) as sel:
            sel.upload(hll, aux, cards)
            cand = None
            # (bits, double-buffered DPP batches, forced verify fallback, query side 1 = LDS tile / 0 = DPP, waves per block,
            #  tile height)
            for bits, db, fb, q, wpb, qt in ((16, 1, 0, 1, 4, 64), (16, 1, 0, 1, 8, 128), (16, 1, 0, 1, 4, 16), (16, 1, 0, 1, 8, 256),
                                             (16, 1, 0, 0, 1, 128), (16, 0, 0, 0, 4, 96), (32, 0, 0, 0, 1, 128),
                                             (16, 1, 1, 1, 4, 64), (16, 0, 1, 0, 4, 128),
                                             (15, 1, 0, 1, 4, 64), (15, 1, 0, 1, 8, 128), (15, 1, 0, 0, 1, 128), (15, 1, 1, 1, 4, 32)):
                sel.set_param("join_bits", bits); sel.set_param("join_db", db); sel.set_param("verify_fb", fb)
                sel.set_param("join_q", q); sel.set_param("join_wpb", wpb); sel.set_param("join_qt", qt)
                assert_same_pairs(sel.run(tau, MODE_CB_SMH, r, b, algo=ALGO_SIG), want)
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]
                assert_same_pairs(sel.run(tau, MODE_SMH, r, b, algo=ALGO_SIG), want_all)
                s = sel.stats()
                assert s["survivors"] == st_all["survivors"]
                cand = s["candidates"] if cand is None else cand
                assert s["candidates"] == cand          # the 32-bit candidate set is the same whichever join produced it
            sel.set_param("join_bits", 16); sel.set_param("join_db", 1); sel.set_param("verify_fb", 0); sel.set_param("join_wpb", 4)
            sel.set_param("join_q", 1); sel.set_param("join_qt", 0)
            # inner loop of the LDS-tile join: zero-half test (default) vs packed minimum -- same 16-bit matches, hence the same candidates
            for form, wpb, qt in ((0, 4, 32), (1, 8, 64), (0, 8, 128), (1, 4, 0)):
                sel.set_param("join_form", form); sel.set_param("join_wpb", wpb); sel.set_param("join_qt", qt)
                assert_same_pairs(sel.run(tau, MODE_CB_SMH, r, b, algo=ALGO_SIG), want)
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"] and s["candidates"] == cand
                assert_same_pairs(sel.run(tau, MODE_SMH, r, b, algo=ALGO_SIG), want_all)
            sel.set_param("join_form", 0); sel.set_param("join_wpb", 4); sel.set_param("join_qt", 0)
            # grid of the LDS-tile join: only the units above the diagonal (default) vs the whole rectangle, whole set and row ranges
            for tri, qt in ((0, 32), (1, 16), (1, 64), (0, 128), (1, 128), (1, 0)):
                sel.set_param("join_tri", tri); sel.set_param("join_qt", qt)
                assert_same_pairs(sel.run(tau, MODE_CB_SMH, r, b, algo=ALGO_SIG), want)
                assert sel.stats()["evaluated"] == st["evaluated"]
                assert_same_pairs(sel.run(tau, MODE_SMH, r, b, algo=ALGO_SIG), want_all)
                for lo, hi_ in ((64, 200), (96, 97), (0, 33), (128, want_all["i"].max() + 2)):
                    hi_ = min(int(hi_), hll.shape[0])
                    assert_same_pairs(sel.run(tau, MODE_SMH, r, b, rows=(lo, hi_), algo=ALGO_SIG), want_all[(want_all["i"] >= lo) & (want_all["i"] < hi_)])
            sel.set_param("join_tri", 0); sel.set_param("join_qt", 0)
            for run, blocks, label in ((1, 8, 0), (3, 64, 1), (8, 2048, 0), (1024, 16384, 1), (1, 16384, 0), (0, 16384, 1), (0, 16384, -1)):
                sel.set_param("hist_run", run); sel.set_param("hist_blocks", blocks); sel.set_param("group_label", label)
                assert_same_pairs(sel.run(tau, MODE_CB_SMH, r, b), want)
                assert_same_pairs(sel.run(tau, MODE_CB_SMH, r, b, algo=ALGO_STREAM), want)      # rows tallied by csr_count_kernel
            for bad in (("join_bits", 24), ("join_qt", 24), ("join_wpb", 2), ("init_cap", -1), ("hist_blocks", 12), ("hist_run", -1), ("group_label", 2),
                        ("no_such_param", 1)):
                with pytest.raises(pkg.SelhipError):
                    sel.set_param(*bad)


@pytest.mark.parametrize("algo", [ALGO_SIG, ALGO_STREAM, ALGO_HASHJOIN])
@pytest.mark.parametrize("grouping", ["label", True, False])
def test_list_overflow_grows_and_repeats(oracle, algo, grouping):
    """the candidate / survivor / result lists start far too small ("init_cap" test hook): the pass must notice from its exact
    counters, grow the lists and repeat -- never read or write past a list (the per-row tally of verify16_kernel
    used to count clipped survivors, so the row-grouped copy was written out of bounds) -- and give the oracle's pairs"""
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.5)
    want, st = oracle.select(hll, aux, cards, 0.5, r, b, use_cb=False)
    assert st["survivors"] > 600
    with Selector(0) as sel:
        sel.set_param("init_cap", 128)                       # 2 records per append segment of the join
        sel.set_stage2_grouping(bool(grouping))
        sel.set_param("group_label", 1 if grouping == "label" else 0)
        sel.upload(hll, aux, cards)
        got = sel.run(0.5, MODE_SMH, r, b, algo=algo)
        assert sel.last_attempts() >= 2
        assert_same_pairs(got, want)
        s = sel.stats()
        assert s["survivors"] == st["survivors"] and s["evaluated"] == st["evaluated"]
        got = sel.run(0.5, MODE_SMH, r, b, algo=algo)       # the grown lists are kept: one attempt now
        assert sel.last_attempts() == 1
        assert_same_pairs(got, want)


@pytest.mark.parametrize("algo", [ALGO_SIG, ALGO_STREAM])
def test_harder_workload_vs_oracle(oracle, algo):
    """bench.py's `harder_workload` recipe (pkg.harden: 25 % of the genomes get buckets mod 2, so >= 1 % of ALL pairs pass a
    band and reach stage 2) at N = 3000, m = 512, tau = 0.8: pairs, Jaccard bits and counters equal the oracle's"""
    cfg = SynthConfig("hard", 3000, 512, 0.8, 0x5EED0002)
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    n_deg = pkg.harden(aux)
    assert 600 < n_deg < 900
    r, b = pkg.banding(cfg.m, cfg.tau)
    want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=False)
    assert st["survivors"] > 0.01 * st["evaluated"]          # the point of the recipe
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        got = sel.run(cfg.tau, MODE_SMH, r, b, algo=algo)
        assert_same_pairs(got, want)
        s = sel.stats()
        assert s["survivors"] == st["survivors"] and s["evaluated"] == st["evaluated"]
        # the same irregular pair graph (one huge near-clique of degenerate genomes beside the small clusters) through the
        # label-ordered grouping and the chunk lanes
        sel.set_param("group_label", 1)
        for chunks in (0, 2, 5):
            sel.set_pipeline(chunks)
            assert_same_pairs(sel.run(cfg.tau, MODE_SMH, r, b, algo=algo), want)
            assert sel.stats()["survivors"] == st["survivors"]


@pytest.mark.parametrize("crit", [pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN])
def test_aux_first_criterion_enumerates_in_sub_passes(oracle, crit):
    """hll_a / hll_an as FIRST criterion list the pair space explicitly; more pairs than one list holds are taken in row
    sub-ranges on the stream ("enum_pairs" shrinks the list for the test; the reference's CPU loop, selection.cpp:152-227, has no
    size limit, and round 1 refused more than 2^28 pairs).  Contiguous rows and interleaved row blocks, vs the oracle."""
    cfg = SynthConfig("enum", 700, 128, 0.9, 0x31, p_aux=8, mode=1, n_sh_lo=3000, n_sh_hi=40000)
    hll, aux, cards, _, ah = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    for use_cb in (True, False):
        want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=use_cb, criterion=crit, aux_hll=ah, p_aux=8)
        mode = MODE_CB_SMH if use_cb else MODE_SMH
        with Selector(0) as sel:
            sel.upload(hll, aux, cards)
            sel.upload_aux_hll(ah, 8)
            sel.set_criterion(crit)
            for budget in (1 << 26, 20000, 1500, 1):
                sel.set_param("enum_pairs", budget)
                assert_same_pairs(sel.run(cfg.tau, mode, r, b), want)
                assert sel.stats()["evaluated"] == st["evaluated"]
                assert_same_pairs(sel.run(cfg.tau, mode, r, b, rows=(100, 555)), want[(want["i"] >= 100) & (want["i"] < 555)])
            sel.set_param("enum_pairs", 3000)
            parts = []
            for part in range(3):
                sel.set_row_interleave(32, 3, part)
                parts.append(sel.run(cfg.tau, mode, r, b))
            sel.set_row_interleave(0, 1, 0)
            allp = np.concatenate(parts)
            allp = allp[np.lexsort((allp["k"], allp["i"]))]
            assert_same_pairs(allp, want)


def test_edge_cases(oracle):
    cfg = SynthConfig("edge", 130, 128, 0.9, 77, n_sh_lo=5000, n_sh_hi=5000)
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        # empty set, single genome, two genomes
        for n in (0, 1, 2, 3):
            sel.upload(hll[:n], aux[:n], cards[:n])
            got = sel.run(cfg.tau, MODE_CB_SMH, r, b)
            want, _ = oracle.select(hll[:n], aux[:n], cards[:n], cfg.tau, r, b)
            assert_same_pairs(got, want)
        # genomes with zero cardinality (all-zero registers): `if (e2 == 0) continue` (selection.cpp:281)
        h2, a2 = hll.copy(), aux.copy()
        h2[:5] = 0
        c2 = oracle.cards(h2)
        perm = pkg.sort_by_card(c2)
        h2, a2, c2 = h2[perm], a2[perm], c2[perm]
        assert (c2[:5] == 0).all()
        sel.upload(h2, a2, c2)
        for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
            got = sel.run(0.5, mode, *pkg.banding(cfg.m, 0.5))
            want, st = oracle.select(h2, a2, c2, 0.5, *pkg.banding(cfg.m, 0.5), use_cb=use_cb)
            assert_same_pairs(got, want)
            assert sel.stats()["evaluated"] == st["evaluated"]
        # identical sketches (duplicates): every band equal, J == 1-ish; and tau > 1 selects nothing
        h3, a3 = hll.copy(), aux.copy()
        h3[1] = h3[0]; a3[1] = a3[0]
        c3 = oracle.cards(h3)
        perm = pkg.sort_by_card(c3)
        h3, a3, c3 = h3[perm], a3[perm], c3[perm]
        sel.upload(h3, a3, c3)
        got = sel.run(0.9, MODE_CB_SMH, r, b)
        want, _ = oracle.select(h3, a3, c3, 0.9, r, b)
        assert_same_pairs(got, want)
        assert len(got) >= 1
        # all band shapes of m=128, including rows=1 and rows=m
        for rows in (1, 2, 4, 8, 16, 32, 64, 128):
            want, st = oracle.select(h3, a3, c3, 0.3, rows, 128 // rows, use_cb=False)
            for algo in (ALGO_STREAM, ALGO_AUTO, ALGO_HASHJOIN):
                got = sel.run(0.3, MODE_SMH, rows, 128 // rows, algo=algo)
                assert_same_pairs(got, want)
                assert sel.stats()["survivors"] == st["survivors"], rows
        # bad banding is an error, not a silent empty result
        with pytest.raises(pkg.SelhipError):
            sel.run(0.9, MODE_CB_SMH, 3, 5)


def test_big_bands_and_generic_shapes(oracle):
    """rows >= 128 (a band spans whole 128-bucket chunks) and shapes the fast kernel does not cover"""
    rng = np.random.default_rng(5)
    for m, shapes in ((512, [(128, 4), (256, 2), (512, 1), (64, 8)]), (96, [(3, 32), (32, 3), (1, 96)]), (64, [(8, 8), (64, 1)])):
        cfg = SynthConfig("shape", 96, 128, 0.9, 1234 + m, n_sh_lo=4000, n_sh_hi=4000)
        hll, _, cards, _, _ = sorted_set(cfg, oracle)
        n = hll.shape[0]
        # buckets from a tiny alphabet + planted equal bands so that long bands do match sometimes
        aux = rng.integers(0, 2, size=(n, m), dtype=np.uint64)
        for g in range(1, n, 3):
            aux[g] = aux[g - 1]
            aux[g, rng.integers(0, m)] ^= np.uint64(1)        # differ in exactly one bucket
        with Selector(0) as sel:
            sel.upload(hll, aux, cards)
            for rows, bands in shapes:
                want, st = oracle.select(hll, aux, cards, 0.0, rows, bands, use_cb=False)
                for algo in (ALGO_STREAM, ALGO_AUTO):
                    got = sel.run(0.0, MODE_SMH, rows, bands, algo=algo)
                    assert sel.stats()["survivors"] == st["survivors"], (m, rows, bands, algo)
                    assert_same_pairs(got, want)


def test_building_blocks(oracle):
    import torch
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, aux_hll = sorted_set(cfg, oracle)
    lib = pkg.hip_lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n = hll.shape[0]
    pairs = np.stack([rng.integers(0, n, 500), rng.integers(0, n, 500)], axis=1).astype(np.int32)
    pairs[:50, 1] = pairs[:50, 0] // 10 * 10 + (pairs[:50, 0] + 1) % 10     # same-cluster pairs
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_hll = torch.from_numpy(hll).to(dev)
    d_aux = torch.from_numpy(aux.view(np.int64)).to(dev)
    # union histograms
    d_counts = torch.zeros((500, 64), dtype=torch.int32, device=dev)
    pkg._lib.check(lib.selhip_hll_union_hist(d_hll.data_ptr(), 14, d_pairs.data_ptr(), 500, d_counts.data_ptr(), None))
    counts = d_counts.cpu().numpy().view(np.uint32)
    for j in range(500):
        assert np.array_equal(counts[j], oracle.union_hist(hll[pairs[j, 0]], hll[pairs[j, 1]])), j
    # estimator, both flavours
    for fp in (FP_FMA, FP_STRICT):
        d_est = torch.zeros(500, dtype=torch.float64, device=dev)
        pkg._lib.check(lib.selhip_ertl_estimate(d_counts.data_ptr(), 500, 14, fp, d_est.data_ptr(), None))
        est = d_est.cpu().numpy()
        want = np.array([oracle.estimate(counts[j], 14, fp) for j in range(500)])
        assert np.array_equal(est.view(np.uint64), want.view(np.uint64))
    # p = 8 auxiliary sketches through the same kernels
    d_ah = torch.from_numpy(aux_hll).to(dev)
    pkg._lib.check(lib.selhip_hll_union_hist(d_ah.data_ptr(), 8, d_pairs.data_ptr(), 500, d_counts.data_ptr(), None))
    counts8 = d_counts.cpu().numpy().view(np.uint32)
    d_est = torch.zeros(500, dtype=torch.float64, device=dev)
    pkg._lib.check(lib.selhip_ertl_estimate(d_counts.data_ptr(), 500, 8, FP_FMA, d_est.data_ptr(), None))
    est8 = d_est.cpu().numpy()
    for j in range(500):
        assert np.array_equal(counts8[j], oracle.union_hist(aux_hll[pairs[j, 0]], aux_hll[pairs[j, 1]]))
        assert est8[j] == oracle.union_size(aux_hll[pairs[j, 0]], aux_hll[pairs[j, 1]], 8)
    # smh_a flags and bucket-match counts
    r, b = pkg.banding(cfg.m, cfg.tau)
    d_flags = torch.zeros(500, dtype=torch.uint8, device=dev)
    pkg._lib.check(lib.selhip_smh_a_pairs(d_aux.data_ptr(), cfg.m, r, b, d_pairs.data_ptr(), 500, d_flags.data_ptr(), None))
    flags = d_flags.cpu().numpy()
    d_mc = torch.zeros(500, dtype=torch.int32, device=dev)
    pkg._lib.check(lib.selhip_smh_match_counts(d_aux.data_ptr(), cfg.m, d_pairs.data_ptr(), 500, d_mc.data_ptr(), None))
    mc = d_mc.cpu().numpy()
    for j in range(500):
        assert bool(flags[j]) == oracle.smh_a(aux[pairs[j, 0]], aux[pairs[j, 1]], r, b)
        assert mc[j] == int((aux[pairs[j, 0]] == aux[pairs[j, 1]]).sum())
    assert flags.sum() > 0


def test_aux_precision_limit_and_counter_recovery(oracle):
    """(1) the auxiliary criteria count in 16-bit bins, so p_aux = 15 is the largest precision any entry point takes -- a
    pair of EMPTY auxiliary sketches (one bin holding all 32 768 registers) must still give the oracle's answer there, and 16 is
    refused; (2) an enqueue that fails after it claimed its counter set must not leave stale counters for the passes behind it"""
    cfg = SynthConfig("paux", 60, 128, 0.9, 0x77, p_aux=8, n_sh_lo=6000, n_sh_hi=6000)
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    n = hll.shape[0]
    rng = np.random.default_rng(4)
    for p_aux in (15, 12):
        ah = np.minimum(rng.geometric(0.5, size=(n, 1 << p_aux)), 40).astype(np.uint8)
        ah[0] = 0; ah[1] = 0                          # two empty sketches: union histogram = one bin of 2^p_aux
        ah[2] = 7; ah[3] = 7                          # ... and two constant ones
        ah[5] = ah[4]
        r, b = pkg.banding(cfg.m, 0.5)
        with Selector(0) as sel:
            sel.upload(hll, aux, cards)
            sel.upload_aux_hll(ah, p_aux)
            for crit in (pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN, pkg.CRIT_HLL_A_SMH_A):
                sel.set_criterion(crit)
                want, st = oracle.select(hll, aux, cards, 0.5, r, b, use_cb=False, criterion=crit, aux_hll=ah, p_aux=p_aux)
                assert_same_pairs(sel.run(0.5, MODE_SMH, r, b), want)
                assert sel.stats()["survivors"] == st["survivors"]
            with pytest.raises(pkg.SelhipError):
                sel.upload_aux_hll(np.zeros((n, 1 << 16), dtype=np.uint8), 16)
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.5)
    want, st = oracle.select(hll, aux, cards, 0.5, r, b)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        for _ in range(3):
            assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b), want)
            sel.set_param("fail_after_flip", 1)
            with pytest.raises(pkg.SelhipError):
                sel.run(0.5, MODE_CB_SMH, r, b)
            for _ in range(2):                        # the pass after the failure, and the one after that (which uses the other set)
                assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b), want)
                s = sel.stats()
                assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"] and s["selected"] == len(want)


def test_small_sets_skip_grouping(oracle):
    """the library's own default: sets of up to 2 048 genomes go to stage 2 ungrouped (no grouping launches); same pairs, same counters,
    every criterion, with and without chunk lanes; above the limit the list is grouped again"""
    cfg = SynthConfig("small", 900, 256, 0.9, 0x99, p_aux=8)
    hll, aux, cards, _, ah = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        sel.upload_aux_hll(ah, 8)
        for min_n in (2048, 0, 899, 900):
            sel.set_param("group_min_n", min_n)
            assert sel.get_param("label_order") == (1 if min_n < 900 else 0)
            for crit in (pkg.CRIT_SMH_A, pkg.CRIT_HLL_A_SMH_A, pkg.CRIT_HLL_A):
                sel.set_criterion(crit)
                for lanes in (-1, 2):
                    sel.set_pipeline(lanes)
                    want, st = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=False, criterion=crit, aux_hll=ah, p_aux=8)
                    assert_same_pairs(sel.run(cfg.tau, MODE_SMH, r, b), want)
                    s = sel.stats()
                    assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]


def test_small_pass_one_launch(oracle):
    """sets of up to 2 048 genomes with criterion smh_a take their whole pass in ONE launch (small_pass_kernel): same pairs,
    Jaccard bits and counters as the oracle -- and as the regular chain -- for both modes, row ranges, candidate ranges, both FP
    flavours, empty sketches, band shapes with long bands; an overflowing block list falls back to the regular pass"""
    for name, tau in (("synth_flat_n1000_m256", 0.9), ("synth_spread_n600_m64", 0.5), ("synth_flat_n300_m512", 0.5), ("synth_flat_n200_m128", 0.9)):
        cfg = make_golden.GOLDEN_SYNTH[name]
        for fp_mode in (FP_FMA, FP_STRICT):
            hll, aux, cards, _, _ = sorted_set(cfg, oracle, fp_mode)
            oracle.set_fma(fp_mode)
            try:
                r, b = pkg.banding(cfg.m, tau)
                with Selector(0, fp_mode) as sel:
                    sel.set_param("sig_cache", 1 if fp_mode == FP_STRICT else 0)     # (the one-launch pass rewrites part of the cached signatures)
                    sel.upload(hll, aux, cards)
                    for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                        want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb)
                        for small in (1, 0, -1):
                            sel.set_param("small_pass", small)
                            got = sel.run(tau, mode, r, b)
                            assert sel.get_param("small_pass_used") == (1 if small else 0)
                            assert_same_pairs(got, want)
                            s = sel.stats()
                            assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"] and s["selected"] == len(want), (name, small, s, st)
                        lo, hi_ = 37, min(211, hll.shape[0])
                        assert_same_pairs(sel.run(tau, mode, r, b, rows=(lo, hi_)), want[(want["i"] >= lo) & (want["i"] < hi_)])
                        assert sel.get_param("small_pass_used") == 1
                    sel.set_candidate_begin(100)
                    want_r, _ = oracle.select(hll, aux, cards, tau, r, b)
                    got = sel.run(tau, MODE_CB_SMH, r, b, rows=(0, 100))
                    assert_same_pairs(got, want_r[(want_r["i"] < 100) & (want_r["k"] >= 100)])
                    sel.set_candidate_begin(0)
            finally:
                oracle.set_fma(1)
    # genomes with zero cardinality and duplicates; many passes in a row (the barrier word lives in the double-buffered counter sets)
    cfg = SynthConfig("small-edge", 130, 128, 0.9, 77, n_sh_lo=5000, n_sh_hi=5000)
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    hll[:5] = 0; hll[7] = hll[6]; aux[7] = aux[6]
    cards = oracle.cards(hll)
    perm = pkg.sort_by_card(cards)
    hll, aux, cards = hll[perm], aux[perm], cards[perm]
    with Selector(0) as sel:
        sel.set_param("small_pass", -1)
        sel.upload(hll, aux, cards)
        for rows in (2, 4, 8, 16):
            want, st = oracle.select(hll, aux, cards, 0.3, rows, 128 // rows, use_cb=False)
            for _ in range(3):
                assert_same_pairs(sel.run(0.3, MODE_SMH, rows, 128 // rows), want)
                assert sel.get_param("small_pass_used") == 1 and sel.stats()["survivors"] == st["survivors"]
        for rows in (1, 32, 64):                               # shapes the one-launch pass does not take (one row per band; fewer than 8 bands)
            want, _ = oracle.select(hll, aux, cards, 0.3, rows, 128 // rows, use_cb=False)
            assert_same_pairs(sel.run(0.3, MODE_SMH, rows, 128 // rows), want)
            assert sel.get_param("small_pass_used") == 0
    # the largest sets the pass takes: eight rows per block, two rounds of 1 024 candidates, partial last chunk; with and without CB
    cfg = SynthConfig("small-max", 2041, 64, 0.8, 78, mode=1, n_sh_lo=5000, n_sh_hi=40000)
    hll2, aux2, cards2, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.8)
    with Selector(0) as sel:
        sel.set_param("small_pass", -1)
        sel.upload(hll2, aux2, cards2)
        for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
            want, st = oracle.select(hll2, aux2, cards2, 0.8, r, b, use_cb=use_cb)
            assert_same_pairs(sel.run(0.8, mode, r, b), want)
            s = sel.stats()
            assert sel.get_param("small_pass_used") == 1 and s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"]
            assert_same_pairs(sel.run(0.8, mode, r, b, rows=(1000, 2041)), want[want["i"] >= 1000])
    # every sketch equal: each row keeps all its candidates, a block's LDS list (2 048 pairs) overflows -> regular pass, same result
    n = 1800
    hll1 = np.tile(hll[10], (n, 1)); aux1 = np.tile(aux[10], (n, 1)); cards1 = np.full(n, cards[10])
    with Selector(0) as sel:
        sel.set_param("small_pass", -1)
        sel.upload(hll1, aux1, cards1)
        got = sel.run(0.9, MODE_SMH, 16, 8)
        assert len(got) == n * (n - 1) // 2 and sel.get_param("small_pass_used") == 0 and sel.last_attempts() >= 2
        assert (got["jaccard"] == got["jaccard"][0]).all()
        got = sel.run(0.9, MODE_SMH, 16, 8)                    # ... and the context remembers
        assert len(got) == n * (n - 1) // 2 and sel.last_attempts() == 1
    # the grid barrier's wait is bounded (an ordinary launch does not promise that all blocks are resident together): with no patience
    # at all (test hook) the blocks that are not last give up, the pass is repeated on the regular path, and later passes stay there
    with Selector(0) as sel:
        sel.set_param("small_pass", 3)
        sel.upload(hll2, aux2, cards2)
        want, st = oracle.select(hll2, aux2, cards2, 0.8, r, b, use_cb=False)
        for _ in range(3):
            assert_same_pairs(sel.run(0.8, MODE_SMH, r, b), want)
            assert sel.get_param("small_pass_used") == 0 and sel.stats()["survivors"] == st["survivors"]
        sel.set_param("small_pass", 1)                         # (setting the parameter forgets the failure; the barrier words are intact)
        for _ in range(3):
            assert_same_pairs(sel.run(0.8, MODE_SMH, r, b), want)
            assert sel.get_param("small_pass_used") == 1 and sel.stats()["survivors"] == st["survivors"]


def test_signature_cache(oracle):
    """"sig_cache" = 1 keeps the band signatures across the passes of a context: they must be rebuilt when the band shape, the signature
    width or the sketches change, and only then -- thresholds, modes, row ranges, interleave parts and criteria in any order vs the oracle"""
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    cfg2 = make_golden.GOLDEN_SYNTH["synth_spread_n600_m64"]
    hll2, aux2, cards2, _, _ = sorted_set(cfg2, oracle)
    with Selector(0) as sel:
        sel.set_param("sig_cache", 1)
        sel.upload(hll, aux, cards)
        for tau, shape, bits in ((0.8, None, 16), (0.8, None, 16), (0.5, None, 16), (0.5, (64, 8), 16), (0.5, (64, 8), 15), (0.5, (64, 8), 16), (0.9, None, 16)):
            r, b = shape or pkg.banding(cfg.m, tau)
            sel.set_param("join_bits", bits)
            for mode, use_cb in ((MODE_CB_SMH, True), (MODE_SMH, False)):
                want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb)
                assert_same_pairs(sel.run(tau, mode, r, b), want)
                assert sel.stats()["survivors"] == st["survivors"]
                assert_same_pairs(sel.run(tau, mode, r, b, rows=(40, 211)), want[(want["i"] >= 40) & (want["i"] < 211)])
        r, b = pkg.banding(cfg.m, 0.8)
        want, _ = oracle.select(hll, aux, cards, 0.8, r, b)
        parts = []
        for part in range(3):
            sel.set_row_interleave(32, 3, part)
            parts.append(sel.run(0.8, MODE_CB_SMH, r, b))
        sel.set_row_interleave(0, 1, 0)
        cat = np.concatenate(parts)
        assert_same_pairs(cat[np.lexsort((cat["k"], cat["i"]))], want)
        # other sketches, same shape of call: the cached signatures of the first set must not survive the upload
        sel.upload(hll2, aux2, cards2)
        r2, b2 = pkg.banding(cfg2.m, 0.5)
        want2, _ = oracle.select(hll2, aux2, cards2, 0.5, r2, b2)
        for _ in range(2):
            assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r2, b2), want2)
        for algo in (ALGO_HASHJOIN, ALGO_STREAM, ALGO_SIG):
            assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r2, b2, algo=algo), want2)
        sel.set_param("sig_cache", 0)
        assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r2, b2), want2)


def test_bitplane_histograms(oracle):
    """stage 2a's bit-plane kernel (csrc/kernel_hllbs.cuh) against numpy on register sets whose largest value selects each of its
    instantiations (16 / 24 / 32 / 48 / 64 decoded values), including all-equal rows, an all-zero row and the value 63; and the
    whole pass with the bit planes vs the byte-row LDS kernel ("hist_algo")"""
    import torch
    lib = pkg.hip_lib()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for vmax in (3, 14, 15, 16, 23, 24, 31, 32, 47, 51, 63):
        n = 40
        hll = rng.integers(0, vmax + 1, size=(n, 16384), dtype=np.uint8)
        hll[0] = 0                                   # empty sketch
        hll[1] = vmax                                # every register at the maximum
        hll[2] = np.minimum(rng.geometric(0.5, 16384), vmax).astype(np.uint8)     # HLL-like
        hll[3] = hll[2]
        hll[4, ::7] = vmax
        for g, cap_v in zip(range(5, 14), (3, 15, 17, 19, 20, 23, 24, 27, 31)):        # rows with smaller maxima: every per-pair tree size is taken
            hll[g] = np.minimum(hll[g], min(cap_v, vmax))
        pairs = np.stack([rng.integers(0, n, 300), rng.integers(0, n, 300)], axis=1).astype(np.int32)
        pairs[:8] = [[0, 0], [0, 1], [1, 1], [2, 3], [3, 2], [0, 2], [4, 1], [4, 0]]
        d_hll = torch.from_numpy(hll).to(dev)
        d_pairs = torch.from_numpy(pairs).to(dev)
        d_planes = torch.zeros((n, 6, 512), dtype=torch.int32, device=dev)
        d_gmax = torch.zeros(n, dtype=torch.uint8, device=dev)
        khi = C.c_int(0)
        pkg._lib.check(lib.selhip_hll_bitslice(d_hll.data_ptr(), n, d_planes.data_ptr(), d_gmax.data_ptr(), C.byref(khi), None))
        assert khi.value == vmax + 1
        assert np.array_equal(d_gmax.cpu().numpy(), hll.max(axis=1))
        planes = d_planes.cpu().numpy().view(np.uint32)
        # every register lands on exactly one bit of every plane: plane b's population = number of registers with bit b set
        for g in (0, 1, 2, 4, n - 1):
            for bit in range(6):
                assert int(np.unpackbits(planes[g, bit].view(np.uint8)).sum()) == int(((hll[g] >> bit) & 1).sum()), (vmax, g, bit)
        d_counts = torch.full((300, 64), -1, dtype=torch.int32, device=dev)
        pkg._lib.check(lib.selhip_hll_union_hist_planes(d_planes.data_ptr(), d_gmax.data_ptr(), khi.value, d_pairs.data_ptr(), 300, d_counts.data_ptr(), None))
        counts = d_counts.cpu().numpy().view(np.uint32)
        for j in range(300):
            want = np.bincount(np.maximum(hll[pairs[j, 0]], hll[pairs[j, 1]]), minlength=64).astype(np.uint32)
            assert np.array_equal(counts[j], want), (vmax, j, pairs[j])
            assert np.array_equal(want, oracle.union_hist(hll[pairs[j, 0]], hll[pairs[j, 1]]))
    cfg = make_golden.GOLDEN_SYNTH["synth_spread_n600_m64"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.5)
    want, st = oracle.select(hll, aux, cards, 0.5, r, b)
    with Selector(0) as sel:
        for algo in (0, 1, -1):
            sel.set_param("hist_algo", algo)
            sel.upload(hll, aux, cards)
            # (dense 0: every XCD walks the whole list and takes its slice of the candidate rows; -1: never; 32: the default threshold)
            for run, blocks, label, dense in ((1, 8, 0, -1), (3, 64, 1, 0), (5, 16, 1, 32), (0, 2048, -1, 0), (100, 8, 1, 0), (100, 2048, 1, -1)):
                sel.set_param("hist_run", run); sel.set_param("hist_bs_blocks", blocks); sel.set_param("group_label", label)
                sel.set_param("hist_dense_degree", dense)
                assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b), want)
                assert sel.stats()["survivors"] == st["survivors"]
                assert_same_pairs(sel.run(0.5, MODE_CB_SMH, r, b, rows=(17, 333)), want[(want["i"] >= 17) & (want["i"] < 333)])
        with pytest.raises(pkg.SelhipError):
            sel.set_param("hist_algo", 2)


def test_estimator_rare_branches(oracle):
    """hll.h:642 (all registers saturated -> +inf), hll.h:659 log1p start point (gprev > 1.5*a: registers
    near saturation), empty sketch, single non-empty register -- forced inputs (never hit by random data)"""
    import torch
    lib = pkg.hip_lib()
    dev = torch.device("cuda", 0)
    hists = []
    for p in (14, 8):
        q = 64 - p
        m = 1 << p
        def h(d):
            c = np.zeros(64, dtype=np.uint32)
            for k, v in d.items():
                c[k] = v
            assert c.sum() == m
            return (p, c)
        hists += [h({q + 1: m}), h({0: m}), h({0: m - 1, 1: 1}), h({q + 1: m - 1, q: 1}), h({q + 1: m - 7, q - 1: 7}),
                  h({q: m}), h({q - 2: m // 2, q + 1: m // 2}), h({40: m}), h({1: m}), h({0: 1, q + 1: m - 1})]
    for p in (14, 8):
        sub = np.stack([c for pp, c in hists if pp == p])
        for fp in (FP_FMA, FP_STRICT):
            d_c = torch.from_numpy(sub.view(np.int32)).to(dev)
            d_e = torch.zeros(len(sub), dtype=torch.float64, device=dev)
            pkg._lib.check(lib.selhip_ertl_estimate(d_c.data_ptr(), len(sub), p, fp, d_e.data_ptr(), None))
            got = d_e.cpu().numpy()
            want = np.array([oracle.estimate(c, p, fp) for c in sub])
            assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (p, fp, got, want)
        assert np.isinf(want[0]) and want[1] == 0.0


def test_drop_in_launchers(oracle):
    """launch_kernel_smh / launch_kernel_CBsmh with the reference's parameter list (device pointers)"""
    import torch
    cfg = make_golden.GOLDEN_SYNTH["synth_spread_n600_m64"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    n = hll.shape[0]
    lib = pkg.hip_lib()
    dev = torch.device("cuda", 0)
    ii, kk = np.triu_indices(n, 1)
    pairs = np.stack([ii, kk], axis=1).astype(np.int32)           # selection_cuda.cpp:146-150
    total = pairs.shape[0]
    d_pairs = torch.from_numpy(pairs).to(dev)
    d_hll = torch.from_numpy(hll).to(dev)
    d_aux = torch.from_numpy(aux.view(np.int64)).to(dev)
    d_cards = torch.from_numpy(cards).to(dev)
    d_out = torch.zeros((total, 3), dtype=torch.int32, device=dev)
    d_cnt = torch.full((1,), -1, dtype=torch.int32, device=dev)
    tau = np.float32(cfg.tau)
    r, b = pkg.banding(cfg.m, cfg.tau)
    for fn, use_cb in ((lib.launch_kernel_smh, False), (lib.launch_kernel_CBsmh, True)):
        rc = fn(d_hll.data_ptr(), d_aux.data_ptr(), d_cards.data_ptr(), d_pairs.data_ptr(), total, float(tau),
                cfg.m, 16384, r, b, d_out.data_ptr(), d_cnt.data_ptr(), 256)
        assert rc == 0, lib.selhip_last_error(None)
        torch.cuda.synchronize()
        cnt = int(d_cnt.item())
        rec = d_out[:cnt].cpu().numpy()
        got = sorted((int(x), int(y), np.int32(s).view(np.float32)) for x, y, s in rec)
        want, _ = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=use_cb)
        exp = [(int(w["i"]), int(w["k"]), np.float32(w["jacc"])) for w in want]
        assert got == exp
    # pairs == NULL: the implicit triangle (never materialised), total_pairs = n(n-1)/2; and the 64-bit launchers, with the
    # explicit list (more pairs than one internal chunk would be > 2^20 -- here the list is 179 700 pairs) and without it
    d_cnt64 = torch.full((1,), -1, dtype=torch.int64, device=dev)
    for fn, use_cb, cnt_t, pp in ((lib.launch_kernel_smh, False, d_cnt, None), (lib.launch_kernel_CBsmh, True, d_cnt, None),
                                  (lib.launch_kernel_smh64, False, d_cnt64, None), (lib.launch_kernel_CBsmh64, True, d_cnt64, None),
                                  (lib.launch_kernel_smh64, False, d_cnt64, d_pairs.data_ptr()), (lib.launch_kernel_CBsmh64, True, d_cnt64, d_pairs.data_ptr())):
        d_out.zero_()
        rc = fn(d_hll.data_ptr(), d_aux.data_ptr(), d_cards.data_ptr(), pp, total, float(tau),
                cfg.m, 16384, r, b, d_out.data_ptr(), cnt_t.data_ptr(), 256)
        assert rc == 0, lib.selhip_last_error(None)
        torch.cuda.synchronize()
        cnt = int(cnt_t.item())
        rec = d_out[:cnt].cpu().numpy()
        got = sorted((int(x), int(y), np.int32(s).view(np.float32)) for x, y, s in rec)
        want, _ = oracle.select(hll, aux, cards, cfg.tau, r, b, use_cb=use_cb)
        assert got == [(int(w["i"]), int(w["k"]), np.float32(w["jacc"])) for w in want]
    # pairs == NULL with a total that is not a whole triangle is refused
    assert lib.launch_kernel_smh(d_hll.data_ptr(), d_aux.data_ptr(), d_cards.data_ptr(), None, total - 1, float(tau), cfg.m, 16384, r, b,
                                 d_out.data_ptr(), d_cnt.data_ptr(), 256) < 0
    # error behaviour: status codes instead of the reference's silent void
    assert lib.launch_kernel_smh(None, d_aux.data_ptr(), d_cards.data_ptr(), d_pairs.data_ptr(), total, 0.9, cfg.m, 16384, r, b,
                                 d_out.data_ptr(), d_cnt.data_ptr(), 256) < 0
    assert lib.launch_kernel_smh(d_hll.data_ptr(), d_aux.data_ptr(), d_cards.data_ptr(), d_pairs.data_ptr(), total, 0.9, cfg.m, 16384, 3, 7,
                                 d_out.data_ptr(), d_cnt.data_ptr(), 256) < 0


def test_device_generator_matches_host():
    cfg = SynthConfig("gen", 64, 256, 0.9, 99, p_aux=8, mode=1, n_sh_lo=2000, n_sh_hi=30000)
    hll_h, aux_h, ah_h = pkg.synth_host(cfg)
    hll_d, aux_d, cards_d, _, ah_d = pkg.synth_device(cfg, sort=False)
    assert np.array_equal(hll_d.cpu().numpy(), hll_h)
    assert np.array_equal(aux_d.cpu().numpy().view(np.uint64), aux_h)
    assert np.array_equal(ah_d.cpu().numpy(), ah_h)


def test_attach_device_tensors_and_sorted_generation(oracle):
    cfg = SynthConfig("attach", 500, 256, 0.9, 4242, n_sh_lo=10000, n_sh_hi=10000)
    hll_t, aux_t, cards_t, perm, _ = pkg.synth_device(cfg)
    hll, aux = hll_t.cpu().numpy(), aux_t.cpu().numpy().view(np.uint64)
    cards = cards_t.cpu().numpy()
    assert np.array_equal(cards.view(np.uint64), oracle.cards(hll).view(np.uint64))
    assert (np.diff(cards) >= 0).all()
    r, b = pkg.banding(cfg.m, cfg.tau)
    with Selector(0) as sel:
        sel.attach(hll_t, aux_t, cards_t)
        got = sel.run(cfg.tau, MODE_CB_SMH, r, b)
    want, _ = oracle.select(hll, aux, cards, cfg.tau, r, b)
    assert_same_pairs(got, want)
    assert len(got) > 100


def test_result_frames_on_the_device(oracle):
    """the device-side result hand-over used for the RCCL gather: copy_results_to, the framed copy after finish, and the
    framed copy enqueued behind a pass that is still running (count taken from the device-side counter)"""
    import torch
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.8)
    want, _ = oracle.select(hll, aux, cards, 0.8, r, b)
    assert len(want) > 10

    def unpack(frame):
        rec = frame.cpu().numpy()
        cnt = int(rec[0, 0])
        got = rec[1:1 + cnt].reshape(-1).view(pkg.PAIR_DTYPE)
        return got[np.lexsort((got["k"], got["i"]))]

    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        sel.run(0.8, MODE_CB_SMH, r, b, fetch=False)
        cap = len(want) + 7
        frame = torch.zeros((cap + 1, 2), dtype=torch.int64, device="cuda:0")
        assert sel.copy_results_framed(frame) == len(want)
        torch.cuda.synchronize()
        assert_same_pairs(unpack(frame), want)
        flat = torch.zeros((cap, 2), dtype=torch.int64, device="cuda:0")
        assert sel.copy_results_to(flat) == len(want)
        # behind a running pass
        frame2 = torch.zeros((cap + 1, 2), dtype=torch.int64, device="cuda:0")
        sel.run_async(0.8, MODE_CB_SMH, r, b)
        sel.copy_results_framed_async(frame2)
        sel.finish()
        assert sel.last_attempts() == 1 and sel.result_count() == len(want)
        torch.cuda.synchronize()
        assert_same_pairs(unpack(frame2), want)
        # a frame that is too small reports the true count in its header; the caller sees count > capacity
        small = torch.zeros((4, 2), dtype=torch.int64, device="cuda:0")
        sel.run_async(0.8, MODE_CB_SMH, r, b)
        sel.copy_results_framed_async(small)
        sel.finish()
        torch.cuda.synchronize()
        assert int(small.cpu().numpy()[0, 0]) == len(want) and sel.result_count() > 3


def test_event_timing_levels(oracle):
    """selhip_ctx_timing: level 1 times every kernel scope, level 2 only the dominant stage-1 kernel ("join" for the signature
    algorithms, "stage1" for the stream kernel), 0 nothing; figures are per pass"""
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n300_m512"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, 0.8)
    with Selector(0) as sel:
        sel.upload(hll, aux, cards)
        sel.run(0.8, MODE_SMH, r, b, fetch=False)
        sel.timing(2)
        for _ in range(3):
            sel.run(0.8, MODE_SMH, r, b, fetch=False)
        assert sel.kernel_ms("join") > 0 and sel.kernel_launches("join") == 1.0
        assert sel.kernel_ms("hist") < 0 and sel.kernel_ms("select") < 0 and sel.kernel_ms("total") < 0
        sel.timing(2)
        for _ in range(2):
            sel.run(0.8, MODE_SMH, r, b, algo=ALGO_STREAM, fetch=False)
        assert sel.kernel_ms("stage1") > 0 and sel.kernel_ms("join") < 0
        sel.timing(1)
        for _ in range(2):
            sel.run(0.8, MODE_SMH, r, b, fetch=False)
        for name in ("sigbuild", "join", "verify", "stage1", "group", "hist", "select", "total"):
            assert sel.kernel_ms(name) > 0, name
        assert sel.kernel_ms("total") >= sel.kernel_ms("join") + sel.kernel_ms("hist")
        sel.timing(0)
        sel.run(0.8, MODE_SMH, r, b, fetch=False)
        assert sel.kernel_ms("join") < 0


def test_unsorted_cards_rejected(oracle):
    cfg = SynthConfig("unsorted", 50, 128, 0.9, 5, n_sh_lo=3000, n_sh_hi=3000)
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    with Selector(0) as sel:
        with pytest.raises(pkg.SelhipError):
            sel.upload(hll, aux, cards[::-1].copy())


def test_rectangular_pass_and_out_of_core_driver(oracle):
    """(1) a pass restricted to candidates k >= k_min evaluates exactly rows x [k_min, n); (2) the
    out-of-core driver -- sketches in host memory, block pairs (I, J) uploaded in turn -- returns the in-core result
    (pairs, Jaccard bits, evaluated / survivor counts) for every block size, CB on and off, every algorithm, one and two
    streams, a set with empty sketches, and the auxiliary-HLL criteria"""
    cfg = make_golden.GOLDEN_SYNTH["synth_spread_n600_m64"]
    hll, aux, cards, _, aux_hll = sorted_set(cfg, oracle)
    tau = 0.5
    r, b = pkg.banding(cfg.m, tau)
    n = len(cards)
    for use_cb, mode in ((True, MODE_CB_SMH), (False, MODE_SMH)):
        want, st = oracle.select(hll, aux, cards, tau, r, b, use_cb=use_cb)
        assert len(want) > 50
        with Selector(0) as sel:
            sel.upload(hll, aux, cards)
            for algo in (ALGO_AUTO, ALGO_STREAM, ALGO_HASHJOIN):
                for k_min, rows in ((0, (0, n)), (200, (0, 200)), (333, (100, 333)), (n, (0, n)), (64, (0, 500))):
                    sel.set_candidate_begin(k_min)
                    got = sel.run(tau, mode, r, b, rows=rows, algo=algo)
                    ref = want[(want["i"] >= rows[0]) & (want["i"] < rows[1]) & (want["k"] >= k_min)]
                    assert_same_pairs(got, ref)
            with pytest.raises(pkg.SelhipError):
                sel.set_candidate_begin(n + 1)
            sel.upload(hll, aux, cards)                      # a new upload resets the restriction
            assert_same_pairs(sel.run(tau, mode, r, b), want)
        for block, streams, algo in ((128, 2, ALGO_AUTO), (100, 1, ALGO_AUTO), (599, 2, ALGO_STREAM), (600, 1, ALGO_AUTO), (5000, 2, ALGO_AUTO),
                                     (64, 3, ALGO_HASHJOIN), (250, 2, ALGO_SIG)):
            got, s = pkg.ooc_select(hll, aux, cards, tau, block, mode, r, b, algo=algo, n_streams=streams)
            assert_same_pairs(got, want)
            assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (block, streams, s, st)
    # empty sketches at the front (cardinality 0: skipped as candidates, selection.cpp:281) across block boundaries
    hll0, aux0 = hll.copy(), aux.copy()
    hll0[:150] = 0
    cards0 = oracle.cards(hll0)
    assert (cards0[:150] == 0).all() and (np.diff(cards0) >= 0).all()
    want0, st0 = oracle.select(hll0, aux0, cards0, tau, r, b)
    got0, s0 = pkg.ooc_select(hll0, aux0, cards0, tau, 100, MODE_CB_SMH, r, b)
    assert_same_pairs(got0, want0)
    assert s0["evaluated"] == st0["evaluated"]
    # auxiliary-HLL criteria through the driver
    for crit, code in (("hll_a", pkg.CRIT_HLL_A), ("hll_an", pkg.CRIT_HLL_AN), ("two-stage", pkg.CRIT_HLL_A_SMH_A)):
        wantc, _ = oracle.select(hll, aux, cards, tau, r, b, criterion={"hll_a": 1, "hll_an": 2, "two-stage": 3}[crit], aux_hll=aux_hll, p_aux=cfg.p_aux)
        gotc, _ = pkg.ooc_select(hll, aux, cards, tau, 170, MODE_CB_SMH, r, b, criterion=code, aux_hll=aux_hll, p_aux=cfg.p_aux)
        assert_same_pairs(gotc, wantc)
    with pytest.raises(pkg.SelhipError):
        pkg.ooc_select(hll, aux, cards[::-1].copy(), tau, 100, MODE_CB_SMH, r, b)          # not sorted
    with pytest.raises(pkg.SelhipError):
        pkg.ooc_select(hll, aux, cards, tau, 0, MODE_CB_SMH, r, b)                          # bad block size


def test_multi_device_entry_and_rccl_gather(oracle):
    """selhip_multi_select (device list, one process): RCCL all_gather with one rank, host merge with several
    contexts sharing the one card, RCCL-or-host fallback when RCCL refuses duplicate devices"""
    from cuda_selection_criteria_amd.selection import multi_select
    cfg = make_golden.GOLDEN_SYNTH["synth_flat_n1000_m256"]
    hll, aux, cards, _, _ = sorted_set(cfg, oracle)
    r, b = pkg.banding(cfg.m, cfg.tau)
    want, st = oracle.select(hll, aux, cards, cfg.tau, r, b)
    for devices, gather in (([0], 1), ([0], 0), ([0, 0, 0], 0), ([0, 0], 2)):
        got, s = multi_select(devices, hll, aux, cards, cfg.tau, MODE_CB_SMH, r, b, gather=gather)
        assert_same_pairs(got, want)
        assert s["evaluated"] == st["evaluated"] and s["survivors"] == st["survivors"], (devices, gather, s, st)
    # every criterion goes through the device-list entry (round 1: smh_a only): hll_a, hll_an, and the two-stage criterion of
    # BASELINE configs[4], against the oracle; three contexts on the one card, host merge
    cfgx = SynthConfig("multi-aux", 500, 256, 0.9, 0x77, p_aux=8)
    hllx, auxx, cardsx, _, ahx = sorted_set(cfgx, oracle)
    rx, bx = pkg.banding(cfgx.m, cfgx.tau)
    for crit in (pkg.CRIT_HLL_A, pkg.CRIT_HLL_AN, pkg.CRIT_HLL_A_SMH_A):
        want_c, st_c = oracle.select(hllx, auxx, cardsx, cfgx.tau, rx, bx, criterion=crit, aux_hll=ahx, p_aux=8)
        for devices in ([0], [0, 0, 0]):
            got, s = multi_select(devices, hllx, auxx, cardsx, cfgx.tau, MODE_CB_SMH, rx, bx, gather=0, criterion=crit, aux_hll=ahx, p_aux=8)
            assert_same_pairs(got, want_c)
            assert s["evaluated"] == st_c["evaluated"], (crit, devices, s, st_c)
    with pytest.raises(pkg.SelhipError):
        multi_select([0], hllx, auxx, cardsx, cfgx.tau, MODE_CB_SMH, rx, bx, gather=0, criterion=pkg.CRIT_HLL_A)     # no auxiliary sketches
    with pytest.raises(pkg.SelhipError):
        multi_select([0, 0], hll, aux, cards, cfg.tau, MODE_CB_SMH, r, b, gather=1)      # RCCL required, duplicate GPU
    with pytest.raises(pkg.SelhipError):
        multi_select([7], hll, aux, cards, cfg.tau, MODE_CB_SMH, r, b, gather=0)         # no such device
    # RCCL with the replicas completed by an in-place all-gather (one rank here: its slice is the whole set), every criterion
    for crit in (pkg.CRIT_HLL_A, pkg.CRIT_HLL_A_SMH_A):
        want_c, _ = oracle.select(hllx, auxx, cardsx, cfgx.tau, rx, bx, criterion=crit, aux_hll=ahx, p_aux=8)
        got, _ = multi_select([0], hllx, auxx, cardsx, cfgx.tau, MODE_CB_SMH, rx, bx, gather=1, criterion=crit, aux_hll=ahx, p_aux=8)
        assert_same_pairs(got, want_c)
    # a device thread that fails -- before the replica all-gather, before its pass, before the gather of the records -- must end the
    # call with an error, never leave the other threads waiting in a collective (no ncclCommAbort was bound)
    import os
    for devices, gather in (([0], 1), ([0, 0, 0], 0)):
        for stage in ("upload", "run", "gather"):
            os.environ["SELHIP_TEST_FAIL"] = f"{len(devices) - 1}:{stage}"
            try:
                with pytest.raises(pkg.SelhipError, match="test hook"):
                    multi_select(devices, hll, aux, cards, cfg.tau, MODE_CB_SMH, r, b, gather=gather)
            finally:
                del os.environ["SELHIP_TEST_FAIL"]
            got, _ = multi_select(devices, hll, aux, cards, cfg.tau, MODE_CB_SMH, r, b, gather=gather)      # and the next call is fine
            assert_same_pairs(got, want)
    unsorted = cards.copy(); unsorted[[3, 4]] = unsorted[[4, 3]]
    if unsorted[3] != unsorted[4]:
        with pytest.raises(pkg.SelhipError, match="ascending"):
            multi_select([0], hll, aux, unsorted, cfg.tau, MODE_CB_SMH, r, b, gather=1)
