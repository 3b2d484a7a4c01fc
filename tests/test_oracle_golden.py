"""The oracle (oracle/selection_oracle.c) pinned against outputs of the REFERENCE ITSELF.

tests/golden/expected/*.txt were produced by the reference's own selection.cpp / hll.h compiled from
/root/reference (tests/golden/make_golden.py, oracle/Makefile `ref`), in two builds: `fma`
(g++ -O3 -march=x86-64-v3: what the reference Makefile's -march=native gives on an FMA host) and
`nofma` (same + -ffp-contract=off).  results_reference.txt is the golden output the reference ships.
"""
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

CLI = ROOT / "oracle" / "selection_oracle_cli"
EXP = GOLDEN / "expected"
FLAVOURS = [("fma", 1), ("nofma", 0)]


def cli(args, cwd=GOLDEN):
    return subprocess.run([str(CLI)] + args, cwd=cwd, check=True, capture_output=True, text=True).stdout


def test_reference_results_txt():
    """results.txt (reference repo root): 7 pairs, smh_a m in {4,64}, tau 0.9 -- any aux size gives the same set"""
    want = (GOLDEN / "results_reference.txt").read_text().replace("datasets/test_influenzaA/", "influenza/")
    for a in (32, 512, 2048):
        got = cli(["-l", "influenza_filelist.txt", "-c", "smh_a", "-a", str(a), "-h", "0.9"])
        assert got == want
    assert cli(["-l", "influenza_filelist.txt", "-c", "hll_a", "-a", "256", "-h", "0.9"]) == want


@pytest.mark.parametrize("flavour,fma", FLAVOURS)
def test_influenza_all_cases(flavour, fma):
    files = sorted(EXP.glob(f"influenza_*_a*_h*.{flavour}.txt"))
    assert len(files) >= 30
    for f in files:
        crit_a_h = f.name[len("influenza_"):-len(f".{flavour}.txt")]
        crit, rest = crit_a_h.rsplit("_a", 1)
        a, h = rest.split("_h")
        got = cli(["-l", "influenza_filelist.txt", "-t", "2", "-c", crit, "-a", a, "-h", h, "-F", str(fma)])
        assert got == f.read_text(), f.name


def _read_hll(path):
    import gzip
    import struct
    b = gzip.open(path, "rb").read()
    (np_,) = struct.unpack_from("<I", b, 16)
    return np.frombuffer(b, dtype=np.uint8, offset=28, count=1 << np_).copy(), np_


@pytest.mark.parametrize("flavour,fma", FLAVOURS)
@pytest.mark.parametrize("suffix", [".hll", ".hll_8"])
def test_influenza_estimator_kat(oracle, flavour, fma, suffix):
    """report()/union_size() of the reference's hll_t, hex-exact"""
    files = [l.strip() for l in (GOLDEN / "influenza_filelist.txt").read_text().splitlines() if l.strip()]
    regs = [_read_hll(GOLDEN / (f + suffix)) for f in files]
    oracle.set_fma(fma)
    try:
        n = 0
        for line in (EXP / f"influenza_kat{suffix.replace('.', '_')}.{flavour}.txt").read_text().splitlines():
            t = line.split()
            if t[0] == "R":
                v = oracle.report(regs[int(t[1])][0], regs[int(t[1])][1])
            else:
                i, k = int(t[1]), int(t[2])
                v = oracle.union_size(regs[i][0], regs[k][0], regs[i][1])
            assert v == float.fromhex(t[-1]), line
            n += 1
        assert n == 55
    finally:
        oracle.set_fma(1)


def test_fma_flavours_differ_on_fixtures():
    """the two builds of the reference are NOT bit-identical (documented in DESIGN.md)"""
    a = (EXP / "influenza_kat_hll.fma.txt").read_text()
    b = (EXP / "influenza_kat_hll.nofma.txt").read_text()
    assert a != b


def test_banding_table(oracle):
    # SURVEY.md section 8(a1)/(c4); cross-checked with the r:/b: the reference's time_smh prints
    assert oracle.banding(256, 0.9) == (16, 16)
    assert oracle.banding(512, 0.8) == (8, 64)
    assert oracle.banding(512, 0.9) == (16, 32)
    assert oracle.banding(1024, 0.9) == (16, 64)
    assert oracle.banding(4, 0.5) == (1, 4)
    assert oracle.banding(4, 0.01) == (1, 4)
    assert oracle.banding(4, 0.01, cuda_variant=True) == (4, 1) or oracle.banding(4, 0.01, cuda_variant=True) == (1, 1)


def _synth_dir(tmp_path, name):
    import sys
    sys.path.insert(0, str(GOLDEN))
    import make_golden
    cfg = make_golden.GOLDEN_SYNTH[name]
    make_golden.write_synth_files(cfg, tmp_path)
    return cfg


@pytest.mark.parametrize("name", ["synth_flat_n1000_m256", "synth_spread_n600_m64", "synth_flat_n300_m512",
                                  "synth_flat_n200_m1024", "synth_flat_n200_m128"])
def test_synthetic_sets(tmp_path, oracle, name):
    """oracle stdout == reference stdout on regenerated synthetic sketch files (both flavours, smh_a + hll_a),
    and estimator KATs on the first 24 genomes"""
    cfg = _synth_dir(tmp_path, name)
    for flavour, fma in FLAVOURS:
        for f in sorted(EXP.glob(f"{name}_smh_a_h*.{flavour}.txt")):
            h = f.name.split("_h")[-1][:-len(f".{flavour}.txt")]
            got = cli(["-l", "list.txt", "-c", "smh_a", "-a", str(cfg.m * 8), "-h", h, "-F", str(fma)], cwd=tmp_path)
            assert got == f.read_text(), f.name
        f = EXP / f"{name}_hll_a_h{cfg.tau}.{flavour}.txt"
        got = cli(["-l", "list.txt", "-c", "hll_a", "-a", "256", "-h", str(cfg.tau), "-F", str(fma)], cwd=tmp_path)
        assert got == f.read_text(), f.name
        oracle.set_fma(fma)
        try:
            for suffix, kat in ((".hll", "kat_hll"), (".hll_8", "kat_hll_8")):
                regs = [_read_hll(tmp_path / f"g{g:06d}{suffix}") for g in range(24)]
                for line in (EXP / f"{name}_{kat}.{flavour}.txt").read_text().splitlines():
                    t = line.split()
                    if t[0] == "R":
                        v = oracle.report(regs[int(t[1])][0], regs[int(t[1])][1])
                    else:
                        v = oracle.union_size(regs[int(t[1])][0], regs[int(t[2])][0], regs[int(t[1])][1])
                    assert v == float.fromhex(t[-1]), (name, line)
        finally:
            oracle.set_fma(1)


@pytest.mark.parametrize("flavour,fma", [("fma", 1), ("nofma", 0)])
def test_tie_order_matches_the_reference_binary(tmp_path, flavour, fma):
    """selection.cpp:251-256 sorts by cardinality with the unstable std::sort; with duplicated sketches (equal cardinalities under
    distinct names, 60 list entries) the printed order shows which of two equal genomes ranks first.  The oracle restates GNU
    libstdc++'s introsort (orc_std_sort_perm) and must print what the reference binary printed (tests/golden/expected/ties_*)."""
    import sys
    sys.path.insert(0, str(GOLDEN))
    import make_golden
    make_golden.write_tie_files(tmp_path)
    for crit, a, h in make_golden.TIES_CASES:
        got = cli(["-l", "list.txt", "-c", crit, "-a", str(a), "-h", h, "-F", str(fma)], cwd=tmp_path)
        assert got == (EXP / f"ties_{crit}_a{a}_h{h}.{flavour}.txt").read_text(), (crit, a, h)


def test_tie_set_really_has_ties(oracle):
    import sys
    sys.path.insert(0, str(GOLDEN))
    import make_golden
    order = make_golden.ties_order()
    assert len(order) == 60 and len(set(order.tolist())) == 12
