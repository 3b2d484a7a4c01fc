"""Sketch construction (the `build_sketch` row, producer of the on-disk format).
CPU part: the oracle (oracle/build_sketch_oracle.c) against files written by the REFERENCE's build_sketch --
the 40 it ships (datasets/test_influenzaA) plus files made by oracle/_ref/build_sketch for other sizes and for
synthetic FASTA (tests/golden/make_golden_build.py).  GPU part: the HIP build against the same files."""
import gzip
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

FLU_FASTA = sorted((GOLDEN / "influenza_fasta").glob("*.fna.gz"))
SYN_FASTA = sorted((GOLDEN / "synth_fasta").glob("*.fna.gz"))


def read_hll_file(path):
    raw = gzip.open(path, "rb").read()
    hdr = struct.unpack_from("<4I", raw)
    (np_,) = struct.unpack_from("<I", raw, 16)
    (val,) = struct.unpack_from("<d", raw, 20)
    assert hdr == (0, 2, 2, 1) and val == -1.0 and len(raw) == 28 + (1 << np_)
    return np.frombuffer(raw, dtype=np.uint8, offset=28).copy(), np_


def read_smh_file(path):
    raw = gzip.open(path, "rb").read()
    (cnt,) = struct.unpack_from("<I", raw)
    assert len(raw) == 4 + 8 * cnt
    return np.frombuffer(raw[4:], dtype=np.uint64).copy()


def golden_files(fasta):
    """[(kind, param, path)] of every reference-written sketch file that belongs to this FASTA"""
    d = GOLDEN / "influenza" if fasta.parent.name == "influenza_fasta" else fasta.parent
    out = []
    for f in sorted(d.glob(fasta.name + ".*")):
        suf = f.name[len(fasta.name) + 1:]
        if suf == "hll":
            out.append(("hll", 14, f))
        elif suf.startswith("hll_"):
            out.append(("aux", int(suf[4:]), f))
        elif suf.startswith("smh"):
            out.append(("smh", int(suf[3:]), f))
    return out


@pytest.fixture(scope="module")
def build_oracle():
    import oracle_py
    return oracle_py.BuildOracle()


@pytest.mark.parametrize("fasta", FLU_FASTA + SYN_FASTA, ids=lambda p: p.name[:24])
def test_oracle_reproduces_reference_files(build_oracle, fasta):
    files = golden_files(fasta)
    assert len(files) >= 7
    for kind, param, path in files:
        if kind == "hll":
            want, p = read_hll_file(path)
            got, _, _, n = build_oracle.sketch(fasta)
            assert p == 14 and np.array_equal(got, want)
        elif kind == "aux":
            want, p = read_hll_file(path)
            _, got, _, _ = build_oracle.sketch(fasta, p_aux=param)
            assert p == param and np.array_equal(got, want)
        else:
            want = read_smh_file(path)
            _, _, got, _ = build_oracle.sketch(fasta, m=param)
            assert np.array_equal(got, want), (fasta.name, param)
