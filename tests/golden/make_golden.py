#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference and `make -C oracle ref`): it executes the
reference's own programs compiled from the sources where they lie (oracle/_ref/selection,
oracle/_ref/selection_nofma, oracle/_ref/hll_kat[_nofma]) on
  * the influenza fixtures (tests/golden/influenza/: the sketch files the reference ships in
    datasets/test_influenzaA plus .smh16/.smh256/.smh512/.smh1024 produced by the reference's
    build_sketch, oracle/_ref/build_sketch, from the same genomes), and
  * synthetic sketch sets written to a scratch directory in the reference's on-disk format by
    libselhost (the generator is deterministic, so tests regenerate the inputs instead of storing them),
and stores stdout as text.  Nothing from the reference's source travels: fixtures are inputs + outputs.
"""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
REF = ROOT / "oracle" / "_ref"

import cuda_selection_criteria_amd as pkg  # noqa: E402
from cuda_selection_criteria_amd.synth import SynthConfig  # noqa: E402

# synthetic golden sets (small n_sh keeps generation fast; statistics are scale free)
GOLDEN_SYNTH = {
    "synth_flat_n1000_m256": SynthConfig("golden flat", 1000, 256, 0.9, 0x5EED0001, p_aux=8, n_sh_lo=20000, n_sh_hi=20000),
    "synth_spread_n600_m64": SynthConfig("golden spread", 600, 64, 0.8, 0x5EED0011, p_aux=8, mode=1, n_sh_lo=3000, n_sh_hi=60000),
    "synth_flat_n300_m512": SynthConfig("golden m512", 300, 512, 0.8, 0x5EED0002, p_aux=8, n_sh_lo=20000, n_sh_hi=20000),
    "synth_flat_n200_m1024": SynthConfig("golden m1024", 200, 1024, 0.9, 0x5EED0004, p_aux=8, n_sh_lo=20000, n_sh_hi=20000),
    "synth_flat_n200_m128": SynthConfig("golden m128", 200, 128, 0.9, 0x5EED0005, p_aux=8, n_sh_lo=20000, n_sh_hi=20000),
}


def write_synth_files(cfg, directory: Path):
    """writes g<idx>.hll / .smh<m> / .hll_<p_aux> in the reference's formats; returns the list file"""
    host = pkg.host_lib()
    hll, aux, aux_hll = pkg.synth_host(cfg)
    names = []
    for g in range(cfg.n_genomes):
        base = directory / f"g{g:06d}"
        assert host.selhost_write_hll(str(base).encode() + b".hll", hll[g].ctypes.data, 14) == 0
        assert host.selhost_write_smh(str(base).encode() + f".smh{cfg.m}".encode(), aux[g].ctypes.data, cfg.m) == 0
        if cfg.p_aux:
            assert host.selhost_write_hll(str(base).encode() + f".hll_{cfg.p_aux}".encode(), aux_hll[g].ctypes.data, cfg.p_aux) == 0
        names.append(f"g{g:06d}")
    lst = directory / "list.txt"
    lst.write_text("\n".join(names) + "\n")
    return lst


# ---- tie set: duplicated sketches => exactly equal cardinalities under distinct names.  selection.cpp:251-256 sorts with the
# UNSTABLE std::sort, so which of two equal genomes ranks first (hence prints first) is decided by libstdc++'s introsort;
# 60 entries (> the 16-element insertion-sort threshold) make the quicksort phase matter.
TIES_BASE = SynthConfig("ties base", 12, 128, 0.9, 0x5EED0071, p_aux=8, n_sh_lo=20000, n_sh_hi=20000)
TIES_COPIES = 5
TIES_CASES = [("smh_a", 1024, "0.9"), ("smh_a", 1024, "0.5"), ("hll_a", 256, "0.9"), ("hll_an", 256, "0.9")]


def ties_order():
    """list position -> base genome (seeded shuffle of TIES_COPIES copies of every base genome)"""
    order = np.repeat(np.arange(TIES_BASE.n_genomes), TIES_COPIES)
    np.random.default_rng(0x71E5).shuffle(order)
    return order


def write_tie_files(directory: Path):
    """t<idx>.hll / .smh128 / .hll_8 (copies of the base genomes' sketches, in the reference's formats); returns the list file"""
    host = pkg.host_lib()
    cfg = TIES_BASE
    hll, aux, aux_hll = pkg.synth_host(cfg)
    names = []
    for idx, g in enumerate(ties_order()):
        base = directory / f"t{idx:03d}"
        assert host.selhost_write_hll(str(base).encode() + b".hll", hll[g].ctypes.data, 14) == 0
        assert host.selhost_write_smh(str(base).encode() + f".smh{cfg.m}".encode(), aux[g].ctypes.data, cfg.m) == 0
        assert host.selhost_write_hll(str(base).encode() + f".hll_{cfg.p_aux}".encode(), aux_hll[g].ctypes.data, cfg.p_aux) == 0
        names.append(f"t{idx:03d}")
    lst = directory / "list.txt"
    lst.write_text("\n".join(names) + "\n")
    return lst


def run(binary, args, cwd):
    return subprocess.run([str(REF / binary)] + args, cwd=cwd, check=True, capture_output=True, text=True).stdout


def main():
    if not (REF / "selection").exists():
        sys.exit("oracle/_ref/selection missing: run `make -C oracle ref` in the authoring container")
    exp = HERE / "expected"
    exp.mkdir(exist_ok=True)
    # ---- influenza: reference stdout for every criterion / size the fixtures allow -------------------
    cases = [("smh_a", a, h) for a in (32, 128, 512, 2048, 4096, 8192) for h in ("0.9", "0.8", "0.5", "0.01")]
    cases += [(c, 256, h) for c in ("hll_a", "hll_an") for h in ("0.9", "0.5", "0.01")]
    for flavour, binary in (("fma", "selection"), ("nofma", "selection_nofma")):
        for crit, a, h in cases:
            out = run(binary, ["-l", "influenza_filelist.txt", "-t", "4", "-c", crit, "-a", str(a), "-h", h], HERE)
            (exp / f"influenza_{crit}_a{a}_h{h}.{flavour}.txt").write_text(out)
    files = [l.strip() for l in (HERE / "influenza_filelist.txt").read_text().splitlines() if l.strip()]
    for flavour, binary in (("fma", "hll_kat"), ("nofma", "hll_kat_nofma")):
        for suffix in (".hll", ".hll_8"):
            out = run(binary, [suffix] + files, HERE)
            (exp / f"influenza_kat{suffix.replace('.', '_')}.{flavour}.txt").write_text(out)
    # ---- synthetic sets --------------------------------------------------------------------------------
    for name, cfg in GOLDEN_SYNTH.items():
        with tempfile.TemporaryDirectory() as td:
            td = Path(td)
            write_synth_files(cfg, td)
            a = cfg.m * 8
            for flavour, binary in (("fma", "selection"), ("nofma", "selection_nofma")):
                for h in sorted({f"{cfg.tau}", "0.5"}):
                    out = run(binary, ["-l", "list.txt", "-t", "8", "-c", "smh_a", "-a", str(a), "-h", h], td)
                    (exp / f"{name}_smh_a_h{h}.{flavour}.txt").write_text(out)
                out = run(binary, ["-l", "list.txt", "-t", "8", "-c", "hll_a", "-a", "256", "-h", f"{cfg.tau}"], td)
                (exp / f"{name}_hll_a_h{cfg.tau}.{flavour}.txt").write_text(out)
            # estimator KATs on the first 24 genomes (p=14 and p=8)
            first = [f"g{g:06d}" for g in range(24)]
            for flavour, binary in (("fma", "hll_kat"), ("nofma", "hll_kat_nofma")):
                (exp / f"{name}_kat_hll.{flavour}.txt").write_text(run(binary, [".hll"] + first, td))
                (exp / f"{name}_kat_hll_8.{flavour}.txt").write_text(run(binary, [".hll_8"] + first, td))
        print("golden:", name)
    # ---- tie order of the reference's std::sort ------------------------------------------------------------
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        write_tie_files(td)
        for flavour, binary in (("fma", "selection"), ("nofma", "selection_nofma")):
            for crit, a, h in TIES_CASES:
                out = run(binary, ["-l", "list.txt", "-t", "4", "-c", crit, "-a", str(a), "-h", h], td)
                (exp / f"ties_{crit}_a{a}_h{h}.{flavour}.txt").write_text(out)
    print("golden: ties")
    print("done ->", exp)


if __name__ == "__main__":
    main()
