"""The C++ host programs: `selection` (drop-in CLI of src/selection_cuda.cpp / README.md:60-66) and
`time_smh_hip` (experiments/src/time_smh_cuda.cpp counterpart)."""
import subprocess

import pytest

from conftest import GOLDEN, ROOT

BIN = ROOT / "cuda_selection_criteria_amd" / "bin"
EXP = GOLDEN / "expected"


def test_usage_without_gpu():
    out = subprocess.run([str(BIN / "selection"), "-x"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("Usage: -l -h -a -b")
    out = subprocess.run([str(BIN / "time_smh_hip"), "-x"], capture_output=True, text=True)
    assert out.returncode == 0 and "Usage" in out.stdout


def test_missing_list_is_an_error():
    out = subprocess.run([str(BIN / "selection"), "-l", "/nonexistent/list.txt", "-h", "0.9", "-a", "512"], capture_output=True, text=True)
    assert out.returncode != 0 and "No valid input file provided" in out.stderr     # selection.cpp:48-52


@pytest.mark.gpu
@pytest.mark.parametrize("a,h", [(32, "0.9"), (512, "0.01"), (2048, "0.9"), (4096, "0.01"), (8192, "0.5")])
def test_selection_cli_matches_reference_stdout(a, h):
    for flag, flavour in (("1", "fma"), ("0", "nofma")):
        out = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", h, "-a", str(a), "-b", "128", "-F", flag],
                             cwd=GOLDEN, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout == (EXP / f"influenza_smh_a_a{a}_h{h}.{flavour}.txt").read_text()
    if h == "0.9":
        want = (GOLDEN / "results_reference.txt").read_text().replace("datasets/test_influenzaA/", "influenza/")
        assert out.stdout == want
    # the stage-1 algorithms are interchangeable: -A stream / sig / hashjoin (sort-based) print the same lines
    for algo in ("stream", "sig", "hashjoin"):
        alt = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", h, "-a", str(a), "-b", "128", "-F", "0", "-A", algo],
                             cwd=GOLDEN, capture_output=True, text=True)
        if algo == "sig" and alt.returncode != 0:                # (an explicit -A sig refuses band shapes the signature join does not cover)
            assert "ALGO_SIG needs" in alt.stderr
            continue
        assert alt.returncode == 0 and alt.stdout == out.stdout, (algo, alt.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize("crit", ["hll_a", "hll_an"])
def test_selection_cli_hll_criteria(crit):
    for h in ("0.9", "0.01"):
        out = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", h, "-a", "256", "-c", crit],
                             cwd=GOLDEN, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout == (EXP / f"influenza_{crit}_a256_h{h}.fma.txt").read_text()


@pytest.mark.gpu
def test_time_smh_hip_records():
    out = subprocess.run([str(BIN / "time_smh_hip"), "-l", "influenza_filelist.txt", "-D", "-h", "0.9", "-m", "256", "-b", "256"],
                         cwd=GOLDEN, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    labels = [l.split(";")[1] for l in lines]
    assert labels == ["build_smh", "smh_a", "CB+smh_a"]
    assert ";r:16_b:16;" in lines[1] and "pairs:45;" in lines[1] and "selected:7;" in lines[2]
    out = subprocess.run([str(BIN / "time_smh_hip"), "-N", "2000", "-h", "0.9", "-m", "256"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "synthetic_N2000;smh_a;0.9;" in out.stdout and "pairs:1999000;" in out.stdout


@pytest.mark.gpu
def test_time_smh_hip_rebuilds_superminhash_from_fasta(tmp_path):
    """`time_smh_hip -l` as experiments/src/time_smh_cuda.cpp:181-211 does it: .hll from disk, SuperMinHash rebuilt from the FASTA
    (on the GPU).  The influenza genomes, FASTA and .hll side by side as the reference expects; the rebuilt sketches must select
    what the .smh files on disk select (same record counts as the -D run) -- for m = 256 and for an m that is rounded up (100 -> 128)"""
    import shutil
    names = []
    for f in sorted((GOLDEN / "influenza_fasta").glob("*.fna.gz")):
        shutil.copy(f, tmp_path / f.name)
        shutil.copy(GOLDEN / "influenza" / (f.name + ".hll"), tmp_path / (f.name + ".hll"))
        names.append(f.name)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    disk = subprocess.run([str(BIN / "time_smh_hip"), "-l", "influenza_filelist.txt", "-D", "-h", "0.01", "-m", "256"], cwd=GOLDEN, capture_output=True, text=True)
    assert disk.returncode == 0, disk.stderr
    out = subprocess.run([str(BIN / "time_smh_hip"), "-l", "list.txt", "-h", "0.01", "-m", "256"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines, dlines = out.stdout.strip().splitlines(), disk.stdout.strip().splitlines()
    assert [l.split(";")[1] for l in lines] == ["build_smh", "smh_a", "CB+smh_a"]
    for a, b in zip(lines[1:], dlines[1:]):           # same banding, pairs, survivors, selected as with the files the reference wrote
        assert a.split(";")[4:8] == b.split(";")[4:8], (a, b)
    out = subprocess.run([str(BIN / "time_smh_hip"), "-l", "list.txt", "-h", "0.5", "-m", "100"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode == 0 and ";r:" in out.stdout and "pairs:45;" in out.stdout, out.stderr
    (tmp_path / "bad.txt").write_text("no_such_genome.fna\n")
    out = subprocess.run([str(BIN / "time_smh_hip"), "-l", "bad.txt", "-h", "0.5", "-m", "64"], cwd=tmp_path, capture_output=True, text=True)
    assert out.returncode != 0


@pytest.mark.gpu
def test_selection_cli_out_of_core_and_result_file(tmp_path):
    """-B (sketches in host memory, block pairs uploaded in turn) prints the same lines; -o writes the binary result file
    and -r turns it back into those lines"""
    want = (EXP / "influenza_smh_a_a512_h0.01.fma.txt").read_text()
    for block in ("3", "4", "10", "1000"):
        out = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", "0.01", "-a", "512", "-B", block],
                             cwd=GOLDEN, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert out.stdout == want
    out = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", "0.01", "-a", "256", "-c", "hll_a", "-B", "4"],
                         cwd=GOLDEN, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == (EXP / "influenza_hll_a_a256_h0.01.fma.txt").read_text()
    f = tmp_path / "sel.selr"
    out = subprocess.run([str(BIN / "selection"), "-l", "influenza_filelist.txt", "-h", "0.01", "-a", "512", "-o", str(f)],
                         cwd=GOLDEN, capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == "", out.stderr
    out = subprocess.run([str(BIN / "selection"), "-r", str(f)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == want


@pytest.mark.gpu
def test_selection_cli_tie_order_matches_the_reference_binary(tmp_path):
    """duplicated sketches (equal cardinalities, distinct names): `bin/selection` and the Python driver print the reference binary's
    lines, tie order included (tests/golden/expected/ties_*, made by oracle/_ref/selection; see test_oracle_golden.py)"""
    import sys
    sys.path.insert(0, str(GOLDEN))
    import make_golden
    import cuda_selection_criteria_amd as pkg
    make_golden.write_tie_files(tmp_path)
    for crit, a, h in make_golden.TIES_CASES:
        for flag, flavour in (("1", "fma"), ("0", "nofma")):
            want = (EXP / f"ties_{crit}_a{a}_h{h}.{flavour}.txt").read_text()
            out = subprocess.run([str(BIN / "selection"), "-l", "list.txt", "-h", h, "-a", str(a), "-c", crit, "-F", flag],
                                 cwd=tmp_path, capture_output=True, text=True)
            assert out.returncode == 0, out.stderr
            assert out.stdout == want, (crit, a, h, flavour)
    import os
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        got = pkg.select_from_filelist("list.txt", 0.9, 1024)
        assert got == (EXP / "ties_smh_a_a1024_h0.9.fma.txt").read_text()
    finally:
        os.chdir(cwd)


@pytest.mark.gpu
def test_comparison_and_timing_experiments(tmp_path):
    """scripts/experiments.py: the reference's CPU-vs-GPU comparison method (run_comparison_experiment.sh:57-112: tau = 0.01, keyed
    join of the two outputs, |sim_cpu - sim_gpu| with eps 1e-6) as a runnable artefact -- every pair on both sides, diff 0 -- and
    its timing CSV (run_time_experiment.sh)"""
    import csv
    import sys
    out = tmp_path / "cmp.csv"
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "experiments.py"), "compare", "-l", "influenza_filelist.txt", "-a", "512", "4096",
                        "-h", "0.01", "-o", str(out)], cwd=GOLDEN, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = list(csv.DictReader(open(out)))
    assert len(rows) == 27 + 41 and all(x["diff"] == "0" for x in rows)          # 27 pairs at m=64, 41 at m=512
    out2 = tmp_path / "time.csv"
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "experiments.py"), "time", "-l", "influenza_filelist.txt", "-m", "256", "-h", "0.9",
                        "-t", "2", "-o", str(out2)], cwd=GOLDEN, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rows = list(csv.DictReader(open(out2)))
    assert {(x["impl"], x["criterio"]) for x in rows} >= {("gpu", "smh_a"), ("gpu", "CB+smh_a"), ("cpu", "smh_a"), ("cpu", "CB+smh_a")}
