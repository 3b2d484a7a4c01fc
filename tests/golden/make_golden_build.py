#!/usr/bin/env python3
"""Golden vectors for the sketch-construction step, made by the REFERENCE's own
build_sketch (oracle/_ref/build_sketch, compiled from /root/reference/src/build_sketch.cpp) on small synthetic
FASTA files that exercise what the influenza genomes do not: N runs, lower case, IUPAC codes, records shorter than
k, empty lines, many records, repeated sequence (duplicates), tiny inputs (buckets that stay empty).
Inputs AND outputs are committed under tests/golden/synth_fasta/ (authoring container only)."""
import gzip
import random
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF = HERE.parent.parent / "oracle" / "_ref" / "build_sketch"
OUT = HERE / "synth_fasta"


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def wrap(s, w=70):
    return "\n".join(s[i:i + w] for i in range(0, len(s), w))


def make_cases():
    rng = random.Random(20240607)
    cases = {}
    a = rand_seq(rng, 6000)
    a = a[:1000] + "N" * 7 + a[1000:2500] + "NNN" + a[2500:3000].lower() + "R" + a[3000:4000] + "n" + a[4000:]
    cases["mixed_case_N"] = f">r1 mixed\n{wrap(a)}\n>r2 short\n{rand_seq(rng, 25)}\n>r3 exact_k\n{rand_seq(rng, 31)}\n>r4\n{wrap(rand_seq(rng, 1500), 61)}\n"
    cases["tiny"] = f">only\n{rand_seq(rng, 40)}\n"                                   # 10 k-mers: almost every bucket stays empty
    cases["small_many_records"] = "".join(f">s{i}\n{wrap(rand_seq(rng, rng.randint(20, 90)))}\n" for i in range(60))
    rep = rand_seq(rng, 200)
    cases["repeats"] = f">rep\n{wrap(rep * 40)}\n\n>rep_rc\n{wrap(rep[::-1] * 3)}\n"
    cases["medium"] = f">m\n{wrap(rand_seq(rng, 30000))}\n"
    cases["polyA"] = f">pa\n{wrap('A' * 500)}\n>pt\n{wrap('T' * 100)}\n"
    return cases


def main():
    if not REF.exists():
        sys.exit("oracle/_ref/build_sketch missing: run `make -C oracle ref`")
    OUT.mkdir(exist_ok=True)
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        names = []
        for name, text in make_cases().items():
            with gzip.open(td / f"{name}.fna.gz", "wt") as f:
                f.write(text)
            names.append(f"{name}.fna.gz")
        (td / "list.txt").write_text("\n".join(names) + "\n")
        for a in (32, 512, 2048, 8192):
            subprocess.run([str(REF), "-l", "list.txt", "-t", "2", "-a", str(a), "-c", "smh_a"], cwd=td, check=True, capture_output=True)
        for a in (16, 256):
            subprocess.run([str(REF), "-l", "list.txt", "-t", "2", "-a", str(a), "-c", "hll_a"], cwd=td, check=True, capture_output=True)
        for f in sorted(td.iterdir()):
            if f.name != "list.txt":
                shutil.copy(f, OUT / f.name)
    print("wrote", len(list(OUT.iterdir())), "files to", OUT)


if __name__ == "__main__":
    main()
