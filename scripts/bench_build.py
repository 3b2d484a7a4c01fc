#!/usr/bin/env python3
"""Measurement for the sketch-construction row: k-mers/s of selhip_build_sketches on synthetic
genomes resident in HBM, with the reference's own build_sketch (oracle/_ref, if present) and the oracle timed beside it."""
import gzip, json, os, random, subprocess, sys, tempfile, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
import cuda_selection_criteria_amd as pkg
from cuda_selection_criteria_amd._lib import check

n_genomes = int(os.environ.get("NG", 256)); L = int(os.environ.get("LEN", 1_000_000)); m = int(os.environ.get("M", 512))
lib = pkg.hip_lib()
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
codes = torch.randint(0, 4, (n_genomes, L + 1), dtype=torch.uint8, generator=g)
codes[:, 0] = 4                                     # record start
d_codes = codes.to(dev).contiguous()
d_off = (torch.arange(n_genomes + 1, dtype=torch.int64) * (L + 1)).to(dev)
d_hll = torch.zeros((n_genomes, 16384), dtype=torch.uint8, device=dev)
d_smh = torch.zeros((n_genomes, m), dtype=torch.int64, device=dev)
def run():
    check(lib.selhip_build_sketches(d_codes.data_ptr(), d_off.data_ptr(), n_genomes, 31, m, 0, d_hll.data_ptr(), d_smh.data_ptr(), None, None))
for _ in range(2): run()
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
ev0.record()
for _ in range(reps): run()
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / reps
kmers = n_genomes * (L - 30)
out = {"workload": f"{n_genomes} random genomes x {L} bases, k=31, HLL p=14 + SuperMinHash m={m}", "gpu_ms": ms,
       "gpu_kmers_per_s": kmers / (ms * 1e-3), "bytes_per_kmer_algorithmic": 1, "achieved_GBps": n_genomes * (L + 1) / (ms * 1e-3) / 1e9}
# CPU: the oracle (sequential restatement) and the reference's own program on a few of the same genomes
ns = 4
with tempfile.TemporaryDirectory() as td:
    names = []
    letters = np.array(list("ACGT"))
    for j in range(ns):
        seq = "".join(letters[codes[j, 1:].numpy()])
        with gzip.open(f"{td}/g{j}.fna.gz", "wt", compresslevel=1) as f:
            f.write(f">g{j}\n{seq}\n")
        names.append(f"g{j}.fna.gz")
    Path(td, "list.txt").write_text("\n".join(names) + "\n")
    import oracle_py
    bo = oracle_py.BuildOracle()
    t = time.perf_counter(); res = [bo.sketch(f"{td}/{nm}", m=m) for nm in names]; dt = time.perf_counter() - t
    out["cpu_oracle_kmers_per_s_1thread"] = ns * (L - 30) / dt
    assert all(np.array_equal(res[j][0], d_hll[j].cpu().numpy()) and np.array_equal(res[j][2], d_smh[j].cpu().numpy().view(np.uint64)) for j in range(ns))
    out["parity_vs_oracle"] = True
    ref = ROOT / "oracle" / "_ref" / "build_sketch"
    if ref.exists():
        t = time.perf_counter()
        subprocess.run([str(ref), "-l", "list.txt", "-t", str(ns), "-a", str(m * 8), "-c", "smh_a"], cwd=td, check=True, capture_output=True)
        dt = time.perf_counter() - t
        out["cpu_reference"] = {"kmers_per_s": ns * (L - 30) / dt, "threads": ns, "note": "oracle/_ref/build_sketch (HLL pass + SuperMinHash pass, gz FASTA through SeqAn), wall"}
print(json.dumps(out))
