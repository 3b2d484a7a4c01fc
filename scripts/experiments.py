#!/usr/bin/env python3
"""The reference's two experiment methods as runnable artefacts (run_comparison_experiment.sh:57-112, run_time_experiment.sh).

  experiments.py compare -l LIST [-a AUX_BYTES ...] [-h TAU] [-o comparison_cpu_gpu.csv] [--cpu oracle|reference]
      runs a CPU program (the oracle CLI, or the reference's own `selection` where oracle/_ref exists) and the MI355X
      `bin/selection` on the same file list, keys every output line by "name1_name2" and writes
          cfg,card1,card2,sim_cpu,sim_gpu,diff          (diff < eps printed as 0, eps = 1e-6: the reference's columns)
      Unlike the reference's `join`, pairs present on one side only are NOT dropped: they are written with an empty
      similarity on the other side, and the exit code is 1 if there is any such pair or any diff >= eps.
  experiments.py time -l LIST | -N GENOMES [-m BUCKETS ...] [-h TAU] [-R REPS] [-t THREADS] [-o experiment_smh.csv]
      impl,threads,mh_size,rep,criterio,tiempo    rows for cpu (oracle library: the OpenMP loop of selection.cpp:270-291,
      modes smh_a and CB+smh_a of time_smh.cpp) and gpu (bin/time_smh_hip records `list;label;tau;seconds`)
"""
import argparse
import csv
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
BIN = ROOT / "cuda_selection_criteria_amd" / "bin"
EPS = 1e-6


def lines_to_map(text):
    out = {}
    for ln in text.splitlines():
        f = ln.split()
        if len(f) == 3:
            out[f"{f[0]}_{f[1]}"] = (f[0], f[1], float(f[2]))
    return out


def compare(args):
    cpu_bin = ROOT / "oracle" / ("_ref/selection" if args.cpu == "reference" else "selection_oracle_cli")
    if not cpu_bin.exists():
        sys.exit(f"{cpu_bin} missing")
    rows, bad = [], 0
    for a in args.a:
        cfg = f"t{args.t}_m{a}"
        cpu = subprocess.run([str(cpu_bin), "-l", args.l, "-t", str(args.t), "-h", args.tau, "-a", str(a), "-c", args.c],
                             capture_output=True, text=True, check=True).stdout
        gpu = subprocess.run([str(BIN / "selection"), "-l", args.l, "-h", args.tau, "-a", str(a), "-b", "128", "-c", args.c],
                             capture_output=True, text=True, check=True).stdout
        mc, mg = lines_to_map(cpu), lines_to_map(gpu)
        for key in sorted(set(mc) | set(mg)):
            c, g = mc.get(key), mg.get(key)
            n1, n2 = (c or g)[0], (c or g)[1]
            if c is None or g is None:
                rows.append([cfg, n1, n2, "" if c is None else c[2], "" if g is None else g[2], "missing"])
                bad += 1
                continue
            d = abs(c[2] - g[2])
            if d < EPS:
                d = 0
            else:
                bad += 1
            rows.append([cfg, n1, n2, c[2], g[2], d])
    with open(args.o, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["cfg", "card1", "card2", "sim_cpu", "sim_gpu", "diff"])
        w.writerows(rows)
    print(f"comparison complete: {len(rows)} pairs in '{args.o}', {bad} mismatching")
    return 1 if bad else 0


def timing(args):
    import numpy as np
    import oracle_py
    import cuda_selection_criteria_amd as pkg
    orc = oracle_py.Oracle()
    rows = []
    for m in args.m:
        for rep in range(1, args.R + 1):
            # ---- gpu: bin/time_smh_hip prints list;label;tau;seconds
            cmd = [str(BIN / "time_smh_hip"), "-h", args.tau, "-m", str(m), "-b", "256"]
            cmd += ["-l", args.l, "-D"] if args.l else ["-N", str(args.N)]       # -D: sketch files from disk (the CPU side below loads the same files)
            out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout
            for ln in out.splitlines():
                f = ln.split(";")
                if len(f) >= 4 and f[1] in ("build_smh", "smh_a", "CB+smh_a"):
                    rows.append(["gpu", 256, m, rep, f[1], f[3]])
            # ---- cpu: the oracle's OpenMP loop on the same sketches
            if args.l:
                ds = pkg.load_dataset(args.l, m, 0)
                hll, aux, cards = ds.hll, ds.aux, ds.cards
            else:
                cfg = pkg.SynthConfig("time", args.N, m, float(args.tau), 0x5EED0000)
                hll, aux, _ = pkg.synth_host(cfg)
                cards = orc.cards(hll)
                perm = pkg.sort_by_card(cards)
                hll, aux, cards = hll[perm], aux[perm], cards[perm]
            r, b = pkg.banding(m, float(args.tau))
            for label, use_cb in (("smh_a", False), ("CB+smh_a", True)):
                t0 = time.perf_counter()
                orc.select(hll, aux, cards, float(args.tau), r, b, use_cb=use_cb, threads=args.t)
                rows.append(["cpu", args.t, m, rep, label, f"{time.perf_counter() - t0:.6f}"])
    with open(args.o, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["impl", "threads", "mh_size", "rep", "criterio", "tiempo"])
        w.writerows(rows)
    print(f"done, results in {args.o}")
    return 0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter, add_help=False)
    ap.add_argument("--help", action="help")
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("compare", add_help=False)
    c.add_argument("-l", required=True); c.add_argument("-a", type=int, nargs="+", default=[512]); c.add_argument("-h", dest="tau", default="0.01")
    c.add_argument("-t", type=int, default=8); c.add_argument("-c", default="smh_a"); c.add_argument("-o", default="comparison_cpu_gpu.csv")
    c.add_argument("--cpu", choices=["oracle", "reference"], default="oracle")
    t = sub.add_parser("time", add_help=False)
    t.add_argument("-l", default=""); t.add_argument("-N", type=int, default=0); t.add_argument("-m", type=int, nargs="+", default=[512])
    t.add_argument("-h", dest="tau", default="0.9"); t.add_argument("-R", type=int, default=1); t.add_argument("-t", type=int, default=8)
    t.add_argument("-o", default="experiment_smh_comparative.csv")
    args = ap.parse_args()
    if args.cmd == "time" and not args.l and not args.N:
        sys.exit("time: give -l LIST or -N GENOMES")
    sys.exit(compare(args) if args.cmd == "compare" else timing(args))


if __name__ == "__main__":
    main()
