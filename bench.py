#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: sketch pair-comparisons/sec (N genomes x m buckets).

A "step" = one pass of the hot path (mode "smh_a" of experiments/src/time_smh.cpp:229-257: every pair
i<k goes through the smh_a band predicate, survivors through the HLL-14 union estimate and the Jaccard
test) over one synthetic sketch set that is already resident in HBM, ending with the selected-pair
list of all ranks gathered on every rank (RCCL all_gather over xGMI when --gpus > 1).

Workloads
  default, N = 1        BASELINE.json configs[2]: 10 000 synthetic genomes, smh_a m=512, tau=0.8.
  default, N > 1        the same configuration weak-scaled: genome count x sqrt(N), so that every GPU keeps
                        configs[2]'s pair count ("scaling": "weak").
  --workload cfg4 --scaling strong
                        BASELINE.json configs[3]: 50 000 genomes, m=512, the SAME 1.25e9 pairs sharded over
                        the N GPUs ("scaling": "strong") -- the second scaling line.
  --workload cfg5 [--scaling strong]   configs[4]: 100 000 genomes, hll_a(p=8) prefilter + smh_a m=1024.
The pair space is sharded by query rows: interleaved blocks of 128 rows dealt to the ranks boustrophedon (0 .. N-1, N-1 .. 0, ...); every rank holds a
full replica of the sketches.

`python bench.py --gpus N` with N > 1 and no torch.distributed environment starts its own ranks: it runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>` as a CHILD process,
before anything in this process has touched the GPU, and exits with the child's code.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline          the binding resource of the kernel that is the LONGEST of the step by measured duration (the all-pairs signature
                    join, or stage 2a's union histograms): vector-instruction issue, `frac` against the mix-weighted peak plus the
                    same rate against the guide's raw issue rate and the measured mixed rate, the model count beside PMC's SQ_INSTS_VALU;
  stage2_roofline / stage1_roofline   the other stage's roofline; roofline_choice says which kernel was chosen and why;
  hbm_algorithmic   SURVEY.md 8(d)'s nominal 8*m bytes per pair-comparison against the HBM peak (NOT a bound of this design);
  stream_kernel     the literal north_star kernel (query tile in LDS, candidates streamed from HBM) on the same inputs;
  harder_workload   the same configuration with 25 % degenerate genomes (1.4 % of all pairs pass a band);
  cpu_baseline      the oracle (OpenMP port of selection.cpp's loop) on a bounded sample, all cores; cpu_baseline_8t: 8 threads.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# ---- hardware constants (every one traceable to a file under profiles/ or to the guide) -----------------------------
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_MEASURED_GBS = 6290.0      # same table: measured float4 copy (79 % of the spec)
FABRIC_GATHER_GBS = 7400.0     # same guide, gather table: uniformly random whole rows of a 151 MB table (Infinity Cache) 7.4-7.9 TB/s
N_SIMD = 256 * 4               # 256 CUs x 4 SIMDs
SHADER_HZ = 2.4e9              # profiles/r02_valu_rate.txt: s_memtime tick rate 2407 MHz, rates quoted at 2400 MHz
# profiles/r02_valu_rate.txt (scripts/microbench/valu_rate.hip), cycles per wave64 instruction per SIMD at >= 4 waves/SIMD:
CYC_VALU_PLAIN = 2.07          # VGPR-only VOP2 (v_xor_b32, v_and_b32, v_add_u32, v_min_u16, v_fma_f32 ...) -- two waves co-issue
CYC_VALU_FULL = 4.07           # anything with DPP / SDWA / an SGPR operand / three sources (VOP3) / packed math (VOP3P)
CYC_VALU_RAW = 2.0             # MI355X_MICROARCH.md: a wave64 VALU instruction issues in 2 cycles per SIMD (4 for one wave alone)
CYC_MIX_MEASURED = 3.85        # profiles/r02_valu_rate.txt, rows "mix: v_xor_b32 (VGPR) / v_pk_min_u16": 3.76-3.95 cycles per instruction
CYC_BITOP3 = 2.2               # profiles/r03_bitplane_rate.txt: v_bitop3_b32 2.17-2.27 cycles (it pairs like a VGPR-only VOP2); v_bcnt_u32_b32 4.05
CYC_DS_2DWORD = 4.0            # MI355X_MICROARCH.md LDS table: a DS op moving 2 dwords per lane (ds_add_u32, ds_write_b32) = 4 cycles/CU
IL_BLOCK = 128                 # rows per interleave block of the multi-GPU partition.  (256 rows -- the candidates of a join block -- were measured
                               # in round 3: the slowest of 2 / 4 / 8 ranks of the weak-scaled workload 2 % faster, configs[3] x 8 the same, configs[4] x 8
                               # 15 % SLOWER (1.52 against 1.32 ms: 128-row join tiles want one tile per block): scripts/il_block_parts.py, emulate_strong.py --block=)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", help="key of cuda_selection_criteria_amd.synth.SYNTH_CONFIGS")
    ap.add_argument("--genomes", type=int, default=0, help="override the genome count (0 = config value; weak scaling multiplies it by sqrt(gpus))")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--mode", choices=["smh_a", "CB+smh_a"], default="smh_a")
    ap.add_argument("--algo", choices=["auto", "stream", "sig", "hashjoin"], default="auto",
                    help="hashjoin = sub-quadratic sort-based candidate generation: NOT the brute-force pair-comparison metric")
    ap.add_argument("--join-q", type=int, default=-1, help="signature join query side: 1 LDS tile (default), 0 DPP broadcast")
    ap.add_argument("--hard", action="store_true", help="run the harder variant of the workload (25 %% degenerate genomes) as the main line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip stream_kernel / harder_workload (profiling runs)")
    ap.add_argument("--no-grouping", action="store_true", help="stage 2 without bucketing the survivors by query row")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 auto (2 chunk lanes from 5e8 pairs per pass), 0 off, 2..8 row chunks, each a whole chain on one of two streams")
    ap.add_argument("--pcie", action="store_true", help="also time a PCIe-inclusive pass (host buffers -> upload -> run)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the collectives even with one rank (RCCL smoke test)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the multi-rank logic on a box with fewer GPUs than ranks (records staged through the host)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target CPU time of each baseline sample")
    ap.add_argument("--param", action="append", default=[], metavar="NAME=VALUE", help="selhip_ctx_set_param (development A/B runs), repeatable")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as a child process tree.  Nothing in THIS
    process has initialised the GPU (no torch import yet), and the child is a fresh interpreter -- no exec of a GPU process."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def join_issue_model(n_bands, join_q):
    """VALU wave-instructions of the signature join per (query row x 64 candidates), split by issue cost -- counted from the ISA
    (csrc/kernel_sigjoin.cuh: per packed dword one xor and one v_pk_min_u16; per row 3 v_pk_min_u16 + v_min_u32_sdwa + v_cmp)."""
    nd = n_bands // 2
    if join_q:
        return {"plain": nd, "full": nd + 5}        # LDS form: the xor is a VGPR-only VOP2
    return {"plain": 0, "full": 2 * nd + 5}         # DPP form: v_xor_b32_dpp is a full-cost instruction


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist

    import cuda_selection_criteria_amd as pkg
    from cuda_selection_criteria_amd import distributed as D
    from cuda_selection_criteria_amd.selection import PAIR_DTYPE

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the product path has no CPU fallback")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        # stdout carries exactly one JSON line.  NCCL_DEBUG=VERSION (set on this pool) makes RCCL printf a five-line banner
        # to stdout at init; that level logs nothing else, so it is dropped here -- any other level the user set is kept,
        # with its output sent to a file.
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/rccl_debug.%h.%p.log")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")       # where the collectives run

    base = pkg.SYNTH_CONFIGS[args.workload]
    n_genomes = args.genomes or base.n_genomes
    if args.scaling == "weak" and world > 1 and not args.genomes:
        n_genomes = int(round(base.n_genomes * math.sqrt(world) / base.cluster_size)) * base.cluster_size
    cfg = base.scaled(n_genomes) if n_genomes != base.n_genomes else base
    mode = pkg.MODE_SMH if args.mode == "smh_a" else pkg.MODE_CB_SMH
    algo = {"auto": pkg.ALGO_AUTO, "stream": pkg.ALGO_STREAM, "sig": pkg.ALGO_SIG, "hashjoin": pkg.ALGO_HASHJOIN}[args.algo]
    n_rows, n_bands = pkg.banding(cfg.m, cfg.tau)

    # ---- inputs: generated in HBM, sorted into rank order (identical replica on every rank) --------------
    hll_t, aux_t, cards_t, _, aux_hll_t = pkg.synth_device(cfg, device=local_rank)
    n_degenerate = pkg.harden(aux_t) if args.hard else 0
    cards = cards_t.cpu().numpy()
    sel = pkg.Selector(local_rank)
    sel.attach(hll_t, aux_t, cards_t)
    sel.set_pipeline(args.pipeline)
    join_q = 1
    if args.join_q >= 0:
        join_q = int(args.join_q != 0)
        sel.set_param("join_q", join_q)
    for kv in args.param:
        name, _, val = kv.partition("=")
        sel.set_param(name, int(val))
    if world > 1:
        # shard the pair space by interleaved blocks of query rows, dealt boustrophedon (csrc/common.cuh, RowMap), i.e. an
        # equal share of the pairs AND of the survivors (stage 2) -- a contiguous equal-pair cut would leave the last rank
        # with a third of all rows, hence of all stage-2 work
        sel.set_row_interleave(IL_BLOCK, world, rank)
    sel.set_stage2_grouping(not args.no_grouping)
    two_stage = cfg.p_aux > 0                      # BASELINE configs[4]: hll_a prefilter + smh_a
    if two_stage:
        sel.attach_aux_hll(aux_hll_t, cfg.p_aux)
        sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)
    row_lo, row_hi = 0, n_genomes                  # the interleave (not a row range) partitions the rows

    # gather: ONE all_gather per step of a fixed-capacity record buffer whose record 0 carries the count.
    # The capacity is sized from the first (untimed) step: 1.25 x the largest per-rank count, so the timed loop
    # allocates nothing and exchanges ~16 B per selected pair, not a worst-case buffer.
    # Two frame/gather buffer sets alternate, and the RCCL all_gather of step k is left in flight (async_op) while step k+1
    # computes: it is waited for before its buffers are reused and at the end of the timed region.
    state = {"cap": 0, "send": None, "recv": None, "host": None, "k": 0, "work": [None, None], "last": 0}

    def size_gather(local_count):
        for w in state["work"]:
            if w is not None:
                w.wait()
        state["work"] = [None, None]
        mx = torch.tensor([local_count], dtype=torch.int64, device=cdev)
        if dist_on:
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        cap = (int(mx.item()) * 5 // 4 + 4096) // 4096 * 4096
        state["cap"] = cap
        state["send"] = [torch.zeros((cap + 1, 2), dtype=torch.int64, device=dev) for _ in range(2)]              # 16 B records
        state["recv"] = [torch.zeros((world, cap + 1, 2), dtype=torch.int64, device=cdev) for _ in range(2)] if dist_on else None
        state["host"] = torch.zeros((cap + 1, 2), dtype=torch.int64).pin_memory() if args.backend == "gloo" else None

    def step():
        if state["send"] is not None and (not dist_on or args.backend == "nccl"):
            # steady state: the frame copy and the collective are enqueued BEHIND the running pass, before the host waits for
            # it, so their launch cost overlaps the pass (the header carries the device-side count; the payload copy moves the
            # whole frame capacity).  The pass is deterministic, so after the first (synchronous) step nothing can overflow.
            sel.run_async(cfg.tau, mode, n_rows, n_bands, rows=(row_lo, row_hi), algo=algo)
            slot = state["k"] & 1
            state["k"] += 1
            if state["work"][slot] is not None:
                state["work"][slot].wait()
                state["work"][slot] = None
            send = state["send"][slot]
            sel.copy_results_framed_async(send)
            if dist_on:
                recv = state["recv"][slot]
                state["work"][slot] = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=True)   # RCCL over xGMI
            sel.finish()
            cnt = sel.result_count()
            if cnt > state["cap"] or sel.last_attempts() != 1:
                raise RuntimeError(f"frame stale: count {cnt} > capacity {state['cap']} or the pass was repeated ({sel.last_attempts()})")
            state["last"] = slot
            return cnt
        sel.run(cfg.tau, mode, n_rows, n_bands, rows=(row_lo, row_hi), algo=algo, fetch=False)
        cnt = sel.result_count()
        if state["send"] is None or cnt > state["cap"]:
            size_gather(cnt)
        slot = state["k"] & 1
        state["k"] += 1
        if state["work"][slot] is not None:
            state["work"][slot].wait()
            state["work"][slot] = None
        send = state["send"][slot]
        sel.copy_results_framed(send)                       # header {count} + records, device-to-device, no host hop
        if dist_on:
            recv = state["recv"][slot]
            if args.backend == "nccl":
                state["work"][slot] = dist.all_gather_into_tensor(recv.view(-1), send.view(-1), async_op=True)   # RCCL over xGMI
            else:
                state["host"].copy_(send)
                dist.all_gather_into_tensor(recv.view(-1), state["host"].view(-1))
        state["last"] = slot
        return cnt

    def sync_all():
        for i, w in enumerate(state["work"]):
            if w is not None:
                w.wait()
                state["work"][i] = None
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # warm-up with every kernel timed: decides which kernel is the longest of the step -- the one whose HIP events stay on inside the
    # timed region (timing level 2: an event pair costs ~10 us of stream time, so only one kernel keeps them there; the other
    # kernels' figures are collected in extra passes after the timed region)
    sel.timing(1)
    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize(dev)
    used_sig = sel.kernel_ms("join") > 0
    dom_key = "join" if used_sig else "stage1"
    timed_hist = 1 if sel.kernel_ms("hist") > sel.kernel_ms(dom_key) else 0
    sel.set_param("timed_kernel", timed_hist)
    khi = sel.get_param("hll_khi")
    info = {"bitplanes": bool(sel.get_param("hist_bitplanes")), "khi": khi, "planes": 4 if khi <= 16 else (5 if khi <= 32 else 6),
            "label": bool(sel.get_param("label_order")), "small": bool(sel.get_param("small_pass_used"))}
    info["layout"] = (f"p=14 registers as 6 bit planes of 512 dwords per genome (12 KiB; {info['planes']} planes non-zero in this set), written once at attach"
                      if info["bitplanes"] else "p=14 registers as bytes (16 KiB per genome)")
    sel.timing(2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    st = sel.stats()
    timed_key = "hist" if timed_hist else dom_key
    dom_pass_ms = sel.kernel_ms(timed_key)                       # all launches of one pass (a pipelined pass: one per row chunk)
    dom_launches = max(1.0, sel.kernel_launches(timed_key))
    dom_span_ms = sel.kernel_ms("join_span") if (used_sig and not timed_hist) else -1.0   # first start -> last end of the pass's join launches (chunk lanes overlap them)
    sel.timing(1)                                                # outside the timed region: every kernel scope, a few passes
    for _ in range(5):
        step()
    torch.cuda.synchronize(dev)
    detail_ms = {k: sel.kernel_ms(k) for k in ("prep", "sigbuild", "join", "verify", "stage1", "aux", "group", "hist", "select", "total")}
    stage1_launches = max(1.0, sel.kernel_launches(dom_key))
    sel.timing(0)

    t_max = torch.tensor([dt], dtype=torch.float64, device=cdev)
    totals = torch.tensor([st["evaluated"], st["survivors"], st["selected"], 1], dtype=torch.int64, device=cdev)
    if dist_on:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    dt = float(t_max.item())
    pairs_per_step = int(totals[0].item())
    n_ranks_seen = int(totals[3].item())
    value = pairs_per_step * args.steps / dt

    # ---- result check outside the timed region: the gathered list holds every rank's records -------------------
    if dist_on:
        sync_all()
        rec = state["recv"][state["last"]].cpu().numpy()
        cts = rec[:, 0, 0]
        assert int(cts.sum()) == int(totals[2].item()), (cts, totals)
        allp = np.concatenate([rec[r, 1:1 + int(cts[r])].reshape(-1).view(PAIR_DTYPE) for r in range(world)])
        key = allp["i"].astype(np.int64) * n_genomes + allp["k"]
        assert len(np.unique(key)) == len(key), "duplicate pairs across shards"
        for r in range(world):                        # every record sits in a row block owned by the rank that sent it
            ii = rec[r, 1:1 + int(cts[r])].reshape(-1).view(PAIR_DTYPE)["i"]
            assert (world == 1) or (D.interleave_owner(ii, IL_BLOCK, world) == r).all()
        if mode == pkg.MODE_SMH:                      # the shards tile the whole triangle
            assert pairs_per_step == n_genomes * (n_genomes - 1) // 2, (pairs_per_step, n_genomes)

    out = None
    if rank == 0:
        pairs_rank0 = st["evaluated"]
        launches = stage1_launches
        dom_ms = dom_pass_ms / dom_launches                      # average launch duration of the timed kernel, measured inside the timed region
        alg_bytes = pairs_rank0 * 8 * cfg.m / launches           # 8*m bytes per pair-comparison, per launch
        wkey = args.workload + ("hard" if args.hard else "")
        kernels = {k: v for k, v in detail_ms.items() if v > 0}
        kernels[timed_key + "_in_timed_region"] = dom_pass_ms
        stage1_ms = detail_ms["join"] if used_sig else detail_ms["stage1"]
        achieved_hbm = alg_bytes / (stage1_ms / launches * 1e-3) / 1e9 if stage1_ms > 0 else None
        out = {
            "metric": "sketch pair-comparisons/sec (N genomes x m buckets)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{cfg.name}{' +25% degenerate genomes' if args.hard else ''}; mode {args.mode}; bands {n_bands} x {n_rows} rows; "
                                   f"pair space sharded by query rows over {world} GPU(s), selected pairs all_gathered",
                       "n_genomes": n_genomes, "m": cfg.m, "tau": cfg.tau,
                       "algo": "hashjoin (pairs are NOT compared one by one: equivalent pairs/s)" if args.algo == "hashjoin" else ("sig" if used_sig else "stream"),
                       "criterion": "hll_a+smh_a" if two_stage else "smh_a",
                       "pairs_per_step": pairs_per_step, "selected_pairs": int(totals[2].item()),
                       "stage1_survivors": int(totals[1].item()), "n_ranks_seen": n_ranks_seen,
                       "hll_layout": info["layout"], "stage2_grouping": "label order" if info["label"] else "query-row order",
                       **({"pass": "one launch (small_pass_kernel)"} if info["small"] else {})},
            "bucket_pair_comparisons_per_s_nominal": value * cfg.m,
            "kernel_ms": kernels,
        }
        # ---- rooflines: one per stage; `roofline` = that of the kernel that is the longest of the step (durations of the passes
        # with every kernel timed, outside the timed region; the one chosen for the timed region's events is stated)
        r1 = None
        if used_sig and args.algo != "hashjoin" and stage1_ms > 0:
            out["bucket_pair_comparisons_note"] = ("nominal = pairs/s x m: the signature join decides every pair from n_bands 16-bit band "
                                                   "signatures, it does not execute m bucket compares per pair (the stream kernel's figure in "
                                                   "`stream_kernel` is the executed one)")
            in_region = dom_key == "join" and timed_hist == 0
            r1 = join_roofline(pairs_rank0, n_bands, join_q, dom_pass_ms if in_region else detail_ms["join"], launches, dom_span_ms if in_region else -1.0,
                               wkey, "HIP events inside the timed region" if in_region else "HIP events of 5 extra passes after the timed region")
        elif info["small"]:
            r1 = {"bound": "latency", "kernel": "small_pass_kernel (the whole pass of a set of <= 2 048 genomes in one launch: bounds + signatures, "
                                                "a grid barrier, then join, verification, union histograms and estimator inside each block)",
                  "avg_launch_ms": stage1_ms / launches if stage1_ms > 0 else None, "launches_per_step": launches, "achieved": None, "peak": None,
                  "unit": "n/a", "frac": None, "traffic": None,
                  "note": "a chain of dependent memory round trips on a few waves per CU: no throughput resource is near its limit at this size"}
        elif not used_sig and cfg.m % 128 == 0 and (n_rows & (n_rows - 1)) == 0 and stage1_ms > 0:
            r1 = stream_issue_roofline(cfg.m, n_rows, pairs_rank0 / launches, stage1_ms / launches)
            if r1:
                r1.update({"avg_launch_ms": stage1_ms / launches, "launches_per_step": launches, "traffic": traffic_of("stage1", f"{args.workload}:stream")[0]})
        r2 = None
        if detail_ms["hist"] > 0:
            h_ms = dom_pass_ms if timed_hist else detail_ms["hist"]
            r2 = hist_roofline(st["survivors"], h_ms, max(1.0, sel.kernel_launches("hist")), info, n_genomes, wkey,
                               "HIP events inside the timed region" if timed_hist else "HIP events of 5 extra passes after the timed region")
        hist_longest = r2 is not None and detail_ms["hist"] > stage1_ms
        if hist_longest:
            out["roofline"], out["stage1_roofline"] = r2, r1
        else:
            out["roofline"] = r1 or {"bound": "n/a", "kernel": "sort-based join" if args.algo == "hashjoin" else "stage 1", "avg_launch_ms": dom_ms,
                                     "launches_per_step": launches, "achieved": None, "peak": None, "unit": "G instr/s", "frac": None, "traffic": None}
            out["stage2_roofline"] = r2
        out["roofline_choice"] = {"longest_kernel": "stage 2a (union histograms)" if hist_longest else "stage 1",
                                  "stage1_ms_per_step": stage1_ms, "stage2a_ms_per_step": detail_ms["hist"],
                                  "timed_region_events_on": "hist" if timed_hist else dom_key}
        out["hbm_algorithmic"] = {"bytes_per_pair": 8 * cfg.m, "achieved_GBs": achieved_hbm, "peak_GBs": HBM_PEAK_GBS,
                                  "ratio": (achieved_hbm / HBM_PEAK_GBS) if achieved_hbm else None,
                                  "note": "SURVEY.md 8(d) nominal: one candidate sketch streamed per pair-comparison.  NOT a bound of this design: "
                                          "both stage-1 kernels reuse every byte they load against a tile of queries, so the ratio exceeds 1; "
                                          "measured HBM traffic per launch is `traffic` of the stage-1 roofline"}
        if n_degenerate:
            out["config"]["degenerate_genomes"] = n_degenerate

    extras = rank == 0 and world == 1 and not args.no_extras
    # ---- the literal north_star kernel (ALGO_STREAM: query tile staged in LDS, candidates streamed row-major, lane-mask
    # reduction) measured beside the default algorithm, outside the timed region (rank 0, N=1, same inputs)
    if extras and args.algo == "auto" and not two_stage:
        for _ in range(2):
            sel.run(cfg.tau, mode, n_rows, n_bands, algo=pkg.ALGO_STREAM, fetch=False)
        sel.timing(True)
        t0s = time.perf_counter()
        ks = 5
        for _ in range(ks):
            sel.run(cfg.tau, mode, n_rows, n_bands, algo=pkg.ALGO_STREAM, fetch=False)
        torch.cuda.synchronize(dev)
        dts = (time.perf_counter() - t0s) / ks
        s_ms = sel.kernel_ms("stage1")
        s_st = sel.stats()
        s_bytes = s_st["evaluated"] * 8 * cfg.m
        s_traffic = None
        try:
            s_traffic = json.loads((ROOT / "profiles" / "stage1_traffic.json").read_text()).get(f"{args.workload}:stream")
        except Exception:
            pass
        out["stream_kernel"] = {"algo": "stream", "pairs_per_s": s_st["evaluated"] / dts, "ms_per_step": dts * 1e3, "selected_pairs": s_st["selected"],
                                "bucket_pair_comparisons_per_s_executed": s_st["evaluated"] * cfg.m / (s_ms * 1e-3),
                                "avg_launch_ms": s_ms, "traffic": s_traffic,
                                "hbm_algorithmic": {"achieved_GBs": s_bytes / (s_ms * 1e-3) / 1e9, "ratio": s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                                "roofline": stream_issue_roofline(cfg.m, n_rows, s_st["evaluated"], s_ms)}
        sel.timing(False)

    # ---- the harder variant of the same configuration, measured beside the main line (never instead of it)
    if extras and not args.hard and not two_stage and args.algo == "auto":
        aux_h = aux_t.clone()
        n_deg = pkg.harden(aux_h)
        with pkg.Selector(local_rank) as s3:
            s3.attach(hll_t, aux_h, cards_t)
            for _ in range(2):
                s3.run(cfg.tau, mode, n_rows, n_bands, fetch=False)
            torch.cuda.synchronize(dev)
            th = time.perf_counter()
            kh = 10
            for _ in range(kh):
                s3.run(cfg.tau, mode, n_rows, n_bands, fetch=False)
            torch.cuda.synchronize(dev)
            dth = (time.perf_counter() - th) / kh
            hs = s3.stats()
            s3.timing(1)
            for _ in range(3):
                s3.run(cfg.tau, mode, n_rows, n_bands, fetch=False)
            hk = {k: s3.kernel_ms(k) for k in ("sigbuild", "join", "verify", "group", "hist", "select")}
            h_launches = {k: max(1.0, s3.kernel_launches(k)) for k in ("join", "hist")}
            s3.timing(0)
        clock_h = "HIP events of 3 passes with every kernel timed"
        hr2 = hist_roofline(hs["survivors"], hk["hist"], h_launches["hist"], info, n_genomes, args.workload + "hard", clock_h) if hk["hist"] > 0 else None
        hr1 = join_roofline(hs["evaluated"], n_bands, join_q, hk["join"], h_launches["join"], -1.0, args.workload + "hard", clock_h) if hk["join"] > 0 else None
        out["harder_workload"] = {"workload": f"{cfg.name} with {n_deg} of {n_genomes} genomes degenerate (buckets mod 2): pairs among them pass a band by chance",
                                  "value": hs["evaluated"] / dth, "unit": "pairs/s", "ms_per_step": dth * 1e3,
                                  "stage1_survivors": hs["survivors"], "survivor_fraction": hs["survivors"] / max(1, hs["evaluated"]),
                                  "selected_pairs": hs["selected"], "kernel_ms": {k: v for k, v in hk.items() if v > 0},
                                  "roofline": hr2 if hk["hist"] >= hk["join"] else hr1,
                                  "other_stage_roofline": hr1 if hk["hist"] >= hk["join"] else hr2}
        del aux_h

    # ---- CPU baseline: the oracle (OpenMP port of selection.cpp:270-291 / time_smh.cpp:229-257) on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_py
        orc = oracle_py.Oracle()
        cores = os.cpu_count() or 1
        use_cb = mode == pkg.MODE_CB_SMH

        def cpu_run(ns, threads, reps=1):
            h = hll_t[:ns].cpu().numpy()
            a = aux_t[:ns].cpu().numpy().view(np.uint64)
            ah = aux_hll_t[:ns].cpu().numpy() if two_stage else None
            c = cards[:ns]
            t = time.perf_counter()
            for _ in range(reps):
                pairs, s = orc.select(h, a, c, cfg.tau, n_rows, n_bands, use_cb=use_cb, threads=threads,
                                      criterion=3 if two_stage else 0, aux_hll=ah, p_aux=cfg.p_aux or 8)
            return time.perf_counter() - t, s["evaluated"] * reps, len(pairs)

        def cpu_row(threads):
            ns = min(n_genomes, 2000)
            t_probe, ev, _ = cpu_run(ns, threads)
            rate = ev / max(t_probe, 1e-9)
            ns = int(min(n_genomes, max(ns, math.sqrt(2 * rate * args.cpu_seconds))))
            t1, ev1, _ = cpu_run(ns, threads)                    # one pass at the chosen size, then repeat to fill the budget
            reps = int(max(1, min(200, args.cpu_seconds / max(t1, 1e-3))))
            t_probe, ev, nsel = cpu_run(ns, threads, reps)
            return {"value": ev / t_probe, "unit": "pairs/s", "cores": threads, "kind": "port",
                    "sample": f"first {ns} genomes (rank order) of the same set, {reps} pass(es): {ev} pairs in {t_probe:.2f} s; "
                              f"oracle/liboracle.so orc_select = OpenMP port of src/selection.cpp:270-291, schedule(dynamic) over rows"}

        out["cpu_baseline"] = cpu_row(cores)
        out["cpu_baseline_8t"] = cpu_row(min(8, cores))          # the reference's default thread count (selection.cpp:79)
        # ---- the REFERENCE's own CPU program (oracle/_ref/selection, built from the unmodified sources in the
        # authoring container and shipped prebuilt) on sketch FILES written from the same set: two sizes, so that
        # file loading cancels in the difference (the program does not time its loop separately).
        ref_bin = ROOT / "oracle" / "_ref" / "selection"
        if ref_bin.exists() and not two_stage:
            import tempfile
            try:
                host = pkg.host_lib()
                n2 = min(n_genomes, 6000)
                n1 = n2 // 2
                with tempfile.TemporaryDirectory() as td:
                    h = hll_t[:n2].cpu().numpy()
                    a = aux_t[:n2].cpu().numpy().view(np.uint64)
                    for g in range(n2):
                        base_ = f"{td}/g{g:06d}".encode()
                        host.selhost_write_hll(base_ + b".hll", h[g].ctypes.data, 14)
                        host.selhost_write_smh(base_ + f".smh{cfg.m}".encode(), a[g].ctypes.data, cfg.m)
                    times = []
                    for nn in (n1, n2):
                        Path(td, f"list{nn}.txt").write_text("".join(f"g{g:06d}\n" for g in range(nn)))
                        t = time.perf_counter()
                        subprocess.run([str(ref_bin), "-l", f"list{nn}.txt", "-t", str(min(cores, 64)), "-a", str(cfg.m * 8),
                                        "-h", str(cfg.tau), "-c", "smh_a"], cwd=td, check=True, capture_output=True)
                        times.append(time.perf_counter() - t)
                dp = n2 * (n2 - 1) // 2 - n1 * (n1 - 1) // 2
                dtm = times[1] - times[0]
                if dtm <= 0:
                    out["cpu_reference"] = {"error": f"non-positive time difference: {times}"}
                else:
                    out["cpu_reference"] = {"value": dp / dtm, "unit": "pairs/s", "cores": min(cores, 64), "kind": "reference",
                                            "sample": f"oracle/_ref/selection (the reference's src/selection.cpp, g++ -O3 -march=x86-64-v3 -fopenmp) "
                                                      f"-c smh_a on the first {n1} and {n2} genomes written as .hll/.smh{cfg.m} files: "
                                                      f"{times[0]:.2f} s and {times[1]:.2f} s wall, difference = {dp} pairs (CB+smh_a mode, the program's only mode)"}
            except Exception as e:                     # the reference binary is optional equipment
                out["cpu_reference"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and (args.pcie or not args.no_extras):
        # PCIe-inclusive variant (never `value`): the boundary handed HOST buffers (selhip_ctx_upload: copies, bit planes of the HLL
        # registers, cardinalities given), then one pass.  First call = with the context's allocations; second = the same context again.
        h = hll_t.cpu().numpy(); a = aux_t.cpu().numpy().view(np.uint64)
        nbytes = int(h.nbytes + a.nbytes + cards.nbytes)
        with pkg.Selector(local_rank) as s2:
            times = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t = time.perf_counter()
                s2.upload(h, a, cards)
                t_up = time.perf_counter() - t
                s2.run(cfg.tau, mode, n_rows, n_bands, algo=algo, fetch=False)
                times.append((time.perf_counter() - t, t_up))
            ev2 = s2.stats()["evaluated"]
        best = min(times[1:])
        out["pcie_inclusive"] = {"pairs_per_s": ev2 / best[0], "seconds": best[0], "upload_seconds": best[1], "first_call_seconds": times[0][0],
                                 "bytes_uploaded": nbytes, "upload_GBs": nbytes / best[1] / 1e9,
                                 "note": "pageable host buffers -> selhip_ctx_upload (copies + bit planes) -> one pass; best of two calls on a warm context, "
                                         "first_call_seconds includes the context's allocations.  Never the headline.  The link moves 54-56 GB/s from pageable "
                                         "and pinned memory alike (scripts/pcie_probe.py), so these bytes cannot arrive in less than "
                                         f"{nbytes / 55e9 * 1e3:.1f} ms"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    sel.close()
    if dist_on:
        dist.destroy_process_group()


def _json(path):
    try:
        return json.loads(path.read_text())
    except Exception:
        return {}


def traffic_of(stage, key):
    """bytes per step from beyond L2 recorded by an EARLIER rocprofv3 --pmc run of the same command (scripts/gpu_profile.sh +
    scripts/summarize_profiles.py write profiles/stage1_traffic.json / stage2_traffic.json); PMC needs the profiler, so an
    ordinary run cannot measure it"""
    tj = _json(ROOT / "profiles" / f"{stage}_traffic.json")
    v = tj.get(key)
    return (v, (tj.get("source", "profiles/") + " -- an earlier rocprofv3 --pmc run of this command, NOT measured by this run") if v else None)


def pmc_valu(wkey, kernel):
    """SQ_INSTS_VALU per launch of `kernel` in the committed PMC summary of this workload (profiles/pmc_counts.json)"""
    return _json(ROOT / "profiles" / "pmc_counts.json").get(wkey, {}).get(kernel, {}).get("SQ_INSTS_VALU")


def issue_fracs(wave_instr_per_s, cyc_model):
    """the same achieved rate against three denominators: the model's mix-weighted cost, the guide's raw wave64 issue rate (2 cycles
    per instruction per SIMD) and, for a plain/full mix, the pairing rate measured for such a mix"""
    return {"frac": wave_instr_per_s / (N_SIMD * SHADER_HZ / cyc_model),
            "frac_vs_raw_issue_rate": wave_instr_per_s / (N_SIMD * SHADER_HZ / CYC_VALU_RAW),
            "frac_vs_measured_mix_rate": wave_instr_per_s / (N_SIMD * SHADER_HZ / CYC_MIX_MEASURED)}


def join_roofline(pairs, n_bands, join_q, pass_ms, launches, span_ms, wkey, clock):
    """VALU issue of the all-pairs signature join: model instruction count (ISA) over the measured duration"""
    row_waves = pairs / 64.0                                   # one query row against one group of 64 candidates
    mdl = join_issue_model(n_bands, join_q)
    n_instr = mdl["plain"] + mdl["full"]
    cyc_mix = (mdl["plain"] * CYC_VALU_PLAIN + mdl["full"] * CYC_VALU_FULL) / n_instr
    peak = N_SIMD * SHADER_HZ / cyc_mix
    achieved = row_waves * n_instr / (pass_ms * 1e-3)
    traffic, traffic_src = traffic_of("stage1", f"{wkey}:sig")
    pmc = pmc_valu(wkey, "join")
    r = {"bound": "valu_issue", "kernel": (f"sigl_join_kernel<{n_bands // 2}> (query tile in LDS)" if join_q else f"sig16_join_kernel<{n_bands // 2}> (DPP broadcast)") + " (stage 1, all-pairs)",
         "achieved": achieved / 1e9, "peak": peak / 1e9, "unit": "G wave-instr/s", **issue_fracs(achieved, cyc_mix),
         "traffic": traffic / launches if traffic else None, "traffic_source": traffic_src,
         "avg_launch_ms": pass_ms / launches, "launches_per_step": launches, "clock": clock,
         "instr_per_row_wave": mdl, "cycles_per_instr_at_peak": cyc_mix,
         "model_vs_pmc": {"model_wave_instr_per_step": row_waves * n_instr, "pmc_SQ_INSTS_VALU_per_step": pmc,
                          "pmc_over_model": (pmc / (row_waves * n_instr)) if pmc else None, "source": "profiles/pmc_counts.json"},
         "denominators": f"frac: {N_SIMD} SIMDs x {SHADER_HZ / 1e9} GHz / mix-weighted cost ({mdl['plain']} VGPR-only VOP2 at {CYC_VALU_PLAIN} cycles + {mdl['full']} "
                         f"VOP3/VOP3P/DPP/SDWA instructions at {CYC_VALU_FULL}, profiles/r02_valu_rate.txt) -- not reachable with a packed min in the loop, because a "
                         f"plain instruction only pairs with another plain one; frac_vs_raw_issue_rate: the guide's {CYC_VALU_RAW} cycles per wave64 instruction; "
                         f"frac_vs_measured_mix_rate: {CYC_MIX_MEASURED} cycles, the measured rate of the alternating xor / packed-min mix (3.76-3.95)"}
    if launches > 1.5 and span_ms > 0:
        r["concurrent_launches"] = {"span_ms": span_ms, "frac_of_chip_while_running": row_waves * n_instr / (span_ms * 1e-3) / peak,
                                    "note": "this pass is cut into row chunks whose chains run on two streams (selhip_ctx_set_pipeline, automatic from 5e8 pairs): the join "
                                            "launches run side by side, each on about half of the chip, so `frac` (work of a launch / its own duration / whole-chip peak) is "
                                            "about half of what the chip delivers while they run; span_ms = first start to last end of the pass's join launches.  --pipeline 0 "
                                            "gives the single-launch figure"}
    return r


def hist_model_bitplanes(khi):
    """VALU wave-instructions per pair of hll_union_hist_bs_kernel on its always-taken walk (csrc/kernel_hllbs.cuh): per column 2 NB
    booleans for the bit-serial maximum + 4 per group of four values, one accumulating population count per decoded value; then packing,
    the transposing reduction and the store.  Pairs with a register above 23 (15 with four planes) take further walks, not counted."""
    nb = 4 if khi <= 16 else (5 if khi <= 32 else 6)
    groups = 4 if nb == 4 else 6
    return {"planes": nb, "values_decoded": 4 * groups, "boolean": 8 * (2 * nb + 4 * groups + (2 if nb == 6 else 0)),
            "bcnt": 8 * 4 * groups, "tail_full_cost": 5 * groups + 35 + 8}


def hist_roofline(survivors, hist_ms, launches, info, n_genomes, wkey, clock):
    """stage 2a.  Bit planes: VALU issue (model count / time) and the fetches from beyond L2; byte rows: LDS issue and the same fetches"""
    fetch, fetch_src = traffic_of("stage2", wkey)
    row_bytes = (info["planes"] * 2048) if info["bitplanes"] else 16384
    table = n_genomes * row_bytes
    fabric_peak = FABRIC_GATHER_GBS if table <= (256 << 20) else HBM_MEASURED_GBS
    fb = {"bytes_per_step": fetch, "achieved_GBs": (fetch / (hist_ms * 1e-3) / 1e9) if fetch else None, "peak_GBs": fabric_peak,
          "frac": (fetch / (hist_ms * 1e-3) / 1e9 / fabric_peak) if fetch else None, "source": fetch_src,
          "minimum_bytes": table, "bytes_per_pair_no_reuse": row_bytes,
          "note": "peak = what MI355X_MICROARCH.md measures for gathered whole rows: 7 400 GB/s from the Infinity Cache (tables up to ~150 MB), "
                  "6 000-6 300 GB/s from HBM; minimum = every row of the table once"}
    if info["bitplanes"]:
        m = hist_model_bitplanes(info["khi"])
        n_instr = m["boolean"] + m["bcnt"] + m["tail_full_cost"]
        cyc = (m["boolean"] * CYC_BITOP3 + (m["bcnt"] + m["tail_full_cost"]) * CYC_VALU_FULL) / n_instr
        achieved = survivors * n_instr / (hist_ms * 1e-3)
        pmc = pmc_valu(wkey, "hist")
        return {"bound": "valu_issue", "kernel": f"hll_union_hist_bs_kernel<{m['planes']}> (stage 2a: union histograms on bit planes)",
                "achieved": achieved / 1e9, "peak": N_SIMD * SHADER_HZ / cyc / 1e9, "unit": "G wave-instr/s", **issue_fracs(achieved, cyc),
                "traffic": fetch / launches if fetch else None, "traffic_source": fetch_src,
                "avg_launch_ms": hist_ms / launches, "launches_per_step": launches, "clock": clock, "pairs_per_step": survivors,
                "instr_per_pair": m, "cycles_per_instr_at_peak": cyc,
                "model_vs_pmc": {"model_wave_instr_per_step": survivors * n_instr, "pmc_SQ_INSTS_VALU_per_step": pmc,
                                 "pmc_over_model": (pmc / (survivors * n_instr)) if pmc else None, "source": "profiles/pmc_counts.json"},
                "beyond_l2_fetch": fb,
                "denominators": f"frac: v_bitop3_b32 / v_and_b32 at {CYC_BITOP3} cycles, v_bcnt_u32_b32 and the DPP / permlane / VOP3 tail at {CYC_VALU_FULL} "
                                f"(profiles/r03_bitplane_rate.txt); frac_vs_raw_issue_rate: {CYC_VALU_RAW} cycles for every instruction; frac_vs_measured_mix_rate: "
                                f"{CYC_MIX_MEASURED} cycles (the join's mix -- kept for comparison with the stage-1 line)"}
    ds_rate = survivors * 256 / (hist_ms * 1e-3)
    ds_peak = 256 * SHADER_HZ / CYC_DS_2DWORD
    return {"bound": "lds_issue", "kernel": "hll_union_hist_runs_kernel (stage 2a: byte rows, lane-private LDS histogram)", "achieved": ds_rate / 1e9, "peak": ds_peak / 1e9,
            "unit": "G ds_add_u32 wave-instr/s", "frac": ds_rate / ds_peak, "traffic": fetch / launches if fetch else None, "traffic_source": fetch_src,
            "avg_launch_ms": hist_ms / launches, "launches_per_step": launches, "clock": clock, "pairs_per_step": survivors, "beyond_l2_fetch": fb,
            "note": f"one conflict-free ds_add_u32 per 64 register pairs, 256 per pair; peak = 256 CUs x one DS op of 2 dwords per {CYC_DS_2DWORD} cycles "
                    "(MI355X_MICROARCH.md LDS table; profiles/r02_lds_rate.txt measures 4.4-5.1)"}


def stream_issue_roofline(m, n_rows, pairs, launch_ms):
    """the stream kernel's binding resource: instruction issue.  Model counts per (query, candidate) pair from the ISA
    (csrc/kernel_stream.cuh); filled in with the kernel's current shape."""
    from cuda_selection_criteria_amd import stream_model
    mdl = stream_model(m, n_rows)
    if launch_ms <= 0:
        return None
    salu = pairs * mdl["salu_per_pair"] / (launch_ms * 1e-3)
    valu = pairs * mdl["valu_wave_instr_per_pair"] / (launch_ms * 1e-3)
    salu_peak = 256 * SHADER_HZ                               # one scalar instruction per cycle per CU
    valu_peak = N_SIMD * SHADER_HZ / CYC_VALU_FULL            # v_cmp / DPP / ballot forms: full-cost instructions
    bound = "salu_issue" if salu / salu_peak >= valu / valu_peak else "valu_issue"
    return {"bound": bound, "kernel": "smh_stream_kernel", "unit": "G instr/s",
            "achieved": (salu if bound == "salu_issue" else valu) / 1e9, "peak": (salu_peak if bound == "salu_issue" else valu_peak) / 1e9,
            "frac": max(salu / salu_peak, valu / valu_peak), "salu_issue_frac": salu / salu_peak, "valu_issue_frac": valu / valu_peak,
            "model": mdl}


if __name__ == "__main__":
    main()
