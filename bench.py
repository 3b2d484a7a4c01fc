#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric: sketch pair-comparisons/sec (N genomes x m buckets).

A "step" = one pass of the hot path (mode "smh_a" of experiments/src/time_smh.cpp:229-257: every pair
i<k goes through the smh_a band predicate, survivors through the HLL-14 union estimate and the Jaccard
test) over one synthetic sketch set that is already resident in HBM, ending with the selected-pair
list of all ranks gathered on every rank (RCCL all_gather over xGMI when --gpus > 1).

Workload at 1 GPU: BASELINE.json configs[2] = 10 000 synthetic genomes, smh_a m=512, tau=0.8 -- the
configuration the north_star target (>= 1e10 m=512 bucket-pair-comparisons/s, >= 40 % of the HBM
roofline) is quoted on.  At N GPUs the genome count is scaled by sqrt(N) (per-GPU pair count fixed:
weak scaling) and the pair space is sharded by query rows, equal pairs per rank; every rank holds a
full replica of the sketches; rows are dealt to ranks in interleaved blocks of 128.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects `roofline`
(dominant kernel = stage 1, HIP-event timed inside the timed region) and `cpu_baseline` (the oracle,
an OpenMP port of selection.cpp's loop, on a bounded sample; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="cfg3", help="key of cuda_selection_criteria_amd.synth.SYNTH_CONFIGS")
    ap.add_argument("--genomes", type=int, default=0, help="override the genome count (0 = config value, scaled by sqrt(gpus))")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--mode", choices=["smh_a", "CB+smh_a"], default="smh_a")
    ap.add_argument("--algo", choices=["auto", "stream", "sig", "hashjoin"], default="auto",
                    help="hashjoin = sub-quadratic sort-based candidate generation: NOT the brute-force pair-comparison metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grouping", action="store_true", help="stage 2 without bucketing the survivors by query row")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 auto, 0 off, 2..8 row chunks (stage 1 of chunk c+1 overlaps stage 2 of chunk c)")
    ap.add_argument("--pcie", action="store_true", help="also time a PCIe-inclusive pass (host buffers -> upload -> run)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the collectives even with one rank (RCCL smoke test)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo = rehearsal of the multi-rank logic on a box with fewer GPUs than ranks (records staged through the host)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    import cuda_selection_criteria_amd as pkg
    from cuda_selection_criteria_amd import _lib
    from cuda_selection_criteria_amd.selection import PAIR_DTYPE

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
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
    cards = cards_t.cpu().numpy()
    sel = pkg.Selector(local_rank)
    sel.attach(hll_t, aux_t, cards_t)
    sel.set_pipeline(args.pipeline)
    IL_BLOCK = 128
    if world > 1:
        # shard the pair space by interleaved blocks of query rows: rank r owns the blocks b with b % world == r, i.e. an
        # equal share of the pairs AND of the survivors (stage 2) -- a contiguous equal-pair cut would leave the last rank
        # with a third of all rows, hence of all stage-2 work
        sel.set_row_interleave(IL_BLOCK, world, rank)
    sel.set_stage2_grouping(not args.no_grouping)
    two_stage = cfg.p_aux > 0                      # BASELINE configs[4]: hll_a prefilter + smh_a
    if two_stage:
        sel.attach_aux_hll(aux_hll_t, cfg.p_aux)
        sel.set_criterion(pkg.CRIT_HLL_A_SMH_A)

    # ---- shard the pair space by query rows: equal pair counts per rank ------------------------------------
    bounds = np.zeros(world + 1, dtype=np.int64)
    hi = None
    if mode == pkg.MODE_CB_SMH:
        e = cards.astype(np.uint64)          # truncation like (size_t)card
        tau64 = float(np.float32(cfg.tau))
        hi = np.empty(n_genomes, dtype=np.int32)
        for i in range(n_genomes):           # monotone: binary search per row (host plan, outside the timed region)
            lo_, hi_ = i, n_genomes - 1
            while lo_ < hi_:
                mid = (lo_ + hi_ + 1) // 2
                ok = e[mid] == 0 or (float(e[i]) / float(e[mid]) >= tau64)
                if ok:
                    lo_ = mid
                else:
                    hi_ = mid - 1
            hi[i] = lo_
    z0 = int(np.argmax(cards >= 1.0)) if (cards >= 1.0).any() else n_genomes
    rc = pkg.host_lib().selhost_shard_rows(n_genomes, hi.ctypes.data if hi is not None else None, z0, world, bounds.ctypes.data)
    assert rc == 0
    row_lo, row_hi = (0, n_genomes) if world > 1 else (int(bounds[rank]), int(bounds[rank + 1]))

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

    for _ in range(args.warmup):
        step()
    # HIP events over the timed region on the dominant stage-1 kernel only (timing level 2): an event pair costs ~10 us of
    # stream time, so the other kernels' figures are collected in extra passes after the timed region
    sel.timing(2)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_sel_local = step()
    sync_all()
    dt = time.perf_counter() - t0
    st = sel.stats()
    used_sig = sel.kernel_ms("join") > 0
    dom_key = "join" if used_sig else "stage1"
    dom_pass_ms = sel.kernel_ms(dom_key)                         # all launches of one pass (a pipelined pass: one per row chunk)
    dom_launches = max(1.0, sel.kernel_launches(dom_key))
    stage1_ms = dom_pass_ms
    sel.timing(1)                                                # outside the timed region: every kernel scope, a few passes
    for _ in range(5):
        step()
    torch.cuda.synchronize(dev)
    detail_ms = {k: sel.kernel_ms(k) for k in ("prep", "sigbuild", "join", "verify", "stage1", "aux", "group", "hist", "select", "total")}
    sel.timing(0)

    t_max = torch.tensor([dt], dtype=torch.float64, device=cdev)
    totals = torch.tensor([st["evaluated"], st["survivors"], st["selected"]], dtype=torch.int64, device=cdev)
    s1 = torch.tensor([stage1_ms], dtype=torch.float64, device=cdev)
    if dist_on:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
        dist.all_reduce(s1, op=dist.ReduceOp.MAX)
    dt = float(t_max.item())
    pairs_per_step = int(totals[0].item())
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
            assert (world == 1) or (((ii // IL_BLOCK) % world) == r).all()

    out = None
    if rank == 0:
        pairs_rank0 = st["evaluated"]
        alg_bytes = pairs_rank0 * 8 * cfg.m                     # 8*m bytes per pair-comparison
        launches = dom_launches
        dom_ms = dom_pass_ms / launches                          # average launch duration, measured inside the timed region
        alg_bytes = alg_bytes / launches                         # algorithmic bytes of one launch
        dom_name = (f"sig16_join_kernel<{n_bands // 2}> ({n_bands} bands, 16-bit signatures two per dword)" if args.algo != "hashjoin" else "sort-based join") if used_sig else "smh_stream_kernel"
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else None
        traffic = None
        tfile = ROOT / "profiles" / "stage1_traffic.json"
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get(f"{args.workload}:{'sig' if used_sig else 'stream'}")
                traffic = traffic / launches if traffic else traffic        # stored per step
            except Exception:
                traffic = None
        hist_ms = detail_ms["hist"]
        surv0 = st["survivors"]
        kernels = {k: v for k, v in detail_ms.items() if v > 0}
        kernels[dom_key + "_in_timed_region"] = dom_pass_ms
        out = {
            "metric": "sketch pair-comparisons/sec (N genomes x m buckets)",
            "value": value, "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{cfg.name}; mode {args.mode}; bands {n_bands} x {n_rows} rows; "
                                   f"pair space sharded by query rows over {world} GPU(s), selected pairs all_gathered",
                       "n_genomes": n_genomes, "m": cfg.m, "tau": cfg.tau,
                       "algo": "hashjoin (pairs are NOT compared one by one: equivalent pairs/s)" if args.algo == "hashjoin" else ("sig" if used_sig else "stream"),
                       "criterion": "hll_a+smh_a" if two_stage else "smh_a",
                       "pairs_per_step": pairs_per_step, "selected_pairs": int(totals[2].item()),
                       "stage1_survivors": int(totals[1].item())},
            "bucket_pair_comparisons_per_s": value * cfg.m,
            "roofline": {"bound": "hbm", "kernel": dom_name + " (stage 1, all-pairs)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms, "launches_per_step": launches,
                         "note": "algorithmic bytes = 8*m per pair-comparison (SURVEY.md 8d: one candidate sketch streamed per "
                                 "pair, query on chip). Both stage-1 kernels reuse every byte they load across a tile of "
                                 "queries, so the algorithmic rate exceeds the HBM peak; `traffic` = measured HBM bytes/launch "
                                 "(rocprofv3 FETCH_SIZE/WRITE_SIZE, profiles/). The signature join is bound by VALU issue "
                                 "(1.5 VALU per band and 64 pairs), see `valu`."},
            "stage2_roofline": {"bound": "lds", "kernel": "hll_union_hist_runs_kernel",
                                "achieved": (surv0 * 256 / (hist_ms * 1e-3)) if hist_ms > 0 else None, "peak": 256 * 2.4e9 / 4.5,
                                "unit": "ds_add_u32 wave-instructions/s",
                                "frac": (surv0 * 256 / (hist_ms * 1e-3)) / (256 * 2.4e9 / 4.5) if hist_ms > 0 else None,
                                "ms_per_step": hist_ms, "row_bytes_per_s": (surv0 * 32768 / (hist_ms * 1e-3)) if hist_ms > 0 else None,
                                "note": "one conflict-free ds_add_u32 per 64 register pairs, 256 per surviving pair; peak = 256 CUs x one "
                                        "such instruction per ~4.5 cycles (scripts/microbench/lds_atomic_rate.hip on this part); "
                                        "row_bytes_per_s = the 2 x 16 KiB of HLL registers per pair, served mostly by L2/MALL"},
            "kernel_ms": kernels,
        }
        if used_sig and dom_ms > 0:
            groups = (n_genomes + 63) // 64
            wave_queries = pairs_rank0 / 64.0                    # one query against one 64-candidate group
            valu = wave_queries * n_bands * 1.0                 # v_xor_b32_dpp + v_pk_min_u16 per TWO bands
            dom_ms = dom_pass_ms
            out["roofline"]["valu"] = {"achieved_wave_instr_per_s": valu / (dom_ms * 1e-3), "peak_wave_instr_per_s": 256 * 4 * 2.4e9 / 4,
                                       "frac": valu / (dom_ms * 1e-3) / (256 * 4 * 2.4e9 / 4),
                                       "note": "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction"}

    # ---- the literal north_star kernel (ALGO_STREAM: query tile staged in LDS, candidates streamed row-major, lane-mask
    # reduction) measured beside the default algorithm, outside the timed region (rank 0, N=1, same inputs)
    if rank == 0 and world == 1 and args.algo == "auto" and not two_stage:
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
                                "roofline": {"bound": "hbm", "kernel": "smh_stream_kernel", "achieved": s_bytes / (s_ms * 1e-3) / 1e9,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s_bytes / (s_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                             "traffic": s_traffic, "avg_launch_ms": s_ms, "algorithmic_bytes_per_launch": s_bytes}}
        sel.timing(False)

    # ---- CPU baseline: the oracle (OpenMP port of selection.cpp:270-291 / time_smh.cpp:229-257) on a bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, str(ROOT / "tests"))
        import oracle_py
        orc = oracle_py.Oracle()
        cores = os.cpu_count() or 1
        use_cb = mode == pkg.MODE_CB_SMH

        def cpu_run(ns, reps=1):
            h = hll_t[:ns].cpu().numpy()
            a = aux_t[:ns].cpu().numpy().view(np.uint64)
            ah = aux_hll_t[:ns].cpu().numpy() if two_stage else None
            c = cards[:ns]
            t = time.perf_counter()
            for _ in range(reps):
                pairs, s = orc.select(h, a, c, cfg.tau, n_rows, n_bands, use_cb=use_cb, threads=cores,
                                      criterion=3 if two_stage else 0, aux_hll=ah, p_aux=cfg.p_aux or 8)
            return time.perf_counter() - t, s["evaluated"] * reps, len(pairs)

        ns = min(n_genomes, 2000)
        t_probe, ev, _ = cpu_run(ns)
        rate = ev / max(t_probe, 1e-9)
        ns = int(min(n_genomes, max(ns, math.sqrt(2 * rate * args.cpu_seconds))))
        t1, ev1, _ = cpu_run(ns)                                 # one pass at the chosen size, then repeat to fill the budget
        reps = int(max(1, min(200, args.cpu_seconds / max(t1, 1e-3))))
        t_probe, ev, nsel = cpu_run(ns, reps)
        out["cpu_baseline"] = {"value": ev / t_probe, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"first {ns} genomes (rank order) of the same set, {reps} pass(es): {ev} pairs in {t_probe:.2f} s; "
                                         f"oracle/liboracle.so orc_select = OpenMP port of src/selection.cpp:270-291, schedule(dynamic) over rows"}
        # ---- the REFERENCE's own CPU program (oracle/_ref/selection, built from the unmodified sources in the
        # authoring container and shipped prebuilt) on sketch FILES written from the same set: two sizes, so that
        # file loading cancels in the difference (the program does not time its loop separately).
        ref_bin = ROOT / "oracle" / "_ref" / "selection"
        if ref_bin.exists() and not use_cb is None and not two_stage:
            import subprocess
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
    if rank == 0 and world == 1 and args.pcie:
        # PCIe-inclusive variant (never `value`): host buffers handed to selhip_ctx_upload, then one pass
        h = hll_t.cpu().numpy(); a = aux_t.cpu().numpy().view(np.uint64)
        with pkg.Selector(local_rank) as s2:
            t = time.perf_counter()
            s2.upload(h, a, cards)
            s2.run(cfg.tau, mode, n_rows, n_bands, algo=algo, fetch=False)
            dt2 = time.perf_counter() - t
            ev2 = s2.stats()["evaluated"]
        out["pcie_inclusive"] = {"pairs_per_s": ev2 / dt2, "seconds": dt2, "bytes_uploaded": int(h.nbytes + a.nbytes + cards.nbytes),
                                 "note": "pageable host buffers -> selhip_ctx_upload -> one pass; not the headline"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    sel.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
