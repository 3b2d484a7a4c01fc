"""N>1 path on CPU: world_size-2 gloo.  The GPU compute is replaced by the CPU oracle as a stand-in producer (this is a
test of the partition + gather logic in cuda_selection_criteria_amd/distributed.py, which is backend-agnostic).  Rows are
dealt out the way the product deals them -- interleaved blocks of 128 rows, dealt to the ranks boustrophedon
(selhip_ctx_set_row_interleave) -- and the union of the shards must equal the single-rank answer, with no duplicates and
every record on the rank that owns its row.  The GPU side of the same partition (Selector.set_row_interleave, parts 0..P-1 on
one device tile the full result and its statistics) is tests/test_gpu_parity.py::test_interleaved_row_blocks_tile_the_pair_space."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import torch.distributed as dist
    import cuda_selection_criteria_amd as pkg
    from cuda_selection_criteria_amd import distributed as D
    from cuda_selection_criteria_amd.selection import PAIR_DTYPE
    from cuda_selection_criteria_amd.synth import SynthConfig
    import oracle_py

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        orc = oracle_py.Oracle()
        cfg = SynthConfig("gloo", 300, 128, 0.9, 31, mode=1, n_sh_lo=2000, n_sh_hi=20000)
        hll, aux, _ = pkg.synth_host(cfg, threads=2)
        cards = orc.cards(hll)
        perm = pkg.sort_by_card(cards)
        hll, aux, cards = hll[perm], aux[perm], cards[perm]
        r, b = pkg.banding(cfg.m, cfg.tau)
        hi = D.cb_bounds(cards, cfg.tau)
        BLOCK = 32                                       # small blocks so that 300 rows give both ranks several (the product uses 128)
        # stand-in for Selector.set_row_interleave(BLOCK, world, rank) + run(): the oracle on the full set, filtered to the rows this rank owns
        full, st = orc.select(hll, aux, cards, cfg.tau, r, b, use_cb=True, threads=2)
        mine = full[D.interleave_owner(full["i"], BLOCK, world) == rank]
        # the partition is a partition of the PAIR SPACE too: per-rank pair counts add up to the oracle's evaluated count
        pc = D.interleave_pair_counts(len(cards), BLOCK, world, hi, D.first_nonzero(cards))
        assert int(pc.sum()) == st["evaluated"] and (pc > 0).all(), (pc, st)
        # the contiguous equal-pair cut (selhost_shard_rows) covers the same space
        bounds = D.shard_rows(len(cards), world, hi, D.first_nonzero(cards))
        assert int(D.pair_counts(len(cards), bounds, hi, D.first_nonzero(cards)).sum()) == st["evaluated"]
        local = np.zeros(len(mine), dtype=PAIR_DTYPE)
        local["i"], local["k"], local["jaccard"] = mine["i"], mine["k"], mine["jacc"]
        gathered = D.gather_pairs(local, dist)
        key = gathered["i"].astype(np.int64) * len(cards) + gathered["k"]
        assert len(np.unique(key)) == len(key)
        ok = (len(gathered) == len(full) and np.array_equal(gathered["i"], full["i"]) and np.array_equal(gathered["k"], full["k"])
              and np.array_equal(gathered["jaccard"].view(np.uint64), full["jacc"].view(np.uint64)))
        Path(out_dir, f"rank{rank}.txt").write_text(f"{int(ok)} {len(local)} {len(full)}")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gather_equals_whole(tmp_path):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [Path(tmp_path, f"rank{r}.txt").read_text().split() for r in range(2)]
    assert all(r[0] == "1" for r in res), res
    assert int(res[0][1]) + int(res[1][1]) == int(res[0][2]) and int(res[0][1]) > 0 and int(res[1][1]) > 0
