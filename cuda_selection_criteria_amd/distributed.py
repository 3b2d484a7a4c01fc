"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in CPU tests).  The pair space shards naturally -- every pair is independent given
read-only sketches -- so each rank holds a full replica of the sketches and evaluates
the query rows it owns.  Ownership is an INTERLEAVE: rows are cut into blocks of `block` rows (128), dealt to the ranks
boustrophedon -- 0 .. world-1, then world-1 .. 0, and so on (selhip_ctx_set_row_interleave; `interleave_owner` below) -- every rank then gets the
same share of pairs AND of survivors whatever the shape of the (triangular or CB-banded) pair space; the
contiguous equal-pair cut (`shard_rows`, libselhost selhost_shard_rows) stays for callers that want row ranges.
The only exchange step is the gather of the selected-pair lists.  No collective touches the sketch data path."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import host_lib
from .selection import PAIR_DTYPE


def cb_bounds(cards: np.ndarray, tau: float) -> np.ndarray:
    """hi[i] = last rank k that still passes CB against row i (criteria_sketch.hpp:45-49 on the
    size_t-truncated cardinalities of selection.cpp:275,280; monotone because cards are ascending).
    One binary search per row, all rows at once (log2 N vectorised steps, the same IEEE double divide as the predicate)."""
    e = np.asarray(cards, dtype=np.float64).astype(np.uint64).astype(np.float64)   # (double)(size_t)card
    n = e.shape[0]
    tau64 = float(np.float32(tau))
    lo = np.arange(n, dtype=np.int64)                     # invariant: the predicate holds at lo (k = i counts as true)
    hi = np.full(n, n - 1, dtype=np.int64)
    ei = e.copy()
    with np.errstate(divide="ignore", invalid="ignore"):
        while True:
            act = lo < hi
            if not act.any():
                break
            mid = (lo + hi + 1) // 2
            em = e[mid]
            ok = (em == 0) | (ei / em >= tau64)
            lo = np.where(act & ok, mid, lo)
            hi = np.where(act & ~ok, mid - 1, hi)
    return lo.astype(np.int32)


def interleave_owner(rows, block: int, world: int):
    """rank that owns query row(s) `rows` under the row interleave: blocks are dealt boustrophedon, block b = q * world + r goes to
    rank r in even cycles q and to rank world - 1 - r in odd ones (csrc/common.cuh, RowMap)"""
    b = np.asarray(rows, dtype=np.int64) // block
    q, r = b // world, b % world
    return np.where(q & 1, world - 1 - r, r)


def interleave_pair_counts(n: int, block: int, world: int, hi: Optional[np.ndarray] = None, z0: int = 0) -> np.ndarray:
    """pairs each rank evaluates under the row interleave (the product's own counter is stats()['evaluated'])"""
    i = np.arange(n, dtype=np.int64)
    h = np.full(n, n - 1, dtype=np.int64) if hi is None else np.asarray(hi, dtype=np.int64)
    cnt = np.maximum(h - np.maximum(i + 1, z0) + 1, 0)
    own = interleave_owner(i, block, world)
    return np.array([cnt[own == r].sum() for r in range(world)], dtype=np.int64)


def first_nonzero(cards: np.ndarray) -> int:
    nz = np.nonzero(np.asarray(cards) >= 1.0)[0]
    return int(nz[0]) if nz.size else int(len(cards))


def shard_rows(n: int, world: int, hi: Optional[np.ndarray] = None, z0: int = 0) -> np.ndarray:
    """row boundaries [world+1] with (nearly) equal pair counts per shard (libselhost selhost_shard_rows)"""
    bounds = np.zeros(world + 1, dtype=np.int64)
    hp = None
    if hi is not None:
        hi = np.ascontiguousarray(hi, dtype=np.int32)
        hp = hi.ctypes.data
    rc = host_lib().selhost_shard_rows(n, hp, z0, world, bounds.ctypes.data)
    if rc:
        raise RuntimeError(host_lib().selhost_last_error().decode())
    return bounds


def pair_counts(n: int, bounds: np.ndarray, hi: Optional[np.ndarray] = None, z0: int = 0) -> np.ndarray:
    i = np.arange(n, dtype=np.int64)
    h = np.full(n, n - 1, dtype=np.int64) if hi is None else hi.astype(np.int64)
    cnt = np.maximum(h - np.maximum(i + 1, z0) + 1, 0)
    return np.array([cnt[bounds[r]:bounds[r + 1]].sum() for r in range(len(bounds) - 1)], dtype=np.int64)


def gather_pairs(local_pairs, dist, device=None, capacity: Optional[int] = None) -> np.ndarray:
    """all_gather of variable-length selhip_pair_t lists: returns the concatenation over ranks, sorted by
    (i,k) = the reference's print order.  `local_pairs` is a PAIR_DTYPE array (host) -- the device-resident
    variant used by bench.py avoids the host hop (Selector.copy_results_to + all_gather_into_tensor)."""
    import torch

    world = dist.get_world_size()
    dev = device if device is not None else torch.device("cpu")
    cnt = torch.tensor([len(local_pairs)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    cap = capacity if capacity is not None else max(max(counts), 1)
    if max(counts) > cap:
        raise RuntimeError("gather capacity too small")
    buf = np.zeros(cap, dtype=PAIR_DTYPE)
    buf[:len(local_pairs)] = local_pairs
    send = torch.from_numpy(buf.view(np.int64).reshape(cap, 2).copy()).to(dev)
    recv = [torch.zeros_like(send) for _ in range(world)]
    dist.all_gather(recv, send)
    parts = [r.cpu().numpy().reshape(-1).view(PAIR_DTYPE)[:c] for r, c in zip(recv, counts)]
    out = np.concatenate(parts) if parts else np.zeros(0, dtype=PAIR_DTYPE)
    order = np.lexsort((out["k"], out["i"]))
    return out[order]
