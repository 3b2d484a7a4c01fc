"""Synthetic sketch sets (csrc/synth.hpp) for the BASELINE.json configurations.

The reference's timing harness rebuilds SuperMinHash sketches from FASTA (experiments/src/
time_smh_cuda.cpp:181-211); there are no genomes here, so sketches are synthesised statistically:
clusters of `cluster_size` genomes share n_sh random elements, every member adds n_sh*f private ones
(f in {0.005, 0.02, 0.1}), each element is hashed once into the p=14 HLL, the auxiliary HLL and one
SuperMinHash bucket.  Host (libselhost) and device (libselhip) generators are bit-identical.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from ._lib import Synth, check, hip_lib, host_lib


@dataclass(frozen=True)
class SynthConfig:
    name: str
    n_genomes: int
    m: int
    tau: float
    seed: int
    p_aux: int = 0
    cluster_size: int = 10
    mode: int = 0                 # 0 flat (every pair passes CB), 1 spread (CB prunes)
    n_sh_lo: int = 100_000
    n_sh_hi: int = 100_000

    def struct(self) -> Synth:
        return Synth(self.seed, self.n_genomes, self.m, self.p_aux, self.cluster_size, self.mode, self.n_sh_lo, self.n_sh_hi)

    def scaled(self, n_genomes: int) -> "SynthConfig":
        return SynthConfig(f"{self.name}@N={n_genomes}", n_genomes, self.m, self.tau, self.seed, self.p_aux,
                           self.cluster_size, self.mode, self.n_sh_lo, self.n_sh_hi)


# BASELINE.json `configs` (index = position there); seeds 0x5EED0000 + index as in SURVEY.md section 8(d)
SYNTH_CONFIGS = {
    "cfg2": SynthConfig("cfg2: 1000 genomes, smh_a m=256, tau=0.9", 1_000, 256, 0.9, 0x5EED0001),
    "cfg3": SynthConfig("cfg3: 10000 genomes, smh_a m=512, tau=0.8", 10_000, 512, 0.8, 0x5EED0002),
    "cfg4": SynthConfig("cfg4: 50000 genomes, smh_a m=512, tau=0.8", 50_000, 512, 0.8, 0x5EED0003),
    "cfg5": SynthConfig("cfg5: 100000 genomes, hll_a p=8 + smh_a m=1024, tau=0.9", 100_000, 1024, 0.9, 0x5EED0004, p_aux=8),
    # CB-pruned variants (log-uniform shared-set sizes)
    "cfg2-spread": SynthConfig("cfg2-spread", 1_000, 256, 0.9, 0x5EED0011, mode=1, n_sh_lo=8_000, n_sh_hi=200_000),
    "cfg3-spread": SynthConfig("cfg3-spread", 10_000, 512, 0.8, 0x5EED0012, mode=1, n_sh_lo=8_000, n_sh_hi=200_000),
}


def synth_host(cfg: SynthConfig, g_range: Optional[Tuple[int, int]] = None, threads: int = 8):
    """(hll u8 [g,16384], aux u64 [g,m], aux_hll u8 [g,1<<p_aux]) in GENERATION order, on the host."""
    b, e = g_range if g_range else (0, cfg.n_genomes)
    n = e - b
    hll = np.empty((n, 16384), dtype=np.uint8)
    aux = np.empty((n, cfg.m), dtype=np.uint64)
    aux_hll = np.empty((n, (1 << cfg.p_aux) if cfg.p_aux else 0), dtype=np.uint8)
    sp = cfg.struct()
    rc = host_lib().selhost_synth_generate(C.byref(sp), b, e, hll.ctypes.data, aux.ctypes.data,
                                           aux_hll.ctypes.data if cfg.p_aux else None, threads)
    if rc:
        raise RuntimeError(host_lib().selhost_last_error().decode())
    return hll, aux, aux_hll


def synth_device(cfg: SynthConfig, device: int = 0, sort: bool = True):
    """Generates the set directly in HBM (torch tensors) and, if `sort`, brings it into ascending-
    cardinality order the way the reference driver does (report() -> std::sort -> flatten,
    selection_cuda.cpp:106-143): returns (hll, aux(int64 view of the u64 buckets), cards, perm, aux_hll)."""
    import torch

    lib = hip_lib()
    dev = torch.device("cuda", device)
    n = cfg.n_genomes
    with torch.cuda.device(dev):
        hll = torch.empty((n, 16384), dtype=torch.uint8, device=dev)
        aux = torch.empty((n, cfg.m), dtype=torch.int64, device=dev)
        aux_hll = torch.empty((n, (1 << cfg.p_aux) if cfg.p_aux else 0), dtype=torch.uint8, device=dev)
        sp = cfg.struct()
        torch.cuda.synchronize(dev)
        check(lib.selhip_synth_generate(C.byref(sp), 0, n, hll.data_ptr(), aux.data_ptr(),
                                        aux_hll.data_ptr() if cfg.p_aux else None, None))
        check(lib.selhip_device_synchronize())
        ctx = C.c_void_p()
        check(lib.selhip_ctx_create(C.byref(ctx), device))
        try:
            cards = torch.empty(n, dtype=torch.float64, device=dev)
            check(lib.selhip_hll_cards(ctx, hll.data_ptr(), n, 14, cards.data_ptr()), ctx)
        finally:
            lib.selhip_ctx_destroy(ctx)
        if not sort:
            return hll, aux, cards, None, aux_hll
        from .selection import sort_by_card
        perm = sort_by_card(cards.cpu().numpy())
        perm_t = torch.from_numpy(perm).to(dev)
        hll_s = torch.empty_like(hll)
        aux_s = torch.empty_like(aux)
        check(lib.selhip_permute_rows(hll.data_ptr(), hll_s.data_ptr(), perm_t.data_ptr(), n, 16384, None))
        check(lib.selhip_permute_rows(aux.data_ptr(), aux_s.data_ptr(), perm_t.data_ptr(), n, cfg.m * 8, None))
        aux_hll_s = aux_hll
        if cfg.p_aux:
            aux_hll_s = torch.empty_like(aux_hll)
            check(lib.selhip_permute_rows(aux_hll.data_ptr(), aux_hll_s.data_ptr(), perm_t.data_ptr(), n, 1 << cfg.p_aux, None))
        check(lib.selhip_device_synchronize())
        cards_s = cards[perm_t.long()].contiguous()
        return hll_s, aux_s, cards_s, perm, aux_hll_s


def harden(aux, frac: float = 0.25, alphabet: int = 2, seed: int = 0xD15EA5E) -> int:
    """The "harder" workload: a seeded `frac` of the genomes (by position in `aux`) gets degenerate SuperMinHash buckets
    (values mod `alphabet`), so that pairs among them pass a band by chance -- alphabet 2 and bands of 8 rows: 22 % of those
    pairs -- i.e. ~frac^2 * 0.22 of ALL pairs reach stage 2 instead of the 0.09 % of the plain recipe.  Works in place on a
    numpy uint64 array or on a torch int64 tensor (device or host); returns the number of degenerate genomes."""
    n = aux.shape[0]
    mask = np.random.default_rng(seed).random(n) < frac
    if isinstance(aux, np.ndarray):
        aux[mask] = aux[mask] % np.uint64(alphabet)
    else:
        import torch
        idx = torch.from_numpy(np.nonzero(mask)[0]).to(aux.device)
        aux[idx] = aux[idx] % alphabet              # int64 view of u64 bucket values < 2^32
    return int(mask.sum())


def stream_model(m: int, n_rows: int) -> dict:
    """instruction counts of smh_stream_kernel per (query, candidate) pair, from its ISA (csrc/kernel_stream.cuh): lane l owns
    B = m/64 contiguous buckets; B v_cmp_eq_u64; B-1 s_and_b64 (r >= B: + 2 per shift-AND step over r/B lanes); 2 scalar
    instructions for the test and the branch"""
    B = max(2, m // 64)
    steps = 0
    L = n_rows // B
    while L > 1:
        steps += 1
        L //= 2
    return {"valu_wave_instr_per_pair": B, "salu_per_pair": (B - 1) + 2 * steps + 2, "buckets_per_lane": B}
