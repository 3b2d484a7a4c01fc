"""Sketch construction on the GPU: the `build_sketch` step of the reference (src/build_sketch.cpp) behind
selhip_build_sketches (include/selection_hip.h section 4b).  FASTA parsing happens in libselhost (C++)."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from ._lib import check, hip_lib, host_lib


def fasta_codes(path: str) -> np.ndarray:
    """one byte per base: 0..3 = A,C,G,T (either case), 4 = k-mer window reset (other character / record start)"""
    h = host_lib()
    n = h.selhost_fasta_codes(str(path).encode(), None, 0)
    if n < 0:
        raise RuntimeError(h.selhost_last_error().decode())
    out = np.empty(max(n, 1), dtype=np.uint8)
    h.selhost_fasta_codes(str(path).encode(), out.ctypes.data, n)
    return out[:n]


def smh_vecsize(m_arg: int) -> int:
    return int(host_lib().selhost_smh_vecsize(m_arg))


def build_sketches(fasta_paths: Sequence[str], m: int = 0, p_aux: int = 0, device: int = 0, k: int = 31):
    """returns (hll u8 [n,16384], smh u64 [n, vecsize(m)] or None, aux u8 [n, 1<<p_aux] or None) as numpy arrays"""
    import torch

    lib = hip_lib()
    codes = [fasta_codes(p) for p in fasta_paths]
    n = len(codes)
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(c) for c in codes])
    flat = np.concatenate(codes) if n else np.zeros(0, dtype=np.uint8)
    dev = torch.device("cuda", device)
    mv = smh_vecsize(m) if m else 0
    with torch.cuda.device(dev):
        d_codes = torch.from_numpy(flat if flat.size else np.zeros(1, dtype=np.uint8)).to(dev)
        d_off = torch.from_numpy(offsets).to(dev)
        d_hll = torch.zeros((n, 16384), dtype=torch.uint8, device=dev)
        d_smh = torch.zeros((n, mv), dtype=torch.int64, device=dev) if mv else None
        d_aux = torch.zeros((n, 1 << p_aux), dtype=torch.uint8, device=dev) if p_aux else None
        torch.cuda.synchronize(dev)
        check(lib.selhip_build_sketches(d_codes.data_ptr(), d_off.data_ptr(), n, k, mv, p_aux, d_hll.data_ptr(),
                                        d_smh.data_ptr() if mv else None, d_aux.data_ptr() if p_aux else None, None))
        check(lib.selhip_device_synchronize())
        return (d_hll.cpu().numpy(), d_smh.cpu().numpy().view(np.uint64) if mv else None,
                d_aux.cpu().numpy() if p_aux else None)
