"""Host-side mirror of the reference's GPU selection driver (src/selection_cuda.cpp:59-189) on top of
the C ABI.  Same steps, same names: load_file_list -> read sketches -> report() -> sort by cardinality
-> banding -> flatten -> upload -> launch -> copy back -> print; all the work happens in
libselhost.so (C++) and libselhip.so (HIP)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import (ALGO_AUTO, BANDING_CPU, CRIT_HLL_A, CRIT_HLL_AN, CRIT_HLL_A_SMH_A, CRIT_SMH_A, FP_FMA, MODE_CB_SMH, Pair,
                   check, hip_lib, host_lib)

# layout of selhip_pair_t {int32 i, k; double jaccard}
PAIR_DTYPE = np.dtype([("i", "<i4"), ("k", "<i4"), ("jaccard", "<f8")], align=True)
assert PAIR_DTYPE.itemsize == C.sizeof(Pair) == 16


def banding(m: int, tau: float, variant: int = BANDING_CPU) -> Tuple[int, int]:
    """(n_rows, n_bands) of src/selection.cpp:258-267 (CPU variant) or selection_cuda.cpp:119-128."""
    r, b = C.c_int(), C.c_int()
    host_lib().selhost_banding(m, np.float32(tau), variant, C.byref(r), C.byref(b))
    return r.value, b.value


def sort_by_card(cards: np.ndarray) -> np.ndarray:
    """perm[rank] = original index; libstdc++ std::sort with the comparator of selection.cpp:251-256."""
    cards = np.ascontiguousarray(cards, dtype=np.float64)
    perm = np.empty(cards.shape[0], dtype=np.int32)
    rc = host_lib().selhost_sort_by_card(cards.ctypes.data, cards.shape[0], perm.ctypes.data)
    if rc:
        raise RuntimeError(host_lib().selhost_last_error().decode())
    return perm


@dataclass
class Dataset:
    names: list
    hll: np.ndarray        # [n, 16384] u8, rank order
    aux: np.ndarray        # [n, m] u64
    aux_hll: np.ndarray    # [n, 1 << p_aux] u8 (empty when p_aux == 0)
    cards: np.ndarray      # [n] f64 ascending


def load_dataset(list_file: str, m: int, p_aux: int = 0, fp_mode: int = FP_FMA, threads: int = 8) -> Dataset:
    """selection_cuda.cpp:90-143: read <name>.hll / <name>.smh<m>, report(), sort, flatten."""
    h = host_lib()
    ds = C.c_void_p()
    rc = h.selhost_dataset_load(C.byref(ds), list_file.encode(), m, p_aux, fp_mode, threads)
    if rc:
        raise RuntimeError(f"selhost error {rc}: {h.selhost_last_error().decode()}")
    try:
        n = h.selhost_dataset_size(ds)
        def arr(ptr, shape, dt):
            count = int(np.prod(shape))
            if n == 0 or count == 0:
                return np.zeros(shape, dtype=dt)
            buf = (C.c_uint8 * (count * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt).reshape(shape).copy()
        hll = arr(h.selhost_dataset_hll(ds), (n, 16384), np.uint8)
        aux = arr(h.selhost_dataset_aux(ds), (n, m), np.uint64)
        aux_hll = arr(h.selhost_dataset_aux_hll(ds), (n, (1 << p_aux) if p_aux else 0), np.uint8)
        cards = arr(h.selhost_dataset_cards(ds), (n,), np.float64)
        names = [h.selhost_dataset_name(ds, r).decode() for r in range(n)]
    finally:
        h.selhost_dataset_free(ds)
    return Dataset(names, hll, aux, aux_hll, cards)


def format_lines(names: Sequence[str], pairs: np.ndarray) -> str:
    """'fn1 fn2 <std::to_string(J)>\\n' per selected pair (selection.cpp:288)."""
    h = host_lib()
    buf = C.create_string_buffer(16384)
    out = []
    for rec in pairs:
        w = h.selhost_format_line(names[rec["i"]].encode(), names[rec["k"]].encode(), float(rec["jaccard"]), buf, len(buf))
        if w < 0:
            raise RuntimeError("line too long")
        out.append(buf.raw[:w].decode())
    return "".join(out)


def write_results(path: str, pairs: np.ndarray, names: Sequence[str], tau: float = 0.0):
    """binary result file of include/selection_host.h (records + the name table their ranks refer to)"""
    h = host_lib()
    rec = np.ascontiguousarray(pairs, dtype=PAIR_DTYPE)
    enc = [n.encode() for n in names]
    arr = (C.c_char_p * len(enc))(*enc)
    rc = h.selhost_write_results(str(path).encode(), rec.ctypes.data if len(rec) else None, len(rec), arr if enc else None, len(enc),
                                 np.float32(tau))
    if rc:
        raise RuntimeError(h.selhost_last_error().decode())


def read_results(path: str):
    """-> (pairs[PAIR_DTYPE], names, tau, text) of a result file; text = the reference's stdout form"""
    h = host_lib()
    r = C.c_void_p()
    rc = h.selhost_read_results(C.byref(r), str(path).encode())
    if rc:
        raise RuntimeError(h.selhost_last_error().decode())
    try:
        cnt = h.selhost_results_count(r)
        nn = h.selhost_results_names(r)
        pairs = np.zeros(cnt, dtype=PAIR_DTYPE)
        if cnt:
            C.memmove(pairs.ctypes.data, h.selhost_results_pairs(r), cnt * PAIR_DTYPE.itemsize)
        names = [h.selhost_results_name(r, g).decode() for g in range(nn)]
        need = h.selhost_results_text(r, None, 0)
        text = ""
        if need > 0:
            buf = C.create_string_buffer(need + 1)
            h.selhost_results_text(r, buf, need + 1)
            text = buf.raw[:need].decode()
        return pairs, names, float(h.selhost_results_tau(r)), text
    finally:
        h.selhost_results_free(r)


class Selector:
    """One selhip context = one GPU (`selhip_ctx_*`, include/selection_hip.h section 2)."""

    DEFAULT_PARAMS: dict = {}      # selhip_ctx_set_param applied to every new context (the test-suite groups small sets too: "group_min_n" = 0)

    def __init__(self, device: int = 0, fp_mode: int = FP_FMA, stream: Optional[int] = None):
        self._lib = hip_lib()
        self._ctx = C.c_void_p()
        check(self._lib.selhip_ctx_create(C.byref(self._ctx), device))
        check(self._lib.selhip_ctx_set_fp_mode(self._ctx, fp_mode), self._ctx)
        for name, value in self.DEFAULT_PARAMS.items():
            check(self._lib.selhip_ctx_set_param(self._ctx, name.encode(), int(value)), self._ctx)
        if stream is not None:
            check(self._lib.selhip_ctx_set_stream(self._ctx, C.c_void_p(stream)), self._ctx)
        self.n = 0
        self.m = 0
        self._keep = None

    def close(self):
        if self._ctx:
            self._lib.selhip_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- sketches ---------------------------------------------------------------------------------
    def upload(self, hll: np.ndarray, aux: np.ndarray, cards: Optional[np.ndarray] = None, p_hll: int = 14):
        hll = np.ascontiguousarray(hll, dtype=np.uint8)
        aux = np.ascontiguousarray(aux, dtype=np.uint64)
        n, m = aux.shape
        assert hll.shape == (n, 1 << p_hll), (hll.shape, n, p_hll)
        cp = None
        if cards is not None:
            cards = np.ascontiguousarray(cards, dtype=np.float64)
            assert cards.shape == (n,)
            cp = cards.ctypes.data
        check(self._lib.selhip_ctx_upload(self._ctx, hll.ctypes.data, aux.ctypes.data, cp, n, m, p_hll), self._ctx)
        self.n, self.m = n, m

    def attach(self, hll_t, aux_t, cards_t=None, p_hll: int = 14):
        """torch CUDA tensors: hll uint8 [n, 1<<p], aux int64 [n, m] (bit pattern of the u64 buckets),
        cards float64 [n] ascending or None."""
        n, m = aux_t.shape
        assert hll_t.is_cuda and aux_t.is_cuda and hll_t.is_contiguous() and aux_t.is_contiguous()
        assert tuple(hll_t.shape) == (n, 1 << p_hll) and aux_t.element_size() == 8 and hll_t.element_size() == 1
        cp = None
        if cards_t is not None:
            assert cards_t.is_cuda and cards_t.is_contiguous() and cards_t.element_size() == 8 and cards_t.shape[0] == n
            cp = cards_t.data_ptr()
        check(self._lib.selhip_ctx_attach(self._ctx, hll_t.data_ptr(), aux_t.data_ptr(), cp, n, m, p_hll), self._ctx)
        self._keep = (hll_t, aux_t, cards_t)
        self.n, self.m = n, m

    def upload_aux_hll(self, aux_hll: np.ndarray, p_aux: int):
        """auxiliary HLL sketches (.hll_<p> files) for the hll_a / hll_an criteria, rank order"""
        aux_hll = np.ascontiguousarray(aux_hll, dtype=np.uint8)
        assert aux_hll.shape == (self.n, 1 << p_aux)
        check(self._lib.selhip_ctx_upload_aux_hll(self._ctx, aux_hll.ctypes.data, p_aux), self._ctx)

    def attach_aux_hll(self, aux_hll_t, p_aux: int):
        assert aux_hll_t.is_cuda and aux_hll_t.is_contiguous() and tuple(aux_hll_t.shape) == (self.n, 1 << p_aux)
        check(self._lib.selhip_ctx_attach_aux_hll(self._ctx, aux_hll_t.data_ptr(), p_aux), self._ctx)
        self._keep_aux = aux_hll_t

    def set_pipeline(self, chunks: int):
        """-1 auto, 0 off, 2..8 forced: overlap of stage 1 (next row chunk) with stage 2 (previous chunk)"""
        check(self._lib.selhip_ctx_set_pipeline(self._ctx, chunks), self._ctx)

    def set_row_interleave(self, block_rows: int, n_parts: int, part: int):
        """the following runs evaluate only this part's row blocks (of block_rows rows, dealt to the n_parts parts boustrophedon:
        distributed.interleave_owner)"""
        check(self._lib.selhip_ctx_set_row_interleave(self._ctx, block_rows, n_parts, part), self._ctx)

    def set_candidate_begin(self, k_min: int):
        """rectangular passes: the following runs only take candidates k >= k_min (reset by upload/attach)"""
        check(self._lib.selhip_ctx_set_candidate_begin(self._ctx, k_min), self._ctx)

    def set_param(self, name: str, value: int):
        check(self._lib.selhip_ctx_set_param(self._ctx, name.encode(), value), self._ctx)

    def get_param(self, name: str) -> int:
        v = C.c_int(0)
        check(self._lib.selhip_ctx_get_param(self._ctx, name.encode(), C.byref(v)), self._ctx)
        return int(v.value)

    def set_stage2_grouping(self, enable: bool):
        check(self._lib.selhip_ctx_set_stage2_grouping(self._ctx, 1 if enable else 0), self._ctx)

    def set_criterion(self, criterion: int):
        check(self._lib.selhip_ctx_set_criterion(self._ctx, criterion), self._ctx)
        self.criterion = criterion

    def cards(self) -> np.ndarray:
        out = np.empty(self.n, dtype=np.float64)
        check(self._lib.selhip_ctx_get_cards(self._ctx, out.ctypes.data), self._ctx)
        return out

    # -- the hot path -------------------------------------------------------------------------------
    def run(self, tau: float, mode: int = MODE_CB_SMH, n_rows: Optional[int] = None, n_bands: Optional[int] = None,
            rows: Optional[Tuple[int, int]] = None, algo: int = ALGO_AUTO, fetch: bool = True):
        if n_rows is None or n_bands is None:
            n_rows, n_bands = banding(self.m, tau) if self.m else (1, 1)
        rb, re = rows if rows is not None else (0, self.n)
        check(self._lib.selhip_ctx_run(self._ctx, mode, algo, np.float32(tau), n_rows, n_bands, rb, re), self._ctx)
        return self.fetch() if fetch else None

    def run_async(self, tau: float, mode: int = MODE_CB_SMH, n_rows: Optional[int] = None, n_bands: Optional[int] = None,
                  rows: Optional[Tuple[int, int]] = None, algo: int = ALGO_AUTO):
        if n_rows is None or n_bands is None:
            n_rows, n_bands = banding(self.m, tau)
        rb, re = rows if rows is not None else (0, self.n)
        check(self._lib.selhip_ctx_run_async(self._ctx, mode, algo, np.float32(tau), n_rows, n_bands, rb, re), self._ctx)

    def finish(self):
        check(self._lib.selhip_ctx_finish(self._ctx), self._ctx)

    def result_count(self) -> int:
        return int(check(self._lib.selhip_ctx_result_count(self._ctx), self._ctx))

    def fetch(self) -> np.ndarray:
        cnt = self.result_count()
        out = np.zeros(cnt, dtype=PAIR_DTYPE)
        check(self._lib.selhip_ctx_fetch(self._ctx, out.ctypes.data if cnt else None, cnt), self._ctx)
        return out

    def result_device(self) -> Tuple[int, int]:
        """(device pointer to the unsorted selhip_pair_t list, count)"""
        p, cnt = C.c_void_p(), C.c_int64()
        check(self._lib.selhip_ctx_result_device(self._ctx, C.byref(p), C.byref(cnt)), self._ctx)
        return (p.value or 0), cnt.value

    def copy_results_to(self, tensor) -> int:
        """D2D copy of the unsorted result records into a torch CUDA uint8/any tensor (>= 16 B per record)."""
        cap = tensor.numel() * tensor.element_size() // PAIR_DTYPE.itemsize
        check(self._lib.selhip_ctx_copy_results(self._ctx, tensor.data_ptr(), cap), self._ctx)
        return min(cap, self.result_count())

    def copy_results_framed(self, tensor) -> int:
        """D2D: tensor[0] (16 B) = {count, 0}, records from tensor[1:]; returns the (host-known) count"""
        cap = tensor.numel() * tensor.element_size() // PAIR_DTYPE.itemsize - 1
        rc = self._lib.selhip_ctx_copy_results_framed(self._ctx, tensor.data_ptr(), cap)
        if rc not in (0, -3):
            check(rc, self._ctx)
        return self.result_count()

    def copy_results_framed_async(self, tensor):
        """the same frame enqueued behind a pass that is still running (between run_async and finish): header from the
        device-side counter, payload = the whole capacity of `tensor`; check result_count() and last_attempts() after finish"""
        cap = tensor.numel() * tensor.element_size() // PAIR_DTYPE.itemsize - 1
        check(self._lib.selhip_ctx_copy_results_framed_async(self._ctx, tensor.data_ptr(), cap), self._ctx)
        return cap

    def last_attempts(self) -> int:
        return int(check(self._lib.selhip_ctx_last_attempts(self._ctx), self._ctx))

    def stats(self) -> dict:
        st = (C.c_int64 * 4)()
        check(self._lib.selhip_ctx_stats(self._ctx, st), self._ctx)
        return {"evaluated": st[0], "survivors": st[1], "selected": st[2], "candidates": st[3]}

    def timing(self, enable=True):
        """False/0 = off, True/1 = every kernel scope, 2 = only the dominant stage-1 kernel (resets the figures)"""
        check(self._lib.selhip_ctx_timing(self._ctx, int(enable)), self._ctx)

    def kernel_ms(self, name: str) -> float:
        """device ms per pass spent in the named kernel (sum over its launches of one pass)"""
        return float(self._lib.selhip_ctx_kernel_ms(self._ctx, name.encode()))

    def kernel_launches(self, name: str) -> float:
        return float(self._lib.selhip_ctx_kernel_launches(self._ctx, name.encode()))


def select_from_filelist(list_file: str, tau: float, aux_bytes: int, mode: int = MODE_CB_SMH, device: int = 0,
                         fp_mode: int = FP_FMA, algo: int = ALGO_AUTO, criterion: str = "smh_a") -> str:
    """The whole of selection_cuda.cpp main() (criterion smh_a) -- and of selection.cpp's hll_a / hll_an
    branches (:122-227): returns the text the CPU reference prints for `-c criterion -a aux_bytes -h tau`."""
    if criterion == "smh_a":
        m, p_aux, crit = aux_bytes // 8, 0, CRIT_SMH_A                       # selection.cpp:231
    elif criterion in ("hll_a", "hll_an"):
        m, p_aux = 0, (aux_bytes & -aux_bytes).bit_length() - 1               # __builtin_ctz(aux_bytes), selection.cpp:125
        crit = CRIT_HLL_A if criterion == "hll_a" else CRIT_HLL_AN
    else:
        raise ValueError("Option -c invalid. The accepted criteria are hll_a, hll_an and smh_a.")
    ds = load_dataset(list_file, m, p_aux, fp_mode)
    n_rows, n_bands = banding(m, tau) if m else (1, 1)
    with Selector(device, fp_mode) as sel:
        aux = ds.aux if m else np.zeros((len(ds.names), 1), dtype=np.uint64)
        sel.upload(ds.hll, aux, ds.cards)
        if p_aux:
            sel.upload_aux_hll(ds.aux_hll, p_aux)
        sel.set_criterion(crit)
        pairs = sel.run(tau, mode, n_rows, n_bands, algo=algo)
    return format_lines(ds.names, pairs)


def multi_select(devices: Sequence[int], hll: np.ndarray, aux: np.ndarray, cards: np.ndarray, tau: float,
                 mode: int = MODE_CB_SMH, n_rows: Optional[int] = None, n_bands: Optional[int] = None, algo: int = ALGO_AUTO,
                 fp_mode: int = FP_FMA, gather: int = 2, criterion: int = 0, aux_hll: Optional[np.ndarray] = None, p_aux: int = 0):
    """selhip_multi_select: one process, one context per listed device, interleaved row blocks, RCCL (or host) gather.
    gather: 0 host merge, 1 RCCL required, 2 RCCL if available; criterion / aux_hll / p_aux as for Selector (hll_a, hll_an,
    the two-stage criterion of BASELINE configs[4]).  Returns (pairs sorted by (i,k), stats dict)."""
    lib = hip_lib()
    hll = np.ascontiguousarray(hll, dtype=np.uint8)
    aux = np.ascontiguousarray(aux, dtype=np.uint64)
    cards = np.ascontiguousarray(cards, dtype=np.float64)
    n, m = aux.shape
    if n_rows is None or n_bands is None:
        n_rows, n_bands = banding(m, tau)
    devs = (C.c_int * len(devices))(*devices)
    ah = np.ascontiguousarray(aux_hll, dtype=np.uint8) if aux_hll is not None and criterion != 0 else None
    cap = 1 << 16
    while True:
        out = np.zeros(cap, dtype=PAIR_DTYPE)
        cnt = C.c_int64()
        st = (C.c_int64 * 4)()
        rc = lib.selhip_multi_select(devs, len(devices), hll.ctypes.data, aux.ctypes.data, cards.ctypes.data,
                                     ah.ctypes.data if ah is not None else None, p_aux, criterion, n, m, 14, mode, algo,
                                     fp_mode, np.float32(tau), n_rows, n_bands, gather, out.ctypes.data, cap, C.byref(cnt), st)
        if rc == -3:
            cap = int(cnt.value)
            continue
        check(rc)
        return out[:cnt.value], {"evaluated": st[0], "survivors": st[1], "selected": st[2], "candidates": st[3]}


def ooc_select(hll: np.ndarray, aux: np.ndarray, cards: np.ndarray, tau: float, block_genomes: int,
               mode: int = MODE_CB_SMH, n_rows: Optional[int] = None, n_bands: Optional[int] = None, algo: int = ALGO_AUTO,
               fp_mode: int = FP_FMA, n_streams: int = 2, device: int = 0, criterion: int = 0,
               aux_hll: Optional[np.ndarray] = None, p_aux: int = 0):
    """selhip_ooc_select: the sketches stay in host memory, at most n_streams * 2 * block_genomes of them are on the device
    at a time; same result as one in-core pass.  Returns (pairs with global ranks sorted by (i,k), stats dict)."""
    lib = hip_lib()
    hll = np.ascontiguousarray(hll, dtype=np.uint8)
    aux = np.ascontiguousarray(aux, dtype=np.uint64)
    cards = np.ascontiguousarray(cards, dtype=np.float64)
    n, m = aux.shape
    if n_rows is None or n_bands is None:
        n_rows, n_bands = banding(m, tau)
    ah = np.ascontiguousarray(aux_hll, dtype=np.uint8) if aux_hll is not None and criterion != 0 else None
    cap = 1 << 16
    while True:
        out = np.zeros(cap, dtype=PAIR_DTYPE)
        cnt = C.c_int64()
        st = (C.c_int64 * 4)()
        rc = lib.selhip_ooc_select(device, hll.ctypes.data, aux.ctypes.data, cards.ctypes.data,
                                   ah.ctypes.data if ah is not None else None, p_aux, criterion, n, m, 14, mode, algo, fp_mode,
                                   np.float32(tau), n_rows, n_bands, block_genomes, n_streams, out.ctypes.data, cap, C.byref(cnt), st)
        if rc == -3:
            cap = int(cnt.value)
            continue
        check(rc)
        return out[:cnt.value], {"evaluated": st[0], "survivors": st[1], "selected": st[2], "candidates": st[3]}
