"""ctypes view of oracle/liboracle.so -- the CPU ORACLE (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORC_PAIR = np.dtype([("i", "<i4"), ("k", "<i4"), ("jacc", "<f8")], align=True)


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(os.environ.get("ORACLE_LIB", str(ROOT / "oracle" / "liboracle.so")))   # ORACLE_LIB: a sanitizer build (scripts/asan_host.sh)
        L = self.lib
        L.orc_ertl_ml_estimate_ex.restype = C.c_double
        L.orc_ertl_ml_estimate_ex.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_double, C.c_int]
        L.orc_hll_report.restype = C.c_double
        L.orc_hll_report.argtypes = [C.c_void_p, C.c_uint]
        L.orc_hll_union_size.restype = C.c_double
        L.orc_hll_union_size.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        L.orc_union_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_smh_a.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint]
        L.orc_banding.argtypes = [C.c_uint, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_banding_cuda_variant.argtypes = [C.c_uint, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_select.restype = C.c_int64
        L.orc_select.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.c_void_p, C.c_int64,
                                 C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        L.orc_set_fma.argtypes = [C.c_int]
        L.orc_std_sort_perm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.orc_std_sort_perm.restype = None
        L.orc_cb.argtypes = [C.c_double, C.c_double, C.c_double]

    def set_fma(self, on):
        self.lib.orc_set_fma(1 if on else 0)

    def report(self, core, p=14):
        core = np.ascontiguousarray(core, dtype=np.uint8)
        return self.lib.orc_hll_report(core.ctypes.data, p)

    def cards(self, hll, p=14):
        return np.array([self.report(hll[g], p) for g in range(hll.shape[0])], dtype=np.float64)

    def union_size(self, a, b, p=14):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        return self.lib.orc_hll_union_size(a.ctypes.data, b.ctypes.data, p)

    def union_hist(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        out = np.zeros(64, dtype=np.uint32)
        self.lib.orc_union_histogram(a.ctypes.data, b.ctypes.data, a.size, out.ctypes.data)
        return out

    def estimate(self, counts, p, fma=1):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        return self.lib.orc_ertl_ml_estimate_ex(counts.ctypes.data, p, 64 - p, 1e-2, fma)

    def smh_a(self, v1, v2, n_rows, n_bands):
        v1 = np.ascontiguousarray(v1, dtype=np.uint64)
        v2 = np.ascontiguousarray(v2, dtype=np.uint64)
        return bool(self.lib.orc_smh_a(v1.ctypes.data, v2.ctypes.data, v1.size, n_rows, n_bands))

    def std_sort_perm(self, cards):
        """permutation of selection.cpp:251-256's std::sort (GNU libstdc++ introsort restated), ties included"""
        cards = np.ascontiguousarray(cards, dtype=np.float64)
        perm = np.empty(len(cards), dtype=np.int64)
        self.lib.orc_std_sort_perm(cards.ctypes.data, len(cards), perm.ctypes.data)
        return perm

    def banding(self, m, tau, cuda_variant=False):
        r, b = C.c_int(), C.c_int()
        (self.lib.orc_banding_cuda_variant if cuda_variant else self.lib.orc_banding)(m, np.float32(tau), C.byref(r), C.byref(b))
        return r.value, b.value

    def select(self, hll, aux, cards, tau, n_rows, n_bands, use_cb=True, criterion=0, aux_hll=None, p=14, p_aux=8,
               threads=8):
        """orc_select on arrays in rank order; returns (pairs[ORC_PAIR], stats dict)"""
        hll = np.ascontiguousarray(hll, dtype=np.uint8)
        cards = np.ascontiguousarray(cards, dtype=np.float64)
        n = cards.shape[0]
        m = 0
        auxp = None
        if aux is not None:
            aux = np.ascontiguousarray(aux, dtype=np.uint64)
            m = aux.shape[1]
            auxp = aux.ctypes.data
        ahp = None
        if aux_hll is not None and aux_hll.size:
            aux_hll = np.ascontiguousarray(aux_hll, dtype=np.uint8)
            ahp = aux_hll.ctypes.data
        cap = 1 << 16
        while True:
            out = np.zeros(cap, dtype=ORC_PAIR)
            st = (C.c_int64 * 2)()
            cnt = self.lib.orc_select(hll.ctypes.data, p, auxp, m, ahp, p_aux, cards.ctypes.data, n, np.float32(tau),
                                      n_rows, n_bands, 1 if use_cb else 0, criterion, out.ctypes.data, cap, st, threads)
            if cnt < 0:
                raise RuntimeError("orc_select failed")
            if cnt <= cap:
                return out[:cnt], {"evaluated": st[0], "survivors": st[1], "selected": cnt}
            cap = int(cnt)


class BuildOracle:
    """oracle/build_sketch_oracle.c: sequential restatement of src/build_sketch.cpp + SuperMinHash/HLL addh"""

    def __init__(self):
        self.lib = C.CDLL(os.environ.get("ORACLE_LIB", str(ROOT / "oracle" / "liboracle.so")))   # ORACLE_LIB: a sanitizer build (scripts/asan_host.sh)
        L = self.lib
        L.orcb_smh_new.restype = C.c_void_p
        L.orcb_smh_new.argtypes = [C.c_size_t]
        L.orcb_smh_data.restype = C.c_void_p
        L.orcb_smh_data.argtypes = [C.c_void_p]
        L.orcb_smh_size.restype = C.c_uint32
        L.orcb_smh_size.argtypes = [C.c_void_p]
        L.orcb_smh_free.argtypes = [C.c_void_p]
        L.orcb_sketch_file.restype = C.c_longlong
        L.orcb_sketch_file.argtypes = [C.c_char_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]

    def sketch(self, path, m=0, p_aux=0, k=31):
        """returns (hll14 u8[16384], aux u8[1<<p_aux] or None, smh u64[m'] or None, n_kmers)"""
        hll = np.zeros(16384, dtype=np.uint8)
        aux = np.zeros(1 << p_aux, dtype=np.uint8) if p_aux else None
        smh = self.lib.orcb_smh_new(m) if m else None
        n = self.lib.orcb_sketch_file(str(path).encode(), k, hll.ctypes.data, aux.ctypes.data if p_aux else None, p_aux, smh)
        if n < 0:
            raise RuntimeError(f"cannot read {path}")
        out = None
        if smh:
            cnt = self.lib.orcb_smh_size(smh)
            out = np.frombuffer((C.c_uint8 * (8 * cnt)).from_address(self.lib.orcb_smh_data(smh)), dtype=np.uint64).copy()
            self.lib.orcb_smh_free(smh)
        return hll, aux, out, n
