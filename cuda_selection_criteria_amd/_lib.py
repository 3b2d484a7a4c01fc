"""ctypes bindings of the two in-tree native libraries.

libselhip.so  (include/selection_hip.h)  -- gfx950 kernels + C ABI; the ONLY compute path.
libselhost.so (include/selection_host.h) -- host-side driver logic (sketch I/O, sort, banding, ...).

There is deliberately no Python/NumPy fallback for anything the HIP library does: if the library is
missing, or no MI355X is present, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_DIR = _PKG / "lib"


class SelhipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"selhip error {code}: {msg}")
        self.code = code


class Result(C.Structure):            # struct Result, src/selection_kernels_wrapper.hpp:6-9
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("sim", C.c_float)]


class Int2(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32)]


class Pair(C.Structure):
    _fields_ = [("i", C.c_int32), ("k", C.c_int32), ("jaccard", C.c_double)]


class Synth(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_genomes", C.c_int32), ("m", C.c_int32), ("p_aux", C.c_int32),
                ("cluster_size", C.c_int32), ("mode", C.c_int32), ("n_sh_lo", C.c_uint32), ("n_sh_hi", C.c_uint32)]


MODE_SMH, MODE_CB_SMH = 0, 1
ALGO_AUTO, ALGO_STREAM, ALGO_SIG, ALGO_HASHJOIN = 0, 1, 2, 3
FP_STRICT, FP_FMA = 0, 1
CRIT_SMH_A, CRIT_HLL_A, CRIT_HLL_AN, CRIT_HLL_A_SMH_A = 0, 1, 2, 3
BANDING_CPU, BANDING_CUDA = 0, 1

_vp, _i, _i64, _d, _sz, _cp = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t, C.c_char_p

# name -> (restype, argtypes): every symbol declared in include/selection_hip.h
HIP_SYMBOLS = {
    "launch_kernel_smh": (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _i, _i, _i, _vp, _vp, _i]),
    "launch_kernel_CBsmh": (_i, [_vp, _vp, _vp, _vp, _i, _d, _i, _i, _i, _i, _vp, _vp, _i]),
    "launch_kernel_smh64": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _i, _i, _i, _i, _vp, _vp, _i]),
    "launch_kernel_CBsmh64": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _i, _i, _i, _i, _vp, _vp, _i]),
    "selhip_device_count": (_i, []),
    "selhip_ctx_create": (_i, [C.POINTER(_vp), _i]),
    "selhip_ctx_destroy": (None, [_vp]),
    "selhip_last_error": (_cp, [_vp]),
    "selhip_ctx_set_stream": (_i, [_vp, _vp]),
    "selhip_ctx_set_fp_mode": (_i, [_vp, _i]),
    "selhip_ctx_set_pipeline": (_i, [_vp, _i]),
    "selhip_ctx_set_stage2_grouping": (_i, [_vp, _i]),
    "selhip_ctx_set_param": (_i, [_vp, _cp, _i]),
    "selhip_ctx_get_param": (_i, [_vp, _cp, _vp]),
    "selhip_ctx_set_row_interleave": (_i, [_vp, _i, _i, _i]),
    "selhip_ctx_upload": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i]),
    "selhip_ctx_attach": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _i]),
    "selhip_ctx_upload_aux_hll": (_i, [_vp, _vp, _i]),
    "selhip_ctx_attach_aux_hll": (_i, [_vp, _vp, _i]),
    "selhip_ctx_set_criterion": (_i, [_vp, _i]),
    "selhip_hll_cards": (_i, [_vp, _vp, _i64, _i, _vp]),
    "selhip_ctx_get_cards": (_i, [_vp, _vp]),
    "selhip_ctx_run": (_i, [_vp, _i, _i, C.c_float, _i, _i, _i64, _i64]),
    "selhip_ctx_run_async": (_i, [_vp, _i, _i, C.c_float, _i, _i, _i64, _i64]),
    "selhip_ctx_finish": (_i, [_vp]),
    "selhip_ctx_stats": (_i, [_vp, C.POINTER(_i64)]),
    "selhip_ctx_result_count": (_i64, [_vp]),
    "selhip_ctx_fetch": (_i, [_vp, _vp, _i64]),
    "selhip_ctx_result_device": (_i, [_vp, C.POINTER(_vp), C.POINTER(_i64)]),
    "selhip_ctx_copy_results": (_i, [_vp, _vp, _i64]),
    "selhip_ctx_copy_results_framed": (_i, [_vp, _vp, _i64]),
    "selhip_ctx_copy_results_framed_async": (_i, [_vp, _vp, _i64]),
    "selhip_ctx_last_attempts": (_i, [_vp]),
    "selhip_ctx_kernel_ms": (_d, [_vp, _cp]),
    "selhip_ctx_kernel_launches": (_d, [_vp, _cp]),
    "selhip_ctx_timing": (_i, [_vp, _i]),
    "selhip_multi_select": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _i, _i, _i, C.c_float, _i, _i, _i, _vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "selhip_ooc_select": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _i, _i, _i, C.c_float, _i, _i, _i64, _i,
                               _vp, _i64, C.POINTER(_i64), C.POINTER(_i64)]),
    "selhip_ctx_set_candidate_begin": (_i, [_vp, _i64]),
    "selhip_smh_a_pairs": (_i, [_vp, _i, _i, _i, _vp, _i64, _vp, _vp]),
    "selhip_hll_union_hist": (_i, [_vp, _i, _vp, _i64, _vp, _vp]),
    "selhip_hll_bitslice": (_i, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "selhip_hll_union_hist_planes": (_i, [_vp, _vp, _i, _vp, _i64, _vp, _vp]),
    "selhip_ertl_estimate": (_i, [_vp, _i64, _i, _i, _vp, _vp]),
    "selhip_smh_match_counts": (_i, [_vp, _i, _vp, _i64, _vp, _vp]),
    "selhip_synth_generate": (_i, [C.POINTER(Synth), _i64, _i64, _vp, _vp, _vp, _vp]),
    "selhip_permute_rows": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "selhip_build_sketches": (_i, [_vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "selhip_malloc": (_i, [C.POINTER(_vp), _sz]),
    "selhip_free": (_i, [_vp]),
    "selhip_memcpy_h2d": (_i, [_vp, _vp, _sz]),
    "selhip_memcpy_d2h": (_i, [_vp, _vp, _sz]),
    "selhip_device_synchronize": (_i, []),
    "selhip_version": (_cp, []),
}

HOST_SYMBOLS = {
    "selhost_last_error": (_cp, []),
    "selhost_read_hll": (_i, [_cp, _vp, _sz, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(_d)]),
    "selhost_write_hll": (_i, [_cp, _vp, C.c_uint32]),
    "selhost_read_smh": (_i64, [_cp, _vp, _sz]),
    "selhost_write_smh": (_i, [_cp, _vp, C.c_uint32]),
    "selhost_fasta_codes": (_i64, [_cp, _vp, _sz]),
    "selhost_smh_vecsize": (C.c_uint32, [C.c_uint32]),
    "selhost_hll_report": (_d, [_vp, C.c_uint, _i]),
    "selhost_hll_union_size": (_d, [_vp, _vp, C.c_uint, _i]),
    "selhost_ertl_estimate": (_d, [_vp, C.c_uint, _i]),
    "selhost_log1p": (_d, [_d]),
    "selhost_banding": (None, [C.c_uint, C.c_float, _i, C.POINTER(_i), C.POINTER(_i)]),
    "selhost_sort_by_card": (_i, [_vp, _i64, _vp]),
    "selhost_dataset_load": (_i, [C.POINTER(_vp), _cp, C.c_uint, C.c_uint, _i, _i]),
    "selhost_dataset_free": (None, [_vp]),
    "selhost_dataset_size": (_i64, [_vp]),
    "selhost_dataset_hll": (_vp, [_vp]),
    "selhost_dataset_aux": (_vp, [_vp]),
    "selhost_dataset_aux_hll": (_vp, [_vp]),
    "selhost_dataset_cards": (_vp, [_vp]),
    "selhost_dataset_name": (_cp, [_vp, _i64]),
    "selhost_format_line": (_i, [_cp, _cp, _d, _vp, _sz]),
    "selhost_write_results": (_i, [C.c_char_p, _vp, _i64, _vp, _i64, C.c_float]),
    "selhost_read_results": (_i, [_vp, C.c_char_p]),
    "selhost_results_free": (None, [_vp]),
    "selhost_results_count": (_i64, [_vp]),
    "selhost_results_names": (_i64, [_vp]),
    "selhost_results_tau": (C.c_float, [_vp]),
    "selhost_results_pairs": (_vp, [_vp]),
    "selhost_results_name": (C.c_char_p, [_vp, _i64]),
    "selhost_results_text": (_i64, [_vp, _vp, C.c_size_t]),
    "selhost_synth_generate": (_i, [C.POINTER(Synth), _i64, _i64, _vp, _vp, _vp, _i]),
    "selhost_shard_rows": (_i, [_i64, _vp, _i64, _i, _vp]),
    "selhost_version": (_cp, []),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return lib


_hip = None
_host = None


def hip_lib():
    """libselhip.so, loaded lazily.  Raises (never falls back) if it has not been built."""
    global _hip
    if _hip is None:
        path = Path(os.environ.get("SELHIP_LIB", LIB_DIR / "libselhip.so"))      # SELHIP_LIB: development A/B builds of the same library
        if not path.exists():
            raise ImportError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              f"or `make -C {_PKG / 'csrc'}` -- there is no CPU fallback")
        _hip = _bind(C.CDLL(str(path), mode=os.RTLD_GLOBAL if hasattr(os, "RTLD_GLOBAL") else 0), HIP_SYMBOLS)
    return _hip


def host_lib():
    global _host
    if _host is None:
        path = Path(os.environ.get("SELHOST_LIB", LIB_DIR / "libselhost.so"))    # SELHOST_LIB: e.g. a sanitizer build (scripts/asan_host.sh)
        if not path.exists():
            raise ImportError(f"{path} is missing: build it with `make -C {_PKG / 'csrc'}`")
        _host = _bind(C.CDLL(str(path)), HOST_SYMBOLS)
    return _host


def check(rc: int, ctx=None):
    if rc < 0:
        msg = hip_lib().selhip_last_error(ctx).decode(errors="replace")
        raise SelhipError(rc, msg)
    return rc
