"""MI355X-native all-pairs SuperMinHash/HLL sketch selection (drop-in for the hot path of
sanhue903/CUDA_Selection_Criteria: `selection` / `time_smh_cuda`).

Python here is plumbing only (device memory via torch, torch.distributed for the multi-GPU gather,
ctypes bindings); the product is csrc/ (HIP kernels + C ABI in include/selection_hip.h, C++ host code).
"""
from ._lib import (ALGO_AUTO, ALGO_HASHJOIN, ALGO_SIG, ALGO_STREAM, BANDING_CPU, BANDING_CUDA, CRIT_HLL_A, CRIT_HLL_A_SMH_A,  # noqa: F401
                   CRIT_HLL_AN, CRIT_SMH_A, FP_FMA, FP_STRICT, MODE_CB_SMH, MODE_SMH, SelhipError, hip_lib, host_lib)
from .selection import (PAIR_DTYPE, Selector, banding, format_lines, load_dataset, ooc_select, read_results,  # noqa: F401
                        select_from_filelist, sort_by_card, write_results)
from .synth import SYNTH_CONFIGS, SynthConfig, harden, stream_model, synth_device, synth_host  # noqa: F401

__all__ = ["Selector", "ooc_select", "write_results", "read_results", "banding", "select_from_filelist", "load_dataset", "sort_by_card", "format_lines",
           "SynthConfig", "SYNTH_CONFIGS", "synth_device", "synth_host", "harden", "stream_model", "hip_lib", "host_lib", "SelhipError",
           "MODE_SMH", "MODE_CB_SMH", "ALGO_AUTO", "ALGO_STREAM", "ALGO_SIG", "ALGO_HASHJOIN", "FP_FMA", "FP_STRICT", "PAIR_DTYPE"]
