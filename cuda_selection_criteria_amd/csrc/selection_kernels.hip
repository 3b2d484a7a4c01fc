// selection_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels and the C ABI of libselhip.so.
//
// Replaces, for the all-pairs sketch-selection path of sanhue903/CUDA_Selection_Criteria:
//   src/selection_kernels.cu:13-177 (kernel_smh, kernel_CBsmh, launchers)
//   include/criteria_sketch_cuda.cuh:11-65 (device CB / smh_a / hll_union_card)
// with the RESULT SEMANTICS of the CPU path src/selection.cpp:270-291 (see include/selection_hip.h).
//
// This is the kernel translation unit of the library: it only includes.  The kernels live in the kernel_*.cuh headers (all integer
// except the estimator; no MFMA), the host side in host_context.hpp (context), host_pass.hpp (pass scheduler) and abi_*.inc (C ABI);
// the multi-GPU and out-of-core drivers are translation units of their own (selhip_multi.hip, selhip_ooc.hip):
//   common.cuh          launch constants, per-pass counters, WaveAppender (LDS-staged appends, one atomic per flush)
//   kernel_bounds.cuh   cb_bounds_kernel      e_i = (size_t)card_i, CB cut-off hi(i), first non-zero rank
//   kernel_stream.cuh   smh_stream_kernel     stage 1 ALGO_STREAM: query tile in LDS/VGPRs, candidates streamed row-major,
//                                             v_cmp_eq_u64 lane masks folded on the scalar unit; smh_generic_kernel
//   kernel_sigjoin.cuh  sig_build / sig_join / verify   stage 1 ALGO_SIG: all-pairs band-signature join (DPP broadcast) + exact verify
//   kernel_hll.cuh      hll_union_hist_kernel, ertl_select_kernel (stage 2), enum_pairs / aux_fused (hll_a, hll_an)
//   kernel_hllbs.cuh    hll_bitslice_kernel, hll_union_hist_bs_kernel: stage 2a on bit-sliced registers (bit-serial max, decode tree, v_bcnt)
//   kernel_pairlist.cuh explicit pair lists (drop-in launch_kernel_* path, test building blocks)
//   kernel_sketch.cuh   synth_kernel, sketch_build_kernel (build_sketch on the GPU), permute_rows
//   kernel_small.cuh    small_pass_kernel: the whole pass of a set of <= 2 048 genomes in one cooperative launch
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>         // exclusive scan of the per-row survivor counts (grouping for stage 2)
#include <rocprim/device/device_radix_sort.hpp>   // ALGO_HASHJOIN only: the key sort is a library call, everything else is hand-written

#include <algorithm>
#include <type_traits>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/selection_hip.h"
#include "selhip_internal.h"
#include "ertl_mle.hpp"
#include "synth.hpp"

#include "common.cuh"
#include "kernel_bounds.cuh"
#include "kernel_stream.cuh"
#include "kernel_sigjoin.cuh"
#include "kernel_hll.cuh"
#include "kernel_hllbs.cuh"
#include "kernel_pairlist.cuh"
#include "kernel_sketch.cuh"
#include "kernel_small.cuh"

#include "host_context.hpp"      // struct selhip_ctx, device buffers, timers, helpers
#include "host_pass.hpp"         // pass scheduler: dispatch of every stage, chunk lanes, scratch sizing
#include "abi_context.inc"       // C ABI: context (create, upload / attach, run, results, timing)
#include "abi_blocks.inc"        // C ABI: building blocks, synthetic sketches, sketch construction, memory helpers
#include "abi_compat.inc"        // C ABI: drop-in launch_kernel_smh / launch_kernel_CBsmh
