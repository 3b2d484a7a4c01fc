// selection_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels and the C ABI of libselhip.so.
//
// Replaces, for the all-pairs sketch-selection path of sanhue903/CUDA_Selection_Criteria:
//   src/selection_kernels.cu:13-177 (kernel_smh, kernel_CBsmh, launchers)
//   include/criteria_sketch_cuda.cuh:11-65 (device CB / smh_a / hll_union_card)
// with the RESULT SEMANTICS of the CPU path src/selection.cpp:270-291 (see include/selection_hip.h).
//
// This file holds the HOST side of the library (context, dispatch, C ABI).  The kernels live in the headers it
// includes (one translation unit; all integer except the estimator; no MFMA):
//   common.cuh          launch constants, per-pass counters, WaveAppender (LDS-staged appends, one atomic per flush)
//   kernel_bounds.cuh   cb_bounds_kernel      e_i = (size_t)card_i, CB cut-off hi(i), first non-zero rank
//   kernel_stream.cuh   smh_stream_kernel     stage 1 ALGO_STREAM: query tile in LDS/VGPRs, candidates streamed row-major,
//                                             v_cmp_eq_u64 lane masks folded on the scalar unit; smh_generic_kernel
//   kernel_sigjoin.cuh  sig_build / sig_join / verify   stage 1 ALGO_SIG: all-pairs band-signature join (DPP broadcast) + exact verify
//   kernel_hll.cuh      hll_union_hist_kernel, ertl_select_kernel (stage 2), enum_pairs / aux_fused (hll_a, hll_an)
//   kernel_hllbs.cuh    hll_bitslice_kernel, hll_union_hist_bs_kernel: stage 2a on bit-sliced registers (bit-serial max, decode tree, v_bcnt)
//   kernel_pairlist.cuh explicit pair lists (drop-in launch_kernel_* path, test building blocks)
//   kernel_sketch.cuh   synth_kernel, sketch_build_kernel (build_sketch on the GPU), permute_rows
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

namespace {

// =============================================================================================
// host side
// =============================================================================================
thread_local std::string g_last_error = "";

void set_err(std::string* dst, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (dst) *dst = buf;
    g_last_error = buf;
}

#define HIPCHK(ctx_err, expr)                                                                  \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            set_err(ctx_err, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return SELHIP_E_HIP;                                                               \
        }                                                                                      \
    } while (0)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct KernelTimer {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<long> ev_pass;          // the pass each pair was recorded in
    double total_ms = 0;
    double span_ms = 0;                 // per pass: first start -> last end (chunk lanes run a kernel's launches side by side)
    long launches = 0;
};

constexpr int kMaxChunks = 8;
constexpr size_t kSegCounterSlots = (size_t)(kMaxChunks + 1) * kAppendSegs * kSegStride;    // the join's append-segment counters (u64 slots)
constexpr double kAutoChunkPairs = 1e9;  // pairs per pass from which the automatic setting splits a pass into two chunk lanes
static_assert(kCounterBlocks == kMaxChunks + 1, "common.cuh: counter blocks per pass");
constexpr int kMaxAuxP = SELHIP_MAX_AUX_P;   // auxiliary HLL precision accepted by every entry point: aux_fused_kernel counts in 16-bit bins (a bin holds up to 2^p_aux)
constexpr long long kEnumPairs = 1ll << 26;    // hll_a / hll_an as first criterion: pairs listed per sub-pass (512 MiB of int2)

enum { T_PREP = 0, T_STAGE1, T_HIST, T_SELECT, T_TOTAL, T_SIGBUILD, T_JOIN, T_VERIFY, T_AUX, T_GROUP, T_COUNT };
const char* kTimerNames[T_COUNT] = {"prep", "stage1", "hist", "select", "total", "sigbuild", "join", "verify", "aux", "group"};

}  // namespace

struct selhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int fp_mode = SELHIP_FP_FMA;
    std::string err;

    // sketches (owned or attached)
    bool owns_sketches = false;
    const uint8_t* d_hll = nullptr;
    const u64* d_aux = nullptr;
    const double* d_cards = nullptr;
    DevBuf<uint8_t> own_hll;
    DevBuf<u64> own_aux;
    DevBuf<double> own_cards;
    int64_t n = 0;
    int m = 0, p = 14;

    // derived / scratch
    DevBuf<u64> ecard;
    DevBuf<int> hi;
    DevBuf<PassCounters> pc;            // TWO sets of kMaxChunks + 1 counter blocks: pass k uses set k & 1 and its first kernel clears the other
    PassCounters* pcb = nullptr;        // the set of the pass enqueued last
    int pc_flip = 0;
    bool pc_dirty = false;              // a pass claimed a counter set and did not get to the end of its enqueue (or the stream changed): clear both sets first
    int fail_after_flip = 0;            // test hook ("fail_after_flip"): the next enqueue returns an error right after claiming its counter set
    DevBuf<u64> seg_cnt;                // the join's append-segment counters: (kMaxChunks + 1) x kAppendSegs x kSegStride
    DevBuf<selhip_int2_t> surv;
    DevBuf<uint32_t> counts;
    DevBuf<selhip_pair_t> results;
    DevBuf<selhip_int2_t> self_pairs;
    DevBuf<selhip_int2_t> cand;         // ALGO_SIG: signature-join candidates; aux criteria: enumerated pairs
    DevBuf<selhip_int2_t> fin;          // aux criteria: pairs that passed hll_a / hll_an
    const uint8_t* d_aux_hll = nullptr; // auxiliary HLL registers [n][1 << p_aux]
    DevBuf<uint8_t> own_aux_hll;
    int p_aux = 0;
    int criterion = 0;
    DevBuf<u64> aux_il;                 // ALGO_STREAM: bucket-interleaved copy of the sketches (kernel_stream.cuh)
    DevBuf<uint32_t> sigQ, sigT, sigP, sigG;  // ALGO_SIG: band signatures, genome-major / band-major / band-major 16-bit pairs / genome-major 16-bit pairs
    DevBuf<u64> hj_keys_in, hj_keys_out;   // ALGO_HASHJOIN: (band << 32 | signature) keys, before / after the sort
    DevBuf<int> hj_vals_in, hj_vals_out;   //                genome ranks carried by the keys
    DevBuf<char> hj_tmp;                   //                rocPRIM temporary storage
    DevBuf<int> csr_cnt, csr_start;        // stage 2 grouping: survivors per query row (cnt[0..n) counts, cnt[n..2n) fill cursors), offsets
    DevBuf<selhip_int2_t> grouped;         //                   the final pair list bucketed by query row
    DevBuf<char> scan_tmp;
    size_t scan_tmp_stride = 0;         // bytes of rocPRIM scan scratch per chunk
    PassCounters* h_pc = nullptr;       // pinned host mirror of the kMaxChunks + 1 counter blocks
    // stage pipeline: stage 1 of row chunk c+1 (VALU-bound) overlaps stage 2 of chunk c (memory/LDS-bound)
    hipStream_t st_stage1 = nullptr;    // internal non-blocking stream: the second chunk lane (the first is `stream`)
    hipEvent_t ev_start = nullptr, ev_end = nullptr;       // fork / join of the second lane
    int n_chunks_last = 1;
    int pipeline = -1;                  // -1 auto, 0 off, >0 forced chunk count
    int64_t cand_begin = 0;             // candidates restricted to ranks >= cand_begin (selhip_ctx_set_candidate_begin)
    int il_block = 128, il_parts = 1, il_part = 0;    // row interleave (selhip_ctx_set_row_interleave); il_parts 1 = contiguous
    int hist_pad = 0;                   // stage 2a: extra LDS bytes per one-wave block (lowers the number of resident waves per CU)
    int hist_run = 0, hist_blocks = kHistSpanBlocks;   // stage 2a: pairs per task (0 = automatic: 1, or 4 with the label order), one-wave blocks (multiple of 8)
    // stage 2a on bit planes (kernel_hllbs.cuh): the p = 14 registers of every genome as 6 bit planes, written when the sketches
    // are uploaded / attached (selhip_ctx_upload / _attach; the caller's arrays must not change behind an attached context)
    DevBuf<uint32_t> hll_bs;            // [n][6][512]
    DevBuf<uint8_t> hll_gmax;           // [n] largest register value of each genome
    DevBuf<int> hll_bs_max;             // largest register value of the set (device side)
    int hll_khi = 0;                    // 0 = no planes; else max register value + 1
    int hist_algo = -1;                 // -1 automatic (bit planes when p = 14), 0 = byte rows + LDS histogram (hll_union_hist_runs_kernel), 1 = bit planes
    int hist_bs_blocks = 2048;          // bit-plane kernel: 4-wave blocks (multiple of 8)
    int group_label = -1;               // grouping: lay the query-row buckets out by label (kernel_hll.cuh): -1 = automatic (HLL rows beyond kLabelOrderBytes), 0 off, 1 on
    int verify_fb = 0;                  // test hook: force the collision fallback of verify16_kernel
    int join_wpb = 4;                   // 16-bit join: waves per block (DPP form: 1 or 4; LDS form: 4 or 8 -- the waves of a block share the staged query tile)
    int join_db = 1;                    // 16-bit join: double-buffered query batches
    int join_tri = 0;                   // LDS-tile join: 1 = launch only the (tile, candidate block) units above the diagonal (measured: no gain, see JoinTriangle); 0 = the rectangle
    int join_form = 0;                  // 16-bit LDS-tile join, inner loop: 0 = xor + v_pk_min_u16, 1 = zero-half test (xor, sub, v_bitop3_b32; measured slower, see kernel_sigjoin.cuh)
    int join_bits = 16;                 // signature width of the all-pairs join: 16 (packed min), 15 (LDS form only: flag arithmetic, all plain VOP2) or 32
    int join_q = 1;                     // 16-bit join, query side: 1 = tile staged in LDS, broadcast reads (sigl_join_kernel), 0 = DPP row broadcast (sig16_join_kernel)
    long long enum_pairs = kEnumPairs;  // hll_a / hll_an as first criterion: pairs listed per sub-pass (test hook "enum_pairs")
    int sig_cache = 0;                  // keep the band signatures across passes ("sig_cache"); sig_key = what the arrays hold (0 = nothing)
    long long sig_key = 0;
    int sig_tile = 1;                   // signature build: tiled form (0 = one thread per bucket, the round-1 kernel)
    int init_cap = 0;                   // test hook: initial capacity of the survivor / candidate lists (0 = sized from the workload)
    int join_qt = 0;                    // query rows per signature-join block (multiple of 16); 0 = automatic: 32 rows below 1e8 pairs per pass, 64 up to 4.5e8
                                        // (30 000 genomes on one GPU; 2e8 since round 3), 128 beyond.  With the segmented appends: cfg3 112 / 114 / 127 us at 64 / 96 / 128 rows (finer tiles balance
                                        // the 1 024 SIMDs better), cfg4 2.12 / 2.10 / 2.07 ms (a block's prologue -- 32 candidate loads per lane,
                                        // tile staging -- is amortised over more rows), cfg5 8.31 / 8.16 / 8.20 ms
    bool group_stage2 = true;           // bucket survivors by query row before stage 2a (hll_union_hist_runs_kernel)

    // last run parameters (for overflow re-runs)
    bool have_run = false, pending = false;
    int mode = 0, algo = 0, n_rows = 0, n_bands = 0;
    float tau_f = 0;
    int64_t row_begin = 0, row_end = 0;
    PassCounters last{};

    int timing = 0;                     // 0 off, 1 every kernel scope, 2 dominant stage-1 kernel only
    int dominant_timer = T_STAGE1;
    int timed_kernel = 0;               // timing level 2 keeps the events of: 0 = the stage-1 kernel (join / stream), 1 = stage 2a ("timed_kernel")
    long timed_passes = 0;
    int last_attempts = 0;              // enqueues the last finished run needed (1 = nothing overflowed)
    KernelTimer timers[T_COUNT];
};

namespace {

int check_device(std::string* err) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) {
        set_err(err, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return SELHIP_E_NODEVICE;
    }
    return SELHIP_OK;
}

// timing level 1: every scope; level 2: only the dominant stage-1 kernel (an event pair costs ~10 us of stream time, and a
// pass of the default workload is ~0.4 ms)
struct TimerScope {
    selhip_ctx* c; int id; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on;
    TimerScope(selhip_ctx* c_, int id_) : TimerScope(c_, id_, c_->stream) {}
    TimerScope(selhip_ctx* c_, int id_, hipStream_t st_) : c(c_), id(id_), st(st_) {
        on = c->timing == 1 || (c->timing == 2 && id == c->dominant_timer);
        if (on) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, st);
        }
    }
    ~TimerScope() {
        if (on) {
            (void)hipEventRecord(b, st);
            c->timers[id].ev.emplace_back(a, b);
            c->timers[id].ev_pass.push_back(c->timed_passes);
        }
    }
};

// where one stage-1 launch (a chunk of query rows) writes: its stream, its slice of the candidate / survivor lists
// and its own counter block; pc0 (the pass's block 0) carries what every chunk reads (z0) and the result counter
struct StageIO {
    hipStream_t st;
    selhip_int2_t* cand;
    selhip_int2_t* surv;
    u64 cap;
    PassCounters* pc;
    int* row_cnt = nullptr;     // if set, the producer of `surv` also tallies survivors per query row (stage-2 grouping)
    int* row_lab = nullptr;     // ... and every row's smallest partner (label order of the grouping)
    u64* seg_cnt = nullptr;     // 16-bit join: this launch's kAppendSegs append counters
};

void drain_timers(selhip_ctx* c) {
    for (int t = 0; t < T_COUNT; ++t) {
        KernelTimer& kt = c->timers[t];
        for (size_t j = 0; j < kt.ev.size();) {
            // the launches of one pass: sum of their durations, and the span they cover together
            size_t e = j;
            float lo = 0, hi = 0;
            for (; e < kt.ev.size() && kt.ev_pass[e] == kt.ev_pass[j]; ++e) {
                float ms = 0, a_off = 0, b_off = 0;
                if (hipEventSynchronize(kt.ev[e].second) != hipSuccess) continue;
                if (hipEventElapsedTime(&ms, kt.ev[e].first, kt.ev[e].second) == hipSuccess) { kt.total_ms += ms; kt.launches += 1; }
                if (hipEventElapsedTime(&a_off, kt.ev[j].first, kt.ev[e].first) == hipSuccess &&
                    hipEventElapsedTime(&b_off, kt.ev[j].first, kt.ev[e].second) == hipSuccess) { lo = std::min(lo, a_off); hi = std::max(hi, b_off); }
            }
            kt.span_ms += hi - lo;
            j = e;
        }
        for (auto& pr : kt.ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
        kt.ev.clear(); kt.ev_pass.clear();
    }
}

double relerr_scaled_for(int p) {
    // hll.h:662  relerr /= std::sqrt(m), relerr = 1e-2 (hll.h:211 default, :257)
    return 1e-2 / std::sqrt((double)(1ull << p));
}

bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// RowMap of the query rows [rb, re) under the context's interleave setting (selhip_ctx_set_row_interleave)
RowMap row_map(const selhip_ctx* c, int rb, int re) {
    RowMap rm;
    rm.row_begin = rb; rm.row_end = re;
    if (c->il_parts > 1) { rm.block_rows = c->il_block; rm.n_parts = c->il_parts; rm.part = c->il_part; }
    else                 { rm.block_rows = std::max(1, re - rb); rm.n_parts = 1; rm.part = 0; }
    return rm;
}

// ---- stage-1 dispatch ------------------------------------------------------------------------
template <int NCH, int LOG2R>
hipError_t launch_stream(selhip_ctx* c, const StageIO& io, const RowMap& rm) {
    constexpr int Q = kQueryVgprBudget / NCH;
    const int n = (int)c->n;
    const long long n_tiles_ll = rm.n_tiles(Q);
    if (n_tiles_ll > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int n_tiles = (int)n_tiles_ll;
    // candidate columns that can matter: k in (row_begin, n)
    const int chunk_base = ((rm.row_begin + 1) / kChunk) * kChunk;
    const int n_chunks = (n - chunk_base + kChunk - 1) / kChunk;
    if (n_tiles <= 0 || n_chunks <= 0) return hipSuccess;
    const long long blocks = (long long)n_tiles * n_chunks;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((smh_stream_kernel<NCH, LOG2R>), dim3((unsigned)blocks), dim3(kBlock), 0, io.st,
                       reinterpret_cast<const u64x2*>(c->aux_il.p), n, c->hi.p, c->pcb,
                       rm, n_tiles, chunk_base, io.surv, io.cap, io.pc);
    return hipGetLastError();
}

// LOG2R runs over 0 .. log2(m) = log2(128 * NCH)
template <int NCH, int LOG2R>
hipError_t launch_stream_r(selhip_ctx* c, const StageIO& io, int l, const RowMap& rm) {
    if (l == LOG2R) return launch_stream<NCH, LOG2R>(c, io, rm);
    if constexpr ((1 << LOG2R) < 128 * NCH) return launch_stream_r<NCH, LOG2R + 1>(c, io, l, rm);
    return hipErrorInvalidValue;
}

bool stream_supported(int m, int n_rows) {
    return is_pow2(m) && m >= 128 && m <= 2048 && is_pow2(n_rows) && n_rows <= m;
}

hipError_t launch_stage1(selhip_ctx* c, const StageIO& io, int n_rows, int n_bands, const RowMap& rm) {
    if (stream_supported(c->m, n_rows)) {
        const int nch = c->m / 128;
        const int l = ilog2(n_rows);
        switch (nch) {
            case 1: return launch_stream_r<1, 0>(c, io, l, rm);
            case 2: return launch_stream_r<2, 0>(c, io, l, rm);
            case 4: return launch_stream_r<4, 0>(c, io, l, rm);
            case 8: return launch_stream_r<8, 0>(c, io, l, rm);
            case 16: return launch_stream_r<16, 0>(c, io, l, rm);
        }
    }
    const long long rows = rm.n_tiles(1);
    const int n = (int)c->n;
    const int chunks = (n + kBlock - 1) / kBlock;
    const long long blocks = rows * chunks;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(smh_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, io.st,
                       c->d_aux, n, c->m, n_rows, n_bands, c->hi.p, c->pcb, rm, (int)rows,
                       io.surv, io.cap, io.pc);
    return hipGetLastError();
}


unsigned grid_for(u64 items, unsigned per_block, unsigned max_blocks);

// tile height of the signature joins: the configured one, or the automatic choice (see selhip_ctx::join_qt)
int join_tile_rows(const selhip_ctx* c) {
    const double pairs_here = 0.5 * (double)c->n * (double)c->n / std::max(1, c->il_parts);       // this context's share of the triangle
    // (< 1e8 pairs: 32-row tiles -- twice the work units for the 8 192 wave slots, a shorter tail: cfg3's join 108.5 -> 104.3 us)
    int qt = c->join_qt > 0 ? c->join_qt : (pairs_here >= 4.5e8 ? 128 : pairs_here >= 2e8 ? 64 : 32);      // (one of 8 ranks of cfg4, 1.6e8 pairs: 32 rows 0.450 ms, 64 rows 0.481)
    if (c->il_parts > 1) { qt = std::min(qt, c->il_block); while (c->il_block % qt) qt -= 16; }
    return qt;
}

bool sig_supported(int m, int n_rows, int n_bands) {
    (void)m;
    return is_pow2(n_rows) && (n_bands == 8 || n_bands == 16 || n_bands == 32 || n_bands == 64 || n_bands == 128);
}

template <int NB>
hipError_t launch_join(selhip_ctx* c, const StageIO& io, int n_pad, const RowMap& rm) {
    const int n = (int)c->n;
    const int qt = join_tile_rows(c);   // query rows per block (multiple of 16)
    const long long n_tiles_ll = rm.n_tiles(qt);
    if (n_tiles_ll > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int n_tiles = (int)n_tiles_ll;
    const int group_base = (std::max(rm.row_begin + 1, (int)c->cand_begin) / kWave / kWavesPerBlock) * kWavesPerBlock;   // candidates k > row_begin, k >= cand_begin
    const int n_groups = (n + kWave - 1) / kWave - group_base;
    const int n_gblocks = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (n_tiles <= 0 || n_gblocks <= 0) return hipSuccess;
    const long long blocks = (long long)n_tiles * n_gblocks;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((sig_join_kernel<NB>), dim3((unsigned)blocks), dim3(kBlock), 0, io.st,
                       c->sigT.p, n, n_pad, c->hi.p, c->pcb, rm, n_tiles, group_base, qt,
                       io.cand, io.cap, io.pc);
    return hipGetLastError();
}

template <int ND, bool DB, int WPB>
hipError_t launch_join16_w(selhip_ctx* c, const StageIO& io, int n_pad, const RowMap& rm) {
    const int n = (int)c->n;
    if ((long long)ND * n_pad * 4 >= (1ll << 31)) return hipErrorInvalidValue;          // 32-bit offsets into the band-major signature array
    const int qt = join_tile_rows(c);
    const long long n_tiles_ll = rm.n_tiles(qt);
    if (n_tiles_ll > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int n_tiles = (int)n_tiles_ll;
    const int group_base = (std::max(rm.row_begin + 1, (int)c->cand_begin) / kWave / WPB) * WPB;   // candidates k > row_begin, k >= cand_begin
    const int n_groups = (n + kWave - 1) / kWave - group_base;
    const int n_gblocks = (n_groups + WPB - 1) / WPB;
    if (n_tiles <= 0 || n_gblocks <= 0) return hipSuccess;
    const long long blocks = (long long)n_tiles * n_gblocks;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((sig16_join_kernel<ND, DB, WPB>), dim3((unsigned)blocks), dim3(WPB * kWave), 0, io.st,
                       c->sigP.p, n, n_pad, c->hi.p, c->pcb, rm, n_tiles, group_base, qt,
                       io.cand, io.cap, io.seg_cnt);
    return hipGetLastError();
}

template <int ND, int T, int WPB>
hipError_t launch_joinl_w(selhip_ctx* c, const StageIO& io, int n_pad, const RowMap& rm) {
    const int n = (int)c->n;
    // tile height: the configured one, capped so that the tile (+ appenders) fits 64 KiB of LDS; a multiple of 16 that divides the
    // interleave block when rows are interleaved
    int qt = std::min(join_tile_rows(c), (int)((64 * 1024 - WPB * kAppendCap * sizeof(selhip_int2_t)) / (ND * 4 + 4) - kJoinTilePadRows) / 16 * 16);
    if (c->il_parts > 1) while (c->il_block % qt) qt -= 16;
    const long long n_tiles_ll = rm.n_tiles(qt);
    if (n_tiles_ll > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const int n_tiles = (int)n_tiles_ll;
    if ((long long)ND * n_pad * 4 >= (1ll << 31)) return hipErrorInvalidValue;          // 32-bit offsets into the band-major signature array
    constexpr int GPB = WPB * T;                                                          // candidate groups per block
    const int group_base = (std::max(rm.row_begin + 1, (int)c->cand_begin) / kWave / GPB) * GPB;   // candidates k > row_begin, k >= cand_begin
    const int n_groups = (n + kWave - 1) / kWave - group_base;
    const int n_gblocks = (n_groups + GPB - 1) / GPB;
    if (n_tiles <= 0 || n_gblocks <= 0) return hipSuccess;
    long long blocks = (long long)n_tiles * n_gblocks;
    // only the units above the diagonal (JoinTriangle, kernel_sigjoin.cuh) when the rows are contiguous and the tiles line up with the
    // 256-candidate blocks; otherwise the rectangle, whose blocks under the diagonal leave at once
    JoinTriangle tri{0, 0, 0, 0};
    constexpr int kCand = GPB * kWave;
    if (c->join_tri && rm.n_parts == 1 && kCand % qt == 0 && rm.row_begin % qt == 0 && blocks < 0x7FFFFFFFll) {
        const int a = kCand / qt, g_lo = group_base / GPB, rbq = rm.row_begin / qt;
        const long long c0 = (long long)a * (g_lo + 1) - rbq;
        if (c0 >= 1) {
            // columns k = 0 .. K-1 hold c0 + a k < n_tiles units
            long long K = c0 >= n_tiles ? 0 : ((long long)n_tiles - c0 + a - 1) / a;
            K = std::min<long long>(K, n_gblocks);
            const long long SK = K * c0 + (long long)a * K * (K - 1) / 2;
            const long long total = SK + (long long)(n_gblocks - K) * n_tiles;
            if (total > 0 && total < 0x7FFFFFFFll) { tri = JoinTriangle{a, (int)c0, (int)K, (int)SK}; blocks = total; }
        }
    }
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    const size_t smem = (size_t)WPB * kAppendCap * sizeof(selhip_int2_t) + (size_t)((qt + 3) & ~3) * 4 + (size_t)(qt + kJoinTilePadRows) * ND * 4;
    if (smem > 64 * 1024) return hipErrorInvalidValue;                                   // join_qt is capped so that this cannot happen
#define SELHIP_JOINL_LAUNCH(FORM) hipLaunchKernelGGL((sigl_join_kernel<ND, T, WPB, FORM>), dim3((unsigned)blocks), dim3(WPB * kWave), smem, io.st, \
                           c->sigP.p, c->sigG.p, n, n_pad, c->hi.p, c->pcb, rm, n_tiles, group_base, qt, \
                           io.cand, io.cap, io.seg_cnt, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, tri)
    if (c->join_bits == 15)    SELHIP_JOINL_LAUNCH(1);
    else if (c->join_form == 0) SELHIP_JOINL_LAUNCH(0);
    else                        SELHIP_JOINL_LAUNCH(2);
#undef SELHIP_JOINL_LAUNCH
    return hipGetLastError();
}

// (T = 2 groups of candidates per wave -- half the LDS reads -- was measured twice: 130 VGPRs, 3 waves per SIMD, cfg3 157 vs 127 us,
// cfg4 2.37 vs 2.07 ms; and, after the wait counts left the row loop, capped at 128 VGPRs / 4 waves per SIMD: cfg3 121 vs 101 us, cfg4
// 2.28 vs 2.02 ms -- the join wants waves, not fewer LDS reads; the template keeps the parameter, only T = 1 is instantiated)
template <int ND>
hipError_t launch_joinl(selhip_ctx* c, const StageIO& io, int n_pad, const RowMap& rm) {
    return c->join_wpb == 8 ? launch_joinl_w<ND, 1, 8>(c, io, n_pad, rm) : launch_joinl_w<ND, 1, 4>(c, io, n_pad, rm);
}

template <int ND, bool DB>
hipError_t launch_join16(selhip_ctx* c, const StageIO& io, int n_pad, const RowMap& rm) {
    if (c->join_q) return launch_joinl<ND>(c, io, n_pad, rm);
    return c->join_wpb == 1 ? launch_join16_w<ND, DB, 1>(c, io, n_pad, rm) : launch_join16_w<ND, DB, 4>(c, io, n_pad, rm);
}

// sig_build with the pass's bounds computation riding in its first blocks (with_bounds) or alone
hipError_t launch_sig_build(selhip_ctx* c, int n_rows, int n_bands, bool with_bounds, double tau, int rb, int re, PassCounters* zero_pc) {
    const int n = (int)c->n;
    const int n_pad = ((n + kWave - 1) / kWave) * kWave;
    TimerScope t(c, T_SIGBUILD);
    const int bounds_blocks = with_bounds ? (n + kBlock - 1) / kBlock : 0;
    // tiled build (kSigTileG genomes per block, LDS transpose) for the shapes of the all-pairs joins; the per-bucket form otherwise
    const bool tile_mode = is_pow2(c->m) && is_pow2(n_bands) && n_bands <= 128 && n_rows >= 2 && n_rows <= 32 && c->m >= 4 && c->sig_tile;
    const long long threads = n_rows <= kWave ? (long long)n * c->m : (long long)n * n_bands;
    // "sig_cache": the signatures depend on the sketches and the band shape only, so a context that runs many passes over the same
    // sketches (the ranks of a strong-scaled job, a threshold sweep) builds them once; upload / attach and any reallocation of the
    // signature arrays invalidate them.  The bounds blocks still run every pass (they depend on tau, the mode and the rows).
    const long long sig_key = ((long long)n_rows << 40) | ((long long)n_bands << 20) | ((long long)(c->join_bits == 15) << 2) | (tile_mode ? 2 : 0) | 1;
    const bool cached = c->sig_cache && c->sig_key == sig_key && with_bounds;
    const unsigned work_blocks = cached ? 0u : tile_mode ? (unsigned)((n + kSigTileG - 1) / kSigTileG) : (unsigned)((threads + kBlock - 1) / kBlock);
    c->sig_key = c->sig_cache ? sig_key : 0;
    if (work_blocks + (unsigned)bounds_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(sig_build_kernel, dim3(work_blocks + (unsigned)bounds_blocks), dim3(kBlock), 0, c->stream,
                       c->d_aux, n, c->m, n_rows, n_bands, n_pad, c->sigQ.p, c->sigT.p, c->sigP.p, c->sigG.p,
                       bounds_blocks, c->d_cards, tau, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, row_map(c, rb, re), c->ecard.p, c->hi.p, c->pcb,
                       (c->p == 14 && c->group_stage2) ? c->csr_cnt.p : nullptr, (c->p == 14 && c->group_stage2) ? (int)c->csr_cnt.cap : 0, (int)c->cand_begin,
                       with_bounds ? c->seg_cnt.p : nullptr, with_bounds ? (int)c->seg_cnt.cap : 0, c->join_bits == 15 ? 17 : 16,
                       zero_pc, tile_mode ? 1 : 0);
    return hipGetLastError();
}

// signature join + exact verification of the query rows [rb, re) (sig_build must have run)
hipError_t launch_stage1_sig(selhip_ctx* c, const StageIO& io, int n_rows, int n_bands, const RowMap& rm) {
    const int n = (int)c->n;
    const int n_pad = ((n + kWave - 1) / kWave) * kWave;
    hipError_t e = hipSuccess;
    if (c->join_bits == 16 || c->join_bits == 15) {
        {
            TimerScope t(c, T_JOIN, io.st);
            switch (n_bands) {
                case 8: e = c->join_db ? launch_join16<4, true>(c, io, n_pad, rm) : launch_join16<4, false>(c, io, n_pad, rm); break;
                case 16: e = c->join_db ? launch_join16<8, true>(c, io, n_pad, rm) : launch_join16<8, false>(c, io, n_pad, rm); break;
                case 32: e = c->join_db ? launch_join16<16, true>(c, io, n_pad, rm) : launch_join16<16, false>(c, io, n_pad, rm); break;
                case 64: e = c->join_db ? launch_join16<32, true>(c, io, n_pad, rm) : launch_join16<32, false>(c, io, n_pad, rm); break;
                case 128: e = c->join_db ? launch_join16<64, true>(c, io, n_pad, rm) : launch_join16<64, false>(c, io, n_pad, rm); break;
                default: return hipErrorInvalidValue;
            }
        }
        if (e != hipSuccess) return e;
        // the 16-bit matches were staged in the candidate list; survivors go to the survivor list as usual
        TimerScope t(c, T_VERIFY, io.st);
        static_assert(1024 % kAppendSegs == 0, "verify16_kernel: the grid is a multiple of the segment count");
        hipLaunchKernelGGL(verify16_kernel, dim3(1024), dim3(kVerifyBlock), 0, io.st, c->d_aux, c->m, n_rows, n_bands, c->sigQ.p,
                           io.cand, io.seg_cnt, io.cap, io.surv, io.cap, io.pc, c->verify_fb, io.row_cnt, io.row_lab, n);
        return hipGetLastError();
    } else {
        TimerScope t(c, T_JOIN, io.st);
        switch (n_bands) {
            case 8: e = launch_join<8>(c, io, n_pad, rm); break;
            case 16: e = launch_join<16>(c, io, n_pad, rm); break;
            case 32: e = launch_join<32>(c, io, n_pad, rm); break;
            case 64: e = launch_join<64>(c, io, n_pad, rm); break;
            case 128: e = launch_join<128>(c, io, n_pad, rm); break;
            default: return hipErrorInvalidValue;
        }
    }
    if (e != hipSuccess) return e;
    TimerScope t(c, T_VERIFY, io.st);
    hipLaunchKernelGGL(verify_kernel, dim3(1024), dim3(kBlock), 0, io.st, c->d_aux, c->m, n_rows, n_bands,
                       io.cand, &io.pc->n_candidates, io.cap, io.surv, io.cap, io.pc);
    return hipGetLastError();
}

// sort-based join of the band signatures (sig_build must have run); rows [rb, re)
hipError_t launch_stage1_hashjoin(selhip_ctx* c, const StageIO& io, int n_rows, int n_bands, const RowMap& rm) {
    const int n = (int)c->n;
    const int n_pad = ((n + kWave - 1) / kWave) * kWave;
    const long long total = (long long)n * n_bands;
    if (total <= 0) return hipSuccess;
    TimerScope t(c, T_JOIN, io.st);
    hipLaunchKernelGGL(sigkey_build_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, io.st,
                       c->sigT.p, n, n_pad, n_bands, c->hj_keys_in.p, c->hj_vals_in.p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t tmp_bytes = c->hj_tmp.cap;
    const unsigned end_bit = 32u + (unsigned)ilog2(n_bands) + 1u;
    e = rocprim::radix_sort_pairs(c->hj_tmp.p, tmp_bytes, c->hj_keys_in.p, c->hj_keys_out.p, c->hj_vals_in.p, c->hj_vals_out.p,
                                  (size_t)total, 0u, end_bit, io.st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(run_emit_kernel, dim3(grid_for((u64)total, kBlock, 8192)), dim3(kBlock), 0, io.st,
                       c->hj_keys_out.p, c->hj_vals_out.p, total, c->sigQ.p, n_bands, c->d_aux, c->m, n_rows, n_bands,
                       n, c->hi.p, c->pcb, rm, io.surv, io.cap, io.pc);
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_select(bool fma, hipStream_t st, unsigned grid, const uint32_t* counts, const u64* n_dev, u64 n_host,
                         u64 cap, int p, double* est, const selhip_int2_t* pairs, const u64* ecard, double tau,
                         selhip_pair_t* results, u64 results_cap, PassCounters* pc,
                         selhip_result_t* rf32, int* out_count, u64 chunk_off = 0, u64 chunk_len = ~0ull) {
    const double rs = relerr_scaled_for(p);
    if (fma)
        hipLaunchKernelGGL((ertl_select_kernel<true, MODE>), dim3((grid + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, st, counts, n_dev, n_host, cap,
                           p, rs, est, pairs, ecard, tau, results, results_cap, pc, rf32, out_count, chunk_off, chunk_len);
    else
        hipLaunchKernelGGL((ertl_select_kernel<false, MODE>), dim3((grid + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, st, counts, n_dev, n_host, cap,
                           p, rs, est, pairs, ecard, tau, results, results_cap, pc, rf32, out_count, chunk_off, chunk_len);
    return hipGetLastError();
}

unsigned grid_for(u64 items, unsigned per_block, unsigned max_blocks) {
    u64 b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

// ---- stage 2a on bit planes (kernel_hllbs.cuh) -------------------------------------------------
// the instantiation for a set whose largest register value is khi - 1 = the number of bit planes that can be non-zero
hipError_t launch_hist_bs(int khi, unsigned blocks, hipStream_t st, const uint32_t* bs, const uint8_t* gmax, const selhip_int2_t* list, const u64* count,
                          u64 cap, uint32_t* counts, u64 off, u64 window, int run) {
#define SELHIP_BS_LAUNCH(NB) hipLaunchKernelGGL((hll_union_hist_bs_kernel<NB>), dim3(blocks), dim3(kBlock), 0, st, bs, gmax, list, count, cap, counts, off, window, run)
    if (khi <= 16)      SELHIP_BS_LAUNCH(4);
    else if (khi <= 32) SELHIP_BS_LAUNCH(5);
    else                SELHIP_BS_LAUNCH(6);
#undef SELHIP_BS_LAUNCH
    return hipGetLastError();
}

// writes the bit planes of n genomes and returns max register value + 1 through *khi (waits for the stream)
int build_bitslices(std::string* err, hipStream_t st, const uint8_t* d_hll, int64_t n, uint32_t* d_bs, uint8_t* d_gmax, int* d_max, int* khi) {
    HIPCHK(err, hipMemsetAsync(d_max, 0, sizeof(int), st));
    hipLaunchKernelGGL(hll_bitslice_kernel, dim3(grid_for((u64)n, kWavesPerBlock, 8192)), dim3(kBlock), 0, st, d_hll, (long long)n, d_bs, d_gmax, d_max);
    HIPCHK(err, hipGetLastError());
    int mx = 0;
    HIPCHK(err, hipMemcpyAsync(&mx, d_max, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(err, hipStreamSynchronize(st));
    *khi = mx + 1;
    return SELHIP_OK;
}

bool use_bitslices(const selhip_ctx* c) { return c->p == 14 && c->hll_khi > 0 && c->hist_algo != 0; }

int compute_cards(selhip_ctx* c, const uint8_t* d_hll, int64_t n, int p, double* d_out) {
    if (n <= 0) return SELHIP_OK;
    HIPCHK(&c->err, c->self_pairs.ensure((size_t)n));
    HIPCHK(&c->err, c->counts.ensure((size_t)n * 64));
    hipLaunchKernelGGL(iota_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->self_pairs.p, (int)n);
    HIPCHK(&c->err, hipGetLastError());
    hipLaunchKernelGGL(hll_union_hist_kernel, dim3(grid_for((u64)n, kWavesPerBlock, 4096)), dim3(kBlock), 0, c->stream,
                       d_hll, p, c->self_pairs.p, (const u64*)nullptr, (u64)n, (u64)n, c->counts.p);
    HIPCHK(&c->err, hipGetLastError());
    HIPCHK(&c->err, launch_select<0>(c->fp_mode == SELHIP_FP_FMA, c->stream, grid_for((u64)n, kWave, 8192), c->counts.p,
                                     nullptr, (u64)n, (u64)n, p, d_out, nullptr, nullptr, 0.0, nullptr, 0, nullptr,
                                     nullptr, nullptr));
    return SELHIP_OK;
}

// criteria_sketch.hpp:7-20 sigma(p): a double expression narrowed to float by the return type
float sigma_p_of(int p) {
    switch (p) {
        case 4: return (float)(1.106 / std::sqrt((double)(1 << p)));
        case 5: return (float)(1.07 / std::sqrt((double)(1 << p)));
        case 6: return (float)(1.054 / std::sqrt((double)(1 << p)));
        case 7: return (float)(1.046 / std::sqrt((double)(1 << p)));
    }
    return (float)(1.039 / std::sqrt((double)(1 << p)));
}

// upper bound of the pair space of rows [rb, re): the triangle (CB can only shrink it)
long long pair_bound(long long n, long long rb, long long re) {
    long long cnt = 0;
    // sum_{i=rb}^{re-1} (n-1-i)
    const long long rows = re - rb;
    cnt = rows * (n - 1) - (rb + re - 1) * rows / 2;
    return cnt < 0 ? 0 : cnt;
}

template <int CRIT>
hipError_t launch_aux_fused(selhip_ctx* c, hipStream_t st, const selhip_int2_t* list, const u64* n_dev, u64 cap, u64 bound, double tau,
                            selhip_int2_t* out, u64 out_cap, u64* out_count) {
    const float Z = 1.96f;                                   // z_score, selection.cpp:76
    const float zs_f = Z * sigma_p_of(c->p_aux);             // float * float (criteria_sketch.hpp:29,40)
    const double zs = (double)zs_f;
    const double S_sum = zs;                                 // order_n = 1 (selection.cpp:77): S = Z*sigma_p
    const double rs = relerr_scaled_for(c->p_aux);
    const unsigned grid = grid_for(bound, kWave, 32768);
    if (c->fp_mode == SELHIP_FP_FMA)
        hipLaunchKernelGGL((aux_fused_kernel<true, CRIT>), dim3(grid), dim3(kWave), 0, st, c->d_aux_hll, c->p_aux, list, n_dev, cap,
                           rs, c->ecard.p, tau, zs, S_sum, out, out_cap, out_count);
    else
        hipLaunchKernelGGL((aux_fused_kernel<false, CRIT>), dim3(grid), dim3(kWave), 0, st, c->d_aux_hll, c->p_aux, list, n_dev, cap,
                           rs, c->ecard.p, tau, zs, S_sum, out, out_cap, out_count);
    return hipGetLastError();
}

// equal-pair row boundaries of the triangle rows [rb, re) x columns (row, n): the same cut the multi-GPU drivers use
void chunk_rows(long long n, long long rb, long long re, int chunks, long long period, long long* bnd) {
    // boundaries fall on whole interleave periods counted from rb (row ownership is defined relative to the range's first row)
    const double total = (double)pair_bound(n, rb, re);
    bnd[0] = rb;
    long long i = rb;
    double acc = 0;
    for (int c = 1; c < chunks; ++c) {
        const double target = total * c / chunks;
        while (i < re && acc < target) {
            const long long e = std::min(re, i + period);
            acc += (double)pair_bound(n, i, e);
            i = e;
        }
        bnd[c] = i;
    }
    bnd[chunks] = re;
}

int pipeline_chunks(const selhip_ctx* c) {
    // Round 1's pipeline (stage 1 of every chunk on one stream, stage 2 on another) lost on every configuration and was replaced
    // by whole-chain lanes (enqueue_pass).  Automatic setting: two chunks for the signature join once a pass is large enough for
    // the second set of tail launches to cost less than the overlap wins (measured: profiles/r02_chunk_lanes.txt).
    const bool smh = c->criterion == SELHIP_CRIT_SMH_A || c->criterion == SELHIP_CRIT_HLL_A_SMH_A;
    if (!smh || c->pipeline == 0 || c->pipeline == 1 || c->algo == SELHIP_ALGO_HASHJOIN) return 1;   // (the sort join works on all rows at once)
    if (c->pipeline > 1) return std::min(c->pipeline, kMaxChunks);
    const bool sig = c->algo != SELHIP_ALGO_STREAM && c->join_bits <= 16 && sig_supported(c->m, c->n_rows, c->n_bands);
    if (!sig || !(c->p == 14 && c->group_stage2)) return 1;
    const double pairs = (double)pair_bound(c->n, c->row_begin, c->row_end) / std::max(1, c->il_parts);
    return pairs >= kAutoChunkPairs ? 2 : 1;
}

// Wait for the context's stream with low wake-up latency: poll for up to ~2 ms (a pass of the BASELINE single-GPU
// configurations takes 0.5-20 ms and the blocking wait's wake-up costs tens of microseconds), then block.
hipError_t wait_stream(hipStream_t st) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    return hipStreamSynchronize(st);
}

// one chain of a pass: the stream it runs on, the query rows it covers and its slices of the pass's buffers
struct Chain {
    StageIO io;
    int rb, re;
    selhip_int2_t* fin; u64 fin_cap;        // output of the auxiliary criterion
    int* csr_cnt; int* csr_start; char* scan_tmp;
    selhip_int2_t* grouped;
    uint32_t* counts; u64 window;           // histogram scratch: `window` pairs at a time
};

// per-chain row arrays of the grouping: counts | fill cursors | labels | label-group sums (then bucket starts) | roots, n ints each
size_t csr_stride(int n) { return 5 * (size_t)n + 2; }

constexpr double kLabelOrderPairs = 4e8;
constexpr size_t kLabelOrderBytes = (size_t)192 << 20;     // HLL rows beyond this (the Infinity Cache holds 256 MiB): label order
bool label_order(const selhip_ctx* c) {
    if (!(c->p == 14 && c->group_stage2)) return false;
    if (c->group_label >= 0) return c->group_label == 1;
    // the bit-plane kernel is bound by its fetches from beyond L2 at every size (cfg3: stage 2a 79 -> 54 us with the label order)
    if (use_bitslices(c)) return true;
    // its three extra launches (~15 us) only pay where stage 2a is bound by fetches from beyond L2 AND has enough pairs: HLL rows
    // beyond the Infinity Cache and -- the proxy known here -- a large pair space (one of 8 ranks of cfg4, 1.6e8 pairs: 0.515 -> 0.529 ms
    // with it; one of 8 ranks of cfg5, 6.2e8: 1.613 -> 1.577 ms; cfg3, whose rows fit the Infinity Cache: 0.311 -> 0.313 ms)
    return (size_t)c->n * 16384 > kLabelOrderBytes && (double)pair_bound(c->n, c->row_begin, c->row_end) / std::max(1, c->il_parts) >= kLabelOrderPairs;
}

int enqueue_tail(selhip_ctx* c, const Chain& ch, const selhip_int2_t* final_list, const u64* final_count, u64 final_cap,
                 bool counted, double tau, PassCounters* pc0) {
    const int n = (int)c->n;
    hipStream_t st = ch.io.st;
    const bool grouped = c->p == 14 && c->group_stage2;
    if (grouped) {
        // bucket the final list by query row so that stage 2a can keep that row in registers across its pairs
        TimerScope t(c, T_GROUP, st);
        // (the counters were cleared by the pass's first kernel)
        const bool label = label_order(c);
        int* const cnt = ch.csr_cnt; int* const fill = cnt + n; int* const lab = cnt + 2 * (size_t)n; int* const gsum = cnt + 3 * (size_t)n;
        if (!counted) {
            hipLaunchKernelGGL(csr_count_kernel, dim3(512), dim3(kBlock), 0, st, final_list, final_count, final_cap, cnt, label ? lab : nullptr, n);
            HIPCHK(&c->err, hipGetLastError());
        }
        if (label && n <= kSmallScanMax) {
            int* const root = cnt + 4 * (size_t)n;
            hipLaunchKernelGGL(csr_label_offsets_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, cnt, lab, n, gsum, root, ch.csr_start);
            HIPCHK(&c->err, hipGetLastError());
            if ((size_t)n * sizeof(int) > 48 * 1024)
                HIPCHK(&c->err, hipFuncSetAttribute((const void*)csr_label_scan_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmallScanMax * 4));
            hipLaunchKernelGGL(csr_label_scan_fill_kernel, dim3(256), dim3(1024), (size_t)n * sizeof(int), st, gsum, n, root, ch.csr_start,
                               final_list, final_count, final_cap, fill, ch.grouped);
            HIPCHK(&c->err, hipGetLastError());
        } else if (label) {
            const unsigned row_blocks = (unsigned)((n + kBlock - 1) / kBlock);
            hipLaunchKernelGGL(csr_label_sum_kernel, dim3(row_blocks), dim3(kBlock), 0, st, cnt, lab, n, gsum);
            HIPCHK(&c->err, hipGetLastError());
            size_t tmp_bytes = c->scan_tmp_stride;
            HIPCHK(&c->err, rocprim::exclusive_scan(ch.scan_tmp, tmp_bytes, gsum, ch.csr_start, 0, (size_t)n, rocprim::plus<int>(), st));
            hipLaunchKernelGGL(csr_label_assign_kernel, dim3(row_blocks), dim3(kBlock), 0, st, cnt, lab, n, ch.csr_start, gsum);   // gsum := bucket starts
            HIPCHK(&c->err, hipGetLastError());
            hipLaunchKernelGGL(csr_fill_kernel, dim3(512), dim3(kBlock), 0, st, final_list, final_count, final_cap, gsum, fill, ch.grouped);
            HIPCHK(&c->err, hipGetLastError());
        } else if (n <= kSmallScanMax) {
            if ((size_t)n * sizeof(int) > 48 * 1024)       // per device, so not cached in a process-wide flag (selhip_multi_select drives several)
                HIPCHK(&c->err, hipFuncSetAttribute((const void*)csr_scan_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmallScanMax * 4));
            hipLaunchKernelGGL(csr_scan_fill_kernel, dim3(256), dim3(1024), (size_t)n * sizeof(int), st, cnt, n,
                               final_list, final_count, final_cap, fill, ch.grouped);
            HIPCHK(&c->err, hipGetLastError());
        } else {
            size_t tmp_bytes = c->scan_tmp_stride;
            HIPCHK(&c->err, rocprim::exclusive_scan(ch.scan_tmp, tmp_bytes, cnt, ch.csr_start, 0, (size_t)n, rocprim::plus<int>(), st));
            hipLaunchKernelGGL(csr_fill_kernel, dim3(512), dim3(kBlock), 0, st, final_list, final_count, final_cap,
                               ch.csr_start, fill, ch.grouped);
            HIPCHK(&c->err, hipGetLastError());
        }
        final_list = ch.grouped;
    }
    for (u64 off = 0; off < final_cap; off += ch.window) {
        {
            TimerScope t(c, T_HIST, st);
            if (use_bitslices(c))
                HIPCHK(&c->err, launch_hist_bs(c->hll_khi, (unsigned)c->hist_bs_blocks, st, c->hll_bs.p, c->hll_gmax.p, final_list, final_count, final_cap, ch.counts, off, ch.window,
                                               c->hist_run > 0 ? c->hist_run : (grouped ? 4 : 1)));
            else if (c->p == 14)
                hipLaunchKernelGGL(hll_union_hist_runs_kernel, dim3(c->hist_blocks), dim3(kWave), (size_t)c->hist_pad, st,
                                   c->d_hll, final_list, final_count, final_cap, ch.counts, off, ch.window,
                                   c->hist_run > 0 ? c->hist_run : (grouped && label_order(c) ? 4 : 1));
            else
                hipLaunchKernelGGL(hll_union_hist_kernel, dim3(2048), dim3(kBlock), 0, st,
                                   c->d_hll, c->p, final_list, final_count, (u64)0, final_cap, ch.counts, off, ch.window);
            HIPCHK(&c->err, hipGetLastError());
        }
        TimerScope t(c, T_SELECT, st);
        HIPCHK(&c->err, launch_select<1>(c->fp_mode == SELHIP_FP_FMA, st, 4096, ch.counts, final_count, 0,
                                         final_cap, c->p, nullptr, final_list, c->ecard.p, tau,
                                         c->results.p, (u64)c->results.cap, pc0, nullptr, nullptr, off, ch.window));
    }
    return SELHIP_OK;
}

// smh_a (alone or before the auxiliary criterion) over the query rows of one chain, then the final criterion.  [rb, re) is the
// pass's whole row range (row ownership under the interleave is counted from its first row)
int enqueue_chain(selhip_ctx* c, const Chain& ch, int rb, int re, double tau, bool use_hash, bool use_sig, bool count_in_verify,
                  PassCounters* pc0) {
    const StageIO& io = ch.io;
    RowMap rm = row_map(c, rb, re);
    if (c->il_parts <= 1) rm = row_map(c, ch.rb, ch.re);
    else { rm.row_begin = ch.rb; rm.row_end = ch.re; }          // ch.rb - rb is a multiple of the interleave period
    {
        TimerScope t(c, T_STAGE1, io.st);
        if (use_hash)     HIPCHK(&c->err, launch_stage1_hashjoin(c, io, c->n_rows, c->n_bands, rm));
        else if (use_sig) HIPCHK(&c->err, launch_stage1_sig(c, io, c->n_rows, c->n_bands, rm));
        else              HIPCHK(&c->err, launch_stage1(c, io, c->n_rows, c->n_bands, rm));
    }
    const selhip_int2_t* final_list = io.surv;
    const u64* final_count = &io.pc->n_survivors;
    u64 final_cap = io.cap;
    if (c->criterion == SELHIP_CRIT_HLL_A_SMH_A) {
        // two-stage form (BASELINE configs[4]): the auxiliary criterion (histogram + estimator + test fused, one lane per pair)
        // sees the survivors of the smh_a join
        TimerScope t(c, T_AUX, io.st);
        HIPCHK(&c->err, launch_aux_fused<1>(c, io.st, io.surv, &io.pc->n_survivors, io.cap, io.cap, tau, ch.fin, ch.fin_cap, &io.pc->n_final));
        final_list = ch.fin;
        final_count = &io.pc->n_final;
        final_cap = ch.fin_cap;
    }
    return enqueue_tail(c, ch, final_list, final_count, final_cap, count_in_verify, tau, pc0);
}

int enqueue_pass(selhip_ctx* c) {
    const int n = (int)c->n;
    const int rb = (int)c->row_begin, re = (int)c->row_end;
    const double tau = (double)c->tau_f;            // float threshold widened, selection.cpp:81,164
    const int crit = c->criterion;
    {
        const bool smh = crit == SELHIP_CRIT_SMH_A || crit == SELHIP_CRIT_HLL_A_SMH_A;
        const bool sig = smh && (c->algo == SELHIP_ALGO_HASHJOIN ||
                                 ((c->algo == SELHIP_ALGO_SIG || c->algo == SELHIP_ALGO_AUTO) && sig_supported(c->m, c->n_rows, c->n_bands)));
        c->dominant_timer = c->timed_kernel == 1 ? T_HIST : (sig ? T_JOIN : T_STAGE1);
        if (c->timing) c->timed_passes += 1;
    }
    TimerScope total(c, T_TOTAL);
    const bool smh_crit = crit == SELHIP_CRIT_SMH_A || crit == SELHIP_CRIT_HLL_A_SMH_A;
    const bool use_hash = smh_crit && c->algo == SELHIP_ALGO_HASHJOIN;
    const bool use_sig = use_hash || (smh_crit && (c->algo == SELHIP_ALGO_SIG || c->algo == SELHIP_ALGO_AUTO) &&
                                      sig_supported(c->m, c->n_rows, c->n_bands));
    if (smh_crit && c->algo == SELHIP_ALGO_SIG && !use_sig) {
        set_err(&c->err, "ALGO_SIG needs power-of-two rows and 8..128 bands (got %d x %d)", c->n_rows, c->n_bands);
        return SELHIP_E_BADARG;
    }
    if (use_hash && (!is_pow2(c->n_rows) || c->n_bands > 65536)) {
        set_err(&c->err, "ALGO_HASHJOIN needs power-of-two rows (got %d x %d)", c->n_rows, c->n_bands);
        return SELHIP_E_BADARG;
    }
    // counter set of this pass (block 0: z0, evaluated, results; blocks 1.. : one per row chunk); the other set is cleared by
    // this pass's first kernel for the next pass -- no memset dispatch on the stream.  (Chosen only now: nothing above launches, and
    // an argument error must not consume a set that no kernel has cleared.)
    if (c->pc_dirty) {
        // the previous enqueue failed after it had claimed its counter set (its first kernel, which clears the other set for this pass,
        // may never have run), or the stream changed behind the initial memset: clear both sets here, once
        HIPCHK(&c->err, hipMemsetAsync(c->pc.p, 0, sizeof(PassCounters) * 2 * (kMaxChunks + 1), c->stream));
    }
    c->pcb = c->pc.p + (size_t)c->pc_flip * (kMaxChunks + 1);
    PassCounters* const pc_next = c->pc.p + (size_t)(c->pc_flip ^ 1) * (kMaxChunks + 1);
    c->pc_flip ^= 1;
    c->pc_dirty = true;                                  // until this function returns SELHIP_OK
    PassCounters* pc0 = c->pcb;
    if (c->fail_after_flip) { c->fail_after_flip = 0; set_err(&c->err, "test hook: enqueue failed after the counter flip"); return SELHIP_E_HIP; }
    if (use_sig) {
        // bounds (truncated cards, CB cut-offs, z0, evaluated count) ride in the first blocks of the signature build
        HIPCHK(&c->err, launch_sig_build(c, c->n_rows, c->n_bands, true, tau, rb, re, pc_next));
    } else {
        TimerScope t(c, T_PREP);
        hipLaunchKernelGGL(cb_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_cards, n, tau, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, row_map(c, rb, re), c->ecard.p, c->hi.p, pc0,
                           (c->p == 14 && c->group_stage2) ? c->csr_cnt.p : nullptr, (c->p == 14 && c->group_stage2) ? (int)c->csr_cnt.cap : 0, (int)c->cand_begin, pc_next,
                           c->seg_cnt.p, (int)c->seg_cnt.cap);
        HIPCHK(&c->err, hipGetLastError());
        if (smh_crit && stream_supported(c->m, c->n_rows)) {
            // ALGO_STREAM: the bucket-interleaved copy of the sketches (lane l = buckets [l*B, (l+1)*B)), rebuilt every pass
            const int nch = c->m / 128;
            const long long total = (long long)c->n * nch * kWave;
            hipLaunchKernelGGL(stream_interleave_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                               reinterpret_cast<const u64x2*>(c->d_aux), reinterpret_cast<u64x2*>(c->aux_il.p), total, nch);
            HIPCHK(&c->err, hipGetLastError());
        }
    }

    const int chunks = pipeline_chunks(c);
    c->n_chunks_last = chunks;
    const bool count_in_verify = crit == SELHIP_CRIT_SMH_A && use_sig && !use_hash && c->join_bits <= 16 && c->p == 14 && c->group_stage2;
    // chunk k's slices of the pass's buffers (one chunk = the whole of each)
    auto chain_of = [&](int k, hipStream_t st, long long b, long long e) {
        const u64 slice = (u64)c->surv.cap / (u64)chunks;
        Chain ch;
        ch.io = StageIO{st, c->cand.p + (size_t)k * slice, c->surv.p + (size_t)k * slice, slice, pc0 + 1 + k};
        ch.io.seg_cnt = c->seg_cnt.p + (size_t)(1 + k) * kAppendSegs * kSegStride;
        ch.rb = (int)b; ch.re = (int)e;
        ch.fin = c->fin.p ? c->fin.p + (size_t)k * ((u64)c->fin.cap / (u64)chunks) : nullptr;
        ch.fin_cap = (u64)c->fin.cap / (u64)chunks;
        ch.csr_cnt = c->csr_cnt.p ? c->csr_cnt.p + (size_t)k * csr_stride(n) : nullptr;
        ch.csr_start = c->csr_start.p ? c->csr_start.p + (size_t)k * ((size_t)n + 2) : nullptr;
        ch.scan_tmp = c->scan_tmp.p ? c->scan_tmp.p + (size_t)k * c->scan_tmp_stride : nullptr;
        ch.grouped = c->grouped.p ? c->grouped.p + (size_t)k * slice : nullptr;
        ch.window = ((u64)c->counts.cap / 64) / (u64)chunks;
        ch.counts = c->counts.p + (size_t)k * ch.window * 64;
        if (count_in_verify) { ch.io.row_cnt = ch.csr_cnt; if (label_order(c)) ch.io.row_lab = ch.csr_cnt + 2 * (size_t)n; }
        return ch;
    };
    if (chunks > 1) {
        // ---- row chunks, each a whole chain (join -> verify -> [auxiliary criterion] -> grouping -> histogram -> estimate) on one of
        // two streams: while one chunk's short tail kernels (tens of microseconds each, far too few waves to fill the chip) run, the
        // other chunk's join has the vector units, and the join's own ramp and tail overlap with the neighbour.  Measured with two
        // contexts on two streams before it was built (scripts/overlap_probe.py): cfg4 on one of 8 ranks 0.551 -> 0.544 ms even with
        // the signature build done twice.
        long long bnd[kMaxChunks + 1];
        chunk_rows(n, rb, re, chunks, c->il_parts > 1 ? (long long)c->il_block * c->il_parts : 1, bnd);
        // lane 0 is the context's own stream (cross-stream waits cost ~10 us each: one to start lane 1, one to join it).
        // (Staggering the lanes -- chunk k's join waits for chunk k-1's join, so that every tail runs beside the NEXT join and only the
        // last tail is exposed -- was measured: cfg4 2.78 vs 2.73 ms, cfg5 9.79 vs 9.74 ms with 2 chunks, no better with 4: the tail
        // kernels take from the join what they use, the chip is not idle in either phase.  profiles/r02_chunk_lanes.txt)
        hipStream_t lane[2] = {c->stream, c->st_stage1};
        HIPCHK(&c->err, hipEventRecord(c->ev_start, c->stream));
        HIPCHK(&c->err, hipStreamWaitEvent(lane[1], c->ev_start, 0));
        for (int k = 0; k < chunks; ++k) {
            // odd chunks first in program order so that lane 1's work is queued before lane 0's blocks the host thread's view
            const Chain ch = chain_of(k, lane[(k & 1) ^ 1], bnd[k], bnd[k + 1]);
            const int rc = enqueue_chain(c, ch, rb, re, tau, use_hash, use_sig, count_in_verify, pc0);
            if (rc) return rc;
        }
        HIPCHK(&c->err, hipEventRecord(c->ev_end, lane[1]));
        HIPCHK(&c->err, hipStreamWaitEvent(c->stream, c->ev_end, 0));
        HIPCHK(&c->err, hipMemcpyAsync(c->h_pc, c->pcb, sizeof(PassCounters) * (kMaxChunks + 1), hipMemcpyDeviceToHost, c->stream));
        c->pc_dirty = false;
        return SELHIP_OK;
    }

    // ---- single chunk: everything in order on the context's stream (counter block 1)
    const Chain ch = chain_of(0, c->stream, rb, re);
    if (smh_crit) {
        const int rc = enqueue_chain(c, ch, rb, re, tau, use_hash, use_sig, count_in_verify, pc0);
        if (rc) return rc;
    } else {
        // hll_a / hll_an as FIRST criterion (selection.cpp:152-173, 206-227): the (CB-pruned) pair space of the rows is listed
        // explicitly, kEnumPairs pairs at a time -- row sub-ranges in turn on the stream, each listed into the same buffer and
        // filtered into `fin` before the next one overwrites it (the reference has no limit on N here; round 1 refused
        // more than 2^28 pairs per call).  Sub-range boundaries fall on whole interleave periods so that row ownership
        // (RowMap blocks are counted from the range's first row) is the same as for the whole range.
        const StageIO& io = ch.io;
        const long long period = c->il_parts > 1 ? (long long)c->il_block * c->il_parts : 1;
        long long sb = rb;
        while (sb < re) {
            long long se = sb;
            long long acc = 0;
            while (se < re) {
                const long long step_end = std::min<long long>(re, se + period);
                const long long add = pair_bound(n, se, step_end);
                if (se > sb && acc + add > c->enum_pairs) break;
                acc += add; se = step_end;
            }
            if ((u64)acc + 1024 > (u64)c->cand.cap) { set_err(&c->err, "internal: enumeration buffer too small for rows [%lld,%lld)", sb, se); return SELHIP_E_OVERFLOW; }
            {
                TimerScope t(c, T_STAGE1);
                HIPCHK(&c->err, hipMemsetAsync(&io.pc->n_aux_in, 0, sizeof(u64), c->stream));
                RowMap rm = row_map(c, rb, re);
                if (c->il_parts <= 1) { rm = row_map(c, (int)sb, (int)se); }
                else { rm.row_begin = (int)sb; rm.row_end = (int)se; }            // sb - rb is a multiple of the interleave period
                const long long rows = rm.n_tiles(1);
                const long long blocks = rows * ((n + kEnumSpan - 1) / kEnumSpan);
                if (blocks > 0x7FFFFFFFll) { set_err(&c->err, "row range too large"); return SELHIP_E_BADARG; }
                if (blocks > 0) {
                    hipLaunchKernelGGL(enum_pairs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, n, c->hi.p, pc0,
                                       rm, (int)rows, c->cand.p, (u64)c->cand.cap, io.pc);
                    HIPCHK(&c->err, hipGetLastError());
                }
            }
            TimerScope t(c, T_AUX);
            const u64 bound = std::min<u64>((u64)c->cand.cap, (u64)acc);
            if (crit == SELHIP_CRIT_HLL_AN) HIPCHK(&c->err, launch_aux_fused<2>(c, c->stream, c->cand.p, &io.pc->n_aux_in, (u64)c->cand.cap, bound, tau, ch.fin, ch.fin_cap, &io.pc->n_final));
            else                            HIPCHK(&c->err, launch_aux_fused<1>(c, c->stream, c->cand.p, &io.pc->n_aux_in, (u64)c->cand.cap, bound, tau, ch.fin, ch.fin_cap, &io.pc->n_final));
            sb = se;
        }
        const int rc = enqueue_tail(c, ch, ch.fin, &io.pc->n_final, ch.fin_cap, false, tau, pc0);
        if (rc) return rc;
    }
    // (handing the counters to the host from the last block of the final kernel instead of this copy was tried: the 1 024
    // "block done" atomics on one address cost 16 us, the copy dispatch 4)
    HIPCHK(&c->err, hipMemcpyAsync(c->h_pc, c->pcb, sizeof(PassCounters) * (kMaxChunks + 1), hipMemcpyDeviceToHost, c->stream));
    c->pc_dirty = false;
    return SELHIP_OK;
}

int ensure_scratch(selhip_ctx* c, size_t surv_cap, size_t res_cap) {
    HIPCHK(&c->err, c->ecard.ensure((size_t)c->n));
    HIPCHK(&c->err, c->hi.ensure((size_t)c->n));
    if (!c->pc.p) {
        HIPCHK(&c->err, c->pc.ensure(2 * (kMaxChunks + 1)));
        HIPCHK(&c->err, hipMemsetAsync(c->pc.p, 0, sizeof(PassCounters) * 2 * (kMaxChunks + 1), c->stream));
        c->pc_flip = 0;
    }
    HIPCHK(&c->err, c->seg_cnt.ensure(kSegCounterSlots));
    if (!c->st_stage1) {
        HIPCHK(&c->err, hipStreamCreateWithFlags(&c->st_stage1, hipStreamNonBlocking));
        HIPCHK(&c->err, hipEventCreateWithFlags(&c->ev_start, hipEventDisableTiming));
        HIPCHK(&c->err, hipEventCreateWithFlags(&c->ev_end, hipEventDisableTiming));
    }
    HIPCHK(&c->err, c->surv.ensure(surv_cap));
    HIPCHK(&c->err, c->cand.ensure(surv_cap));
    if (c->criterion != SELHIP_CRIT_SMH_A) HIPCHK(&c->err, c->fin.ensure(surv_cap));
    if (c->criterion == SELHIP_CRIT_HLL_A || c->criterion == SELHIP_CRIT_HLL_AN) {
        // the explicit pair space is materialised kEnumPairs pairs at a time (8 B per pair); one interleave period of rows is the
        // smallest unit, so the buffer holds at least that
        const long long period = c->il_parts > 1 ? (long long)c->il_block * c->il_parts : 1;
        long long unit = 0;
        for (long long s = c->row_begin; s < c->row_end; s += period) unit = std::max(unit, pair_bound(c->n, s, std::min<long long>(c->row_end, s + period)));
        const long long bound = std::min(pair_bound(c->n, c->row_begin, c->row_end), std::max(c->enum_pairs, unit));
        HIPCHK(&c->err, c->cand.ensure((size_t)bound + 1024));
    }
    {
        const size_t n_pad = (((size_t)c->n + kWave - 1) / kWave) * kWave;
        const size_t nb = (size_t)std::max(c->n_bands, 1);
        const bool hash = c->algo == SELHIP_ALGO_HASHJOIN;
        const bool smh = c->criterion == SELHIP_CRIT_SMH_A || c->criterion == SELHIP_CRIT_HLL_A_SMH_A;
        const bool sig_path = hash || ((c->algo == SELHIP_ALGO_SIG || c->algo == SELHIP_ALGO_AUTO) && sig_supported(c->m, c->n_rows, c->n_bands));
        if (smh && !sig_path && stream_supported(c->m, c->n_rows)) HIPCHK(&c->err, c->aux_il.ensure((size_t)c->n * c->m));
        if (nb <= 128 || hash) {
            const uint32_t* const old_sig[4] = {c->sigQ.p, c->sigT.p, c->sigP.p, c->sigG.p};
            struct SigGuard { selhip_ctx* c; const uint32_t* const* o; ~SigGuard() { if (c->sigQ.p != o[0] || c->sigT.p != o[1] || c->sigP.p != o[2] || c->sigG.p != o[3]) c->sig_key = 0; } } guard{c, old_sig};
            HIPCHK(&c->err, c->sigQ.ensure((size_t)c->n * nb));
            HIPCHK(&c->err, c->sigT.ensure(n_pad * nb));
            HIPCHK(&c->err, c->sigP.ensure(n_pad * (size_t)((nb + 1) / 2)));
            HIPCHK(&c->err, c->sigG.ensure((n_pad + 2) * (size_t)((nb + 1) / 2)));
        }
        if (hash) {
            const size_t total = (size_t)c->n * nb;
            HIPCHK(&c->err, c->hj_keys_in.ensure(total)); HIPCHK(&c->err, c->hj_keys_out.ensure(total));
            HIPCHK(&c->err, c->hj_vals_in.ensure(total)); HIPCHK(&c->err, c->hj_vals_out.ensure(total));
            size_t tmp_bytes = 0;
            HIPCHK(&c->err, rocprim::radix_sort_pairs(nullptr, tmp_bytes, c->hj_keys_in.p, c->hj_keys_out.p, c->hj_vals_in.p,
                                                      c->hj_vals_out.p, total, 0u, 64u, c->stream));
            HIPCHK(&c->err, c->hj_tmp.ensure(tmp_bytes + 256));
        }
    }
    // histogram scratch: 256 B per pair, at most 4 Mi pairs per window (1 GiB of 288; the lists are sized for the join's 16-bit
    // matches, several times the final list, so a smaller window only adds empty histogram + estimate launches: 6 -> 2 per chain at cfg5)
    HIPCHK(&c->err, c->counts.ensure(std::min<size_t>(std::max(c->surv.cap, (size_t)c->n), (size_t)1 << 22) * 64));
    HIPCHK(&c->err, c->results.ensure(res_cap));
    if (c->group_stage2 && c->p == 14) {
        const size_t chunks = (size_t)pipeline_chunks(c);               // every chunk lane has its own row counters and scan scratch
        HIPCHK(&c->err, c->csr_cnt.ensure(chunks * csr_stride((int)c->n)));
        HIPCHK(&c->err, c->csr_start.ensure(chunks * ((size_t)c->n + 2)));
        HIPCHK(&c->err, c->grouped.ensure(surv_cap));
        size_t tmp_bytes = 0;
        HIPCHK(&c->err, rocprim::exclusive_scan(nullptr, tmp_bytes, c->csr_cnt.p, c->csr_start.p, 0, (size_t)c->n, rocprim::plus<int>(), c->stream));
        c->scan_tmp_stride = std::max(c->scan_tmp_stride, (tmp_bytes + 511) / 256 * 256);
        HIPCHK(&c->err, c->scan_tmp.ensure(chunks * c->scan_tmp_stride));
    }
    if (!c->h_pc) HIPCHK(&c->err, hipHostMalloc((void**)&c->h_pc, sizeof(PassCounters) * (kMaxChunks + 1), hipHostMallocDefault));
    return SELHIP_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

const char* selhip_version(void) { return "selhip 0.1 (gfx950)"; }

int selhip_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

const char* selhip_last_error(const selhip_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    return g_last_error.c_str();
}

int selhip_ctx_create(selhip_ctx** out, int device) {
    if (!out) return SELHIP_E_BADARG;
    *out = nullptr;
    int rc = check_device(nullptr);
    if (rc) return rc;
    int cnt = 0;
    (void)hipGetDeviceCount(&cnt);
    if (device < 0 || device >= cnt) { set_err(nullptr, "device %d out of range (0..%d)", device, cnt - 1); return SELHIP_E_BADARG; }
    HIPCHK(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_err(nullptr, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        return SELHIP_E_NODEVICE;
    }
    selhip_ctx* c = new selhip_ctx();
    c->device = device;
    *out = c;
    return SELHIP_OK;
}

void selhip_ctx_destroy(selhip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    c->own_hll.release(); c->own_aux.release(); c->own_cards.release();
    c->ecard.release(); c->hi.release(); c->pc.release(); c->seg_cnt.release(); c->surv.release();
    c->counts.release(); c->results.release(); c->self_pairs.release();
    c->aux_il.release(); c->cand.release(); c->sigQ.release(); c->sigT.release(); c->sigP.release(); c->sigG.release(); c->fin.release(); c->own_aux_hll.release();
    c->hj_keys_in.release(); c->hj_keys_out.release(); c->hj_vals_in.release(); c->hj_vals_out.release(); c->hj_tmp.release();
    c->csr_cnt.release(); c->csr_start.release(); c->grouped.release(); c->scan_tmp.release();
    c->hll_bs.release(); c->hll_bs_max.release(); c->hll_gmax.release();
    if (c->h_pc) (void)hipHostFree(c->h_pc);
    if (c->st_stage1) {
        (void)hipStreamDestroy(c->st_stage1);
        (void)hipEventDestroy(c->ev_start); (void)hipEventDestroy(c->ev_end);
    }
    delete c;
}

int selhip_ctx_set_stream(selhip_ctx* c, void* hip_stream) {
    if (!c) return SELHIP_E_BADARG;
    if (c->stream != (hipStream_t)hip_stream && c->pc.p) {
        // the counter sets were cleared by work on the old stream, which the new one is not ordered behind
        (void)hipStreamSynchronize(c->stream);
        c->pc_dirty = true;
    }
    c->stream = (hipStream_t)hip_stream;
    return SELHIP_OK;
}

int selhip_ctx_set_fp_mode(selhip_ctx* c, int fp_mode) {
    if (!c || (fp_mode != SELHIP_FP_FMA && fp_mode != SELHIP_FP_STRICT)) return SELHIP_E_BADARG;
    c->fp_mode = fp_mode;
    return SELHIP_OK;
}

int selhip_ctx_set_row_interleave(selhip_ctx* c, int block_rows, int n_parts, int part) {
    if (!c) return SELHIP_E_BADARG;
    if (n_parts <= 1) { c->il_parts = 1; c->il_part = 0; return SELHIP_OK; }
    if (block_rows < 32 || block_rows % 32 || part < 0 || part >= n_parts) {
        set_err(&c->err, "row interleave: block_rows must be a multiple of 32 (>= 32) and 0 <= part < n_parts");
        return SELHIP_E_BADARG;
    }
    c->il_block = block_rows; c->il_parts = n_parts; c->il_part = part;     // (the join's tile height is fitted to the block in join_tile_rows)
    return SELHIP_OK;
}

// Names beyond the documented ones (include/selection_hip.h) -- hooks of the test-suite and measurement knobs, not interface:
//   "verify_fb"        1 sends every candidate through the hash-collision fallback of the verification
//   "init_cap"         initial capacity of the candidate / survivor / result lists (they grow and the pass repeats)
//   "enum_pairs"       pairs listed per sub-pass when hll_a / hll_an is the first criterion (default 2^26)
//   "fail_after_flip"  the next enqueue fails right after claiming its counter set (recovery of the double-buffered counters)
//   "hist_pad"         extra LDS bytes per block of the byte-row stage-2a kernel (lowers the resident waves per CU)
int selhip_ctx_set_param(selhip_ctx* c, const char* name, int value) {
    if (!c || !name) return SELHIP_E_BADARG;
    if (!std::strcmp(name, "join_qt")) {
        if (value != 0 && (value < 16 || value > 4096 || value % 16)) { set_err(&c->err, "join_qt must be 0 (automatic) or a multiple of 16 in [16, 4096]"); return SELHIP_E_BADARG; }
        c->join_qt = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "verify_fb")) { c->verify_fb = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "fail_after_flip")) { c->fail_after_flip = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "join_wpb")) {
        if (value != 1 && value != 4 && value != 8) { set_err(&c->err, "join_wpb must be 1, 4 or 8"); return SELHIP_E_BADARG; }
        c->join_wpb = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "join_form")) { c->join_form = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "join_tri")) { c->join_tri = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "join_db")) { c->join_db = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "join_q")) { c->join_q = value != 0; return SELHIP_OK; }
    if (!std::strcmp(name, "sig_tile")) { c->sig_tile = value != 0; c->sig_key = 0; return SELHIP_OK; }
    if (!std::strcmp(name, "sig_cache")) { c->sig_cache = value != 0; c->sig_key = 0; return SELHIP_OK; }
    if (!std::strcmp(name, "enum_pairs")) {
        if (value < 1) { set_err(&c->err, "enum_pairs must be >= 1"); return SELHIP_E_BADARG; }
        c->enum_pairs = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "init_cap")) {
        if (value < 0) { set_err(&c->err, "init_cap must be >= 0"); return SELHIP_E_BADARG; }
        c->init_cap = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "join_bits")) {
        if (value != 15 && value != 16 && value != 32) { set_err(&c->err, "join_bits must be 15, 16 or 32"); return SELHIP_E_BADARG; }
        c->join_bits = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "group_label")) {
        if (value < -1 || value > 1) { set_err(&c->err, "group_label must be -1 (automatic), 0 or 1"); return SELHIP_E_BADARG; }
        c->group_label = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "hist_run")) {
        if (value < 0 || value > 1024) { set_err(&c->err, "hist_run must be in [0, 1024] (0 = automatic)"); return SELHIP_E_BADARG; }
        c->hist_run = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "hist_pad")) {
        if (value < 0 || value > 48 * 1024) { set_err(&c->err, "hist_pad must be in [0, 49152]"); return SELHIP_E_BADARG; }
        c->hist_pad = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "timed_kernel")) {
        if (value < 0 || value > 1) { set_err(&c->err, "timed_kernel must be 0 (stage-1 kernel) or 1 (stage 2a)"); return SELHIP_E_BADARG; }
        c->timed_kernel = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "hist_algo")) {
        // takes effect at the next selhip_ctx_upload / _attach (the bit planes are written there)
        if (value < -1 || value > 1) { set_err(&c->err, "hist_algo must be -1 (automatic), 0 (byte rows, LDS histogram) or 1 (bit planes)"); return SELHIP_E_BADARG; }
        c->hist_algo = value;
        if (value == 0) c->hll_khi = 0;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "hist_bs_blocks")) {
        if (value < 8 || value > 65536 || value % 8) { set_err(&c->err, "hist_bs_blocks must be a multiple of 8 in [8, 65536]"); return SELHIP_E_BADARG; }
        c->hist_bs_blocks = value;
        return SELHIP_OK;
    }
    if (!std::strcmp(name, "hist_blocks")) {
        if (value < 8 || value > 65536 || value % 8) { set_err(&c->err, "hist_blocks must be a multiple of 8 in [8, 65536]"); return SELHIP_E_BADARG; }
        c->hist_blocks = value;
        return SELHIP_OK;
    }
    set_err(&c->err, "unknown parameter '%s'", name);
    return SELHIP_E_BADARG;
}

int selhip_ctx_get_param(const selhip_ctx* c, const char* name, int* value) {
    if (!c || !name || !value) return SELHIP_E_BADARG;
    if (!std::strcmp(name, "hll_khi"))          { *value = c->hll_khi; return SELHIP_OK; }             // largest p = 14 register value + 1 (0: no bit planes)
    if (!std::strcmp(name, "hist_bitplanes"))   { *value = use_bitslices(c) ? 1 : 0; return SELHIP_OK; }
    if (!std::strcmp(name, "label_order"))      { *value = label_order(c) ? 1 : 0; return SELHIP_OK; }
    if (!std::strcmp(name, "join_tile_rows"))   { *value = join_tile_rows(c); return SELHIP_OK; }
    if (!std::strcmp(name, "chunks"))           { *value = c->n_chunks_last; return SELHIP_OK; }
    return SELHIP_E_BADARG;
}

int selhip_ctx_set_candidate_begin(selhip_ctx* c, int64_t k_min) {
    if (!c) return SELHIP_E_BADARG;
    if (k_min < 0 || k_min > c->n) { set_err(&c->err, "candidate_begin %lld outside [0, n]", (long long)k_min); return SELHIP_E_BADARG; }
    c->cand_begin = k_min;
    return SELHIP_OK;
}

int selhip_ctx_set_stage2_grouping(selhip_ctx* c, int enable) {
    if (!c) return SELHIP_E_BADARG;
    c->group_stage2 = enable != 0;
    return SELHIP_OK;
}

int selhip_ctx_set_pipeline(selhip_ctx* c, int chunks) {
    if (!c || chunks < -1 || chunks > kMaxChunks) return SELHIP_E_BADARG;
    c->pipeline = chunks;
    return SELHIP_OK;
}

int selhip_ctx_set_criterion(selhip_ctx* c, int criterion) {
    if (!c || criterion < SELHIP_CRIT_SMH_A || criterion > SELHIP_CRIT_HLL_A_SMH_A) return SELHIP_E_BADARG;
    c->criterion = criterion;
    return SELHIP_OK;
}

int selhip_ctx_upload_aux_hll(selhip_ctx* c, const uint8_t* h_aux_hll, int p_aux) {
    if (!c || !h_aux_hll || p_aux < 4 || p_aux > kMaxAuxP) return SELHIP_E_BADARG;     // (aux_fused_kernel's bins are 16 bits wide)
    if (!c->d_hll && c->n) { set_err(&c->err, "upload the primary sketches first"); return SELHIP_E_STATE; }
    HIPCHK(&c->err, hipSetDevice(c->device));
    const size_t bytes = (size_t)c->n << p_aux;
    HIPCHK(&c->err, c->own_aux_hll.ensure(bytes ? bytes : 1));
    if (bytes) HIPCHK(&c->err, hipMemcpyAsync(c->own_aux_hll.p, h_aux_hll, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    c->d_aux_hll = c->own_aux_hll.p; c->p_aux = p_aux;
    return SELHIP_OK;
}

int selhip_ctx_attach_aux_hll(selhip_ctx* c, const uint8_t* d_aux_hll, int p_aux) {
    if (!c || !d_aux_hll || p_aux < 4 || p_aux > kMaxAuxP) return SELHIP_E_BADARG;
    c->d_aux_hll = d_aux_hll; c->p_aux = p_aux;
    return SELHIP_OK;
}

static int validate_shape(selhip_ctx* c, int64_t n, int m, int p) {
    if (n < 0 || n > 0x7FFFFFF0ll) { set_err(&c->err, "n_genomes %lld out of range", (long long)n); return SELHIP_E_BADARG; }
    if (m <= 0) { set_err(&c->err, "m must be > 0"); return SELHIP_E_BADARG; }
    if (p < 4 || p > 20) { set_err(&c->err, "p_hll %d out of range [4,20]", p); return SELHIP_E_BADARG; }
    return SELHIP_OK;
}

static int after_sketches(selhip_ctx* c, const double* cards_src, bool cards_on_host) {
    // cards: given or computed with the device estimator
    c->hll_khi = 0;
    if (c->n == 0) return SELHIP_OK;
    if (c->p == 14 && c->hist_algo != 0) {
        // the registers once more as bit planes: what stage 2a reads (12 KiB per genome; the byte rows stay for report() and the callers)
        HIPCHK(&c->err, c->hll_bs.ensure((size_t)c->n * kBsGenomeDwords));
        HIPCHK(&c->err, c->hll_bs_max.ensure(1));
        HIPCHK(&c->err, c->hll_gmax.ensure((size_t)c->n));
        const int rc = build_bitslices(&c->err, c->stream, c->d_hll, c->n, c->hll_bs.p, c->hll_gmax.p, c->hll_bs_max.p, &c->hll_khi);
        if (rc) return rc;
    }
    if (!cards_src) {
        HIPCHK(&c->err, c->own_cards.ensure((size_t)c->n));
        int rc = compute_cards(c, c->d_hll, c->n, c->p, c->own_cards.p);
        if (rc) return rc;
        c->d_cards = c->own_cards.p;
    } else if (cards_on_host) {
        for (int64_t i = 0; i < c->n; ++i) {
            double v = cards_src[i];
            if (!(v >= 0.0) || !(v < 9.2e18)) { set_err(&c->err, "cards[%lld] = %g is not a finite value in [0, 2^63)", (long long)i, v); return SELHIP_E_BADARG; }
            if (i && v < cards_src[i - 1]) { set_err(&c->err, "cards are not in ascending order at rank %lld", (long long)i); return SELHIP_E_BADARG; }
        }
        HIPCHK(&c->err, c->own_cards.ensure((size_t)c->n));
        HIPCHK(&c->err, hipMemcpyAsync(c->own_cards.p, cards_src, (size_t)c->n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        c->d_cards = c->own_cards.p;
    } else {
        c->d_cards = cards_src;
    }
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    return SELHIP_OK;
}

int selhip_ctx_upload(selhip_ctx* c, const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                      int64_t n, int m, int p_hll) {
    if (!c) return SELHIP_E_BADARG;
    HIPCHK(&c->err, hipSetDevice(c->device));
    int rc = validate_shape(c, n, m, p_hll);
    if (rc) return rc;
    if (n > 0 && (!h_hll || !h_aux)) { set_err(&c->err, "null sketch pointer"); return SELHIP_E_BADARG; }
    c->n = n; c->m = m; c->p = p_hll; c->have_run = false; c->pending = false; c->cand_begin = 0; c->sig_key = 0;
    const size_t hb = (size_t)1 << p_hll;
    if (n > 0) {
        HIPCHK(&c->err, c->own_hll.ensure((size_t)n * hb));
        HIPCHK(&c->err, c->own_aux.ensure((size_t)n * m));
        HIPCHK(&c->err, hipMemcpyAsync(c->own_hll.p, h_hll, (size_t)n * hb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(&c->err, hipMemcpyAsync(c->own_aux.p, h_aux, (size_t)n * m * 8, hipMemcpyHostToDevice, c->stream));
    }
    c->d_hll = c->own_hll.p; c->d_aux = (const u64*)c->own_aux.p; c->owns_sketches = true;
    c->d_aux_hll = nullptr; c->p_aux = 0;
    return after_sketches(c, h_cards, true);
}

int selhip_ctx_attach(selhip_ctx* c, const uint8_t* d_hll, const uint64_t* d_aux, const double* d_cards,
                      int64_t n, int m, int p_hll) {
    if (!c) return SELHIP_E_BADARG;
    HIPCHK(&c->err, hipSetDevice(c->device));
    int rc = validate_shape(c, n, m, p_hll);
    if (rc) return rc;
    if (n > 0 && (!d_hll || !d_aux)) { set_err(&c->err, "null sketch pointer"); return SELHIP_E_BADARG; }
    if (((uintptr_t)d_hll & 15) || ((uintptr_t)d_aux & 15)) { set_err(&c->err, "sketch pointers must be 16-byte aligned"); return SELHIP_E_BADARG; }
    c->n = n; c->m = m; c->p = p_hll; c->have_run = false; c->pending = false; c->cand_begin = 0; c->sig_key = 0;
    c->d_hll = d_hll; c->d_aux = (const u64*)d_aux; c->owns_sketches = false;
    c->d_aux_hll = nullptr; c->p_aux = 0;
    return after_sketches(c, d_cards, false);
}

int selhip_hll_cards(selhip_ctx* c, const uint8_t* d_hll, int64_t n, int p, double* d_cards_out) {
    if (!c || !d_hll || !d_cards_out || n < 0 || p < 4 || p > 20) return SELHIP_E_BADARG;
    HIPCHK(&c->err, hipSetDevice(c->device));
    int rc = compute_cards(c, d_hll, n, p, d_cards_out);
    if (rc) return rc;
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    return SELHIP_OK;
}

int selhip_ctx_get_cards(selhip_ctx* c, double* h_out) {
    if (!c || !h_out) return SELHIP_E_BADARG;
    if (!c->d_cards && c->n) { set_err(&c->err, "no sketches uploaded"); return SELHIP_E_STATE; }
    HIPCHK(&c->err, hipSetDevice(c->device));
    if (c->n) HIPCHK(&c->err, hipMemcpyAsync(h_out, c->d_cards, (size_t)c->n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    return SELHIP_OK;
}

int selhip_ctx_run_async(selhip_ctx* c, int mode, int algo, float tau_f, int n_rows, int n_bands,
                         int64_t row_begin, int64_t row_end) {
    if (!c) return SELHIP_E_BADARG;
    if (!c->d_aux && c->n) { set_err(&c->err, "run before upload/attach"); return SELHIP_E_STATE; }
    if (mode != SELHIP_MODE_SMH && mode != SELHIP_MODE_CB_SMH) { set_err(&c->err, "bad mode %d", mode); return SELHIP_E_BADARG; }
    if (algo != SELHIP_ALGO_AUTO && algo != SELHIP_ALGO_STREAM && algo != SELHIP_ALGO_SIG && algo != SELHIP_ALGO_HASHJOIN) { set_err(&c->err, "bad algo %d", algo); return SELHIP_E_BADARG; }
    if (c->criterion != SELHIP_CRIT_SMH_A && !c->d_aux_hll && c->n) {
        set_err(&c->err, "criterion %d needs auxiliary HLL sketches (selhip_ctx_upload_aux_hll)", c->criterion);
        return SELHIP_E_STATE;
    }
    const bool needs_smh = c->criterion == SELHIP_CRIT_SMH_A || c->criterion == SELHIP_CRIT_HLL_A_SMH_A;
    if (needs_smh && (n_rows <= 0 || n_bands <= 0 || (long long)n_rows * n_bands != c->m)) {
        // criteria_sketch.hpp:67-70: the reference prints an error and selects nothing; the ABI reports it
        set_err(&c->err, "n_rows*n_bands (%d*%d) != m (%d)", n_rows, n_bands, c->m);
        return SELHIP_E_BADARG;
    }
    if (row_begin < 0 || row_end > c->n || row_begin > row_end) { set_err(&c->err, "bad row range [%lld,%lld)", (long long)row_begin, (long long)row_end); return SELHIP_E_BADARG; }
    HIPCHK(&c->err, hipSetDevice(c->device));
    c->mode = mode; c->algo = algo; c->tau_f = tau_f; c->n_rows = n_rows; c->n_bands = n_bands;
    c->row_begin = row_begin; c->row_end = row_end;
    c->have_run = false;
    std::memset(&c->last, 0, sizeof c->last);
    if (c->n == 0 || row_begin == row_end) { c->pending = false; c->have_run = true; return SELHIP_OK; }
    size_t surv_cap = std::max<size_t>(c->surv.cap, std::max<size_t>((size_t)1 << 20, (size_t)c->n * 16));
    if (needs_smh && c->join_bits <= 16 && algo != SELHIP_ALGO_STREAM && algo != SELHIP_ALGO_HASHJOIN && sig_supported(c->m, n_rows, n_bands)) {
        // the 16-bit join passes ~n_bands * 2^-16 of the pairs it compares on to the 32-bit filter: size the lists for that
        // up front (an overflow would only cost one repeated pass)
        const double expect = (double)pair_bound(c->n, (int)row_begin, (int)row_end) / std::max(1, c->il_parts) * n_bands / (c->join_bits == 15 ? 32768.0 : 65536.0);
        surv_cap = std::max(surv_cap, (size_t)std::min(expect * 1.25 + 65536.0, (double)((size_t)1 << 26)));
    }
    if (c->init_cap > 0) surv_cap = std::max<size_t>(c->surv.cap, (size_t)c->init_cap);       // test hook: start small, grow on overflow
    size_t res_cap = std::max<size_t>(c->results.cap, surv_cap);
    int rc = ensure_scratch(c, surv_cap, res_cap);
    if (rc) return rc;
    rc = enqueue_pass(c);
    if (rc) return rc;
    c->pending = true;
    return SELHIP_OK;
}

int selhip_ctx_finish(selhip_ctx* c) {
    if (!c) return SELHIP_E_BADARG;
    if (!c->pending) return c->have_run ? SELHIP_OK : SELHIP_E_STATE;
    HIPCHK(&c->err, hipSetDevice(c->device));
    for (int attempt = 0; attempt < 8; ++attempt) {
        HIPCHK(&c->err, wait_stream(c->stream));
        // block 0 = pass-wide counters; blocks 1..chunks = per-row-chunk list counters (each list slice = cap / chunks)
        PassCounters pc = c->h_pc[0];
        const int chunks = c->n_chunks_last;
        if (pc.unsorted) { c->pending = false; set_err(&c->err, "cards are not in ascending order"); return SELHIP_E_BADARG; }
        bool grow = false;
        size_t surv_cap = c->surv.cap, res_cap = c->results.cap;
        const size_t slice = c->surv.cap / (size_t)chunks;
        for (int k = 1; k <= chunks; ++k) {
            const PassCounters& q = c->h_pc[k];
            pc.n_survivors += q.n_survivors; pc.n_candidates += q.n_candidates; pc.n_aux_in += q.n_aux_in; pc.n_final += q.n_final;
            // the 16-bit join's list is kAppendSegs equal slices: it overflows when its fullest slice does
            const u64 worst = std::max(std::max(std::max(q.n_survivors, q.n_candidates), q.n_pre_segmax * (u64)kAppendSegs), c->criterion != SELHIP_CRIT_SMH_A ? q.n_final : 0);
            if (worst > slice) { surv_cap = std::max(surv_cap, (size_t)((worst + worst / 8 + 1024) * (u64)chunks)); grow = true; }
            // (n_aux_in is zeroed before every enumeration sub-pass, so what arrives here is the LAST sub-pass's count only: it proves
            //  nothing about the others.  The guarantee is the host-side bound in enqueue_pass -- every sub-pass lists at most
            //  cand.cap - 1024 pairs by construction -- and enum_pairs_kernel never writes past out_cap.)
        }
        if (pc.n_results > c->results.cap) { res_cap = (size_t)(pc.n_results + pc.n_results / 8 + 1024); grow = true; }
        if (!grow) {
            c->last = pc; c->pending = false; c->have_run = true; c->last_attempts = attempt + 1;
            // (event pairs are read lazily -- selhip_ctx_kernel_ms / _timing / destroy -- so that a timed run does not stall the
            // host between passes; a pass records at most ~20 of them)
            return SELHIP_OK;
        }
        // an output list was too small: counts are exact, so grow once and repeat the pass
        res_cap = std::max(res_cap, surv_cap);
        int rc = ensure_scratch(c, surv_cap, res_cap);
        if (rc) { c->pending = false; return rc; }
        rc = enqueue_pass(c);
        if (rc) { c->pending = false; return rc; }
    }
    c->pending = false;
    set_err(&c->err, "output buffers kept overflowing");
    return SELHIP_E_OVERFLOW;
}

int selhip_ctx_run(selhip_ctx* c, int mode, int algo, float tau_f, int n_rows, int n_bands,
                   int64_t row_begin, int64_t row_end) {
    int rc = selhip_ctx_run_async(c, mode, algo, tau_f, n_rows, n_bands, row_begin, row_end);
    if (rc) return rc;
    return selhip_ctx_finish(c);
}

int selhip_ctx_stats(const selhip_ctx* c, int64_t stats[4]) {
    if (!c || !stats) return SELHIP_E_BADARG;
    if (!c->have_run) return SELHIP_E_STATE;
    stats[0] = (int64_t)c->last.n_evaluated;
    stats[1] = (int64_t)(c->criterion == SELHIP_CRIT_SMH_A ? c->last.n_survivors : c->last.n_final);
    stats[2] = (int64_t)c->last.n_results;
    stats[3] = (int64_t)(c->last.n_candidates ? c->last.n_candidates : c->last.n_survivors);
    return SELHIP_OK;
}

int64_t selhip_ctx_result_count(const selhip_ctx* c) {
    if (!c || !c->have_run) return SELHIP_E_STATE;
    return (int64_t)c->last.n_results;
}

int selhip_ctx_fetch(selhip_ctx* c, selhip_pair_t* h_out, int64_t cap) {
    if (!c || (cap > 0 && !h_out) || cap < 0) return SELHIP_E_BADARG;
    if (!c->have_run) return SELHIP_E_STATE;
    const int64_t cnt = (int64_t)c->last.n_results;
    if (cnt == 0) return SELHIP_OK;
    HIPCHK(&c->err, hipSetDevice(c->device));
    std::vector<selhip_pair_t> tmp((size_t)cnt);
    HIPCHK(&c->err, hipMemcpyAsync(tmp.data(), c->results.p, (size_t)cnt * sizeof(selhip_pair_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(&c->err, hipStreamSynchronize(c->stream));
    std::sort(tmp.begin(), tmp.end(), [](const selhip_pair_t& a, const selhip_pair_t& b) {
        return a.i != b.i ? a.i < b.i : a.k < b.k;
    });
    std::memcpy(h_out, tmp.data(), (size_t)std::min(cnt, cap) * sizeof(selhip_pair_t));
    return cnt > cap ? SELHIP_E_OVERFLOW : SELHIP_OK;
}

int selhip_ctx_result_device(selhip_ctx* c, const selhip_pair_t** d_results, int64_t* count) {
    if (!c || !d_results || !count) return SELHIP_E_BADARG;
    if (!c->have_run) return SELHIP_E_STATE;
    *d_results = c->results.p;
    *count = (int64_t)c->last.n_results;
    return SELHIP_OK;
}

int selhip_ctx_copy_results(selhip_ctx* c, selhip_pair_t* d_dst, int64_t cap) {
    if (!c || cap < 0 || (cap > 0 && !d_dst)) return SELHIP_E_BADARG;
    if (!c->have_run) return SELHIP_E_STATE;
    const int64_t cnt = std::min<int64_t>((int64_t)c->last.n_results, cap);
    if (cnt > 0) {
        HIPCHK(&c->err, hipSetDevice(c->device));
        HIPCHK(&c->err, hipMemcpyAsync(d_dst, c->results.p, (size_t)cnt * sizeof(selhip_pair_t), hipMemcpyDeviceToDevice, c->stream));
    }
    return SELHIP_OK;
}

int selhip_ctx_copy_results_framed(selhip_ctx* c, void* d_dst, int64_t cap_records) {
    // frame = one 16-byte header record {u64 count, u64 0} followed by the records; both copies are device-to-device
    if (!c || !d_dst || cap_records < 0) return SELHIP_E_BADARG;
    if (!c->have_run) return SELHIP_E_STATE;
    HIPCHK(&c->err, hipSetDevice(c->device));
    HIPCHK(&c->err, hipMemcpyAsync(d_dst, &c->pcb->n_results, sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
    const int64_t cnt = std::min<int64_t>((int64_t)c->last.n_results, cap_records);
    if (cnt > 0)
        HIPCHK(&c->err, hipMemcpyAsync((char*)d_dst + sizeof(selhip_pair_t), c->results.p, (size_t)cnt * sizeof(selhip_pair_t),
                                       hipMemcpyDeviceToDevice, c->stream));
    return (int64_t)c->last.n_results > cap_records ? SELHIP_E_OVERFLOW : SELHIP_OK;
}

int selhip_ctx_copy_results_framed_async(selhip_ctx* c, void* d_dst, int64_t cap_records) {
    if (!c || !d_dst || cap_records < 0) return SELHIP_E_BADARG;
    if (!c->pending && !c->have_run) return SELHIP_E_STATE;
    if (!c->results.p || !c->pcb) return SELHIP_E_STATE;
    HIPCHK(&c->err, hipSetDevice(c->device));
    if ((uintptr_t)d_dst & 15) { set_err(&c->err, "frame buffer must be 16-byte aligned"); return SELHIP_E_BADARG; }
    static_assert(sizeof(selhip_pair_t) == 16, "frame records are 16 bytes");
    hipLaunchKernelGGL(frame_results_kernel, dim3(256), dim3(kBlock), 0, c->stream, c->results.p, &c->pcb->n_results,
                       (u64)c->results.cap, (u64)cap_records, (uint4*)d_dst);
    HIPCHK(&c->err, hipGetLastError());
    return SELHIP_OK;
}

int selhip_ctx_last_attempts(const selhip_ctx* c) { return c ? c->last_attempts : SELHIP_E_BADARG; }

int selhip_ctx_timing(selhip_ctx* c, int enable) {
    if (!c) return SELHIP_E_BADARG;
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (int t = 0; t < T_COUNT; ++t) { c->timers[t].total_ms = 0; c->timers[t].launches = 0; c->timers[t].span_ms = 0; }
    c->timing = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    c->timed_passes = 0;
    return SELHIP_OK;
}

double selhip_ctx_kernel_ms(const selhip_ctx* c, const char* name) {
    if (!c || !name) return -1.0;
    if (!c->pending) drain_timers(const_cast<selhip_ctx*>(c));
    const long passes = c->timed_passes;
    for (int t = 0; t < T_COUNT; ++t)
        if (!std::strcmp(name, kTimerNames[t]))
            return (c->timers[t].launches && passes) ? c->timers[t].total_ms / (double)passes : -1.0;
    if (!std::strcmp(name, "join_span"))
        return (c->timers[T_JOIN].launches && passes) ? c->timers[T_JOIN].span_ms / (double)passes : -1.0;
    return -1.0;
}

double selhip_ctx_kernel_launches(const selhip_ctx* c, const char* name) {
    if (!c || !name) return -1.0;
    if (!c->pending) drain_timers(const_cast<selhip_ctx*>(c));
    const long passes = c->timed_passes;
    for (int t = 0; t < T_COUNT; ++t)
        if (!std::strcmp(name, kTimerNames[t]))
            return passes ? (double)c->timers[t].launches / (double)passes : 0.0;
    return -1.0;
}

// ---- building blocks -------------------------------------------------------------------------
int selhip_smh_a_pairs(const uint64_t* d_aux, int m, int n_rows, int n_bands,
                       const selhip_int2_t* d_pairs, int64_t n_pairs, uint8_t* d_flags, void* hip_stream) {
    if (!d_aux || !d_pairs || !d_flags || n_pairs < 0 || m <= 0) return SELHIP_E_BADARG;
    if (n_rows <= 0 || n_bands <= 0 || (long long)n_rows * n_bands != m) return SELHIP_E_BADARG;
    if (n_pairs == 0) return SELHIP_OK;
    hipLaunchKernelGGL(pairlist_smh_kernel, dim3((unsigned)((n_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)hip_stream,
                       (const u64*)d_aux, m, n_rows, n_bands, d_pairs, (long long)n_pairs, (const double*)nullptr, 0.0, 0, 0,
                       d_flags, (selhip_int2_t*)nullptr, (u64)0, (u64*)nullptr);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

int selhip_hll_union_hist(const uint8_t* d_hll, int p, const selhip_int2_t* d_pairs, int64_t n_pairs,
                          uint32_t* d_counts, void* hip_stream) {
    if (!d_hll || !d_pairs || !d_counts || n_pairs < 0 || p < 4 || p > 20) return SELHIP_E_BADARG;
    if (n_pairs == 0) return SELHIP_OK;
    hipLaunchKernelGGL(hll_union_hist_kernel, dim3(grid_for((u64)n_pairs, kWavesPerBlock, 4096)), dim3(kBlock), 0,
                       (hipStream_t)hip_stream, d_hll, p, d_pairs, (const u64*)nullptr, (u64)n_pairs, (u64)n_pairs, d_counts);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

int selhip_hll_bitslice(const uint8_t* d_hll, int64_t n, uint32_t* d_planes, uint8_t* d_gmax, int* khi_out, void* hip_stream) {
    if (!d_hll || !d_planes || !d_gmax || !khi_out || n < 0) return SELHIP_E_BADARG;
    *khi_out = 1;
    if (n == 0) return SELHIP_OK;
    int* d_max = nullptr;
    HIPCHK(nullptr, hipMalloc((void**)&d_max, sizeof(int)));
    const int rc = build_bitslices(nullptr, (hipStream_t)hip_stream, d_hll, n, d_planes, d_gmax, d_max, khi_out);
    (void)hipFree(d_max);
    return rc;
}

int selhip_hll_union_hist_planes(const uint32_t* d_planes, const uint8_t* d_gmax, int khi, const selhip_int2_t* d_pairs, int64_t n_pairs,
                                 uint32_t* d_counts, void* hip_stream) {
    if (!d_planes || !d_gmax || !d_pairs || !d_counts || n_pairs < 0 || khi < 1 || khi > 64) return SELHIP_E_BADARG;
    if (n_pairs == 0) return SELHIP_OK;
    u64* d_n = nullptr;
    HIPCHK(nullptr, hipMalloc((void**)&d_n, sizeof(u64)));
    const u64 np = (u64)n_pairs;
    hipError_t e = hipMemcpyAsync(d_n, &np, sizeof(u64), hipMemcpyHostToDevice, (hipStream_t)hip_stream);
    if (e == hipSuccess) e = launch_hist_bs(khi, 2048, (hipStream_t)hip_stream, d_planes, d_gmax, d_pairs, d_n, np, d_counts, 0, ~0ull, 1);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)hip_stream);
    (void)hipFree(d_n);
    HIPCHK(nullptr, e);
    return SELHIP_OK;
}

int selhip_ertl_estimate(const uint32_t* d_counts, int64_t n, int p, int fp_mode, double* d_est, void* hip_stream) {
    if (!d_counts || !d_est || n < 0 || p < 4 || p > 20) return SELHIP_E_BADARG;
    if (n == 0) return SELHIP_OK;
    HIPCHK(nullptr, launch_select<0>(fp_mode == SELHIP_FP_FMA, (hipStream_t)hip_stream, grid_for((u64)n, kWave, 8192), d_counts,
                                     nullptr, (u64)n, (u64)n, p, d_est, nullptr, nullptr, 0.0, nullptr, 0, nullptr, nullptr, nullptr));
    return SELHIP_OK;
}

int selhip_smh_match_counts(const uint64_t* d_aux, int m, const selhip_int2_t* d_pairs, int64_t n_pairs,
                            int32_t* d_matches, void* hip_stream) {
    if (!d_aux || !d_pairs || !d_matches || n_pairs < 0 || m <= 0) return SELHIP_E_BADARG;
    if (n_pairs == 0) return SELHIP_OK;
    const long long threads = (long long)n_pairs * kWave;
    hipLaunchKernelGGL(match_count_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)hip_stream,
                       (const u64*)d_aux, m, d_pairs, (long long)n_pairs, d_matches);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

// ---- drop-in launchers (src/selection_kernels_wrapper.hpp:11-45) -------------------------------
}  // extern "C"

namespace {
struct CompatWs {
    std::mutex mu;
    DevBuf<selhip_int2_t> surv;
    DevBuf<uint32_t> counts;
    DevBuf<u64> surv_count;
    selhip_ctx* ctx[64] = {};          // implicit-enumeration path: one cached context per device (process lifetime)
};
CompatWs g_ws;
constexpr long long kCompatChunk = 1ll << 20;      // explicit pair lists are processed this many pairs at a time (256 MiB of histograms)

// select for the drop-in launchers: the reference signature carries no genome count, so the truncated
// cardinalities are taken on the fly from cards[rank]; output record = struct Result {x, y, (float)J}.
template <bool FMA, typename CountT>
__global__ __launch_bounds__(kWave)
void compat_select_kernel(const uint32_t* __restrict__ counts, const u64* __restrict__ n_dev, u64 cap, int p,
                          double relerr_scaled, const selhip_int2_t* __restrict__ pairs,
                          const double* __restrict__ cards, double tau,
                          selhip_result_t* __restrict__ out, CountT* __restrict__ out_count) {
    __shared__ uint32_t lds[64 * 65];
    const int lane = threadIdx.x;
    u64 n = *n_dev;
    if (n > cap) n = cap;
    for (u64 base = (u64)blockIdx.x * kWave; base < n; base += (u64)gridDim.x * kWave) {
        __syncthreads();
        for (int r = 0; r < kWave; ++r) {
            u64 j = base + r;
            lds[lane * 65 + r] = (j < n) ? counts[j * 64 + lane] : (lane == 0 ? (1u << p) : 0u);
        }
        __syncthreads();
        const u64 j = base + lane;
        LdsCounts c{lds + lane};
        double t = selhip::ertl_ml_estimate<FMA>(c, (unsigned)p, (unsigned)(64 - p), relerr_scaled);
        if (j < n) {
            const selhip_int2_t pr = pairs[j];
            const double e1 = (double)selhip::trunc_card(cards[pr.x]), e2 = (double)selhip::trunc_card(cards[pr.y]);
            const double jacc = (e1 + e2 - t) / t;                           // selection.cpp:287
            if (jacc >= tau) {                                               // selection.cpp:288
                const CountT idx = atomicAdd(out_count, (CountT)1);
                out[idx].x = pr.x; out[idx].y = pr.y; out[idx].sim = (float)jacc;
            }
        }
    }
}

// implicit enumeration: the context's result records {i, k, double J} -> struct Result {x, y, (float)J}, and the count
template <typename CountT>
__global__ __launch_bounds__(kBlock)
void compat_convert_kernel(const selhip_pair_t* __restrict__ res, u64 n, selhip_result_t* __restrict__ out, CountT* __restrict__ out_count) {
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < n; j += (u64)gridDim.x * kBlock) {
        const selhip_pair_t r = res[j];
        out[j].x = r.i; out[j].y = r.k; out[j].sim = (float)r.jaccard;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *out_count = (CountT)n;
}

template <typename CountT>
int compat_launch(bool use_cb, const uint8_t* main_sketches, const uint64_t* aux, const double* cards,
                  const selhip_int2_t* pairs, long long total_pairs, double tau, int m_aux, int m_hll,
                  int n_rows, int n_bands, selhip_result_t* out, CountT* out_count) {
    if (!main_sketches || !aux || !cards || !out || !out_count) { set_err(nullptr, "null pointer argument"); return SELHIP_E_BADARG; }
    if (total_pairs < 0 || m_aux <= 0 || !is_pow2(m_hll) || m_hll < 16) { set_err(nullptr, "bad sizes"); return SELHIP_E_BADARG; }
    if (n_rows <= 0 || n_bands <= 0 || (long long)n_rows * n_bands != m_aux) { set_err(nullptr, "n_rows*n_bands != m_aux"); return SELHIP_E_BADARG; }
    hipStream_t st = nullptr;                                        // default stream, like the reference
    HIPCHK(nullptr, hipMemsetAsync(out_count, 0, sizeof(CountT), st));  // selection_kernels.cu:137,166
    if (total_pairs == 0) return SELHIP_OK;
    const int p = ilog2(m_hll);
    std::lock_guard<std::mutex> lk(g_ws.mu);
    if (!pairs) {
        // pairs == NULL: the list is the implicit triangle i < k < n of the reference driver (selection_cuda.cpp:146-150), which is
        // never materialised -- total_pairs must be n(n-1)/2 and fixes n.  Runs through a cached context attached to the caller's
        // device arrays (all-pairs signature join + stage 2); this variant waits for the pass before it returns.
        const long long nn = (long long)((1.0 + std::sqrt(1.0 + 8.0 * (double)total_pairs)) / 2.0);
        long long n = nn;
        for (long long c = nn - 1; c <= nn + 1; ++c) if (c >= 2 && c * (c - 1) / 2 == total_pairs) n = c;
        if (n < 2 || n * (n - 1) / 2 != total_pairs) { set_err(nullptr, "pairs == NULL needs total_pairs = n(n-1)/2 (the whole triangle); got %lld", total_pairs); return SELHIP_E_BADARG; }
        int dev = 0;
        HIPCHK(nullptr, hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) { set_err(nullptr, "device index out of range"); return SELHIP_E_BADARG; }
        if (!g_ws.ctx[dev]) { int r = selhip_ctx_create(&g_ws.ctx[dev], dev); if (r) return r; }
        selhip_ctx* c = g_ws.ctx[dev];
        int r = selhip_ctx_attach(c, main_sketches, aux, cards, n, m_aux, p);
        if (!r) r = selhip_ctx_run(c, use_cb ? SELHIP_MODE_CB_SMH : SELHIP_MODE_SMH, SELHIP_ALGO_AUTO, (float)tau, n_rows, n_bands, 0, n);
        if (r) { set_err(nullptr, "%s", selhip_last_error(c)); return r; }
        const u64 cnt = c->last.n_results;
        if (sizeof(CountT) == 4 && cnt > 0x7FFFFFFFull) { set_err(nullptr, "more than 2^31 selected pairs: use the 64-bit launcher"); return SELHIP_E_OVERFLOW; }
        hipLaunchKernelGGL((compat_convert_kernel<CountT>), dim3(grid_for(cnt, kBlock, 4096)), dim3(kBlock), 0, c->stream, c->results.p, cnt, out, out_count);
        HIPCHK(nullptr, hipGetLastError());
        return SELHIP_OK;
    }
    const long long chunk = std::min(total_pairs, kCompatChunk);
    HIPCHK(nullptr, g_ws.surv.ensure((size_t)chunk));
    HIPCHK(nullptr, g_ws.counts.ensure((size_t)chunk * 64));
    HIPCHK(nullptr, g_ws.surv_count.ensure(1));
    for (long long off = 0; off < total_pairs; off += chunk) {
        const long long len = std::min(chunk, total_pairs - off);
        HIPCHK(nullptr, hipMemsetAsync(g_ws.surv_count.p, 0, sizeof(u64), st));
        hipLaunchKernelGGL(pairlist_smh_kernel, dim3((unsigned)((len + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           (const u64*)aux, m_aux, n_rows, n_bands, pairs + off, len, cards, tau, 1, use_cb ? 1 : 0,
                           (uint8_t*)nullptr, g_ws.surv.p, (u64)g_ws.surv.cap, g_ws.surv_count.p);
        HIPCHK(nullptr, hipGetLastError());
        hipLaunchKernelGGL(hll_union_hist_kernel, dim3(2048), dim3(kBlock), 0, st, main_sketches, p, g_ws.surv.p,
                           g_ws.surv_count.p, (u64)0, (u64)g_ws.surv.cap, g_ws.counts.p);
        HIPCHK(nullptr, hipGetLastError());
        hipLaunchKernelGGL((compat_select_kernel<true, CountT>), dim3(4096), dim3(kWave), 0, st,
                           g_ws.counts.p, g_ws.surv_count.p, (u64)g_ws.surv.cap, p, relerr_scaled_for(p), g_ws.surv.p,
                           cards, tau, out, out_count);
        HIPCHK(nullptr, hipGetLastError());
    }
    return SELHIP_OK;
}
}  // namespace

extern "C" {

int launch_kernel_smh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                      const selhip_int2_t* pairs, int total_pairs, double tau,
                      int m_aux, int m_hll, int n_rows, int n_bands,
                      selhip_result_t* out, int* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch<int>(false, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out, out_count);
}

int launch_kernel_CBsmh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                        const selhip_int2_t* pairs, int total_pairs, double tau,
                        int m_aux, int m_hll, int n_rows, int n_bands,
                        selhip_result_t* out, int* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch<int>(true, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out, out_count);
}

// 64-bit variants: the reference's `int total_pairs` / `int idx` (selection_kernels.cu:29-30) overflow beyond 2^31 pairs
// (N > 65 536 genomes); same parameter lists with int64_t total_pairs and a 64-bit *out_count.
int launch_kernel_smh64(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                        const selhip_int2_t* pairs, int64_t total_pairs, double tau,
                        int m_aux, int m_hll, int n_rows, int n_bands,
                        selhip_result_t* out, int64_t* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch<unsigned long long>(false, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out,
                                             reinterpret_cast<unsigned long long*>(out_count));
}

int launch_kernel_CBsmh64(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                          const selhip_int2_t* pairs, int64_t total_pairs, double tau,
                          int m_aux, int m_hll, int n_rows, int n_bands,
                          selhip_result_t* out, int64_t* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch<unsigned long long>(true, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out,
                                             reinterpret_cast<unsigned long long*>(out_count));
}

// (selhip_multi_select lives in selhip_multi.hip, selhip_ooc_select in selhip_ooc.hip: host-only translation units of this library
// built on the context API above; what they need beyond it is declared in selhip_internal.h and defined here)
void selhip_internal_set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int selhip_internal_reserve_replica(selhip_ctx* c, int64_t rows_padded, int m, int p_hll, int p_aux,
                                    uint8_t** d_hll, uint64_t** d_aux, double** d_cards, uint8_t** d_aux_hll) {
    if (!c || rows_padded < 0 || m <= 0 || p_hll < 4 || p_hll > 20 || !d_hll || !d_aux || !d_cards) return SELHIP_E_BADARG;
    HIPCHK(&c->err, hipSetDevice(c->device));
    const size_t rows = (size_t)std::max<int64_t>(rows_padded, 1);
    HIPCHK(&c->err, c->own_hll.ensure(rows << p_hll));
    HIPCHK(&c->err, c->own_aux.ensure(rows * (size_t)m));
    HIPCHK(&c->err, c->own_cards.ensure(rows));
    *d_hll = c->own_hll.p; *d_aux = (uint64_t*)c->own_aux.p; *d_cards = c->own_cards.p;
    if (p_aux > 0) {
        if (!d_aux_hll || p_aux > kMaxAuxP) return SELHIP_E_BADARG;
        HIPCHK(&c->err, c->own_aux_hll.ensure(rows << p_aux));
        *d_aux_hll = c->own_aux_hll.p;
    }
    return SELHIP_OK;
}

void* selhip_internal_stream(selhip_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ---- synthetic data ----------------------------------------------------------------------------
int selhip_synth_generate(const selhip_synth_t* sp_in, int64_t g_begin, int64_t g_end,
                          uint8_t* d_hll, uint64_t* d_aux, uint8_t* d_aux_hll, void* hip_stream) {
    if (!sp_in || !d_hll || !d_aux || g_begin < 0 || g_end < g_begin) return SELHIP_E_BADARG;
    if (!is_pow2(sp_in->m) || sp_in->m > 4096 || sp_in->cluster_size < 1 || sp_in->p_aux < 0 || sp_in->p_aux > 12) return SELHIP_E_BADARG;
    if (g_end == g_begin) return SELHIP_OK;
    selhip::SynthParams sp;
    sp.seed = sp_in->seed; sp.n_genomes = sp_in->n_genomes; sp.m = sp_in->m; sp.p_aux = sp_in->p_aux;
    sp.cluster_size = sp_in->cluster_size; sp.mode = sp_in->mode; sp.n_sh_lo = sp_in->n_sh_lo; sp.n_sh_hi = sp_in->n_sh_hi;
    const size_t smem = (size_t)sp.m * 8 + 16384 * 4 + (sp.p_aux ? ((size_t)4 << sp.p_aux) : 0);
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)synth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    HIPCHK(nullptr, attr_err);
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)(g_end - g_begin)), dim3(kBlock), smem, (hipStream_t)hip_stream,
                       sp, (long long)g_begin, (long long)g_end, d_hll, (u64*)d_aux, d_aux_hll);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}


int selhip_build_sketches(const uint8_t* d_codes, const int64_t* d_offsets, int64_t n_genomes, int k, int m, int p_aux,
                          uint8_t* d_hll, uint64_t* d_smh, uint8_t* d_aux_hll, void* hip_stream) {
    if (!d_codes || !d_offsets || !d_hll || n_genomes < 0 || k < 1 || k > 32) { set_err(nullptr, "bad argument"); return SELHIP_E_BADARG; }
    if (d_smh && (!is_pow2(m) || m > 2048)) { set_err(nullptr, "m must be a power of two <= 2048 (SizePow2Policy rounds up: pass the rounded value)"); return SELHIP_E_BADARG; }
    if (d_aux_hll && (p_aux < 4 || p_aux > 12)) { set_err(nullptr, "p_aux out of range [4,12]"); return SELHIP_E_BADARG; }
    if (n_genomes == 0) return SELHIP_OK;
    const size_t ms = d_smh ? (size_t)m : 0;
    const size_t smem = ms * 8 + 16384 + (d_aux_hll ? ((((size_t)1 << p_aux) + 3) / 4 * 4) : 0) + ms * 12 + 16;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] { attr_err = hipFuncSetAttribute((const void*)sketch_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    HIPCHK(nullptr, attr_err);
    // one block per genome; few genomes: 16 waves per block, so that a CU that holds a single block still has 4 waves per SIMD
    const unsigned threads = n_genomes < 2048 ? 1024u : (unsigned)kBlock;
    hipLaunchKernelGGL(sketch_build_kernel, dim3((unsigned)n_genomes), dim3(threads), smem, (hipStream_t)hip_stream,
                       d_codes, (const long long*)d_offsets, k, m, p_aux, d_hll, (u64*)d_smh, d_aux_hll);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

#ifdef SELHIP_JOIN_TRACE
int selhip_debug_join_trace(unsigned long long* h_out, int clear) {        // development build only (scripts/join_trace.py)
    if (clear) {
        void* p = nullptr;
        HIPCHK(nullptr, hipGetSymbolAddress(&p, HIP_SYMBOL(g_join_trace)));
        HIPCHK(nullptr, hipMemset(p, 0, sizeof(unsigned long long) * 4 * (1 << 17)));
        return SELHIP_OK;
    }
    HIPCHK(nullptr, hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_join_trace), sizeof(unsigned long long) * 4 * (1 << 17)));
    return SELHIP_OK;
}
#endif

int selhip_malloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return SELHIP_E_BADARG;
    HIPCHK(nullptr, hipMalloc(d_ptr, bytes ? bytes : 1));
    return SELHIP_OK;
}
int selhip_free(void* d_ptr) {
    if (d_ptr) HIPCHK(nullptr, hipFree(d_ptr));
    return SELHIP_OK;
}
int selhip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes) HIPCHK(nullptr, hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return SELHIP_OK;
}
int selhip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes) HIPCHK(nullptr, hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SELHIP_OK;
}
int selhip_device_synchronize(void) {
    HIPCHK(nullptr, hipDeviceSynchronize());
    return SELHIP_OK;
}

int selhip_permute_rows(const void* d_src, void* d_dst, const int32_t* d_perm, int64_t n_rows,
                        int64_t row_bytes, void* hip_stream) {
    if (!d_src || !d_dst || !d_perm || n_rows < 0 || row_bytes <= 0) return SELHIP_E_BADARG;
    if (n_rows == 0) return SELHIP_OK;
    const unsigned grid = (unsigned)std::min<int64_t>(n_rows, 65536);
    if (row_bytes % 16 == 0 && !((uintptr_t)d_src & 15) && !((uintptr_t)d_dst & 15))
        hipLaunchKernelGGL(permute_rows_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)hip_stream,
                           (const uint4*)d_src, (uint4*)d_dst, d_perm, (long long)n_rows, (long long)(row_bytes / 16));
    else
        hipLaunchKernelGGL(permute_bytes_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)hip_stream,
                           (const uint8_t*)d_src, (uint8_t*)d_dst, d_perm, (long long)n_rows, (long long)row_bytes);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

}  // extern "C"
