// host_context.hpp -- the context of libselhip.so (struct selhip_ctx), its device buffers, kernel timers and small helpers.
// Part of the kernel translation unit selection_kernels.hip (included there, after the kernel headers); not a stand-alone header.
#pragma once

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
// pairs per pass from which the automatic setting splits a pass into two chunk lanes (one MI355X, cfg3 data at growing genome counts,
// lanes off -> 2: 4.0e8 pairs 1.033 -> 1.034 ms, 5.4e8 1.319 -> 1.289, 7.2e8 1.648 -> 1.622, 1.0e9 2.290 -> 2.154; an eighth of cfg4's rows,
// 6.2e8 pairs, 0.446 -> 0.431 ms)
constexpr double kAutoChunkPairs = 5e8;
static_assert(kCounterBlocks == kMaxChunks + 1, "common.cuh: counter blocks per pass");
constexpr int kMaxAuxP = SELHIP_MAX_AUX_P;   // auxiliary HLL precision accepted by every entry point: aux_fused_kernel counts in 16-bit bins (a bin holds up to 2^p_aux)
constexpr long long kEnumPairs = 1ll << 26;    // hll_a / hll_an as first criterion: pairs listed per sub-pass (512 MiB of int2)

enum { T_PREP = 0, T_STAGE1, T_HIST, T_SELECT, T_TOTAL, T_SIGBUILD, T_JOIN, T_VERIFY, T_AUX, T_GROUP, T_COUNT };
const char* kTimerNames[T_COUNT] = {"prep", "stage1", "hist", "select", "total", "sigbuild", "join", "verify", "aux", "group"};

}  // namespace

struct selhip_ctx {
    int device = 0;
    int dev_cus = 0;                    // compute units of the device (the one-launch pass of a small set needs all 256 of an MI355X)
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
    DevBuf<u64> small_bar;              // the one-launch pass's barrier: 16 group words, 128 bytes apart
    DevBuf<uint8_t> hll_gmax;           // [n] largest register value of each genome
    DevBuf<int> hll_bs_max;             // largest register value of the set (device side)
    int hll_khi = 0;                    // 0 = no planes; else max register value + 1
    int hist_algo = -1;                 // -1 automatic (bit planes when p = 14), 0 = byte rows + LDS histogram (hll_union_hist_runs_kernel), 1 = bit planes
    int hist_dense_degree = 32;         // bit-plane kernel: survivors per query row from which a grouped list is walked by candidate slice per XCD (-1 = never)
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
    // genomes per tile of the tiled signature build ("sig_tile_g": 8 / 16 / 32).  Measured, build alone, 8 / 16 / 32: cfg3 15.9 / 16.4 / 21.3 us,
    // 28 280 genomes 36.9 / 35.0 / 42.2, cfg4 64.5 / 60.4 / 67.2 (gpurun_out/r03/o_*): full 128-byte band-major segments (32) do not pay
    int sig_tile_g = 16;
    int sig_cache = 0;                  // keep the band signatures across passes ("sig_cache"); sig_key = what the arrays hold (0 = nothing)
    long long sig_key = 0;
    int sig_tile = 1;                   // signature build: tiled form (0 = one thread per bucket, the round-1 kernel)
    int init_cap = 0;                   // test hook: initial capacity of the survivor / candidate lists (0 = sized from the workload)
    int join_qt = 0;                    // query rows per signature-join block (multiple of 16); 0 = automatic: 32 rows below 1e8 pairs per pass, 64 up to 4.5e8
                                        // (30 000 genomes on one GPU; 2e8 since round 3), 128 beyond.  With the segmented appends: cfg3 112 / 114 / 127 us at 64 / 96 / 128 rows (finer tiles balance
                                        // the 1 024 SIMDs better), cfg4 2.12 / 2.10 / 2.07 ms (a block's prologue -- 32 candidate loads per lane,
                                        // tile staging -- is amortised over more rows), cfg5 8.31 / 8.16 / 8.20 ms
    bool group_stage2 = true;           // bucket survivors by query row before stage 2a (hll_union_hist_runs_kernel)
    int small_pass = -1;                // sets of <= 2 048 genomes: the whole pass in one cooperative launch (kernel_small.cuh); -1 automatic, 0 off
    bool small_used = false, small_pass_failed = false;   // the last enqueue took it / a block's survivor list overflowed once: regular passes from then on
    int group_min_n = 2048;             // ... for sets of more than this many genomes ("group_min_n"; see grouping_on)

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

// timing level 1: every scope; level 2: only the dominant kernel of the pass (an event pair costs ~5 us of stream time, and a pass of
// the default workload is 0.24 ms).  Handing the pair to the launch itself (hipExtLaunchKernelGGL: the dispatch's own start / end
// timestamps) was built and measured in round 3: same step (0.2427 / 0.2404 against 0.2399 / 0.2414 ms), same kernel figure
// (103.9 against 103.3 us) -- the runtime brackets the dispatch with the same packets either way (gpurun_out/r03/t_*).  Not kept.
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

// stage 2 works on a list bucketed by query row (and laid out by label) unless the caller switched that off -- or the set is small:
// up to group_min_n genomes (2 048) the whole table of bit planes (<= 20 MB) stays in L2 / the Infinity Cache whatever the order, and
// the two grouping launches are 13 us of a 95 us step (BASELINE configs[1])
bool grouping_on(const selhip_ctx* c) { return c->p == 14 && c->group_stage2 && c->n > c->group_min_n; }

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

}  // namespace
