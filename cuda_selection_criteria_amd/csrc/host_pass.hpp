// host_pass.hpp -- the pass scheduler of libselhip.so: kernel dispatch of stage 1 (stream / signature join / sort join), the
// auxiliary criteria, grouping, stage 2 (union histograms on bit planes or byte rows, estimator), chunk lanes, scratch sizing.
// Part of the kernel translation unit selection_kernels.hip (included there, after host_context.hpp); not a stand-alone header.
#pragma once

namespace {

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
    const int tg = c->sig_tile_g;                                       // genomes per tile
    const unsigned work_blocks = cached ? 0u : tile_mode ? (unsigned)((n + tg - 1) / tg) : (unsigned)((threads + kBlock - 1) / kBlock);
    c->sig_key = c->sig_cache ? sig_key : 0;
    if (work_blocks + (unsigned)bounds_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(sig_build_kernel, dim3(work_blocks + (unsigned)bounds_blocks), dim3(kBlock), 0, c->stream,
                       c->d_aux, n, c->m, n_rows, n_bands, n_pad, c->sigQ.p, c->sigT.p, c->sigP.p, c->sigG.p,
                       bounds_blocks, c->d_cards, tau, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, row_map(c, rb, re), c->ecard.p, c->hi.p, c->pcb,
                       grouping_on(c) ? c->csr_cnt.p : nullptr, grouping_on(c) ? (int)c->csr_cnt.cap : 0, (int)c->cand_begin,
                       with_bounds ? c->seg_cnt.p : nullptr, with_bounds ? (int)c->seg_cnt.cap : 0, c->join_bits == 15 ? 17 : 16,
                       zero_pc, tile_mode ? tg : 0);
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
                          u64 cap, uint32_t* counts, u64 off, u64 window, int run, u64 dense_pairs) {
#define SELHIP_BS_LAUNCH(NB) hipLaunchKernelGGL((hll_union_hist_bs_kernel<NB>), dim3(blocks), dim3(kBlock), 0, st, bs, gmax, list, count, cap, counts, off, window, run, dense_pairs)
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
// rows after which the deal of the row interleave repeats: two cycles of n_parts blocks (the snake turns round every cycle)
long long interleave_period(const selhip_ctx* c) { return c->il_parts > 1 ? 2ll * c->il_block * c->il_parts : 1; }

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
    if (!sig || !grouping_on(c)) return 1;
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
    if (!grouping_on(c)) return false;
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
    const bool grouped = grouping_on(c);
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
                                               c->hist_run > 0 ? c->hist_run : (grouped ? 4 : 1),
                                               // a dense survivor graph is walked by candidate-row slice per XCD (query-major list only)
                                               // ("dense" = survivors per QUERY ROW of this chain: a rank's or a lane's share of the rows sees
                                               //  its share of the pairs and all of the candidate rows)
                                               grouped && c->hist_dense_degree >= 0
                                                   ? (u64)c->hist_dense_degree * (u64)std::max<long long>(1, ((long long)ch.re - ch.rb) / std::max(1, c->il_parts)) : ~0ull));
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

// ---- the whole pass of a small set in one cooperative launch (kernel_small.cuh) ------------------------------------------------
bool small_pass_ok(const selhip_ctx* c) {
    if (c->small_pass == 0 || c->small_pass_failed) return false;
    if (c->n < 2 || c->n > kSmallPassMaxN || c->criterion != SELHIP_CRIT_SMH_A || c->il_parts > 1) return false;
    if (!(c->algo == SELHIP_ALGO_AUTO || c->algo == SELHIP_ALGO_SIG) || !sig_supported(c->m, c->n_rows, c->n_bands)) return false;
    if (!(is_pow2(c->m) && c->n_rows >= 2 && c->n_rows <= 32 && c->m >= 4)) return false;          // the tiled signature build
    if ((c->row_end - c->row_begin + 255) / 256 > kSmallRows) return false;                        // rows per block of the 256-block grid
    if (c->dev_cus < 256) return false;                                                            // (a partitioned or masked device: the regular pass)
    return use_bitslices(c) && c->hll_khi <= 32;                                                   // bit planes, at most five of them non-zero
}

hipError_t hipModuleLaunchKernel_like(selhip_ctx* c, const void* fn, unsigned grid, void** args) {
    return hipLaunchKernel(fn, dim3(grid), dim3(kBlock), args, 0, c->stream);
}

template <bool FMA, int NB>
hipError_t launch_small_pass(selhip_ctx* c, PassCounters* pc_next, double tau) {
    int n = (int)c->n;
    int n_pad = ((n + kWave - 1) / kWave) * kWave;
    RowMap rm = row_map(c, (int)c->row_begin, (int)c->row_end);
    int m = c->m, r = c->n_rows, nb = c->n_bands, use_cb = c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, cand_begin = (int)c->cand_begin, fb = c->verify_fb;
    const u64* aux = c->d_aux; const double* cards = c->d_cards; const uint32_t* bs = c->hll_bs.p; const uint8_t* gmax = c->hll_gmax.p;
    uint32_t *sQ = c->sigQ.p, *sT = c->sigT.p, *sP = c->sigP.p, *sG = c->sigG.p;
    u64* ecard = c->ecard.p; int* hi = c->hi.p; PassCounters* pc = c->pcb;
    u64* barrier_word = &c->pcb[kMaxChunks].n_aux_in;                   // a word of this pass's counter set that nothing else uses (cleared by the previous pass)
    if (!c->small_bar.p) {                                               // the barrier's group words: zero between passes (the kernel puts them back)
        hipError_t e = c->small_bar.ensure((size_t)kSmallBarGroups * kSmallBarStride);
        if (e == hipSuccess) e = hipMemsetAsync(c->small_bar.p, 0, sizeof(u64) * kSmallBarGroups * kSmallBarStride, c->stream);
        if (e != hipSuccess) return e;
    }
    u64* barrier_groups = c->small_bar.p;
    u64 barrier_ticks = c->small_pass == 3 ? 0 : 2000000;                // 20 ms of the 100 MHz wall clock ("small_pass" = 3, test hook: no wait at all)
    double rs = relerr_scaled_for(14);
    selhip_pair_t* results = c->results.p; u64 results_cap = (u64)c->results.cap;
    void* args[] = {&aux, &cards, &bs, &gmax, &n, &m, &r, &nb, &n_pad, &tau, &use_cb, &rm, &cand_begin, &sQ, &sT, &sP, &sG, &ecard, &hi, &pc, &pc_next,
                    &barrier_word, &barrier_groups, &barrier_ticks, &rs, &results, &results_cap, &fb};
    const unsigned grid = 256;                                           // one block per CU (small_pass_ok checked that the device has 256)
    // an ordinary launch: 256 blocks of 256 threads with 60 KB of LDS and <= 250 registers -- a CU holds two, the device 512 -- all become
    // resident as soon as whatever else is running drains, which is all the kernel's one barrier needs (work of the same stream is over
    // by then; nothing another stream runs waits for this kernel); hipLaunchCooperativeKernel ("small_pass" = 2) asks the runtime for
    // that guarantee and costs ~15 us more per launch
    if (c->small_pass == 2) return hipLaunchCooperativeKernel((const void*)small_pass_kernel<FMA, NB>, dim3(grid), dim3(kBlock), args, 0, c->stream);
    return hipModuleLaunchKernel_like(c, (const void*)small_pass_kernel<FMA, NB>, grid, args);
}

int enqueue_small_pass(selhip_ctx* c, PassCounters* pc_next, double tau) {
    TimerScope t(c, T_STAGE1);
    const bool fma = c->fp_mode == SELHIP_FP_FMA;
    c->sig_key = 0;                                                      // (the kernel rewrites the 32-bit signature layouts only)
    hipError_t e;
    if (c->hll_khi <= 16) e = fma ? launch_small_pass<true, 4>(c, pc_next, tau) : launch_small_pass<false, 4>(c, pc_next, tau);
    else                  e = fma ? launch_small_pass<true, 5>(c, pc_next, tau) : launch_small_pass<false, 5>(c, pc_next, tau);
    HIPCHK(&c->err, e);
    return SELHIP_OK;
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
    c->small_used = small_pass_ok(c);
    if (c->small_used) {
        c->n_chunks_last = 1;
        const int rc = enqueue_small_pass(c, pc_next, tau);
        if (rc) return rc;
        HIPCHK(&c->err, hipMemcpyAsync(c->h_pc, c->pcb, sizeof(PassCounters) * (kMaxChunks + 1), hipMemcpyDeviceToHost, c->stream));
        c->pc_dirty = false;
        return SELHIP_OK;
    }
    if (use_sig) {
        // bounds (truncated cards, CB cut-offs, z0, evaluated count) ride in the first blocks of the signature build
        HIPCHK(&c->err, launch_sig_build(c, c->n_rows, c->n_bands, true, tau, rb, re, pc_next));
    } else {
        TimerScope t(c, T_PREP);
        hipLaunchKernelGGL(cb_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_cards, n, tau, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, row_map(c, rb, re), c->ecard.p, c->hi.p, pc0,
                           grouping_on(c) ? c->csr_cnt.p : nullptr, grouping_on(c) ? (int)c->csr_cnt.cap : 0, (int)c->cand_begin, pc_next,
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
    const bool count_in_verify = crit == SELHIP_CRIT_SMH_A && use_sig && !use_hash && c->join_bits <= 16 && grouping_on(c);
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
        chunk_rows(n, rb, re, chunks, interleave_period(c), bnd);
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
        const long long period = interleave_period(c);
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
        const long long period = interleave_period(c);
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
    if (grouping_on(c)) {
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
