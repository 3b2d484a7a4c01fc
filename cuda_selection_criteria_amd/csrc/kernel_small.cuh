// kernel_small.cuh -- small_pass_kernel: a whole pass over a SMALL set (<= 2 048 genomes; BASELINE configs[1]) in ONE launch.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
//
// At 1 000 genomes the regular pass is five or six dependent launches of 8-20 us each, every one a chain of two to six memory round
// trips on a handful of waves (profiles/r03cfg2_*: 0.09 ms for 499 500 pairs, 0.003-1.3 waves per SIMD): what the step costs is launches and
// latency, not work.  Here the chain is one kernel with ONE grid-wide barrier:
//   phase 0  every block: its share of the bounds (cb_bounds_body), of the signature tiles (sig_build_tile_body) and of the counter clearing;
//   -------- grid barrier (two levels of arrival counters + a spin: 256 blocks, one per CU, all resident) --------
//   phase 1  block b owns up to eight query rows (dealt boustrophedon, so the triangle's work is even): lane = candidate, "some band's
//            32-bit signature equal" over the band-major signatures, matches queued in LDS and verified sixteen lanes to a pair on the
//            flagged band (the literal smh_a after a 32-bit collision: rare), survivors into the block's LDS list;
//   phase 2  the block's survivors, 64 at a time: a wave per pair builds the union histogram from the bit planes (bs_pair_hist, the code
//            of stage 2a) into an LDS tile, then one wave runs the estimator with a lane per pair and appends the selected ones.
// Nothing after the barrier leaves the block, so the stages of different blocks overlap freely.  Same results as the regular pass
// (pairs, Jaccard bits, evaluated / candidate / survivor counters); criterion smh_a, contiguous rows, band shapes of the signature join,
// bit planes with at most five non-zero planes -- anything else takes the regular pass.
#pragma once

namespace {

constexpr int kSmallPassMaxN = 2048;
constexpr int kSmallRows = 8;                // query rows per block: 2 048 genomes over 256 blocks
constexpr int kSmallListCap = 2048;          // survivors a block can hold (8 rows x up to 2 047 candidates: the host falls back if it overflows)
constexpr uint32_t kSmallNoBand = 255u;      // "no band of this candidate matched the row yet" (bands are numbered below 128)
constexpr int kSmallBands = 16;              // bands of a candidate's signature a lane requests together
constexpr int kSmallChunks = 4;              // chunks of 256 candidates a lane has in flight in the join
constexpr int kSmallQueueCap = kSmallChunks * 256 * kSmallRows;      // flagged candidates awaiting verification: the worst case of one round
static_assert(kSmallQueueCap >= 64 * 65, "the histogram tile of phase 2 reuses the queue's LDS");
constexpr u64 kSmallOverflow = ~0ull;        // published in n_pre_segmax of counter block 0 when a block's list overflowed or its barrier wait ran out

// The grid barrier, two levels: 256 arrivals on ONE word are served one after the other by the memory side (9 us between the last arrival
// and the release, measured); here a block arrives on its group's word (16 groups, 128 bytes apart), a group's last arrival puts the
// word back to zero -- ready for the next pass -- and arrives on the top word, which everyone polls.
// The wait is BOUNDED: an ordinary launch does not promise that all 256 blocks are resident together -- three such kernels of three
// contexts dispatched at once could each hold a part of the device's 512 block slots and wait for the rest for ever.  A block that has
// waited `max_ticks` (100 MHz wall clock; 20 ms from the host) gives up: it returns false, the kernel flags the pass for the regular path
// (as for a list overflow) and ends; every block still arrives exactly once, so the words are left as the next pass needs them.
constexpr int kSmallBarGroups = 16, kSmallBarStride = 16;                    // (u64 words)
__device__ __forceinline__ bool small_grid_barrier(u64* top, u64* groups, unsigned n_blocks, u64 max_ticks) {
    __shared__ int released;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned grp = blockIdx.x % kSmallBarGroups;
        const unsigned members = (n_blocks - 1u - grp) / kSmallBarGroups + 1u;          // blocks b with b % 16 == grp (n_blocks >= 16)
        u64* gw = groups + grp * kSmallBarStride;
        if (atomicAdd(gw, 1ull) == (u64)(members - 1u)) {
            atomicExch(gw, 0ull);
            atomicAdd(top, 1ull);
        }
        const u64 t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (u64)kSmallBarGroups) {
            if (wall_clock64() - t0 >= max_ticks) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        released = ok;
        __threadfence();
    }
    __syncthreads();
    return released != 0;
}

#ifdef SELHIP_JOIN_TRACE
#define SELHIP_SMALL_STAMP(K) do { if (threadIdx.x == 0) g_join_trace[blockIdx.x * 3 + (K) / 4][(K) % 4] = wall_clock64(); } while (0)
#else
#define SELHIP_SMALL_STAMP(K) do { } while (0)
#endif

template <bool FMA, int NB>
__global__ __launch_bounds__(kBlock)
void small_pass_kernel(const u64* __restrict__ aux, const double* __restrict__ cards, const uint32_t* __restrict__ bs, const uint8_t* __restrict__ gmax,
                       int n, int m, int r, int nb, int n_pad, double tau, int use_cb, RowMap rm, int cand_begin,
                       uint32_t* __restrict__ sigQ, uint32_t* __restrict__ sigT, uint32_t* __restrict__ sigP, uint32_t* __restrict__ sigG,
                       u64* __restrict__ ecard, int* __restrict__ hi, PassCounters* __restrict__ pc, PassCounters* __restrict__ zero_pc,
                       u64* __restrict__ barrier_word, u64* __restrict__ barrier_groups, u64 barrier_ticks, double relerr_scaled, selhip_pair_t* __restrict__ results, u64 results_cap, int force_fallback) {
    __shared__ selhip_int2_t list_lds[kSmallListCap];
    __shared__ uint32_t scratch_lds[kSmallQueueCap];                         // phase 1: flagged (row, band, candidate) triples awaiting
    uint32_t* const counts_lds = scratch_lds;                                // verification; phase 2: 64 pairs' histograms (64 x 65 words)
    __shared__ uint32_t q_lds[kSmallRows * 128];                             // the block's query rows' 32-bit band signatures
    __shared__ int hi_lds[kSmallRows], row_lds[kSmallRows];
    __shared__ int n_list, n_cand_blk, n_queue;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int G = (int)gridDim.x;

    SELHIP_SMALL_STAMP(0);
    // ---- phase 0: bounds, signatures, next pass's counters
    {
        const int t = (int)(blockIdx.x * kBlock + threadIdx.x);
        zero_next_counters(t, G * kBlock, zero_pc, kCounterBlocks);
        // (the rows' pair counts are summed per wave first: one atomic per wave instead of one per row on the same address)
        long long ev = 0;
        for (int i = t; i < n; i += G * kBlock) ev += cb_bounds_body(i, cards, n, tau, use_cb, rm, ecard, hi, pc, cand_begin, true);
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) ev += __shfl_xor(ev, sft, kWave);
        if (lane == 0 && ev > 0) atomicAdd(&pc->n_evaluated, (u64)ev);
        // the signatures, every block its (n + G - 1) / G genomes (<= 8): one load round trip instead of the four a 16-genome tile takes
        const int tg = (n + G - 1) / G;
        sig_build_tile_body<kSmallRows>((int)blockIdx.x, aux, n, m, r, nb, n_pad, sigQ, sigT, sigP, sigG, 16, tg, false);
        if (threadIdx.x == 0) { n_list = 0; n_cand_blk = 0; n_queue = 0; }
    }
    SELHIP_SMALL_STAMP(1);
    if (!small_grid_barrier(barrier_word, barrier_groups, (unsigned)G, barrier_ticks)) {
        if (threadIdx.x == 0) pc->n_pre_segmax = kSmallOverflow;             // the host repeats the pass on the regular path
        return;
    }
    SELHIP_SMALL_STAMP(2);

    // ---- phase 1: the block's (up to kSmallRows) query rows against every candidate.  Rows are dealt to the blocks boustrophedon
    // (row slot j G + b for even j, j G + G-1-b for odd j), so every block meets about the same number of candidates although row i only
    // has n-1-i of them.  The rows' signatures sit in LDS; kSmallChunks x 256 candidates read kSmallBands bands of their band-major signatures at
    // a time (64 coalesced loads in flight per lane: the loads' round trips, not their bytes, are what this phase costs) and meet all the
    // rows; per row a lane keeps the first band whose 32-bit signatures agree and queues (row, band, candidate) in LDS.  The queue is
    // then verified sixteen lanes to a candidate: the flagged band alone is compared on the full sketches, one bucket per lane, so a
    // block's few dozen candidates cost ONE round trip.  The literal lane-serial smh_a only decides after a 32-bit collision, as in
    // verify16_kernel
    const int z0 = pc->z0p1 ? pc->z0p1 - 1 : n;
    const int rows_total = rm.row_end - rm.row_begin;
    auto row_of = [&](int j) {                                               // ascending in j; -1 past the block's last row
        const int slot = j * G + ((j & 1) ? G - 1 - (int)blockIdx.x : (int)blockIdx.x);
        return slot < rows_total ? rm.row_begin + slot : -1;
    };
    int nr = 0;
#pragma unroll
    for (int j = 0; j < kSmallRows; ++j) nr += row_of(j) >= 0 ? 1 : 0;
    // the rows' signatures and cut-offs are REQUESTED here and parked in LDS only after the first round of candidate loads has been
    // issued (the first pass through the loop below): one round trip for both
    constexpr int kQPerLane = kSmallRows * 128 / kBlock;
    uint32_t qreg[kQPerLane];
#pragma unroll
    for (int u = 0; u < kQPerLane; ++u) {
        const int t = (int)threadIdx.x + u * kBlock;
        qreg[u] = t < nr * nb ? sigQ[(size_t)row_of(t / nb) * nb + (t % nb)] : 0u;
    }
    int hreg = -1;
    if ((int)threadIdx.x < nr) hreg = hi[row_of((int)threadIdx.x)];
    bool parked = false;
    const __amdgpu_buffer_rsrc_t sigT_rsrc = __builtin_amdgcn_make_buffer_rsrc(sigT, 0, nb * n_pad * 4, 0x00020000);
    const int row_bytes = n_pad * 4;
    if (nr > 0) {
        const int k_first = (max(row_of(0) + 1, z0) / kBlock) * kBlock;
        for (int k0 = k_first; k0 < n; k0 += kSmallChunks * kBlock) {
            // (first matching band = the smallest: a compare, a select and an unsigned minimum per (row, band, candidate), all on vector
            //  registers -- written as "not found yet && equal" the conditions were kept as 64-bit lane masks, 32 of them alive at once,
            //  and the scalar registers spilled: 9 us of compares)
            uint32_t fband[kSmallChunks][kSmallRows];
#pragma unroll
            for (int c = 0; c < kSmallChunks; ++c)
#pragma unroll
                for (int ri = 0; ri < kSmallRows; ++ri) fband[c][ri] = kSmallNoBand;
            for (int b0 = 0; b0 < nb; b0 += kSmallBands) {                    // (nb is 8 or a multiple of 16: sig_supported)
                // (through a buffer resource: the band's row rides in the scalar offset, the lane's column in the 32-bit vector offset --
                //  guarded 64-bit addressing put two branches and a 64-bit multiply around every one of these 64 loads.  Columns past
                //  the last genome and bands past the last band are clamped: such a value is never compared (bands) or never a hit
                //  (columns: `k < n` below))
                uint32_t cv[kSmallChunks][kSmallBands];
#pragma unroll
                for (int c = 0; c < kSmallChunks; ++c) {
                    const int kk = min(k0 + c * kBlock + (int)threadIdx.x, n_pad - 1) * 4;
#pragma unroll
                    for (int u = 0; u < kSmallBands; ++u)
                        cv[c][u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(sigT_rsrc, kk, min(b0 + u, nb - 1) * row_bytes, 0);
                }
                if (!parked) {                                                // block-uniform
                    parked = true;
#pragma unroll
                    for (int u = 0; u < kQPerLane; ++u) q_lds[(int)threadIdx.x + u * kBlock] = qreg[u];
                    if ((int)threadIdx.x < kSmallRows) { row_lds[threadIdx.x] = row_of((int)threadIdx.x); hi_lds[threadIdx.x] = hreg; }
                    __syncthreads();
                    SELHIP_SMALL_STAMP(7);
                }
#pragma unroll
                for (int u = 0; u < kSmallBands; ++u)
#pragma unroll
                    for (int ri = 0; ri < kSmallRows; ++ri) {
                        if (ri >= nr || b0 + u >= nb) break;                  // block-uniform
                        const uint32_t qv = q_lds[ri * nb + b0 + u];
#pragma unroll
                        for (int c = 0; c < kSmallChunks; ++c)
                            fband[c][ri] = min(fband[c][ri], cv[c][u] == qv ? (uint32_t)(b0 + u) : kSmallNoBand);
                    }
            }
#pragma unroll
            for (int c = 0; c < kSmallChunks; ++c) {
                const int k = k0 + c * kBlock + (int)threadIdx.x;
                if (k0 + c * kBlock >= n) break;                              // block-uniform
#pragma unroll
                for (int ri = 0; ri < kSmallRows; ++ri) {
                    if (ri >= nr) break;                                      // block-uniform
                    const int i = row_lds[ri];
                    const bool hit = k < n && fband[c][ri] != kSmallNoBand && k >= max(i + 1, z0) && k <= min(hi_lds[ri], n - 1);
                    const u64 hm = __ballot(hit);
                    if (hm) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&n_queue, (int)__popcll(hm));
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (hit) scratch_lds[base + (int)__popcll(hm & ((1ull << lane) - 1ull))] =
                                     ((uint32_t)ri << 18) | (fband[c][ri] << 11) | (uint32_t)k;
                    }
                }
            }
            SELHIP_SMALL_STAMP(8);
            __syncthreads();
            SELHIP_SMALL_STAMP(9);
            const int nq = n_queue;                                           // <= kSmallChunks x 256 x kSmallRows = kSmallQueueCap
            for (int q0 = 0; q0 < nq; q0 += kBlock / 16) {
                const int q = q0 + (int)threadIdx.x / 16, sub = (int)threadIdx.x & 15;
                const bool live = q < nq;
                const uint32_t e = live ? scratch_lds[q] : 0u;
                const int ri = (int)(e >> 18), fb = (int)((e >> 11) & 127u), kk = (int)(e & 2047u);
                const int i = row_lds[ri];
                bool eq = true;
                if (live) {
                    const u64* x = aux + (long long)i * m + (long long)fb * r;
                    const u64* y = aux + (long long)kk * m + (long long)fb * r;
                    for (int j2 = sub; j2 < r; j2 += 16) eq &= x[j2] == y[j2];
                }
                const u64 em = __ballot(eq);
                bool ok = live && !force_fallback && (uint32_t)((em >> (lane & 48)) & 0xFFFFull) == 0xFFFFu;
                // a 32-bit collision (or the forced fallback): the literal predicate decides
                if (live && sub == 0 && !ok) ok = smh_a_lane(aux + (long long)i * m, aux + (long long)kk * m, r, nb);
                ok = ok && sub == 0;
                const u64 om = __ballot(ok);
                if (om) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&n_list, (int)__popcll(om));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (ok) {
                        const int pos = base + (int)__popcll(om & ((1ull << lane) - 1ull));
                        if (pos < kSmallListCap) list_lds[pos] = selhip_int2_t{i, kk};
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) { n_cand_blk += nq; n_queue = 0; }
            __syncthreads();
        }
    }
    __syncthreads();
    SELHIP_SMALL_STAMP(3);
    const int n_surv = n_list;
    if (threadIdx.x == 0) {
        if (n_surv) atomicAdd(&pc[1].n_survivors, (u64)n_surv);              // (counter block 1 = the pass's only chain, as in the regular pass)
        if (n_cand_blk) atomicAdd(&pc[1].n_candidates, (u64)n_cand_blk);
        if (n_surv > kSmallListCap) pc->n_pre_segmax = kSmallOverflow;       // the host repeats the pass on the regular path
    }
    const int n_mine = min(n_surv, kSmallListCap);

    // ---- phase 2: union histograms (a wave per pair, from the bit planes) and the estimator (a lane per pair), 64 pairs at a time
    const int my_bin = ((lane & 2) ? 32 : 0) + 2 * bs_pidx(lane) + (lane & 1);
    for (int base = 0; base < n_mine; base += kWave) {
        const int batch = min(kWave, n_mine - base);
        // (a wave's pairs one after the other, the NEXT pair's planes requested before the current one is decoded: one wave per SIMD here,
        //  nothing else hides a memory round trip, and registers are plentiful)
        uint32_t xa[NB][8], yb[NB][8], xn[NB][8], yn[NB][8];
        int kp = 0, kpn = 0;
        if (wave < batch) {
            const selhip_int2_t pr = list_lds[base + wave];
            const int px = __builtin_amdgcn_readfirstlane(pr.x), py = __builtin_amdgcn_readfirstlane(pr.y);
            bs_load<NB>(bs, px, lane, xa); bs_load<NB>(bs, py, lane, yb);
            kp = max((int)gmax[px], (int)gmax[py]) + 1;
        }
        for (int p = wave; p < batch; p += kWavesPerBlock) {
            const bool more = p + kWavesPerBlock < batch;
            if (more) {
                const selhip_int2_t pr = list_lds[base + p + kWavesPerBlock];
                const int px = __builtin_amdgcn_readfirstlane(pr.x), py = __builtin_amdgcn_readfirstlane(pr.y);
                bs_load<NB>(bs, px, lane, xn); bs_load<NB>(bs, py, lane, yn);
                kpn = max((int)gmax[px], (int)gmax[py]) + 1;
            }
            const uint32_t tot = bs_pair_hist<NB>(xa, yb, __builtin_amdgcn_readfirstlane(kp), lane);
            counts_lds[my_bin * 65 + p] = (lane & 1) ? (tot >> 16) : (tot & 0xFFFFu);
            if (more) {
#pragma unroll
                for (int b2 = 0; b2 < NB; ++b2)
#pragma unroll
                    for (int c2 = 0; c2 < 8; ++c2) { xa[b2][c2] = xn[b2][c2]; yb[b2][c2] = yn[b2][c2]; }
                kp = kpn;
            }
        }
        __syncthreads();
        SELHIP_SMALL_STAMP(4);
        if (wave == 0) {
            const bool live = lane < batch;
            selhip_int2_t pr{0, 0};
            u64 e1 = 0, e2 = 0;
            if (live) { pr = list_lds[base + lane]; e1 = ecard[pr.x]; e2 = ecard[pr.y]; }
            else {                                                            // an empty sketch for the idle lanes: estimate 0, result unused
                for (int kq = 0; kq < 64; ++kq) counts_lds[kq * 65 + lane] = kq == 0 ? 16384u : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            LdsCounts c{counts_lds + lane};
            const double t = selhip::ertl_ml_estimate<FMA>(c, 14u, 50u, relerr_scaled);
            const double jacc = ((double)e1 + (double)e2 - t) / t;            // selection.cpp:287
            const bool keep = live && jacc >= tau;                            // selection.cpp:288
            const u64 km = __ballot(keep);
            if (km) {
                u64 gb = 0;
                if (lane == 0) gb = atomicAdd(&pc->n_results, (u64)__popcll(km));
                gb = ((u64)__builtin_amdgcn_readfirstlane((int)(gb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)gb);
                if (keep) {
                    const u64 idx = gb + (u64)__popcll(km & ((1ull << lane) - 1ull));
                    if (idx < results_cap) { results[idx].i = pr.x; results[idx].k = pr.y; results[idx].jaccard = jacc; }
                }
            }
        }
        __syncthreads();
        SELHIP_SMALL_STAMP(5);
    }
    SELHIP_SMALL_STAMP(6);
}

}  // namespace
