// kernel_stream.cuh -- stage 1, ALGO_STREAM: full m-bucket compare with the query tile in LDS/VGPRs; generic lane-per-candidate kernel.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// ALGO_STREAM -- the literal design of BASELINE.json's north_star: the m-bucket sketches are read coalesced from HBM, one
// tile of query sketches is staged in LDS per workgroup, and the per-pair bucket compares are reduced with wavefront
// primitives (v_cmp_eq_u64 lane masks, scalar AND/shift folds).
//
// Lane <-> bucket mapping.  A sketch is m = 64 * B buckets; lane l owns the B CONTIGUOUS buckets [l*B, (l+1)*B), i.e.
// NCH = B/2 registers of type u64x2.  To keep the loads coalesced (lane l reading 16 bytes next to lane l+1's) the pass
// first writes a bucket-interleaved copy of the sketch array (stream_interleave_kernel):
//     P[row][c*64 + l] = R[row][l*NCH + c]        (u64x2 units, c < NCH, l < 64)
// so that `global_load_dwordx4` number c of a wave fetches 1 KiB contiguous and hands lane l its buckets l*B + 2c, 2c+1.
// With the band of r consecutive buckets inside ONE lane (r <= B) or spanning r/B whole lanes (r >= B), the band predicate is
//     r >= B:  M = AND of the 2*NCH v_cmp_eq_u64 lane masks (bit l = "all of lane l's buckets equal"), then a band is r/B
//              consecutive set bits at an aligned position: log2(r/B) shift-AND steps + one alignment mask, on the scalar unit;
//     r <  B:  per group of r buckets inside the lane the AND of its masks, OR over the groups.
// Per (query, candidate): 2*NCH vector compares and ~2*NCH scalar ANDs -- at m = 512, r = 8 (BASELINE configs[2]): 8 + ~10.
// (Round 1 mapped lane l to buckets (2l, 2l+1) of each 128-bucket chunk: a band then always straddled lanes and every
//  CHUNK needed its own shift-AND fold -- 33 scalar instructions per pair, 81 % of the scalar issue rate, 3.35 ms at cfg3.)
// ---------------------------------------------------------------------------------------------
template <int L>
__host__ __device__ constexpr u64 align_mask() {
    // one bit at every multiple of L
    u64 v = 0;
    for (int b = 0; b < 64; b += L) v |= 1ull << b;
    return v;
}

__global__ __launch_bounds__(kBlock)
void stream_interleave_kernel(const u64x2* __restrict__ in, u64x2* __restrict__ out, long long total, int nch) {
    const long long o = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (o >= total) return;
    const int per_row = nch * kWave;
    const long long row = o / per_row;
    const int rem = (int)(o - row * per_row);
    const int c = rem / kWave, l = rem % kWave;
    out[o] = in[row * per_row + l * nch + c];
}

template <int NCH, int LOG2R>
__device__ __forceinline__ bool stream_band_pass(const u64x2 (&cand)[NCH], const u64x2 (&q)[NCH]) {
    constexpr int B = 2 * NCH, R = 1 << LOG2R;
    if constexpr (R >= B) {
        u64 M = ~0ull;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            M &= __ballot(cand[c].x == q[c].x);
            M &= __ballot(cand[c].y == q[c].y);
        }
        constexpr int L = R / B;                      // lanes per band
        if constexpr (L >= 64) {
            return M == ~0ull;
        } else {
#pragma unroll
            for (int s = 1; s < L; s <<= 1) M &= M >> s;
            return (M & align_mask<L>()) != 0;
        }
    } else {
        u64 any = 0;
#pragma unroll
        for (int g = 0; g < B / R; ++g) {
            u64 Mg = ~0ull;
#pragma unroll
            for (int j = g * R; j < (g + 1) * R; ++j)
                Mg &= (j & 1) ? __ballot(cand[j >> 1].y == q[j >> 1].y) : __ballot(cand[j >> 1].x == q[j >> 1].x);
            any |= Mg;
        }
        return any != 0;
    }
}

// ---------------------------------------------------------------------------------------------
// smh_stream_kernel<NCH, LOG2R>: m = 128*NCH buckets, bands of 2^LOG2R rows.
//   block  = 4 waves; one block = (query tile of Q = 24/NCH rows, kQueryVgprBudget) x (chunk of kChunk candidates)
//   LDS    = the Q query sketches (24 KiB), staged once per block with coalesced 16-B loads, then copied to VGPRs by each wave
//   stream = each wave walks its candidates (stride 4), NCH x global_load_dwordx4 (1 KiB each) per candidate, kStreamAhead candidates ahead
// blockIdx.x -> (tile = b % n_tiles, chunk = b / n_tiles): blocks b and b+8 (same XCD under round-robin
// dispatch) work on the same candidate chunk, so the chunk is served by that XCD's L2.
// Registers: Q*NCH = 24 query + a ring of 3*NCH candidate u64x2 (96 + 48 VGPRs at NCH = 4): no scratch under the 168-VGPR cap of
// 3 waves per SIMD (round 1 held 32 query registers and spilled 10 VGPRs: 222 MB of scratch writes per launch).
// Tried and dropped (round 2): the candidate rows shared by the block's four waves through an LDS ring filled by LDS-DMA loads
// (`global_load_lds_dwordx4`, 8 rows deep, every wave comparing every row with its own 7 queries, LDS reads double-buffered in
// registers): 4x fewer L2 reads, but one block barrier per candidate row keeps four waves on four SIMDs in lockstep --
// 1.78 ms against 1.48 ms for this kernel at cfg3 (bit-identical results).
// ---------------------------------------------------------------------------------------------
template <int NCH, int LOG2R>
__global__ __launch_bounds__(kBlock, (NCH <= 4 ? 3 : 2))
void smh_stream_kernel(const u64x2* __restrict__ aux, int n,
                       const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                       RowMap rm, int n_tiles, int chunk_base,
                       selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    constexpr int Q = kQueryVgprBudget / NCH;
    constexpr int ROWV = NCH * kWave;                 // u64x2 per sketch row
    __shared__ u64x2 qs[Q * ROWV];
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];

    const int tile = blockIdx.x % n_tiles;
    const int chunk = blockIdx.x / n_tiles;
    int i0, i_end;
    rm.tile_rows(tile, Q, &i0, &i_end);
    if (i0 >= i_end) return;
    const int i_last = i_end - 1;
    const int z0 = pc_in->z0p1 ? pc_in->z0p1 - 1 : n;
    const int k0 = chunk_base + chunk * kChunk;
    const int kmax = hi[i_last];                      // hi is non-decreasing in i
    const int kmin = max(i0 + 1, z0);
    if (k0 > kmax || k0 + kChunk - 1 < kmin) return;

    // stage the query tile: rows i0 .. i0+Q-1 are contiguous in memory
    {
        const long long base = (long long)i0 * ROWV;
        const long long limit = (long long)n * ROWV;
        for (int t = threadIdx.x; t < Q * ROWV; t += kBlock) {
            long long src = base + t;
            if (src >= limit) src = limit - 1;        // rows past the end: never valid, any data will do
            qs[t] = aux[src];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);     // wave-uniform -> SGPR loop counter
    u64x2 q[Q][NCH];
#pragma unroll
    for (int a = 0; a < Q; ++a)
#pragma unroll
        for (int c = 0; c < NCH; ++c) q[a][c] = qs[(a * NCH + c) * kWave + lane];

    const int k_end = min(min(k0 + kChunk, n), kmax + 1);
    int k = max(k0, kmin) + wave;
    if (k >= k_end) return;
    WaveAppender app;
    app.init(app_lds, wave, surv, surv_cap, &pc->n_survivors);
    // software pipeline: kStreamAhead candidates' loads are in flight while one is compared -- a ring of kStreamAhead + 1 register
    // sets, the loop unrolled over the ring so that every set is addressed statically
    constexpr int AHEAD = NCH <= 4 ? kStreamAhead : 1;                         // (m >= 1024: the registers allow one row ahead)
    constexpr int RING = AHEAD + 1;
    u64x2 ring[RING][NCH];
    auto load_row = [&](u64x2 (&dst)[NCH], int kk) {
        const int kc = min(kk, k_end - 1);                                    // clamped: prefetches past the end re-read a valid row
        const u64x2* row = aux + (long long)kc * ROWV + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) dst[c] = row[c * kWave];
    };
    auto compare_row = [&](const u64x2 (&cand)[NCH], int kk) {
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            if (stream_band_pass<NCH, LOG2R>(cand, q[a])) {
                // (the empty volatile asm pins the branch on the band mask alone: merged with the range tests below, the compiler
                //  spent 10 scalar instructions per pair on a condition that is false for all but ~1 pair in 1 000)
                asm volatile("");
                const int i = i0 + a;
                if (i < i_end && kk > i && kk >= z0 && kk <= hi[i]) app.push_uniform(i, kk, lane);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) load_row(ring[s], k + s * kWavesPerBlock);
    for (; k < k_end; k += RING * kWavesPerBlock) {
#pragma unroll
        for (int s = 0; s < RING; ++s) {
            const int kk = k + s * kWavesPerBlock;
            if (kk >= k_end) break;
            load_row(ring[(s + AHEAD) % RING], kk + AHEAD * kWavesPerBlock);
            compare_row(ring[s], kk);
        }
    }
    app.flush(lane);
}

// ---------------------------------------------------------------------------------------------
// smh_a for one pair evaluated by ONE LANE (any m, rows, bands): criteria_sketch.hpp:66-81 literally.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool smh_a_lane(const u64* __restrict__ v1, const u64* __restrict__ v2,
                                           int n_rows, int n_bands) {
    for (int b = 0; b < n_bands; ++b) {
        const u64* x = v1 + (long long)b * n_rows;
        const u64* y = v2 + (long long)b * n_rows;
        int j = 0;
        while (j < n_rows && x[j] == y[j]) ++j;
        if (j == n_rows) return true;
    }
    return false;
}

// generic stage 1: block = 256 lanes = 256 candidates of one query row; grid = (chunks, rows)
__global__ __launch_bounds__(kBlock)
void smh_generic_kernel(const u64* __restrict__ aux, int n, int m, int n_rows, int n_bands,
                        const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                        RowMap rm, int n_rows_grid,
                        selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];
    int i, i_e;
    rm.tile_rows((int)(blockIdx.x % n_rows_grid), 1, &i, &i_e);
    const int chunk = blockIdx.x / n_rows_grid;
    if (i >= i_e) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int z0 = pc_in->z0p1 ? pc_in->z0p1 - 1 : n;
    const int kmin = max(i + 1, z0);
    const int kmax = hi[i];
    const int k = kmin + chunk * kBlock + (int)threadIdx.x;
    const bool in_range = k <= kmax && k < n;
    const bool ok = in_range && smh_a_lane(aux + (long long)i * m, aux + (long long)k * m, n_rows, n_bands);
    WaveAppender app;
    app.init(app_lds, wave, surv, surv_cap, &pc->n_survivors);
    app.push(ok, i, k, lane);
    app.flush(lane);
}

}  // namespace
