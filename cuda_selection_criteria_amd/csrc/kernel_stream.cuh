// kernel_stream.cuh -- stage 1, ALGO_STREAM: full m-bucket compare with the query tile in LDS/VGPRs; generic lane-per-candidate kernel.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// Band predicate on lane masks.  A candidate chunk of 128 buckets is held as one u64x2 per lane:
// lane l owns buckets (2l, 2l+1).  m0/m1 are the v_cmp_eq_u64 lane masks of the even/odd bucket.
// A band of r = 2^LOG2R consecutive buckets is, for r >= 2, r/2 consecutive lanes of (m0 & m1).
// Returns a mask with a bit set for every fully equal band (r <= 128).
// ---------------------------------------------------------------------------------------------
template <int HALF>
__host__ __device__ constexpr u64 align_mask() {
    // one bit at every multiple of HALF
    u64 v = 0;
    for (int b = 0; b < 64; b += HALF) v |= 1ull << b;
    return v;
}

template <int LOG2R>
__device__ __forceinline__ u64 band_fold(u64 m0, u64 m1) {
    if constexpr (LOG2R == 0) {
        return m0 | m1;
    } else {
        constexpr int HALF = 1 << (LOG2R - 1);
        u64 t = m0 & m1;
#pragma unroll
        for (int s = 1; s < HALF; s <<= 1) t &= t >> s;
        return t & align_mask<HALF>();
    }
}

// ---------------------------------------------------------------------------------------------
// smh_stream_kernel<NCH, LOG2R>: m = 128*NCH buckets, bands of 2^LOG2R rows (LOG2R == 7: r >= 128,
// runtime r_rt, a band covers r_rt/128 whole chunks).
//   block  = 4 waves; one block = (query tile of Q = 32/NCH rows) x (chunk of kChunk candidates)
//   LDS    = the Q query sketches (32 KiB), staged once per block, then copied to VGPRs by each wave
//   stream = each wave walks its candidates (stride 4), NCH x global_load_dwordx4 per candidate
// blockIdx.x -> (tile = b % n_tiles, chunk = b / n_tiles): blocks b and b+8 (same XCD under round-robin
// dispatch) work on the same candidate chunk, so the chunk is served by that XCD's L2.
// ---------------------------------------------------------------------------------------------
template <int NCH, int LOG2R>
__global__ __launch_bounds__(kBlock, (NCH <= 4 ? 3 : 2))      // 3 waves/SIMD = at most 168 VGPRs (measured: 2 waves cost 17 %)
void smh_stream_kernel(const u64x2* __restrict__ aux, int n, int r_rt,
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
    // software pipeline: the next candidate's loads are in flight while the current one is compared
    u64x2 cand[NCH], nxt[NCH];
    {
        const u64x2* row = aux + (long long)k * ROWV + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) nxt[c] = row[c * kWave];
    }
    for (; k < k_end; k += kWavesPerBlock) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) cand[c] = nxt[c];
        {
            const int kn = min(k + kWavesPerBlock, k_end - 1);                // clamped: last prefetch re-reads a valid row
            const u64x2* row = aux + (long long)kn * ROWV + lane;
#pragma unroll
            for (int c = 0; c < NCH; ++c) nxt[c] = row[c * kWave];
        }

#pragma unroll
        for (int a = 0; a < Q; ++a) {
            bool pass;
            if constexpr (LOG2R < 7) {
                u64 acc = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    u64 m0 = __ballot(cand[c].x == q[a][c].x);
                    u64 m1 = __ballot(cand[c].y == q[a][c].y);
                    acc |= band_fold<LOG2R>(m0, m1);
                }
                pass = acc != 0;
            } else {
                // r_rt >= 128: a band is r_rt/128 consecutive chunks, all 128 buckets of each equal
                const int G = r_rt >> 7;
                pass = false;
                bool run = true;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    u64 m0 = __ballot(cand[c].x == q[a][c].x);
                    u64 m1 = __ballot(cand[c].y == q[a][c].y);
                    bool full = (m0 & m1) == ~0ull;
                    if ((c % G) == 0) run = true;
                    run = run && full;
                    if ((c % G) == G - 1 && run) pass = true;
                }
            }
            if (pass) {
                const int i = i0 + a;
                if (i < i_end && k > i && k >= z0 && k <= hi[i]) app.push_uniform(i, k, lane);
            }
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
