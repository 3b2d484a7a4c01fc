// kernel_sigjoin.cuh -- stage 1, ALGO_SIG: band signatures, all-pairs signature join (DPP broadcast), exact verification.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

// =============================================================================================
// ALGO_SIG -- stage 1 as a signature join (exact):
//   a pair passes smh_a iff SOME band of r buckets is entirely equal (criteria_sketch.hpp:66-81).  Equal bands
//   have equal 32-bit signatures (a hash of the band's r u64 values), so "some band signature equal" is a
//   necessary condition; pairs that meet it are CANDIDATES and are verified with the literal predicate on the
//   full sketches (verify_kernel).  A hash collision only adds a candidate that the verification rejects
//   (expected n_bands * 2^-32 per pair), it can never drop a pair: the survivor set is identical to the
//   stream kernel's.  The all-pairs part then costs n_bands 32-bit compares per pair instead of m 64-bit ones.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}

// sig_build_kernel: one thread per bucket, coalesced 8-B loads; the r lanes of a band add their position-salted
// mixes with DPP row shifts (r <= 16) or xor-shuffles (r <= 64) -- or one thread walks the band (r > 64).  Writes the layouts:
//   sigQ[g][NB] (genome-major), sigT[b][n_pad] (band-major, lane = candidate) and sigP[b/2][n_pad]: the top 16 bits of
//   the signatures of bands 2d (low half) and 2d+1 (high half) packed in one dword, for the 16-bit join -- plus sigG[g][NB/2],
//   the same packed dwords genome-major (the query side of sigs_join_kernel, read through the scalar cache).
template <int S>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t x) {
    // DPP row_shr:S -- lane i of a 16-lane row reads lane i-S of the row, 0 for i < S
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x110 + S, 0xF, 0xF, true);
}

// signature of a band = xor of the two 32-bit halves of  sum_j mix64(v_j + salt_j)  taken half by half (no carry between
// the halves, so that the lanes of a band can add them with 32-bit DPP shifts).  With m and nb powers of two (always, for the
// all-pairs joins) bucket index -> (genome, band, position) is shifts and masks instead of 64-bit divisions.
__device__ __forceinline__ void sig_build_body(long long block, const u64* __restrict__ aux, int n, int m, int r, int nb, int n_pad,
                                               uint32_t* __restrict__ sigQ, uint32_t* __restrict__ sigT, uint32_t* __restrict__ sigP,
                                               uint32_t* __restrict__ sigG, int pk_shift) {
    uint16_t* const sigP16 = reinterpret_cast<uint16_t*>(sigP);
    uint16_t* const sigG16 = reinterpret_cast<uint16_t*>(sigG);          // genome-major twin of sigP: [g][(nb + 1) / 2] dwords
    const long long g_pitch = 2ll * ((nb + 1) / 2);
    // r is a power of two; m and nb are too for the all-pairs joins (sig_supported) but not necessarily for the sort-based join
    const bool pow2 = (m & (m - 1)) == 0 && (nb & (nb - 1)) == 0;
    const int lr = __builtin_ctz((unsigned)r), lm = __builtin_ctz((unsigned)m);
    if (r <= kWave) {
        const long long t = block * kBlock + threadIdx.x;                      // global bucket index
        const long long total = (long long)n * m;
        uint32_t hl = 0, hh = 0;
        const int j = (int)(t & (r - 1));                                      // position inside the band
        if (t < total) {
            const u64 x = mix64(aux[t] + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
            hl = (uint32_t)x; hh = (uint32_t)(x >> 32);
        }
        int writer = r - 1;                                                    // the band's lane that ends up with the sums
        if (r <= 16) {                                                         // a band lies inside one 16-lane row: prefix adds by DPP shifts
            if (r > 1) { hl += dpp_row_shr<1>(hl); hh += dpp_row_shr<1>(hh); }
            if (r > 2) { hl += dpp_row_shr<2>(hl); hh += dpp_row_shr<2>(hh); }
            if (r > 4) { hl += dpp_row_shr<4>(hl); hh += dpp_row_shr<4>(hh); }
            if (r > 8) { hl += dpp_row_shr<8>(hl); hh += dpp_row_shr<8>(hh); }
        } else {
            for (int s = 1; s < r; s <<= 1) { hl += __shfl_xor(hl, s, kWave); hh += __shfl_xor(hh, s, kWave); }
            writer = 0;
        }
        if (t < total && j == writer) {
            const int g = pow2 ? (int)(t >> lm) : (int)(t / m);
            const int b = (pow2 ? (int)(t & (m - 1)) : (int)(t % m)) >> lr;
            const uint32_t sig = hl ^ hh;
            sigQ[(long long)g * nb + b] = sig;
            sigT[(long long)b * n_pad + g] = sig;
            sigP16[((long long)(b >> 1) * n_pad + g) * 2 + (b & 1)] = (uint16_t)(sig >> pk_shift);
            sigG16[(long long)g * g_pitch + b] = (uint16_t)(sig >> pk_shift);
        }
    } else {
        const long long t = block * kBlock + threadIdx.x;                      // (genome, band)
        if (t >= (long long)n * nb) return;
        const int g = pow2 ? (int)(t >> __builtin_ctz((unsigned)nb)) : (int)(t / nb);
        const int b = pow2 ? (int)(t & (nb - 1)) : (int)(t % nb);
        const u64* v = aux + (long long)g * m + ((long long)b << lr);
        uint32_t hl = 0, hh = 0;
        for (int j = 0; j < r; ++j) {
            const u64 x = mix64(v[j] + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
            hl += (uint32_t)x; hh += (uint32_t)(x >> 32);
        }
        const uint32_t sig = hl ^ hh;
        sigQ[(long long)g * nb + b] = sig;
        sigT[(long long)b * n_pad + g] = sig;
        sigP16[((long long)(b >> 1) * n_pad + g) * 2 + (b & 1)] = (uint16_t)(sig >> pk_shift);
        sigG16[(long long)g * g_pitch + b] = (uint16_t)(sig >> pk_shift);
    }
}

// sig_build_tile_body: the same signatures, built kSigTileG genomes per block (power-of-two m and nb <= 128, 2 <= r <= 32).
//  * a thread takes TWO buckets per 16-byte load and four loads are in flight per thread before the first is hashed (the
//    one-bucket-per-thread form above left ~4 short-lived waves per SIMD waiting on one 8-byte load each: 18 us for 41 MB);
//  * the band sums are formed with DPP row shifts over the r/2 lanes of a band as before, but land in an LDS tile
//    [genome][band], from which all four layouts are written in full segments: sigQ / sigG rows contiguous, sigT / sigP as
//    kSigTileG consecutive genomes per band (the per-genome form wrote each band-major entry as a lone 4-byte store: 19.5 MB
//    of HBM writes for 7.7 MB of signatures, PMC WRITE_SIZE).
constexpr int kSigTileG = 32;             // LDS capacity of a tile; the launch says how many genomes a tile holds (16 by default)

// 16-byte loads a thread has in flight (8 and 16 measured no better: build alone 17.3 / 17.5 / 18.9 us at cfg3, 60.4 / 60.1 / 63.0 at cfg4,
// gpurun_out/r03/p_*)
constexpr int kSigLoads = 4;
template <int CAP>
__device__ __forceinline__ void sig_build_tile_body(int tile, const u64* __restrict__ aux, int n, int m, int r, int nb, int n_pad,
                                                    uint32_t* __restrict__ sigQ, uint32_t* __restrict__ sigT, uint32_t* __restrict__ sigP,
                                                    uint32_t* __restrict__ sigG, int pk_shift,
                                                    int tg, bool all_layouts = true) {
    // (tg <= CAP genomes per tile: the one-launch pass of a small set spreads ITS genomes over all its blocks; it only reads the
    //  32-bit layouts sigQ / sigT and leaves the packed ones alone)
    __shared__ uint32_t sig_lds[CAP][129];                          // pitch 129: the band-major read-out is conflict-free
    const int g0 = tile * tg;
    const int ng = min(tg, n - g0);
    if (ng <= 0) return;
    const int lr = __builtin_ctz((unsigned)r), lm = __builtin_ctz((unsigned)m);
    const int half_m = m >> 1;                                            // bucket pairs per genome
    const int total = ng * half_m;                                        // bucket pairs of this tile
    const u64x2* src = reinterpret_cast<const u64x2*>(aux + (size_t)g0 * m);
    const int L = r >> 1;                                                 // lanes per band (1..16)
    for (int base = 0; base < total; base += kSigLoads * kBlock) {
        u64x2 v[kSigLoads];
#pragma unroll
        for (int u = 0; u < kSigLoads; ++u) {
            const int idx = base + u * kBlock + (int)threadIdx.x;
            v[u] = idx < total ? src[idx] : u64x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < kSigLoads; ++u) {
            const int idx = base + u * kBlock + (int)threadIdx.x;         // blocks of 256 pairs never straddle a band (L <= 16 divides 256)
            const int bucket = (idx << 1) & (m - 1);
            const int j = bucket & (r - 1);                               // position of the first of the two buckets inside its band
            const u64 x0 = mix64(v[u].x + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
            const u64 x1 = mix64(v[u].y + 0x9E3779B97F4A7C15ull * (u64)(j + 2));
            uint32_t hl = (uint32_t)x0 + (uint32_t)x1, hh = (uint32_t)(x0 >> 32) + (uint32_t)(x1 >> 32);
            if (L > 1) { hl += dpp_row_shr<1>(hl); hh += dpp_row_shr<1>(hh); }
            if (L > 2) { hl += dpp_row_shr<2>(hl); hh += dpp_row_shr<2>(hh); }
            if (L > 4) { hl += dpp_row_shr<4>(hl); hh += dpp_row_shr<4>(hh); }
            if (L > 8) { hl += dpp_row_shr<8>(hl); hh += dpp_row_shr<8>(hh); }
            if (idx < total && (j + 2) == r) sig_lds[idx >> (lm - 1)][bucket >> lr] = hl ^ hh;      // the band's last lane holds the sums
        }
    }
    __syncthreads();
    uint16_t* const sigG16 = reinterpret_cast<uint16_t*>(sigG);
    const int ndw = (nb + 1) >> 1;
    for (int idx = threadIdx.x; idx < ng * nb; idx += kBlock) {            // genome-major: rows contiguous
        const int gl = idx / nb, b = idx - gl * nb;
        const uint32_t sig = sig_lds[gl][b];
        sigQ[(size_t)(g0 + gl) * nb + b] = sig;
        if (all_layouts) sigG16[(size_t)(g0 + gl) * (2 * ndw) + b] = (uint16_t)(sig >> pk_shift);
    }
    for (int idx = threadIdx.x; idx < nb * tg; idx += kBlock) {            // band-major: tg consecutive genomes per band
        const int b = idx / tg, gl = idx - b * tg;
        if (gl < ng) sigT[(size_t)b * n_pad + g0 + gl] = sig_lds[gl][b];
    }
    if (!all_layouts) return;
    for (int idx = threadIdx.x; idx < ndw * tg; idx += kBlock) {
        const int d = idx / tg, gl = idx - d * tg;
        if (gl < ng) {
            const uint32_t lo = sig_lds[gl][2 * d] >> pk_shift, hi2 = (2 * d + 1 < nb) ? (sig_lds[gl][2 * d + 1] >> pk_shift) : 0u;
            sigP[(size_t)d * n_pad + g0 + gl] = (lo & 0xFFFFu) | (hi2 << 16);
        }
    }
}

// One launch for the two kernels every signature pass starts with: the first `bounds_blocks` blocks run cb_bounds_body (a few
// waves of binary searches, latency-bound), the others build the signatures -- the bounds then cost nothing on the stream.
__global__ __launch_bounds__(kBlock)
void sig_build_kernel(const u64* __restrict__ aux, int n, int m, int r, int nb, int n_pad,
                      uint32_t* __restrict__ sigQ, uint32_t* __restrict__ sigT, uint32_t* __restrict__ sigP, uint32_t* __restrict__ sigG,
                      int bounds_blocks, const double* __restrict__ cards, double tau, int use_cb, RowMap rm,
                      u64* __restrict__ ecard, int* __restrict__ hi, PassCounters* __restrict__ pc, int* __restrict__ csr_zero, int csr_zero_n, int cand_begin,
                      u64* __restrict__ seg_zero, int seg_zero_n, int pk_shift, PassCounters* __restrict__ zero_pc, int tile_mode) {
    // (tile_mode = genomes per tile of the tiled build, 0 = the per-bucket form)
    if ((int)blockIdx.x < bounds_blocks) {
        const int t = (int)(blockIdx.x * kBlock + threadIdx.x);
        zero_next_counters(t, bounds_blocks * kBlock, zero_pc, kCounterBlocks);
        for (int j = t; j < seg_zero_n; j += bounds_blocks * kBlock) seg_zero[j] = 0;      // the join's append-segment counters of this pass
        for (int j = t; j < csr_zero_n; j += bounds_blocks * kBlock) csr_zero[j] = 0;      // stage 2's per-row counters, one set per chunk lane
        cb_bounds_body(t, cards, n, tau, use_cb, rm, ecard, hi, pc, cand_begin);
        return;
    }
    if (tile_mode) sig_build_tile_body<kSigTileG>((int)blockIdx.x - bounds_blocks, aux, n, m, r, nb, n_pad, sigQ, sigT, sigP, sigG, pk_shift, tile_mode);
    else           sig_build_body((long long)blockIdx.x - bounds_blocks, aux, n, m, r, nb, n_pad, sigQ, sigT, sigP, sigG, pk_shift);
}

// sig_join_kernel<NB>: all-pairs "some band signature equal", entirely on the vector unit.
//   lane = candidate k: its NB signatures live in VGPRs c[0..NB)            (loaded once per wave)
//   queries come 16 at a time: lane l holds the signatures of query i16 + (l & 15) in qv[0..NB) (the four
//   16-lane rows hold the same 16 queries); query j of the batch is broadcast to every lane by the DPP
//   modifier row_newbcast:j ON the xor itself (v_xor_b32_dpp), so a band compare costs
//       t = c[b] ^ bcast_j(qv[b]);  acc = min(acc, t)          (v_xor_b32_dpp + v_min_u32 / v_min3_u32)
//   with no LDS, scalar-cache or SGPR traffic in the inner loop; acc == 0 iff some band matched.
// Three earlier forms measured 0.5-0.65 ms on cfg3 and are recorded in DESIGN.md section 4: v_cmp_eq_u32 -> SGPR
// mask -> s_or_b64 per band; query signatures streamed through scalar loads (the scalar-cache miss path
// sustains ~0.5 B/clk/CU); query tile in LDS read back with broadcast ds_read_b128 (latency-bound at the
// occupancy its registers allow).
template <int J>
__device__ __forceinline__ uint32_t dpp_row_bcast(uint32_t x) {
    // DPP_ROW_NEWBCAST (gfx90a+): every lane reads lane J of its own 16-lane row
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + J, 0xF, 0xF, true);
}

template <int NB, int J>
__device__ __forceinline__ void join_one_query(const uint32_t (&c)[NB], const uint32_t (&qv)[NB], int i, int i_hi,
                                               int k, int lane, int z0, int n, const int* __restrict__ hi,
                                               WaveAppender& app) {
    // independent chains of v_min3_u32(acc, x, y): 1.5 VALU per band; four chains (when there are enough bands) so that
    // a wave has issue work while a chain waits on its own result
    uint32_t acc0 = 0xFFFFFFFFu, acc1 = 0xFFFFFFFFu, acc2 = 0xFFFFFFFFu, acc3 = 0xFFFFFFFFu;
    if constexpr (NB >= 16) {
#pragma unroll
        for (int b = 0; b < NB; b += 8) {
            acc0 = min(min(acc0, c[b] ^ dpp_row_bcast<J>(qv[b])), c[b + 1] ^ dpp_row_bcast<J>(qv[b + 1]));
            acc1 = min(min(acc1, c[b + 2] ^ dpp_row_bcast<J>(qv[b + 2])), c[b + 3] ^ dpp_row_bcast<J>(qv[b + 3]));
            acc2 = min(min(acc2, c[b + 4] ^ dpp_row_bcast<J>(qv[b + 4])), c[b + 5] ^ dpp_row_bcast<J>(qv[b + 5]));
            acc3 = min(min(acc3, c[b + 6] ^ dpp_row_bcast<J>(qv[b + 6])), c[b + 7] ^ dpp_row_bcast<J>(qv[b + 7]));
        }
    } else {
#pragma unroll
        for (int b = 0; b < NB; b += 4) {
            acc0 = min(min(acc0, c[b] ^ dpp_row_bcast<J>(qv[b])), c[b + 1] ^ dpp_row_bcast<J>(qv[b + 1]));
            acc1 = min(min(acc1, c[b + 2] ^ dpp_row_bcast<J>(qv[b + 2])), c[b + 3] ^ dpp_row_bcast<J>(qv[b + 3]));
        }
    }
    const u64 mm = __ballot(min(min(acc0, acc1), min(acc2, acc3)) == 0u);
    if (mm && i < i_hi) {
        const int lo = max(i + 1, z0);
        const int hk = min(hi[i], n - 1);
        app.push(((mm >> lane) & 1ull) && k >= lo && k <= hk, i, k, lane);
    }
}

template <int NB>
__global__ __launch_bounds__(kBlock)
void sig_join_kernel(const uint32_t* __restrict__ sigT, int n, int n_pad,
                     const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                     RowMap rm, int n_tiles, int group_base, int qt,
                     selhip_int2_t* __restrict__ cand, u64 cand_cap, PassCounters* __restrict__ pc) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int tile = blockIdx.x % n_tiles;
    const int grp = group_base + (blockIdx.x / n_tiles) * kWavesPerBlock + wave;
    const int k_base = grp * kWave;
    if (k_base >= n) return;
    const int z0 = pc_in->z0p1 ? pc_in->z0p1 - 1 : n;
    const int k_last = k_base + kWave - 1;
    int i_lo, i_end;
    rm.tile_rows(tile, qt, &i_lo, &i_end);                                    // qt is a multiple of 16
    const int i_hi = min(i_end, k_last);                                      // need i < k for some lane
    if (i_lo >= i_hi || k_last < z0) return;
    if (hi[i_hi - 1] < k_base) return;                                        // hi is non-decreasing

    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];
    WaveAppender app;
    app.init(app_lds, wave, cand, cand_cap, &pc->n_candidates);
    const int k = k_base + lane;                                              // < n_pad
    uint32_t c[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) c[b] = sigT[(long long)b * n_pad + k];

    for (int i16 = i_lo; i16 < i_hi; i16 += 16) {
        const int qi = min(i16 + (lane & 15), n_pad - 1);
        uint32_t qv[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) qv[b] = sigT[(long long)b * n_pad + qi];
#define SELHIP_JQ(J) join_one_query<NB, J>(c, qv, i16 + J, i_hi, k, lane, z0, n, hi, app);
        SELHIP_JQ(0) SELHIP_JQ(1) SELHIP_JQ(2) SELHIP_JQ(3) SELHIP_JQ(4) SELHIP_JQ(5) SELHIP_JQ(6) SELHIP_JQ(7)
        SELHIP_JQ(8) SELHIP_JQ(9) SELHIP_JQ(10) SELHIP_JQ(11) SELHIP_JQ(12) SELHIP_JQ(13) SELHIP_JQ(14) SELHIP_JQ(15)
#undef SELHIP_JQ
    }
    app.flush(lane);
}

// sig16_join_kernel<ND>: the same all-pairs join on 16-bit signatures packed two bands per dword (sigP): one band
// compare is half a `v_xor_b32_dpp` plus half a `v_pk_min_u16` -- 1.0 instead of 1.5 VALU instructions per band -- and a
// lane holds ND = NB/2 registers per side.  A zero 16-bit half of the running minimum = some band's 16-bit signature
// equal: a superset (NB * 2^-16 per pair) of the 32-bit candidates, cut back to exactly that set by verify16_kernel.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

template <int ND, int J>
__device__ __forceinline__ void join16_one_query(const uint32_t (&c)[ND], const uint32_t (&qv)[ND], int i, int i_hi,
                                                 int k, int lane, int z0, int n, const int* __restrict__ hi,
                                                 WaveAppender& app) {
    us2_t acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = us2_t{0xFFFF, 0xFFFF};
#pragma unroll
    for (int d = 0; d < ND; d += 4) {
        uint32_t x[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) x[a] = c[d + a] ^ dpp_row_bcast<J>(qv[d + a]);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_elementwise_min(acc[a], __builtin_bit_cast(us2_t, x[a]));
    }
    const uint32_t mv = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_min(acc[0], acc[1]),
                                                                                 __builtin_elementwise_min(acc[2], acc[3])));
    const u64 mm = __ballot(min(mv & 0xFFFFu, mv >> 16) == 0u);
    if (mm && i < i_hi) {
        const int lo = max(i + 1, z0);
        const int hk = min(hi[i], n - 1);
        app.push(((mm >> lane) & 1ull) && k >= lo && k <= hk, i, k, lane);
    }
}

template <int ND, bool DB, int WPB>
__global__ __launch_bounds__(WPB * kWave)
void sig16_join_kernel(const uint32_t* __restrict__ sigP, int n, int n_pad,
                       const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                       RowMap rm, int n_tiles, int group_base, int qt,
                       selhip_int2_t* __restrict__ pre, u64 pre_cap, u64* __restrict__ seg_cnt) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int tile = blockIdx.x % n_tiles;
    const int grp = group_base + (blockIdx.x / n_tiles) * WPB + wave;
    const int k_base = grp * kWave;
    if (k_base >= n) return;
    const int z0 = pc_in->z0p1 ? pc_in->z0p1 - 1 : n;
    const int k_last = k_base + kWave - 1;
    int i_lo, i_end;
    rm.tile_rows(tile, qt, &i_lo, &i_end);                                    // qt is a multiple of 16
    const int i_hi = min(i_end, k_last);                                      // need i < k for some lane
    if (i_lo >= i_hi || k_last < z0) return;
    if (hi[i_hi - 1] < k_base) return;                                        // hi is non-decreasing

    __shared__ selhip_int2_t app_lds[WPB * kAppendCap];
    WaveAppender app;
    {
        const int seg = blockIdx.x % kAppendSegs;
        const u64 seg_cap = pre_cap / kAppendSegs;
        app.init(app_lds, wave, pre + (size_t)seg * seg_cap, seg_cap, seg_cnt + seg * kSegStride);
    }
    const int k = k_base + lane;                                              // < n_pad
    // sigP through a buffer resource: the band's row offset d * n_pad * 4 rides in the instruction's scalar offset and the
    // lane's column in its 32-bit vector offset, so a load costs no 64-bit address arithmetic on the vector unit (it was
    // ~7 % of the kernel's VALU instructions); out-of-range offsets read 0 instead of faulting
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(sigP), 0, ND * n_pad * 4, 0x00020000);
    const int row_bytes = n_pad * 4;
    uint32_t c[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) c[d] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, k * 4, d * row_bytes, 0);

#define SELHIP_JQ16(Q, I0) \
    { join16_one_query<ND, 0>(c, Q, (I0) + 0, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 1>(c, Q, (I0) + 1, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 2>(c, Q, (I0) + 2, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 3>(c, Q, (I0) + 3, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 4>(c, Q, (I0) + 4, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 5>(c, Q, (I0) + 5, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 6>(c, Q, (I0) + 6, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 7>(c, Q, (I0) + 7, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 8>(c, Q, (I0) + 8, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 9>(c, Q, (I0) + 9, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 10>(c, Q, (I0) + 10, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 11>(c, Q, (I0) + 11, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 12>(c, Q, (I0) + 12, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 13>(c, Q, (I0) + 13, i_hi, k, lane, z0, n, hi, app); \
      join16_one_query<ND, 14>(c, Q, (I0) + 14, i_hi, k, lane, z0, n, hi, app); join16_one_query<ND, 15>(c, Q, (I0) + 15, i_hi, k, lane, z0, n, hi, app); }
#define SELHIP_LOADQ(Q, I0) \
    { const int qo_ = min((I0) + (lane & 15), n_pad - 1) * 4; \
      _Pragma("unroll") for (int d = 0; d < ND; ++d) Q[d] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, qo_, d * row_bytes, 0); }

    if constexpr (DB) {
        // two register sets: the next 16 queries are in flight while the current 16 are compared
        uint32_t qa[ND], qb[ND];
        SELHIP_LOADQ(qa, i_lo)
        for (int i16 = i_lo; i16 < i_hi; i16 += 32) {
            SELHIP_LOADQ(qb, i16 + 16)
            SELHIP_JQ16(qa, i16)
            if (i16 + 16 >= i_hi) break;
            SELHIP_LOADQ(qa, i16 + 32)
            SELHIP_JQ16(qb, i16 + 16)
        }
    } else {
        for (int i16 = i_lo; i16 < i_hi; i16 += 16) {
            uint32_t qv[ND];
            SELHIP_LOADQ(qv, i16)
            SELHIP_JQ16(qv, i16)
        }
    }
#undef SELHIP_JQ16
#undef SELHIP_LOADQ
    app.flush(lane);
}

// sigl_join_kernel<ND, T, WPB>: the 16-bit join with the QUERY TILE STAGED IN LDS (one tile per workgroup) and broadcast
// to the lanes by plain LDS reads, so that the per-band work on the vector unit is a plain VOP2 xor plus the packed min.
// Why (profiles/r02_valu_rate.txt, scripts/microbench/valu_rate.hip): on gfx950 a VGPR-VGPR VOP2 (`v_xor_b32`, `v_and_b32`,
// `v_add_u32`, `v_min_u16`, `v_fma_f32`) issues every ~2.1 cycles per SIMD once >= 4 waves share it, but ANY instruction with
// a DPP or SDWA modifier, an SGPR operand, three sources (VOP3) or packed math (VOP3P) costs 4.07 cycles at every occupancy.
// The DPP form above therefore pays 4.07 (xor_dpp) + 4.07 (pk_min) = 8.1 cycles per dword (two bands) and 64 pairs; a query
// dword fetched by `s_load` into an SGPR makes the xor cost 4.07 as well (tried: 2.19 vs 2.27 ms at cfg4, no gain).  With
// the query dword in a VGPR the xor is the 2.1-cycle VOP2 -- 6.2 cycles per dword -- and the only way to put it there without
// spending vector-unit time is an LDS read whose 64 lanes share one address (`ds_read_b128`, 4 dwords per instruction, no
// bank conflict): the broadcast moves to the LDS pipe, which the join otherwise leaves idle.
// Layout: the block copies its tile of query rows (qt rows x ND dwords, genome-major sigG, coalesced 16-B loads) into LDS
// once; each of its WPB waves holds T groups of 64 candidates in VGPRs (lane = candidate) and walks the rows, reading row
// chunks of 16 dwords one chunk ahead of the chunk being compared (two register sets).  Results identical to
// sig16_join_kernel (same predicate on the same signatures).
template <int ND, int T, int OFF, int CNT>
__device__ __forceinline__ void joinl_accum(us2_t (&acc)[T][4], const uint32_t (&c)[T][ND], const uint32_t (&q)[CNT]) {
#pragma unroll
    for (int d = 0; d < CNT; d += 4) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t x[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) x[a] = c[t][OFF + d + a] ^ q[d + a];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[t][a] = __builtin_elementwise_min(acc[t][a], __builtin_bit_cast(us2_t, x[a]));
        }
    }
}

// 15-bit form ("join_bits" = 15): every instruction of the loop is a VGPR-only VOP2, the class that two waves co-issue
// (2.07 instead of 4.07 cycles, profiles/r02_valu_rate.txt) -- and they only co-issue with EACH OTHER, which is why the packed
// min above never sees the benefit of its plain xor (same file, rows "mix: v_xor_b32 / v_pk_min_u16": 3.75-3.93 cycles per
// instruction).  The packed dword holds two 15-bit signatures with a clear top bit per half, so
//     t = c ^ q                    both halves in [0, 0x7FFF]
//     u = t + 0x7FFF7FFF           no carry between the halves; bit 15 / bit 31 = "half != 0"
//     acc &= u                     a cleared flag bit = some band's 15-bit signature was equal
// Inline asm keeps the compiler from fusing xor+add into v_xad_u32 or and+or forms (VOP3: full cost, and unpairable).
template <int ND, int T, int OFF, int CNT>
__device__ __forceinline__ void joinl_accum15(uint32_t (&acc)[T][4], const uint32_t (&c)[T][ND], const uint32_t (&q)[CNT], uint32_t k7) {
#pragma unroll
    for (int d = 0; d < CNT; d += 4) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t x[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) asm("v_xor_b32 %0, %1, %2" : "=v"(x[a]) : "v"(c[t][OFF + d + a]), "v"(q[d + a]));
#pragma unroll
            for (int a = 0; a < 4; ++a) asm("v_add_u32 %0, %1, %2" : "=v"(x[a]) : "v"(x[a]), "v"(k7));
#pragma unroll
            for (int a = 0; a < 4; ++a) asm("v_and_b32 %0, %1, %2" : "=v"(acc[t][a]) : "v"(acc[t][a]), "v"(x[a]));
        }
    }
}

// 16-bit "zero half" form ("join_form" = 1, opt-in): the signatures keep all 16 bits and the loop is three 2-cycle instructions per
// dword -- t = c ^ q, u = t - 0x00010001, acc = acc | (u & ~t) (one v_bitop3_b32: profiles/r03_bitplane_rate.txt, 2.2 cycles alone).
// (u & ~t) & 0x80008000 != 0 exactly when a half of t is zero (the classic zero-byte test: a borrow can only raise a false flag ABOVE
// a half that really is zero, so "some half is zero" is exact), and that mask is applied once per row.  Same candidate set as the
// packed-minimum form.  MEASURED SLOWER (gpurun_out/r03/d_form*: cfg3 112.0 vs 105.9 us, cfg4 2.12 vs 2.03 ms): 101 instructions per
// row-wave at 2.64 cycles against 69 at 3.7 -- inside the join's loop, beside the LDS broadcast reads, the three-source v_bitop3_b32
// does not reach its stand-alone rate.  Not the default.
template <int ND, int T, int OFF, int CNT>
__device__ __forceinline__ void joinl_accum_z(uint32_t (&acc)[T][4], const uint32_t (&c)[T][ND], const uint32_t (&q)[CNT], uint32_t k1) {
#pragma unroll
    for (int d = 0; d < CNT; d += 4) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint32_t x[4], u[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) asm("v_xor_b32 %0, %1, %2" : "=v"(x[a]) : "v"(c[t][OFF + d + a]), "v"(q[d + a]));
#pragma unroll
            for (int a = 0; a < 4; ++a) asm("v_sub_u32 %0, %1, %2" : "=v"(u[a]) : "v"(x[a]), "v"(k1));
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[t][a] = __builtin_amdgcn_bitop3_b32(acc[t][a], u[a], x[a], 0xF4);     // a | (b & ~c)
        }
    }
}

template <int T>
__device__ __forceinline__ void joinl_reset_z(uint32_t (&acc)[T][4]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = 0u;
}

template <int T>
__device__ __forceinline__ void joinl_test_z(const uint32_t (&acc)[T][4], int i, int k0, int lane, int z0, int n,
                                             const int* hi_rows, int r, WaveAppender& app) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t f = ((acc[t][0] | acc[t][1]) | (acc[t][2] | acc[t][3])) & 0x80008000u;
        const u64 mm = __ballot(f != 0u);
        if (mm) {
            const int lo = max(i + 1, z0);
            const int hk = min(hi_rows[r], n - 1);
            const int k = k0 + t * kWave;
            app.push(((mm >> lane) & 1ull) && k >= lo && k <= hk, i, k, lane);
        }
    }
}

template <int T>
__device__ __forceinline__ void joinl_reset15(uint32_t (&acc)[T][4]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = 0xFFFFFFFFu;
}

template <int T>
__device__ __forceinline__ void joinl_test15(const uint32_t (&acc)[T][4], int i, int k0, int lane, int z0, int n,
                                             const int* hi_rows, int r, WaveAppender& app) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t f = (acc[t][0] & acc[t][1]) & (acc[t][2] & acc[t][3]) & 0x80008000u;
        const u64 mm = __ballot(f != 0x80008000u);
        if (mm) {
            const int lo = max(i + 1, z0);
            const int hk = min(hi_rows[r], n - 1);
            const int k = k0 + t * kWave;
            app.push(((mm >> lane) & 1ull) && k >= lo && k <= hk, i, k, lane);
        }
    }
}

template <int T>
__device__ __forceinline__ void joinl_reset(us2_t (&acc)[T][4]) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[t][a] = us2_t{0xFFFF, 0xFFFF};
}

// (a hit reads the row's CB cut-off from the block's LDS copy: a global load here stalled the wave for a memory round trip on the
//  ~6 % of rows that have a 16-bit match somewhere in the wave)
template <int T>
__device__ __forceinline__ void joinl_test(const us2_t (&acc)[T][4], int i, int k0, int lane, int z0, int n,
                                           const int* hi_rows, int r, WaveAppender& app) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t mv = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_min(acc[t][0], acc[t][1]),
                                                                                     __builtin_elementwise_min(acc[t][2], acc[t][3])));
        const u64 mm = __ballot(min(mv & 0xFFFFu, mv >> 16) == 0u);
        if (mm) {
            const int lo = max(i + 1, z0);
            const int hk = min(hi_rows[r], n - 1);
            const int k = k0 + t * kWave;
            app.push(((mm >> lane) & 1ull) && k >= lo && k <= hk, i, k, lane);
        }
    }
}

// CH dwords of a query row from the LDS tile: every lane reads the same address (broadcast)
template <int CH>
__device__ __forceinline__ void joinl_load(uint32_t (&q)[CH], const uint32_t* row_chunk) {
    if constexpr (CH % 4 == 0) {
        const uint4* p = reinterpret_cast<const uint4*>(row_chunk);
#pragma unroll
        for (int j = 0; j < CH / 4; ++j) { const uint4 v = p[j]; q[4 * j] = v.x; q[4 * j + 1] = v.y; q[4 * j + 2] = v.z; q[4 * j + 3] = v.w; }
    } else {
#pragma unroll
        for (int j = 0; j < CH; ++j) q[j] = row_chunk[j];
    }
}

#ifdef SELHIP_JOIN_TRACE
// development build only (scripts/join_trace.py): per-wave {start, end} wall-clock ticks, rows compared and hardware id -- a timeline of the join
__device__ unsigned long long g_join_trace[1 << 17][4];
#endif
constexpr int kJoinTilePadRows = 2;        // look-ahead reads past the last staged row stay inside the allocation

// Which (tile of query rows, block of 256 candidates) unit a block of the LDS-tile join takes.  The plain grid is the rectangle
// tiles x candidate blocks, and on the triangle i < k the blocks under the diagonal leave at once -- but they are dispatched first
// (candidate blocks ascending: block column g has only ~g * 256 / qt tiles above the diagonal), 37 700 dead waves of 50 080 at cfg3,
// and the waves that do work start late: the timeline (profiles/r02_join_timeline.txt) shows the last of them starting 55 us into a
// 104 us kernel.  For contiguous rows (one part, first row a multiple of the tile height, tile height dividing 256) column g holds
//     count(g) = min(n_tiles, c0 + a g),   a = 256 / qt,   c0 = a (g_lo + 1) - row_begin / qt
// units, so the grid is launched with exactly the sum of those and a block finds its unit from the closed form of the partial sums
// (a square root and a fix-up).  a == 0: the rectangle (interleaved row blocks of the multi-GPU partition, odd tile heights).
// MEASURED (gpurun_out/r03/d_tri*): cfg3 join 107.0 vs 106.6 us, cfg4 2.009 vs 2.013 ms -- nothing.  The late
// starts of the timeline are not dispatch time lost on dead blocks: 12 400 working wave-units meet 8 192 wave slots, so a third of
// them can only start when a first-round wave ends (~45-55 us in).  Kept behind "join_tri" = 1 (tested), not the default.
struct JoinTriangle {
    int a, c0, K, SK;          // K columns grow by `a` units each (SK units together), the rest hold n_tiles
    __device__ __forceinline__ long long upto(int k) const { return (long long)k * c0 + (long long)a * k * (k - 1) / 2; }
    __device__ __forceinline__ void unit(int b, int n_tiles, int* tile, int* col) const {
        if (a == 0) { *tile = b % n_tiles; *col = b / n_tiles; return; }
        if (b >= SK) { *tile = (b - SK) % n_tiles; *col = K + (b - SK) / n_tiles; return; }
        const double h = (double)c0 - 0.5 * (double)a;
        int k = (int)((sqrt(h * h + 2.0 * (double)a * (double)b) - h) / (double)a);
        k = max(0, min(k, K - 1));
        while (k + 1 < K && upto(k + 1) <= b) ++k;
        while (k > 0 && upto(k) > b) --k;
        *tile = b - (int)upto(k); *col = k;
    }
};

// FORM: 0 = packed minimum (v_xor_b32 + v_pk_min_u16), 1 = 15-bit signatures with flag arithmetic, 2 = 16-bit zero-half test
template <int ND, int T, int WPB, int FORM>
__global__ __launch_bounds__(WPB * kWave)
void sigl_join_kernel(const uint32_t* __restrict__ sigP, const uint32_t* __restrict__ sigG, int n, int n_pad,
                      const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                      RowMap rm, int n_tiles, int group_base, int qt,
                      selhip_int2_t* __restrict__ pre, u64 pre_cap, u64* __restrict__ seg_cnt, int cb_pruned, JoinTriangle tri) {
    extern __shared__ __attribute__((aligned(16))) uint32_t joinl_smem[];
    selhip_int2_t* const app_lds = reinterpret_cast<selhip_int2_t*>(joinl_smem);                  // WPB * kAppendCap records
    int* const hi_lds = reinterpret_cast<int*>(joinl_smem + WPB * kAppendCap * 2);                // hi[] of the tile's rows
    uint32_t* const tile_lds = joinl_smem + WPB * kAppendCap * 2 + ((qt + 3) & ~3);               // (qt + pad) rows x ND dwords
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
#ifdef SELHIP_JOIN_TRACE
    const unsigned long long trace_t0 = wall_clock64();
#endif
    int tile, col;
    tri.unit((int)blockIdx.x, n_tiles, &tile, &col);
    const int grp_b = group_base + col * (WPB * T);                                               // the block's first candidate group
    // (walking the candidate blocks from the highest ranks down, so that the grid tapers off on the short columns, was measured:
    //  cfg3 119 vs 109 us -- kept in ascending order.  So was a grid resident all at once with the valid (tile, candidate block)
    //  units dealt out in equal contiguous shares -- tiles of 16 / 32 / 64 rows double-buffered in LDS, candidates reloaded only when
    //  the block changes: 122-128 us with 2 048 blocks, 116 us with 4 096, bit-identical -- one barrier and one dependent tile
    //  fetch per unit cost more than the uneven tail of the plain grid: profiles/r02_join_timeline.txt)
    // ---- block-uniform part, arithmetic only: which rows can meet this block's candidates at all (half of the grid lies below
    // the diagonal and leaves here without touching memory)
    if (grp_b * kWave >= n) return;
    const int kb_last = (grp_b + WPB * T) * kWave - 1;
    int i_lo, i_end;
    rm.tile_rows(tile, qt, &i_lo, &i_end);
    const int ib_hi = min(i_end, kb_last);                                    // need i < k for some lane of the block
    if (i_lo >= ib_hi) return;
    // (with CB pruning most blocks lie beyond the rows' cut-offs: one dependent load decides that before anything is fetched; in
    //  the all-pairs mode hi = n-1 everywhere and the test is skipped)
    if (cb_pruned && hi[ib_hi - 1] < grp_b * kWave) return;                   // hi is non-decreasing
    // ---- every load of the prologue is issued before anything waits: the wave's candidate signatures, z0, the tile rows and
    // their CB cut-offs are independent of each other (one memory round trip instead of four in a row at the head of a block)
    const int k_base = (grp_b + wave * T) * kWave;
    const int k0 = k_base + lane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(sigP), 0, ND * n_pad * 4, 0x00020000);
    const int row_bytes = n_pad * 4;
    uint32_t c[T][ND];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int kk = min(k0 + t * kWave, n_pad - 1);                        // lanes past the end repeat the last column (never pushed: k > hk)
#pragma unroll
        for (int d = 0; d < ND; ++d) c[t][d] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, kk * 4, d * row_bytes, 0);
    }
    const int z0p1 = pc_in->z0p1;
    {   // stage the rows [i_lo, ib_hi) of sigG: contiguous, 16-byte aligned (ND is a multiple of 4)
        const uint4* src = reinterpret_cast<const uint4*>(sigG + (size_t)i_lo * ND);
        uint4* dst = reinterpret_cast<uint4*>(tile_lds);
        const int n16 = (ib_hi - i_lo) * (ND / 4);
        for (int t = threadIdx.x; t < n16; t += WPB * kWave) dst[t] = src[t];
        for (int t = threadIdx.x; t < ib_hi - i_lo; t += WPB * kWave) hi_lds[t] = hi[i_lo + t];
    }
    __syncthreads();
    // The candidate loads were issued before the tile's, and vector loads return in order: they have all arrived by now.  Say so
    // to the compiler (an empty asm that consumes each register) -- otherwise its wait-count pass, which must assume the loop below
    // can be entered with loads in flight, puts an `s_waitcnt vmcnt(k)` in front of every one of the 32 xors of EVERY row.
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int d = 0; d < ND; ++d) asm volatile("" :: "v"(c[t][d]));
    // ---- per wave (no block-wide synchronisation below)
    const int z0 = z0p1 ? z0p1 - 1 : n;
    if (k_base >= n) return;
    const int k_last = k_base + T * kWave - 1;
    const int i_hi = min(i_end, k_last);
    if (i_lo >= i_hi || k_last < z0) return;
    if (hi_lds[i_hi - 1 - i_lo] < k_base) return;                             // hi is non-decreasing

    WaveAppender app;
    {
        const int seg = blockIdx.x % kAppendSegs;
        const u64 seg_cap = pre_cap / kAppendSegs;
        app.init(app_lds, wave, pre + (size_t)seg * seg_cap, seg_cap, seg_cnt + seg * kSegStride);
    }
    constexpr int CH = ND < 16 ? ND : 16;                                      // dwords per chunk (one register set)
    constexpr int NCH = ND / CH;                                               // chunks per row: 1 (<= 32 bands), 2 (64), 4 (128)
    const int rows = i_hi - i_lo;
    using acc_t = typename std::conditional<FORM != 0, uint32_t, us2_t>::type;
    acc_t acc[T][4];
    uint32_t k7 = FORM == 2 ? 0x00010001u : 0x7FFF7FFFu;
    asm volatile("" : "+v"(k7));                                               // keep the constant in a VGPR (a literal operand is not a plain VOP2)
#define SELHIP_JL_RESET()            do { if constexpr (FORM == 2) joinl_reset_z<T>(acc); else if constexpr (FORM == 1) joinl_reset15<T>(acc); else joinl_reset<T>(acc); } while (0)
#define SELHIP_JL_ACCUM(OFF, Q)      do { if constexpr (FORM == 2) joinl_accum_z<ND, T, OFF, CH>(acc, c, Q, k7); else if constexpr (FORM == 1) joinl_accum15<ND, T, OFF, CH>(acc, c, Q, k7); else joinl_accum<ND, T, OFF, CH>(acc, c, Q); } while (0)
#define SELHIP_JL_TEST(R)            do { if constexpr (FORM == 2) joinl_test_z<T>(acc, i_lo + (R), k0, lane, z0, n, hi_lds, R, app); else if constexpr (FORM == 1) joinl_test15<T>(acc, i_lo + (R), k0, lane, z0, n, hi_lds, R, app); else joinl_test<T>(acc, i_lo + (R), k0, lane, z0, n, hi_lds, R, app); } while (0)
    uint32_t qa[CH], qb[CH];
    joinl_load<CH>(qa, tile_lds);
    if constexpr (NCH == 1) {
        for (int r = 0; r < rows; r += 2) {
            joinl_load<CH>(qb, tile_lds + (r + 1) * ND);
            SELHIP_JL_RESET();
            SELHIP_JL_ACCUM(0, qa);
            SELHIP_JL_TEST(r);
            if (r + 1 >= rows) break;
            joinl_load<CH>(qa, tile_lds + (r + 2) * ND);
            SELHIP_JL_RESET();
            SELHIP_JL_ACCUM(0, qb);
            SELHIP_JL_TEST(r + 1);
        }
    } else {
        for (int r = 0; r < rows; ++r) {
            const uint32_t* row = tile_lds + r * ND;
            SELHIP_JL_RESET();
            joinl_load<CH>(qb, row + CH);
            SELHIP_JL_ACCUM(0, qa);
            if constexpr (NCH == 2) {
                joinl_load<CH>(qa, row + ND);                                  // chunk 0 of the next row
                SELHIP_JL_ACCUM(CH, qb);
            } else {
                joinl_load<CH>(qa, row + 2 * CH);
                SELHIP_JL_ACCUM(CH, qb);
                joinl_load<CH>(qb, row + 3 * CH);
                SELHIP_JL_ACCUM(2 * CH, qa);
                joinl_load<CH>(qa, row + ND);
                SELHIP_JL_ACCUM(3 * CH, qb);
            }
            SELHIP_JL_TEST(r);
        }
    }
#undef SELHIP_JL_RESET
#undef SELHIP_JL_ACCUM
#undef SELHIP_JL_TEST
#ifdef SELHIP_JOIN_TRACE
    if (lane == 0) {
        const unsigned w = blockIdx.x * WPB + wave;
        if (w < (1u << 17)) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            g_join_trace[w][0] = trace_t0; g_join_trace[w][1] = wall_clock64(); g_join_trace[w][2] = (unsigned long long)rows; g_join_trace[w][3] = hwid;
        }
    }
#endif
    app.flush(lane);
}

// verify_kernel: the literal smh_a on every candidate (one lane per candidate), survivors compacted.
__global__ __launch_bounds__(kBlock)
void verify_kernel(const u64* __restrict__ aux, int m, int n_rows, int n_bands,
                   const selhip_int2_t* __restrict__ cand, const u64* __restrict__ n_cand_dev, u64 cand_cap,
                   selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    u64 n_cand = *n_cand_dev;
    if (n_cand > cand_cap) n_cand = cand_cap;
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < n_cand; j += (u64)gridDim.x * kBlock) {
        const selhip_int2_t pr = cand[j];
        if (smh_a_lane(aux + (long long)pr.x * m, aux + (long long)pr.y * m, n_rows, n_bands)) {
            u64 idx = atomicAdd(&pc->n_survivors, 1ull);
            if (idx < surv_cap) surv[idx] = pr;
        }
    }
}

// (Round 3 built and measured the alternative of verifying INSIDE the join: every wave of sigl_join_kernel checked its own 16-bit matches
// at its end -- the candidate registers are free there -- and a light kernel compacted the survivors; bit-identical in all 90 parity tests.
// It lost: cfg3 join 101 -> 128 us for a verification 25 -> 7 us (step 0.250 -> 0.261 ms), cfg2 join 12.5 -> 26.5 us, cfg4 even
// (gpurun_out/r03/f_jv*).  A wave meets ~8 matches, so its own verification is two or three dependent memory round trips for a
// handful of pairs -- 6-8 us added to a 45 us wave that holds one of the 8 192 slots the second round of waves is waiting for -- and
// keeping the row loop at 64 registers beside it cost spills.  The separate kernel stays.)
// verify16_kernel: verification behind the 16-bit join.  A wave takes 64 pairs of the join's output at a time and works
// on them four at a time, 16 lanes per pair (step s: quarter-wave q has pair 4s+q).
//  1. 32-bit signatures: the lanes of a quarter read the two genomes' signature rows from the genome-major copy sigQ
//     (coalesced 16-B loads, 2 x n_bands*4 bytes per pair, L2-resident) and keep one bit per band "32-bit signature
//     equal".  Pairs with a bit set are exactly the candidate set of the 32-bit join (counted in n_candidates).
//  2. A band that is entirely equal has an equal signature, so only bands with a bit set can make smh_a true: the first
//     such band of each pair is compared on the full sketches (n_rows u64 per genome, 16 lanes).  Equal -> the pair
//     survives.  Not equal (a 32-bit hash collision, ~2^-32 per band) -> the pair takes the literal smh_a on one lane.
// All loads of a phase are independent across the steps, so a batch costs a handful of memory round trips instead of the
// ~n_bands dependent ones of the lane-serial literal check (verify_kernel: 32 us for 45 000 candidates at cfg3).
// Recorded alternatives: a separate filter kernel appending the passing pairs to a list (58 us at cfg3 -- one
// single-address atomic per wave, ~85 of those per microsecond); one lane per pair for step 1 (uncoalesced loads: cfg4
// verification 90 -> 370 us); literal check in place on each batch's few passing lanes (cfg4 355 us) or on lanes packed
// through LDS (cfg4 205 us, cfg3 60 us).
// Output: the survivors of a block's 512 pairs (256: 27 us, 512: 24 us, 1024: 26 us at cfg3) are gathered in LDS and appended with ONE global atomic per block and
// batch (plus one for the candidate tally): appends are single-address atomics, ~85 per microsecond on this part, and a
// per-wave append (1 500 waves at cfg3) costs more than the whole check (52 us vs 15 us).
// force_fallback (test hook): treat every first-band comparison as a collision.
constexpr int kVerifyBlock = 512;

__global__ __launch_bounds__(kVerifyBlock)
void verify16_kernel(const u64* __restrict__ aux, int m, int n_rows, int n_bands, const uint32_t* __restrict__ sigQ,
                     const selhip_int2_t* __restrict__ pre_all, const u64* __restrict__ seg_cnt, u64 pre_cap,
                     selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc, int force_fallback,
                     int* __restrict__ row_cnt, int* __restrict__ row_lab, int n) {
    __shared__ selhip_int2_t out_lds[kVerifyBlock];
    __shared__ uint32_t blk_count, blk_cand;
    __shared__ u64 blk_base;
    // the join's output comes in kAppendSegs lists; block b works on list b % kAppendSegs (gridDim.x is a multiple of kAppendSegs)
    const int seg = blockIdx.x % kAppendSegs;
    const u64 seg_cap = pre_cap / kAppendSegs;
    const selhip_int2_t* __restrict__ pre = pre_all + (size_t)seg * seg_cap;
    const u64 n_seg = seg_cnt[seg * kSegStride];
    if (blockIdx.x < kAppendSegs && threadIdx.x == 0 && n_seg) {             // exact totals for the host (overflow test, statistics)
        atomicAdd(&pc->n_pre, n_seg);
        atomicMax(&pc->n_pre_segmax, n_seg);
    }
    const u64 n_pre = n_seg > seg_cap ? seg_cap : n_seg;
    const int lane = threadIdx.x & (kWave - 1);
    const int sub = lane & 15, quarter = lane >> 4, qshift = quarter * 16;
    const int nq = n_bands >> 2;                                              // 16-byte groups per genome (n_bands % 8 == 0, <= 32)
    if (threadIdx.x == 0) { blk_count = 0; blk_cand = 0; }
    __syncthreads();
    for (u64 base = (u64)(blockIdx.x / kAppendSegs) * kVerifyBlock; base < n_pre; base += (u64)(gridDim.x / kAppendSegs) * kVerifyBlock) {
        const u64 j = base + threadIdx.x;
        const bool live = j < n_pre;
        selhip_int2_t pr{0, 0};
        if (live) pr = pre[j];
        const u64 live_mask = __ballot(live);
        u64 has_mask = 0, ok_mask = 0, fb_mask = 0;                           // bit p: pair p of this wave's 64 (wave-uniform)
#pragma unroll 1
        for (int s0 = 0; s0 < 16; s0 += 8) {
            int px[8], py[8];
            uint32_t lm[8];           // bit t (0..3): band 4*sub+t equal; bit 4+t: band 4*(sub+16)+t equal
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int src = (s0 + s) * 4 + quarter;
                px[s] = __shfl(pr.x, src, kWave);
                py[s] = __shfl(pr.y, src, kWave);
                const uint4* a = reinterpret_cast<const uint4*>(sigQ + (long long)px[s] * n_bands);
                const uint4* b = reinterpret_cast<const uint4*>(sigQ + (long long)py[s] * n_bands);
                uint32_t bits = 0;
                if (sub < nq) {
                    const uint4 u = a[sub], v = b[sub];
                    bits |= (u.x == v.x ? 1u : 0u) | (u.y == v.y ? 2u : 0u) | (u.z == v.z ? 4u : 0u) | (u.w == v.w ? 8u : 0u);
                }
                if (sub + 16 < nq) {
                    const uint4 u = a[sub + 16], v = b[sub + 16];
                    bits |= (u.x == v.x ? 16u : 0u) | (u.y == v.y ? 32u : 0u) | (u.z == v.z ? 64u : 0u) | (u.w == v.w ? 128u : 0u);
                }
                lm[s] = ((live_mask >> src) & 1ull) ? bits : 0u;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const uint32_t mine = (uint32_t)(__ballot(lm[s] != 0u) >> qshift) & 0xFFFFu;   // lanes of my quarter with a bit
                const bool has = mine != 0u;
                const int src_sub = has ? __builtin_ctz(mine) : 0;
                const uint32_t lmv = (uint32_t)__shfl((int)lm[s], qshift + src_sub, kWave);
                const int t = has ? __builtin_ctz(lmv) : 0;
                const int band = t < 4 ? src_sub * 4 + t : (src_sub + 16) * 4 + (t - 4);
                const u64* x = aux + (long long)px[s] * m + (long long)band * n_rows;
                const u64* y = aux + (long long)py[s] * m + (long long)band * n_rows;
                bool eq = true;
                for (int j0 = sub; j0 < n_rows; j0 += 16)
                    if (has) eq &= x[j0] == y[j0];
                const bool all_eq = ((uint32_t)(__ballot(eq) >> qshift) & 0xFFFFu) == 0xFFFFu && !force_fallback;
                // one bit per quarter (lanes 0, 16, 32, 48) -> bits 4(s0+s) .. 4(s0+s)+3 of the pair masks
                const u64 mh = __ballot(has && sub == 0), mo = __ballot(has && all_eq && sub == 0);
                const int sh = (s0 + s) * 4;
                has_mask |= (((mh >> 0) & 1ull) | (((mh >> 16) & 1ull) << 1) | (((mh >> 32) & 1ull) << 2) | (((mh >> 48) & 1ull) << 3)) << sh;
                ok_mask |= (((mo >> 0) & 1ull) | (((mo >> 16) & 1ull) << 1) | (((mo >> 32) & 1ull) << 2) | (((mo >> 48) & 1ull) << 3)) << sh;
            }
        }
        fb_mask = has_mask & ~ok_mask;                                        // signature collision: the literal predicate decides
        bool ok = (ok_mask >> lane) & 1ull;
        if ((fb_mask >> lane) & 1ull) ok = smh_a_lane(aux + (long long)pr.x * m, aux + (long long)pr.y * m, n_rows, n_bands);
        const u64 okb = __ballot(ok);
        if (okb) {
            uint32_t wbase = 0;
            if (lane == 0) wbase = atomicAdd(&blk_count, (uint32_t)__popcll(okb));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            if (ok) out_lds[wbase + (uint32_t)__popcll(okb & ((1ull << lane) - 1ull))] = pr;
        }
        if (lane == 0 && has_mask) atomicAdd(&blk_cand, (uint32_t)__popcll(has_mask));
        __syncthreads();
        const uint32_t cnt = blk_count;
        if (threadIdx.x == 0) {
            if (cnt) blk_base = atomicAdd(&pc->n_survivors, (u64)cnt);
            if (blk_cand) atomicAdd(&pc->n_candidates, (u64)blk_cand);
        }
        __syncthreads();
        if (threadIdx.x < cnt) {
            const u64 dst = blk_base + threadIdx.x;
            const selhip_int2_t q = out_lds[threadIdx.x];
            if (dst < surv_cap) {
                surv[dst] = q;
                if (row_cnt) atomicAdd(&row_cnt[q.x], 1);                    // stage 2 grouping: survivors per query row, STORED ones only
                if (row_lab) atomicMax(&row_lab[q.y], n - q.x);              // ... and every row's smallest partner (csr_label_* in kernel_hll.cuh)
            }                                                                //   (like csr_count_kernel: the offsets must stay inside `grouped`)
        }
        __syncthreads();
        if (threadIdx.x == 0) { blk_count = 0; blk_cand = 0; }
        __syncthreads();
    }
}

// =============================================================================================
// ALGO_HASHJOIN -- sub-quadratic candidate generation (NOT the brute-force metric's path):
// instead of comparing every pair's signatures, sort the (band, signature) keys of all genomes (rocPRIM radix sort,
// 32 + log2(NB) key bits) and read the candidates off the runs of equal keys.  A pair that shares t band signatures
// sits in t runs; it is taken from the run of its FIRST matching band only (the bands before it are re-checked on the
// query-major copy), so no de-duplication pass is needed.  Candidates are then verified with the literal smh_a exactly
// as in ALGO_SIG: same candidate set, same survivor set.  Cost O(N*NB log) + output instead of O(N^2 * NB).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void sigkey_build_kernel(const uint32_t* __restrict__ sigT, int n, int n_pad, int nb,
                         u64* __restrict__ keys, int* __restrict__ vals) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)n * nb) return;
    const int b = (int)(t / n), g = (int)(t % n);
    keys[t] = ((u64)b << 32) | sigT[(long long)b * n_pad + g];
    vals[t] = g;
}

__global__ __launch_bounds__(kBlock)
void run_emit_kernel(const u64* __restrict__ keys, const int* __restrict__ vals, long long total,
                     const uint32_t* __restrict__ sigQ, int nb, const u64* __restrict__ aux, int m, int n_rows, int n_bands,
                     int n, const int* __restrict__ hi, const PassCounters* __restrict__ pc_in, RowMap rm,
                     selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int z0 = pc_in->z0p1 ? pc_in->z0p1 - 1 : n;
    WaveAppender app;
    app.init(app_lds, wave, surv, surv_cap, &pc->n_survivors);
    int n_cand = 0;
    for (long long base = (long long)blockIdx.x * kBlock; base < total; base += (long long)gridDim.x * kBlock) {
        const long long p = base + threadIdx.x;
        const bool live = p < total;
        const u64 key = live ? keys[p] : 0;
        const int g = live ? vals[p] : 0;
        const int b = (int)(key >> 32);
        for (long long step = 1;; ++step) {                       // walk the run of equal keys that starts after p
            const bool act = live && p + step < total && keys[p + step] == key;
            if (!__any(act)) break;
            bool cand = false;
            int i = 0, k = 0;
            if (act) {
                const int g2 = vals[p + step];
                i = min(g, g2); k = max(g, g2);
                cand = i != k && rm.owns(i) && k >= max(i + 1, z0) && k <= min(hi[i], n - 1);
                if (cand) {                                        // take the pair from its first matching band only
                    const uint32_t* qi = sigQ + (long long)i * nb;
                    const uint32_t* qk = sigQ + (long long)k * nb;
                    for (int bb = 0; bb < b && cand; ++bb) cand = qi[bb] != qk[bb];
                }
            }
            n_cand += (int)__popcll(__ballot(cand));
            const bool ok = cand && smh_a_lane(aux + (long long)i * m, aux + (long long)k * m, n_rows, n_bands);
            app.push(ok, i, k, lane);
        }
    }
    app.flush(lane);
    if (lane == 0 && n_cand) atomicAdd(&pc->n_candidates, (u64)n_cand);
}

}  // namespace
