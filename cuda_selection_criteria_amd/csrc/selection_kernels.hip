// selection_kernels.hip -- gfx950 (MI355X, CDNA4, wave64) kernels and the C ABI of libselhip.so.
//
// Replaces, for the all-pairs sketch-selection path of sanhue903/CUDA_Selection_Criteria:
//   src/selection_kernels.cu:13-177 (kernel_smh, kernel_CBsmh, launchers)
//   include/criteria_sketch_cuda.cuh:11-65 (device CB / smh_a / hll_union_card)
// with the RESULT SEMANTICS of the CPU path src/selection.cpp:270-291 (see include/selection_hip.h).
//
// Kernels (all integer except the estimator):
//   cb_bounds_kernel        e_i = (size_t)card_i, CB cut-off hi(i) by binary search (CB is monotone on
//                           sorted cards, so the reference's `break` == a per-row upper bound)
//   smh_stream_kernel       stage 1, "stream" algorithm: a tile of Q query sketches is staged in LDS and
//                           held in VGPRs, candidate sketches are streamed row-major with 16 B/lane
//                           coalesced loads, v_cmp_eq_u64 lane masks are folded on the scalar unit into
//                           the band predicate of criteria_sketch.hpp:66-81
//   smh_generic_kernel      any (m, n_rows, n_bands): lane-per-candidate, used for m < 128 or odd shapes
//   hll_union_hist_kernel   stage 2a: per surviving pair, histogram of max(reg_i, reg_k) (hll.h:1188-1204)
//   ertl_select_kernel      stage 2b: lane-per-pair Ertl MLE (hll.h:629-688) + Jaccard test (selection.cpp:286-288)
//   pairlist_kernel         self-contained wave-per-pair path behind the drop-in launch_kernel_* entry points
//   synth_kernel            synthetic sketches generated in HBM (csrc/synth.hpp)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see csrc/Makefile).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/selection_hip.h"
#include "ertl_mle.hpp"
#include "synth.hpp"

namespace {

using u64 = unsigned long long;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kChunk = 256;            // candidates per stage-1 block
constexpr int kQueryVgprBudget = 32;   // u64x2 query registers per lane  (Q * NCH)

// ---------------------------------------------------------------------------------------------
// device-side counters of one pass
// ---------------------------------------------------------------------------------------------
struct PassCounters {
    u64 n_survivors;     // stage-1 survivors appended (may exceed capacity: exact count, stores clipped)
    u64 n_results;       // selected pairs appended (same convention)
    u64 n_evaluated;     // pairs inside the (triangular / CB-banded) pair space of this pass
    u64 n_candidates;    // ALGO_SIG: signature-join candidates
    u64 n_aux_in;        // pairs handed to the auxiliary-HLL criterion (hll_a / hll_an)
    u64 n_final;         // pairs handed to the final HLL-14 Jaccard stage
    int z0;              // first rank with e != 0
    int unsorted;        // set if cards are not ascending
    int pad[2];
};


// ---------------------------------------------------------------------------------------------
// WaveAppender: per-wave staging of output records in LDS, flushed with ONE global atomic per >= 64 records.
// A returning atomic on a single address sustains only ~90 operations/us chip-wide (MI355X_MICROARCH.md,
// row "dequeue"), so appending survivors one atomicAdd at a time caps a pass at ~90 survivors/us
// (45 000 survivors = 0.5 ms -- measured: it was THE cost of the first signature-join kernels).
// ---------------------------------------------------------------------------------------------
constexpr int kAppendCap = 2 * kWave;          // count < 64 before a push, a push adds <= 64

struct WaveAppender {
    selhip_int2_t* buf;        // this wave's LDS staging area [kAppendCap]
    int count;                 // wave-uniform
    selhip_int2_t* out;
    u64 out_cap;
    u64* out_count;

    __device__ __forceinline__ void init(selhip_int2_t* lds_block, int wave, selhip_int2_t* o, u64 cap, u64* cnt) {
        buf = lds_block + wave * kAppendCap; count = 0; out = o; out_cap = cap; out_count = cnt;
    }
    __device__ __forceinline__ void flush(int lane) {
        if (count == 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        u64 base = 0;
        if (lane == 0) base = atomicAdd(out_count, (u64)count);
        base = ((u64)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int t = lane; t < count; t += kWave)
            if (base + (u64)t < out_cap) out[base + (u64)t] = buf[t];
        count = 0;
    }
    // lanes with pred push (x,y); the call must be wave-uniformly reached
    __device__ __forceinline__ void push(bool pred, int x, int y, int lane) {
        const u64 m = __ballot(pred);
        if (m == 0) return;
        if (pred) {
            const int off = count + (int)__popcll(m & ((1ull << lane) - 1ull));
            buf[off].x = x; buf[off].y = y;
        }
        count += (int)__popcll(m);
        if (count >= kWave) flush(lane);
    }
    // wave-uniform single record
    __device__ __forceinline__ void push_uniform(int x, int y, int lane) {
        if (lane == 0) { buf[count].x = x; buf[count].y = y; }
        count += 1;
        if (count >= kWave) flush(lane);
    }
};

// ---------------------------------------------------------------------------------------------
// cb_bounds_kernel: one thread per genome rank.
//   ecard[i] = (size_t)cards[i]                                         (selection.cpp:275,280)
//   hi[i]    = last k such that CB(tau, e_i, e_k) holds, or N-1 without CB  (criteria_sketch.hpp:45-49;
//              the loop `break`s at the first failing k (selection.cpp:282-283); e is ascending so the
//              predicate is monotone and the break is exactly "k <= hi(i)")
//   z0       = first rank with e != 0  (`if(e2 == 0) continue`, selection.cpp:281)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cb_pred(double tau, u64 e1, u64 e2) {
    double gamma = (double)e1 / (double)e2;      // criteria_sketch.hpp:47 (size_t -> double, IEEE divide)
    return gamma >= tau;
}

__global__ void cb_bounds_kernel(const double* __restrict__ cards, int n, double tau, int use_cb,
                                 int row_begin, int row_end, u64* __restrict__ ecard, int* __restrict__ hi,
                                 PassCounters* __restrict__ pc) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c = cards[i];
    u64 e1 = selhip::trunc_card(c);
    ecard[i] = e1;
    if (i > 0) {
        double cp = cards[i - 1];
        if (c < cp) pc->unsorted = 1;
        if (e1 != 0 && selhip::trunc_card(cp) == 0) pc->z0 = i;
    } else if (e1 != 0) {
        pc->z0 = 0;
    }
    int h = n - 1;
    if (use_cb) {
        // largest k in (i, n) with (e_k == 0 || CB(e1, e_k)); predicate is true on a prefix
        int lo = i, hi_ = n - 1;      // invariant: pred(lo) true (k = i itself counts as true), answer in [lo, hi_]
        while (lo < hi_) {
            int mid = lo + (hi_ - lo + 1) / 2;
            u64 e2 = selhip::trunc_card(cards[mid]);
            bool ok = (e2 == 0) || cb_pred(tau, e1, e2);
            if (ok) lo = mid; else hi_ = mid - 1;
        }
        h = lo;
    }
    hi[i] = h;
    if (i >= row_begin && i < row_end) {
        // pairs of this row inside the pair space: k in [max(i+1, z0'), h]; z0 may not be published yet,
        // so count candidates with e_k != 0 directly from the sorted property: e_k == 0 only for k < z0.
        // first k > i with e_k != 0: if e1 != 0 it is i+1, else binary search.
        int first = i + 1;
        if (e1 == 0) {
            int lo = i + 1, hi2 = n;          // first index in [i+1, n) with e != 0
            while (lo < hi2) {
                int mid = lo + (hi2 - lo) / 2;
                if (selhip::trunc_card(cards[mid]) != 0) hi2 = mid; else lo = mid + 1;
            }
            first = lo;
        }
        long long cnt = (long long)h - first + 1;
        if (cnt > 0) atomicAdd(&pc->n_evaluated, (u64)cnt);
    }
}

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
                       int row_begin, int row_end, int n_tiles, int chunk_base,
                       selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    constexpr int Q = kQueryVgprBudget / NCH;
    constexpr int ROWV = NCH * kWave;                 // u64x2 per sketch row
    __shared__ u64x2 qs[Q * ROWV];
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];

    const int tile = blockIdx.x % n_tiles;
    const int chunk = blockIdx.x / n_tiles;
    const int i0 = row_begin + tile * Q;
    const int i_last = min(i0 + Q, row_end) - 1;
    const int z0 = pc_in->z0;
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
                if (i < row_end && k > i && k >= z0 && k <= hi[i]) app.push_uniform(i, k, lane);
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
                        int row_begin, int row_end, int n_rows_grid,
                        selhip_int2_t* __restrict__ surv, u64 surv_cap, PassCounters* __restrict__ pc) {
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];
    const int i = row_begin + (int)(blockIdx.x % n_rows_grid);
    const int chunk = blockIdx.x / n_rows_grid;
    if (i >= row_end) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int z0 = pc_in->z0;
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
// mixes with xor-shuffles (r <= 64) -- or one thread walks the band (r > 64).  Writes both layouts:
//   sigQ[g][NB] (query-major, read with scalar loads) and sigT[b][n_pad] (band-major, lane = candidate).
__global__ __launch_bounds__(kBlock)
void sig_build_kernel(const u64* __restrict__ aux, int n, int m, int r, int nb, int n_pad,
                      uint32_t* __restrict__ sigQ, uint32_t* __restrict__ sigT) {
    if (r <= kWave) {
        const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;      // global bucket index
        const long long total = (long long)n * m;
        u64 h = 0;
        int j = 0;
        if (t < total) {
            j = (int)(t % r);                                                  // position inside the band
            h = mix64(aux[t] + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
        }
        for (int s = 1; s < r; s <<= 1) {                                      // r is a power of two here
            h += __shfl_xor(h, s, kWave);
        }
        if (t < total && j == 0) {
            const int g = (int)(t / m);
            const int b = (int)((t % m) / r);
            const uint32_t sig = (uint32_t)(h ^ (h >> 32));
            sigQ[(long long)g * nb + b] = sig;
            sigT[(long long)b * n_pad + g] = sig;
        }
    } else {
        const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;      // (genome, band)
        if (t >= (long long)n * nb) return;
        const int g = (int)(t / nb), b = (int)(t % nb);
        const u64* v = aux + (long long)g * m + (long long)b * r;
        u64 h = 0;
        for (int j = 0; j < r; ++j) h += mix64(v[j] + 0x9E3779B97F4A7C15ull * (u64)(j + 1));
        const uint32_t sig = (uint32_t)(h ^ (h >> 32));
        sigQ[(long long)g * nb + b] = sig;
        sigT[(long long)b * n_pad + g] = sig;
    }
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
    // two independent chains of v_min3_u32(acc, x, y): 1.5 VALU per band
    uint32_t acc0 = 0xFFFFFFFFu, acc1 = 0xFFFFFFFFu;
#pragma unroll
    for (int b = 0; b < NB; b += 4) {
        acc0 = min(min(acc0, c[b] ^ dpp_row_bcast<J>(qv[b])), c[b + 1] ^ dpp_row_bcast<J>(qv[b + 1]));
        acc1 = min(min(acc1, c[b + 2] ^ dpp_row_bcast<J>(qv[b + 2])), c[b + 3] ^ dpp_row_bcast<J>(qv[b + 3]));
    }
    const u64 mm = __ballot(min(acc0, acc1) == 0u);
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
                     int row_begin, int row_end, int n_tiles, int group_base, int qt,
                     selhip_int2_t* __restrict__ cand, u64 cand_cap, PassCounters* __restrict__ pc) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int tile = blockIdx.x % n_tiles;
    const int grp = group_base + (blockIdx.x / n_tiles) * kWavesPerBlock + wave;
    const int k_base = grp * kWave;
    if (k_base >= n) return;
    const int z0 = pc_in->z0;
    const int k_last = k_base + kWave - 1;
    const int i_lo = row_begin + tile * qt;                                   // qt is a multiple of 16
    const int i_hi = min(min(i_lo + qt, row_end), k_last);                    // need i < k for some lane
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

// ---------------------------------------------------------------------------------------------
// hll_union_hist_kernel: one wave per pair.  LDS holds a lane-private 64-bin histogram per wave
// ([bin][lane], conflict-free ds_add_u32), reduced with a rotated column walk.
// counts[j][0..63] = #registers whose max(reg_x, reg_y) equals the bin    (hll.h:1188-1204)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t max_u8x4(uint32_t a, uint32_t b) {
    // per-byte unsigned max without carries between bytes
    uint32_t r = 0;
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        uint32_t x = (a >> s) & 0xFF, y = (b >> s) & 0xFF;
        r |= (x > y ? x : y) << s;
    }
    return r;
}

__device__ __forceinline__ void hist_add_word(uint32_t* __restrict__ col, uint32_t w) {
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        uint32_t v = (w >> s) & 63u;              // register values are <= 64-p+1 < 64
        __hip_atomic_fetch_add(col + v * kWave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ __launch_bounds__(kBlock)
void hll_union_hist_kernel(const uint8_t* __restrict__ hll, int p,
                           const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ n_pairs_dev,
                           u64 n_pairs_host, u64 cap, uint32_t* __restrict__ counts,
                           u64 chunk_off = 0, u64 chunk_len = ~0ull) {
    __shared__ uint32_t hist[kWavesPerBlock][64 * kWave];     // 64 KiB
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    u64 n_pairs = n_pairs_dev ? *n_pairs_dev : n_pairs_host;
    if (n_pairs > cap) n_pairs = cap;
    // optional window [chunk_off, chunk_off + chunk_len) of the list; counts are indexed from the window start
    n_pairs = n_pairs > chunk_off ? min(n_pairs - chunk_off, chunk_len) : 0;
    pairs += chunk_off;
    const long long nreg = 1ll << p;
    uint32_t* my = hist[wave];
    uint32_t* col = my + lane;

    for (u64 base = (u64)blockIdx.x * kWavesPerBlock; base < n_pairs; base += (u64)gridDim.x * kWavesPerBlock) {
        const u64 j = base + wave;
        const bool active = j < n_pairs;
        // zero this wave's histogram
#pragma unroll 8
        for (int b = 0; b < 64; ++b) col[b * kWave] = 0;
        __syncthreads();
        if (active) {
            const selhip_int2_t pr = pairs[j];
            const uint8_t* a = hll + (long long)pr.x * nreg;
            const uint8_t* b = hll + (long long)pr.y * nreg;
            if (nreg == 16384) {
                // p = 14: both rows (2 x 16 KiB) are requested up front -- 32 x 16-B loads in flight per lane --
                // before any LDS work starts; the kernel is bound by bytes in flight otherwise
                const uint4* a4 = reinterpret_cast<const uint4*>(a);
                const uint4* b4 = reinterpret_cast<const uint4*>(b);
                uint4 xa[16], xb[16];
#pragma unroll
                for (int it = 0; it < 16; ++it) { xa[it] = a4[it * kWave + lane]; xb[it] = b4[it * kWave + lane]; }
#pragma unroll
                for (int it = 0; it < 16; ++it) {
                    hist_add_word(col, max_u8x4(xa[it].x, xb[it].x));
                    hist_add_word(col, max_u8x4(xa[it].y, xb[it].y));
                    hist_add_word(col, max_u8x4(xa[it].z, xb[it].z));
                    hist_add_word(col, max_u8x4(xa[it].w, xb[it].w));
                }
            } else if (nreg >= 1024) {
                const uint4* a4 = reinterpret_cast<const uint4*>(a);
                const uint4* b4 = reinterpret_cast<const uint4*>(b);
                const int iters = (int)(nreg / (16 * kWave));
                for (int it = 0; it < iters; ++it) {
                    uint4 x = a4[it * kWave + lane], y = b4[it * kWave + lane];
                    hist_add_word(col, max_u8x4(x.x, y.x));
                    hist_add_word(col, max_u8x4(x.y, y.y));
                    hist_add_word(col, max_u8x4(x.z, y.z));
                    hist_add_word(col, max_u8x4(x.w, y.w));
                }
            } else {
                for (long long t = lane; t < nreg; t += kWave) {
                    uint32_t x = a[t], y = b[t];
                    uint32_t v = (x > y ? x : y) & 63u;
                    __hip_atomic_fetch_add(col + v * kWave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
        __syncthreads();
        if (active) {
            // lane l sums bin l over the 64 lane-columns, rotated so that lanes hit distinct banks
            uint32_t s = 0;
            const uint32_t* rowp = my + lane * kWave;
#pragma unroll 8
            for (int t = 0; t < kWave; ++t) s += rowp[(t + lane) & (kWave - 1)];
            counts[j * 64 + lane] = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// ertl_select_kernel: one LANE per histogram.  The 64 histograms of a wave are staged in LDS
// (pitch 65 -> conflict-free) because the estimator indexes them with run-time k.
//   MODE 0: est[j] = estimate                                     (selhip_ertl_estimate, cards)
//   MODE 1: t = estimate; J = ((double)e_x + (double)e_y - t)/t; if (J >= tau) append   (selection.cpp:286-288)
//   MODE 2: like 1 but writes selhip_result_t{x,y,(float)J} (drop-in launchers)
// ---------------------------------------------------------------------------------------------
struct LdsCounts {
    const uint32_t* base;       // &lds[lane]
    __device__ __forceinline__ uint32_t operator[](int k) const { return base[k * 65]; }
};

template <bool FMA, int MODE>
__global__ __launch_bounds__(kWave)
void ertl_select_kernel(const uint32_t* __restrict__ counts, const u64* __restrict__ n_dev, u64 n_host, u64 cap,
                        int p, double relerr_scaled,
                        double* __restrict__ est,
                        const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ ecard, double tau,
                        selhip_pair_t* __restrict__ results, u64 results_cap, PassCounters* __restrict__ pc,
                        selhip_result_t* __restrict__ results_f32, int* __restrict__ out_count_i32,
                        u64 chunk_off, u64 chunk_len) {
    __shared__ uint32_t lds[64 * 65];
    const int lane = threadIdx.x;
    u64 n = n_dev ? *n_dev : n_host;
    if (n > cap) n = cap;
    n = n > chunk_off ? min(n - chunk_off, chunk_len) : 0;       // window of the list; counts indexed from its start
    if (pairs) pairs += chunk_off;
    if (est) est += chunk_off;
    for (u64 base = (u64)blockIdx.x * kWave; base < n; base += (u64)gridDim.x * kWave) {
        __syncthreads();
        // row r of the tile = histogram base+r; lane = bin -> coalesced 256 B reads
        for (int r = 0; r < kWave; ++r) {
            u64 j = base + r;
            uint32_t v = (j < n) ? counts[j * 64 + lane] : (lane == 0 ? (1u << p) : 0u);
            lds[lane * 65 + r] = v;
        }
        __syncthreads();
        const u64 j = base + lane;
        LdsCounts c{lds + lane};
        double t = selhip::ertl_ml_estimate<FMA>(c, (unsigned)p, (unsigned)(64 - p), relerr_scaled);
        if (j < n) {
            if constexpr (MODE == 0) {
                est[j] = t;
            } else {
                const selhip_int2_t pr = pairs[j];
                const double e1 = (double)ecard[pr.x], e2 = (double)ecard[pr.y];
                const double jacc = (e1 + e2 - t) / t;                       // selection.cpp:287
                if (jacc >= tau) {                                           // selection.cpp:288
                    if constexpr (MODE == 1) {
                        u64 idx = atomicAdd(&pc->n_results, 1ull);
                        if (idx < results_cap) { results[idx].i = pr.x; results[idx].k = pr.y; results[idx].jaccard = jacc; }
                    } else {
                        int idx = atomicAdd(out_count_i32, 1);
                        results_f32[idx].x = pr.x; results_f32[idx].y = pr.y; results_f32[idx].sim = (float)jacc;
                    }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Auxiliary-HLL criteria (src/selection.cpp:152-173 hll_a, :206-227 hll_an; criteria_sketch.hpp:22-64).
// enum_pairs_kernel lists the (CB-pruned) pair space of the rows explicitly -- only used when hll_a / hll_an
// is the FIRST criterion; in the two-stage form (BASELINE config 5) the cheap smh_a join runs first and the
// auxiliary criterion sees its survivors only: the selected set is the intersection either way.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void enum_pairs_kernel(int n, const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                       int row_begin, int row_end, int n_rows_grid,
                       selhip_int2_t* __restrict__ out, u64 out_cap, PassCounters* __restrict__ pc) {
    __shared__ selhip_int2_t app_lds[kWavesPerBlock * kAppendCap];
    const int i = row_begin + (int)(blockIdx.x % n_rows_grid);
    const int chunk = blockIdx.x / n_rows_grid;
    if (i >= row_end) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int kmin = max(i + 1, pc_in->z0);
    const int k = kmin + chunk * kBlock + (int)threadIdx.x;
    WaveAppender app;
    app.init(app_lds, wave, out, out_cap, &pc->n_aux_in);
    app.push(k <= hi[i] && k < n, i, k, lane);
    app.flush(lane);
}

// aux_filter_kernel<FMA, CRIT>: one LANE per pair; counts = union histogram of the two AUXILIARY sketches.
//   CRIT 1 (hll_a):  t_hat = (size_t)U;  t+ = t_hat / (1 + Z*sigma_p);  K+ = ((1+gamma)*e_k - t+)/t+ >= tau
//   CRIT 2 (hll_an): J = ((double)(e_i+e_k) - U)/U;  C = min(1, (1+Z*sigma_p)*e_k/U) * (1+gamma) * S;  J + C >= tau
// zs = (double)(float)(Z*sigma_p) and S (= zs for order_n = 1) are computed on the host in float/double exactly as
// criteria_sketch.hpp:7-20,25-31,39-40 do.  FMA flavour: g++ fuses (1+gamma)*card_B - t_hat_mas (criteria_sketch.hpp:41).
template <bool FMA, int CRIT>
__global__ __launch_bounds__(kWave)
void aux_filter_kernel(const uint32_t* __restrict__ counts, const selhip_int2_t* __restrict__ pairs,
                       const u64* __restrict__ n_dev, u64 chunk_off, u64 chunk_len, u64 cap,
                       int p_aux, double relerr_scaled, const u64* __restrict__ ecard, double tau,
                       double zs, double S_sum,
                       selhip_int2_t* __restrict__ out, u64 out_cap, u64* __restrict__ out_count) {
    __shared__ uint32_t lds[64 * 65];
    __shared__ selhip_int2_t app_lds[kAppendCap];
    const int lane = threadIdx.x;
    u64 total = *n_dev;
    if (total > cap) total = cap;
    const u64 n = total > chunk_off ? min(total - chunk_off, chunk_len) : 0;     // pairs of this chunk
    WaveAppender app;
    app.init(app_lds, 0, out, out_cap, out_count);
    for (u64 base = (u64)blockIdx.x * kWave; base < n; base += (u64)gridDim.x * kWave) {
        __syncthreads();
        for (int r = 0; r < kWave; ++r) {
            u64 j = base + r;
            lds[lane * 65 + r] = (j < n) ? counts[j * 64 + lane] : (lane == 0 ? (1u << p_aux) : 0u);
        }
        __syncthreads();
        const u64 j = base + lane;
        LdsCounts c{lds + lane};
        const double U = selhip::ertl_ml_estimate<FMA>(c, (unsigned)p_aux, (unsigned)(64 - p_aux), relerr_scaled);
        bool sel = false;
        selhip_int2_t pr{0, 0};
        if (j < n) {
            pr = pairs[chunk_off + j];
            const u64 ea = ecard[pr.x], eb = ecard[pr.y];
            const double gamma = (double)ea / (double)eb;                         // criteria_sketch.hpp:24,38
            if constexpr (CRIT == 1) {
                const double t_hat = (double)(u64)(long long)U;                   // size_t t_hat = union_size()  (:61)
                const double t_mas = t_hat / (1.0 + zs);                          // :40
                const double K = selhip::muladd<FMA>(1.0 + gamma, (double)eb, -t_mas) / t_mas;   // :41
                sel = K >= tau;                                                   // :63
            } else {
                const double J = ((double)(ea + eb) - U) / U;                     // :55
                const double candv = (1.0 + zs) * (double)eb / U;                 // :32
                const double minimo = candv < 1.0 ? candv : 1.0;                  // std::min(1.0, .)
                const double C = minimo * (1 + gamma) * S_sum;                    // :33
                sel = (J + C) >= tau;                                             // :57
            }
        }
        app.push(sel, pr.x, pr.y, lane);
    }
    app.flush(lane);
}

// ---------------------------------------------------------------------------------------------
// explicit pair lists (drop-in launchers and test building blocks): one LANE per pair.
//   flags[j] = pair passes [e_y != 0] [CB] smh_a ;  optionally compacts survivors.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void pairlist_smh_kernel(const u64* __restrict__ aux, int m, int n_rows, int n_bands,
                         const selhip_int2_t* __restrict__ pairs, long long n_pairs,
                         const double* __restrict__ cards, double tau, int check_cards, int use_cb,
                         uint8_t* __restrict__ flags,
                         selhip_int2_t* __restrict__ surv, u64 surv_cap, u64* __restrict__ surv_count) {
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pairs) return;
    const selhip_int2_t pr = pairs[j];
    bool ok = true;
    if (check_cards) {
        const u64 e1 = selhip::trunc_card(cards[pr.x]), e2 = selhip::trunc_card(cards[pr.y]);
        if (e2 == 0) ok = false;                                             // selection.cpp:281
        else if (use_cb && !cb_pred(tau, e1, e2)) ok = false;                // selection.cpp:282
    }
    if (ok) ok = smh_a_lane(aux + (long long)pr.x * m, aux + (long long)pr.y * m, n_rows, n_bands);
    if (flags) flags[j] = ok ? 1 : 0;
    if (ok && surv) {
        u64 idx = atomicAdd(surv_count, 1ull);
        if (idx < surv_cap) surv[idx] = pr;
    }
}

__global__ __launch_bounds__(kBlock)
void match_count_kernel(const u64* __restrict__ aux, int m, const selhip_int2_t* __restrict__ pairs,
                        long long n_pairs, int32_t* __restrict__ matches) {
    // one wave per pair: lanes stride the buckets, v_cmp_eq_u64 masks counted with s_bcnt1
    const int lane = threadIdx.x & (kWave - 1);
    long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (j >= n_pairs) return;
    const selhip_int2_t pr = pairs[j];
    const u64* a = aux + (long long)pr.x * m;
    const u64* b = aux + (long long)pr.y * m;
    int cnt = 0;
    for (int t0 = 0; t0 < m; t0 += kWave) {
        int t = t0 + lane;
        bool eq = (t < m) && (a[t] == b[t]);
        cnt += __popcll(__ballot(eq));
    }
    if (lane == 0) matches[j] = cnt;
}

__global__ void truncate_cards_kernel(const double* __restrict__ cards, int n, u64* __restrict__ ecard) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ecard[i] = selhip::trunc_card(cards[i]);
}

__global__ void iota_pairs_kernel(selhip_int2_t* pairs, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { pairs[i].x = i; pairs[i].y = i; }
}

// ---------------------------------------------------------------------------------------------
// synth_kernel: one block per genome; registers / buckets are built in LDS with ds_max / ds_min.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void synth_kernel(selhip::SynthParams sp, long long g_begin, long long g_end,
                  uint8_t* __restrict__ hll, u64* __restrict__ aux, uint8_t* __restrict__ aux_hll) {
    extern __shared__ unsigned char smem_raw[];
    // layout: u64 smh[m] | u32 regs[16384] | u32 aregs[1<<p_aux]
    u64* smh = reinterpret_cast<u64*>(smem_raw);
    uint32_t* regs = reinterpret_cast<uint32_t*>(smem_raw + (size_t)sp.m * 8);
    uint32_t* aregs = regs + 16384;
    const int n_aux = sp.p_aux > 0 ? (1 << sp.p_aux) : 0;

    const long long g = g_begin + blockIdx.x;
    if (g >= g_end) return;
    for (int t = threadIdx.x; t < sp.m; t += kBlock) smh[t] = ~0ull;
    for (int t = threadIdx.x; t < 16384; t += kBlock) regs[t] = 0;
    for (int t = threadIdx.x; t < n_aux; t += kBlock) aregs[t] = 0;
    __syncthreads();

    const uint32_t cluster = (uint32_t)(g / sp.cluster_size);
    const uint32_t n_sh = selhip::synth_n_shared(sp, cluster);
    const uint32_t n_pr = selhip::synth_n_private(sp, (uint32_t)g, n_sh);
    for (int part = 0; part < 2; ++part) {
        const uint32_t cnt = part == 0 ? n_sh : n_pr;
        const u64 stream = part == 0 ? 2ull * cluster : 2ull * (u64)g + 1;
        for (uint32_t e = threadIdx.x; e < cnt; e += kBlock) {
            const u64 h = selhip::synth_element(sp, stream, e);
            uint32_t idx, rank;
            selhip::synth_hll_slot(h, 14, &idx, &rank);
            atomicMax(&regs[idx], rank);
            if (n_aux) {
                selhip::synth_hll_slot(h, sp.p_aux, &idx, &rank);
                atomicMax(&aregs[idx], rank);
            }
            uint32_t bucket; uint64_t value;
            selhip::synth_smh_slot(h, sp.m, &bucket, &value);
            atomicMin(&smh[bucket], (u64)value);
        }
    }
    __syncthreads();
    const long long r = g - g_begin;
    for (int t = threadIdx.x; t < sp.m; t += kBlock) aux[r * sp.m + t] = smh[t];
    uint32_t* out32 = reinterpret_cast<uint32_t*>(hll + r * 16384);
    for (int t = threadIdx.x; t < 16384 / 4; t += kBlock)
        out32[t] = regs[4 * t] | (regs[4 * t + 1] << 8) | (regs[4 * t + 2] << 16) | (regs[4 * t + 3] << 24);
    if (n_aux && aux_hll)
        for (int t = threadIdx.x; t < n_aux; t += kBlock) aux_hll[r * n_aux + t] = (uint8_t)aregs[t];
}


// =============================================================================================
// Sketch construction on the GPU (reference: src/build_sketch.cpp:26-151).
//   input : per genome, its FASTA records as one byte per base: 0..3 = A,C,G,T (either case), 4 = anything that
//           resets the k-mer window (non-ACGT character, record boundary)        (build_sketch.cpp:68-84)
//   output: HLL p=14 registers (hll.h:886-904 add/addh with WangHash, hash.h:42-53), auxiliary HLL p_aux registers,
//           SuperMinHash h_[m] (bbmh.h:639-670)
// SuperMinHash in parallel.  The reference's addh is sequential per sketch, but its RESULT is order-free: for an
// element e the draws (k_j, r_j) come from a generator seeded with e alone, step j swaps p[k_j] <-> p[j] in a
// permutation that starts as the identity for every element, and bucket p[j] is offered the value (j << 32) | r_j;
// h[bucket] keeps the minimum.  The running bound a_ (largest integer part still present) only SKIPS offers that
// cannot win.  Hence h = min over all elements and all steps j <= a_final, and it is computed here as
//   pass 0: every k-mer in parallel offers its step-0 value (bucket k_0) with a 64-bit LDS atomic min;
//   while a = max_b min(m-1, h[b] >> 32) exceeds the number of steps J offered so far: J = a and every k-mer re-runs
//   its own chain up to step J (a handful of entries of p, kept in a tiny per-thread map) -- or, when a is large
//   (few k-mers per bucket: tiny inputs), ONE lane runs the reference's sequential algorithm literally.
// Either way the bytes equal the reference's (tests/test_build_sketch.py: 128 reference-written files).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ u64 canonical_kmer(u64 kmer, unsigned k) {           // build_sketch.cpp:26-39
    const u64 b_kmer = kmer;
    kmer = ((kmer >> 2) & 0x3333333333333333ull) | ((kmer & 0x3333333333333333ull) << 2);
    kmer = ((kmer >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((kmer & 0x0F0F0F0F0F0F0F0Full) << 4);
    kmer = ((kmer >> 8) & 0x00FF00FF00FF00FFull) | ((kmer & 0x00FF00FF00FF00FFull) << 8);
    kmer = ((kmer >> 16) & 0x0000FFFF0000FFFFull) | ((kmer & 0x0000FFFF0000FFFFull) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    const u64 reverse = (~0ull - kmer) >> (64 - (k << 1));
    return b_kmer < reverse ? b_kmer : reverse;
}
__host__ __device__ __forceinline__ u64 wang_hash(u64 key) {                              // hash.h:42-53
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}
__device__ __forceinline__ u64 wyhash64_next(u64& state) {                               // aesctr/wy.h:44-59
    state += 0x60bee2bee120fc15ull;
    const u64 x = state ^ 0xe7037ed1a0b428dbull, y = state;
    return (x * y) ^ __umul64hi(x, y);
}
__device__ __forceinline__ void hll_slot(u64 h, int p, uint32_t* idx, uint32_t* rank) {   // hll.h:886-888
    *idx = (uint32_t)(h >> (64 - p));
    *rank = (uint32_t)__clzll((long long)(((h << 1) | 1) << (p - 1))) + 1;
}

// k-mer ending at position i of the genome (codes[0..L)); false if the window holds a reset code
__device__ __forceinline__ bool kmer_at(const uint8_t* __restrict__ codes, long long i, int k, u64* out) {
    u64 kmer = 0;
    bool ok = true;
    for (int t = 0; t < k; ++t) {
        const uint32_t c = codes[i - (k - 1) + t];
        ok = ok && (c < 4);
        kmer = (kmer << 2) | (c & 3);
    }
    *out = kmer;
    return ok;
}

// byte-wide HLL registers packed four to an LDS word: max via read-check + CAS (updates become rare once the
// registers have warmed up, so the CAS loop almost never runs)
__device__ __forceinline__ void lds_byte_max(uint32_t* words, uint32_t idx, uint32_t rank) {
    uint32_t* w = words + (idx >> 2);
    const int sh = (idx & 3) * 8;
    uint32_t cur = *(volatile uint32_t*)w;
    while (((cur >> sh) & 0xFFu) < rank) {
        const uint32_t want = (cur & ~(0xFFu << sh)) | (rank << sh);
        const uint32_t prev = atomicCAS(w, cur, want);
        if (prev == cur) break;
        cur = prev;
    }
}

constexpr int kSketchJmaxParallel = 15;
constexpr int kSketchSeg = 64;          // consecutive k-mer end positions rolled by one thread

// visits every valid k-mer of the genome once: thread t owns segments t, t+256, ... of kSketchSeg end positions and
// rolls the 2-bit window through them (30 warm-up bases per segment)
template <typename F>
__device__ __forceinline__ void for_each_kmer(const uint8_t* __restrict__ codes, long long L, int k, F&& f) {
    const u64 kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
    for (long long seg = (long long)(k - 1) + (long long)threadIdx.x * kSketchSeg; seg < L; seg += (long long)kBlock * kSketchSeg) {
        const long long end = min(seg + kSketchSeg, L);
        u64 kmer = 0;
        int bases = 0;                                       // valid bases in the window, capped at k
        for (long long i = seg - (k - 1); i < end; ++i) {
            const uint32_t c = codes[i];
            if (c < 4) { kmer = ((kmer << 2) | c) & kmask; bases = min(bases + 1, k); }
            else       { kmer = 0; bases = 0; }                                       // build_sketch.cpp:83
            if (i >= seg && bases == k) f(kmer);
        }
    }
}

__global__ __launch_bounds__(kBlock)
void sketch_build_kernel(const uint8_t* __restrict__ codes_all, const long long* __restrict__ offsets, int k,
                         int m, int p_aux, uint8_t* __restrict__ hll_out, u64* __restrict__ smh_out,
                         uint8_t* __restrict__ aux_out) {
    extern __shared__ unsigned char smem_raw[];
    // layout: u64 h[m] | u32 regs[16384/4] (byte registers) | u32 aregs[(1<<p_aux)/4] | u32 p[m] | u32 q[m] | i32 b[m] | i32 ctl[4]
    const int n_aux = (aux_out && p_aux > 0) ? (1 << p_aux) : 0;
    const int ms = smh_out ? m : 0;
    u64* h = reinterpret_cast<u64*>(smem_raw);
    uint32_t* regs = reinterpret_cast<uint32_t*>(smem_raw + (size_t)ms * 8);
    uint32_t* aregs = regs + 16384 / 4;
    uint32_t* pp = aregs + (n_aux + 3) / 4;
    uint32_t* qq = pp + ms;
    int* bb = reinterpret_cast<int*>(qq + ms);
    int* ctl = bb + ms;

    const long long g = blockIdx.x;
    const uint8_t* codes = codes_all + offsets[g];
    const long long L = offsets[g + 1] - offsets[g];
    const uint32_t mask = (uint32_t)(m - 1);

    for (int t = threadIdx.x; t < ms; t += kBlock) h[t] = ~0ull;
    for (int t = threadIdx.x; t < 16384 / 4; t += kBlock) regs[t] = 0;
    for (int t = threadIdx.x; t < (n_aux + 3) / 4; t += kBlock) aregs[t] = 0;
    if (threadIdx.x == 0) ctl[0] = 0;
    __syncthreads();

    // pass 0: HLL registers and the step-0 offer of every k-mer
    for_each_kmer(codes, L, k, [&](u64 kmer) {
        const u64 canon = canonical_kmer(kmer, (unsigned)k);
        const u64 hv = wang_hash(canon);                                              // hll.h:901-904 addh
        uint32_t idx, rank;
        hll_slot(hv, 14, &idx, &rank);
        lds_byte_max(regs, idx, rank);
        if (n_aux) { hll_slot(hv, p_aux, &idx, &rank); lds_byte_max(aregs, idx, rank); }
        if (ms) {
            u64 st = canon ? canon : 1337ull;                                         // WyRand(seed ? seed : 1337)
            const u64 v = wyhash64_next(st);
            const u64 offer = v >> 32;                                                // j = 0: value (0<<32)|r_0
            u64* slot = &h[(uint32_t)v & mask];                                       //        bucket k_0
            if (offer < *(volatile u64*)slot) atomicMin(slot, offer);
        }
    });
    __syncthreads();

    if (ms) {
        int J = 0;
        while (true) {
            // a = max_b min(m-1, h[b] >> 32)    (bbmh.h:657-664: b_ / a_ bookkeeping, stated directly)
            int la = 0;
            for (int t = threadIdx.x; t < ms; t += kBlock) la = max(la, (int)min((u64)(m - 1), h[t] >> 32));
            atomicMax(&ctl[0], la);
            __syncthreads();
            const int a = ctl[0];
            __syncthreads();
            if (threadIdx.x == 0) ctl[0] = 0;
            if (a <= J) break;
            if (a > kSketchJmaxParallel) {
                // few k-mers per bucket: run the reference's sequential algorithm literally on one lane
                for (int t = threadIdx.x; t < ms; t += kBlock) { h[t] = ~0ull; qq[t] = 0xFFFFFFFFu; pp[t] = 0; bb[t] = 0; }
                __syncthreads();
                if (threadIdx.x == 0) {
                    bb[m - 1] = m;                                                    // bbmh.h:575
                    u64 aa = (u64)(m - 1), ii = 0;
                    for (long long i = k - 1; i < L; ++i) {
                        u64 kmer;
                        if (!kmer_at(codes, i, k, &kmer)) continue;
                        const u64 canon = canonical_kmer(kmer, (unsigned)k);
                        u64 st = canon ? canon : 1337ull;
                        u64 j = 0;
                        while (j <= aa) {                                             // bbmh.h:643-668
                            const u64 v = wyhash64_next(st);
                            const uint32_t kk = (uint32_t)v & mask;
                            if ((u64)qq[j] != ii) { qq[j] = (uint32_t)ii; pp[j] = (uint32_t)j; }
                            if ((u64)qq[kk] != ii) { qq[kk] = (uint32_t)ii; pp[kk] = kk; }
                            const uint32_t tmp = pp[kk]; pp[kk] = pp[j]; pp[j] = tmp;
                            const u64 crj = (j << 32) | (v >> 32);
                            if (crj < h[pp[j]]) {
                                const uint32_t jprime = min((uint32_t)(m - 1), (uint32_t)(h[pp[j]] >> 32));
                                h[pp[j]] = crj;
                                if (j < jprime) {
                                    --bb[jprime];
                                    ++bb[j];
                                    while (bb[aa] == 0) --aa;
                                }
                            }
                            ++j;
                        }
                        ++ii;
                    }
                }
                __syncthreads();
                break;
            }
            J = a;
            // every k-mer re-runs its chain up to step J; only steps >= 1 can be new (atomic min is idempotent)
            for_each_kmer(codes, L, k, [&](u64 kmer) {
                const u64 canon = canonical_kmer(kmer, (unsigned)k);
                u64 st = canon ? canon : 1337ull;
                uint32_t pos[2 * (kSketchJmaxParallel + 1)], val[2 * (kSketchJmaxParallel + 1)];
                int cnt = 0;
                for (int j = 0; j <= J; ++j) {
                    const u64 v = wyhash64_next(st);
                    const uint32_t kk = (uint32_t)v & mask;
                    uint32_t pj = (uint32_t)j, pk = kk;
                    int ij = -1, ik = -1;
                    for (int t = 0; t < cnt; ++t) {
                        if (pos[t] == (uint32_t)j) { pj = val[t]; ij = t; }
                        if (pos[t] == kk) { pk = val[t]; ik = t; }
                    }
                    // swap(p[kk], p[j])
                    if (ik >= 0) val[ik] = pj; else { pos[cnt] = kk; val[cnt] = pj; ik = cnt++; }
                    if (kk != (uint32_t)j) {
                        if (ij >= 0) val[ij] = pk; else { pos[cnt] = (uint32_t)j; val[cnt] = pk; cnt++; }
                    }
                    const uint32_t bucket = (kk == (uint32_t)j) ? pj : pk;            // p[j] after the swap
                    atomicMin(&h[bucket], ((u64)j << 32) | (v >> 32));
                }
            });
            __syncthreads();
        }
    }
    __syncthreads();
    uint32_t* out32 = reinterpret_cast<uint32_t*>(hll_out + g * 16384);
    for (int t = threadIdx.x; t < 16384 / 4; t += kBlock) out32[t] = regs[t];
    for (int t = threadIdx.x; t < n_aux; t += kBlock) aux_out[g * n_aux + t] = (uint8_t)(aregs[t >> 2] >> ((t & 3) * 8));
    for (int t = threadIdx.x; t < ms; t += kBlock) smh_out[g * (long long)m + t] = h[t];
}

__global__ __launch_bounds__(kBlock)
void permute_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int32_t* __restrict__ perm,
                         long long n_rows, long long row_vec) {
    // one block per destination row (grid-stride), 16 B per lane
    for (long long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint4* s = src + (long long)perm[r] * row_vec;
        uint4* d = dst + r * row_vec;
        for (long long t = threadIdx.x; t < row_vec; t += kBlock) d[t] = s[t];
    }
}

__global__ void permute_bytes_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                     const int32_t* __restrict__ perm, long long n_rows, long long row_bytes) {
    for (long long r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const uint8_t* s = src + (long long)perm[r] * row_bytes;
        uint8_t* d = dst + r * row_bytes;
        for (long long t = threadIdx.x; t < row_bytes; t += blockDim.x) d[t] = s[t];
    }
}

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
    double total_ms = 0;
    long launches = 0;
};

enum { T_PREP = 0, T_STAGE1, T_HIST, T_SELECT, T_TOTAL, T_SIGBUILD, T_JOIN, T_VERIFY, T_AUX, T_COUNT };
const char* kTimerNames[T_COUNT] = {"prep", "stage1", "hist", "select", "total", "sigbuild", "join", "verify", "aux"};

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
    DevBuf<PassCounters> pc;
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
    DevBuf<uint32_t> sigQ, sigT;        // ALGO_SIG: band signatures, query-major / band-major
    PassCounters* h_pc = nullptr;       // pinned host mirror

    // last run parameters (for overflow re-runs)
    bool have_run = false, pending = false;
    int mode = 0, algo = 0, n_rows = 0, n_bands = 0;
    float tau_f = 0;
    int64_t row_begin = 0, row_end = 0;
    PassCounters last{};

    bool timing = false;
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

struct TimerScope {
    selhip_ctx* c; int id; hipEvent_t a = nullptr, b = nullptr;
    TimerScope(selhip_ctx* c_, int id_) : c(c_), id(id_) {
        if (c->timing) {
            (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a, c->stream);
        }
    }
    ~TimerScope() {
        if (c->timing) {
            (void)hipEventRecord(b, c->stream);
            c->timers[id].ev.emplace_back(a, b);
        }
    }
};

void drain_timers(selhip_ctx* c) {
    for (int t = 0; t < T_COUNT; ++t) {
        for (auto& pr : c->timers[t].ev) {
            float ms = 0;
            if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                c->timers[t].total_ms += ms;
                c->timers[t].launches += 1;
            }
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c->timers[t].ev.clear();
    }
}

double relerr_scaled_for(int p) {
    // hll.h:662  relerr /= std::sqrt(m), relerr = 1e-2 (hll.h:211 default, :257)
    return 1e-2 / std::sqrt((double)(1ull << p));
}

bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }

// ---- stage-1 dispatch ------------------------------------------------------------------------
template <int NCH, int LOG2R>
hipError_t launch_stream(selhip_ctx* c, int r_rt, int row_begin, int row_end) {
    constexpr int Q = kQueryVgprBudget / NCH;
    const int n = (int)c->n;
    const int n_tiles = (row_end - row_begin + Q - 1) / Q;
    // candidate columns that can matter: k in (row_begin, n)
    const int chunk_base = ((row_begin + 1) / kChunk) * kChunk;
    const int n_chunks = (n - chunk_base + kChunk - 1) / kChunk;
    if (n_tiles <= 0 || n_chunks <= 0) return hipSuccess;
    const long long blocks = (long long)n_tiles * n_chunks;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((smh_stream_kernel<NCH, LOG2R>), dim3((unsigned)blocks), dim3(kBlock), 0, c->stream,
                       reinterpret_cast<const u64x2*>(c->d_aux), n, r_rt, c->hi.p, c->pc.p,
                       row_begin, row_end, n_tiles, chunk_base, c->surv.p, (u64)c->surv.cap, c->pc.p);
    return hipGetLastError();
}

template <int NCH>
hipError_t launch_stream_r(selhip_ctx* c, int n_rows, int rb, int re) {
    const int l = n_rows >= 128 ? 7 : ilog2(n_rows);
    switch (l) {
        case 0: return launch_stream<NCH, 0>(c, n_rows, rb, re);
        case 1: return launch_stream<NCH, 1>(c, n_rows, rb, re);
        case 2: return launch_stream<NCH, 2>(c, n_rows, rb, re);
        case 3: return launch_stream<NCH, 3>(c, n_rows, rb, re);
        case 4: return launch_stream<NCH, 4>(c, n_rows, rb, re);
        case 5: return launch_stream<NCH, 5>(c, n_rows, rb, re);
        case 6: return launch_stream<NCH, 6>(c, n_rows, rb, re);
        default: return launch_stream<NCH, 7>(c, n_rows, rb, re);
    }
}

bool stream_supported(int m, int n_rows) {
    return is_pow2(m) && m >= 128 && m <= 2048 && is_pow2(n_rows) && n_rows <= m;
}

hipError_t launch_stage1(selhip_ctx* c, int n_rows, int n_bands, int rb, int re) {
    if (stream_supported(c->m, n_rows)) {
        switch (c->m / 128) {
            case 1: return launch_stream_r<1>(c, n_rows, rb, re);
            case 2: return launch_stream_r<2>(c, n_rows, rb, re);
            case 4: return launch_stream_r<4>(c, n_rows, rb, re);
            case 8: return launch_stream_r<8>(c, n_rows, rb, re);
            case 16: return launch_stream_r<16>(c, n_rows, rb, re);
        }
    }
    const int rows = re - rb;
    const int n = (int)c->n;
    const int chunks = (n + kBlock - 1) / kBlock;
    const long long blocks = (long long)rows * chunks;
    if (blocks <= 0) return hipSuccess;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL(smh_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream,
                       c->d_aux, n, c->m, n_rows, n_bands, c->hi.p, c->pc.p, rb, re, rows,
                       c->surv.p, (u64)c->surv.cap, c->pc.p);
    return hipGetLastError();
}


bool sig_supported(int m, int n_rows, int n_bands) {
    (void)m;
    return is_pow2(n_rows) && (n_bands == 8 || n_bands == 16 || n_bands == 32 || n_bands == 64 || n_bands == 128);
}

int env_int(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

template <int NB>
hipError_t launch_join(selhip_ctx* c, int n_pad, int rb, int re) {
    const int n = (int)c->n;
    int qt = std::max(16, env_int("SELHIP_JOIN_QT", 128));                            // development knob
    qt = (qt + 15) / 16 * 16;
    const int n_tiles = (re - rb + qt - 1) / qt;
    const int group_base = ((rb + 1) / kWave / kWavesPerBlock) * kWavesPerBlock;      // candidates k > row_begin
    const int n_groups = (n + kWave - 1) / kWave - group_base;
    const int n_gblocks = (n_groups + kWavesPerBlock - 1) / kWavesPerBlock;
    if (n_tiles <= 0 || n_gblocks <= 0) return hipSuccess;
    const long long blocks = (long long)n_tiles * n_gblocks;
    if (blocks > 0x7FFFFFFFll) return hipErrorInvalidValue;
    hipLaunchKernelGGL((sig_join_kernel<NB>), dim3((unsigned)blocks), dim3(kBlock), 0, c->stream,
                       c->sigT.p, n, n_pad, c->hi.p, c->pc.p, rb, re, n_tiles, group_base, qt,
                       c->cand.p, (u64)c->cand.cap, c->pc.p);
    return hipGetLastError();
}

hipError_t launch_stage1_sig(selhip_ctx* c, int n_rows, int n_bands, int rb, int re) {
    const int n = (int)c->n;
    const int n_pad = ((n + kWave - 1) / kWave) * kWave;
    {
        TimerScope t(c, T_SIGBUILD);
        const long long threads = n_rows <= kWave ? (long long)n * c->m : (long long)n * n_bands;
        hipLaunchKernelGGL(sig_build_kernel, dim3((unsigned)((threads + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           c->d_aux, n, c->m, n_rows, n_bands, n_pad, c->sigQ.p, c->sigT.p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipSuccess;
    {
    TimerScope t(c, T_JOIN);
    switch (n_bands) {
        case 8: e = launch_join<8>(c, n_pad, rb, re); break;
        case 16: e = launch_join<16>(c, n_pad, rb, re); break;
        case 32: e = launch_join<32>(c, n_pad, rb, re); break;
        case 64: e = launch_join<64>(c, n_pad, rb, re); break;
        case 128: e = launch_join<128>(c, n_pad, rb, re); break;
        default: return hipErrorInvalidValue;
    }
    }
    if (e != hipSuccess) return e;
    TimerScope t(c, T_VERIFY);
    hipLaunchKernelGGL(verify_kernel, dim3(1024), dim3(kBlock), 0, c->stream, c->d_aux, c->m, n_rows, n_bands,
                       c->cand.p, &c->pc.p->n_candidates, (u64)c->cand.cap, c->surv.p, (u64)c->surv.cap, c->pc.p);
    return hipGetLastError();
}

template <int MODE>
hipError_t launch_select(bool fma, hipStream_t st, unsigned grid, const uint32_t* counts, const u64* n_dev, u64 n_host,
                         u64 cap, int p, double* est, const selhip_int2_t* pairs, const u64* ecard, double tau,
                         selhip_pair_t* results, u64 results_cap, PassCounters* pc,
                         selhip_result_t* rf32, int* out_count, u64 chunk_off = 0, u64 chunk_len = ~0ull) {
    const double rs = relerr_scaled_for(p);
    if (fma)
        hipLaunchKernelGGL((ertl_select_kernel<true, MODE>), dim3(grid), dim3(kWave), 0, st, counts, n_dev, n_host, cap,
                           p, rs, est, pairs, ecard, tau, results, results_cap, pc, rf32, out_count, chunk_off, chunk_len);
    else
        hipLaunchKernelGGL((ertl_select_kernel<false, MODE>), dim3(grid), dim3(kWave), 0, st, counts, n_dev, n_host, cap,
                           p, rs, est, pairs, ecard, tau, results, results_cap, pc, rf32, out_count, chunk_off, chunk_len);
    return hipGetLastError();
}

unsigned grid_for(u64 items, unsigned per_block, unsigned max_blocks) {
    u64 b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (unsigned)b;
}

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
hipError_t launch_aux_filter(selhip_ctx* c, const selhip_int2_t* list, const u64* n_dev, u64 off, u64 len, u64 cap, double tau) {
    const float Z = 1.96f;                                   // z_score, selection.cpp:76
    const float zs_f = Z * sigma_p_of(c->p_aux);             // float * float (criteria_sketch.hpp:29,40)
    const double zs = (double)zs_f;
    const double S_sum = zs;                                 // order_n = 1 (selection.cpp:77): S = Z*sigma_p
    const double rs = relerr_scaled_for(c->p_aux);
    const unsigned grid = 2048;
    if (c->fp_mode == SELHIP_FP_FMA)
        hipLaunchKernelGGL((aux_filter_kernel<true, CRIT>), dim3(grid), dim3(kWave), 0, c->stream, c->counts.p, list, n_dev, off, len, cap,
                           c->p_aux, rs, c->ecard.p, tau, zs, S_sum, c->fin.p, (u64)c->fin.cap, &c->pc.p->n_final);
    else
        hipLaunchKernelGGL((aux_filter_kernel<false, CRIT>), dim3(grid), dim3(kWave), 0, c->stream, c->counts.p, list, n_dev, off, len, cap,
                           c->p_aux, rs, c->ecard.p, tau, zs, S_sum, c->fin.p, (u64)c->fin.cap, &c->pc.p->n_final);
    return hipGetLastError();
}

int enqueue_pass(selhip_ctx* c) {
    const int n = (int)c->n;
    const int rb = (int)c->row_begin, re = (int)c->row_end;
    const double tau = (double)c->tau_f;            // float threshold widened, selection.cpp:81,164
    const int crit = c->criterion;
    TimerScope total(c, T_TOTAL);
    HIPCHK(&c->err, hipMemsetAsync(c->pc.p, 0, sizeof(PassCounters), c->stream));
    {
        TimerScope t(c, T_PREP);
        // z0 defaults to n ("no genome with e != 0"): written before the kernel
        int z0_init = n;
        HIPCHK(&c->err, hipMemcpyAsync(&c->pc.p->z0, &z0_init, sizeof(int), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(cb_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->d_cards, n, tau, c->mode == SELHIP_MODE_CB_SMH ? 1 : 0, rb, re, c->ecard.p, c->hi.p, c->pc.p);
        HIPCHK(&c->err, hipGetLastError());
    }
    // ---- first criterion: smh_a (stream / signature join) or the explicit pair space for hll_a / hll_an
    const selhip_int2_t* final_list = c->surv.p;
    const u64* final_count = &c->pc.p->n_survivors;
    u64 final_cap = (u64)c->surv.cap;
    if (crit == SELHIP_CRIT_SMH_A || crit == SELHIP_CRIT_HLL_A_SMH_A) {
        TimerScope t(c, T_STAGE1);
        const bool use_sig = (c->algo == SELHIP_ALGO_SIG || c->algo == SELHIP_ALGO_AUTO) && sig_supported(c->m, c->n_rows, c->n_bands);
        if (c->algo == SELHIP_ALGO_SIG && !use_sig) {
            set_err(&c->err, "ALGO_SIG needs power-of-two rows and 8..128 bands (got %d x %d)", c->n_rows, c->n_bands);
            return SELHIP_E_BADARG;
        }
        if (use_sig) HIPCHK(&c->err, launch_stage1_sig(c, c->n_rows, c->n_bands, rb, re));
        else         HIPCHK(&c->err, launch_stage1(c, c->n_rows, c->n_bands, rb, re));
    } else {
        TimerScope t(c, T_STAGE1);
        const int rows = re - rb;
        const long long blocks = (long long)rows * ((n + kBlock - 1) / kBlock);
        if (blocks > 0x7FFFFFFFll) { set_err(&c->err, "row range too large"); return SELHIP_E_BADARG; }
        if (blocks > 0) {
            hipLaunchKernelGGL(enum_pairs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, c->stream, n, c->hi.p, c->pc.p,
                               rb, re, rows, c->cand.p, (u64)c->cand.cap, c->pc.p);
            HIPCHK(&c->err, hipGetLastError());
        }
    }
    // ---- auxiliary-HLL criterion (hll_a / hll_an), in windows of the counts buffer
    if (crit != SELHIP_CRIT_SMH_A) {
        TimerScope t(c, T_AUX);
        const selhip_int2_t* list = crit == SELHIP_CRIT_HLL_A_SMH_A ? c->surv.p : c->cand.p;
        const u64* n_dev = crit == SELHIP_CRIT_HLL_A_SMH_A ? &c->pc.p->n_survivors : &c->pc.p->n_aux_in;
        const u64 cap = crit == SELHIP_CRIT_HLL_A_SMH_A ? (u64)c->surv.cap : (u64)c->cand.cap;
        const u64 window = (u64)c->counts.cap / 64;
        const u64 bound = crit == SELHIP_CRIT_HLL_A_SMH_A ? cap : std::min<u64>(cap, (u64)pair_bound(n, rb, re));
        for (u64 off = 0; off < bound; off += window) {
            hipLaunchKernelGGL(hll_union_hist_kernel, dim3(2048), dim3(kBlock), 0, c->stream,
                               c->d_aux_hll, c->p_aux, list, n_dev, (u64)0, cap, c->counts.p, off, window);
            HIPCHK(&c->err, hipGetLastError());
            if (crit == SELHIP_CRIT_HLL_AN) HIPCHK(&c->err, launch_aux_filter<2>(c, list, n_dev, off, window, cap, tau));
            else                            HIPCHK(&c->err, launch_aux_filter<1>(c, list, n_dev, off, window, cap, tau));
        }
        final_list = c->fin.p;
        final_count = &c->pc.p->n_final;
        final_cap = (u64)c->fin.cap;
    }
    // ---- final criterion: HLL-14 union estimate + Jaccard (selection.cpp:286-288), windows of the counts buffer
    {
        const u64 window = (u64)c->counts.cap / 64;
        for (u64 off = 0; off < final_cap; off += window) {
            {
                TimerScope t(c, T_HIST);
                hipLaunchKernelGGL(hll_union_hist_kernel, dim3(2048), dim3(kBlock), 0, c->stream,
                                   c->d_hll, c->p, final_list, final_count, (u64)0, final_cap, c->counts.p, off, window);
                HIPCHK(&c->err, hipGetLastError());
            }
            TimerScope t(c, T_SELECT);
            HIPCHK(&c->err, launch_select<1>(c->fp_mode == SELHIP_FP_FMA, c->stream, 4096, c->counts.p, final_count, 0,
                                             final_cap, c->p, nullptr, final_list, c->ecard.p, tau,
                                             c->results.p, (u64)c->results.cap, c->pc.p, nullptr, nullptr, off, window));
        }
    }
    HIPCHK(&c->err, hipMemcpyAsync(c->h_pc, c->pc.p, sizeof(PassCounters), hipMemcpyDeviceToHost, c->stream));
    return SELHIP_OK;
}

int ensure_scratch(selhip_ctx* c, size_t surv_cap, size_t res_cap) {
    HIPCHK(&c->err, c->ecard.ensure((size_t)c->n));
    HIPCHK(&c->err, c->hi.ensure((size_t)c->n));
    HIPCHK(&c->err, c->pc.ensure(1));
    HIPCHK(&c->err, c->surv.ensure(surv_cap));
    HIPCHK(&c->err, c->cand.ensure(surv_cap));
    if (c->criterion != SELHIP_CRIT_SMH_A) HIPCHK(&c->err, c->fin.ensure(surv_cap));
    if (c->criterion == SELHIP_CRIT_HLL_A || c->criterion == SELHIP_CRIT_HLL_AN) {
        // the explicit pair space of the row range is materialised (8 B per pair)
        const long long bound = pair_bound(c->n, c->row_begin, c->row_end);
        if (bound > (1ll << 28)) {
            set_err(&c->err, "hll_a/hll_an as first criterion enumerates %lld pairs for this row range; pass sub-ranges of rows (<= 2^28 pairs each)", bound);
            return SELHIP_E_BADARG;
        }
        HIPCHK(&c->err, c->cand.ensure((size_t)bound + 1024));
    }
    {
        const size_t n_pad = (((size_t)c->n + kWave - 1) / kWave) * kWave;
        const size_t nb = (size_t)std::max(c->n_bands, 1);
        if (nb <= 128) {
            HIPCHK(&c->err, c->sigQ.ensure((size_t)c->n * nb));
            HIPCHK(&c->err, c->sigT.ensure(n_pad * nb));
        }
    }
    // histogram scratch: 256 B per pair, at most 1 Mi pairs per window (256 MiB)
    HIPCHK(&c->err, c->counts.ensure(std::min<size_t>(std::max(c->surv.cap, (size_t)c->n), (size_t)1 << 20) * 64));
    HIPCHK(&c->err, c->results.ensure(res_cap));
    if (!c->h_pc) HIPCHK(&c->err, hipHostMalloc((void**)&c->h_pc, sizeof(PassCounters), hipHostMallocDefault));
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
    c->ecard.release(); c->hi.release(); c->pc.release(); c->surv.release();
    c->counts.release(); c->results.release(); c->self_pairs.release();
    c->cand.release(); c->sigQ.release(); c->sigT.release(); c->fin.release(); c->own_aux_hll.release();
    if (c->h_pc) (void)hipHostFree(c->h_pc);
    delete c;
}

int selhip_ctx_set_stream(selhip_ctx* c, void* hip_stream) {
    if (!c) return SELHIP_E_BADARG;
    c->stream = (hipStream_t)hip_stream;
    return SELHIP_OK;
}

int selhip_ctx_set_fp_mode(selhip_ctx* c, int fp_mode) {
    if (!c || (fp_mode != SELHIP_FP_FMA && fp_mode != SELHIP_FP_STRICT)) return SELHIP_E_BADARG;
    c->fp_mode = fp_mode;
    return SELHIP_OK;
}

int selhip_ctx_set_criterion(selhip_ctx* c, int criterion) {
    if (!c || criterion < SELHIP_CRIT_SMH_A || criterion > SELHIP_CRIT_HLL_A_SMH_A) return SELHIP_E_BADARG;
    c->criterion = criterion;
    return SELHIP_OK;
}

int selhip_ctx_upload_aux_hll(selhip_ctx* c, const uint8_t* h_aux_hll, int p_aux) {
    if (!c || !h_aux_hll || p_aux < 4 || p_aux > 16) return SELHIP_E_BADARG;
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
    if (!c || !d_aux_hll || p_aux < 4 || p_aux > 16) return SELHIP_E_BADARG;
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
    if (c->n == 0) return SELHIP_OK;
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
    c->n = n; c->m = m; c->p = p_hll; c->have_run = false; c->pending = false;
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
    c->n = n; c->m = m; c->p = p_hll; c->have_run = false; c->pending = false;
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
    if (algo != SELHIP_ALGO_AUTO && algo != SELHIP_ALGO_STREAM && algo != SELHIP_ALGO_SIG) { set_err(&c->err, "bad algo %d", algo); return SELHIP_E_BADARG; }
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
        HIPCHK(&c->err, hipStreamSynchronize(c->stream));
        PassCounters pc = *c->h_pc;
        if (pc.unsorted) { c->pending = false; set_err(&c->err, "cards are not in ascending order"); return SELHIP_E_BADARG; }
        bool grow = false;
        size_t surv_cap = c->surv.cap, res_cap = c->results.cap;
        if (pc.n_survivors > c->surv.cap) { surv_cap = (size_t)(pc.n_survivors + pc.n_survivors / 8 + 1024); grow = true; }
        if (pc.n_candidates > c->cand.cap) { surv_cap = std::max(surv_cap, (size_t)(pc.n_candidates + pc.n_candidates / 8 + 1024)); grow = true; }
        if (c->criterion != SELHIP_CRIT_SMH_A && pc.n_final > c->fin.cap) { surv_cap = std::max(surv_cap, (size_t)(pc.n_final + pc.n_final / 8 + 1024)); grow = true; }
        if (pc.n_aux_in > c->cand.cap) { c->pending = false; set_err(&c->err, "internal: enumerated pair list overflow"); return SELHIP_E_OVERFLOW; }
        if (pc.n_results > c->results.cap) { res_cap = (size_t)(pc.n_results + pc.n_results / 8 + 1024); grow = true; }
        if (!grow) {
            c->last = pc; c->pending = false; c->have_run = true;
            if (c->timing) drain_timers(c);
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

int selhip_ctx_timing(selhip_ctx* c, int enable) {
    if (!c) return SELHIP_E_BADARG;
    (void)hipStreamSynchronize(c->stream);
    drain_timers(c);
    for (int t = 0; t < T_COUNT; ++t) { c->timers[t].total_ms = 0; c->timers[t].launches = 0; }
    c->timing = enable != 0;
    return SELHIP_OK;
}

double selhip_ctx_kernel_ms(const selhip_ctx* c, const char* name) {
    if (!c || !name) return -1.0;
    for (int t = 0; t < T_COUNT; ++t)
        if (!std::strcmp(name, kTimerNames[t]))
            return c->timers[t].launches ? c->timers[t].total_ms / (double)c->timers[t].launches : -1.0;
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
};
CompatWs g_ws;

// select for the drop-in launchers: the reference signature carries no genome count, so the truncated
// cardinalities are taken on the fly from cards[rank]; output record = struct Result {x, y, (float)J}.
template <bool FMA>
__global__ __launch_bounds__(kWave)
void compat_select_kernel(const uint32_t* __restrict__ counts, const u64* __restrict__ n_dev, u64 cap, int p,
                          double relerr_scaled, const selhip_int2_t* __restrict__ pairs,
                          const double* __restrict__ cards, double tau,
                          selhip_result_t* __restrict__ out, int* __restrict__ out_count) {
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
                int idx = atomicAdd(out_count, 1);
                out[idx].x = pr.x; out[idx].y = pr.y; out[idx].sim = (float)jacc;
            }
        }
    }
}

int compat_launch(bool use_cb, const uint8_t* main_sketches, const uint64_t* aux, const double* cards,
                  const selhip_int2_t* pairs, int total_pairs, double tau, int m_aux, int m_hll,
                  int n_rows, int n_bands, selhip_result_t* out, int* out_count) {
    if (!main_sketches || !aux || !cards || !out || !out_count) { set_err(nullptr, "null pointer argument"); return SELHIP_E_BADARG; }
    if (total_pairs < 0 || m_aux <= 0 || !is_pow2(m_hll) || m_hll < 16) { set_err(nullptr, "bad sizes"); return SELHIP_E_BADARG; }
    if (n_rows <= 0 || n_bands <= 0 || (long long)n_rows * n_bands != m_aux) { set_err(nullptr, "n_rows*n_bands != m_aux"); return SELHIP_E_BADARG; }
    if (!pairs && total_pairs > 0) { set_err(nullptr, "pairs == NULL (use the selhip_ctx_* API for implicit all-pairs enumeration)"); return SELHIP_E_BADARG; }
    hipStream_t st = nullptr;                                        // default stream, like the reference
    HIPCHK(nullptr, hipMemsetAsync(out_count, 0, sizeof(int), st));  // selection_kernels.cu:137,166
    if (total_pairs == 0) return SELHIP_OK;
    std::lock_guard<std::mutex> lk(g_ws.mu);
    HIPCHK(nullptr, g_ws.surv.ensure((size_t)total_pairs));
    HIPCHK(nullptr, g_ws.counts.ensure((size_t)total_pairs * 64));
    HIPCHK(nullptr, g_ws.surv_count.ensure(1));
    HIPCHK(nullptr, hipMemsetAsync(g_ws.surv_count.p, 0, sizeof(u64), st));
    hipLaunchKernelGGL(pairlist_smh_kernel, dim3((unsigned)((total_pairs + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const u64*)aux, m_aux, n_rows, n_bands, pairs, (long long)total_pairs, cards, tau, 1, use_cb ? 1 : 0,
                       (uint8_t*)nullptr, g_ws.surv.p, (u64)g_ws.surv.cap, g_ws.surv_count.p);
    HIPCHK(nullptr, hipGetLastError());
    const int p = ilog2(m_hll);
    hipLaunchKernelGGL(hll_union_hist_kernel, dim3(2048), dim3(kBlock), 0, st, main_sketches, p, g_ws.surv.p,
                       g_ws.surv_count.p, (u64)0, (u64)g_ws.surv.cap, g_ws.counts.p);
    HIPCHK(nullptr, hipGetLastError());
    hipLaunchKernelGGL((compat_select_kernel<true>), dim3(4096), dim3(kWave), 0, st,
                       g_ws.counts.p, g_ws.surv_count.p, (u64)g_ws.surv.cap, p, relerr_scaled_for(p), g_ws.surv.p,
                       cards, tau, out, out_count);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}
}  // namespace

extern "C" {

int launch_kernel_smh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                      const selhip_int2_t* pairs, int total_pairs, double tau,
                      int m_aux, int m_hll, int n_rows, int n_bands,
                      selhip_result_t* out, int* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch(false, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out, out_count);
}

int launch_kernel_CBsmh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                        const selhip_int2_t* pairs, int total_pairs, double tau,
                        int m_aux, int m_hll, int n_rows, int n_bands,
                        selhip_result_t* out, int* out_count, int blockSize) {
    (void)blockSize;
    return compat_launch(true, main_sketches, aux_sketches, cards, pairs, total_pairs, tau, m_aux, m_hll, n_rows, n_bands, out, out_count);
}

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
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(nullptr, hipFuncSetAttribute((const void*)synth_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
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
    static bool attr_set = false;
    if (!attr_set) {
        HIPCHK(nullptr, hipFuncSetAttribute((const void*)sketch_build_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(sketch_build_kernel, dim3((unsigned)n_genomes), dim3(kBlock), smem, (hipStream_t)hip_stream,
                       d_codes, (const long long*)d_offsets, k, m, p_aux, d_hll, (u64*)d_smh, d_aux_hll);
    HIPCHK(nullptr, hipGetLastError());
    return SELHIP_OK;
}

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
