// common.cuh -- types, launch constants, per-pass counters and the per-wave output appender.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

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
    int z0p1;            // 1 + first rank with e != 0; 0 (the memset value) = none, i.e. z0 = n
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

}  // namespace
