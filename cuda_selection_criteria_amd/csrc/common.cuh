// common.cuh -- types, launch constants, per-pass counters and the per-wave output appender.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

using u64 = unsigned long long;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kChunk = 1024;           // candidates per block of the stream kernel (256: every block re-fetches its query tile from beyond L2 for a
                                       // quarter of the work -- 1.1 GB per launch at cfg3, 1.52 ms; 512-1024: 1.43 ms; 2048: 1.48 ms)
#ifndef SELHIP_QBUDGET
#define SELHIP_QBUDGET 24
#endif
#ifndef SELHIP_AHEAD
#define SELHIP_AHEAD 2
#endif
constexpr int kQueryVgprBudget = SELHIP_QBUDGET;   // u64x2 query registers per lane  (Q * NCH): 96 VGPRs + the ring of candidate registers fit 168 without scratch
constexpr int kStreamAhead = SELHIP_AHEAD;        // ALGO_STREAM: candidate rows in flight ahead of the one being compared

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
    u64 n_pre;           // ALGO_SIG, 16-bit join: pairs with an equal 16-bit band signature (filtered down to n_candidates); sum over the append segments
    u64 n_pre_segmax;    //   the fullest append segment's count (overflow test: every segment has cap / kAppendSegs slots)
    int z0p1;            // 1 + first rank with e != 0; 0 (the memset value) = none, i.e. z0 = n
    int unsorted;        // set if cards are not ascending
};
static_assert(sizeof(PassCounters) % 8 == 0, "counter blocks are cleared and copied as u64 words");


// ---------------------------------------------------------------------------------------------
// RowMap: which query rows a pass owns.  Rows [row_begin, row_end) are cut into blocks of block_rows rows, dealt to the n_parts
// parts BOUSTROPHEDON: cycle q (blocks q n_parts .. (q + 1) n_parts - 1) hands its r-th block to part r when q is even and to part
// n_parts - 1 - r when q is odd.  Row i has n - 1 - i candidates, so under the plain deal (block b to part b % n_parts) part 0 always
// got the longest rows of a cycle: 51.6e6 pairs against 48.4e6 for part 7 of 8 at 28 280 genomes, and the slowest rank sets the step
// (0.262 ms against a mean of 0.243, profiles/r03_scaling_emulation.txt); the snake evens that out to second order.
// n_parts = 1 (one block spanning the range) is the plain contiguous range; the multi-GPU drivers use n_parts = world size with
// small blocks, so that every rank gets the same share of pairs AND of survivors (a contiguous equal-pair cut gives the last rank
// ~35 % of all rows, hence of all stage-2 work, at 8 ranks).
// Kernels address their query rows through "local tiles": tile t of height tile_h inside the owned blocks.
// ---------------------------------------------------------------------------------------------
struct RowMap {
    int row_begin, row_end, block_rows, n_parts, part;
    __host__ __device__ int total_blocks() const { return (int)(((long long)row_end - row_begin + block_rows - 1) / block_rows); }
    __host__ __device__ int pos_in_cycle(int q) const { return (q & 1) ? n_parts - 1 - part : part; }     // this part's block inside cycle q
    __host__ __device__ int local_blocks() const { const int tb = total_blocks(), full = tb / n_parts; return full + (pos_in_cycle(full) < tb % n_parts ? 1 : 0); }
    __host__ __device__ int tiles_per_block(int tile_h) const { return (block_rows + tile_h - 1) / tile_h; }
    __host__ __device__ long long n_tiles(int tile_h) const { return (long long)local_blocks() * tiles_per_block(tile_h); }
    __host__ __device__ bool owns(int i) const {
        if (i < row_begin || i >= row_end) return false;
        const int b = (i - row_begin) / block_rows;
        return b % n_parts == pos_in_cycle(b / n_parts);
    }
    // rows [*lo, *end) of local tile t; empty (lo == end) past the range
    __device__ __forceinline__ void tile_rows(int t, int tile_h, int* lo, int* end) const {
        const int tpb = tiles_per_block(tile_h);
        const int lb = t / tpb, w = t % tpb;
        const long long bs = (long long)row_begin + ((long long)lb * n_parts + pos_in_cycle(lb)) * block_rows;     // one block per cycle
        long long l = bs + (long long)w * tile_h;
        long long e = l + tile_h;
        if (e > bs + block_rows) e = bs + block_rows;
        if (e > row_end) e = row_end;
        if (l > row_end) l = row_end;
        if (e < l) e = l;
        *lo = (int)l; *end = (int)e;
    }
};

// ---------------------------------------------------------------------------------------------
// WaveAppender: per-wave staging of output records in LDS, flushed with ONE global atomic per >= 64 records.
// A returning atomic on a single address sustains only ~90 operations/us chip-wide (MI355X_MICROARCH.md,
// row "dequeue"), so appending survivors one atomicAdd at a time caps a pass at ~90 survivors/us
// (45 000 survivors = 0.5 ms -- measured: it was THE cost of the first signature-join kernels).
// ---------------------------------------------------------------------------------------------
constexpr int kAppendCap = 2 * kWave;          // count < 64 before a push, a push adds <= 64

// The all-pairs join appends through kAppendSegs INDEPENDENT lists (own counter, own slice of the output array; a block
// uses segment blockIdx % kAppendSegs).  Measured (gpurun_out/r02/join_dbg.txt, PMC in DESIGN.md section 4): with ONE counter
// the join of cfg4 was bound by the ~87 returning atomics per microsecond a single address sustains -- every wave flushes
// at least once, 153 000 waves = 1.8 ms of a 2.1 ms kernel, and halving the tile height doubled the kernel time.
constexpr int kCounterBlocks = 9;               // counter blocks per pass: block 0 + one per pipeline chunk (kMaxChunks = 8)
constexpr int kAppendSegs = 64;
constexpr int kSegStride = 16;                 // u64 slots between two segment counters (128 B: one cache line each)

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
