// kernel_hll.cuh -- stage 2: HLL union histograms, Ertl-MLE + Jaccard selection, auxiliary-HLL criteria (hll_a / hll_an).
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

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
// Grouping survivors by query row, and the histogram kernel that exploits it.  Stage 2a issues one conflict-free
// `ds_add_u32` per 64 register pairs and the LDS retires one such wave-instruction every ~5 cycles per CU
// (scripts/microbench/lds_atomic_rate.hip): 256 of them per pair = the kernel's floor (cfg3: 45 000 pairs -> ~95 us).  To sit
// on that floor the 32 KiB of rows per pair must come from L2 as often as possible, so csr_count / csr_fill bucket the pair
// list by its first rank (counting sort, order inside a bucket free) and the histogram kernel hands neighbouring pairs of
// the grouped list to waves that run at the same time on the same XCD: the query row of a group is then an L2 hit for
// all but one of them (ungrouped list: 188 us instead of 118 us at cfg3).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void csr_count_kernel(const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ n_dev, u64 cap, int* __restrict__ cnt,
                      int* __restrict__ lab, int n_rows_total) {
    u64 n = *n_dev;
    if (n > cap) n = cap;
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < n; j += (u64)gridDim.x * kBlock) {
        const selhip_int2_t pr = pairs[j];
        atomicAdd(&cnt[pr.x], 1);
        if (lab) atomicMax(&lab[pr.y], n_rows_total - pr.x);
    }
}

// Locality order of the query rows (sets whose HLL rows do not fit the 256 MiB Infinity Cache).  Bucketing by query row alone
// leaves the buckets in rank = cardinality order, and similar genomes -- the pairs that reach stage 2 -- have similar but not
// adjacent ranks: the rows of one cluster come up as candidates again and again, each time long after they left the XCD's
// 4 MiB L2 (measured, profiles/r02_hist_locality.txt: every pair fetches ~1 row from beyond L2 at cfg3/cfg4, 2 at cfg5, and the
// kernel then runs at the fabric's 6.5 TB/s; with every cluster contiguous in the list it takes 482 instead of 705 us at cfg4,
// 2.3 instead of 5.5 ns per pair at cfg5).  So the buckets are laid out by LABEL first: label(i) = the smallest rank that
// row i is paired with (itself if none is smaller) -- for a clique of similar genomes that is the clique's first member, so
// the clique's buckets become neighbours and stage 2a's waves, which walk the list in order XCD by XCD, meet each of its
// rows while it is still in L2.  Any other pair graph only gets a different (still valid) order: results do not depend on it.
//   lab[k] = max over pairs (i, k) of n - i   (0 = no smaller partner; tallied where the pairs are counted);
//   label(i) = the root of i's smallest-partner chain (at most 8 hops)
//   csr_label_sum:    gsum[label(i)] += cnt[i]
//   (exclusive scan of gsum -> gbase, rocPRIM)
//   csr_label_assign: start[i] = atomicAdd(&gbase[label(i)], cnt[i])     (order of the rows inside a label group: free)
__device__ __forceinline__ int csr_label_of(const int* __restrict__ lab, int i, int n) {
    // follow the smallest-partner links to their root (a row with no smaller partner): in a cluster that is not a full clique --
    // e.g. after the auxiliary criterion removed a third of its pairs -- a member's smallest partner need not be the cluster's
    // first member, and one hop would split the cluster into several label groups far apart in the list
    int cur = i;
#pragma unroll 1
    for (int hop = 0; hop < 8; ++hop) {
        const int v = lab[cur];
        if (!v) break;
        cur = n - v;                      // strictly smaller rank: the walk ends
    }
    return cur;
}

__global__ __launch_bounds__(kBlock)
void csr_label_sum_kernel(const int* __restrict__ cnt, const int* __restrict__ lab, int n, int* __restrict__ gsum) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n) return;
    const int c = cnt[i];
    if (c) atomicAdd(&gsum[csr_label_of(lab, i, n)], c);
}

__global__ __launch_bounds__(kBlock)
void csr_label_assign_kernel(const int* __restrict__ cnt, const int* __restrict__ lab, int n, int* __restrict__ gbase,
                             int* __restrict__ start) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n) return;
    const int c = cnt[i];
    start[i] = c ? atomicAdd(&gbase[csr_label_of(lab, i, n)], c) : 0;
}

__global__ __launch_bounds__(kBlock)
void csr_fill_kernel(const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ n_dev, u64 cap,
                     const int* __restrict__ start, int* __restrict__ fill, selhip_int2_t* __restrict__ grouped) {
    u64 n = *n_dev;
    if (n > cap) n = cap;
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < n; j += (u64)gridDim.x * kBlock) {
        const selhip_int2_t pr = pairs[j];
        const u64 pos = (u64)start[pr.x] + (u64)atomicAdd(&fill[pr.x], 1);
        if (pos < cap) grouped[pos] = pr;                                    // (always true when the counts tallied min(n, cap) pairs)
    }
}

// The label order in TWO launches for n <= kSmallScanMax (the four above cost ~10 us more than the plain order at cfg3):
//   csr_label_offsets_kernel: per row its root and, from the SAME atomic that accumulates the group's size, the offset of its bucket
//                             inside the group  (off[i] = atomicAdd(&gsum[root(i)], cnt[i]): the old value is the offset);
//   csr_label_scan_fill_kernel: every block scans the group sizes into its own LDS copy (as csr_scan_fill_kernel does with the row
//                             counts) and scatters its share of the pairs to gbase[root(x)] + off[x] + (cursor of row x).
__global__ __launch_bounds__(kBlock)
void csr_label_offsets_kernel(const int* __restrict__ cnt, const int* __restrict__ lab, int n, int* __restrict__ gsum,
                              int* __restrict__ root, int* __restrict__ off) {
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= n) return;
    const int c = cnt[i];
    const int r = csr_label_of(lab, i, n);
    root[i] = r;
    off[i] = c ? atomicAdd(&gsum[r], c) : 0;
}

// (The three steps as ONE single-block launch with the group sums in LDS, for N <= 32 768, were built and measured at cfg3: grouping
// 9.5 -> 20 us (five dependent memory round trips on one CU), stage 2a 114 -> 106 us, step 0.311 -> 0.313 ms -- dropped; sets that fit
// the Infinity Cache keep the query-row order.)
constexpr int kSmallScanMax = 32768;        // rows whose counts one block scans in LDS (128 KiB); larger N: rocPRIM scan + csr_fill_kernel

// csr_scan_fill_kernel: the two grouping steps in ONE launch for n <= kSmallScanMax: every block repeats the exclusive scan of
// the per-row counts into its own LDS copy (n ints; a few microseconds, all blocks at once) and then scatters its share of the
// pairs -- one dispatch and one single-block kernel less on the stream than scan + fill.
__global__ __launch_bounds__(1024)
void csr_scan_fill_kernel(const int* __restrict__ cnt, int n, const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ n_dev,
                          u64 cap, int* __restrict__ fill, selhip_int2_t* __restrict__ grouped) {
    extern __shared__ int csr_start_lds[];                                  // n ints
    __shared__ int wave_sum[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int per = (n + 1023) / 1024;                                      // <= 32
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, n);
    int v[32];
    int mine = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) { v[t] = (t < per && i0 + t < i1) ? cnt[i0 + t] : 0; mine += v[t]; }
    int inc = mine;
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) { const int o = __shfl_up(inc, s, kWave); if (lane >= s) inc += o; }
    if (lane == kWave - 1) wave_sum[wave] = inc;
    __syncthreads();
    int run = inc - mine;
    for (int w = 0; w < wave; ++w) run += wave_sum[w];
#pragma unroll
    for (int t = 0; t < 32; ++t) { if (t < per && i0 + t < i1) csr_start_lds[i0 + t] = run; run += v[t]; }
    __syncthreads();
    u64 np = *n_dev;
    if (np > cap) np = cap;
    for (u64 j = (u64)blockIdx.x * 1024 + threadIdx.x; j < np; j += (u64)gridDim.x * 1024) {
        const selhip_int2_t pr = pairs[j];
        const u64 pos = (u64)csr_start_lds[pr.x] + (u64)atomicAdd(&fill[pr.x], 1);
        if (pos < cap) grouped[pos] = pr;
    }
}

__global__ __launch_bounds__(1024)
void csr_label_scan_fill_kernel(const int* __restrict__ gsum, int n, const int* __restrict__ root, const int* __restrict__ off,
                                const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ n_dev,
                                u64 cap, int* __restrict__ fill, selhip_int2_t* __restrict__ grouped) {
    extern __shared__ int csr_start_lds[];                                  // n ints: start of every label group
    __shared__ int wave_sum[16];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int per = (n + 1023) / 1024;                                      // <= 32
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, n);
    int v[32];
    int mine = 0;
#pragma unroll
    for (int t = 0; t < 32; ++t) { v[t] = (t < per && i0 + t < i1) ? gsum[i0 + t] : 0; mine += v[t]; }
    int inc = mine;
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) { const int o = __shfl_up(inc, s, kWave); if (lane >= s) inc += o; }
    if (lane == kWave - 1) wave_sum[wave] = inc;
    __syncthreads();
    int run = inc - mine;
    for (int w = 0; w < wave; ++w) run += wave_sum[w];
#pragma unroll
    for (int t = 0; t < 32; ++t) { if (t < per && i0 + t < i1) csr_start_lds[i0 + t] = run; run += v[t]; }
    __syncthreads();
    u64 np = *n_dev;
    if (np > cap) np = cap;
    for (u64 j = (u64)blockIdx.x * 1024 + threadIdx.x; j < np; j += (u64)gridDim.x * 1024) {
        const selhip_int2_t pr = pairs[j];
        const u64 pos = (u64)csr_start_lds[root[pr.x]] + (u64)off[pr.x] + (u64)atomicAdd(&fill[pr.x], 1);
        if (pos < cap) grouped[pos] = pr;
    }
}

// hll_union_hist_runs_kernel (p = 14): one wave per block with a lane-private [bin][lane] histogram (16 KiB of LDS).
//  * block b works in the (b % 8)-th eighth of the list (round-robin block dispatch puts it on XCD b % 8) and the waves
//    of an XCD stride through that eighth together, `run_len` (default 1) consecutive pairs at a time;
//  * the LDS address of a register pair is formed by ONE VALU instruction: with the histogram at LDS offset 0, [bin][lane]
//    puts lane*4 in byte 0 and the bin in byte 1 of the address, and `v_max_u32_sdwa dst_sel:BYTE_1
//    dst_unused:UNUSED_PRESERVE` writes max(byte_s(x), byte_s(y)) straight into byte 1 of a register that keeps lane*4
//    (the compiler's own sequence -- max, shift, mask, add -- is kept as the path for a histogram not at offset 0);
//  * (binning a pair in two halves with the other half's loads in flight -- same registers, no wait on memory inside a
//    wave -- changed nothing at cfg3 and cost 10 % at cfg4: the kernel waits on the LDS, not on memory);
//  * (non-temporal loads for the candidate rows -- each is used by one pair of this wave only -- were measured: 149 instead of 115 us
//    at cfg3, +8 / +15 % at cfg4 / cfg5: the rows ARE reused through L2 by the other waves of the bucket's neighbourhood);
//  * the histogram is zeroed once per wave: bins only grow, and a pair's counts are the difference of the running column
//    sums before and after it (unsigned arithmetic: wrap-around cancels), which removes 64 LDS stores per pair.
constexpr int kHistSpanBlocks = 16384;      // one-wave blocks (a multiple of 8); ~8 resident per CU, the rest balance the tail
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

#define SELHIP_SDWA_MAX_B1(addr, a, b, SEL)                                                                    \
    asm("v_max_u32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:" SEL " src1_sel:" SEL   \
        : "+v"(addr) : "v"(a), "v"(b))

__device__ __forceinline__ void hist_add_max_word(uint32_t& a0, uint32_t& a1, uint32_t& a2, uint32_t& a3, uint32_t x, uint32_t y) {
    SELHIP_SDWA_MAX_B1(a0, x, y, "BYTE_0");
    __hip_atomic_fetch_add((lds_u32_t*)(uintptr_t)a0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    SELHIP_SDWA_MAX_B1(a1, x, y, "BYTE_1");
    __hip_atomic_fetch_add((lds_u32_t*)(uintptr_t)a1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    SELHIP_SDWA_MAX_B1(a2, x, y, "BYTE_2");
    __hip_atomic_fetch_add((lds_u32_t*)(uintptr_t)a2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    SELHIP_SDWA_MAX_B1(a3, x, y, "BYTE_3");
    __hip_atomic_fetch_add((lds_u32_t*)(uintptr_t)a3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(kWave)
void hll_union_hist_runs_kernel(const uint8_t* __restrict__ hll, const selhip_int2_t* __restrict__ pairs,
                                const u64* __restrict__ n_pairs_dev, u64 cap, uint32_t* __restrict__ counts,
                                u64 chunk_off, u64 chunk_len, int run_len) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[64 * kWave];     // [bin][lane], 16 KiB: the only LDS object -> offset 0
    const int lane = threadIdx.x;
    u64 n_pairs = *n_pairs_dev;
    if (n_pairs > cap) n_pairs = cap;
    n_pairs = n_pairs > chunk_off ? min(n_pairs - chunk_off, chunk_len) : 0;
    pairs += chunk_off;
    const u64 per_xcd = gridDim.x >> 3;                                   // host launches a multiple of 8 blocks
    const u64 n_tasks = (n_pairs + run_len - 1) / run_len;
    const u64 tasks_per_xcd = (n_tasks + 7) >> 3;
    const u64 t_begin = (u64)(blockIdx.x & 7) * tasks_per_xcd, t_end = min(t_begin + tasks_per_xcd, n_tasks);
    const bool lds_at_zero = (uint32_t)(uintptr_t)hist == 0u;
    uint4* const row4 = reinterpret_cast<uint4*>(hist + lane * kWave);     // lane l owns bin l in the zero / reduce steps
    uint32_t* const col = hist + lane;
    uint32_t a0 = (uint32_t)lane * 4u, a1 = a0, a2 = a0, a3 = a0;
    constexpr uint32_t kMask = 0x3F3F3F3Fu;                               // register values are <= 64-p+1 < 64
    int cur_x = -1;
    uint4 xa[16];
    uint32_t prev = 0;
#pragma unroll
    for (int t = 0; t < 16; ++t) row4[(t + lane) & 15] = make_uint4(0u, 0u, 0u, 0u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    // (handing the tasks of an XCD's eighth out through a queue counter instead of this static stride -- so that the resident waves
    // always work on neighbouring tasks -- was measured: cfg4 475-478 vs 480 us, cfg5 927-942 vs 934 us, cfg3 slower; dropped)
    for (u64 task = t_begin + (blockIdx.x >> 3); task < t_end; task += per_xcd) {
    const u64 j0 = task * run_len, j1 = min(j0 + run_len, n_pairs);
    selhip_int2_t pr = pairs[j0];
    for (u64 j = j0; j < j1; ++j) {
        const uint4* b4 = reinterpret_cast<const uint4*>(hll + (long long)pr.y * 16384);
        uint4 xb[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) xb[it] = b4[it * kWave + lane];
        if (pr.x != cur_x) {                                              // wave-uniform: the query row changes
            const uint4* a4 = reinterpret_cast<const uint4*>(hll + (long long)pr.x * 16384);
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                uint4 v = a4[it * kWave + lane];
                xa[it] = make_uint4(v.x & kMask, v.y & kMask, v.z & kMask, v.w & kMask);
            }
            cur_x = pr.x;
        }
        if (j + 1 < j1) pr = pairs[j + 1];                                // next pair's ranks arrive while this one is binned
        if (lds_at_zero) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                hist_add_max_word(a0, a1, a2, a3, xa[it].x, xb[it].x & kMask);
                hist_add_max_word(a0, a1, a2, a3, xa[it].y, xb[it].y & kMask);
                hist_add_max_word(a0, a1, a2, a3, xa[it].z, xb[it].z & kMask);
                hist_add_max_word(a0, a1, a2, a3, xa[it].w, xb[it].w & kMask);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                hist_add_word(col, max_u8x4(xa[it].x, xb[it].x));
                hist_add_word(col, max_u8x4(xa[it].y, xb[it].y));
                hist_add_word(col, max_u8x4(xa[it].z, xb[it].z));
                hist_add_word(col, max_u8x4(xa[it].w, xb[it].w));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t sacc = 0;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const uint4 v = row4[(t + lane) & 15];
            sacc += (v.x + v.y) + (v.z + v.w);
        }
        counts[j * 64 + lane] = sacc - prev;                              // the histogram is never re-zeroed: difference of running sums
        prev = sacc;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    }
}

// frame_results_kernel: the result hand-over for a fixed-size collective in ONE launch -- record 0 of `dst` = {count, 0}
// (the device-side counter: the host need not know it yet), then min(count, cap) result records.
__global__ __launch_bounds__(kBlock)
void frame_results_kernel(const selhip_pair_t* __restrict__ results, const u64* __restrict__ n_results, u64 results_cap,
                          u64 cap_records, uint4* __restrict__ dst) {
    const u64 cnt_all = *n_results;
    u64 cnt = cnt_all < cap_records ? cnt_all : cap_records;
    if (cnt > results_cap) cnt = results_cap;
    const uint4* src = reinterpret_cast<const uint4*>(results);               // 16-byte records
    if (blockIdx.x == 0 && threadIdx.x == 0) dst[0] = make_uint4((uint32_t)cnt_all, (uint32_t)(cnt_all >> 32), 0u, 0u);
    for (u64 j = (u64)blockIdx.x * kBlock + threadIdx.x; j < cnt; j += (u64)gridDim.x * kBlock) dst[1 + j] = src[j];
}

// ---------------------------------------------------------------------------------------------
// (Round 3 built stage 2a + 2b as ONE kernel -- a block takes 64 consecutive pairs, its four waves build 16 histograms each into an LDS
// tile, wave 0 runs the estimator and appends; a dense list falls back to stage 2a's slice walk and a device word tells the select kernel
// behind it whether anything is left -- bit-identical in the parity suite, and slower everywhere: cfg3 77 + 7 us (the select launch that
// finds nothing to do) against 52 + 20, cfg4 0.425 + 0.13 against 0.289 + 0.144 ms, configs[1] 43 against 12 + 18 us
// (gpurun_out/r03/x_*).  Whole batches per block cannot be dealt as evenly as stage 2a's one- and two-pair tasks (703 blocks on 1 024
// slots: 2.75 waves per SIMD), and the solve's latency then sits behind the slowest block's sixteen pairs instead of behind the last
// pair.  Not kept.)
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
__global__ __launch_bounds__(kBlock)
void ertl_select_kernel(const uint32_t* __restrict__ counts, const u64* __restrict__ n_dev, u64 n_host, u64 cap,
                        int p, double relerr_scaled,
                        double* __restrict__ est,
                        const selhip_int2_t* __restrict__ pairs, const u64* __restrict__ ecard, double tau,
                        selhip_pair_t* __restrict__ results, u64 results_cap, PassCounters* __restrict__ pc,
                        selhip_result_t* __restrict__ results_f32, int* __restrict__ out_count_i32,
                        u64 chunk_off, u64 chunk_len) {
    __shared__ uint32_t lds_all[kWavesPerBlock][64 * 65];
    __shared__ uint32_t blk_count;
    __shared__ u64 blk_base;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    uint32_t* const lds = lds_all[wave];
    u64 n = n_dev ? *n_dev : n_host;
    if (n > cap) n = cap;
    n = n > chunk_off ? min(n - chunk_off, chunk_len) : 0;       // window of the list; counts indexed from its start
    if (pairs) pairs += chunk_off;
    if (est) est += chunk_off;
    if (threadIdx.x == 0) blk_count = 0;
    for (u64 base = (u64)blockIdx.x * kBlock; base < n; base += (u64)gridDim.x * kBlock) {
        __syncthreads();
        // thread = histogram base+threadIdx.x: its 64 counts (256 contiguous bytes) are requested with 16 independent 16-B
        // loads and parked in column `lane` of the wave's LDS tile (pitch 65: conflict-free) -- the estimator indexes them
        // with run-time k.  (Reading the tile row by row, lane = bin, serialised 64 dependent round trips: 33 us for
        // 45 000 histograms.)
        const u64 j = base + threadIdx.x;
        uint4 row[16];
        if (j < n) {
            const uint4* src = reinterpret_cast<const uint4*>(counts + j * 64);
#pragma unroll
            for (int t4 = 0; t4 < 16; ++t4) row[t4] = src[t4];
        } else {
#pragma unroll
            for (int t4 = 0; t4 < 16; ++t4) row[t4] = make_uint4(0u, 0u, 0u, 0u);
            row[0].x = 1u << p;                                              // an empty sketch: estimate 0, result unused
        }
        selhip_int2_t pr{0, 0};
        u64 ec1 = 0, ec2 = 0;
        if constexpr (MODE != 0) {
            if (j < n) { pr = pairs[j]; ec1 = ecard[pr.x]; ec2 = ecard[pr.y]; }
        }
#pragma unroll
        for (int t4 = 0; t4 < 16; ++t4) {
            lds[(4 * t4 + 0) * 65 + lane] = row[t4].x; lds[(4 * t4 + 1) * 65 + lane] = row[t4].y;
            lds[(4 * t4 + 2) * 65 + lane] = row[t4].z; lds[(4 * t4 + 3) * 65 + lane] = row[t4].w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");               // a tile is private to its wave
        LdsCounts c{lds + lane};
        const double t = selhip::ertl_ml_estimate<FMA>(c, (unsigned)p, (unsigned)(64 - p), relerr_scaled);
        if constexpr (MODE == 0) {
            if (j < n) est[j] = t;
        } else {
            const double e1 = (double)ec1, e2 = (double)ec2;
            const double jacc = (e1 + e2 - t) / t;                           // selection.cpp:287
            const bool keep = j < n && jacc >= tau;                          // selection.cpp:288
            // one global append per block: a wave reserves its slots in the block's tally (LDS), thread 0 reserves the block's
            // range in the output list
            const u64 km = __ballot(keep);
            uint32_t wbase = 0;
            if (lane == 0 && km) wbase = atomicAdd(&blk_count, (uint32_t)__popcll(km));
            wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t cnt = blk_count;
                if (cnt) {
                    if constexpr (MODE == 1) blk_base = atomicAdd(&pc->n_results, (u64)cnt);
                    else                     blk_base = (u64)atomicAdd(out_count_i32, (int)cnt);
                }
                blk_count = 0;
            }
            __syncthreads();
            if (keep) {
                const u64 idx = blk_base + wbase + (u64)__popcll(km & ((1ull << lane) - 1ull));
                if constexpr (MODE == 1) {
                    if (idx < results_cap) { results[idx].i = pr.x; results[idx].k = pr.y; results[idx].jaccard = jacc; }
                } else {
                    results_f32[idx].x = pr.x; results_f32[idx].y = pr.y; results_f32[idx].sim = (float)jacc;
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
// One block lists up to kEnumSpan consecutive candidates of ONE row: the count is known from the row's cut-off, so the block
// reserves its stretch of the list with a single atomic and writes it coalesced.  (The first form appended per wave through one
// counter: 781 000 single-address atomics for cfg3's 5e7 pairs, 9.5 ms of a 15.5 ms `-c hll_a` pass -- the ~87 atomics/us wall of
// DESIGN.md section 4.1 once more; now 0.2 ms.)
constexpr int kEnumSpan = 16384;

__global__ __launch_bounds__(kBlock)
void enum_pairs_kernel(int n, const int* __restrict__ hi, const PassCounters* __restrict__ pc_in,
                       RowMap rm, int n_rows_grid,
                       selhip_int2_t* __restrict__ out, u64 out_cap, PassCounters* __restrict__ pc) {
    __shared__ u64 base_lds;
    int i, i_e;
    rm.tile_rows((int)(blockIdx.x % n_rows_grid), 1, &i, &i_e);
    const int chunk = blockIdx.x / n_rows_grid;
    if (i >= i_e) return;
    const int kmin = max(i + 1, pc_in->z0p1 ? pc_in->z0p1 - 1 : n);
    const long long k0 = (long long)kmin + (long long)chunk * kEnumSpan;
    const long long k_last = min((long long)min(hi[i], n - 1), k0 + kEnumSpan - 1);
    const long long cnt = k_last - k0 + 1;
    if (cnt <= 0) return;                                                     // block-uniform
    if (threadIdx.x == 0) base_lds = atomicAdd(&pc->n_aux_in, (u64)cnt);      // exact total even when the list is too small
    __syncthreads();
    const u64 base = base_lds;
    for (long long t = threadIdx.x; t < cnt; t += kBlock)
        if (base + (u64)t < out_cap) out[base + (u64)t] = selhip_int2_t{i, (int)(k0 + t)};
}

// aux_fused_kernel<FMA, CRIT>: one LANE per pair; U = Ertl-MLE of the union histogram of the two AUXILIARY sketches.
//   CRIT 1 (hll_a):  t_hat = (size_t)U;  t+ = t_hat / (1 + Z*sigma_p);  K+ = ((1+gamma)*e_k - t+)/t+ >= tau
//   CRIT 2 (hll_an): J = ((double)(e_i+e_k) - U)/U;  C = min(1, (1+Z*sigma_p)*e_k/U) * (1+gamma) * S;  J + C >= tau
// zs = (double)(float)(Z*sigma_p) and S (= zs for order_n = 1) are computed on the host in float/double exactly as
// criteria_sketch.hpp:7-20,25-31,39-40 do.  FMA flavour: g++ fuses (1+gamma)*card_B - t_hat_mas (criteria_sketch.hpp:41).
// The lane's histogram column packs TWO bins per dword (bin k in half k & 1 of word k >> 1): a count is at most 2^p_aux <= 32 768
// (every entry point refuses p_aux > 15: at 16 a pair of empty sketches would put 65 536 into one 16-bit bin), and
// the adds of one instruction still go to 64 different dwords (one per lane).  8 KiB per one-wave block instead of 16: twice the
// resident waves, which is what the kernel is short of -- most of a wave's life is the serial f64 solve, not the binning
// (cfg3's 5e7 pairs with hll_a as first criterion: 5.4 -> 4.8 ms; the kernel is then held at 4 waves per SIMD by its 114 VGPRs).
struct LdsColumn {
    const uint32_t* base;       // &hist[lane]
    __device__ __forceinline__ uint32_t operator[](int k) const { return (base[(k >> 1) * kWave] >> ((k & 1) << 4)) & 0xFFFFu; }
};

__device__ __forceinline__ void hist_add_word_packed(uint32_t* __restrict__ col, uint32_t w) {
#pragma unroll
    for (int s = 0; s < 32; s += 8) {
        const uint32_t v = (w >> s) & 63u;        // register values are <= 64-p+1 < 64
        __hip_atomic_fetch_add(col + (v >> 1) * kWave, 1u << ((v & 1u) << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// One LANE per pair, histogram and estimator fused: the auxiliary sketches are small (2^p_aux <= 4096 registers), so a lane
// reads its pair's two rows itself (16-B loads), bins the per-byte max into ITS column of the wave's [bin][lane] LDS
// histogram (one ds_add_u32 per register for 64 pairs at once, same one-instruction address formation as stage 2a) and
// runs the estimator straight from that column -- no counts buffer, no second kernel.  (The earlier form -- one WAVE per
// pair in hll_union_hist_kernel, 128 LDS operations of zeroing and reduction around 4 useful adds at p_aux = 8, then
// aux_filter_kernel on the 256-B histograms -- took 0.89 ms of cfg5's pass.)
template <bool FMA, int CRIT>
__global__ __launch_bounds__(kWave)
void aux_fused_kernel(const uint8_t* __restrict__ aux_hll, int p_aux, const selhip_int2_t* __restrict__ pairs,
                      const u64* __restrict__ n_dev, u64 cap, double relerr_scaled, const u64* __restrict__ ecard, double tau,
                      double zs, double S_sum,
                      selhip_int2_t* __restrict__ out, u64 out_cap, u64* __restrict__ out_count) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[32 * kWave];     // [bin pair][lane], 8 KiB
    __shared__ selhip_int2_t app_lds[kAppendCap];
    const int lane = threadIdx.x;
    u64 n = *n_dev;
    if (n > cap) n = cap;
    const long long nreg = 1ll << p_aux;
    const int n16 = (int)(nreg >> 4);                                        // 16-byte groups per row (p_aux >= 4)
    uint32_t* const col = hist + lane;
    WaveAppender app;
    app.init(app_lds, 0, out, out_cap, out_count);
    for (u64 base = (u64)blockIdx.x * kWave; base < n; base += (u64)gridDim.x * kWave) {
        const u64 j = base + lane;
        const bool live = j < n;
        selhip_int2_t pr{0, 0};
        u64 ea = 0, eb = 1;
        if (live) { pr = pairs[j]; ea = ecard[pr.x]; eb = ecard[pr.y]; }
#pragma unroll 8
        for (int k = 0; k < 32; ++k) col[k * kWave] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const uint4* ra = reinterpret_cast<const uint4*>(aux_hll + (long long)pr.x * nreg);
        const uint4* rb = reinterpret_cast<const uint4*>(aux_hll + (long long)pr.y * nreg);
        for (int c0 = 0; c0 < n16; c0 += 8) {                                // wave-uniform trip count
            uint4 xa[8], xb[8];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (c0 + t < n16) { xa[t] = ra[c0 + t]; xb[t] = rb[c0 + t]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (c0 + t < n16) {
                    hist_add_word_packed(col, max_u8x4(xa[t].x, xb[t].x));
                    hist_add_word_packed(col, max_u8x4(xa[t].y, xb[t].y));
                    hist_add_word_packed(col, max_u8x4(xa[t].z, xb[t].z));
                    hist_add_word_packed(col, max_u8x4(xa[t].w, xb[t].w));
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        LdsColumn c{col};
        const double U = selhip::ertl_ml_estimate<FMA>(c, (unsigned)p_aux, (unsigned)(64 - p_aux), relerr_scaled);
        bool sel = false;
        if (live) {
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        app.push(sel, pr.x, pr.y, lane);
    }
    app.flush(lane);
}

}  // namespace
