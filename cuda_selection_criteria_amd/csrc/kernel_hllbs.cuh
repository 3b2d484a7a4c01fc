// kernel_hllbs.cuh -- stage 2a on BIT-SLICED HLL registers (p = 14): union histograms without LDS atomics.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
//
// What it computes is hll.h:1188-1204 (counts[max(a_r, b_r)]++ over the 16 384 registers of two sketches), bit for bit.
// How: the byte-per-register kernel (hll_union_hist_runs_kernel) issues one `ds_add_u32` per 64 registers and the LDS
// retires one of those every ~4.5 cycles per CU -- 1 150 cycles per pair per CU, whatever the memory system does.  Here a
// genome's registers are resident in HBM as SIX BIT PLANES (plane b = bit b of every register, 512 dwords each: 12 KiB per
// genome instead of 16, written once when the sketches are uploaded / attached), so that one 32-bit VALU operation
// handles 32 registers:
//   * max(a, b) bit-serially:  g = borrow chain of (b - a) from the LSB  (g = a > b),  M_b = g ? A_b : B_b      (3 ops per plane)
//   * the histogram of M by decoding its planes four values at a time: one v_bitop3_b32 picks the registers whose upper bits spell
//     the group, three more masks and four accumulating population counts (`v_bcnt_u32_b32`) give the group's four counts
//     (bs_group_node below).  Only the values below KHI (the pair's largest register value, rounded up to a multiple of 4) are decoded.
//   * the lanes' partial counts (<= 256 each) are packed two per dword and summed over the wave by a TRANSPOSING reduction
//     (v_permlane32_swap / v_permlane16_swap / DPP mirrors: each step halves the number of values a lane carries), 35 VALU
//     operations for all bins, after which lane L holds the totals of two bins and stores them.
// Per 32 x 64 registers: 2 NB + KHI boolean instructions (v_bitop3_b32, 2.2 cycles per SIMD: profiles/r03_bitplane_rate.txt) and KHI
// population counts (4.05 cycles), no LDS at all: ~1 500 cycles per pair per SIMD at KHI = 24, i.e. ~380 per CU.
#pragma once

namespace {

constexpr int kBsPlanes = 6;                                    // register values are < 64 (6 bits)
constexpr int kBsPlaneDwords = 512;                             // 16 384 registers / 32
constexpr int kBsGenomeDwords = kBsPlanes * kBsPlaneDwords;     // 12 KiB per genome

// hll_bitslice_kernel: one wave per genome.  Register r = 4 * ((8 o + jj) * 64 + lane) + s  (byte s of the dword the lane
// loads in step 8 o + jj) becomes bit 8 s + jj of dword o * 64 + lane of every plane: which bit a register lands on is
// irrelevant (a pair's counts are sums over all bits) as long as it is the same for every genome.
// gmax[g] = the largest register value of genome g, *max_val that of the set (values are taken modulo 64, like the byte kernel does).
__global__ __launch_bounds__(kBlock)
void hll_bitslice_kernel(const uint8_t* __restrict__ hll, long long n, uint32_t* __restrict__ bs, uint8_t* __restrict__ gmax,
                         int* __restrict__ max_val) {
    const int lane = threadIdx.x & (kWave - 1);
    uint32_t mx_all = 0;
    for (long long g = (long long)blockIdx.x * kWavesPerBlock + threadIdx.x / kWave; g < n; g += (long long)gridDim.x * kWavesPerBlock) {
        const uint32_t* row = reinterpret_cast<const uint32_t*>(hll + g * 16384);
        uint32_t* dst = bs + g * kBsGenomeDwords;
        uint32_t mx = 0;
#pragma unroll 1
        for (int o = 0; o < 8; ++o) {
            uint32_t d[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) d[jj] = row[(o * 8 + jj) * kWave + lane] & 0x3F3F3F3Fu;
            uint32_t out[kBsPlanes];
#pragma unroll
            for (int b = 0; b < kBsPlanes; ++b) out[b] = 0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const uint32_t w = d[jj];
                mx = max(max(mx, w & 0xFFu), max((w >> 8) & 0xFFu, max((w >> 16) & 0xFFu, w >> 24)));
#pragma unroll
                for (int b = 0; b < kBsPlanes; ++b) out[b] |= ((w >> b) & 0x01010101u) << jj;
            }
#pragma unroll
            for (int b = 0; b < kBsPlanes; ++b) dst[b * kBsPlaneDwords + o * kWave + lane] = out[b];
        }
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, s, kWave));
        if (lane == 0) gmax[g] = (uint8_t)mx;
        mx_all = max(mx_all, mx);
    }
    if (lane == 0 && mx_all) atomicMax(max_val, (int)mx_all);
}

// the lane's 8 dwords of each of the NB low planes of genome g (two 16-byte loads per plane, 1 KiB per wave instruction)
template <int NB>
__device__ __forceinline__ void bs_load(const uint32_t* __restrict__ bs, int g, int lane, uint32_t (&r)[NB][8]) {
    const uint4* p = reinterpret_cast<const uint4*>(bs + (size_t)g * kBsGenomeDwords);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint4 lo = p[b * (kBsPlaneDwords / 4) + lane], hi = p[b * (kBsPlaneDwords / 4) + kWave + lane];
        r[b][0] = lo.x; r[b][1] = lo.y; r[b][2] = lo.z; r[b][3] = lo.w;
        r[b][4] = hi.x; r[b][5] = hi.y; r[b][6] = hi.z; r[b][7] = hi.w;
    }
}

// ---- decoding the histogram of M = max(a, b) from its planes, four values at a time ----------------------------------------
// Group G = the values 4G .. 4G+3.  node = registers whose bits NB-1 .. 2 spell G: ONE v_bitop3_b32 from the planes (NB = 4: two
// inputs, NB = 5: three, NB = 6: a 2-input pre-node of bits 5, 4 and then three inputs).  Four population counts per group,
//   A = |node|,  B = |node & M1|,  C0 = |node & ~M1 & M0|,  C1 = |node & M1 & M0|     (each mask one more instruction)
// give  count(4G) = A - B - C0,  count(4G+1) = C0,  count(4G+2) = B - C1,  count(4G+3) = C1:  per decoded value one 2-cycle
// boolean instruction and one 4-cycle accumulating v_bcnt_u32_b32, and nothing for values nobody holds.
template <int TBL>
__device__ __forceinline__ uint32_t bs_op3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, TBL); }
__device__ __forceinline__ void bs_count(uint32_t& acc, uint32_t mask) {
    // (the accumulating form spelled out: left to itself the compiler counts into fresh registers and sums them with v_add3_u32)
    asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(mask));
}

template <int NB, int G>
__device__ __forceinline__ uint32_t bs_group_node(const uint32_t (&M)[NB], const uint32_t (&pre)[4]) {
    constexpr int s1 = (G & 2) ? 0xCC : 0x33, s0 = (G & 1) ? 0xAA : 0x55;
    if constexpr (NB == 4)      return bs_op3<((G & 2) ? 0xF0 : 0x0F) & ((G & 1) ? 0xCC : 0x33)>(M[3], M[2], M[2]);
    else if constexpr (NB == 5) return bs_op3<((G & 4) ? 0xF0 : 0x0F) & s1 & s0>(M[4], M[3], M[2]);
    else                        return bs_op3<0xF0 & s1 & s0>(pre[G >> 2], M[3], M[2]);
}

// groups [G, GEND) of one column (compile-time recursion)
template <int NB, int G0, int G, int GEND>
__device__ __forceinline__ void bs_groups(const uint32_t (&M)[NB], const uint32_t (&pre)[4], uint32_t (&A)[GEND - G0], uint32_t (&B)[GEND - G0],
                                          uint32_t (&C0)[GEND - G0], uint32_t (&C1)[GEND - G0]) {
    if constexpr (G < GEND) {
        const uint32_t node = bs_group_node<NB, G>(M, pre);
        bs_count(A[G - G0], node);
        bs_count(B[G - G0], node & M[1]);
        bs_count(C0[G - G0], bs_op3<0xF0 & 0x33 & 0xAA>(node, M[1], M[0]));
        bs_count(C1[G - G0], bs_op3<0xF0 & 0xCC & 0xAA>(node, M[1], M[0]));
        bs_groups<NB, G0, G + 1, GEND>(M, pre, A, B, C0, C1);
    }
}

// the groups [G0, GE) of a pair (at most 6: 24 accumulators), all 8 columns; leaves the lane's packed partial counts in
// P[2 G0 .. 2 GE) (two values per dword).  A pair with more values takes another walk over the columns with the maximum recomputed
// (2 NB instructions per column) -- registers, not instructions, are what the kernel is short of.
template <int NB, int G0, int GE, int NP>
__device__ __forceinline__ void bs_decode(const uint32_t (&xa)[NB][8], const uint32_t (&yb)[NB][8], uint32_t (&P)[NP]) {
    constexpr int NG = GE - G0;
    static_assert(NG >= 1 && NG <= 6 && 2 * GE <= NP, "a chunk is 1..6 groups");
    uint32_t A[NG], B[NG], C0[NG], C1[NG];
#pragma unroll
    for (int v = 0; v < NG; ++v) { A[v] = 0; B[v] = 0; C0[v] = 0; C1[v] = 0; }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        // g = a > b: the borrow of b - a, from the LSB: g' = (a & ~b) | (~(a ^ b) & g); then M = g ? a : b.  One v_bitop3_b32 each
        // (truth table = the expression on 0xF0, 0xCC, 0xAA; the compiler's own choice for the select is the 4-cycle v_bfi_b32)
        uint32_t g = 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) g = bs_op3<0xB2>(xa[b][c], yb[b][c], g);
        uint32_t M[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) M[b] = bs_op3<0xCA>(g, xa[b][c], yb[b][c]);
        uint32_t pre[4] = {0u, 0u, 0u, 0u};
        if constexpr (NB == 6) {
            if constexpr (G0 < 4 && GE > 0)   pre[0] = bs_op3<0x0F & 0x33>(M[5], M[4], M[4]);
            if constexpr (G0 < 8 && GE > 4)   pre[1] = bs_op3<0x0F & 0xCC>(M[5], M[4], M[4]);
            if constexpr (G0 < 12 && GE > 8)  pre[2] = bs_op3<0xF0 & 0x33>(M[5], M[4], M[4]);
            if constexpr (G0 < 16 && GE > 12) pre[3] = bs_op3<0xF0 & 0xCC>(M[5], M[4], M[4]);
        }
        bs_groups<NB, G0, G0, GE>(M, pre, A, B, C0, C1);
    }
#pragma unroll
    for (int v = 0; v < NG; ++v) {
        P[2 * (G0 + v)]     = (A[v] - B[v] - C0[v]) | (C0[v] << 16);
        P[2 * (G0 + v) + 1] = (B[v] - C1[v]) | (C1[v] << 16);
    }
}

template <int CTRL>
__device__ __forceinline__ uint32_t bs_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kDppRowMirror = 0x140, kDppRowHalfMirror = 0x141, kDppQuad3210 = 0x1B, kDppQuad1032 = 0xB1;

// Transposing wave reduction of 16 per-lane values: returns, in lane L, the sum over all 64 lanes of P[bs_pidx(L)].
// Each step exchanges half of the values a lane still carries for the partner's copies of the other half and adds:
//   v_permlane32_swap (lanes 32-63 of a <-> lanes 0-31 of b), v_permlane16_swap (odd rows of a <-> even rows of b),
//   then inside a row of 16 lanes: DPP row_mirror / row_half_mirror / quad_perm with a select per pair of values.
__device__ __forceinline__ uint32_t bs_reduce16(const uint32_t (&P)[16], int lane) {
    uint32_t Q[8], R[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const auto s = __builtin_amdgcn_permlane32_swap(P[2 * j], P[2 * j + 1], false, false);
        Q[j] = s[0] + s[1];                 // lanes 0-31: P[2j] over {i, i+32}; lanes 32-63: P[2j+1]
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const auto s = __builtin_amdgcn_permlane16_swap(Q[2 * j], Q[2 * j + 1], false, false);
        R[j] = s[0] + s[1];                 // rows 0..3: P[4j], P[4j+2], P[4j+1], P[4j+3], each over 4 source lanes per lane
    }
    const bool h = (lane & 8) != 0, q = (lane & 4) != 0;
    const uint32_t t0 = R[0] + bs_dpp<kDppRowMirror>(R[0]), t1 = R[1] + bs_dpp<kDppRowMirror>(R[1]);
    const uint32_t t2 = R[2] + bs_dpp<kDppRowMirror>(R[2]), t3 = R[3] + bs_dpp<kDppRowMirror>(R[3]);
    const uint32_t S0 = h ? t1 : t0, S1 = h ? t3 : t2;
    const uint32_t u0 = S0 + bs_dpp<kDppRowHalfMirror>(S0), u1 = S1 + bs_dpp<kDppRowHalfMirror>(S1);
    const uint32_t U = q ? u1 : u0;
    const uint32_t V = U + bs_dpp<kDppQuad3210>(U);
    return V + bs_dpp<kDppQuad1032>(V);
}
__device__ __forceinline__ int bs_pidx(int lane) {
    const int r = lane >> 4;
    return 4 * (2 * ((lane >> 2) & 1) + ((lane >> 3) & 1)) + (((r & 1) << 1) | (r >> 1));
}

// union histogram of one pair from the two rows' planes: returns, in lane L, the packed totals of the bins 2 * pidx(L), 2 * pidx(L) + 1
// (lanes 0, 1 of a quad) or 32 + ... (lanes 2, 3).  kp (wave-uniform) = values the pair can hold: every register of both rows is < kp.
// The first 24 values (16 with four planes) are always decoded; the walks for 24.., 32.., 48.. only when the pair can hold such values
// -- a typical sketch's largest register is ~log2(cardinality) - 11 while a set of 10 000 reaches 30.  (One code path with optional
// walks, not a body per size: with several bodies in one kernel the compiler hoists their common maximum computation above the
// branch and keeps it live -- 84 VGPRs for one body, 166 for two.)
template <int NB>
__device__ __forceinline__ uint32_t bs_pair_hist(const uint32_t (&xa)[NB][8], const uint32_t (&yb)[NB][8], int kp, int lane) {
    constexpr int NP = NB == 6 ? 32 : 16;                                 // packed values: two bins per dword (a wave total is <= 16 384)
    uint32_t P[NP];
#pragma unroll
    for (int v = 0; v < NP; ++v) P[v] = 0u;
    if constexpr (NB == 4) {
        bs_decode<NB, 0, 4, NP>(xa, yb, P);
    } else {
        bs_decode<NB, 0, 6, NP>(xa, yb, P);
        if (kp > 24) bs_decode<NB, 6, 8, NP>(xa, yb, P);
        if constexpr (NB == 6) {
            if (kp > 32) bs_decode<NB, 8, 12, NP>(xa, yb, P);
            if (kp > 48) bs_decode<NB, 12, 16, NP>(xa, yb, P);
        }
    }
    // the transposing reduction, 16 packed values at a time
    uint32_t lo[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) lo[v] = P[v];
    uint32_t tot = bs_reduce16(lo, lane);
    if constexpr (NP > 16) {
        uint32_t tot_hi = 0;
        if (kp > 32) {
#pragma unroll
            for (int v = 0; v < 16; ++v) lo[v] = P[16 + v];
            tot_hi = bs_reduce16(lo, lane);
        }
        tot = (lane & 2) ? tot_hi : tot;
    } else {
        tot = (lane & 2) ? 0u : tot;
    }
    return tot;
}

// hll_union_hist_bs_kernel<NB>: same contract as hll_union_hist_runs_kernel (window [chunk_off, chunk_off + chunk_len) of the
// pair list, counts indexed from the window start, tasks of run_len (at most 64) consecutive pairs, block b works in the (b % 8)-th eighth of
// the tasks = on XCD b % 8), on the bit planes.  NB = planes that can be non-zero in the SET (4, 5 or 6); how many values are
// decoded is decided PER PAIR from the two rows' largest register values (gmax, written with the planes).
// One wave per pair: a lane owns 8 dwords (256 registers) of every plane; the query row's planes stay in registers across a run.
template <int NB>
__global__ __launch_bounds__(kBlock, NB <= 5 ? 4 : 2)
void hll_union_hist_bs_kernel(const uint32_t* __restrict__ bs, const uint8_t* __restrict__ gmax, const selhip_int2_t* __restrict__ pairs,
                              const u64* __restrict__ n_pairs_dev, u64 cap, uint32_t* __restrict__ counts,
                              u64 chunk_off, u64 chunk_len, int run_len, u64 dense_pairs) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    u64 n_pairs = *n_pairs_dev;
    if (n_pairs > cap) n_pairs = cap;
    n_pairs = n_pairs > chunk_off ? min(n_pairs - chunk_off, chunk_len) : 0;
    pairs += chunk_off;
    // A SHORT list runs on part of the grid: the device holds 1 024 SIMDs x (4 or 2) waves of this kernel at a time, and a wave's first
    // pair costs three dependent round trips (count, pair record, rows) before any arithmetic.  With 8 192 waves for 16 000 pairs (one of
    // 8 ranks of the weak-scaled workload) every wave took two pairs and half of the waves could only start when the first half had
    // ended: 44 us for 16 us of arithmetic.  Up to eight pairs per resident wave the list is therefore dealt to at most ONE round of
    // waves (the other blocks leave at once); longer lists keep the whole grid, whose two rounds of shorter waves end more evenly.
    u64 blocks_active = gridDim.x;
    {
        const u64 resident = 1024ull * (NB <= 5 ? 4 : 2);
        if (n_pairs <= 8 * resident) {
            const u64 waves = min(resident, max(n_pairs, (u64)64));
            blocks_active = min((u64)gridDim.x, ((waves + 8 * kWavesPerBlock - 1) / (8 * kWavesPerBlock)) * 8);
        }
    }
    if (blockIdx.x >= blocks_active) return;
    const u64 stride = (blocks_active >> 3) * kWavesPerBlock;             // waves per XCD (the host launches a multiple of 8 blocks)
    const int my_bin = ((lane & 2) ? 32 : 0) + 2 * bs_pidx(lane) + (lane & 1);
    // A DENSE survivor graph (n_pairs >= dense_pairs; the host sets 32 pairs per query row of the pass on a grouped list, "hist_dense_degree"): a query
    // row has tens to hundreds of partners spread over the whole table, the table does not fit one L2 (4 MiB = ~400 rows) and in list
    // order every pair fetches its candidate row from beyond it (the 25 %-degenerate set of bench.py: 730 000 pairs among 2 516 genomes,
    // 6.2 GB per pass at 92 % of the Infinity Cache's gather rate).  Then every XCD walks the WHOLE list, 64 pairs per task, and takes
    // the pairs whose candidate row hashes to it: its L2 only ever sees an eighth of the candidate rows (they stay) plus the query rows,
    // which all its waves pass through together (the list is query-major).  Same counts in the same slots either way.
    const bool dense = n_pairs >= dense_pairs;
    const uint32_t xcd = blockIdx.x & 7u;
    // Runs only pay (the query row stays in registers) once every wave has several of them: on the whole grid a list is dealt out in
    // runs of n_pairs / 16 per wave at most (one of 8 ranks of cfg4, 56 000 pairs on 8 192 waves: 49.9 us with single pairs, 58.3 us with
    // runs of 4).  A short list on its one round of waves gives every wave ONE run of its n_pairs / waves consecutive pairs: with single
    // pairs every pair fetched both of its rows (one of 8 ranks at 28 280 genomes: 16 000 pairs, 41 us for 16 us of arithmetic).
    const u64 waves_active = 8 * stride;
    // (runs of at most 2 there: longer ones put the uses of a candidate row at different points of different waves' serial chains
    //  and it leaves L2 in between -- one chunk lane of a rank of cfg4, 28 000 pairs, table 512 MB: 47.3 / 49.2 / 56.3 us with runs of
    //  1 / 2 / 4; the 16 000-pair share above: 40.1 / 37.1 / 37.3)
    const bool short_list = blocks_active < gridDim.x;
    const u64 per_wave = short_list ? (n_pairs + waves_active - 1) / waves_active : n_pairs / (16 * stride);
    run_len = dense ? kWave : (int)max((u64)1, min((u64)min(run_len, short_list ? 2 : kWave), per_wave));
    const u64 n_tasks = (n_pairs + run_len - 1) / run_len;
    const u64 tasks_per_xcd = dense ? n_tasks : (n_tasks + 7) >> 3;
    const u64 t_begin = dense ? 0 : (u64)xcd * tasks_per_xcd, t_end = min(t_begin + tasks_per_xcd, n_tasks);
    int cur_x = -1;
    uint32_t xa[NB][8];
    for (u64 task = t_begin + (u64)(blockIdx.x >> 3) * kWavesPerBlock + wave; task < t_end; task += stride) {
        const u64 j0 = task * run_len;
        const int cnt = (int)min((u64)run_len, n_pairs - j0);
        selhip_int2_t pr{0, 0};
        if (lane < cnt) pr = pairs[j0 + lane];                            // the task's pairs, one per lane
        u64 mine = __ballot(lane < cnt && (!dense || (((uint32_t)pr.y * 0x9E3779B1u) >> 29) == xcd));
        while (mine) {
            const int src = (int)__builtin_ctzll(mine);
            mine &= mine - 1;
            const int px = __builtin_amdgcn_readlane(pr.x, src), py = __builtin_amdgcn_readlane(pr.y, src);   // wave-uniform: scalar row addresses
            uint32_t yb[NB][8];
            bs_load<NB>(bs, py, lane, yb);
            if (px != cur_x) { bs_load<NB>(bs, px, lane, xa); cur_x = px; }
            const int kp = max((int)gmax[px], (int)gmax[py]) + 1;         // values this pair can hold: [0, kp)
            const uint32_t tot = bs_pair_hist<NB>(xa, yb, kp, lane);
            counts[(j0 + src) * 64 + my_bin] = (lane & 1) ? (tot >> 16) : (tot & 0xFFFFu);
        }
    }
}

// (Round 3 also built this kernel with the NEXT pair's candidate planes prefetched through LDS -- every wave a slot of NB x 2 KiB filled by
// LDS-DMA (`global_load_lds_dwordx4`) while the current pair is decoded, pair records and maxima read two pairs ahead; bit-identical in
// the parity suite.  It changed nothing: cfg3 53.7 -> 56.6 us, hard 0.874 -> 0.878 ms, cfg4 / cfg5 equal (gpurun_out/r03/g_pf*).  The PMC
// pass says why (profiles/r03b_pmc_summary.json): the vector units are ACTIVE for 0.71 of the kernel at cfg3 and 0.83 on the hard set --
// ~4.1 cycles per instruction: in this mix of dependent chains the boolean instructions do not reach the 2.2 cycles they show alone --
// so the waves are not waiting for memory, they are waiting for each other's arithmetic.  Dropped; what is left to gain here is
// instruction count.)

}  // namespace
