// kernel_sketch.cuh -- synthetic sketches, sketch construction from k-mers (build_sketch), row permutation.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

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
    for (long long seg = (long long)(k - 1) + (long long)threadIdx.x * kSketchSeg; seg < L; seg += (long long)blockDim.x * kSketchSeg) {
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

// Block size: 256 or 1 024 threads (the host picks 1 024 when there are too few genomes to put several blocks on every CU: one
// 4-wave block per CU is ONE wave per SIMD, and a lone wave issues a vector instruction every ~12 cycles -- rocprofv3 PMC, 256 genomes:
// 2.57 ms at 0.98 waves per SIMD).
__global__ __launch_bounds__(1024)
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

    for (int t = threadIdx.x; t < ms; t += (int)blockDim.x) h[t] = ~0ull;
    for (int t = threadIdx.x; t < 16384 / 4; t += (int)blockDim.x) regs[t] = 0;
    for (int t = threadIdx.x; t < (n_aux + 3) / 4; t += (int)blockDim.x) aregs[t] = 0;
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
            for (int t = threadIdx.x; t < ms; t += (int)blockDim.x) la = max(la, (int)min((u64)(m - 1), h[t] >> 32));
            atomicMax(&ctl[0], la);
            __syncthreads();
            const int a = ctl[0];
            __syncthreads();
            if (threadIdx.x == 0) ctl[0] = 0;
            if (a <= J) break;
            if (a > kSketchJmaxParallel) {
                // few k-mers per bucket: run the reference's sequential algorithm literally on one lane
                for (int t = threadIdx.x; t < ms; t += (int)blockDim.x) { h[t] = ~0ull; qq[t] = 0xFFFFFFFFu; pp[t] = 0; bb[t] = 0; }
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
    for (int t = threadIdx.x; t < 16384 / 4; t += (int)blockDim.x) out32[t] = regs[t];
    for (int t = threadIdx.x; t < n_aux; t += (int)blockDim.x) aux_out[g * n_aux + t] = (uint8_t)(aregs[t >> 2] >> ((t & 3) * 8));
    for (int t = threadIdx.x; t < ms; t += (int)blockDim.x) smh_out[g * (long long)m + t] = h[t];
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

}  // namespace
