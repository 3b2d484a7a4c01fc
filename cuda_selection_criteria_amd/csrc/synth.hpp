// synth.hpp -- deterministic synthetic sketch generator (host + gfx950 device), integer-only.
//
// Replaces the FASTA -> k-mer -> sketch step of the reference's timing harness
// (experiments/src/time_smh_cuda.cpp:181-211 rebuilds SuperMinHash sketches from FASTA through SeqAn)
// with a statistical model so that the selection path can be exercised at BASELINE.json sizes without
// genomes: genomes come in clusters; a cluster owns n_sh shared random elements, each member adds
// n_pr private ones.  Every element is hashed once (splitmix64 of a counter -- no floating point, so
// host and device produce identical bytes):
//   * primary HLL p=14 and auxiliary HLL p_aux: register = max(clz-rank) exactly as
//     sketch/include/sketch/hll.h:886-899 (`add`): index = h >> (64-p), rank = clz(((h<<1)|1) << (p-1)) + 1
//   * SuperMinHash-shaped array of m u64 buckets (format of src/build_sketch.cpp:9-20 / bbmh.h:560):
//     one-permutation hashing, bucket = low bits of a second hash, value = (bucket_rank_hash >> 32),
//     high word 0; empty bucket = ~0ull (bbmh.h:566 initialises h_ to all-ones).
// Expected Jaccard of two members of one cluster = n_sh / (n_sh + n_pr_a + n_pr_b).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SYNTH_HD __host__ __device__ __forceinline__
#else
#define SYNTH_HD inline
#endif

namespace selhip {

struct SynthParams {
    uint64_t seed;
    int32_t  n_genomes;
    int32_t  m;            // SuperMinHash buckets (power of two)
    int32_t  p_aux;        // auxiliary HLL precision (0 = none)
    int32_t  cluster_size; // genomes per cluster (>=1)
    int32_t  mode;         // 0 = "flat": n_sh identical for all clusters; 1 = "spread": log-uniform per cluster
    uint32_t n_sh_lo;      // flat: n_sh = n_sh_lo ; spread: n_sh in [n_sh_lo, n_sh_hi)
    uint32_t n_sh_hi;
};

SYNTH_HD uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// shared-set size of cluster c
SYNTH_HD uint32_t synth_n_shared(const SynthParams& sp, uint32_t cluster) {
    if (sp.mode == 0 || sp.n_sh_hi <= sp.n_sh_lo) return sp.n_sh_lo;
    // integer "log-uniform": pick an octave uniformly, then uniform inside the octave, clipped
    uint64_t h = splitmix64(sp.seed ^ (0xC1u + ((uint64_t)cluster << 8)));
    uint32_t lo = sp.n_sh_lo, hi = sp.n_sh_hi;
    uint32_t octaves = 0;
    while (((uint64_t)lo << (octaves + 1)) <= hi) ++octaves;
    uint32_t o = (uint32_t)(h % (octaves + 1));
    uint64_t base = (uint64_t)lo << o;
    uint64_t span = base;                     // [base, 2*base)
    uint64_t v = base + ((h >> 20) % span);
    if (v >= hi) v = hi - 1;
    return (uint32_t)v;
}

// private-set size of genome g: n_sh * f, f in {0.005, 0.02, 0.1}
SYNTH_HD uint32_t synth_n_private(const SynthParams& sp, uint32_t genome, uint32_t n_sh) {
    uint64_t h = splitmix64(sp.seed ^ (0xA7u + ((uint64_t)genome << 8)));
    uint32_t sel = (uint32_t)(h % 3);
    uint64_t num = sel == 0 ? 5 : (sel == 1 ? 20 : 100);
    return (uint32_t)(((uint64_t)n_sh * num) / 1000);
}

// element e of stream `stream_id` (cluster stream: 2*cluster, private stream: 2*genome+1)
SYNTH_HD uint64_t synth_element(const SynthParams& sp, uint64_t stream_id, uint32_t e) {
    return splitmix64(splitmix64(sp.seed + 0x51ED270B9F3Cull * (stream_id + 1)) + e);
}

// hll.h:886-899
SYNTH_HD void synth_hll_slot(uint64_t h, int p, uint32_t* index, uint32_t* rank) {
    *index = (uint32_t)(h >> (64 - p));
    uint64_t t = ((h << 1) | 1) << (p - 1);
    *rank = (uint32_t)__builtin_clzll(t) + 1;
}

SYNTH_HD void synth_smh_slot(uint64_t elem, int m, uint32_t* bucket, uint64_t* value) {
    uint64_t h2 = splitmix64(elem ^ 0x5DEECE66Dull);
    *bucket = (uint32_t)(h2 & (uint64_t)(m - 1));
    *value = h2 >> 32;
}

}  // namespace selhip
