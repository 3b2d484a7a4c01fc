/*
 * build_sketch_oracle.c -- CPU ORACLE for the sketch-construction step (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's src/build_sketch.cpp:26-151 (canonical 31-mers of every FASTA record,
 * HyperLogLog `addh`, SuperMinHash `addh`) and of the vendored dnbaker/sketch v0.19.0 pieces it executes:
 *   sketch/include/sketch/hash.h:42-53      WangHash
 *   sketch/include/sketch/hll.h:886-904     hll add / addh
 *   sketch/include/sketch/bbmh.h:639-670    SuperMinHash::addh  (literal, sequential)
 *   sketch/include/aesctr/wy.h:44-59,98-143 WyHash<uint32_t,1>  (one 64-bit wyhash value = two 32-bit draws, low half first)
 *   sketch/include/sketch/policy.h:8-27     SizePow2Policy (m rounded up to a power of two, mod = & mask)
 * Parity: PINNED by the reference's own output files -- the 40 sketch files it ships in datasets/test_influenzaA
 * (.hll, .hll_8, .smh4, .smh64) and files produced here by oracle/_ref/build_sketch (the reference's program compiled
 * from its sources) for further sizes and for synthetic FASTA with N runs / lower case / short records
 * (tests/golden/, tests/test_build_sketch.py): byte-identical after gunzip.
 *
 * FASTA reading: the reference goes through SeqAn (readRecord into an IupacString); the restatement implements the
 * subset that matters for the k-mer stream: '>' starts a record, sequence lines are concatenated, A/C/G/T in either
 * case are bases (SeqAn upper-cases on conversion to Iupac), every other character resets the k-mer window
 * (build_sketch.cpp:83 `default:`), white space inside sequence lines is skipped.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* src/build_sketch.cpp:26-39 */
uint64_t orcb_canonical_kmer(uint64_t kmer, unsigned k)
{
    uint64_t reverse = 0;
    uint64_t b_kmer = kmer;
    kmer = ((kmer >> 2) & 0x3333333333333333UL) | ((kmer & 0x3333333333333333UL) << 2);
    kmer = ((kmer >> 4) & 0x0F0F0F0F0F0F0F0FUL) | ((kmer & 0x0F0F0F0F0F0F0F0FUL) << 4);
    kmer = ((kmer >> 8) & 0x00FF00FF00FF00FFUL) | ((kmer & 0x00FF00FF00FF00FFUL) << 8);
    kmer = ((kmer >> 16) & 0x0000FFFF0000FFFFUL) | ((kmer & 0x0000FFFF0000FFFFUL) << 16);
    kmer = (kmer >> 32) | (kmer << 32);
    reverse = (((uint64_t)-1) - kmer) >> (8 * sizeof(kmer) - (k << 1));
    return (b_kmer < reverse) ? b_kmer : reverse;
}

/* sketch/hash.h:42-53 */
uint64_t orcb_wang_hash(uint64_t key)
{
    key = (~key) + (key << 21);
    key = key ^ (key >> 24);
    key = (key + (key << 3)) + (key << 8);
    key = key ^ (key >> 14);
    key = (key + (key << 2)) + (key << 4);
    key = key ^ (key >> 28);
    key = key + (key << 31);
    return key;
}

/* sketch/hll.h:886-899 add(hashval) */
void orcb_hll_add(uint8_t *core, unsigned p, uint64_t hashval)
{
    const unsigned q = 64 - p;
    const uint32_t index = (uint32_t)(hashval >> q);
    const uint8_t lzt = (uint8_t)(__builtin_clzll(((hashval << 1) | 1) << (p - 1)) + 1);
    if (core[index] < lzt) core[index] = lzt;
}

/* aesctr/wy.h:44-59 */
static uint64_t wymum(uint64_t x, uint64_t y)
{
    __uint128_t l = x;
    l *= y;
    return (uint64_t)(l ^ (l >> 64));
}
static uint64_t wyhash64_stateless(uint64_t *seed)
{
    *seed += 0x60bee2bee120fc15ull;
    return wymum(*seed ^ 0xe7037ed1a0b428dbull, *seed);
}

/* wy::WyHash<uint32_t, 1>: aesctr/wy.h:98-143 with UNROLL_COUNT = 1 */
typedef struct { uint64_t state, buf; unsigned off; } wy32_t;
static void wy32_init(wy32_t *g, uint64_t seed) { g->state = seed ? seed : 1337ull; g->buf = 0; g->off = 8; }
static uint32_t wy32_next(wy32_t *g)
{
    if (g->off + 4 > 8) { g->buf = wyhash64_stateless(&g->state); g->off = 0; }
    uint32_t r;
    memcpy(&r, (const uint8_t *)&g->buf + g->off, 4);
    g->off += 4;
    return r;
}

/* sketch/bbmh.h:530-670  SuperMinHash<SizePow2Policy, WyHash<uint32_t,1>, uint32_t> */
typedef struct {
    uint64_t a, i;
    uint32_t m, mask;
    uint32_t *p, *q;
    int32_t *b;
    uint64_t *h;
} orcb_smh_t;

static unsigned ilog2u(size_t x) { unsigned l = 0; while (((size_t)1 << (l + 1)) <= x) ++l; return l; }

orcb_smh_t *orcb_smh_new(size_t arg)
{
    /* policy.h:12-19: nelem2arg = ilog2(n) + (n not a power of two); vecsize = 1 << that */
    unsigned lg = ilog2u(arg) + ((arg & (arg - 1)) != 0);
    uint32_t m = (uint32_t)1 << lg;
    orcb_smh_t *s = (orcb_smh_t *)calloc(1, sizeof *s);
    s->m = m; s->mask = m - 1; s->a = m - 1; s->i = 0;
    s->p = (uint32_t *)calloc(m, 4);
    s->q = (uint32_t *)malloc((size_t)m * 4);
    s->b = (int32_t *)calloc(m, 4);
    s->h = (uint64_t *)malloc((size_t)m * 8);
    for (uint32_t t = 0; t < m; ++t) { s->q[t] = (uint32_t)-1; s->h[t] = (uint64_t)-1; }   /* bbmh.h:565-566 */
    s->b[m - 1] = (int32_t)m;                                                                 /* bbmh.h:575 */
    return s;
}
void orcb_smh_free(orcb_smh_t *s) { if (s) { free(s->p); free(s->q); free(s->b); free(s->h); free(s); } }
uint32_t orcb_smh_size(const orcb_smh_t *s) { return s->m; }
const uint64_t *orcb_smh_data(const orcb_smh_t *s) { return s->h; }

/* bbmh.h:639-670 */
void orcb_smh_addh(orcb_smh_t *s, uint64_t item)
{
    wy32_t gen;
    wy32_init(&gen, item ^ 0 /* seed_ */);
    uint64_t j = 0;
    while (j <= s->a) {
        uint32_t k = wy32_next(&gen) & s->mask;
        if ((uint64_t)s->q[j] != s->i) { s->q[j] = (uint32_t)s->i; s->p[j] = (uint32_t)j; }
        if ((uint64_t)s->q[k] != s->i) { s->q[k] = (uint32_t)s->i; s->p[k] = k; }
        uint32_t t = s->p[k]; s->p[k] = s->p[j]; s->p[j] = t;
        uint64_t crj = ((uint64_t)j << 32) | wy32_next(&gen);
        if (crj < s->h[s->p[j]]) {
            uint32_t hi = (uint32_t)(s->h[s->p[j]] >> 32);
            uint32_t jprime = s->m - 1 < hi ? s->m - 1 : hi;
            s->h[s->p[j]] = crj;
            if (j < jprime) {
                --s->b[jprime];
                ++s->b[j];
                while (s->b[s->a] == 0) --s->a;
            }
        }
        ++j;
    }
    ++s->i;
}

/* src/build_sketch.cpp:41-95 / :97-151 on one FASTA(.gz) file: every canonical k-mer goes to the sketches that are
 * not NULL.  hll14: 16384 registers; aux: 1 << p_aux registers.  Returns the number of k-mers added or <0. */
long long orcb_sketch_file(const char *path, unsigned k, uint8_t *hll14, uint8_t *aux, unsigned p_aux, orcb_smh_t *smh)
{
    gzFile fp = gzopen(path, "rb");
    if (!fp) return -1;
    long long added = 0;
    uint64_t kmer = 0;
    unsigned bases = 0;
    int in_header = 0, at_line_start = 1;
    char buf[1 << 16];
    int n;
    while ((n = gzread(fp, buf, sizeof buf)) > 0) {
        for (int t = 0; t < n; ++t) {
            char c = buf[t];
            if (c == '\n') { in_header = 0; at_line_start = 1; continue; }
            if (at_line_start && c == '>') { in_header = 1; at_line_start = 0; kmer = 0; bases = 0; continue; }   /* new record: :60-61 */
            at_line_start = 0;
            if (in_header || c == '\r' || c == ' ' || c == '\t') continue;
            uint8_t two_bit = 0;
            bases++;
            switch (c) {                                                                     /* :68-84 */
                case 'A': case 'a': two_bit = 0; break;
                case 'C': case 'c': two_bit = 1; break;
                case 'G': case 'g': two_bit = 2; break;
                case 'T': case 't': two_bit = 3; break;
                default: two_bit = 0; bases = 0; kmer = 0; break;
            }
            kmer = (kmer << 2) | two_bit;
            kmer = kmer & ((1ULL << (k << 1)) - 1);
            if (bases == k) {
                uint64_t canon = orcb_canonical_kmer(kmer, k);
                if (hll14) orcb_hll_add(hll14, 14, orcb_wang_hash(canon));                    /* hll.h:901-904 addh */
                if (aux) orcb_hll_add(aux, p_aux, orcb_wang_hash(canon));
                if (smh) orcb_smh_addh(smh, canon);
                bases--;
                ++added;
            }
        }
    }
    gzclose(fp);
    return added;
}
