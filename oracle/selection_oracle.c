/*
 * selection_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).  See selection_oracle.h.
 *
 * Every function is a plain-C restatement of the reference code it cites (paths relative to the
 * reference repository root).  Build with -ffp-contract=off: the reference's arithmetic is written
 * as separate IEEE double operations and the restatement keeps them separate.
 */
#include "selection_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * sketch/include/sketch/hll.h:629-688   detail::ertl_ml_estimate(c, p, q, relerr)
 * c is std::array<uint32_t,64>; every mixed uint32/double expression below converts the integer
 * operand to double first, exactly as the usual arithmetic conversions do in the reference.
 *
 * Floating-point contraction.  The reference is built by its Makefile with g++ -O3 -march=native
 * (Makefile:32), i.e. GCC's default -ffp-contract=fast: on any FMA-capable x86 host GCC fuses every
 * product whose only use is an add/sub in the same basic block.  The secant loop stops at a coarse
 * relative step (1e-2/sqrt(m)), so the fused and unfused builds differ in the last bits of the
 * result (seen on the influenza fixtures: report() of genomes 4, 6, 7).  Both flavours are restated:
 *   fma = 1  the products GCC fuses (marked F below) are evaluated with fma()  -> bit-identical to
 *            oracle/_ref/selection, oracle/_ref/hll_kat           (g++ -O3 -march=x86-64-v3)
 *   fma = 0  every operation rounded separately                  -> bit-identical to
 *            oracle/_ref/selection_nofma, oracle/_ref/hll_kat_nofma   (same + -ffp-contract=off)
 * Products whose fusion cannot change the value (0.5*z + c, 0.5*gprev + a: exact scalings) are
 * written unfused in both flavours.
 * ------------------------------------------------------------------------------------------ */
static int g_orc_fma = 1;
void orc_set_fma(int on) { g_orc_fma = on ? 1 : 0; }
int orc_get_fma(void) { return g_orc_fma; }

static inline double muladd(double a, double b, double c, int use_fma)
{
    return use_fma ? __builtin_fma(a, b, c) : a * b + c;
}

double orc_ertl_ml_estimate_ex(const uint32_t *c, unsigned p, unsigned q, double relerr, int use_fma)
{
    const uint64_t m = 1ull << p;                                  /* :641 */
    if (c[q + 1] == m) return INFINITY;                            /* :642 */

    int kMin, kMax;
    for (kMin = 0; c[kMin] == 0; ++kMin) {}                        /* :645 */
    int kMinPrime = kMin > 1 ? kMin : 1;                           /* :646 */
    for (kMax = (int)q + 1; kMax && c[kMax] == 0; --kMax) {}       /* :647 */
    int kMaxPrime = (int)q < kMax ? (int)q : kMax;                 /* :648 */
    double z = 0.;
    for (int k = kMaxPrime; k >= kMinPrime; --k)                   /* :650 */
        z = 0.5 * z + (double)c[k];
    z = ldexp(z, -kMinPrime);                                      /* :651 */
    unsigned cPrime = c[q + 1];                                    /* :652 */
    if (q) cPrime += c[kMaxPrime];                                 /* :653 */
    double gprev;
    double x;
    double a = z + (double)c[0];                                   /* :656 */
    int mPrime = (int)(m - c[0]);                                  /* :657 */
    gprev = z + ldexp((double)c[q + 1], -(int)q);                  /* :658 */
    x = gprev <= 1.5 * a ? (double)mPrime / (0.5 * gprev + a)      /* :659 */
                         : ((double)mPrime / gprev) * log1p(gprev / a);
    gprev = 0;
    double deltaX = x;
    relerr /= sqrt((double)m);                                     /* :662 */
    while (deltaX > x * relerr) {                                  /* :663 */
        int kappaMinus1;
        frexp(x, &kappaMinus1);                                    /* :665 */
        int sh = kMaxPrime + 1 > kappaMinus1 + 2 ? kMaxPrime + 1 : kappaMinus1 + 2;
        double xPrime = ldexp(x, -sh);                             /* :666 */
        double xPrime2 = xPrime * xPrime;
        /* :668  h = xPrime - xPrime2/3 + (xPrime2*xPrime2)*(1./45. - xPrime2/472.5) */
        double h = muladd(xPrime2 * xPrime2, 1. / 45. - xPrime2 / 472.5, xPrime - xPrime2 / 3, use_fma); /* F */
        for (int k = kappaMinus1; k >= kMaxPrime; --k) {           /* :669 */
            double hPrime = 1. - h;
            h = muladd(h, hPrime, xPrime, use_fma) / (xPrime + hPrime);   /* F :671 */
            xPrime += xPrime;
        }
        double g = (double)cPrime * h;                             /* :674 */
        for (int k = kMaxPrime - 1; k >= kMinPrime; --k) {         /* :675 */
            double hPrime = 1. - h;
            h = muladd(h, hPrime, xPrime, use_fma) / (xPrime + hPrime);   /* F :677 */
            xPrime += xPrime;
            g = muladd((double)c[k], h, g, use_fma);               /* F :679 */
        }
        g = muladd(x, a, g, use_fma);                              /* F :681 */
        if (gprev < g && g <= (double)mPrime) deltaX *= (g - (double)mPrime) / (gprev - g); /* :682 */
        else                                  deltaX = 0;
        x += deltaX;
        gprev = g;
    }
    return x * (double)m;                                          /* :687 */
}

double orc_ertl_ml_estimate(const uint32_t *c, unsigned p, unsigned q, double relerr)
{
    return orc_ertl_ml_estimate_ex(c, p, q, relerr, g_orc_fma);
}

/* hll.h:564-581 sum_counts: a histogram of the register bytes (the SIMD code only reorders adds) */
void orc_histogram(const uint8_t *core, size_t n, uint32_t counts[64])
{
    memset(counts, 0, 64 * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) ++counts[core[i] & 63];
}

/* hll.h:1188-1204: counts[max(a[i], b[i])]++ */
void orc_union_histogram(const uint8_t *a, const uint8_t *b, size_t n, uint32_t counts[64])
{
    memset(counts, 0, 64 * sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        uint8_t v = a[i] > b[i] ? a[i] : b[i];
        ++counts[v & 63];
    }
}

/* hll.h:834-837 sum() -> :210-263 calculate_estimate(ERTL_MLE) -> ertl_ml_estimate(counts,p,64-p,1e-2) */
double orc_hll_report(const uint8_t *core, unsigned p)
{
    uint32_t counts[64];
    orc_histogram(core, (size_t)1 << p, counts);
    return orc_ertl_ml_estimate(counts, p, 64 - p, 1e-2);
}

double orc_hll_union_size(const uint8_t *a, const uint8_t *b, unsigned p)
{
    uint32_t counts[64];
    orc_union_histogram(a, b, (size_t)1 << p, counts);
    return orc_ertl_ml_estimate(counts, p, 64 - p, 1e-2);
}

/* include/criteria_sketch.hpp:45-49 */
int orc_cb(double tau, double card_a, double card_b)
{
    double gamma = (double)card_a / card_b;
    return gamma >= tau;
}

/* include/criteria_sketch.hpp:66-81 */
int orc_smh_a(const uint64_t *v1, const uint64_t *v2, unsigned m, unsigned n_rows, unsigned n_bands)
{
    if (n_rows * n_bands != m) return 0;                            /* :67-70 */
    for (unsigned band = 0; band < n_bands; ++band) {
        const uint64_t *x = v1 + (size_t)band * n_rows, *y = v2 + (size_t)band * n_rows;
        unsigned j = 0;
        while (j < n_rows && x[j] == y[j]) ++j;                     /* std::equal */
        if (j == n_rows) return 1;
    }
    return 0;
}

/* include/criteria_sketch.hpp:7-20 : double expression narrowed to float by the return type */
float orc_sigma(int p)
{
    switch (p) {
        case 4: return (float)(1.106 / sqrt((double)(1 << p)));
        case 5: return (float)(1.07 / sqrt((double)(1 << p)));
        case 6: return (float)(1.054 / sqrt((double)(1 << p)));
        case 7: return (float)(1.046 / sqrt((double)(1 << p)));
    }
    return (float)(1.039 / sqrt((double)(1 << p)));
}

/* include/criteria_sketch.hpp:36-43 */
double orc_kota_mas(size_t card_a, size_t card_b, double t_hat, int p, float Z)
{
    double gamma = (double)card_a / (double)card_b;
    float sigma_p = orc_sigma(p);
    float zs = Z * sigma_p;                                         /* float * float */
    double t_hat_mas = t_hat / (1.0 + (double)zs);
    /* :41  ((1.0+gamma)*card_B - t_hat_mas) / t_hat_mas -- the product's only use is the subtraction, so the
     * reference's FMA build fuses it (see the note above orc_ertl_ml_estimate_ex) */
    double K_mas = muladd(1.0 + gamma, (double)card_b, -t_hat_mas, g_orc_fma) / t_hat_mas;
    return K_mas;
}

/* include/criteria_sketch.hpp:60-64 : size_t t_hat = union_size (truncation), then kota_mas >= tau */
int orc_hll_a_from_union(double tau, size_t card_a, size_t card_b, double union_est, int p, float Z)
{
    size_t t_hat = (size_t)union_est;
    double K_mas = orc_kota_mas(card_a, card_b, (double)t_hat, p, Z);
    return K_mas >= tau;
}

/* include/criteria_sketch.hpp:22-34 */
double orc_cota_n(size_t card_a, size_t card_b, double t_hat, int p, float Z, int order_n)
{
    double gamma = (double)card_a / (double)card_b;
    float sigma_p = orc_sigma(p);
    float zs = Z * sigma_p;
    double S = 0;
    double num = 1;
    for (int k = 1; k < order_n + 1; k++) {
        num *= (double)zs;
        S += num;
    }
    double cand = (1.0 + (double)zs) * (double)card_b / t_hat;
    double minimo = cand < 1.0 ? cand : 1.0;                        /* std::min(1.0, cand) */
    return minimo * (1 + gamma) * S;
}

/* include/criteria_sketch.hpp:52-58 */
int orc_hll_an_from_union(double tau, size_t card_a, size_t card_b, double union_est, int p, float Z, int order_n)
{
    double t_hat = union_est;
    double J_hat = ((double)(card_a + card_b) - t_hat) / t_hat;
    double C = orc_cota_n(card_a, card_b, t_hat, p, Z, order_n);
    return (J_hat + C) >= tau;
}

/* src/selection.cpp:258-267.  threshold is a float; pow() is the double libm function, its
 * arguments are float expressions widened to double; P_r is narrowed to float, then compared
 * with the double constant 0.95. */
void orc_banding(unsigned m, float tau_f, int *n_rows, int *n_bands)
{
    int rows = 1, bands = 1;
    for (unsigned band = 1; band <= m; band++) {
        if (m % band != 0) continue;
        bands = (int)band;
        rows = (int)(m / band);
        float P_r = (float)(1.0 - pow(1.0 - pow((double)tau_f, (double)((float)m / (float)band)), (double)(float)band));
        if ((double)P_r >= 0.95) break;
    }
    *n_rows = rows;
    *n_bands = bands;
}

/* src/selection_cuda.cpp:119-128 / experiments/src/time_smh*.cpp: assigns only on success */
void orc_banding_cuda_variant(unsigned m, float tau_f, int *n_rows, int *n_bands)
{
    int rows = 1, bands = 1;
    for (unsigned band = 1; band <= m; band++) {
        if (m % band != 0) continue;
        float P_r = (float)(1.0 - pow(1.0 - pow((double)tau_f, (double)((float)m / (float)band)), (double)(float)band));
        if ((double)P_r >= 0.95) {
            bands = (int)band;
            rows = (int)(m / band);
            break;
        }
    }
    *n_rows = rows;
    *n_bands = bands;
}

/* hll.h:1126-1143 read(gzFile): u32 bf[4]; u32 np_; f64 value_; u8 core[1<<np_] */
int orc_read_hll(const char *path, uint8_t *core, size_t cap_bytes, uint32_t *p_out,
                 uint32_t hdr_out[4], double *value_out)
{
    gzFile fp = gzopen(path, "rb");
    if (!fp) return -1;
    uint32_t bf[4];
    uint32_t np;
    double value;
    int rc = 0;
    if (gzread(fp, bf, sizeof bf) != (int)sizeof bf) rc = -2;
    else if (gzread(fp, &np, sizeof np) != (int)sizeof np) rc = -2;
    else if (gzread(fp, &value, sizeof value) != (int)sizeof value) rc = -2;
    else if (np > 30 || ((size_t)1 << np) > cap_bytes) rc = -3;
    else if (gzread(fp, core, (unsigned)((size_t)1 << np)) != (int)((size_t)1 << np)) rc = -2;
    gzclose(fp);
    if (rc) return rc;
    if (p_out) *p_out = np;
    if (hdr_out) memcpy(hdr_out, bf, sizeof bf);
    if (value_out) *value_out = value;
    return 0;
}

/* src/selection.cpp:12-33 read_smh: u32 count, then count x u64 */
int64_t orc_read_smh(const char *path, uint64_t *out, size_t cap)
{
    gzFile fp = gzopen(path, "rb");
    if (!fp) return -1;
    uint32_t n;
    if (gzread(fp, &n, sizeof n) != (int)sizeof n) { gzclose(fp); return -2; }
    size_t take = n < cap ? n : cap;
    if (take && gzread(fp, out, (unsigned)(take * 8)) != (int)(take * 8)) { gzclose(fp); return -2; }
    gzclose(fp);
    return (int64_t)n;
}

int orc_format_jacc(double j, char *buf, size_t cap)
{
    return snprintf(buf, cap, "%f", j);
}

/* ------------------------------------------------------------------------------------------
 * src/selection.cpp:270-291 (smh_a), :152-173 (hll_a), :206-227 (hll_an) on flattened arrays.
 * The reference indexes maps by file name; here index i IS the sorted rank, which is what the
 * reference's card_name[i] denotes after the sort at :251-256.
 * ------------------------------------------------------------------------------------------ */
typedef struct { orc_pair_t *v; int64_t n, cap; } rowbuf_t;

static void rb_push(rowbuf_t *rb, int32_t i, int32_t k, double j)
{
    if (rb->n == rb->cap) {
        rb->cap = rb->cap ? rb->cap * 2 : 8;
        rb->v = (orc_pair_t *)realloc(rb->v, (size_t)rb->cap * sizeof(orc_pair_t));
    }
    rb->v[rb->n].i = i; rb->v[rb->n].k = k; rb->v[rb->n].jacc = j;
    rb->n++;
}

int64_t orc_select(const uint8_t *hll, unsigned p, const uint64_t *aux_smh, unsigned m,
                   const uint8_t *aux_hll, unsigned p_aux, const double *cards, int64_t N,
                   float tau_f, int n_rows, int n_bands, int use_cb, int criterion,
                   orc_pair_t *out, int64_t cap, int64_t stats[2], int nthreads)
{
    if (N < 0 || criterion < 0 || criterion > 3) return -1;
    const double threshold = (double)tau_f;          /* float threshold widened at each use */
    const float z_score = 1.96f;                     /* selection.cpp:76 */
    const int order_n = 1;                           /* selection.cpp:77 */
    const size_t hll_bytes = (size_t)1 << p;
    const size_t aux_bytes = (size_t)1 << p_aux;
    rowbuf_t *rows = (rowbuf_t *)calloc((size_t)(N > 0 ? N : 1), sizeof(rowbuf_t));
    int64_t evaluated = 0, survivors = 0;
    if (nthreads < 1) nthreads = 1;

    /* card_name.size() - 1 is size_t arithmetic in the reference: N == 0 would wrap; the
     * restatement treats N < 2 as "no pairs". */
#pragma omp parallel for schedule(dynamic) num_threads(nthreads) reduction(+:evaluated, survivors)
    for (int64_t i = 0; i < N - 1; ++i) {
        size_t e1 = (size_t)cards[i];                               /* :275 truncation */
        for (int64_t k = i + 1; k < N; ++k) {
            size_t e2 = (size_t)cards[k];                           /* :280 */
            if (e2 == 0) continue;                                  /* :281 */
            if (use_cb && !orc_cb(threshold, (double)e1, (double)e2)) break;   /* :282-283 */
            ++evaluated;
            int sel = 1;
            if (criterion == 1 || criterion == 3) {
                double u = orc_hll_union_size(aux_hll + (size_t)i * aux_bytes, aux_hll + (size_t)k * aux_bytes, p_aux);
                sel = orc_hll_a_from_union(threshold, e1, e2, u, (int)p_aux, z_score);
            } else if (criterion == 2) {
                double u = orc_hll_union_size(aux_hll + (size_t)i * aux_bytes, aux_hll + (size_t)k * aux_bytes, p_aux);
                sel = orc_hll_an_from_union(threshold, e1, e2, u, (int)p_aux, z_score, order_n);
            }
            if (sel && (criterion == 0 || criterion == 3))
                sel = orc_smh_a(aux_smh + (size_t)i * m, aux_smh + (size_t)k * m, m, (unsigned)n_rows, (unsigned)n_bands);
            if (!sel) continue;                                     /* :285 */
            ++survivors;
            double t = orc_hll_union_size(hll + (size_t)i * hll_bytes, hll + (size_t)k * hll_bytes, p); /* :286 */
            double jacc14 = ((double)e1 + (double)e2 - t) / t;      /* :287 */
            if (jacc14 >= threshold) rb_push(&rows[i], (int32_t)i, (int32_t)k, jacc14);  /* :288 */
        }
    }

    int64_t total = 0;
    for (int64_t i = 0; i < N; ++i) {                               /* :297-300 print order */
        for (int64_t j = 0; j < rows[i].n; ++j) {
            if (total < cap && out) out[total] = rows[i].v[j];
            ++total;
        }
        free(rows[i].v);
    }
    free(rows);
    if (stats) { stats[0] = evaluated; stats[1] = survivors; }
    return total;
}


/* =================================================================================================
 * Sort by cardinality: selection.cpp:251-256 calls std::sort(card_name.begin(), card_name.end(), by .second ascending).
 * std::sort is NOT stable, so where cardinalities tie the order the reference prints depends on the algorithm of the
 * C++ library it was built with.  That algorithm lives outside /root/reference: GNU libstdc++ (the g++ 11.4 of this image),
 * <bits/stl_algo.h> std::__sort = introsort: median-of-three quicksort (std::__introsort_loop, threshold 16, depth limit
 * 2*floor(log2 n), heap-sort fallback std::__partial_sort) followed by std::__final_insertion_sort.  Restated here on an
 * index array (moving an index = moving the element): comp(a, b) = card[a] < card[b]; perm starts as the file-list order,
 * as card_name does (selection.cpp:241-249).  Pinned by tests/golden/expected/ties_* (the reference binary's stdout on a
 * set with duplicated sketches) and against the product's libstdc++ std::sort on random arrays with ties.
 * ================================================================================================= */
#define ORC_LESS(a, b) (card[(a)] < card[(b)])

static void orc_ss_push_heap(int64_t *f, int64_t hole, int64_t top, int64_t value, const double *card)
{
    int64_t parent = (hole - 1) / 2;
    while (hole > top && ORC_LESS(f[parent], value)) { f[hole] = f[parent]; hole = parent; parent = (hole - 1) / 2; }
    f[hole] = value;
}

static void orc_ss_adjust_heap(int64_t *f, int64_t hole, int64_t len, int64_t value, const double *card)
{
    const int64_t top = hole;
    int64_t child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (ORC_LESS(f[child], f[child - 1])) child--;
        f[hole] = f[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        f[hole] = f[child - 1];
        hole = child - 1;
    }
    orc_ss_push_heap(f, hole, top, value, card);
}

static void orc_ss_heapsort(int64_t *f, int64_t n, const double *card)
{   /* std::__partial_sort(first, last, last): __heap_select (= make_heap, the select loop is empty) + __sort_heap */
    if (n >= 2) {
        int64_t parent = (n - 2) / 2;
        for (;;) { orc_ss_adjust_heap(f, parent, n, f[parent], card); if (parent == 0) break; parent--; }
    }
    int64_t last = n;
    while (last > 1) { --last; const int64_t v = f[last]; f[last] = f[0]; orc_ss_adjust_heap(f, 0, last, v, card); }
}

static void orc_ss_unguarded_linear_insert(int64_t *last, const double *card)
{
    const int64_t val = *last;
    int64_t *next = last - 1;
    while (ORC_LESS(val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}

static void orc_ss_insertion_sort(int64_t *first, int64_t *last, const double *card)
{
    if (first == last) return;
    for (int64_t *i = first + 1; i != last; ++i) {
        if (ORC_LESS(*i, *first)) {
            const int64_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(int64_t));
            *first = val;
        } else {
            orc_ss_unguarded_linear_insert(i, card);
        }
    }
}

static void orc_ss_introsort_loop(int64_t *first, int64_t *last, int64_t depth, const double *card)
{
    while (last - first > 16) {
        if (depth == 0) { orc_ss_heapsort(first, last - first, card); return; }
        --depth;
        /* __unguarded_partition_pivot: median of first+1, mid, last-1 moved to *first */
        int64_t *mid = first + (last - first) / 2;
        int64_t *a = first + 1, *b = mid, *c = last - 1, *r = first, t;
#define ORC_SWAP(x, y) (t = *(x), *(x) = *(y), *(y) = t)
        if (ORC_LESS(*a, *b)) {
            if (ORC_LESS(*b, *c)) ORC_SWAP(r, b);
            else if (ORC_LESS(*a, *c)) ORC_SWAP(r, c);
            else ORC_SWAP(r, a);
        } else if (ORC_LESS(*a, *c)) ORC_SWAP(r, a);
        else if (ORC_LESS(*b, *c)) ORC_SWAP(r, c);
        else ORC_SWAP(r, b);
        /* __unguarded_partition(first + 1, last, pivot = first) */
        int64_t *lo = first + 1, *hi = last;
        for (;;) {
            while (ORC_LESS(*lo, *first)) ++lo;
            --hi;
            while (ORC_LESS(*first, *hi)) --hi;
            if (!(lo < hi)) break;
            ORC_SWAP(lo, hi);
            ++lo;
        }
#undef ORC_SWAP
        orc_ss_introsort_loop(lo, last, depth, card);
        last = lo;
    }
}

void orc_std_sort_perm(const double *card, int64_t n, int64_t *perm)
{
    for (int64_t i = 0; i < n; ++i) perm[i] = i;
    if (n <= 0) return;
    int64_t lg = 0;
    for (int64_t t = n; t > 1; t >>= 1) ++lg;                  /* std::__lg */
    orc_ss_introsort_loop(perm, perm + n, 2 * lg, card);
    if (n > 16) {                                              /* __final_insertion_sort */
        orc_ss_insertion_sort(perm, perm + 16, card);
        for (int64_t *i = perm + 16; i != perm + n; ++i) orc_ss_unguarded_linear_insert(i, card);
    } else {
        orc_ss_insertion_sort(perm, perm + n, card);
    }
}
#undef ORC_LESS
