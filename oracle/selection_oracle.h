/*
 * selection_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's CPU selection path
 * (sanhue903/CUDA_Selection_Criteria: src/selection.cpp, include/criteria_sketch.hpp and the
 * parts of the vendored dnbaker/sketch v0.19.0 hll.h that the path executes).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * (cuda_selection_criteria_amd/) never does.
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_*.py) against
 *   - results.txt (reference golden output, 7 pairs),
 *   - stdout of the reference's own selection.cpp compiled here from /root/reference
 *     (oracle/Makefile -> oracle/_ref/selection) on the influenza fixtures and on synthetic sets,
 *   - hex-exact report()/union_size() values produced by the reference's hll.h (oracle/_ref/hll_kat),
 *   committed as fixtures under tests/golden/.
 */
#ifndef SELECTION_ORACLE_H
#define SELECTION_ORACLE_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int32_t i, k; double jacc; } orc_pair_t;

/* sketch/include/sketch/hll.h:629-688 */
double orc_ertl_ml_estimate(const uint32_t *c /*[64]*/, unsigned p, unsigned q, double relerr);
/* use_fma: 1 = fused at the sites g++ -O3 -march=<FMA host> fuses (reference Makefile build),
 *          0 = every operation rounded separately (-ffp-contract=off build of the reference) */
double orc_ertl_ml_estimate_ex(const uint32_t *c, unsigned p, unsigned q, double relerr, int use_fma);
/* process-wide default flavour used by every other entry point (initially 1) */
void orc_set_fma(int on);
int orc_get_fma(void);
/* hll.h:564-581 (sum_counts): plain byte histogram */
void orc_histogram(const uint8_t *core, size_t n, uint32_t counts[64]);
/* hll.h:1188-1210 (non-joint branch): histogram of max(a[j], b[j]) */
void orc_union_histogram(const uint8_t *a, const uint8_t *b, size_t n, uint32_t counts[64]);
/* hll.h:834-837,862 : report() for estim_ == ERTL_MLE */
double orc_hll_report(const uint8_t *core, unsigned p);
/* hll.h:1188-1210 -> :255-258 */
double orc_hll_union_size(const uint8_t *a, const uint8_t *b, unsigned p);

/* include/criteria_sketch.hpp:45-49 */
int orc_cb(double tau, double card_a, double card_b);
/* include/criteria_sketch.hpp:66-81 */
int orc_smh_a(const uint64_t *v1, const uint64_t *v2, unsigned m, unsigned n_rows, unsigned n_bands);
/* include/criteria_sketch.hpp:7-20 */
float orc_sigma(int p);
/* include/criteria_sketch.hpp:36-43 */
double orc_kota_mas(size_t card_a, size_t card_b, double t_hat, int p, float Z);
/* include/criteria_sketch.hpp:60-64 (union estimate handed in as a double) */
int orc_hll_a_from_union(double tau, size_t card_a, size_t card_b, double union_est, int p, float Z);
/* include/criteria_sketch.hpp:22-34,52-58 */
double orc_cota_n(size_t card_a, size_t card_b, double t_hat, int p, float Z, int order_n);
int orc_hll_an_from_union(double tau, size_t card_a, size_t card_b, double union_est, int p, float Z, int order_n);

/* src/selection.cpp:258-267 (CPU variant: falls through to rows=1,bands=m) */
void orc_banding(unsigned m, float tau_f, int *n_rows, int *n_bands);
/* src/selection_cuda.cpp:119-128 (GPU-driver variant: stays (1,1) when nothing reaches 0.95) */
void orc_banding_cuda_variant(unsigned m, float tau_f, int *n_rows, int *n_bands);

/* on-disk formats: hll.h:1126-1143 (.hll / .hll_<p>), src/selection.cpp:12-33 (.smh<m>) */
/* returns 0 on success; *p_out = np_, core must hold >= 2^np bytes (cap_bytes) */
int orc_read_hll(const char *path, uint8_t *core, size_t cap_bytes, uint32_t *p_out,
                 uint32_t hdr_out[4], double *value_out);
/* returns element count (>=0) or -1; reads min(count, cap) elements */
int64_t orc_read_smh(const char *path, uint64_t *out, size_t cap);

/*
 * The hot loop, src/selection.cpp:270-291 (use_cb=1) and experiments/src/time_smh.cpp:229-257
 * (use_cb=0), on flattened arrays that are ALREADY in ascending-cardinality order.
 *   criterion: 0 = smh_a, 1 = hll_a (selection.cpp:152-173), 2 = hll_an (:206-227),
 *              3 = hll_a AND smh_a (two-stage, BASELINE config 5: intersection of 0 and 1)
 *   hll      : [N][2^p]      primary HLL registers (p = 14 in the reference)
 *   aux_smh  : [N][m]        SuperMinHash buckets          (criterion 0,3)
 *   aux_hll  : [N][2^p_aux]  auxiliary HLL registers       (criterion 1,2,3)
 * Output pairs are in the reference's print order (row i ascending, then k ascending).
 * Returns the number of selected pairs (may exceed cap: only cap are stored), or <0 on error.
 * stats[0] = pairs that reached the aux criterion (after e2==0 / CB), stats[1] = aux survivors.
 */
int64_t orc_select(const uint8_t *hll, unsigned p, const uint64_t *aux_smh, unsigned m,
                   const uint8_t *aux_hll, unsigned p_aux, const double *cards, int64_t N,
                   float tau_f, int n_rows, int n_bands, int use_cb, int criterion,
                   orc_pair_t *out, int64_t cap, int64_t stats[2], int nthreads);

/* std::to_string(double) == sprintf("%f") (selection.cpp:288); returns strlen */
int orc_format_jacc(double j, char *buf, size_t cap);

/* selection.cpp:251-256: the permutation GNU libstdc++'s std::sort produces for `card` ascending, starting from the identity
 * (file-list order) -- tie order included */
void orc_std_sort_perm(const double *card, int64_t n, int64_t *perm);

#ifdef __cplusplus
}
#endif
#endif
