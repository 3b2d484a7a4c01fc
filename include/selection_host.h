/*
 * selection_host.h -- C ABI of libselhost.so: the host-side (CPU, no GPU needed) half of the
 * MI355X-native selection path.  It mirrors what the reference's GPU drivers do around the kernel
 * launch (paths relative to sanhue903/CUDA_Selection_Criteria):
 *   src/selection_cuda.cpp:19-35,  src/selection.cpp:12-33     read_smh          (.smh<m> files)
 *   sketch/include/sketch/hll.h:1103-1143                      hll_t write/read  (.hll, .hll_<p> files)
 *   src/selection_cuda.cpp:37-57,  src/selection.cpp:36-63     load_file_list
 *   src/selection_cuda.cpp:106-116, src/selection.cpp:247-256  report() + sort by cardinality
 *   src/selection.cpp:258-267 / src/selection_cuda.cpp:119-128 banding parameters (CPU / GPU-driver variant)
 *   src/selection_cuda.cpp:131-143                             flatten in rank order
 *   src/selection.cpp:288,297-300                              output lines "fn1 fn2 <to_string(J)>"
 * plus the host twin of the synthetic generator (csrc/synth.hpp).
 * All functions return 0 on success or a negative SELHOST_E_* code; nothing throws across the ABI.
 */
#ifndef SELECTION_HOST_H
#define SELECTION_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELHOST_OK          0
#define SELHOST_E_BADARG   -1
#define SELHOST_E_IO       -2   /* cannot open / short read (reference: std::runtime_error / ZlibError)      */
#define SELHOST_E_FORMAT   -3   /* header not produced by build_sketch (estimator != ERTL_MLE, wrong p, ...) */
#define SELHOST_E_NOMEM    -4

#define SELHOST_BANDING_CPU   0  /* src/selection.cpp:258-267: falls through to (rows 1, bands m)            */
#define SELHOST_BANDING_CUDA  1  /* src/selection_cuda.cpp:119-128: stays (1,1) if no divisor reaches 0.95    */

const char* selhost_last_error(void);

/* ---- on-disk sketch formats ------------------------------------------------------------------ */
/* .hll / .hll_<p>: gz stream  u32 bf[4] = {is_calculated, estim, jestim, 1}; u32 np; f64 value; u8 core[1<<np] */
int selhost_read_hll(const char* path, uint8_t* core, size_t cap_bytes, uint32_t* p_out,
                     uint32_t hdr_out[4], double* value_out);
int selhost_write_hll(const char* path, const uint8_t* core, uint32_t p);   /* header (0,2,2,1), value -1.0 */
/* .smh<m>: gz stream  u32 count; u64 h_[count]      (src/build_sketch.cpp:9-20) */
int64_t selhost_read_smh(const char* path, uint64_t* out, size_t cap);      /* returns count or <0 */
int selhost_write_smh(const char* path, const uint64_t* v, uint32_t count);

/* FASTA(.gz) -> one byte per base for selhip_build_sketches: 0..3 = A,C,G,T in either case, 4 = k-mer window reset
 * (any other sequence character; one is also emitted at every record start).  Header lines ('>'...) and white space
 * are skipped.  Mirrors what src/build_sketch.cpp:49-84 feeds its k-mer loop through SeqAn's readRecord.
 * Returns the number of codes (only the first `cap` are stored: call again with a larger buffer) or <0. */
int64_t selhost_fasta_codes(const char* path, uint8_t* out, size_t cap);
/* SizePow2Policy::arg2vecsize (sketch/policy.h:12-19): the bucket count a SuperMinHash<>(arg) really has */
uint32_t selhost_smh_vecsize(uint32_t arg);

/* ---- estimator on the host (same header as the device code: csrc/ertl_mle.hpp) ---------------- */
/* fp_mode: 1 = fused like the reference built by its Makefile on an FMA host, 0 = strict */
double selhost_hll_report(const uint8_t* core, unsigned p, int fp_mode);
double selhost_hll_union_size(const uint8_t* a, const uint8_t* b, unsigned p, int fp_mode);
double selhost_ertl_estimate(const uint32_t counts[64], unsigned p, int fp_mode);
double selhost_log1p(double x);     /* the log1p restatement used inside the estimator (test hook) */

/* ---- driver logic ------------------------------------------------------------------------------ */
void selhost_banding(unsigned m, float tau_f, int variant, int* n_rows, int* n_bands);
/* perm[r] = original index of the genome with rank r; std::sort with the reference's comparator */
int selhost_sort_by_card(const double* cards, int64_t n, int32_t* perm);

typedef struct selhost_dataset selhost_dataset;
/* Loads every genome of `list_file` (<name>.hll, <name>.smh<m> and, if p_aux > 0, <name>.hll_<p_aux>),
 * computes report(), sorts ascending and flattens in rank order.  m == 0 skips the .smh files. */
int selhost_dataset_load(selhost_dataset** out, const char* list_file, unsigned m, unsigned p_aux,
                         int fp_mode, int n_threads);
void selhost_dataset_free(selhost_dataset* ds);
int64_t selhost_dataset_size(const selhost_dataset* ds);
const uint8_t*  selhost_dataset_hll(const selhost_dataset* ds);      /* [n][16384]      */
const uint64_t* selhost_dataset_aux(const selhost_dataset* ds);      /* [n][m]          */
const uint8_t*  selhost_dataset_aux_hll(const selhost_dataset* ds);  /* [n][1<<p_aux]   */
const double*   selhost_dataset_cards(const selhost_dataset* ds);    /* [n] ascending   */
const char*     selhost_dataset_name(const selhost_dataset* ds, int64_t rank);

/* "fn1 fn2 0.946107\n" (std::to_string(double) == "%f"); returns bytes written (excluding NUL) or <0 */
int selhost_format_line(const char* fn1, const char* fn2, double jaccard, char* buf, size_t cap);

/* ---- on-disk result format (the reference only prints text, selection.cpp:297-300) ---------
 * A self-contained binary file: the selected pairs as 16-byte records plus the table of genome names their ranks refer to.
 *   header (40 B, little endian): char magic[4] = "SELR"; u32 version = 1; u64 n_pairs; u64 n_names; u64 names_bytes;
 *                                 f32 tau; u32 reserved = 0
 *   names   : n_names NUL-terminated strings, rank order (names_bytes bytes in total)
 *   records : n_pairs x { i32 i; i32 k; f64 jaccard }  (selhost_pair_t == selhip_pair_t), sorted by (i, k)
 * selhost_results_text(file) reproduces the reference's stdout lines "fn1 fn2 <to_string(J)>" byte for byte. */
typedef struct { int32_t i, k; double jaccard; } selhost_pair_t;
int selhost_write_results(const char* path, const selhost_pair_t* pairs, int64_t n_pairs,
                          const char* const* names, int64_t n_names, float tau);
typedef struct selhost_results selhost_results;
int selhost_read_results(selhost_results** out, const char* path);
void selhost_results_free(selhost_results* r);
int64_t selhost_results_count(const selhost_results* r);
int64_t selhost_results_names(const selhost_results* r);
float selhost_results_tau(const selhost_results* r);
const selhost_pair_t* selhost_results_pairs(const selhost_results* r);
const char* selhost_results_name(const selhost_results* r, int64_t rank);
/* text of the whole file into buf (NUL-terminated if it fits); returns the number of bytes needed (excluding NUL) or <0 */
int64_t selhost_results_text(const selhost_results* r, char* buf, size_t cap);

/* ---- synthetic sketches on the host (bit-identical to selhip_synth_generate) ------------------- */
typedef struct {
    uint64_t seed;
    int32_t  n_genomes, m, p_aux, cluster_size, mode;
    uint32_t n_sh_lo, n_sh_hi;
} selhost_synth_t;
int selhost_synth_generate(const selhost_synth_t* sp, int64_t g_begin, int64_t g_end,
                           uint8_t* hll, uint64_t* aux, uint8_t* aux_hll, int n_threads);

/* ---- pair-space sharding (rows are independent; used by the multi-GPU drivers) ------------------
 * Splits query rows [0, n) into `parts` contiguous ranges of (nearly) equal PAIR count given the
 * per-row candidate counts implied by hi[] (hi[i] = last candidate rank of row i; NULL = n-1) and z0.
 * bounds[parts+1] receives the row boundaries. */
int selhost_shard_rows(int64_t n, const int32_t* hi, int64_t z0, int parts, int64_t* bounds);

const char* selhost_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SELECTION_HOST_H */
