/*
 * selection_hip.h -- C ABI of libselhip.so: the MI355X (gfx950) replacement for the reference's GPU
 * kernel boundary of the all-pairs sketch-selection step.
 *
 * What it replaces in sanhue903/CUDA_Selection_Criteria (paths relative to that repository):
 *   src/selection_kernels_wrapper.hpp:6-45   struct Result, launch_kernel_smh, launch_kernel_CBsmh
 *   src/selection_kernels.cu:13-177          kernel_smh / kernel_CBsmh and their launchers
 *   include/criteria_sketch_cuda.cuh:11-65   device CB / smh_a / hll_union_card
 *   src/selection_cuda.cpp:152-182           the raw cudaMalloc/cudaMemcpy/launch/copy-back sequence
 *                                            (here: the selhip_ctx_* lifecycle)
 *
 * RESULT SEMANTICS are those of the reference's CPU path src/selection.cpp:270-291 (the parity target
 * named by BASELINE.json), NOT of its CUDA kernels (which use a different HLL estimator, never apply
 * CB and index out of bounds -- SURVEY.md section 2.3):
 *   pair (i,k), i<k in ascending-cardinality rank order, is selected iff
 *     e_k != 0                                   (selection.cpp:281; e = (size_t)cardinality, :275,:280)
 *     [CB modes]  (double)e_i / (double)e_k >= (double)tau_f            (criteria_sketch.hpp:45-49)
 *     smh_a: some band of n_rows consecutive u64 buckets is entirely equal   (criteria_sketch.hpp:66-81)
 *     J = ((double)e_i + (double)e_k - U) / U >= (double)tau_f, U = Ertl-MLE estimate of the register-wise
 *         max of the two p=14 HyperLogLog sketches                     (hll.h:1188-1210 -> :629-688)
 *
 * All functions return 0 on success or a negative SELHIP_E_* code; nothing throws across the ABI.
 * No HIP, torch or C++ types appear in any signature.  Pointers named d_* are DEVICE pointers owned by
 * the caller (hipMalloc, or a torch tensor's data_ptr()); h_* are host pointers.
 */
#ifndef SELECTION_HIP_H
#define SELECTION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELHIP_OK              0
#define SELHIP_E_BADARG       -1   /* null pointer, bad size, rows*bands != m, ...                      */
#define SELHIP_E_HIP          -2   /* a HIP runtime call failed; see selhip_last_error()                */
#define SELHIP_E_OVERFLOW     -3   /* an output buffer was too small; nothing was lost: counts are exact */
#define SELHIP_E_NODEVICE     -4   /* no usable gfx950 device                                           */
#define SELHIP_E_STATE        -5   /* call sequence error (e.g. run before upload)                      */

/* == struct Result of src/selection_kernels_wrapper.hpp:6-9 (same layout: 12 bytes) */
typedef struct { int32_t x, y; float sim; } selhip_result_t;
/* == CUDA int2 used for the pair list (wrapper.hpp:16) */
typedef struct { int32_t x, y; } selhip_int2_t;
/* full-precision output record of the context API: the CPU path prints J with std::to_string(double) */
typedef struct { int32_t i, k; double jaccard; } selhip_pair_t;

/* modes, after the two timed regions of experiments/src/time_smh.cpp:229-257 / :261-292 */
#define SELHIP_MODE_SMH      0     /* "smh_a": every pair i<k (e_k==0 skipped), no CB                    */
#define SELHIP_MODE_CB_SMH   1     /* "CB+smh_a": = src/selection.cpp:270-291                            */

/* stage-1 algorithm */
#define SELHIP_ALGO_AUTO     0
#define SELHIP_ALGO_STREAM   1     /* full m-bucket compare, candidates streamed row-major, query tile in LDS */
#define SELHIP_ALGO_SIG      2     /* 32-bit band signatures joined all-pairs, exact verify of candidates  */
#define SELHIP_ALGO_HASHJOIN 3     /* sub-quadratic: (band, signature) keys radix-sorted, candidates read off the runs of equal
                                      keys, exact verify -- same survivors, but pairs are no longer compared one by one
                                      (never chosen by AUTO; not what the pair-comparisons/s metric measures)          */

/* selection criterion applied before the final HLL-14 Jaccard test (src/selection.cpp -c ...):
 *   SMH_A        src/selection.cpp:228-291   (the north_star path)
 *   HLL_A        src/selection.cpp:122-173   auxiliary HLL p = ctz(aux_bytes), criteria_sketch.hpp:36-43,60-64
 *   HLL_AN       src/selection.cpp:175-227   criteria_sketch.hpp:22-34,52-58 (order_n = 1, Z = 1.96)
 *   HLL_A_SMH_A  both hll_a and smh_a must select the pair (BASELINE.json config 5: "hll_a prefilter +
 *                smh_a two-stage criterion"); evaluated smh_a first, hll_a on its survivors -- the selected
 *                set is the intersection either way */
#define SELHIP_CRIT_SMH_A        0
#define SELHIP_CRIT_HLL_A        1
#define SELHIP_CRIT_HLL_AN       2
#define SELHIP_CRIT_HLL_A_SMH_A  3

/* estimator arithmetic flavour (see csrc/ertl_mle.hpp) */
#define SELHIP_FP_FMA        1     /* = reference built by its Makefile on an FMA-capable x86 host (default) */
#define SELHIP_FP_STRICT     0     /* = reference built with -ffp-contract=off                              */

/* ---------------------------------------------------------------------------------------------------
 * 1. Drop-in launchers: same names and parameter lists as src/selection_kernels_wrapper.hpp:11-45
 *    (CUDA int2 -> selhip_int2_t, Result -> selhip_result_t, void -> int status).
 *    All pointers are device pointers; out must hold total_pairs records (reference contract,
 *    selection_cuda.cpp:164); *out_count is zeroed by the launcher (selection_kernels.cu:137,166).
 *    pairs == NULL selects the IMPLICIT triangle i < k < n of the reference driver (selection_cuda.cpp:146-150) without
 *    materialising the 8 B/pair list: total_pairs must then be n(n-1)/2, which fixes n; cards must be ascending (the
 *    driver's order); this variant runs the all-pairs path of the context API and returns after the pass.
 *    With an explicit list the launchers are asynchronous on the null stream, like the reference.
 *    m_aux is the number of u64 buckets per sketch, m_hll the number of HLL registers (16384).
 *    launch_kernel_smh evaluates every listed pair; launch_kernel_CBsmh additionally applies CB
 *    (the reference kernel of that name does not, selection_kernels.cu:63-117 -- documented defect).
 *    blockSize is accepted and ignored (the kernels choose their own wave64 geometry).
 *    launch_kernel_smh64 / launch_kernel_CBsmh64: the same with int64_t total_pairs and a 64-bit *out_count -- the
 *    reference's `int total_pairs` / `int idx` (selection_kernels.cu:29-30) overflow beyond 2^31 pairs (n > 65 536).
 * --------------------------------------------------------------------------------------------------- */
int launch_kernel_smh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                      const selhip_int2_t* pairs, int total_pairs, double tau,
                      int m_aux, int m_hll, int n_rows, int n_bands,
                      selhip_result_t* out, int* out_count, int blockSize);
int launch_kernel_CBsmh(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                        const selhip_int2_t* pairs, int total_pairs, double tau,
                        int m_aux, int m_hll, int n_rows, int n_bands,
                        selhip_result_t* out, int* out_count, int blockSize);
int launch_kernel_smh64(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                        const selhip_int2_t* pairs, int64_t total_pairs, double tau,
                        int m_aux, int m_hll, int n_rows, int n_bands,
                        selhip_result_t* out, int64_t* out_count, int blockSize);
int launch_kernel_CBsmh64(const uint8_t* main_sketches, const uint64_t* aux_sketches, const double* cards,
                          const selhip_int2_t* pairs, int64_t total_pairs, double tau,
                          int m_aux, int m_hll, int n_rows, int n_bands,
                          selhip_result_t* out, int64_t* out_count, int blockSize);

/* ---------------------------------------------------------------------------------------------------
 * 2. Context API: owns the derived device buffers (truncated cards, CB bounds, band signatures,
 *    candidate / result lists) that selection_cuda.cpp:152-182 handles with raw CUDA calls.
 * --------------------------------------------------------------------------------------------------- */
typedef struct selhip_ctx selhip_ctx;

int  selhip_device_count(void);
int  selhip_ctx_create(selhip_ctx** out, int device);
void selhip_ctx_destroy(selhip_ctx* ctx);
/* message of the last failure on this context (never NULL); ctx may be NULL for creation errors */
const char* selhip_last_error(const selhip_ctx* ctx);

/* Run all work of this context on `hip_stream` (a hipStream_t passed as void*; NULL = null stream). */
int selhip_ctx_set_stream(selhip_ctx* ctx, void* hip_stream);
/* Chunk lanes of a pass (smh_a / hll_a+smh_a): the query rows are cut into `chunks` equal-pair chunks and every chunk runs its
 * whole chain (join, verify, [auxiliary criterion], grouping, HLL union histograms, estimate) on one of two internal streams
 * (the context's own and one more), so that one chunk's short tail kernels run beside the other chunk's join.
 * -1 = automatic (the default: 2 chunks from 5e8 pairs per pass with the signature join, else 1), 0 / 1 = off, 2..8 = chunk
 * count.  Results and counters do not depend on it.  (Round 1's stage-1-stream / stage-2-stream pipeline was replaced.) */
int selhip_ctx_set_pipeline(selhip_ctx* ctx, int chunks);
/* Row interleave for sharding a pass over several devices/ranks: the rows [row_begin, row_end) of the following runs are
 * cut into blocks of block_rows rows (a multiple of 32), dealt boustrophedon: of the n_parts blocks of cycle q the run evaluates
 * block number `part` when q is even and block number n_parts - 1 - part when q is odd (rows get shorter towards the end of the
 * triangle, so a plain round-robin deal would give part 0 the longest row block of every cycle).  Every rank then gets the
 * same share of the pair space AND of the survivors, whatever the triangle's shape (a contiguous equal-pair cut hands the
 * last of 8 ranks ~35 % of all rows, i.e. of all stage-2 work).  n_parts <= 1 switches it off. */
int selhip_ctx_set_row_interleave(selhip_ctx* ctx, int block_rows, int n_parts, int part);
/* Rectangular passes: the following runs only take candidates k >= k_min (in addition to k > i), i.e. rows [row_begin,
 * row_end) x columns [k_min, n).  With the uploaded array = block I followed by block J of a larger sorted set,
 * rows [0, |I|) and k_min = |I| evaluates exactly the pairs I x J (the out-of-core driver below).  Reset to 0 by upload/attach. */
int selhip_ctx_set_candidate_begin(selhip_ctx* ctx, int64_t k_min);
/* Tunables (integers by name; results never depend on them):
 *   "join_q"      query side of the 16-bit signature join: 1 (default) = tile of query rows staged in LDS, broadcast LDS reads
 *                 (sigl_join_kernel); 0 = DPP row broadcast from registers (sig16_join_kernel, the round-1 form)
 *   "join_qt"     query rows per block of the signature join: 0 (default) = automatic (32 below 1e8 pairs per pass, 64 below 4.5e8, 128
 *                 beyond), or a multiple of 16
 *   "join_bits"   16 (default): all-pairs join on 16-bit band signatures packed two per dword, its matches cut back to the
 *                 32-bit candidate set during verification; 15: the same with 15-bit signatures and flag arithmetic made of
 *                 plain VOP2 instructions only (LDS form); 32: join on the 32-bit signatures directly
 *   "join_form"   inner loop of the 16-bit LDS-tile join: 0 (default) = v_xor_b32 + v_pk_min_u16; 1 = zero-half test, three 2-cycle
 *                 instructions per packed dword (v_xor_b32, v_sub_u32, v_bitop3_b32; measured slower)
 *   "join_tri"    1: the LDS-tile join launches only the (tile, candidate block) units above the diagonal (contiguous rows; measured:
 *                 no gain); 0 (default): the rectangle, whose blocks under the diagonal leave at once
 *   "join_db"     1 (default) / 0: double-buffered query batches in the DPP form of the 16-bit join
 *   "join_wpb"    waves per block of the 16-bit join: LDS form 4 (default) or 8, DPP form 1 or 4
 *   "sig_tile"    1 (default): signature build 16 genomes per block with an LDS transpose; 0: one thread per bucket
 *   "sig_cache"   1: the band signatures are kept across the passes of this context (they depend on the sketches and the band shape
 *                 only; upload / attach drop them) -- for callers that run many passes over one set, e.g. every rank of a
 *                 strong-scaled job; 0 (default): every pass builds them
 *   "hist_algo"   stage 2a: -1 (default) / 1 = on the registers' bit planes, written at upload / attach (hll_union_hist_bs_kernel:
 *                 bit-serial max, decode tree, population counts; p = 14 only); 0 = on the byte rows with a lane-private LDS
 *                 histogram (hll_union_hist_runs_kernel).  Takes effect at the next upload / attach.
 *   "hist_run"    pairs a wave of stage 2a takes at a time (0 = automatic: 4 on a grouped list; the byte-row kernel: 1, or 4 with the label order);
 *                 the bit-plane kernel takes at most 64;
 *   "hist_dense_degree"  bit-plane kernel on a grouped list: from this many survivors per query row of the pass (default 32; 0 = always, -1 = never) every
 *                 XCD walks the whole list and takes the pairs whose candidate row hashes to it, so that its L2 keeps an eighth of the
 *                 candidate rows instead of streaming all of them (a dense survivor graph; bench.py --hard: 6.2 -> 1.5 GB per pass from beyond L2,
 *                 DESIGN.md section 4.3);
 *   "hist_blocks" one-wave blocks of the byte-row kernel, "hist_bs_blocks" four-wave blocks of the bit-plane kernel (multiples of 8);
 *   "small_pass"  -1 (default) / 1: a set of up to 2 048 genomes with criterion smh_a takes its whole pass in ONE launch
 *                 (small_pass_kernel: bounds + signatures, a grid barrier, then join, verification, union histograms and estimator
 *                 inside each block; the barrier's wait is bounded and a pass that runs out of patience is repeated on the regular path);
 *                 2: the same through hipLaunchCooperativeKernel (+15 us per launch); 0: the regular chain of launches
 *   "group_min_n" sets of up to this many genomes (default 2 048) skip the stage-2 grouping: two latency-bound launches that buy nothing
 *                 while the whole table stays in cache; 0 = group always
 *   "group_label" stage-2 grouping lays the query-row buckets out by label = a row's smallest partner, so that the pairs of a
 *                 cluster of similar genomes are neighbours in the list and their HLL rows stay in L2 (-1 = automatic: sets
 *                 whose HLL rows exceed 192 MiB and passes of >= 4e8 pairs; 0 off; 1 on);
 *   "timed_kernel" see selhip_ctx_timing.
 * (The library also answers to a few names that are NOT part of this interface -- hooks of its own test-suite and measurement
 * knobs, listed at selhip_ctx_set_param in csrc/selection_kernels.hip.) */
int selhip_ctx_set_param(selhip_ctx* ctx, const char* name, int value);
/* what the context decided (read-only): "hll_khi" (largest p = 14 register value + 1; 0 = no bit planes), "hist_bitplanes",
 * "label_order", "join_tile_rows", "chunks" (chunk lanes of the last pass), "small_pass_used" (the last pass was the one-launch small pass) */
int selhip_ctx_get_param(const selhip_ctx* ctx, const char* name, int* value);
/* Stage 2 grouping (default on): the pairs that reach the HLL-14 stage are bucketed by query row (counting sort) so
 * that waves running side by side on one XCD share their query row in L2.  0 = off (same kernel, list as produced). */
int selhip_ctx_set_stage2_grouping(selhip_ctx* ctx, int enable);
/* SELHIP_FP_FMA (default) or SELHIP_FP_STRICT */
int selhip_ctx_set_fp_mode(selhip_ctx* ctx, int fp_mode);

/* Host -> device upload of sketches already in ascending-cardinality order
 * (the flatten step of selection_cuda.cpp:131-143 + the H2D copies :167-169).
 *   h_hll [n][1<<p_hll] u8, h_aux [n][m] u64, h_cards [n] f64 (may be NULL: computed on the device
 *   from the registers with the Ertl-MLE estimator, == hll_t::report()). */
int selhip_ctx_upload(selhip_ctx* ctx, const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                      int64_t n_genomes, int m, int p_hll);
/* Same, for sketches that already live in device memory (caller keeps ownership and must keep them
 * alive while the context uses them).  d_cards may be NULL (computed). */
int selhip_ctx_attach(selhip_ctx* ctx, const uint8_t* d_hll, const uint64_t* d_aux, const double* d_cards,
                      int64_t n_genomes, int m, int p_hll);

/* Auxiliary HLL sketches (the .hll_<p> files, rank order, [n][1 << p_aux] u8) for the hll_a / hll_an criteria;
 * call after upload/attach of the primary sketches.  attach: device pointer owned by the caller. */
int selhip_ctx_upload_aux_hll(selhip_ctx* ctx, const uint8_t* h_aux_hll, int p_aux);
int selhip_ctx_attach_aux_hll(selhip_ctx* ctx, const uint8_t* d_aux_hll, int p_aux);
/* criterion used by the following selhip_ctx_run* calls (default SELHIP_CRIT_SMH_A); n_rows/n_bands are
 * ignored by HLL_A / HLL_AN */
int selhip_ctx_set_criterion(selhip_ctx* ctx, int criterion);

/* report() of every genome (Ertl-MLE, hll.h:834-837,862) computed on the device: d_cards_out[n]. */
int selhip_hll_cards(selhip_ctx* ctx, const uint8_t* d_hll, int64_t n_genomes, int p, double* d_cards_out);
/* copies the context's cardinalities to the host */
int selhip_ctx_get_cards(selhip_ctx* ctx, double* h_cards_out);

/* One pass of the hot path over query rows [row_begin, row_end) (0, n_genomes = everything;
 * a sub-range is one rank's shard of the pair space: rows are independent).
 * tau_f is the FLOAT threshold of the reference (`float threshold`, selection.cpp:81,103).
 * Results stay on the device until fetched.  Synchronous w.r.t. the context's stream on return. */
int selhip_ctx_run(selhip_ctx* ctx, int mode, int algo, float tau_f, int n_rows, int n_bands,
                   int64_t row_begin, int64_t row_end);
/* Asynchronous variant: enqueues the pass and returns; selhip_ctx_finish() waits and validates. */
int selhip_ctx_run_async(selhip_ctx* ctx, int mode, int algo, float tau_f, int n_rows, int n_bands,
                         int64_t row_begin, int64_t row_end);
int selhip_ctx_finish(selhip_ctx* ctx);

/* statistics of the last finished run:
 *   stats[0] pairs evaluated by the smh_a predicate (after e_k==0 / CB pruning)
 *   stats[1] pairs that passed the auxiliary criterion/criteria (smh_a: pairs with a fully equal band)
 *   stats[2] selected pairs (J >= tau)
 *   stats[3] candidates produced by the signature join (ALGO_SIG; = stats[1] for ALGO_STREAM) */
int selhip_ctx_stats(const selhip_ctx* ctx, int64_t stats[4]);
int64_t selhip_ctx_result_count(const selhip_ctx* ctx);
/* copies min(count, cap) records to the host, sorted by (i,k) = the reference's print order */
int selhip_ctx_fetch(selhip_ctx* ctx, selhip_pair_t* h_out, int64_t cap);
/* device-side view of the unsorted result list (for RCCL gathers without a host round trip) */
int selhip_ctx_result_device(selhip_ctx* ctx, const selhip_pair_t** d_results, int64_t* count);
/* device-to-device copy of min(count, cap) unsorted records into caller memory (e.g. a torch tensor that
 * is then handed to an RCCL collective); asynchronous on the context's stream. */
int selhip_ctx_copy_results(selhip_ctx* ctx, selhip_pair_t* d_dst, int64_t cap);
/* same, framed for a fixed-size collective: d_dst[0] = 16-byte header {u64 count, u64 0}, records from d_dst + 16;
 * d_dst must hold cap_records + 1 records.  Returns SELHIP_E_OVERFLOW (after copying cap_records) if count > cap. */
int selhip_ctx_copy_results_framed(selhip_ctx* ctx, void* d_dst, int64_t cap_records);
/* The same frame, enqueued BEHIND a pass that is still running (between selhip_ctx_run_async and selhip_ctx_finish), so
 * that the caller can also enqueue its collective before it waits: the count is not known on the host yet, so one small
 * kernel reads the device-side counter, writes the header and copies min(count, cap_records) records (d_dst 16-byte
 * aligned).  After selhip_ctx_finish the caller checks selhip_ctx_result_count() <= cap_records and
 * selhip_ctx_last_attempts() == 1 (an overflowing internal list makes finish repeat the pass, which would leave the
 * frame stale) and otherwise frames again. */
int selhip_ctx_copy_results_framed_async(selhip_ctx* ctx, void* d_dst, int64_t cap_records);
/* number of times the last finished run had to enqueue its pass (1 = no internal list overflowed) */
int selhip_ctx_last_attempts(const selhip_ctx* ctx);

/* device time (ms, HIP events on the stream each kernel is launched on) of the named kernel PER PASS, averaged over
 * the passes since the last reset (a pipelined pass launches a kernel once per row chunk: the figure is their sum);
 * names: "prep", "sigbuild", "join", "verify", "stage1", "aux", "group", "hist", "select", "total"; "join_span" = first start to
 * last end of the pass's join launches (chunk lanes run them side by side).  <0 if never launched.
 * selhip_ctx_kernel_launches: launches of that kernel per pass. */
double selhip_ctx_kernel_ms(const selhip_ctx* ctx, const char* name);
double selhip_ctx_kernel_launches(const selhip_ctx* ctx, const char* name);
/* enable: 0 = off, 1 = every kernel scope, 2 = only ONE kernel: the stage-1 kernel ("join" for the signature algorithms,
 * "stage1" otherwise) or, after selhip_ctx_set_param(ctx, "timed_kernel", 1), stage 2a ("hist") -- an event pair costs ~10 us of
 * stream time, so level 2 is what a throughput measurement leaves on, on whichever kernel is the longest of the step.
 * Every call resets the accumulated figures. */
int    selhip_ctx_timing(selhip_ctx* ctx, int enable);

/* ---------------------------------------------------------------------------------------------------
 * 2b. Multi-GPU entry taking a device list: ONE process, one host thread + one context per
 *     device; the pair space is cut into equal-pair row ranges, every device gets a full replica of the (host)
 *     sketches, and the selected-pair lists are gathered -- over RCCL/xGMI (ncclAllGather of framed record buffers on
 *     communicators from ncclCommInitAll; librccl is dlopen'ed on first use) or through the host.
 *     Any criterion (SELHIP_CRIT_*): h_aux_hll / p_aux carry the auxiliary HLL sketches of hll_a, hll_an and the two-stage
 *     criterion of BASELINE configs[4] (NULL / 0 for smh_a).  Rows are dealt to the devices in interleaved blocks of 128
 *     (selhip_ctx_set_row_interleave).  h_out receives min(count, cap) records sorted by (i,k); stats_out (optional) as
 *     selhip_ctx_stats.
 * --------------------------------------------------------------------------------------------------- */
#define SELHIP_GATHER_HOST           0   /* each device's list is fetched and merged on the host              */
#define SELHIP_GATHER_RCCL           1   /* RCCL all_gather; an error if RCCL cannot be initialised           */
#define SELHIP_GATHER_RCCL_OR_HOST   2   /* RCCL if it initialises, host merge otherwise (note in last_error) */
int selhip_multi_select(const int* devices, int n_devices,
                        const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                        const uint8_t* h_aux_hll, int p_aux, int criterion,
                        int64_t n_genomes, int m, int p_hll, int mode, int algo, int fp_mode, float tau_f, int n_rows, int n_bands,
                        int gather, selhip_pair_t* h_out, int64_t cap, int64_t* count_out, int64_t stats_out[4]);

/* ---------------------------------------------------------------------------------------------------
 * 2c. Out-of-core driver: the sketches stay in HOST memory (all n genomes, ascending
 *     cardinality, h_cards required) and only `block_genomes` of them per block are resident on the device at a time.
 *     The pair space is tiled into block pairs (I, J), I <= J: a diagonal block is an ordinary pass over I; an
 *     off-diagonal one uploads I followed by J and runs rows I x candidates J (selhip_ctx_set_candidate_begin), so every
 *     pair is evaluated exactly once and the result -- pairs, Jaccard values, statistics -- is identical to one in-core
 *     pass over the whole set.  n_streams = 2 runs two block pairs at a time (own buffers, own context, own host thread):
 *     the upload of one overlaps the pass of the other; device memory needed ~ n_streams * 2 * block_genomes sketches.
 *     h_aux_hll/p_aux are only needed for criteria other than SELHIP_CRIT_SMH_A (NULL/0 otherwise).
 *     h_out receives min(count, cap) records with GLOBAL ranks, sorted by (i,k); SELHIP_E_OVERFLOW (count still exact)
 *     if cap was too small.
 * --------------------------------------------------------------------------------------------------- */
int selhip_ooc_select(int device, const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                      const uint8_t* h_aux_hll, int p_aux, int criterion,
                      int64_t n, int m, int p_hll, int mode, int algo, int fp_mode, float tau_f, int n_rows, int n_bands,
                      int64_t block_genomes, int n_streams,
                      selhip_pair_t* h_out, int64_t cap, int64_t* count_out, int64_t stats_out[4]);


/* ---------------------------------------------------------------------------------------------------
 * 3. Building blocks (device pointers), used by the launchers above and exposed for tests.
 * --------------------------------------------------------------------------------------------------- */
/* smh_a on an explicit pair list: d_flags[j] = 1 iff pair j has a fully equal band. */
int selhip_smh_a_pairs(const uint64_t* d_aux, int m, int n_rows, int n_bands,
                       const selhip_int2_t* d_pairs, int64_t n_pairs, uint8_t* d_flags, void* hip_stream);
/* union histogram of an explicit pair list: d_counts[j][64] (u32) = counts of max(reg_x, reg_y). */
int selhip_hll_union_hist(const uint8_t* d_hll, int p, const selhip_int2_t* d_pairs, int64_t n_pairs,
                          uint32_t* d_counts, void* hip_stream);
/* The p = 14 registers of n genomes as bit planes, the layout stage 2a reads (csrc/kernel_hllbs.cuh): d_planes[n][6][512] u32,
 * plane b of a genome = bit b of its 16 384 registers; d_gmax[n] u8 = every genome's largest register value; *khi_out = the
 * set's largest register value + 1.  Waits for the stream. */
int selhip_hll_bitslice(const uint8_t* d_hll, int64_t n_genomes, uint32_t* d_planes, uint8_t* d_gmax, int* khi_out, void* hip_stream);
/* union histograms from the bit planes (same result as selhip_hll_union_hist with p = 14); gmax / khi as returned above.  Waits. */
int selhip_hll_union_hist_planes(const uint32_t* d_planes, const uint8_t* d_gmax, int khi, const selhip_int2_t* d_pairs, int64_t n_pairs,
                                 uint32_t* d_counts, void* hip_stream);
/* Ertl-MLE of n histograms: d_est[j] = ertl_ml_estimate(d_counts[j], p, 64-p, 1e-2). */
int selhip_ertl_estimate(const uint32_t* d_counts, int64_t n, int p, int fp_mode, double* d_est, void* hip_stream);
/* bucket-match counts (the by-product the north_star mentions): d_matches[j] = #{b : aux_x[b]==aux_y[b]} */
int selhip_smh_match_counts(const uint64_t* d_aux, int m, const selhip_int2_t* d_pairs, int64_t n_pairs,
                            int32_t* d_matches, void* hip_stream);

/* ---------------------------------------------------------------------------------------------------
 * 4. Synthetic sketches generated directly in HBM (csrc/synth.hpp; stands in for the FASTA->sketch
 *    rebuild of experiments/src/time_smh_cuda.cpp:181-211).  Output is in GENERATION order (not sorted).
 * --------------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed;
    int32_t  n_genomes, m, p_aux, cluster_size, mode;
    uint32_t n_sh_lo, n_sh_hi;
} selhip_synth_t;
int selhip_synth_generate(const selhip_synth_t* sp, int64_t g_begin, int64_t g_end,
                          uint8_t* d_hll /*[g_end-g_begin][16384]*/, uint64_t* d_aux /*[..][m]*/,
                          uint8_t* d_aux_hll /*[..][1<<p_aux] or NULL*/, void* hip_stream);
/* gather rows into rank order: dst[r] = src[perm[r]] for row_bytes-sized rows (device pointers) */
int selhip_permute_rows(const void* d_src, void* d_dst, const int32_t* d_perm, int64_t n_rows,
                        int64_t row_bytes, void* hip_stream);

/* ---------------------------------------------------------------------------------------------------
 * 4b. Sketch construction (the `build_sketch` step, src/build_sketch.cpp:26-151): canonical k-mers of every
 *     genome -> HLL p=14 registers, optional auxiliary HLL (p_aux) and optional SuperMinHash h_[m], byte-identical
 *     to the files the reference writes.  d_codes: one byte per base (0..3 = A,C,G,T; 4 = window reset: non-ACGT
 *     character or record boundary), genomes concatenated; d_offsets[n_genomes+1] (int64) delimit them
 *     (selhost_fasta_codes produces the codes).  d_smh / d_aux_hll may be NULL.  m must already be rounded up to a
 *     power of two (policy.h:12-19), m <= 2048.  One workgroup per genome.
 * --------------------------------------------------------------------------------------------------- */
int selhip_build_sketches(const uint8_t* d_codes, const int64_t* d_offsets, int64_t n_genomes, int k, int m, int p_aux,
                          uint8_t* d_hll, uint64_t* d_smh, uint8_t* d_aux_hll, void* hip_stream);

/* ---------------------------------------------------------------------------------------------------
 * 5. Plain device-memory helpers so that a host program needs no HIP headers (the reference driver
 *    calls cudaMalloc/cudaMemcpy/cudaFree directly, selection_cuda.cpp:160-182).
 * --------------------------------------------------------------------------------------------------- */
int selhip_malloc(void** d_ptr, size_t bytes);
int selhip_free(void* d_ptr);
int selhip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes);
int selhip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes);
int selhip_device_synchronize(void);

const char* selhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SELECTION_HIP_H */
