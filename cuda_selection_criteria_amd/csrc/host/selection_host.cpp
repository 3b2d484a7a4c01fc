// selection_host.cpp -- libselhost.so: host-side half of the MI355X-native selection path.
// See include/selection_host.h for the reference code each entry point mirrors.
#include "../../../include/selection_host.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../ertl_mle.hpp"
#include "../synth.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

double relerr_scaled_for(unsigned p) { return 1e-2 / std::sqrt((double)(1ull << p)); }   // hll.h:662

double estimate(const uint32_t* counts, unsigned p, int fp_mode) {
    return fp_mode ? selhip::ertl_ml_estimate<true>(counts, p, 64 - p, relerr_scaled_for(p))
                   : selhip::ertl_ml_estimate<false>(counts, p, 64 - p, relerr_scaled_for(p));
}

}  // namespace

struct selhost_dataset {
    int64_t n = 0;
    unsigned m = 0, p_aux = 0;
    std::vector<std::string> names;       // rank order
    std::vector<uint8_t> hll;
    std::vector<uint64_t> aux;
    std::vector<uint8_t> aux_hll;
    std::vector<double> cards;
};

extern "C" {

const char* selhost_version(void) { return "selhost 0.1"; }
const char* selhost_last_error(void) { return g_err.c_str(); }

// ---- formats ------------------------------------------------------------------------------------
int selhost_read_hll(const char* path, uint8_t* core, size_t cap_bytes, uint32_t* p_out,
                     uint32_t hdr_out[4], double* value_out) {
    if (!path || !core) return fail(SELHOST_E_BADARG, "null argument");
    gzFile fp = gzopen(path, "rb");
    if (!fp) return fail(SELHOST_E_IO, "Could not open file at '%s' for reading", path);   // hll.h:1147
    uint32_t bf[4], np = 0;
    double value = 0;
    int rc = SELHOST_OK;
    if (gzread(fp, bf, sizeof bf) != (int)sizeof bf || gzread(fp, &np, sizeof np) != (int)sizeof np ||
        gzread(fp, &value, sizeof value) != (int)sizeof value)
        rc = fail(SELHOST_E_IO, "Error reading from file '%s'", path);
    else if (np > 30 || ((size_t)1 << np) > cap_bytes)
        rc = fail(SELHOST_E_FORMAT, "'%s': precision %u does not fit the caller's buffer (%zu bytes)", path, np, cap_bytes);
    else if (gzread(fp, core, (unsigned)((size_t)1 << np)) != (int)((size_t)1 << np))
        rc = fail(SELHOST_E_IO, "Error reading from file '%s'", path);
    gzclose(fp);
    if (rc) return rc;
    if (p_out) *p_out = np;
    if (hdr_out) std::memcpy(hdr_out, bf, sizeof bf);
    if (value_out) *value_out = value;
    return SELHOST_OK;
}

int selhost_write_hll(const char* path, const uint8_t* core, uint32_t p) {
    if (!path || !core || p > 30) return fail(SELHOST_E_BADARG, "bad argument");
    gzFile fp = gzopen(path, "wb");
    if (!fp) return fail(SELHOST_E_IO, "Could not open file at '%s' for writing", path);
    const uint32_t bf[4] = {0, 2, 2, 1};            // is_calculated=0, ERTL_MLE, J_ERTL_MLE, 1 (hll.h:1105)
    const double value = -1.0;
    int ok = gzwrite(fp, bf, sizeof bf) && gzwrite(fp, &p, sizeof p) && gzwrite(fp, &value, sizeof value) &&
             gzwrite(fp, core, (unsigned)((size_t)1 << p));
    gzclose(fp);
    return ok ? SELHOST_OK : fail(SELHOST_E_IO, "Error writing to file '%s'", path);
}

int64_t selhost_read_smh(const char* path, uint64_t* out, size_t cap) {
    if (!path || (!out && cap)) return fail(SELHOST_E_BADARG, "null argument");
    gzFile fp = gzopen(path, "rb");
    if (!fp) return fail(SELHOST_E_IO, "Could not open file at '%s' for reading", path);   // selection.cpp:16
    uint32_t n = 0;
    if (gzread(fp, &n, sizeof n) != (int)sizeof n) { gzclose(fp); return fail(SELHOST_E_IO, "Error reading from file '%s'", path); }
    size_t take = std::min<size_t>(n, cap);
    if (take && gzread(fp, out, (unsigned)(take * 8)) != (int)(take * 8)) { gzclose(fp); return fail(SELHOST_E_IO, "Error reading from file '%s'", path); }
    gzclose(fp);
    return (int64_t)n;
}

int selhost_write_smh(const char* path, const uint64_t* v, uint32_t count) {
    if (!path || (!v && count)) return fail(SELHOST_E_BADARG, "null argument");
    gzFile fp = gzopen(path, "wb");
    if (!fp) return fail(SELHOST_E_IO, "Could not open file at '%s' for writing", path);
    int ok = gzwrite(fp, &count, sizeof count) != 0;
    if (ok && count) ok = gzwrite(fp, v, (unsigned)((size_t)count * 8)) != 0;
    gzclose(fp);
    return ok ? SELHOST_OK : fail(SELHOST_E_IO, "Error writing to file '%s'", path);
}

int64_t selhost_fasta_codes(const char* path, uint8_t* out, size_t cap) {
    if (!path || (!out && cap)) return fail(SELHOST_E_BADARG, "null argument");
    gzFile fp = gzopen(path, "rb");
    if (!fp) return fail(SELHOST_E_IO, "ERROR: Could not open the file %s.", path);      // build_sketch.cpp:44-48
    int64_t n = 0;
    bool in_header = false, at_line_start = true;
    std::vector<char> buf(1 << 16);
    int got;
    auto emit = [&](uint8_t c) { if ((size_t)n < cap) out[n] = c; ++n; };
    while ((got = gzread(fp, buf.data(), (unsigned)buf.size())) > 0) {
        for (int t = 0; t < got; ++t) {
            const char c = buf[(size_t)t];
            if (c == '\n') { in_header = false; at_line_start = true; continue; }
            if (at_line_start && c == '>') { in_header = true; at_line_start = false; emit(4); continue; }   // new record
            at_line_start = false;
            if (in_header || c == '\r' || c == ' ' || c == '\t') continue;
            switch (c) {                                                                 // build_sketch.cpp:68-84
                case 'A': case 'a': emit(0); break;
                case 'C': case 'c': emit(1); break;
                case 'G': case 'g': emit(2); break;
                case 'T': case 't': emit(3); break;
                default: emit(4); break;
            }
        }
    }
    gzclose(fp);
    return n;
}

uint32_t selhost_smh_vecsize(uint32_t arg) {
    if (arg <= 1) return 1;
    uint32_t lg = 0;
    while ((1ull << (lg + 1)) <= arg) ++lg;
    lg += (arg & (arg - 1)) != 0;
    return 1u << lg;
}

// ---- estimator ------------------------------------------------------------------------------------
double selhost_ertl_estimate(const uint32_t counts[64], unsigned p, int fp_mode) { return estimate(counts, p, fp_mode); }
double selhost_log1p(double x) { return selhip::log1p_fdlibm(x); }

double selhost_hll_report(const uint8_t* core, unsigned p, int fp_mode) {
    uint32_t counts[64] = {0};
    const size_t n = (size_t)1 << p;
    for (size_t i = 0; i < n; ++i) ++counts[core[i] & 63];           // hll.h:834 sum_counts
    return estimate(counts, p, fp_mode);
}

double selhost_hll_union_size(const uint8_t* a, const uint8_t* b, unsigned p, int fp_mode) {
    uint32_t counts[64] = {0};
    const size_t n = (size_t)1 << p;
    for (size_t i = 0; i < n; ++i) ++counts[std::max(a[i], b[i]) & 63];   // hll.h:1188-1204
    return estimate(counts, p, fp_mode);
}

// ---- driver logic -----------------------------------------------------------------------------------
void selhost_banding(unsigned m, float threshold, int variant, int* n_rows_out, int* n_bands_out) {
    int n_rows = 1, n_bands = 1;
    for (unsigned band = 1; band <= m; band++) {
        if (m % band != 0) continue;
        if (variant == SELHOST_BANDING_CPU) {                         // selection.cpp:261-262 assigns first
            n_bands = (int)band;
            n_rows = (int)(m / band);
        }
        // selection.cpp:263: float threshold, float exponent, double pow(), result narrowed to float
        float P_r = (float)(1.0 - std::pow(1.0 - std::pow((double)threshold, (double)((float)m / (float)band)), (double)(float)band));
        if ((double)P_r >= 0.95) {
            n_bands = (int)band;
            n_rows = (int)(m / band);
            break;
        }
    }
    if (n_rows_out) *n_rows_out = n_rows;
    if (n_bands_out) *n_bands_out = n_bands;
}

int selhost_sort_by_card(const double* cards, int64_t n, int32_t* perm) {
    if (n < 0 || (n && (!cards || !perm))) return fail(SELHOST_E_BADARG, "bad argument");
    std::vector<std::pair<int32_t, double>> v((size_t)n);
    for (int64_t i = 0; i < n; ++i) v[(size_t)i] = {(int32_t)i, cards[i]};
    // same algorithm (libstdc++ std::sort) and comparator as selection.cpp:251-256: the sequence of
    // swaps depends only on comparator outcomes, so ties land where the reference puts them
    std::sort(v.begin(), v.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) {
        return x.second < y.second;
    });
    for (int64_t i = 0; i < n; ++i) perm[i] = v[(size_t)i].first;
    return SELHOST_OK;
}

static int load_file_list(const char* list_file, std::vector<std::string>& files) {
    if (!list_file || !*list_file) return fail(SELHOST_E_BADARG, "No input file provided");      // selection.cpp:40-44
    std::ifstream file(list_file);
    if (!file.is_open()) return fail(SELHOST_E_IO, "No valid input file provided");                // selection.cpp:48-52
    std::string line;
    while (getline(file, line)) {
        line.erase(0, line.find_first_not_of(" \t\r\n"));                                           // selection.cpp:56-57
        line.erase(line.find_last_not_of(" \t\r\n") + 1);
        if (!line.empty()) files.push_back(line);
    }
    return SELHOST_OK;
}

int selhost_dataset_load(selhost_dataset** out, const char* list_file, unsigned m, unsigned p_aux,
                         int fp_mode, int n_threads) {
    if (!out) return fail(SELHOST_E_BADARG, "null argument");
    *out = nullptr;
    std::vector<std::string> files;
    int rc = load_file_list(list_file, files);
    if (rc) return rc;
    const int64_t n = (int64_t)files.size();
    const unsigned p = 14;
    const size_t hb = (size_t)1 << p, ab = p_aux ? (size_t)1 << p_aux : 0;
    std::vector<uint8_t> hll((size_t)n * hb), auxh((size_t)n * ab);
    std::vector<uint64_t> aux((size_t)n * m);
    std::vector<double> cards((size_t)n);
    int err = SELHOST_OK;
    std::string err_msg;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for (int64_t i = 0; i < n; ++i) {
        int lrc = SELHOST_OK;
        uint32_t np = 0, hdr[4] = {0, 0, 0, 0};
        double value = 0;
        const std::string& fn = files[(size_t)i];
        lrc = selhost_read_hll((fn + ".hll").c_str(), hll.data() + (size_t)i * hb, hb, &np, hdr, &value);
        if (!lrc && np != p) lrc = fail(SELHOST_E_FORMAT, "'%s.hll': precision %u, expected %u", fn.c_str(), np, p);
        // build_sketch writes (is_calc 0, ERTL_MLE, J_ERTL_MLE) with value -1; anything else would send the
        // reference down a different estimator (hll.h:210-263) that this path does not implement
        if (!lrc && (hdr[1] != 2 || hdr[2] != 2)) lrc = fail(SELHOST_E_FORMAT, "'%s.hll': estimator %u/%u is not ERTL_MLE", fn.c_str(), hdr[1], hdr[2]);
        if (!lrc) cards[(size_t)i] = value >= 0. ? value                                            // hll.h:1114 is_calculated
                                                 : selhost_hll_report(hll.data() + (size_t)i * hb, p, fp_mode);
        if (!lrc && m) {
            int64_t cnt = selhost_read_smh((fn + ".smh" + std::to_string(m)).c_str(), aux.data() + (size_t)i * m, m);
            if (cnt < 0) lrc = (int)cnt;
            else if ((uint64_t)cnt != m) lrc = fail(SELHOST_E_FORMAT, "'%s.smh%u' holds %lld buckets", fn.c_str(), m, (long long)cnt);
        }
        if (!lrc && p_aux) {
            lrc = selhost_read_hll((fn + ".hll_" + std::to_string(p_aux)).c_str(), auxh.data() + (size_t)i * ab, ab, &np, hdr, &value);
            if (!lrc && np != p_aux) lrc = fail(SELHOST_E_FORMAT, "'%s.hll_%u': precision %u", fn.c_str(), p_aux, np);
        }
        if (lrc) {
#pragma omp critical
            { if (!err) { err = lrc; err_msg = g_err; } }
        }
    }
    if (err) { g_err = err_msg; return err; }

    std::vector<int32_t> perm((size_t)n);
    selhost_sort_by_card(cards.data(), n, perm.data());
    auto ds = std::make_unique<selhost_dataset>();
    ds->n = n; ds->m = m; ds->p_aux = p_aux;
    ds->names.resize((size_t)n);
    ds->hll.resize((size_t)n * hb); ds->aux.resize((size_t)n * m); ds->aux_hll.resize((size_t)n * ab); ds->cards.resize((size_t)n);
    for (int64_t r = 0; r < n; ++r) {                                  // selection_cuda.cpp:138-143
        const size_t s = (size_t)perm[(size_t)r];
        ds->names[(size_t)r] = files[s];
        std::memcpy(ds->hll.data() + (size_t)r * hb, hll.data() + s * hb, hb);
        if (m) std::memcpy(ds->aux.data() + (size_t)r * m, aux.data() + s * m, (size_t)m * 8);
        if (ab) std::memcpy(ds->aux_hll.data() + (size_t)r * ab, auxh.data() + s * ab, ab);
        ds->cards[(size_t)r] = cards[s];
    }
    *out = ds.release();
    return SELHOST_OK;
}

void selhost_dataset_free(selhost_dataset* ds) { delete ds; }
int64_t selhost_dataset_size(const selhost_dataset* ds) { return ds ? ds->n : 0; }
const uint8_t* selhost_dataset_hll(const selhost_dataset* ds) { return ds ? ds->hll.data() : nullptr; }
const uint64_t* selhost_dataset_aux(const selhost_dataset* ds) { return ds ? ds->aux.data() : nullptr; }
const uint8_t* selhost_dataset_aux_hll(const selhost_dataset* ds) { return ds ? ds->aux_hll.data() : nullptr; }
const double* selhost_dataset_cards(const selhost_dataset* ds) { return ds ? ds->cards.data() : nullptr; }
const char* selhost_dataset_name(const selhost_dataset* ds, int64_t rank) {
    if (!ds || rank < 0 || rank >= ds->n) return "";
    return ds->names[(size_t)rank].c_str();
}

int selhost_format_line(const char* fn1, const char* fn2, double jaccard, char* buf, size_t cap) {
    if (!fn1 || !fn2 || !buf) return SELHOST_E_BADARG;
    int w = snprintf(buf, cap, "%s %s %f\n", fn1, fn2, jaccard);       // selection.cpp:288 std::to_string(double)
    return (w < 0 || (size_t)w >= cap) ? SELHOST_E_BADARG : w;
}

// ---- on-disk result format ---------------------------------------------------------------------------------
struct selhost_results {
    std::vector<selhost_pair_t> pairs;
    std::vector<std::string> names;
    float tau = 0.f;
};

namespace {
struct ResultHeader { char magic[4]; uint32_t version; uint64_t n_pairs, n_names, names_bytes; float tau; uint32_t reserved; };
static_assert(sizeof(ResultHeader) == 40, "result header layout");
static_assert(sizeof(selhost_pair_t) == 16, "result record layout");
}  // namespace

int selhost_write_results(const char* path, const selhost_pair_t* pairs, int64_t n_pairs,
                          const char* const* names, int64_t n_names, float tau) {
    if (!path || n_pairs < 0 || n_names < 0 || (n_pairs && !pairs) || (n_names && !names)) return fail(SELHOST_E_BADARG, "bad argument");
    for (int64_t j = 0; j < n_pairs; ++j)
        if (pairs[j].i < 0 || pairs[j].k < 0 || (n_names && (pairs[j].i >= n_names || pairs[j].k >= n_names)))
            return fail(SELHOST_E_BADARG, "record %lld refers to a rank outside the name table", (long long)j);
    std::string blob;
    for (int64_t g = 0; g < n_names; ++g) {
        if (!names[g]) return fail(SELHOST_E_BADARG, "null name");
        blob.append(names[g]);
        blob.push_back('\0');
    }
    ResultHeader h;
    std::memcpy(h.magic, "SELR", 4);
    h.version = 1; h.n_pairs = (uint64_t)n_pairs; h.n_names = (uint64_t)n_names; h.names_bytes = blob.size(); h.tau = tau; h.reserved = 0;
    FILE* fp = std::fopen(path, "wb");
    if (!fp) return fail(SELHOST_E_IO, "cannot open %s for writing", path);
    bool ok = std::fwrite(&h, sizeof h, 1, fp) == 1;
    ok = ok && (blob.empty() || std::fwrite(blob.data(), 1, blob.size(), fp) == blob.size());
    ok = ok && (n_pairs == 0 || std::fwrite(pairs, sizeof(selhost_pair_t), (size_t)n_pairs, fp) == (size_t)n_pairs);
    ok = (std::fclose(fp) == 0) && ok;
    return ok ? SELHOST_OK : fail(SELHOST_E_IO, "short write to %s", path);
}

int selhost_read_results(selhost_results** out, const char* path) {
    if (!out || !path) return fail(SELHOST_E_BADARG, "bad argument");
    *out = nullptr;
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return fail(SELHOST_E_IO, "cannot open %s", path);
    std::unique_ptr<FILE, int (*)(FILE*)> guard(fp, std::fclose);
    ResultHeader h;
    if (std::fread(&h, sizeof h, 1, fp) != 1) return fail(SELHOST_E_IO, "short read (header) from %s", path);
    if (std::memcmp(h.magic, "SELR", 4) || h.version != 1) return fail(SELHOST_E_FORMAT, "%s is not a version-1 result file", path);
    if (h.n_pairs > (1ull << 40) || h.n_names > (1ull << 32) || h.names_bytes > (1ull << 40) || h.names_bytes < h.n_names)
        return fail(SELHOST_E_FORMAT, "implausible sizes in %s", path);
    std::unique_ptr<selhost_results> r(new selhost_results);
    r->tau = h.tau;
    std::string blob((size_t)h.names_bytes, '\0');
    if (h.names_bytes && std::fread(&blob[0], 1, blob.size(), fp) != blob.size()) return fail(SELHOST_E_IO, "short read (names) from %s", path);
    size_t pos = 0;
    for (uint64_t g = 0; g < h.n_names; ++g) {
        const void* z = pos < blob.size() ? std::memchr(blob.data() + pos, 0, blob.size() - pos) : nullptr;
        if (!z) return fail(SELHOST_E_FORMAT, "name table of %s is truncated", path);
        const size_t len = (size_t)((const char*)z - (blob.data() + pos));
        r->names.emplace_back(blob.data() + pos, len);
        pos += len + 1;
    }
    r->pairs.resize((size_t)h.n_pairs);
    if (h.n_pairs && std::fread(r->pairs.data(), sizeof(selhost_pair_t), (size_t)h.n_pairs, fp) != (size_t)h.n_pairs)
        return fail(SELHOST_E_IO, "short read (records) from %s", path);
    for (const selhost_pair_t& pr : r->pairs)
        if (pr.i < 0 || pr.k < 0 || (h.n_names && ((uint64_t)pr.i >= h.n_names || (uint64_t)pr.k >= h.n_names)))
            return fail(SELHOST_E_FORMAT, "a record of %s refers to a rank outside the name table", path);
    *out = r.release();
    return SELHOST_OK;
}

void selhost_results_free(selhost_results* r) { delete r; }
int64_t selhost_results_count(const selhost_results* r) { return r ? (int64_t)r->pairs.size() : SELHOST_E_BADARG; }
int64_t selhost_results_names(const selhost_results* r) { return r ? (int64_t)r->names.size() : SELHOST_E_BADARG; }
float selhost_results_tau(const selhost_results* r) { return r ? r->tau : 0.f; }
const selhost_pair_t* selhost_results_pairs(const selhost_results* r) { return r && !r->pairs.empty() ? r->pairs.data() : nullptr; }
const char* selhost_results_name(const selhost_results* r, int64_t rank) {
    if (!r || rank < 0 || rank >= (int64_t)r->names.size()) return nullptr;
    return r->names[(size_t)rank].c_str();
}
int64_t selhost_results_text(const selhost_results* r, char* buf, size_t cap) {
    if (!r) return SELHOST_E_BADARG;
    if (r->names.empty() && !r->pairs.empty()) return fail(SELHOST_E_FORMAT, "the file carries no name table");
    std::string out;
    char line[8192];
    for (const selhost_pair_t& pr : r->pairs) {
        const int w = selhost_format_line(r->names[(size_t)pr.i].c_str(), r->names[(size_t)pr.k].c_str(), pr.jaccard, line, sizeof line);
        if (w < 0) return fail(SELHOST_E_BADARG, "line too long");
        out.append(line, (size_t)w);
    }
    if (buf && cap) {
        const size_t c = std::min(cap - 1, out.size());
        std::memcpy(buf, out.data(), c);
        buf[c] = '\0';
    }
    return (int64_t)out.size();
}

// ---- synthetic sketches -------------------------------------------------------------------------------
int selhost_synth_generate(const selhost_synth_t* in, int64_t g_begin, int64_t g_end,
                           uint8_t* hll, uint64_t* aux, uint8_t* aux_hll, int n_threads) {
    if (!in || !hll || !aux || g_begin < 0 || g_end < g_begin) return fail(SELHOST_E_BADARG, "bad argument");
    if (in->m <= 0 || (in->m & (in->m - 1)) || in->cluster_size < 1 || in->p_aux < 0 || in->p_aux > 12)
        return fail(SELHOST_E_BADARG, "bad synth parameters");
    selhip::SynthParams sp;
    sp.seed = in->seed; sp.n_genomes = in->n_genomes; sp.m = in->m; sp.p_aux = in->p_aux;
    sp.cluster_size = in->cluster_size; sp.mode = in->mode; sp.n_sh_lo = in->n_sh_lo; sp.n_sh_hi = in->n_sh_hi;
    const int n_aux = sp.p_aux ? 1 << sp.p_aux : 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic) num_threads(n_threads)
    for (int64_t g = g_begin; g < g_end; ++g) {
        const int64_t r = g - g_begin;
        uint8_t* regs = hll + (size_t)r * 16384;
        uint64_t* smh = aux + (size_t)r * sp.m;
        uint8_t* aregs = (n_aux && aux_hll) ? aux_hll + (size_t)r * n_aux : nullptr;
        std::memset(regs, 0, 16384);
        for (int t = 0; t < sp.m; ++t) smh[t] = ~0ull;
        if (aregs) std::memset(aregs, 0, (size_t)n_aux);
        const uint32_t cluster = (uint32_t)(g / sp.cluster_size);
        const uint32_t n_sh = selhip::synth_n_shared(sp, cluster);
        const uint32_t n_pr = selhip::synth_n_private(sp, (uint32_t)g, n_sh);
        for (int part = 0; part < 2; ++part) {
            const uint32_t cnt = part == 0 ? n_sh : n_pr;
            const uint64_t stream = part == 0 ? 2ull * cluster : 2ull * (uint64_t)g + 1;
            for (uint32_t e = 0; e < cnt; ++e) {
                const uint64_t h = selhip::synth_element(sp, stream, e);
                uint32_t idx, rank;
                selhip::synth_hll_slot(h, 14, &idx, &rank);
                if (regs[idx] < rank) regs[idx] = (uint8_t)rank;
                if (aregs) {
                    selhip::synth_hll_slot(h, sp.p_aux, &idx, &rank);
                    if (aregs[idx] < rank) aregs[idx] = (uint8_t)rank;
                }
                uint32_t bucket; uint64_t value;
                selhip::synth_smh_slot(h, sp.m, &bucket, &value);
                if (value < smh[bucket]) smh[bucket] = value;
            }
        }
    }
    return SELHOST_OK;
}

// ---- sharding -----------------------------------------------------------------------------------------
int selhost_shard_rows(int64_t n, const int32_t* hi, int64_t z0, int parts, int64_t* bounds) {
    if (n < 0 || parts < 1 || !bounds) return fail(SELHOST_E_BADARG, "bad argument");
    std::vector<double> prefix((size_t)n + 1, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        int64_t h = hi ? hi[i] : n - 1;
        int64_t first = std::max<int64_t>(i + 1, z0);
        int64_t cnt = h - first + 1;
        prefix[(size_t)i + 1] = prefix[(size_t)i] + (double)(cnt > 0 ? cnt : 0);
    }
    const double total = prefix[(size_t)n];
    bounds[0] = 0;
    for (int part = 1; part < parts; ++part) {
        const double target = total * (double)part / (double)parts;
        int64_t b = (int64_t)(std::lower_bound(prefix.begin(), prefix.end(), target) - prefix.begin());
        if (b > n) b = n;
        if (b < bounds[part - 1]) b = bounds[part - 1];
        bounds[part] = b;
    }
    bounds[parts] = n;
    return SELHOST_OK;
}

}  // extern "C"
