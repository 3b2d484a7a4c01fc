// selhip_ooc.hip -- selhip_ooc_select: the out-of-core driver.  Host code only (no kernels), built on the
// public context API: the sketches stay in HOST memory and block pairs (I, J), I <= J, are uploaded in turn; a diagonal tile is an
// ordinary pass, an off-diagonal one runs rows I x candidates J (selhip_ctx_set_candidate_begin).  See include/selection_hip.h 2c.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "selhip_internal.h"

namespace {
struct OocWorker {
    selhip_ctx* ctx = nullptr;
    hipStream_t st = nullptr;                    // the worker's own stream (set on its context)
    uint8_t* d_hll = nullptr; uint64_t* d_aux = nullptr; double* d_cards = nullptr; uint8_t* d_aux_hll = nullptr;
    int64_t resident_i = -1;                     // block whose rows sit at the front of the buffers
    std::vector<selhip_pair_t> out;
    int64_t stats[4] = {0, 0, 0, 0};
    int rc = 0;
    std::string err;
};
}  // namespace

extern "C" int selhip_ooc_select(int device, const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                      const uint8_t* h_aux_hll, int p_aux, int criterion,
                      int64_t n, int m, int p_hll, int mode, int algo, int fp_mode, float tau_f, int n_rows, int n_bands,
                      int64_t block_genomes, int n_streams,
                      selhip_pair_t* h_out, int64_t cap, int64_t* count_out, int64_t stats_out[4]) {
    if (!count_out || cap < 0 || (cap && !h_out) || n < 0 || block_genomes < 1 || n_streams < 1 || n_streams > 4) { selhip_internal_set_error("bad argument"); return SELHIP_E_BADARG; }
    if (n > 0 && (!h_hll || !h_aux || !h_cards)) { selhip_internal_set_error("the out-of-core driver needs host sketches and their cardinalities"); return SELHIP_E_BADARG; }
    const bool need_aux_hll = criterion != SELHIP_CRIT_SMH_A;
    if (need_aux_hll && (!h_aux_hll || p_aux < 4 || p_aux > SELHIP_MAX_AUX_P)) { selhip_internal_set_error("criterion %d needs auxiliary HLL sketches", criterion); return SELHIP_E_BADARG; }
    if (block_genomes * 2 > 0x7FFFFFF0ll) { selhip_internal_set_error("block too large"); return SELHIP_E_BADARG; }
    *count_out = 0;
    if (stats_out) for (int k = 0; k < 4; ++k) stats_out[k] = 0;
    if (n == 0) return SELHIP_OK;
    for (int64_t i = 1; i < n; ++i)
        if (h_cards[i] < h_cards[i - 1]) { selhip_internal_set_error("cards are not in ascending order"); return SELHIP_E_BADARG; }
    const int64_t B = std::min(block_genomes, n);
    const int64_t nb = (n + B - 1) / B;
    const size_t hb = (size_t)1 << p_hll, ab = need_aux_hll ? ((size_t)1 << p_aux) : 0;
    std::vector<std::pair<int64_t, int64_t>> tiles;              // row-major: the resident row block changes rarely
    for (int64_t I = 0; I < nb; ++I) for (int64_t J = I; J < nb; ++J) tiles.emplace_back(I, J);
    const int W = (int)std::min<int64_t>(n_streams, (int64_t)tiles.size());
    std::vector<OocWorker> ws((size_t)W);
    std::vector<std::thread> th;
    for (int w = 0; w < W; ++w) th.emplace_back([&, w] {
        OocWorker& k = ws[(size_t)w];
        auto fail = [&](int rc, const char* what) -> void { k.rc = rc; k.err = what ? what : ""; };
        if (hipSetDevice(device) != hipSuccess) return fail(SELHIP_E_HIP, "hipSetDevice");
        int r = selhip_ctx_create(&k.ctx, device);
        if (r) return fail(r, selhip_last_error(nullptr));
        if (hipStreamCreateWithFlags(&k.st, hipStreamNonBlocking) != hipSuccess) return fail(SELHIP_E_HIP, "hipStreamCreate");
        hipStream_t st = k.st;
        r = selhip_ctx_set_stream(k.ctx, st);
        if (!r) r = selhip_ctx_set_fp_mode(k.ctx, fp_mode);
        if (r) return fail(r, selhip_last_error(k.ctx));
        if (hipMalloc((void**)&k.d_hll, (size_t)2 * B * hb) != hipSuccess || hipMalloc((void**)&k.d_aux, (size_t)2 * B * m * 8) != hipSuccess ||
            hipMalloc((void**)&k.d_cards, (size_t)2 * B * 8) != hipSuccess || (need_aux_hll && hipMalloc((void**)&k.d_aux_hll, (size_t)2 * B * ab) != hipSuccess))
            return fail(SELHIP_E_HIP, "hipMalloc of the out-of-core driver's device buffers failed");
        auto put = [&](int64_t blk, int64_t at) -> bool {         // block `blk` of the host arrays -> device rows [at, at + len)
            const int64_t g0 = blk * B, len = std::min(B, n - g0);
            bool ok = hipMemcpyAsync(k.d_hll + (size_t)at * hb, h_hll + (size_t)g0 * hb, (size_t)len * hb, hipMemcpyHostToDevice, st) == hipSuccess;
            ok = ok && hipMemcpyAsync(k.d_aux + (size_t)at * m, h_aux + (size_t)g0 * m, (size_t)len * m * 8, hipMemcpyHostToDevice, st) == hipSuccess;
            ok = ok && hipMemcpyAsync(k.d_cards + at, h_cards + g0, (size_t)len * 8, hipMemcpyHostToDevice, st) == hipSuccess;
            if (need_aux_hll) ok = ok && hipMemcpyAsync(k.d_aux_hll + (size_t)at * ab, h_aux_hll + (size_t)g0 * ab, (size_t)len * ab, hipMemcpyHostToDevice, st) == hipSuccess;
            return ok;
        };
        std::vector<selhip_pair_t> part;
        for (size_t t = (size_t)w; t < tiles.size(); t += (size_t)W) {
            const int64_t I = tiles[t].first, J = tiles[t].second;
            const int64_t i0 = I * B, nI = std::min(B, n - i0), j0 = J * B, nJ = std::min(B, n - j0);
            if (k.resident_i != I) { if (!put(I, 0)) return fail(SELHIP_E_HIP, "upload of a row block"); k.resident_i = I; }
            const bool diag = I == J;
            if (!diag && !put(J, nI)) return fail(SELHIP_E_HIP, "upload of a candidate block");
            const int64_t tot = diag ? nI : nI + nJ;
            r = selhip_ctx_attach(k.ctx, k.d_hll, (const uint64_t*)k.d_aux, k.d_cards, tot, m, p_hll);
            if (!r && need_aux_hll) r = selhip_ctx_attach_aux_hll(k.ctx, k.d_aux_hll, p_aux);
            if (!r) r = selhip_ctx_set_criterion(k.ctx, criterion);
            if (!r) r = selhip_ctx_set_candidate_begin(k.ctx, diag ? 0 : nI);
            if (!r) r = selhip_ctx_run(k.ctx, mode, algo, tau_f, n_rows, n_bands, 0, nI);
            if (r) return fail(r, selhip_last_error(k.ctx));
            const int64_t cnt = selhip_ctx_result_count(k.ctx);
            part.resize((size_t)cnt);
            int64_t s4[4];
            r = selhip_ctx_fetch(k.ctx, cnt ? part.data() : nullptr, cnt);
            if (!r) r = selhip_ctx_stats(k.ctx, s4);
            if (r) return fail(r, selhip_last_error(k.ctx));
            for (int q = 0; q < 4; ++q) k.stats[q] += s4[q];
            for (auto& pr : part) {                               // local ranks -> global ranks
                pr.i = (int32_t)(i0 + pr.i);
                pr.k = (int32_t)(diag ? i0 + pr.k : j0 + (pr.k - nI));
            }
            k.out.insert(k.out.end(), part.begin(), part.end());
        }
        (void)hipStreamSynchronize(st);
    });
    for (auto& t : th) t.join();
    int fail_rc = 0;
    for (int w = 0; w < W; ++w) {
        OocWorker& k = ws[(size_t)w];
        if (k.rc && !fail_rc) { fail_rc = k.rc; selhip_internal_set_error("out-of-core worker %d: %s", w, k.err.c_str()); }
        (void)hipSetDevice(device);
        if (k.ctx) selhip_ctx_destroy(k.ctx);
        if (k.st) (void)hipStreamDestroy(k.st);
        if (k.d_hll) (void)hipFree(k.d_hll);
        if (k.d_aux) (void)hipFree(k.d_aux);
        if (k.d_cards) (void)hipFree(k.d_cards);
        if (k.d_aux_hll) (void)hipFree(k.d_aux_hll);
    }
    if (fail_rc) return fail_rc;
    std::vector<selhip_pair_t> all;
    for (auto& k : ws) { all.insert(all.end(), k.out.begin(), k.out.end()); if (stats_out) for (int q = 0; q < 4; ++q) stats_out[q] += k.stats[q]; }
    std::sort(all.begin(), all.end(), [](const selhip_pair_t& a, const selhip_pair_t& b) { return a.i != b.i ? a.i < b.i : a.k < b.k; });
    *count_out = (int64_t)all.size();
    if (cap) std::memcpy(h_out, all.data(), (size_t)std::min<int64_t>((int64_t)all.size(), cap) * sizeof(selhip_pair_t));
    if ((int64_t)all.size() > cap) { selhip_internal_set_error("result buffer too small: %lld records", (long long)all.size()); return SELHIP_E_OVERFLOW; }
    return SELHIP_OK;
}

