// selhip_multi.hip -- selhip_multi_select: the multi-GPU entry that takes a device list.
// Host code only (no kernels): ONE process, one host thread + one context per device, built on the public context API.
//   * rows of the pair matrix are dealt to the devices in interleaved blocks of 128 (selhip_ctx_set_row_interleave): every device
//     gets the same share of pairs and of survivors;
//   * every device needs a full replica of the sketches.  With RCCL each device uploads ONE G-th of every array from the host and
//     the replicas are completed by in-place ncclAllGather over xGMI (round 2 had every device thread push the whole set through
//     its own pageable-memory copy: G x 2.4 GB over PCIe at BASELINE configs[4]); without RCCL (host gather) each device uploads all;
//   * the only exchange of the pass itself is the gather of the selected-pair records: one ncclAllGather of framed buffers, or a
//     host merge;
//   * a device thread that fails at any point ABORTS every communicator (ncclCommAbort), so that no other thread stays blocked in a
//     collective the failed rank will never enter (round 2 could strand G-1 threads in ncclAllGather for ever).
// librccl is dlopen'ed on first use so that single-GPU users never load it.
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "selhip_internal.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    int (*CommInitAll)(void**, int, const int*) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*CommAbort)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (int (*)(void**, int, const int*))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
        CommAbort = (int (*)(void*))dlsym(lib, "ncclCommAbort");
        AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
        GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && CommAbort && AllGather && GetErrorString;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;
constexpr int kNcclChar = 0;     // ncclDataType_t ncclChar (rccl.h)

// test hook (tests/test_gpu_parity.py): SELHIP_TEST_FAIL="<rank>:<stage>" makes that device thread fail at "upload" (before the
// replica all-gather), "run" or "gather" (before the gather of the result records)
bool test_fail(int rank, const char* stage) {
    const char* e = std::getenv("SELHIP_TEST_FAIL");
    if (!e) return false;
    const char* colon = std::strchr(e, ':');
    return colon && std::atoi(e) == rank && !std::strcmp(colon + 1, stage);
}

}  // namespace

extern "C" int selhip_multi_select(const int* devices, int n_devices,
                                   const uint8_t* h_hll, const uint64_t* h_aux, const double* h_cards,
                                   const uint8_t* h_aux_hll, int p_aux, int criterion,
                                   int64_t n, int m, int p_hll, int mode, int algo, int fp_mode, float tau_f, int n_rows, int n_bands,
                                   int gather, selhip_pair_t* h_out, int64_t cap, int64_t* count_out, int64_t stats_out[4]) {
    if (!devices || n_devices < 1 || n_devices > 64 || !count_out || cap < 0 || (cap && !h_out)) { selhip_internal_set_error("bad argument"); return SELHIP_E_BADARG; }
    if (criterion < SELHIP_CRIT_SMH_A || criterion > SELHIP_CRIT_HLL_A_SMH_A) { selhip_internal_set_error("bad criterion %d", criterion); return SELHIP_E_BADARG; }
    const bool need_aux = criterion != SELHIP_CRIT_SMH_A && n > 0;
    if (need_aux && (!h_aux_hll || p_aux < 4 || p_aux > SELHIP_MAX_AUX_P)) { selhip_internal_set_error("criterion %d needs auxiliary HLL sketches (h_aux_hll, p_aux in [4,15])", criterion); return SELHIP_E_BADARG; }
    if (gather < SELHIP_GATHER_HOST || gather > SELHIP_GATHER_RCCL_OR_HOST) { selhip_internal_set_error("bad gather mode"); return SELHIP_E_BADARG; }
    if (n < 0 || m <= 0 || p_hll < 4 || p_hll > 20 || (n > 0 && (!h_hll || !h_aux || !h_cards))) { selhip_internal_set_error("bad sketch arguments"); return SELHIP_E_BADARG; }
    *count_out = 0;
    const int G = n_devices;
    // the cardinalities are checked once, here (the sharded upload never shows a context the host array)
    for (int64_t i = 0; i < n; ++i) {
        const double v = h_cards[i];
        if (!(v >= 0.0) || !(v < 9.2e18)) { selhip_internal_set_error("cards[%lld] = %g is not a finite value in [0, 2^63)", (long long)i, v); return SELHIP_E_BADARG; }
        if (i && v < h_cards[i - 1]) { selhip_internal_set_error("cards are not in ascending order at rank %lld", (long long)i); return SELHIP_E_BADARG; }
    }

    // RCCL communicators (single process, one per device)
    std::vector<void*> comms((size_t)G, nullptr);
    bool use_rccl = gather != SELHIP_GATHER_HOST;
    std::string rccl_note;
    if (use_rccl) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!g_rccl.load()) { use_rccl = false; rccl_note = "librccl.so not loadable"; }
        else {
            const int r = g_rccl.CommInitAll(comms.data(), G, devices);
            if (r != 0) { use_rccl = false; rccl_note = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); std::fill(comms.begin(), comms.end(), nullptr); }
        }
        if (!use_rccl && gather == SELHIP_GATHER_RCCL) { selhip_internal_set_error("RCCL gather requested but unavailable: %s", rccl_note.c_str()); return SELHIP_E_HIP; }
    }
    // a failing device thread aborts EVERY communicator: the threads blocked in (or about to enter) a collective return instead of
    // waiting for a rank that will never come
    std::atomic<int> failed{0};
    std::mutex abort_mu;
    bool aborted = false;
    auto abort_all = [&] {
        failed.store(1);
        std::lock_guard<std::mutex> lk(abort_mu);
        if (aborted || !use_rccl) return;
        aborted = true;
        for (int k = 0; k < G; ++k)
            if (comms[(size_t)k]) { (void)g_rccl.CommAbort(comms[(size_t)k]); comms[(size_t)k] = nullptr; }
    };

    auto comm_of = [&](int g) -> void* { std::lock_guard<std::mutex> lk(abort_mu); return comms[(size_t)g]; };     // NULL once aborted

    std::vector<selhip_ctx*> ctxs((size_t)G, nullptr);
    std::vector<int> rc((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    std::vector<int64_t> counts((size_t)G, 0);
    std::vector<std::array<int64_t, 4>> st((size_t)G);
    const size_t hb = (size_t)1 << p_hll, ab = need_aux ? ((size_t)1 << p_aux) : 0;
    const int64_t slice = (n + G - 1) / G, n_pad = slice * G;           // rows per device of the sharded upload (the last one may be short)

    // ---- phase 1: replicas, then every device runs its interleaved share of the rows
    {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g) th.emplace_back([&, g] {
            auto fail = [&](int code, const std::string& what) { rc[(size_t)g] = code; errs[(size_t)g] = what; abort_all(); };
            int r = selhip_ctx_create(&ctxs[(size_t)g], devices[g]);
            if (r) return fail(r, selhip_last_error(nullptr));
            selhip_ctx* c = ctxs[(size_t)g];
            r = selhip_ctx_set_fp_mode(c, fp_mode);
            if (r) return fail(r, selhip_last_error(c));
            if (test_fail(g, "upload")) return fail(SELHIP_E_HIP, "test hook: failure before the replica all-gather");
            if (use_rccl && n > 0) {
                uint8_t* d_hll = nullptr; uint64_t* d_aux = nullptr; double* d_cards = nullptr; uint8_t* d_ah = nullptr;
                r = selhip_internal_reserve_replica(c, n_pad, m, p_hll, need_aux ? p_aux : 0, &d_hll, &d_aux, &d_cards, &d_ah);
                if (r) return fail(r, selhip_last_error(c));
                hipStream_t s = (hipStream_t)selhip_internal_stream(c);
                const int64_t r0 = (int64_t)g * slice, r1 = std::min(n, r0 + slice), rows = std::max<int64_t>(0, r1 - r0);
                hipError_t e = hipSuccess;
                if (rows > 0) {
                    e = hipMemcpyAsync(d_hll + (size_t)r0 * hb, h_hll + (size_t)r0 * hb, (size_t)rows * hb, hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_aux + (size_t)r0 * m, h_aux + (size_t)r0 * m, (size_t)rows * m * 8, hipMemcpyHostToDevice, s);
                    if (e == hipSuccess) e = hipMemcpyAsync(d_cards + r0, h_cards + r0, (size_t)rows * 8, hipMemcpyHostToDevice, s);
                    if (e == hipSuccess && need_aux) e = hipMemcpyAsync(d_ah + (size_t)r0 * ab, h_aux_hll + (size_t)r0 * ab, (size_t)rows * ab, hipMemcpyHostToDevice, s);
                }
                if (e != hipSuccess) return fail(SELHIP_E_HIP, std::string("upload of a replica slice: ") + hipGetErrorString(e));
                // in-place all-gathers: this device's slice sits at recvbuf + rank * bytes already
                struct { void* base; size_t row; } arr[4] = {{d_hll, hb}, {d_aux, (size_t)m * 8}, {d_cards, 8}, {d_ah, ab}};
                for (int a = 0; a < (need_aux ? 4 : 3); ++a) {
                    void* const comm = comm_of(g);
                    if (failed.load() || !comm) return;
                    const size_t bytes = (size_t)slice * arr[a].row;
                    const int nr = g_rccl.AllGather((char*)arr[a].base + (size_t)g * bytes, arr[a].base, bytes, kNcclChar, comm, s);
                    if (nr != 0) return fail(SELHIP_E_HIP, std::string("ncclAllGather of the replicas: ") + g_rccl.GetErrorString(nr));
                }
                e = hipStreamSynchronize(s);
                if (failed.load()) return;
                if (e != hipSuccess) return fail(SELHIP_E_HIP, std::string("replica all-gather: ") + hipGetErrorString(e));
                r = selhip_ctx_attach(c, d_hll, d_aux, d_cards, n, m, p_hll);
                if (!r && need_aux) r = selhip_ctx_attach_aux_hll(c, d_ah, p_aux);
            } else {
                r = selhip_ctx_upload(c, h_hll, h_aux, h_cards, n, m, p_hll);
                if (!r && need_aux) r = selhip_ctx_upload_aux_hll(c, h_aux_hll, p_aux);
            }
            if (!r) r = selhip_ctx_set_criterion(c, criterion);
            // interleaved row blocks: every device gets the same share of pairs and of survivors
            if (!r) r = selhip_ctx_set_row_interleave(c, 128, G, g);
            if (!r && test_fail(g, "run")) return fail(SELHIP_E_HIP, "test hook: failure before the pass");
            if (!r) r = selhip_ctx_run(c, mode, algo, tau_f, n_rows, n_bands, 0, n);
            if (!r) { counts[(size_t)g] = selhip_ctx_result_count(c); r = selhip_ctx_stats(c, st[(size_t)g].data()); }
            if (r) return fail(r, selhip_last_error(c));
        });
        for (auto& t : th) t.join();
    }
    int fail_rc = 0;
    std::string fail_msg;
    for (int g = 0; g < G; ++g)
        if (rc[(size_t)g] && !fail_rc) { fail_rc = rc[(size_t)g]; fail_msg = "device " + std::to_string(devices[g]) + ": " + errs[(size_t)g]; }
    if (!fail_rc && failed.load()) { fail_rc = SELHIP_E_HIP; fail_msg = "a device thread failed"; }
    std::vector<selhip_pair_t> all;
    if (!fail_rc) {
        int64_t max_cnt = 0;
        for (int g = 0; g < G; ++g) max_cnt = std::max(max_cnt, counts[(size_t)g]);
        if (use_rccl) {
            // ---- phase 2: framed send buffers of (max_cnt + 1) records, one all_gather, device 0's copy goes to the host
            const size_t frame = (size_t)(max_cnt + 1) * sizeof(selhip_pair_t);
            std::vector<void*> send((size_t)G, nullptr), recv((size_t)G, nullptr);
            std::vector<std::thread> th;
            for (int g = 0; g < G; ++g) th.emplace_back([&, g] {
                auto fail = [&](int code, const std::string& what) { rc[(size_t)g] = code; errs[(size_t)g] = what; abort_all(); };
                selhip_ctx* c = ctxs[(size_t)g];
                hipStream_t s = (hipStream_t)selhip_internal_stream(c);
                hipError_t e = hipSetDevice(devices[g]);
                if (e == hipSuccess) e = hipMalloc(&send[(size_t)g], frame);
                if (e == hipSuccess) e = hipMalloc(&recv[(size_t)g], frame * (size_t)G);
                if (e == hipSuccess) e = hipMemsetAsync(send[(size_t)g], 0, frame, s);
                if (e != hipSuccess) return fail(SELHIP_E_HIP, hipGetErrorString(e));
                if (test_fail(g, "gather")) return fail(SELHIP_E_HIP, "test hook: failure before the gather of the records");
                const int r = selhip_ctx_copy_results_framed(c, send[(size_t)g], max_cnt);
                if (r) return fail(r, selhip_last_error(c));
                void* const comm = comm_of(g);
                if (failed.load() || !comm) return;
                const int nr = g_rccl.AllGather(send[(size_t)g], recv[(size_t)g], frame, kNcclChar, comm, s);
                if (nr != 0) return fail(SELHIP_E_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(nr));
                e = hipStreamSynchronize(s);
                if (failed.load()) return;
                if (e != hipSuccess) return fail(SELHIP_E_HIP, hipGetErrorString(e));
            });
            for (auto& t : th) t.join();
            for (int g = 0; g < G; ++g)
                if (rc[(size_t)g] && !fail_rc) { fail_rc = rc[(size_t)g]; fail_msg = "gather on device " + std::to_string(devices[g]) + ": " + errs[(size_t)g]; }
            if (!fail_rc && failed.load()) { fail_rc = SELHIP_E_HIP; fail_msg = "a device thread failed in the gather"; }
            if (!fail_rc) {
                std::vector<char> host_recv(frame * (size_t)G);
                (void)hipSetDevice(devices[0]);
                if (hipMemcpy(host_recv.data(), recv[0], host_recv.size(), hipMemcpyDeviceToHost) != hipSuccess) { fail_rc = SELHIP_E_HIP; fail_msg = "copy of the gathered records failed"; }
                for (int g = 0; g < G && !fail_rc; ++g) {
                    const char* f = host_recv.data() + (size_t)g * frame;
                    uint64_t cnt;
                    std::memcpy(&cnt, f, 8);
                    if ((int64_t)cnt != counts[(size_t)g]) { fail_rc = SELHIP_E_HIP; fail_msg = "gathered count mismatch on rank " + std::to_string(g); break; }
                    const selhip_pair_t* rec = reinterpret_cast<const selhip_pair_t*>(f + sizeof(selhip_pair_t));
                    all.insert(all.end(), rec, rec + cnt);
                }
            }
            for (int g = 0; g < G; ++g) { (void)hipSetDevice(devices[g]); if (send[(size_t)g]) (void)hipFree(send[(size_t)g]); if (recv[(size_t)g]) (void)hipFree(recv[(size_t)g]); }
        } else {
            for (int g = 0; g < G && !fail_rc; ++g) {
                std::vector<selhip_pair_t> part((size_t)counts[(size_t)g]);
                if (test_fail(g, "gather")) { fail_rc = SELHIP_E_HIP; fail_msg = "test hook: failure before the gather of the records"; break; }
                const int r = selhip_ctx_fetch(ctxs[(size_t)g], part.data(), counts[(size_t)g]);
                if (r) { fail_rc = r; fail_msg = "fetch on device " + std::to_string(devices[g]) + ": " + selhip_last_error(ctxs[(size_t)g]); }
                all.insert(all.end(), part.begin(), part.end());
            }
        }
        if (!fail_rc) {
            std::sort(all.begin(), all.end(), [](const selhip_pair_t& a, const selhip_pair_t& b) { return a.i != b.i ? a.i < b.i : a.k < b.k; });
            *count_out = (int64_t)all.size();
            if (cap) std::memcpy(h_out, all.data(), (size_t)std::min<int64_t>((int64_t)all.size(), cap) * sizeof(selhip_pair_t));
            if (stats_out) {
                for (int k = 0; k < 4; ++k) { stats_out[k] = 0; for (int g = 0; g < G; ++g) stats_out[k] += st[(size_t)g][(size_t)k]; }
            }
            if ((int64_t)all.size() > cap) { fail_rc = SELHIP_E_OVERFLOW; fail_msg = "result buffer too small: " + std::to_string(all.size()) + " records"; }
        }
    }
    for (int g = 0; g < G; ++g) if (ctxs[(size_t)g]) selhip_ctx_destroy(ctxs[(size_t)g]);
    if (use_rccl) {
        std::lock_guard<std::mutex> lk(abort_mu);
        for (int g = 0; g < G; ++g) if (comms[(size_t)g]) (void)g_rccl.CommDestroy(comms[(size_t)g]);          // (aborted communicators are gone already)
    }
    if (fail_rc) selhip_internal_set_error("%s", fail_msg.c_str());
    else if (!rccl_note.empty()) selhip_internal_set_error("note: host gather used (%s)", rccl_note.c_str());
    return fail_rc;
}
