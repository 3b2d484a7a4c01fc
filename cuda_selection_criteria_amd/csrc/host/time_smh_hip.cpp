// time_smh_hip.cpp -- timing harness, the MI355X counterpart of experiments/src/time_smh_cuda.cpp:141-311.
// Prints the same `list;label;tau;seconds` records (parsed by run_time_experiment.sh:24-26,37-39) for
// the two timed regions "smh_a" (all pairs) and "CB+smh_a", but (a) the device is synchronised before
// the clock stops (the reference stops it right after an asynchronous launch, time_smh_cuda.cpp:279-283),
// and (b) the sketches come from one of three places:
//   -l list          like the reference (time_smh_cuda.cpp:181-211): every entry's <file>.hll is read from disk and its
//                    SuperMinHash is REBUILT from the FASTA <file> -- on the GPU (selhip_build_sketches); that is the
//                    timed "build_smh" record;
//   -l list -D       .hll and .smh<m> both loaded from disk (no FASTA needed);
//   -N n_genomes     synthesised on the GPU.
//   -l list [-D] | -N n_genomes   -h tau   -m buckets   -b block(ignored)   -R repetitions   -S seed   -A algo
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/selection_hip.h"
#include "../../../include/selection_host.h"

static double now_s() {
    return std::chrono::duration<double>(std::chrono::high_resolution_clock::now().time_since_epoch()).count();
}

static void load_file_list(std::vector<std::string>& files, const std::string& list_file) {   // time_smh_cuda.cpp:101-123
    std::ifstream file(list_file);
    if (!file.is_open()) { std::cerr << "No valid input file provided\n"; exit(-1); }
    std::string line;
    while (getline(file, line)) {
        line.erase(0, line.find_first_not_of(" \t\r\n"));
        line.erase(line.find_last_not_of(" \t\r\n") + 1);
        if (!line.empty()) files.push_back(line);
    }
}

int main(int argc, char* argv[]) {
    std::string list_file = "";
    bool from_disk = false;
    float threshold = 0.9f;
    int mh_size = 8, total_rep = 1, algo = SELHIP_ALGO_AUTO;
    long n_synth = 0;
    unsigned long long seed = 0x5EED0000ull;
    int c;
    while ((c = getopt(argc, argv, "xDl:h:m:b:R:N:S:A:")) != -1) {
        switch (c) {
            case 'x': std::cout << "Usage: -l list [-D: .smh files from disk instead of rebuilding from FASTA] | -N genomes, -h tau -m buckets [-b block] [-R reps] [-S seed] [-A auto|stream|sig]\n"; return 0;
            case 'l': list_file = optarg; break;
            case 'D': from_disk = true; break;
            case 'h': threshold = std::stof(optarg); break;
            case 'm': mh_size = std::stoi(optarg); break;
            case 'b': break;
            case 'R': total_rep = std::stoi(optarg); break;
            case 'N': n_synth = std::stol(optarg); break;
            case 'S': seed = std::stoull(optarg, nullptr, 0); break;
            case 'A': algo = !strcmp(optarg, "stream") ? SELHIP_ALGO_STREAM : !strcmp(optarg, "sig") ? SELHIP_ALGO_SIG : SELHIP_ALGO_AUTO; break;
            default: break;
        }
    }
    if (selhip_device_count() <= 0) { std::cerr << "time_smh_hip: no MI355X (gfx950) device available\n"; return 3; }
    selhip_ctx* ctx = nullptr;
    if (selhip_ctx_create(&ctx, 0)) { std::cerr << selhip_last_error(nullptr) << "\n"; return 3; }
    std::string label = list_file;
    int64_t n = 0;
    selhost_dataset* ds = nullptr;
    void *d_hll = nullptr, *d_aux = nullptr, *d_hll_s = nullptr, *d_aux_s = nullptr, *d_perm = nullptr, *d_cards = nullptr, *d_cards_s = nullptr;

    double t0 = now_s();
    if (n_synth > 0) {
        n = n_synth;
        label = "synthetic_N" + std::to_string(n);
        selhip_synth_t sp{seed, (int32_t)n, mh_size, 0, 10, 0, 100000u, 100000u};
        if (selhip_malloc(&d_hll, (size_t)n * 16384) || selhip_malloc(&d_aux, (size_t)n * mh_size * 8) || selhip_malloc(&d_hll_s, (size_t)n * 16384) ||
            selhip_malloc(&d_aux_s, (size_t)n * mh_size * 8) || selhip_malloc(&d_perm, (size_t)n * 4) || selhip_malloc(&d_cards, (size_t)n * 8) ||
            selhip_malloc(&d_cards_s, (size_t)n * 8)) { std::cerr << "device allocation failed: " << selhip_last_error(nullptr) << "\n"; return 4; }
        int rc = selhip_synth_generate(&sp, 0, n, (uint8_t*)d_hll, (uint64_t*)d_aux, nullptr, nullptr);
        if (!rc) rc = selhip_hll_cards(ctx, (const uint8_t*)d_hll, n, 14, (double*)d_cards);
        std::vector<double> cards((size_t)n), cards_s((size_t)n);
        std::vector<int32_t> perm((size_t)n);
        if (!rc) rc = selhip_memcpy_d2h(cards.data(), d_cards, (size_t)n * 8);
        if (!rc) rc = selhost_sort_by_card(cards.data(), n, perm.data());
        for (int64_t r = 0; r < n; ++r) cards_s[(size_t)r] = cards[(size_t)perm[(size_t)r]];
        if (!rc) rc = selhip_memcpy_h2d(d_perm, perm.data(), (size_t)n * 4);
        if (!rc) rc = selhip_memcpy_h2d(d_cards_s, cards_s.data(), (size_t)n * 8);
        if (!rc) rc = selhip_permute_rows(d_hll, d_hll_s, (const int32_t*)d_perm, n, 16384, nullptr);
        if (!rc) rc = selhip_permute_rows(d_aux, d_aux_s, (const int32_t*)d_perm, n, (int64_t)mh_size * 8, nullptr);
        if (!rc) rc = selhip_ctx_attach(ctx, (const uint8_t*)d_hll_s, (const uint64_t*)d_aux_s, (const double*)d_cards_s, n, mh_size, 14);
        if (rc) { std::cerr << "setup failed: " << selhip_last_error(ctx) << "\n"; return 4; }
    } else if (!from_disk) {
        // time_smh_cuda.cpp:181-211: .hll from disk, SuperMinHash rebuilt from the FASTA -- here every k-mer of every genome on the GPU
        std::vector<std::string> files;
        load_file_list(files, list_file);
        n = (int64_t)files.size();
        const uint32_t m_vec = selhost_smh_vecsize((uint32_t)mh_size);             // SuperMinHash<>(mh_size) holds this many buckets (policy.h:12-19)
        mh_size = (int)m_vec;
        std::vector<uint8_t> hll((size_t)n * 16384);
        std::vector<double> cards((size_t)n);
        std::vector<int64_t> offsets{0};
        for (int64_t g = 0; g < n; ++g) {
            uint32_t p = 0, hdr[4]; double val = 0;
            if (selhost_read_hll((files[(size_t)g] + ".hll").c_str(), hll.data() + (size_t)g * 16384, 16384, &p, hdr, &val) || p != 14) {
                std::cerr << "Error opening file: " << files[(size_t)g] << ".hll (" << selhost_last_error() << ")\n"; return 4;
            }
            cards[(size_t)g] = selhost_hll_report(hll.data() + (size_t)g * 16384, 14, 1);
            const int64_t len = selhost_fasta_codes(files[(size_t)g].c_str(), nullptr, 0);
            if (len < 0) { std::cerr << "ERROR: Could not open the file " << files[(size_t)g] << " (" << selhost_last_error() << "); -D loads .smh files instead\n"; return 4; }
            offsets.push_back(offsets.back() + len);
        }
        std::vector<uint8_t> flat((size_t)std::max<int64_t>(offsets.back(), 1));
#pragma omp parallel for schedule(dynamic)
        for (int64_t g = 0; g < n; ++g)
            if (offsets[(size_t)g + 1] > offsets[(size_t)g]) selhost_fasta_codes(files[(size_t)g].c_str(), flat.data() + offsets[(size_t)g], (size_t)(offsets[(size_t)g + 1] - offsets[(size_t)g]));
        std::vector<int32_t> perm((size_t)n);
        std::vector<double> cards_s((size_t)n);
        std::vector<uint8_t> hll_s((size_t)n * 16384);
        int rc = selhost_sort_by_card(cards.data(), n, perm.data());                // time_smh_cuda.cpp:215
        for (int64_t r = 0; r < n && !rc; ++r) {
            cards_s[(size_t)r] = cards[(size_t)perm[(size_t)r]];
            std::memcpy(hll_s.data() + (size_t)r * 16384, hll.data() + (size_t)perm[(size_t)r] * 16384, 16384);
        }
        void *d_codes = nullptr, *d_off = nullptr;
        if (!rc) rc = selhip_malloc(&d_codes, flat.size());
        if (!rc) rc = selhip_malloc(&d_off, offsets.size() * 8);
        if (!rc) rc = selhip_malloc(&d_hll, (size_t)std::max<int64_t>(n, 1) * 16384);         // the builder's own HLL output (not used: the .hll files are)
        if (!rc) rc = selhip_malloc(&d_aux, (size_t)std::max<int64_t>(n, 1) * m_vec * 8);
        if (!rc) rc = selhip_malloc(&d_hll_s, (size_t)std::max<int64_t>(n, 1) * 16384);
        if (!rc) rc = selhip_malloc(&d_aux_s, (size_t)std::max<int64_t>(n, 1) * m_vec * 8);
        if (!rc) rc = selhip_malloc(&d_perm, (size_t)std::max<int64_t>(n, 1) * 4);
        if (!rc) rc = selhip_malloc(&d_cards_s, (size_t)std::max<int64_t>(n, 1) * 8);
        if (!rc) rc = selhip_memcpy_h2d(d_codes, flat.data(), flat.size());
        if (!rc) rc = selhip_memcpy_h2d(d_off, offsets.data(), offsets.size() * 8);
        if (!rc && n) rc = selhip_build_sketches((const uint8_t*)d_codes, (const int64_t*)d_off, n, 31, (int)m_vec, 0, (uint8_t*)d_hll, (uint64_t*)d_aux, nullptr, nullptr);
        if (!rc) rc = selhip_memcpy_h2d(d_perm, perm.data(), (size_t)n * 4);
        if (!rc) rc = selhip_memcpy_h2d(d_cards_s, cards_s.data(), (size_t)n * 8);
        if (!rc) rc = selhip_memcpy_h2d(d_hll_s, hll_s.data(), hll_s.size());
        if (!rc && n) rc = selhip_permute_rows(d_aux, d_aux_s, (const int32_t*)d_perm, n, (int64_t)m_vec * 8, nullptr);
        if (!rc) rc = selhip_device_synchronize();
        selhip_free(d_codes); selhip_free(d_off);
        if (!rc) rc = selhip_ctx_attach(ctx, (const uint8_t*)d_hll_s, (const uint64_t*)d_aux_s, (const double*)d_cards_s, n, (int)m_vec, 14);
        if (rc) { std::cerr << "setup failed: " << selhip_last_error(ctx) << " / " << selhip_last_error(nullptr) << "\n"; return 4; }
    } else {
        if (selhost_dataset_load(&ds, list_file.c_str(), (unsigned)mh_size, 0, 1, 8)) { std::cerr << selhost_last_error() << "\n"; exit(-1); }
        n = selhost_dataset_size(ds);
        if (selhip_ctx_upload(ctx, selhost_dataset_hll(ds), selhost_dataset_aux(ds), selhost_dataset_cards(ds), n, mh_size, 14)) {
            std::cerr << selhip_last_error(ctx) << "\n"; return 4;
        }
    }
    std::cout << label << ";build_smh;" << threshold << ";" << (now_s() - t0) << std::endl;   // time_smh_cuda.cpp:209-211

    int n_rows = 1, n_bands = 1;
    selhost_banding((unsigned)mh_size, threshold, SELHOST_BANDING_CPU, &n_rows, &n_bands);

    for (int rep = 0; rep < total_rep; ++rep) {
        const int modes[2] = {SELHIP_MODE_SMH, SELHIP_MODE_CB_SMH};
        const char* names[2] = {"smh_a", "CB+smh_a"};
        for (int k = 0; k < 2; ++k) {
            std::cout << label << ";" << names[k] << ";" << threshold << ";";
            double a = now_s();
            int rc = selhip_ctx_run(ctx, modes[k], algo, threshold, n_rows, n_bands, 0, n);   // synchronous
            double dt = now_s() - a;
            if (rc) { std::cerr << "run failed: " << selhip_last_error(ctx) << "\n"; return 5; }
            int64_t st[4];
            selhip_ctx_stats(ctx, st);
            std::cout << dt << ";r:" << n_rows << "_b:" << n_bands << ";pairs:" << st[0] << ";survivors:" << st[1]
                      << ";selected:" << st[2] << ";pairs_per_s:" << (dt > 0 ? (double)st[0] / dt : 0.0) << std::endl;
        }
    }
    selhip_ctx_destroy(ctx);
    if (ds) selhost_dataset_free(ds);
    return 0;
}
