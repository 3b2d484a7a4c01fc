// build_sketch_main.cpp -- `build_sketch`: drop-in for the reference's sketch builder (src/build_sketch.cpp:186-295,
// CLI `-l filelist -t nthreads -a aux_memory -c criterion`, README.md:44-56): for every FASTA(.gz) in the list it
// writes <file>.hll (HyperLogLog p=14) and, by criterion, <file>.hll_<p> (hll_a / hll_an, p = ctz(aux_bytes)) or
// <file>.smh<m> (smh_a, m = aux_bytes/8) -- byte-identical (after gunzip) to what the reference writes.
// FASTA parsing and file I/O run on host threads; every k-mer of every genome is sketched on the GPU
// (selhip_build_sketches), genomes batched so that a batch's bases fit a fixed device buffer.
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/selection_hip.h"
#include "../../../include/selection_host.h"

static void load_file_list(std::vector<std::string>& files, const std::string& list_file) {   // build_sketch.cpp:153-180
    if (list_file.empty()) { std::cerr << "No input file provided\n"; exit(-1); }
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
    std::string list_file = "", criterion = "";
    unsigned threads = 8, aux_bytes = 256;
    const int k = 31;                                                           // build_sketch.cpp:190
    int c;
    while ((c = getopt(argc, argv, "l:t:a:c:x")) != -1) {
        switch (c) {
            case 'l': list_file = optarg; break;
            case 't': threads = (unsigned)std::stoi(optarg); break;
            case 'a': aux_bytes = (unsigned)std::stoi(optarg); break;
            case 'c': criterion = optarg; break;
            case 'x': std::cout << "Usage: -l -t -a -c\n"; return 0;
            default: break;
        }
    }
    std::vector<std::string> files;
    load_file_list(files, list_file);
    int mode = 0;                       // 1 = aux hll, 2 = smh
    if (criterion == "hll_a" || criterion == "hll_an") mode = 1;
    else if (criterion == "smh_a") mode = 2;
    const int p_aux = mode == 1 ? __builtin_ctz(aux_bytes ? aux_bytes : 1) : 0;               // build_sketch.cpp:239
    const uint32_t m_arg = aux_bytes / 8;                                                      // build_sketch.cpp:272
    const uint32_t m = mode == 2 ? selhost_smh_vecsize(m_arg) : 0;

    if (selhip_device_count() <= 0) { std::cerr << "build_sketch: no MI355X (gfx950) device available\n"; return 3; }

    const size_t n = files.size();
    const size_t batch_cap = (size_t)1 << 31;             // bases per GPU batch
    size_t done = 0;
    while (done < n) {
        std::vector<int64_t> offsets{0};
        size_t b_end = done, total = 0;
        while (b_end < n) {                                // size files one at a time until the batch is full
            int64_t len = selhost_fasta_codes(files[b_end].c_str(), nullptr, 0);
            if (len < 0) { std::cerr << "ERROR: Could not open the file " << files[b_end] << ".\n"; len = 0; }   // build_sketch.cpp:44-48
            if (total && total + (size_t)len > batch_cap) break;
            total += (size_t)len;
            offsets.push_back((int64_t)total);
            ++b_end;
        }
        std::vector<uint8_t> flat(total ? total : 1);
        const size_t nb = b_end - done;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
        for (size_t j = 0; j < nb; ++j) {
            const size_t len = (size_t)(offsets[j + 1] - offsets[j]);
            if (len) selhost_fasta_codes(files[done + j].c_str(), flat.data() + offsets[j], len);
        }
        void *d_codes = nullptr, *d_off = nullptr, *d_hll = nullptr, *d_smh = nullptr, *d_aux = nullptr;
        std::vector<uint8_t> hll(nb * 16384), aux(mode == 1 ? nb << p_aux : 0);
        std::vector<uint64_t> smh(mode == 2 ? nb * m : 0);
        int rc = selhip_malloc(&d_codes, flat.size());
        if (!rc) rc = selhip_malloc(&d_off, offsets.size() * 8);
        if (!rc) rc = selhip_malloc(&d_hll, hll.size());
        if (!rc && mode == 2) rc = selhip_malloc(&d_smh, smh.size() * 8);
        if (!rc && mode == 1) rc = selhip_malloc(&d_aux, aux.size());
        if (!rc) rc = selhip_memcpy_h2d(d_codes, flat.data(), flat.size());
        if (!rc) rc = selhip_memcpy_h2d(d_off, offsets.data(), offsets.size() * 8);
        if (!rc) rc = selhip_build_sketches((const uint8_t*)d_codes, (const int64_t*)d_off, (int64_t)nb, k, (int)m, p_aux,
                                            (uint8_t*)d_hll, (uint64_t*)d_smh, (uint8_t*)d_aux, nullptr);
        if (!rc) rc = selhip_memcpy_d2h(hll.data(), d_hll, hll.size());
        if (!rc && mode == 2) rc = selhip_memcpy_d2h(smh.data(), d_smh, smh.size() * 8);
        if (!rc && mode == 1) rc = selhip_memcpy_d2h(aux.data(), d_aux, aux.size());
        selhip_free(d_codes); selhip_free(d_off); selhip_free(d_hll); selhip_free(d_smh); selhip_free(d_aux);
        if (rc) { std::cerr << "build_sketch: " << selhip_last_error(nullptr) << "\n"; return 4; }
        int werr = 0;
#pragma omp parallel for schedule(dynamic) num_threads(threads)
        for (size_t j = 0; j < nb; ++j) {
            const std::string& fn = files[done + j];
            int w = selhost_write_hll((fn + ".hll").c_str(), hll.data() + j * 16384, 14);                  // build_sketch.cpp:233
            if (!w && mode == 1) w = selhost_write_hll((fn + ".hll_" + std::to_string(p_aux)).c_str(), aux.data() + (j << p_aux), (uint32_t)p_aux);
            if (!w && mode == 2) w = selhost_write_smh((fn + ".smh" + std::to_string(m_arg)).c_str(), smh.data() + j * m, m);   // :286
            if (w) {
#pragma omp critical
                werr = w;
            }
        }
        if (werr) { std::cerr << "build_sketch: " << selhost_last_error() << "\n"; return 5; }
        done = b_end;
    }
    if (mode == 0) printf("Option -c invalid. The accepted criteria are hll_a, hll_an and smh_a.\n");    // build_sketch.cpp:289
    return 0;
}
