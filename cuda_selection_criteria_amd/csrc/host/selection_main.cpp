// selection_main.cpp -- `selection`: drop-in for the reference's GPU selection driver
// (src/selection_cuda.cpp:59-189; README.md:60-66 documents it as `selection -l -h -a -b`),
// producing the OUTPUT of the reference's CPU program src/selection.cpp:228-300 (criterion smh_a):
// lines "fn1 fn2 <jaccard %f>" in rank order.
//
//   -l <file>   list of genome paths; <path>.hll and <path>.smh<m> must exist (written by build_sketch)
//   -h <tau>    similarity threshold (float, std::stof like selection.cpp:103)
//   -a <bytes>  auxiliary memory per genome; m = bytes/8 SuperMinHash buckets (selection.cpp:231)
//   -b <n>      accepted for CLI compatibility (CUDA block size in the reference); ignored
//   -c <crit>   smh_a (default; the only criterion of the reference's GPU driver, selection_cuda.cpp:64), or hll_a /
//               hll_an as in the CPU program (src/selection.cpp:122-227: auxiliary HLL p = ctz(aux_bytes), file .hll_<p>)
//   -t <n>      host threads for loading sketches (selection.cpp:97)
//   -g <n>      number of GPUs to shard the pair space over (default 1; any criterion); selected pairs gathered over RCCL/xGMI
//   -n          no CB pruning ("smh_a" mode of experiments/src/time_smh.cpp:229-257)
//   -A <algo>   stage-1 algorithm: auto | stream | sig | hashjoin (sort-based: same result, sub-quadratic)
//   -F <0|1>    estimator flavour: 1 = FMA (reference Makefile build on FMA hosts, default), 0 = strict
//   -B <n>      out-of-core: keep the sketches in host memory and process the pair space in blocks of n genomes
//               (selhip_ooc_select; same output); 0 = everything resident on the device (default)
//   -o <file>   write the result as a binary result file (include/selection_host.h: "SELR" format: records + name table)
//               instead of text on stdout
//   -r <file>   no selection: print the text form of a result file written with -o (needs no GPU)
//   -x          usage
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/selection_hip.h"
#include "../../../include/selection_host.h"

int main(int argc, char* argv[]) {
    std::string list_file = "";
    float threshold = 0.9f;              // selection_cuda.cpp:62
    int aux_bytes = 256;                 // selection_cuda.cpp:63
    std::string criterion = "smh_a";
    int threads = 8, n_gpus = 1, mode = SELHIP_MODE_CB_SMH, algo = SELHIP_ALGO_AUTO, fp_mode = SELHIP_FP_FMA;
    long long ooc_block = 0;
    std::string out_file = "", dump_file = "";
    int c;
    while ((c = getopt(argc, argv, "xl:b:a:h:c:t:g:nA:F:B:o:r:")) != -1) {
        switch (c) {
            case 'x': std::cout << "Usage: -l -h -a -b [-c smh_a] [-t threads] [-g gpus] [-n] [-A auto|stream|sig] [-F 0|1] [-B block] [-o file] | -r file\n"; return 0;
            case 'B': ooc_block = std::stoll(optarg); break;
            case 'o': out_file = optarg; break;
            case 'r': dump_file = optarg; break;
            case 'l': list_file = optarg; break;
            case 'b': break;
            case 'a': aux_bytes = std::stoi(optarg); break;
            case 'h': threshold = std::stof(optarg); break;
            case 'c': criterion = optarg; break;
            case 't': threads = std::stoi(optarg); break;
            case 'g': n_gpus = std::stoi(optarg); break;
            case 'n': mode = SELHIP_MODE_SMH; break;
            case 'A': algo = !strcmp(optarg, "stream") ? SELHIP_ALGO_STREAM : !strcmp(optarg, "sig") ? SELHIP_ALGO_SIG : !strcmp(optarg, "hashjoin") ? SELHIP_ALGO_HASHJOIN : SELHIP_ALGO_AUTO; break;
            case 'F': fp_mode = std::stoi(optarg) ? SELHIP_FP_FMA : SELHIP_FP_STRICT; break;
            default: break;
        }
    }
    if (!dump_file.empty()) {
        selhost_results* res = nullptr;
        if (selhost_read_results(&res, dump_file.c_str())) { std::cerr << "selection: " << selhost_last_error() << "\n"; return 5; }
        const int64_t need = selhost_results_text(res, nullptr, 0);
        if (need < 0) { std::cerr << "selection: " << selhost_last_error() << "\n"; selhost_results_free(res); return 5; }
        std::string text((size_t)need + 1, '\0');
        selhost_results_text(res, &text[0], text.size());
        text.resize((size_t)need);
        std::cout << text;
        selhost_results_free(res);
        return 0;
    }
    int crit = SELHIP_CRIT_SMH_A;
    if (criterion == "hll_a") crit = SELHIP_CRIT_HLL_A;
    else if (criterion == "hll_an") crit = SELHIP_CRIT_HLL_AN;
    else if (criterion != "smh_a") {
        std::cout << "Option -c invalid. The accepted criteria are hll_a, hll_an and smh_a.\n";    // selection.cpp:293
        return 0;
    }
    if (list_file.empty()) { std::cerr << "No input file provided\n"; exit(-1); }   // selection.cpp:40-44
    const unsigned m = crit == SELHIP_CRIT_SMH_A ? (unsigned)aux_bytes / 8 : 0;                      // selection.cpp:231
    const unsigned p_aux = crit == SELHIP_CRIT_SMH_A ? 0 : (unsigned)__builtin_ctz(aux_bytes ? aux_bytes : 1);   // :125

    selhost_dataset* ds = nullptr;
    int rc = selhost_dataset_load(&ds, list_file.c_str(), m, p_aux, fp_mode, threads);
    if (rc) { std::cerr << selhost_last_error() << "\n"; exit(-1); }
    const int64_t n = selhost_dataset_size(ds);

    int n_rows = 1, n_bands = 1;
    if (m) selhost_banding(m, threshold, SELHOST_BANDING_CPU, &n_rows, &n_bands);
    // hll_a / hll_an do not read SuperMinHash buckets: a one-bucket placeholder keeps the upload call uniform
    std::vector<uint64_t> no_smh(m ? 0 : (size_t)(n > 0 ? n : 1), 0);
    const uint64_t* aux_ptr = m ? selhost_dataset_aux(ds) : no_smh.data();
    const int m_up = m ? (int)m : 1;

    const int avail = selhip_device_count();
    if (avail <= 0) { std::cerr << "selection: no MI355X (gfx950) device available: " << selhip_last_error(nullptr) << "\n"; return 3; }
    if (n_gpus < 1) n_gpus = 1;
    if (n_gpus > avail) n_gpus = avail;

    std::vector<std::vector<selhip_pair_t>> parts(1);
    if (ooc_block > 0) {
        // sketches stay in host memory; block pairs are uploaded in turn (two at a time: upload overlaps compute)
        int64_t cnt = 0, cap = 1 << 20;
        for (int attempt = 0; attempt < 2; ++attempt) {
            parts[0].resize((size_t)cap);
            int r = selhip_ooc_select(0, selhost_dataset_hll(ds), aux_ptr, selhost_dataset_cards(ds),
                                      p_aux ? selhost_dataset_aux_hll(ds) : nullptr, (int)p_aux, crit, n, m_up, 14,
                                      mode, algo, fp_mode, threshold, n_rows, n_bands, ooc_block, 2, parts[0].data(), cap, &cnt, nullptr);
            if (r == SELHIP_E_OVERFLOW && attempt == 0) { cap = cnt; continue; }
            if (r) { std::cerr << "selection: " << selhip_last_error(nullptr) << "\n"; return 4; }
            break;
        }
        parts[0].resize((size_t)cnt);
        n_gpus = 1;
    } else if (n_gpus > 1) {
        // one process, one thread + context per device, selected pairs gathered over RCCL/xGMI (host merge if RCCL is
        // unavailable): selhip_multi_select
        std::vector<int> devs((size_t)n_gpus);
        for (int g = 0; g < n_gpus; ++g) devs[(size_t)g] = g;
        int64_t cnt = 0, cap = 1 << 20;
        for (int attempt = 0; attempt < 2; ++attempt) {
            parts[0].resize((size_t)cap);
            int r = selhip_multi_select(devs.data(), n_gpus, selhost_dataset_hll(ds), aux_ptr, selhost_dataset_cards(ds),
                                        p_aux ? selhost_dataset_aux_hll(ds) : nullptr, (int)p_aux, crit, n, m_up, 14,
                                        mode, algo, fp_mode, threshold, n_rows, n_bands, SELHIP_GATHER_RCCL_OR_HOST,
                                        parts[0].data(), cap, &cnt, nullptr);
            if (r == SELHIP_E_OVERFLOW && attempt == 0) { cap = cnt; continue; }
            if (r) { std::cerr << "selection: " << selhip_last_error(nullptr) << "\n"; return 4; }
            break;
        }
        parts[0].resize((size_t)cnt);
        n_gpus = 1;                                   // one merged, sorted part
    } else {
        n_gpus = 1;
        selhip_ctx* ctx = nullptr;
        int r = selhip_ctx_create(&ctx, 0);
        if (r) { std::cerr << "selection: " << selhip_last_error(nullptr) << "\n"; return 4; }
        selhip_ctx_set_fp_mode(ctx, fp_mode);
        r = selhip_ctx_upload(ctx, selhost_dataset_hll(ds), aux_ptr, selhost_dataset_cards(ds), n, m_up, 14);
        if (!r && p_aux) r = selhip_ctx_upload_aux_hll(ctx, selhost_dataset_aux_hll(ds), (int)p_aux);
        if (!r) r = selhip_ctx_set_criterion(ctx, crit);
        if (!r) r = selhip_ctx_run(ctx, mode, algo, threshold, n_rows, n_bands, 0, n);
        if (!r) {
            int64_t cnt = selhip_ctx_result_count(ctx);
            parts[0].resize((size_t)cnt);
            r = selhip_ctx_fetch(ctx, parts[0].data(), cnt);
        }
        if (r) { std::cerr << "selection: " << selhip_last_error(ctx) << "\n"; selhip_ctx_destroy(ctx); return 4; }
        selhip_ctx_destroy(ctx);
    }

    if (!out_file.empty()) {
        std::vector<const char*> names((size_t)n);
        for (int64_t g = 0; g < n; ++g) names[(size_t)g] = selhost_dataset_name(ds, g);
        static_assert(sizeof(selhost_pair_t) == sizeof(selhip_pair_t), "record layouts must agree");
        const int r = selhost_write_results(out_file.c_str(), reinterpret_cast<const selhost_pair_t*>(parts[0].data()), (int64_t)parts[0].size(),
                                            names.data(), n, threshold);
        if (r) { std::cerr << "selection: " << selhost_last_error() << "\n"; return 5; }
        selhost_dataset_free(ds);
        return 0;
    }
    // shards are contiguous row ranges and each part is sorted by (i,k): concatenation = print order
    std::string out;
    char line[8192];
    for (int g = 0; g < n_gpus; ++g)
        for (const selhip_pair_t& pr : parts[(size_t)g]) {
            int w = selhost_format_line(selhost_dataset_name(ds, pr.i), selhost_dataset_name(ds, pr.k), pr.jaccard, line, sizeof line);
            if (w > 0) out.append(line, (size_t)w);
        }
    std::cout << out;
    selhost_dataset_free(ds);
    return 0;
}
