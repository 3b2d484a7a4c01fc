// kernel_pairlist.cuh -- explicit pair lists: drop-in launcher path and test building blocks.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// explicit pair lists (drop-in launchers and test building blocks): one LANE per pair.
//   flags[j] = pair passes [e_y != 0] [CB] smh_a ;  optionally compacts survivors.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock)
void pairlist_smh_kernel(const u64* __restrict__ aux, int m, int n_rows, int n_bands,
                         const selhip_int2_t* __restrict__ pairs, long long n_pairs,
                         const double* __restrict__ cards, double tau, int check_cards, int use_cb,
                         uint8_t* __restrict__ flags,
                         selhip_int2_t* __restrict__ surv, u64 surv_cap, u64* __restrict__ surv_count) {
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_pairs) return;
    const selhip_int2_t pr = pairs[j];
    bool ok = true;
    if (check_cards) {
        const u64 e1 = selhip::trunc_card(cards[pr.x]), e2 = selhip::trunc_card(cards[pr.y]);
        if (e2 == 0) ok = false;                                             // selection.cpp:281
        else if (use_cb && !cb_pred(tau, e1, e2)) ok = false;                // selection.cpp:282
    }
    if (ok) ok = smh_a_lane(aux + (long long)pr.x * m, aux + (long long)pr.y * m, n_rows, n_bands);
    if (flags) flags[j] = ok ? 1 : 0;
    if (ok && surv) {
        u64 idx = atomicAdd(surv_count, 1ull);
        if (idx < surv_cap) surv[idx] = pr;
    }
}

__global__ __launch_bounds__(kBlock)
void match_count_kernel(const u64* __restrict__ aux, int m, const selhip_int2_t* __restrict__ pairs,
                        long long n_pairs, int32_t* __restrict__ matches) {
    // one wave per pair: lanes stride the buckets, v_cmp_eq_u64 masks counted with s_bcnt1
    const int lane = threadIdx.x & (kWave - 1);
    long long j = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    if (j >= n_pairs) return;
    const selhip_int2_t pr = pairs[j];
    const u64* a = aux + (long long)pr.x * m;
    const u64* b = aux + (long long)pr.y * m;
    int cnt = 0;
    for (int t0 = 0; t0 < m; t0 += kWave) {
        int t = t0 + lane;
        bool eq = (t < m) && (a[t] == b[t]);
        cnt += __popcll(__ballot(eq));
    }
    if (lane == 0) matches[j] = cnt;
}

__global__ void truncate_cards_kernel(const double* __restrict__ cards, int n, u64* __restrict__ ecard) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ecard[i] = selhip::trunc_card(cards[i]);
}

__global__ void iota_pairs_kernel(selhip_int2_t* pairs, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { pairs[i].x = i; pairs[i].y = i; }
}

}  // namespace
