// kernel_bounds.cuh -- cb_bounds_kernel: truncated cardinalities, CB cut-off per row, evaluated-pair count.
// Part of libselhip.so; included by selection_kernels.hip only (one translation unit, anonymous namespace).
#pragma once

namespace {

// ---------------------------------------------------------------------------------------------
// cb_bounds_kernel: one thread per genome rank.
//   ecard[i] = (size_t)cards[i]                                         (selection.cpp:275,280)
//   hi[i]    = last k such that CB(tau, e_i, e_k) holds, or N-1 without CB  (criteria_sketch.hpp:45-49;
//              the loop `break`s at the first failing k (selection.cpp:282-283); e is ascending so the
//              predicate is monotone and the break is exactly "k <= hi(i)")
//   z0       = first rank with e != 0  (`if(e2 == 0) continue`, selection.cpp:281), raised to cand_begin when the pass is
//              restricted to candidates k >= cand_begin (rectangular passes of the out-of-core driver): every kernel
//              takes its first candidate as max(i+1, z0), so this one number carries the restriction
// It is the first kernel of a pass, so it also clears the per-row counters of the stage-2 grouping (saves a memset).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cb_pred(double tau, u64 e1, u64 e2) {
    double gamma = (double)e1 / (double)e2;      // criteria_sketch.hpp:47 (size_t -> double, IEEE divide)
    return gamma >= tau;
}

// returns the row's pairs inside the pair space when `count_only` (the caller adds them up); otherwise adds them to pc->n_evaluated itself
__device__ __forceinline__ long long cb_bounds_body(int i, const double* __restrict__ cards, int n, double tau, int use_cb,
                                                    RowMap rm, u64* __restrict__ ecard, int* __restrict__ hi,
                                                    PassCounters* __restrict__ pc, int cand_begin, bool count_only = false) {
    if (i >= n) return 0;
    double c = cards[i];
    u64 e1 = selhip::trunc_card(c);
    ecard[i] = e1;
    if (i > 0) {
        double cp = cards[i - 1];
        if (c < cp) pc->unsorted = 1;
        if (e1 != 0 && selhip::trunc_card(cp) == 0) pc->z0p1 = max(i, cand_begin) + 1;
    } else if (e1 != 0) {
        pc->z0p1 = max(0, cand_begin) + 1;
    }
    int h = n - 1;
    if (use_cb) {
        // largest k in (i, n) with (e_k == 0 || CB(e1, e_k)); predicate is true on a prefix
        int lo = i, hi_ = n - 1;      // invariant: pred(lo) true (k = i itself counts as true), answer in [lo, hi_]
        while (lo < hi_) {
            int mid = lo + (hi_ - lo + 1) / 2;
            u64 e2 = selhip::trunc_card(cards[mid]);
            bool ok = (e2 == 0) || cb_pred(tau, e1, e2);
            if (ok) lo = mid; else hi_ = mid - 1;
        }
        h = lo;
    }
    hi[i] = h;
    if (rm.owns(i)) {
        // pairs of this row inside the pair space: k in [max(i+1, z0'), h]; z0 may not be published yet,
        // so count candidates with e_k != 0 directly from the sorted property: e_k == 0 only for k < z0.
        // first k > i with e_k != 0: if e1 != 0 it is i+1, else binary search.
        int first = i + 1;
        if (e1 == 0) {
            int lo = i + 1, hi2 = n;          // first index in [i+1, n) with e != 0
            while (lo < hi2) {
                int mid = lo + (hi2 - lo) / 2;
                if (selhip::trunc_card(cards[mid]) != 0) hi2 = mid; else lo = mid + 1;
            }
            first = lo;
        }
        if (first < cand_begin) first = cand_begin;
        long long cnt = (long long)h - first + 1;
        if (cnt > 0) {
            if (count_only) return cnt;
            atomicAdd(&pc->n_evaluated, (u64)cnt);
        }
    }
    return 0;
}

// the counter blocks of the NEXT pass (the other of the context's two sets) are cleared by the first kernel of this one
__device__ __forceinline__ void zero_next_counters(int t, int n_threads, PassCounters* __restrict__ zero_pc, int n_blocks) {
    if (!zero_pc) return;
    u64* w = reinterpret_cast<u64*>(zero_pc);
    const int words = n_blocks * (int)(sizeof(PassCounters) / 8);
    for (int j = t; j < words; j += n_threads) w[j] = 0;
}

__global__ void cb_bounds_kernel(const double* __restrict__ cards, int n, double tau, int use_cb,
                                 RowMap rm, u64* __restrict__ ecard, int* __restrict__ hi,
                                 PassCounters* __restrict__ pc, int* __restrict__ csr_zero, int csr_zero_n, int cand_begin,
                                 PassCounters* __restrict__ zero_pc, u64* __restrict__ seg_zero, int seg_zero_n) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    zero_next_counters(t, (int)(gridDim.x * blockDim.x), zero_pc, kCounterBlocks);
    for (int j = t; j < seg_zero_n; j += (int)(gridDim.x * blockDim.x)) seg_zero[j] = 0;      // the join's append-segment counters
    // stage 2's per-row counters (count / fill cursors, one set per chunk lane) for this pass
    for (int j = t; j < csr_zero_n; j += (int)(gridDim.x * blockDim.x)) csr_zero[j] = 0;
    cb_bounds_body(t, cards, n, tau, use_cb, rm, ecard, hi, pc, cand_begin);
}

}  // namespace
