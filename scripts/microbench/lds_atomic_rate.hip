// microbenchmark (GPU box): issue rate of conflict-free LDS ops per CU -- ds_add_u32 (no return), ds_write_b32, ds_add_u64,
// ds_write_b128 -- with 10 one-wave blocks per CU as in stage 2a.   hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters, uint32_t seed) {
    __shared__ __attribute__((aligned(16))) uint32_t h[64 * 64];
    const int lane = threadIdx.x;
    for (int i = 0; i < 64; ++i) h[i * 64 + lane] = 0;
    uint32_t r = seed * 2654435761u + lane * 40503u + blockIdx.x;
    uint32_t one = 1;
    uint32_t bins[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { r = r * 1664525u + 1013904223u; bins[u] = (r >> 24) & 63u; }
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4u lds_u128;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t bin = bins[u];
            if (KIND == 0) __hip_atomic_fetch_add((lds_u32*)(uintptr_t)((bin << 8) | (lane * 4)), one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 1) *(volatile lds_u32*)(uintptr_t)((bin << 8) | (lane * 4)) = r;
            else if (KIND == 2) __hip_atomic_fetch_add((lds_u64*)(uintptr_t)(((bin & 31u) << 9) | (lane * 8)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 3) { v4u val = {r, r, r, r}; *(volatile lds_u128*)(uintptr_t)(((bin & 15u) << 10) | (lane * 16)) = val; }
            else if (KIND == 4) { one += *(volatile lds_u32*)(uintptr_t)((bin << 8) | (lane * 4)); }
            else if (KIND == 5) { v4u v = *(volatile lds_u128*)(uintptr_t)(((bin & 15u) << 10) | (lane * 16)); one += v.x + v.w; }
        }
    }
    __syncthreads();
    uint32_t s = one;
    for (int i = 0; i < 64; ++i) s += h[i * 64 + lane];
    out[blockIdx.x * 64 + lane] = s;
}
int main() {
    uint32_t* d; hipMalloc(&d, 65536 * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"ds_add_u32", "ds_write_b32", "ds_add_u64", "ds_write_b128", "ds_read_b32", "ds_read_b128"};
    for (int blocks : {2560, 1024}) {
        for (int kind = 0; kind < 6; ++kind) {
            const int iters = 2000;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (kind) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                    case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(64), 0, 0, d, iters, rep); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            double instr_per_cu = (double)blocks / 256.0 * iters * 16;
            printf("blocks=%d %-14s %.3f ms  -> %.2f cycles/instr/CU at 2.4 GHz\n", blocks, names[kind], best, best * 1e-3 * 2.4e9 / instr_per_cu);
        }
    }
    return 0;
}
