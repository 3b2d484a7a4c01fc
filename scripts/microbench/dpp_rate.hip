// microbenchmark (GPU box): issue rate of v_xor_b32 with and without a DPP row broadcast, and of v_pk_min_u16 / v_min3_u32.
//   hipcc --offload-arch=gfx950 -O3 -o dpp_rate dpp_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    uint32_t a[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i; q[i] = blockIdx.x * 40503u + i * 77u + threadIdx.x; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t nx = a[(i + 1) & 7], qq = q[(i + u) & 7];
                if (KIND == 0) a[i] = qq ^ nx;                                                                                   // v_xor_b32
                else if (KIND == 1) a[i] = qq ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nx, 0x150 + 3, 0xF, 0xF, true);   // v_xor_b32_dpp row_newbcast
                else if (KIND == 2) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, nx), __builtin_bit_cast(us2, qq)));  // v_pk_min_u16
                else if (KIND == 3) a[i] = min(min(nx, qq), q[(i + u + 3) & 7]);                                                // v_min3_u32
                else if (KIND == 4) a[i] = qq ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)nx, 0x111, 0xF, 0xF, true);       // v_xor_b32_dpp row_shr:1
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const char* names[] = {"v_xor_b32", "v_xor_b32_dpp row_newbcast", "v_pk_min_u16", "v_min3_u32", "v_xor_b32_dpp row_shr"};
    const int blocks = 4096, iters = 2000;
    for (int kind = 0; kind < 5; ++kind) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            switch (kind) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters); break;
            }
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * iters * 64;
        printf("%-28s %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[kind], best, best * 1e-3 * 2.4e9 * 1024 / wave_instr);
    }
    return 0;
}
