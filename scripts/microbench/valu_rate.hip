// microbenchmark (GPU box): VALU issue cost of the instructions the signature join is built from, with control rows, at a
// PINNED occupancy of 1, 2, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate > profiles/rNN_valu_rate.txt
//
// Question it answers: MI355X_MICROARCH.md's constants table gives `v_fma_f32 (wave64) 2 cyc;
// one wave alone 4`; the join's roofline had been priced at 4 cycles per wave64 VALU instruction per SIMD.  Which holds for
// v_xor_b32_dpp / v_pk_min_u16?
//
// Method: 256-thread blocks (4 waves, one per SIMD), dynamic LDS = 160 KiB / W so that exactly W blocks fit a CU
// => W waves per SIMD; 256 CUs x W x 8 blocks.  Every row is 64 inline-asm instructions per loop iteration on 8 independent
// register chains (so a chain's own latency never limits issue).  Two clocks: HIP events around the launch (wall: chip-wide
// wave-instructions / s, converted to cycles per instruction per SIMD at the shader clock measured in the same run) and
// s_memtime inside each wave (cycles one wave needs per instruction while W waves share its SIMD; / W = per-SIMD cost).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Kind { K_FMA_F32, K_ADD_U32, K_XOR_B32, K_PK_MIN_U16, K_XOR_DPP, K_MIN3_U32, K_PK_FMA_F32, K_OR3_B32, K_JOINMIX, K_JOINMIX_DEP,
            K_SAD_U16, K_PERM_B32, K_CMP_EQ_U32, K_FMA_F64, K_MQSAD_U32_U8, K_MUL_LO_U32, K_PK_ADD_U16,
            K_MIN_U32, K_AND_B32, K_XOR_SGPR, K_LSHRREV_B32, K_MOV_DPP, K_AND_OR_B32, K_BFI_B32, K_JOINMIX_SGPR, K_MIN_U16, K_XOR_SDWA, K_ADD_F32, K_MAX_U32,
            K_MIX_ALT, K_MIX_G4, K_MIX_G8, K_MIX_XOR_AND, K_MIX_XOR_MIN16, K_OR_B32, K_SUB_U32, K_MIN_I32, K_MIN_F32, K_MUL_F32, K_MOV_B32, K_CMP_EQ_U64, K_CMP_EQ_U32_SGPR,
            K_BITOP3, K_BCNT, K_MIX_AND2_BCNT, K_MIX_BITOP2_BCNT, K_MIX_AND_RUN_BCNT_RUN, K_NKINDS };

const char* kNames[K_NKINDS] = {
    "v_fma_f32", "v_add_u32", "v_xor_b32", "v_pk_min_u16", "v_xor_b32_dpp row_newbcast", "v_min3_u32", "v_pk_fma_f32",
    "v_or3_b32", "join mix: xor_dpp + pk_min_u16 (4 chains)", "join mix, ONE chain (dependent)", "v_sad_u16", "v_perm_b32",
    "v_cmp_eq_u32 (vcc)", "v_fma_f64", "v_mqsad_u32_u8", "v_mul_lo_u32", "v_pk_add_u16",
    "v_min_u32", "v_and_b32", "v_xor_b32 v, s, v (SGPR operand)", "v_lshrrev_b32", "v_mov_b32_dpp row_newbcast", "v_and_or_b32", "v_bfi_b32",
    "join mix 2: v_xor_b32 (SGPR query) + v_pk_min_u16", "v_min_u16", "v_xor_b32_sdwa", "v_add_f32", "v_max_u32",
    "mix: v_xor_b32 (VGPR) / v_pk_min_u16 alternating", "mix: 4 x v_xor_b32 then 4 x v_pk_min_u16", "mix: 8 x v_xor_b32 then 8 x v_pk_min_u16",
    "mix: v_xor_b32 / v_and_b32 alternating (both 2-cycle)", "mix: v_xor_b32 / v_min_u16 alternating (both 2-cycle)", "v_or_b32", "v_sub_u32", "v_min_i32", "v_min_f32", "v_mul_f32", "v_mov_b32", "v_cmp_eq_u64 -> SGPR pair", "v_cmp_eq_u32 -> SGPR pair",
    "v_bitop3_b32 (3-input boolean)", "v_bcnt_u32_b32 (accumulating)", "mix: 2 x v_and_b32 then v_bcnt_u32_b32", "mix: 2 x v_bitop3_b32 then v_bcnt_u32_b32",
    "mix: runs of 16-32 v_and_b32 and of 8 v_bcnt_u32_b32 (3:1)"};

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* __restrict__ out, unsigned long long* __restrict__ cyc, int iters) {
    extern __shared__ uint32_t pin_lds[];        // only pins the occupancy
    uint32_t a[8], q[8], t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 97u + 1u; q[i] = blockIdx.x * 40503u + i * 77u + threadIdx.x + 3u; t[i] = 0; }
    uint32_t sq[8];
    unsigned long long sm[4] = {0, 0, 0, 0}, dq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dq[i] = ((unsigned long long)threadIdx.x << 32) | (unsigned)(i * 2654435761u);
#pragma unroll
    for (int i = 0; i < 8; ++i) sq[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 7919u + i * 104729u + (uint32_t)iters));
    double dd[8], d1 = 1.0000001;
#pragma unroll
    for (int i = 0; i < 8; ++i) dd[i] = (double)threadIdx.x + i;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    v2u p2[8];
    v4u p4[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) p2[i] = v2u{a[i], q[i]};
#pragma unroll
    for (int i = 0; i < 4; ++i) p4[i] = v4u{a[i], q[i], a[i + 4], q[i + 4]};
    if (threadIdx.x == 100000) pin_lds[0] = 1;  // never true; keeps the allocation referenced
    __syncthreads();
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (KIND == K_FMA_F32) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_ADD_U32) {
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_XOR_B32) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_PK_MIN_U16) {
#define X(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_XOR_DPP) {
                // DPP on the operand that is never written in the loop (as in the join: the query registers)
#define X(i) asm volatile("v_xor_b32_dpp %0, %1, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIN3_U32) {
#define X(i) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_PK_FMA_F32) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p2[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_OR3_B32) {
#define X(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_JOINMIX) {
                // exactly the join's inner loop: t = c ^ bcast(q); acc = pk_min(acc, t); 4 accumulator chains
                // (four xors, then four mins, as the compiler schedules the join: no DPP-write -> read wait state is needed)
#define X(i) asm volatile("v_xor_b32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t[i & 3]) : "v"(q[i]), "v"(q[(i + 3) & 7]));
#define Y(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i & 3]) : "v"(t[i & 3]));
                X(0) X(1) X(2) X(3) Y(0) Y(1) Y(2) Y(3)
#undef X
#undef Y
            } else if constexpr (KIND == K_JOINMIX_DEP) {
#define X(i) asm volatile("v_xor_b32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(t[0]) : "v"(q[i]), "v"(q[(i + 3) & 7])); \
             asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[0]) : "v"(t[0]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (KIND == K_SAD_U16) {
#define X(i) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_PERM_B32) {
#define X(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_CMP_EQ_U32) {
#define X(i) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(a[i]), "v"(q[i]) : "vcc");
                REP8(X)
#undef X
            } else if constexpr (KIND == K_FMA_F64) {
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[i]) : "v"(d1));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MQSAD_U32_U8) {
#define X(i) asm volatile("v_mqsad_u32_u8 %0, %1, %2, %0" : "+v"(p4[i & 3]) : "v"(p2[i]), "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MUL_LO_U32) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIN_U32) {
#define X(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MAX_U32) {
#define X(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_AND_B32) {
#define X(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_XOR_SGPR) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "s"(sq[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_LSHRREV_B32) {
#define X(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MOV_DPP) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_AND_OR_B32) {
#define X(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_BFI_B32) {
#define X(i) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_JOINMIX_SGPR) {
                // candidate join form: query dword in an SGPR (s_load), t = c ^ q on the plain VOP2, acc = pk_min(acc, t)
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i & 3]) : "s"(sq[i]), "v"(q[(i + 3) & 7]));
#define Y(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i & 3]) : "v"(t[i & 3]));
                X(0) X(1) X(2) X(3) Y(0) Y(1) Y(2) Y(3)
#undef X
#undef Y
            } else if constexpr (KIND == K_MIN_U16) {
#define X(i) asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_XOR_SDWA) {
#define X(i) asm volatile("v_xor_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_ADD_F32) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIX_ALT) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7])); \
             asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(t[(i + 4) & 7]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (KIND == K_MIX_G4) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i & 3]) : "v"(q[i]), "v"(q[(i + 3) & 7]));
#define Y(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i & 3]) : "v"(t[i & 3]));
                X(0) X(1) X(2) X(3) Y(0) Y(1) Y(2) Y(3)
#undef X
#undef Y
            } else if constexpr (KIND == K_MIX_G8) {
                if (u & 1) {
#define Y(i) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(t[i]));
                    REP8(Y)
#undef Y
                } else {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7]));
                    REP8(X)
#undef X
                }
            } else if constexpr (KIND == K_MIX_XOR_AND) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7])); \
             asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(t[(i + 4) & 7]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (KIND == K_MIX_XOR_MIN16) {
#define X(i) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7])); \
             asm volatile("v_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(t[(i + 4) & 7]));
                X(0) X(1) X(2) X(3)
#undef X
            } else if constexpr (KIND == K_OR_B32) {
#define X(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_SUB_U32) {
#define X(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIN_I32) {
#define X(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIN_F32) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MUL_F32) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MOV_B32) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_CMP_EQ_U64) {
#define X(i) asm volatile("v_cmp_eq_u64 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(dq[i]), "v"(dq[(i + 3) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_CMP_EQ_U32_SGPR) {
#define X(i) asm volatile("v_cmp_eq_u32 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(a[i]), "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_BITOP3) {
#define X(i) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x80" : "+v"(a[i]) : "v"(q[i]), "v"(q[(i + 1) & 7]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_BCNT) {
#define X(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            } else if constexpr (KIND == K_MIX_AND2_BCNT) {
                // the bit-plane histogram's tree in plain VOP2 form: two node masks, one population count (9 instructions x ... = 8 per u: 3 groups of 2+1 minus one)
#define X(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7])); \
             asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[i + 4]) : "v"(q[i]), "v"(q[(i + 5) & 7])); \
             asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(t[i]));
                X(0) X(1)
                asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[2]) : "v"(q[2]), "v"(q[5]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[2]) : "v"(t[2]));
#undef X
            } else if constexpr (KIND == K_MIX_BITOP2_BCNT) {
#define X(i) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x80" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7]), "v"(q[(i + 1) & 7])); \
             asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x40" : "=v"(t[i + 4]) : "v"(q[i]), "v"(q[(i + 5) & 7]), "v"(q[(i + 2) & 7])); \
             asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(t[i]));
                X(0) X(1)
                asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x80" : "=v"(t[2]) : "v"(q[2]), "v"(q[5]), "v"(q[6]));
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[2]) : "v"(t[2]));
#undef X
            } else if constexpr (KIND == K_MIX_AND_RUN_BCNT_RUN) {
                // the same work with the plain instructions in a run of 16 and the counts in a run of 8 (three u-steps = one period)
                if (u % 3 == 2) {
#define Y(i) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(t[i]));
                    REP8(Y)
#undef Y
                } else {
#define X(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[i]) : "v"(q[i]), "v"(q[(i + 3) & 7]));
                    REP8(X)
#undef X
                }
            } else if constexpr (KIND == K_PK_ADD_U16) {
#define X(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(q[i]));
                REP8(X)
#undef X
            }
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    uint32_t s = (uint32_t)(sm[0] + sm[1] + sm[2] + sm[3]);
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (uint32_t)dd[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + t[i] + p2[i].x + p2[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p4[i].x + p4[i].y + p4[i].z + p4[i].w;
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + threadIdx.x / 64] = c1 - c0;
}

// shader clock: s_memtime ticks per wall second (a long single-wave spin, timed with events)
__global__ void clock_kernel(unsigned long long* out, int iters) {
    uint32_t a = threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = a; }
}

template <int KIND>
void run_kind(uint32_t* d_out, unsigned long long* d_cyc, int W, int iters, double ticks_per_s, double shader_hz) {
    const int lds_bytes = (160 * 1024) / W;
    (void)hipFuncSetAttribute((const void*)rate_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks = 256 * W * 8;
    const int per_iter = (KIND == K_JOINMIX || KIND == K_JOINMIX_DEP) ? 8 * 8 : 8 * 8;   // instructions per loop iteration
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), lds_bytes, 0, d_out, d_cyc, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { printf("%-44s W=%d launch failed: %s\n", kNames[KIND], W, hipGetErrorString(e)); return; }
    std::vector<unsigned long long> cyc((size_t)blocks * 4);
    (void)hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    const double med_ticks = (double)cyc[cyc.size() / 2];
    const double wave_instr = (double)blocks * 4 * iters * per_iter;
    // wall: chip-wide wave-instructions per second -> cycles per wave-instruction per SIMD (1024 SIMDs)
    const double wall_cyc = best * 1e-3 * shader_hz * 1024.0 / wave_instr;
    // in-wave: ticks one wave spends per instruction (converted to shader cycles), and / W
    const double wave_cyc = med_ticks / ticks_per_s * shader_hz / ((double)iters * per_iter);
    printf("%-44s W=%d  %8.3f ms  wall: %5.2f cyc/instr/SIMD (%.3e wave-instr/s)   in-wave: %6.2f cyc/instr per wave = %5.2f per SIMD\n",
           kNames[KIND], W, best, wall_cyc, wave_instr / (best * 1e-3), wave_cyc, wave_cyc / W);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
    uint32_t* d_out; unsigned long long* d_cyc;
    (void)hipMalloc(&d_out, (size_t)256 * 8 * 8 * 256 * 4);
    (void)hipMalloc(&d_cyc, (size_t)256 * 8 * 8 * 4 * 8 + 64);
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    printf("device: %s, %d CUs, clockRate %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    // s_memtime ticks per second, and the shader clock: a single wave issues one v_add_u32 per 4 (dependent: ~4-5) cycles -- we
    // only need ticks/s here; shader cycles are taken as ticks IF the tick rate is ~the clock (MI355X_MICROARCH.md: tick = shader cycle)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double ticks_per_s = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(clock_kernel, dim3(1), dim3(64), 0, 0, d_cyc, 2000000);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, d_cyc, 16, hipMemcpyDeviceToHost);
        ticks_per_s = (double)h[0] / (ms * 1e-3);
        printf("clock probe: %llu s_memtime ticks in %.3f ms -> %.1f MHz tick rate (one idle-chip wave)\n", h[0], ms, ticks_per_s / 1e6);
    }
    const char* env = getenv("SHADER_MHZ");
    const double shader_hz = env ? atof(env) * 1e6 : 2.4e9;
    printf("shader clock assumed for the cycle conversion: %.0f MHz (SHADER_MHZ overrides); s_memtime tick rate %.1f MHz\n", shader_hz / 1e6, ticks_per_s / 1e6);
    const int iters = 4000;
    if (getenv("BITPLANE_ONLY")) {                 // the instructions of the bit-plane histogram kernel (csrc/kernel_hllbs.cuh) and their controls
        for (int W : {1, 2, 3, 4, 8}) {
            run_kind<K_AND_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_BITOP3>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_BCNT>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_BFI_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_MIX_AND2_BCNT>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_MIX_BITOP2_BCNT>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
            run_kind<K_MIX_AND_RUN_BCNT_RUN>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        }
        return 0;
    }
    for (int W : {1, 2, 4, 8}) {
        run_kind<K_FMA_F32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_ADD_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_XOR_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_PK_MIN_U16>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_XOR_DPP>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIN3_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_PK_FMA_F32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_OR3_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_JOINMIX>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_JOINMIX_DEP>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_SAD_U16>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_PERM_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_CMP_EQ_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_FMA_F64>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MQSAD_U32_U8>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MUL_LO_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_PK_ADD_U16>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIN_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MAX_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_AND_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_XOR_SGPR>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_LSHRREV_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MOV_DPP>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_AND_OR_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_BFI_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_JOINMIX_SGPR>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIN_U16>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_XOR_SDWA>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_ADD_F32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIX_ALT>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIX_G4>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIX_G8>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIX_XOR_AND>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIX_XOR_MIN16>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_OR_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_SUB_U32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIN_I32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MIN_F32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MUL_F32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_MOV_B32>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_CMP_EQ_U64>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
        run_kind<K_CMP_EQ_U32_SGPR>(d_out, d_cyc, W, iters, ticks_per_s, shader_hz);
    }
    return 0;
}
