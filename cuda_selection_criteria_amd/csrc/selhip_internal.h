// selhip_internal.h -- what the host-only translation units of libselhip.so (selhip_multi.hip, selhip_ooc.hip) share with the
// kernel translation unit (selection_kernels.hip) beyond the public C ABI of include/selection_hip.h.  Not installed, not interface.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/selection_hip.h"

// largest auxiliary-HLL precision any entry point accepts: aux_fused_kernel counts in 16-bit bins (a bin holds up to 2^p_aux)
#define SELHIP_MAX_AUX_P 15

extern "C" {
// message returned by selhip_last_error(NULL) on the calling thread
void selhip_internal_set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// Device arrays OWNED by the context, sized for rows_padded genomes, for a caller that fills them itself (a slice from the host and
// the rest by an RCCL all-gather) and then hands them to selhip_ctx_attach / selhip_ctx_attach_aux_hll.  p_aux = 0: no auxiliary array.
int selhip_internal_reserve_replica(selhip_ctx* ctx, int64_t rows_padded, int m, int p_hll, int p_aux,
                                    uint8_t** d_hll, uint64_t** d_aux, double** d_cards, uint8_t** d_aux_hll);
// the context's stream (hipStream_t as void*)
void* selhip_internal_stream(selhip_ctx* ctx);
}
