// rfq_decode.hip — RFQ -> FASTQ path (stub until the decode kernels land; fails loudly).
#include "rfq_ctx.h"
extern "C" int rfq_decode_batch(rfq_ctx* ctx, const rfq_decode_args* a, rfq_decode_result* res) {
    if (!ctx || !a || !res) return RFQ_E_ARG;
    return rfq_fail(ctx, RFQ_E_STATE, "rfq_decode_batch: not built yet");
}
