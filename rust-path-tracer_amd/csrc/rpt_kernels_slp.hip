/*
 * rpt_kernels_slp.hip — the kernels of librpt_hip.so that are built WITH the compiler's SLP vectorizer.
 *
 * The vectorizer pairs independent f32 multiplies / adds into v_pk_*_f32 and pays for it in v_mov shuffles; on gfx950 a packed
 * op issues in the time of two plain ones (tools/microbench/valu_rates.hip), so for most kernels it only adds the moves.  Built
 * without it (-fno-slp-vectorize, the Makefile's flags for rpt_hip.hip / rpt_comm.hip): k_shade<0, false, false> 64 -> 57 VGPRs
 * and 13.3 -> 11.9 ms per four DarkCornell batches, the streamed LDS walk 61 -> 52 VGPRs and - 2 %, k_sky - 2 ... 4 %, the
 * global-memory walks - 1 % — but the streamed LDS SHADOW walk 50.8 -> 61.2 ms per four batches (same instruction count; the
 * any-hit leaf body schedules worse).  So that one kernel is instantiated here, in a translation unit of its own with the
 * default flags, and launched through rpt_launch_shadow_stream_lds (profiles/r03_slp.txt).  Same IEEE operations either way:
 * the images do not change.
 */
#include <hip/hip_runtime.h>

#include "rpt_ctx.h"
#include "k_traverse.h"

void rpt_launch_shadow_stream_lds(rpt_ctx *c, uint32_t workgroups, size_t lds_bytes, uint32_t span) {
    k_traverse_shadow_stream<16, RPT_LDS_THREADS><<<workgroups, RPT_LDS_THREADS, lds_bytes, c->stream>>>(c->scene, c->state, c->queues, c->dev_stats.p, span);
}
