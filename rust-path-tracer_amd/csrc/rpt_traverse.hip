/*
 * rpt_traverse.hip — the traversal stages of librpt_hip.so: which kernel of k_traverse.h walks the extension rays / the shadow rays of a context's scene,
 * and on which grid.  Its own translation unit since round 6 (the 57 walk kernels are half of the library's compile time; the three units build in parallel).
 * Entry points (rpt_ctx.h): rpt_launch_nearest, rpt_launch_shadow, rpt_launch_trace_debug, rpt_last_walk_attributes.
 */
#include <hip/hip_runtime.h>

#include "rpt_ctx.h"
#include "k_traverse.h"

namespace {

#ifndef RPT_GLOBAL_THREADS
#define RPT_GLOBAL_THREADS 64      /* one wave: no scene staging to share, and a finished wave frees its stack at once (PBRTest traverse -5 %) */
#endif
constexpr int GLOBAL_THREADS = RPT_GLOBAL_THREADS;   /* workgroup size of the global-memory traversal variants */
constexpr int LDS_THREADS = RPT_LDS_THREADS;     /* (rpt_ctx.h) */


template <int STACK>
static int gstream_stack_width(const rpt_ctx *c) {
    /* a tree of at most 15 levels has fewer than 65 536 nodes, one of at most 23 fewer than 2^24: the entry widths a stack capacity can meet
     * (RPT_STACK_BITS, a test aid, widens the entries of deep trees) */
    if (STACK == 16) return 16;
    const int bits = c->knobs.stack_bits;
    const int w = (c->scene.n_nodes < 65536u && bits <= 16) ? 16 : (c->scene.n_nodes < (1u << 21) && bits <= 21) ? 21 : (c->scene.n_nodes < (1u << 24) && bits <= 24) ? 24 : 32;
    return STACK == 24 && w > 24 ? 24 : w;
}
/* slots per wave of the streamed global-memory walks: as many as keep >= gstream_min_waves waves in the launch, at most `most` per lane */
static uint32_t gstream_span(const rpt_ctx *c, uint32_t most) {
    const uint32_t wanted = c->n_slots / (c->gstream_min_waves * RPT_WAVE);
    const uint32_t g = wanted < 1u ? 1u : (wanted > most ? most : wanted);
    return g * RPT_WAVE;
}
/* spans of the streamed LDS walks: persistent workgroups (as many as stay resident: 2 per CU) fetch spans of slots from a launch-wide counter; a span = 1/32 of a
 * workgroup's share, between 1 and 8 slots per lane (measured at 33 M slots: 8192 / 4096 / 2048 / 1024 slots per span: DarkCornell 9 995 / 10 160 / 10 255 / 10 200
 * Mrays/s with 64 pixels per wave; rounds 1-2, two pixels per wave, preferred 4096) */
static uint32_t lds_stream_span(const rpt_ctx *c, uint32_t &grid) {
    const uint32_t wgs = c->stream_max_blocks;
    uint32_t span = c->n_slots / (wgs * 32u);
    span = span < (uint32_t)LDS_THREADS ? (uint32_t)LDS_THREADS : (span > 8u * LDS_THREADS ? 8u * LDS_THREADS : span);
    span = (span + LDS_THREADS - 1) / LDS_THREADS * LDS_THREADS;
    const uint32_t n_spans = (c->n_slots + span - 1) / span;
    grid = n_spans < wgs ? n_spans : wgs;
    return span;
}

/* the streamed global-memory nearest-hit walk for the (stack capacity, entry width) pairs a scene can have */
template <int STACK, bool COOP>
static void launch_nearest_gstream(rpt_ctx *c, uint32_t iteration) {
    const int width = gstream_stack_width<STACK>(c);
    const uint32_t span = gstream_span(c, (uint32_t)gstream_rays_nearest(STACK, width)), blocks = (c->n_slots + span - 1) / span;
#define RPT_LAUNCH_NEAREST(W) k_traverse_nearest_gstream<STACK, W, COOP><<<blocks, RPT_WAVE, 0, c->stream>>>(c->scene, c->state, c->queues, iteration, span)
    if constexpr (STACK == 16) RPT_LAUNCH_NEAREST(16);
    else if constexpr (STACK == 24) { if (width == 16) RPT_LAUNCH_NEAREST(16); else if (width == 21) RPT_LAUNCH_NEAREST(21); else RPT_LAUNCH_NEAREST(24); }
    else { if (width == 16) RPT_LAUNCH_NEAREST(16); else if (width == 21) RPT_LAUNCH_NEAREST(21); else if (width == 24) RPT_LAUNCH_NEAREST(24); else RPT_LAUNCH_NEAREST(32); }
#undef RPT_LAUNCH_NEAREST
}

/* The nearest-hit traversal stage for the context's scene and state: which kernel, which grid.  Used by every iteration of a
 * render call and by rpt_debug_trace_rays_production (per-ray parity of exactly these kernels).
 *   a scene whose traversal image lives in LDS : the persistent streamed LDS walk (FIRST: camera rays; LAST: the last rays of a batch without NEE)
 *   a pair-shaped node pool (every pool the reference's builder makes)  : the streamed global-memory walk over pair records
 *   any other tree (a foreign builder's pool)  : the generic one-ray-per-lane walk over the uploaded nodes */
template <int STACK>
void launch_nearest(rpt_ctx *c, uint32_t iteration, bool last_without_nee = false /* the last extension rays of a batch of known length, no NEE */,
                    bool camera_rays = false /* iteration 0 of a render call: every ray leaves cfg.cam_position */) {
    hipStream_t s = c->stream;
    if (STACK == 16 && c->scene.lds_scene) {
        const size_t lds_bytes = (size_t)c->scene.lds_vecs * sizeof(float4);
        uint32_t grid;
        const uint32_t span = lds_stream_span(c, grid);
        const float *cam = c->cfg.c.cam_position;
        if (last_without_nee && c->scene.last_emit_n <= RPT_LAST_EMIT_MAX)
            k_traverse_nearest_stream<16, LDS_THREADS, RPT_NEAREST_LAST><<<grid, LDS_THREADS, lds_bytes + (size_t)c->scene.last_flip_vecs * sizeof(float4), s>>>(c->scene, c->state, c->queues, iteration, span, 0.0f, 0.0f, 0.0f);
        else if (camera_rays)
            k_traverse_nearest_stream<16, LDS_THREADS, RPT_NEAREST_FIRST><<<grid, LDS_THREADS, lds_bytes, s>>>(c->scene, c->state, c->queues, iteration, span, cam[0], cam[1], cam[2]);
        else k_traverse_nearest_stream<16, LDS_THREADS><<<grid, LDS_THREADS, lds_bytes, s>>>(c->scene, c->state, c->queues, iteration, span, 0.0f, 0.0f, 0.0f);
    } else if (c->scene.gpairs) {
        if (c->fat_leaves) launch_nearest_gstream<STACK, true>(c, iteration);
        else launch_nearest_gstream<STACK, false>(c, iteration);
    } else {
        const uint32_t nb = (c->n_slots + GLOBAL_THREADS - 1) / GLOBAL_THREADS;
        k_traverse_nearest<32, false, GLOBAL_THREADS><<<nb, GLOBAL_THREADS, 0, s>>>(c->scene, c->state, c->queues, iteration);
    }
}

/* the streamed global-memory any-hit walk (FIXED: over the flipped copy, left child first — shadow_order.h) */
template <int STACK, bool COOP>
static void launch_shadow_gstream(rpt_ctx *c, uint32_t q_positions) {
    const int width = gstream_stack_width<STACK>(c);
    const uint32_t span = gstream_span(c, (uint32_t)RPT_GSTREAM_RAYS), blocks = (q_positions + span - 1) / span;
#define RPT_LAUNCH_SHADOW(W)                                                                                                                                \
    do {                                                                                                                                                    \
        if (c->scene.gpairs_shadow) k_traverse_shadow_gstream<STACK, W, COOP, true><<<blocks, RPT_WAVE, 0, c->stream>>>(c->scene, c->state, c->queues, c->cfg, c->dev_stats.p, span); \
        else k_traverse_shadow_gstream<STACK, W, COOP, false><<<blocks, RPT_WAVE, 0, c->stream>>>(c->scene, c->state, c->queues, c->cfg, c->dev_stats.p, span);     \
    } while (0)
    if constexpr (STACK == 16) RPT_LAUNCH_SHADOW(16);
    else if constexpr (STACK == 24) { if (width == 16) RPT_LAUNCH_SHADOW(16); else if (width == 21) RPT_LAUNCH_SHADOW(21); else RPT_LAUNCH_SHADOW(24); }
    else { if (width == 16) RPT_LAUNCH_SHADOW(16); else if (width == 21) RPT_LAUNCH_SHADOW(21); else if (width == 24) RPT_LAUNCH_SHADOW(24); else RPT_LAUNCH_SHADOW(32); }
#undef RPT_LAUNCH_SHADOW
}


template <int STACK>
void launch_shadow(rpt_ctx *c) {
    hipStream_t s = c->stream;
    const size_t lds_bytes = (size_t)c->scene.lds_vecs * sizeof(float4);
    const uint32_t q_positions = c->n_slots + RPT_Q_SLACK, blocks_q = (q_positions + RPT_BLOCK - 1) / RPT_BLOCK;
    {
        if (STACK == 16 && c->scene.lds_scene) {
            uint32_t grid;
            const uint32_t span = lds_stream_span(c, grid);
            if (c->scene.lds_image_shadow) k_traverse_shadow_stream<16, LDS_THREADS, true><<<grid, LDS_THREADS, lds_bytes, s>>>(c->scene, c->state, c->queues, c->dev_stats.p, span);
            else k_traverse_shadow_stream<16, LDS_THREADS, false><<<grid, LDS_THREADS, lds_bytes, s>>>(c->scene, c->state, c->queues, c->dev_stats.p, span);
            k_shadow_resolve<<<blocks_q, RPT_BLOCK, 0, s>>>(c->state, c->queues, c->cfg);
        } else if (c->scene.gpairs) {
            if (c->fat_leaves) launch_shadow_gstream<STACK, true>(c, q_positions);
            else launch_shadow_gstream<STACK, false>(c, q_positions);
        } else {
            const uint32_t nb = (q_positions + GLOBAL_THREADS - 1) / GLOBAL_THREADS;
            k_traverse_shadow<32, false, GLOBAL_THREADS><<<nb, GLOBAL_THREADS, 0, s>>>(c->scene, c->state, c->queues, c->cfg, c->dev_stats.p);
        }
    }
}

}  // namespace

void rpt_launch_nearest(rpt_ctx *c, uint32_t iteration, bool last_without_nee, bool camera_rays) {
    switch (c->stack_cap) {
        case 16: launch_nearest<16>(c, iteration, last_without_nee, camera_rays); break;
        case 24: launch_nearest<24>(c, iteration, last_without_nee, camera_rays); break;
        default: launch_nearest<32>(c, iteration, last_without_nee, camera_rays); break;
    }
}

void rpt_launch_shadow(rpt_ctx *c) {
    switch (c->stack_cap) {
        case 16: launch_shadow<16>(c); break;
        case 24: launch_shadow<24>(c); break;
        default: launch_shadow<32>(c); break;
    }
}

hipError_t rpt_last_walk_attributes(hipFuncAttributes *out) {
    return hipFuncGetAttributes(out, reinterpret_cast<const void *>(&k_traverse_nearest_stream<16, RPT_LDS_THREADS, RPT_NEAREST_LAST>));
}

/* rpt_debug_trace_rays: plain ray arrays through the reference-order walk (traverse_one), out of LDS where the scene lives there */
void rpt_launch_trace_debug(rpt_ctx *c, bool any_hit, uint32_t n, const float *o, const float *d, const float *max_t, float *out_t, uint32_t *out_tri, uint32_t *out_flags) {
    hipStream_t s = c->stream;
    if (c->stack_cap == 16 && c->scene.lds_scene) {
        const unsigned bl = (n + LDS_THREADS - 1) / LDS_THREADS;
        const size_t lds_bytes = (size_t)c->scene.lds_vecs * sizeof(float4);
        if (any_hit) k_trace_debug<16, true, true, LDS_THREADS><<<bl, LDS_THREADS, lds_bytes, s>>>(c->scene, n, o, d, max_t, out_t, out_tri, out_flags);
        else k_trace_debug<16, false, true, LDS_THREADS><<<bl, LDS_THREADS, lds_bytes, s>>>(c->scene, n, o, d, max_t, out_t, out_tri, out_flags);
    } else {
        const unsigned blocks = (n + RPT_BLOCK - 1) / RPT_BLOCK;
        if (any_hit) k_trace_debug<32, true, false, RPT_BLOCK><<<blocks, RPT_BLOCK, 0, s>>>(c->scene, n, o, d, max_t, out_t, out_tri, out_flags);
        else k_trace_debug<32, false, false, RPT_BLOCK><<<blocks, RPT_BLOCK, 0, s>>>(c->scene, n, o, d, max_t, out_t, out_tri, out_flags);
    }
}
