/*
 * rpt_hip.hip — librpt_hip.so: context, upload, wavefront scheduling and the
 * C ABI of include/rpt/rpt.h (the drop-in replacement of the gpgpu-rs/wgpu
 * calls in the reference's trace_gpu, src/trace.rs:136-224).
 *
 * Wavefront iteration (all queues of slot ids, all state SoA, no host round
 * trip per sample):
 *     traverse_nearest -> shade -> [traverse_shadow] -> sky -> generate
 * The host only learns the size of the next extension queue through a lagged
 * asynchronous read-back (pinned ring + events), so the GPU never idles on it.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rpt_ctx.h"
#include "shadow_order.h"

#include "rpt_fastdiv.h"      /* (the walk kernels of k_traverse.h are compiled in rpt_traverse.hip) */
#include "k_shade.h"
#include "k_complete.h"
#include "k_sky_generate.h"
#include "k_bsdf_extra.h"

namespace {

constexpr int LAG = RPT_RING_LAG;
constexpr int RING = RPT_RING;

}  // namespace

/* (read afresh by every rpt_create / scene-preparation call: a test process changes its environment between contexts) */
rpt_knobs rpt_read_knobs() {
    {
        rpt_knobs k;
        auto num = [](const char *name, int lo, int hi, int otherwise) { const char *e = getenv(name); if (!e || !e[0]) return otherwise; const int v = atoi(e); return v < lo ? lo : (v > hi ? hi : v); };
        auto word = [](const char *name) { const char *e = getenv(name); return std::string(e ? e : ""); };
        k.stage_timing = num("RPT_STAGE_TIMING", 0, 2, 0);
        k.upload_timing = num("RPT_UPLOAD_TIMING", 0, 1, 0) == 1;
        k.slot_q_shift = num("RPT_SLOT_Q_SHIFT", 0, 5, -1);
        const std::string so = word("RPT_SHADOW_ORDER"), lo = word("RPT_LAST_ORDER");
        k.shadow_order = so == "near" ? 0 : (so == "fixed" ? 1 : -1);
        k.last_order = lo == "off" ? 4 : (lo == "near" ? 0 : (lo == "opaque" ? 1 : (lo == "small" ? 2 : (lo == "ratio" ? 3 : -1))));
        k.shade_compact = num("RPT_SHADE_COMPACT", 0, 1, -1);
        k.sky_strided = num("RPT_SKY_STRIDED", 0, 1 << 20, -1);
        k.stack_bits = num("RPT_STACK_BITS", 16, 32, 16);
        k.coop_leaves = num("RPT_COOP_LEAVES", 0, 1, -1);
        k.no_lds_scene = num("RPT_NO_LDS_SCENE", 0, 1, 0) == 1;
        k.bvh_team_min = num("RPT_BVH_TEAM_MIN", 2, 1 << 30, 0);
        return k;
    }
}

thread_local std::string g_create_error;
std::string &rpt_create_error() { return g_create_error; }

namespace {

/* rank-local slot order: tiles in ascending id (tile t -> rank t mod world),
 * inside a tile 8x8 pixel blocks row-major, inside a block row-major; pixels
 * outside the image are skipped.  One wave = one 8x8 block on full tiles, so
 * primary rays of a wave are coherent. */
void build_pixel_order(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, std::vector<uint32_t> &out) {
    const uint32_t T = RPT_TILE, B = 8;
    uint32_t tiles_x = (W + T - 1) / T, tiles_y = (H + T - 1) / T;
    /* (count first, then plain stores: four million push_backs were 10 ms of rpt_set_config at 2048^2) */
    size_t total = 0;
    for (uint32_t t = rank; t < tiles_x * tiles_y; t += world) {
        const uint32_t tx = (t % tiles_x) * T, ty = (t / tiles_x) * T;
        total += (size_t)std::min(T, W - tx) * std::min(T, H - ty);
    }
    out.resize(total);
    uint32_t *at = out.data();
    for (uint32_t t = rank; t < tiles_x * tiles_y; t += world) {
        uint32_t tx = (t % tiles_x) * T, ty = (t / tiles_x) * T;
        for (uint32_t by = 0; by < T && ty + by < H; by += B)
            for (uint32_t bx = 0; bx < T && tx + bx < W; bx += B) {
                const uint32_t xs = std::min(B, W - (tx + bx)), ys = std::min(B, H - (ty + by));
                for (uint32_t y = 0; y < ys; ++y)
                    for (uint32_t x = 0; x < xs; ++x) *at++ = (tx + bx + x) | ((ty + by + y) << 16);
            }
    }
}

}  // namespace
void rpt_build_pixel_order(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, std::vector<uint32_t> &out) { build_pixel_order(W, H, rank, world, out); }
namespace {

/* what needs no walk: sizes, and every index of the triangle and light-pick buffers in range */
int validate_scene_flat(rpt_ctx *ctx, const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt, size_t nn, const rpt_material_data *mats, size_t nm,
                        const rpt_light_pick_entry *lp, size_t nlp) {
    (void)pv; (void)mats;
    if (!nv || !nt || !nn || !nm || !nlp) { ctx->error = "empty scene buffer"; return RPT_ESCENE; }
    if (nt >= 0x7ffffff0ull || nn >= 0x7ffffff0ull) { ctx->error = "scene too large for 31-bit indices"; return RPT_ESCENE; }
    for (size_t i = 0; i < nt; ++i)
        if (idx[i].v0 >= nv || idx[i].v1 >= nv || idx[i].v2 >= nv || idx[i].material >= nm) {
            ctx->error = "index buffer entry out of range";
            return RPT_ESCENE;
        }
    bool sentinel = lp[0].ratio < 0.0f;
    if (!sentinel)
        for (size_t i = 0; i < nlp; ++i)
            if (lp[i].triangle_index_a >= nt || lp[i].triangle_index_b >= nt) {
                ctx->error = "light pick entry out of range";
                return RPT_ESCENE;
            }
    return RPT_OK;
}

/* The whole validation on the host (the debug hooks, which have no device): the flat part + the node pool as a TREE — children in range, no node reached twice
 * (a cycle, or a subtree with two parents), leaf ranges inside the index buffer, depth bounded.  rpt_upload_scene checks the tree on the device (device_validate_tree:
 * the same four conditions, level by level; the DFS over the 2 M nodes of the scattered stand-in was 16 - 20 ms of its upload). */
int validate_scene(rpt_ctx *ctx, const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt,
                   const rpt_bvh_node *nodes, size_t nn, const rpt_material_data *mats, size_t nm,
                   const rpt_light_pick_entry *lp, size_t nlp, uint32_t &max_depth) {
    int rc = validate_scene_flat(ctx, pv, nv, idx, nt, nn, mats, nm, lp, nlp);
    if (rc) return rc;
    std::vector<std::pair<uint32_t, uint32_t>> stack{{0u, 0u}};
    std::vector<bool> seen(nn, false);
    max_depth = 0;
    while (!stack.empty()) {
        auto [n, d] = stack.back();
        stack.pop_back();
        if (seen[n]) { ctx->error = "BVH is not a tree"; return RPT_ESCENE; }
        seen[n] = true;
        if (d > max_depth) max_depth = d;
        const rpt_bvh_node &node = nodes[n];
        if (node.triangle_count > 0) {
            if ((size_t)node.left_or_first + node.triangle_count > nt) { ctx->error = "BVH leaf range out of bounds"; return RPT_ESCENE; }
        } else {
            if ((size_t)node.left_or_first + 1 >= nn) { ctx->error = "BVH child index out of bounds"; return RPT_ESCENE; }
            stack.push_back({node.left_or_first, d + 1});
            stack.push_back({node.left_or_first + 1, d + 1});
        }
    }
    if (max_depth > 31) {   /* reference: FixedVec<usize, 32> would overflow (intersection.rs:178) */
        ctx->error = "BVH deeper than the reference's 32-entry traversal stack";
        return RPT_ESCENE;
    }
    return RPT_OK;
}

void rotation_y(float angle, float *m) {   /* Mat3::from_rotation_y, column-major */
    float s, c;
    rptm::sincosr(angle, s, c);
    m[0] = c; m[1] = 0.0f; m[2] = -s;
    m[3] = 0.0f; m[4] = 1.0f; m[5] = 0.0f;
    m[6] = s; m[7] = 0.0f; m[8] = c;
}
void rotation_x(float angle, float *m) {
    float s, c;
    rptm::sincosr(angle, s, c);
    m[0] = 1.0f; m[1] = 0.0f; m[2] = 0.0f;
    m[3] = 0.0f; m[4] = c; m[5] = s;
    m[6] = 0.0f; m[7] = -s; m[8] = c;
}
void mat3_mul_host(const float *a, const float *b, float *out) {   /* Mat3 * Mat3 = cols a*b.col */
    for (int c = 0; c < 3; ++c) {
        float v[3] = {b[3 * c], b[3 * c + 1], b[3 * c + 2]};
        for (int r = 0; r < 3; ++r) {
            float acc = a[r] * v[0];
            acc = acc + a[3 + r] * v[1];
            acc = acc + a[6 + r] * v[2];
            out[3 * c + r] = acc;
        }
    }
}

void release_state(rpt_ctx *c) {
    c->ray_a.release(); c->ray_b.release(); c->hit.release(); c->thr.release(); c->rad.release();
    c->mis_a.release(); c->mis_b.release();
    c->accum.release(); c->rng.release();
    c->q_sky.release(); c->q_count.release(); c->ray_shards.release();
    c->sh_o.release(); c->sh_d.release(); c->sh_c.release();
    c->pixel_xy.release();
    rpt_image_release(c);
    c->has_state = false;
}

/* Stream discipline: the context's stream is NON-BLOCKING, so nothing on the legacy default stream is ordered against it.
 * hipMemset of device memory may return before it has run; issued on the default stream it raced the first kernels of the
 * next render whenever other threads kept that stream busy with their own contexts (lost ray counts in
 * test_contexts_on_different_threads).  Hence: every fill is hipMemsetAsync ON THE CONTEXT'S STREAM; host-to-device copies
 * stay synchronous hipMemcpy (complete when they return) and are only issued after the stream has been drained. */
/* What rpt_set_config allocates: the per-PIXEL state (accumulators, rng, pixel coordinates) and the counters.  The per-SLOT path state is
 * allocated by the first rpt_render that needs it, at the size it needs (ensure_slot_state): a start-up — the reference's
 * trace_gpu(scene, 0 samples), benches/benchmark.rs:11-13 — allocates and touches no path state at all, a configuration without NEE no
 * shadow queue (48 bytes per slot) and no MIS carry (32), and the ceiling of samples in flight can be 256 per pixel without a context
 * that renders 32-sample batches paying for it (rounds 1-5 allocated up to 160 M slots x 180 bytes = 29 GB in rpt_set_config). */
int alloc_pixel_state(rpt_ctx *c) {
    const size_t np = c->n_pixels;
    HIP_TRY(c, c->accum.alloc(np)); HIP_TRY(c, c->rng.alloc(np));
    HIP_TRY(c, c->ray_shards.alloc(RPT_STAT_SHARDS * RPT_STAT_STRIDE));
    HIP_TRY(c, hipMemsetAsync(c->ray_shards.p, 0, RPT_STAT_SHARDS * RPT_STAT_STRIDE * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, c->q_count.alloc(Q_WORDS));
    HIP_TRY(c, c->pixel_xy.alloc(np));
    if (np) HIP_TRY(c, hipMemcpy(c->pixel_xy.p, c->pixel_xy_host.data(), np * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemsetAsync(c->q_count.p, 0, Q_WORDS * sizeof(uint32_t), c->stream));
    DevState &s = c->state;
    s = DevState{};
    s.rng = c->rng.p; s.accum = c->accum.p; s.pixel_xy = c->pixel_xy.p; s.n_slots = c->n_slots;
    s.n_pixels = (uint32_t)np; s.group_shift = c->group_shift; s.q_shift = 0;
    DevQueues &q = c->queues;
    q = DevQueues{};
    q.ray_shards = c->ray_shards.p; q.count = c->q_count.p;
    q.sky_cnt = c->q_count.p + Q_COUNT; q.shadow_cnt = q.sky_cnt + RPT_Q_SHARDS * RPT_Q_SHARD_STRIDE;
    q.host_ring = c->host_ring_dev; q.ring_mask = RING - 1;
    /* up to this many queued misses the sky march runs 16 lanes per miss (re-clamped per call to that call's slot count) */
    q.sky_wide_limit = (uint32_t)std::min<size_t>(c->n_slots / 16, c->sky_wide_cfg);
    /* 1 = shade misses in the iteration that found them.  (Letting them pile up removed most of the near-empty sky launches on closed scenes, but the
     * parked pixels finish later and lengthen the tail: DarkCornell 3650 Mrays/s deferred vs 3928 eager in round 1; the knob went in round 6.) */
    q.sky_threshold = 1u; q.sky_at_end = 0u; q.known_length = 0u;
    c->has_state = true;
    return RPT_OK;
}

/* The per-slot arrays, for a render call over `n` slots: grown (never shrunk) to what the call needs — the path state always, the shadow
 * queue when the configuration has NEE, the MIS carry when it is MIS.  Growing waits for whatever is in flight, frees the old arrays first
 * and leaves every slot idle; between render calls every slot IS idle, so nothing is carried over. */
int ensure_slot_state(rpt_ctx *c, size_t n, bool need_shadow, bool need_mis) {
    const bool grow = c->hit.n < n, grow_shadow = need_shadow && c->sh_o.n < n + RPT_Q_SLACK, grow_mis = need_mis && c->mis_a.n < n;
    if (grow || grow_shadow || grow_mis) {
        SectionTimer sections("path state");
        if (c->async_pending) { int rc = rpt_wait(c); if (rc) return rc; }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        sections.mark("drain");
        if (grow) {
            c->ray_a.release(); c->ray_b.release(); c->hit.release(); c->thr.release(); c->rad.release(); c->q_sky.release();
            HIP_TRY(c, c->ray_a.alloc(n)); HIP_TRY(c, c->ray_b.alloc(n)); HIP_TRY(c, c->hit.alloc(n));
            HIP_TRY(c, c->thr.alloc(n)); HIP_TRY(c, c->rad.alloc(n));
            HIP_TRY(c, c->q_sky.alloc(n + RPT_Q_SLACK));     /* side queues: positions, not entries (k_common.h: sharded queues) */
            sections.mark("alloc_path_state");
            k_fill_idle<<<(unsigned)((n + RPT_BLOCK - 1) / RPT_BLOCK), RPT_BLOCK, 0, c->stream>>>(c->hit.p, (uint32_t)n);   /* nothing in flight */
            HIP_TRY(c, hipGetLastError());
        }
        if (grow_shadow) {
            c->sh_o.release(); c->sh_d.release(); c->sh_c.release();
            HIP_TRY(c, c->sh_o.alloc(n + RPT_Q_SLACK)); HIP_TRY(c, c->sh_d.alloc(n + RPT_Q_SLACK)); HIP_TRY(c, c->sh_c.alloc(n + RPT_Q_SLACK));
        }
        if (grow_mis) {
            c->mis_a.release(); c->mis_b.release();
            HIP_TRY(c, c->mis_a.alloc(n)); HIP_TRY(c, c->mis_b.alloc(n));
        }
        sections.mark("alloc_queues_carries");
    }
    DevState &s = c->state;
    s.ray_a = c->ray_a.p; s.ray_b = c->ray_b.p; s.hit = c->hit.p; s.thr = c->thr.p; s.rad = c->rad.p;
    s.mis_a = c->mis_a.p; s.mis_b = c->mis_b.p;
    DevQueues &q = c->queues;
    q.sky = c->q_sky.p; q.sh_o = c->sh_o.p; q.sh_d = c->sh_d.p; q.sh_c = c->sh_c.p;
    return RPT_OK;
}

/* The LDS-resident traversal image of a small scene (layout and rationale: k_traverse.h, SceneViewLds):
 *   float4 K_A[P], K_B[P] for K = x, y, z   (L.lo, R.lo, L.hi, R.hi) and (L.hi, R.hi, L.lo, R.lo)
 *   u32    D[P] (padded to 16 bytes)        desc(L) | desc(R) << 16
 *   float4 a[T], e1[T], e2[T]
 * with pair p = nodes (2p+1, 2p+2).  Returns false when the node array cannot be represented (not pair-shaped,
 * a box with lo > hi or a NaN bound, a leaf of 64+ triangles, 512+ triangles): such a scene is traversed from
 * global memory by the generic loop. */
bool build_lds_image(const rpt_bvh_node *nodes, size_t nn, const std::vector<float4> &geom, size_t nt,
                     std::vector<float4> &image, uint32_t &pairs, uint32_t &root) {
    if (nn == 0 || (nn & 1u) == 0u || nt > 512 || nn >= 2 * (size_t)LDS_DESC_DEAD) return false;
    auto desc = [&](const rpt_bvh_node &n, uint32_t &out) {
        if (n.triangle_count != 0u) {
            if (n.triangle_count >= 64u || n.left_or_first >= 512u || (size_t)n.left_or_first + n.triangle_count > nt) return false;
            out = LDS_DESC_LEAF | (n.triangle_count << 9) | n.left_or_first;
            return true;
        }
        uint32_t l = n.left_or_first;
        if ((l & 1u) == 0u || (size_t)l + 1 >= nn) return false;
        out = l >> 1;
        return true;
    };
    if (!desc(nodes[0], root)) return false;
    for (size_t i = 0; i < nn; ++i)
        for (int k = 0; k < 3; ++k)
            if (!(nodes[i].aabb_min[k] <= nodes[i].aabb_max[k])) return false;
    const size_t P = (nn - 1) / 2, desc_vecs = (P + 3) / 4;
    pairs = (uint32_t)P;
    image.assign(6 * P + desc_vecs + 3 * nt, make_float4(0, 0, 0, 0));
    uint32_t *descs = reinterpret_cast<uint32_t *>(image.data() + 6 * P);
    for (size_t p = 0; p < P; ++p) {
        const rpt_bvh_node &L = nodes[2 * p + 1], &R = nodes[2 * p + 2];
        uint32_t dl, dr;
        if (!desc(L, dl) || !desc(R, dr)) return false;
        descs[p] = dl | (dr << 16);
        for (int k = 0; k < 3; ++k) {
            image[(2 * k) * P + p] = make_float4(L.aabb_min[k], R.aabb_min[k], L.aabb_max[k], R.aabb_max[k]);
            image[(2 * k + 1) * P + p] = make_float4(L.aabb_max[k], R.aabb_max[k], L.aabb_min[k], R.aabb_min[k]);
        }
    }
    for (size_t t = 0; t < nt; ++t)
        for (int j = 0; j < 3; ++j) image[6 * P + desc_vecs + (size_t)j * nt + t] = geom[3 * t + j];
    return true;
}

static uint32_t padded_pixels(uint32_t n_pixels) { return (n_pixels + 63u) & ~63u; }      /* whole chunks of 64 pixels (k_common.h, slot_pix) */

/* one wave per chunk of 64 pixels (k_complete.h) */
static void launch_complete(rpt_ctx *c, uint32_t iteration, uint32_t final_pass) {
    k_complete<<<padded_pixels(c->n_pixels) / RPT_WAVE, RPT_WAVE, complete_lds_bytes(1u << c->group_shift, c->state.q_shift), c->stream>>>(c->state, c->queues, c->cfg, iteration, final_pass,
                                                                                                                     c->dev_stats.p);
}

template <int STACK, int NEE, bool TEXTURED>
void launch_iteration(rpt_ctx *c, uint32_t iteration, uint32_t blocks, std::vector<hipEvent_t> *ev, size_t &ev_at, bool complete_each, bool sky_now) {
    hipStream_t s = c->stream;
    const bool only_traverse = c->timing_level == 2;
    auto mark = [&](bool traverse_edge = false) {
        if (ev && (!only_traverse || traverse_edge)) (void)hipEventRecord((*ev)[ev_at++], s);
    };
    if (only_traverse) mark(true);
    /* the consumers of a side queue cover its POSITIONS: up to RPT_Q_SLACK more than there are slots (k_common.h) */
    const uint32_t q_positions = c->n_slots + RPT_Q_SLACK, blocks_q = (q_positions + RPT_BLOCK - 1) / RPT_BLOCK;
    /* (the shade stage's last_iteration, k_shade.h: in a batch of known length iteration k is bounce k of every path) */
    rpt_launch_nearest(c, iteration, NEE == RPT_NEE_NONE && c->queues.known_length != 0u && iteration != 0u && iteration + 1u >= c->cfg.c.max_bounces,
                       iteration == 0u);
    mark(true);
    if (c->shade_compact) k_shade<NEE, TEXTURED, true><<<(c->n_slots + RPT_BLOCK * RPT_SHADE_ROUNDS - 1) / (RPT_BLOCK * RPT_SHADE_ROUNDS), RPT_BLOCK, 0, s>>>(c->scene, c->state, c->queues, c->cfg, iteration, c->dev_stats.p, c->call_samples);
    else k_shade<NEE, TEXTURED, false><<<blocks, RPT_BLOCK, 0, s>>>(c->scene, c->state, c->queues, c->cfg, iteration, c->dev_stats.p, c->call_samples);
    /* generations are completed (and the next samples started) after every shade stage only where slots take more than one
     * sample in this call; a batch of known length completes them once, after its last iteration (render_impl) */
    if (complete_each) launch_complete(c, iteration, 0u);
    mark();
    if (NEE != RPT_NEE_NONE) rpt_launch_shadow(c);          /* the any-hit walk of the queued shadow rays (+ k_shadow_resolve behind the streamed LDS walk) */
    mark();
    if (c->queues.sky_at_end == 0u || sky_now) {
        if (c->sky_strided && blocks > c->sky_blocks) k_sky<true><<<c->sky_blocks, RPT_BLOCK, 0, s>>>(c->scene, c->state, c->queues, c->cfg, iteration, c->dev_stats.p);
        else k_sky<false><<<blocks_q, RPT_BLOCK, 0, s>>>(c->scene, c->state, c->queues, c->cfg, iteration, c->dev_stats.p);
    }
    mark();
}

template <int STACK>
void launch_iteration_stack(rpt_ctx *c, uint32_t iteration, uint32_t blocks, std::vector<hipEvent_t> *ev, size_t &ev_at, bool complete_each, bool sky_now) {
    const bool tex = c->scene.textured != 0u;
    switch (c->cfg.nee_mode) {
        case RPT_NEE_MIS:
            if (tex) launch_iteration<STACK, RPT_NEE_MIS, true>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            else launch_iteration<STACK, RPT_NEE_MIS, false>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            break;
        case RPT_NEE_DIRECT:
            if (tex) launch_iteration<STACK, RPT_NEE_DIRECT, true>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            else launch_iteration<STACK, RPT_NEE_DIRECT, false>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            break;
        default:
            if (tex) launch_iteration<STACK, RPT_NEE_NONE, true>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            else launch_iteration<STACK, RPT_NEE_NONE, false>(c, iteration, blocks, ev, ev_at, complete_each, sky_now);
            break;
    }
}

constexpr int EVENTS_PER_ITER = 4;   /* timing level 1: after traverse, shade, shadow, sky; per call two leading events (before and
                                        after k_generate_first) and one after the batch's k_complete;
                                        level 2: before and after the traversal kernel only */
constexpr int EVENTS_LEAD = 2, EVENTS_TAIL = 1;

static size_t timing_events_needed(const rpt_ctx *c, uint64_t iterations) {
    return c->timing_level == 2 ? (size_t)iterations * 2 : EVENTS_LEAD + (size_t)iterations * EVENTS_PER_ITER + EVENTS_TAIL;
}
static void timing_accumulate(rpt_ctx *c, const std::vector<hipEvent_t> &ev, uint64_t iterations, bool complete_timed) {
    float ms;
    if (c->timing_level == 2) {
        for (uint64_t k = 0; k < iterations; ++k)
            if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) c->stats.kernel_ms[RPT_STAGE_TRAVERSE] += ms;
        return;
    }
    if (hipEventElapsedTime(&ms, ev[0], ev[1]) == hipSuccess) c->stats.kernel_ms[RPT_STAGE_GENERATE] += ms;
    const int stage_of[EVENTS_PER_ITER] = {RPT_STAGE_TRAVERSE, RPT_STAGE_SHADE, RPT_STAGE_SHADOW, RPT_STAGE_SKY};
    size_t at = EVENTS_LEAD;
    for (uint64_t k = 0; k < iterations; ++k)
        for (int e = 0; e < EVENTS_PER_ITER; ++e) {
            if (hipEventElapsedTime(&ms, ev[at - 1], ev[at]) == hipSuccess) c->stats.kernel_ms[stage_of[e]] += ms;
            at += 1;
        }
    if (complete_timed && hipEventElapsedTime(&ms, ev[at - 1], ev[at]) == hipSuccess) c->stats.kernel_ms[RPT_STAGE_COMPLETE] += ms;
}

}  // namespace

/* rpt_reset: the caller's row-major seeds (and accumulators, when a render resumes) into the rank's tile-major pixel order */
__global__ __launch_bounds__(RPT_BLOCK) void k_reset_gather(const uint32_t *pixel_xy, uint32_t n_pixels, uint32_t width, const uint2 *seed, const float4 *accum_init /* nullable */,
                                                            uint2 *rng, float4 *accum) {
    const uint32_t s = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (s >= n_pixels) return;
    const uint32_t pxy = pixel_xy[s];
    const size_t i = (size_t)(pxy >> 16) * width + (pxy & 0xffffu);
    rng[s] = seed[i];
    accum[s] = accum_init ? accum_init[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

/* The 64-byte pair records + per-node links of the streamed global-memory walks (k_traverse.h SceneViewPairsT) from the uploaded node pool; with `flip`
 * (shadow_order.h) the two nodes of a flipped pair exchange slots: the copy the fixed-order shadow walks read.  (On the host this loop took 27 ms for 2 M nodes.) */
__global__ __launch_bounds__(RPT_BLOCK) void k_build_pairs(const float4 *nodes, const uint8_t *flip, uint32_t n_pairs, float4 *pairs, uint32_t *links) {
    const uint32_t p = blockIdx.x * RPT_BLOCK + threadIdx.x;
    auto link_of = [](float4 lo, float4 hi) { return (__float_as_uint(lo.w) << 24) | __float_as_uint(hi.w); };      /* triangle_count << 24 | left child / first triangle */
    if (p == 0u) links[0] = link_of(nodes[0], nodes[1]);
    if (p >= n_pairs) return;
    const bool f = flip != nullptr && flip[p] != 0;
    const uint32_t l = f ? 2u * p + 2u : 2u * p + 1u, r = f ? 2u * p + 1u : 2u * p + 2u;
    const float4 llo = nodes[2u * (size_t)l], lhi = nodes[2u * (size_t)l + 1u], rlo = nodes[2u * (size_t)r], rhi = nodes[2u * (size_t)r + 1u];
    const uint32_t kl = link_of(llo, lhi), kr = link_of(rlo, rhi);
    pairs[4u * (size_t)p + 0u] = make_float4(llo.x, llo.y, llo.z, lhi.x);
    pairs[4u * (size_t)p + 1u] = make_float4(lhi.y, lhi.z, rlo.x, rlo.y);
    pairs[4u * (size_t)p + 2u] = make_float4(rlo.z, rhi.x, rhi.y, rhi.z);
    pairs[4u * (size_t)p + 3u] = make_float4(0.0f, 0.0f, __uint_as_float(kl), __uint_as_float(kr));
    links[2u * p + 1u] = kl;
    links[2u * p + 2u] = kr;
}

/* tri_geom / tri_isect / tri_shade of every triangle (DevScene, k_common.h) and |e1 x e2|^2 for the shadow-order probe, from the uploaded reference buffers:
 *   tri_geom : a, e1 = b - a, e2 = c - a (muller_trumbore, intersection.rs:13-14; barycentric v0, v1, util.rs:239-240)
 *              with d00 = e1.e1, d01 = e1.e2, d11 = e2.e2 (util.rs:242-244) in the .w lanes — dot = (x x' + y y') + z z', as glam's
 *   tri_isect: e1, e2, a packed in 36 bytes        tri_shade: the three vertex normals, the three uv0 pairs and the material index in 64 bytes */
__global__ __launch_bounds__(RPT_BLOCK) void k_derive_triangles(const float4 *per_vertex, const uint4 *indices, uint32_t nt, float4 *tri_geom, float *tri_isect,
                                                                float4 *tri_shade, float4 *tri_tangent /* nullable */, float *cross_sq) {
    const uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (i >= nt) return;
    const uint4 t = indices[i];
    const float4 *A = per_vertex + 4u * (size_t)t.x, *B = per_vertex + 4u * (size_t)t.y, *C = per_vertex + 4u * (size_t)t.z;
    const float4 a = A[0], b = B[0], cc = C[0];
    const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const float e2x = cc.x - a.x, e2y = cc.y - a.y, e2z = cc.z - a.z;
    tri_geom[3u * (size_t)i + 0u] = make_float4(a.x, a.y, a.z, (e1x * e1x + e1y * e1y) + e1z * e1z);
    tri_geom[3u * (size_t)i + 1u] = make_float4(e1x, e1y, e1z, (e1x * e2x + e1y * e2y) + e1z * e2z);
    tri_geom[3u * (size_t)i + 2u] = make_float4(e2x, e2y, e2z, (e2x * e2x + e2y * e2y) + e2z * e2z);
    float *p = tri_isect + 9u * (size_t)i;
    p[0] = e1x; p[1] = e1y; p[2] = e1z; p[3] = e2x; p[4] = e2y; p[5] = e2z; p[6] = a.x; p[7] = a.y; p[8] = a.z;
    const float4 na = A[1], nb = B[1], nc = C[1], ua = A[3], ub = B[3], uc = C[3];
    tri_shade[4u * (size_t)i + 0u] = make_float4(na.x, na.y, na.z, ua.x);
    tri_shade[4u * (size_t)i + 1u] = make_float4(nb.x, nb.y, nb.z, ua.y);
    tri_shade[4u * (size_t)i + 2u] = make_float4(nc.x, nc.y, nc.z, __uint_as_float(t.w));
    tri_shade[4u * (size_t)i + 3u] = make_float4(ub.x, ub.y, uc.x, uc.y);
    if (tri_tangent) { tri_tangent[3u * (size_t)i + 0u] = A[2]; tri_tangent[3u * (size_t)i + 1u] = B[2]; tri_tangent[3u * (size_t)i + 2u] = C[2]; }
    const float cx = e1y * e2z - e1z * e2y, cy = e1z * e2x - e1x * e2z, cz = e1x * e2y - e1y * e2x;       /* (the probe's estimate of areas: no part of a result) */
    cross_sq[i] = (cx * cx + cy * cy) + cz * cz;
}

/* ---- the node pool as a tree, checked on the device (rpt_upload_scene) ------------------------------------------------------------------------------------------- */
struct NodeFacts {
    uint32_t error;        /* 1 child index out of bounds, 2 leaf range out of bounds, 4 a node reached twice, 8 deeper than 31 levels */
    uint32_t max_depth;
    uint32_t flags;        /* over ALL nodes of the pool: 1 a leaf of more than RPT_COOP_LEAF_MIN triangles, 2 a node the pair records cannot express, 4 a bound outside the exact-division guard */
};
constexpr uint32_t DEPTH_UNSET = 0xffffffffu;
/* pass p: the nodes at depth p claim their children for depth p + 1 (a child somebody already claimed: not a tree) */
__global__ __launch_bounds__(RPT_BLOCK) void k_validate_pass(const rpt_bvh_node *nodes, uint32_t nn, uint32_t nt, uint32_t *depth_of, uint32_t pass, NodeFacts *facts) {
    const uint32_t n = blockIdx.x * RPT_BLOCK + threadIdx.x;
    uint32_t err = 0u;
    const bool mine = n < nn && depth_of[n] == pass;
    if (mine) {
        const rpt_bvh_node node = nodes[n];
        if (pass > 31u) err = 8u;                  /* reference: FixedVec<usize, 32> would overflow (intersection.rs:178) */
        else if (node.triangle_count != 0u) { if ((size_t)node.left_or_first + node.triangle_count > nt) err = 2u; }
        else if ((size_t)node.left_or_first + 1 >= nn) err = 1u;
        else {
            if (atomicCAS(&depth_of[node.left_or_first], DEPTH_UNSET, pass + 1u) != DEPTH_UNSET) err = 4u;
            if (atomicCAS(&depth_of[node.left_or_first + 1u], DEPTH_UNSET, pass + 1u) != DEPTH_UNSET) err = 4u;
        }
    }
    const unsigned long long any = rpt_ballot(mine);
    if (any != 0ull && __lane_id() == (uint32_t)__ffsll((long long)any) - 1u && pass < 32u) facts->max_depth = pass;      /* (every writer of a launch stores the same value) */
    if (err != 0u) atomicOr(&facts->error, err);
}
__global__ __launch_bounds__(RPT_BLOCK) void k_node_flags(const rpt_bvh_node *nodes, uint32_t nn, NodeFacts *facts) {
    const uint32_t n = blockIdx.x * RPT_BLOCK + threadIdx.x;
    uint32_t bits = 0u;
    if (n < nn) {
        const rpt_bvh_node node = nodes[n];
        if (node.triangle_count > (uint32_t)RPT_COOP_LEAF_MIN) bits |= 1u;
        if (node.triangle_count >= 255u || node.left_or_first >= (1u << 24) || (node.triangle_count == 0u && ((node.left_or_first & 1u) == 0u || (size_t)node.left_or_first + 1 >= nn))) bits |= 2u;
        for (int k = 0; k < 3; ++k)
            if (!rptm::fastdiv_operand_ok(node.aabb_min[k]) || !rptm::fastdiv_operand_ok(node.aabb_max[k])) bits |= 4u;
    }
    uint32_t wave_bits = 0u;
    for (uint32_t b = 1u; b <= 4u; b <<= 1) if (rpt_ballot((bits & b) != 0u) != 0ull) wave_bits |= b;
    if (wave_bits != 0u && __lane_id() == 0u) atomicOr(&facts->flags, wave_bits);
}
/* nodes already on the device (not yet the context's); on RPT_OK `out` holds depth and flags */
static int device_validate_tree(rpt_ctx *c, const rpt_bvh_node *d_nodes, size_t nn, size_t nt, NodeFacts &out) {
    DevBuf<uint32_t> depth_of;
    DevBuf<NodeFacts> facts;
    hipError_t e = depth_of.alloc(nn);
    if (e == hipSuccess) e = facts.alloc(1);
    if (e == hipSuccess) e = hipMemsetAsync(depth_of.p, 0xff, nn * sizeof(uint32_t), nullptr);
    if (e == hipSuccess) e = hipMemsetAsync(depth_of.p, 0, sizeof(uint32_t), nullptr);              /* the root: depth 0 */
    if (e == hipSuccess) e = hipMemsetAsync(facts.p, 0, sizeof(NodeFacts), nullptr);
    if (e == hipSuccess) {
        const unsigned blocks = (unsigned)((nn + RPT_BLOCK - 1) / RPT_BLOCK);
        for (uint32_t pass = 0; pass <= 32u; ++pass) k_validate_pass<<<blocks, RPT_BLOCK>>>(d_nodes, (uint32_t)nn, (uint32_t)nt, depth_of.p, pass, facts.p);
        k_node_flags<<<blocks, RPT_BLOCK>>>(d_nodes, (uint32_t)nn, facts.p);
        e = hipMemcpy(&out, facts.p, sizeof(out), hipMemcpyDeviceToHost);
    }
    if (e == hipSuccess) e = hipGetLastError();
    depth_of.release(); facts.release();
    HIP_TRY(c, e);
    if (out.error & 1u) { c->error = "BVH child index out of bounds"; return RPT_ESCENE; }
    if (out.error & 2u) { c->error = "BVH leaf range out of bounds"; return RPT_ESCENE; }
    if (out.error & 4u) { c->error = "BVH is not a tree"; return RPT_ESCENE; }
    if (out.error & 8u) { c->error = "BVH deeper than the reference's 32-entry traversal stack"; return RPT_ESCENE; }
    return RPT_OK;
}

/* what the pair records of the streamed global-memory walks (k_traverse.h SceneViewPairsT) and the flipped copies can express: children of every inner node
 * are the nodes (2p + 1, 2p + 2) of one pair — every pool the reference's builder makes (src/bvh.rs:296-320) —, leaves of fewer than 255 triangles, links in 24 bits */
static bool pool_is_pair_shaped(const rpt_bvh_node *nodes, size_t nn) {
    if ((nn & 1u) != 1u || nn < 3 || nodes[0].triangle_count != 0u) return false;
    for (size_t i = 0; i < nn; ++i) {
        const rpt_bvh_node &n = nodes[i];
        if (n.triangle_count >= 255u || n.left_or_first >= (1u << 24)) return false;
        if (n.triangle_count == 0u && ((n.left_or_first & 1u) == 0u || (size_t)n.left_or_first + 1 >= nn)) return false;
    }
    return true;
}

/* ---- the order probes as kernels (shadow_order.h: the core is shared with the host driver) ---------------------------------------------------------------- */
namespace order_probe {

constexpr uint32_t LEVEL_UNSET = 0xffffffffu;
struct DevStack {
    uint32_t *column;                                   /* LDS [entry][lane] */
    __device__ __forceinline__ uint32_t &operator()(int k) const { return column[k * RPT_WAVE]; }
};

__global__ __launch_bounds__(RPT_BLOCK) void k_probe_tri_area(const float *cross_sq, uint32_t nt, double *tri_area) {
    const uint32_t t = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (t < nt) tri_area[t] = area_of_cross_sq(cross_sq[t]);
}
/* leaves: the sums over their own triangles, in index order (as host_sums adds them); inner nodes wait for their children */
__global__ __launch_bounds__(RPT_BLOCK) void k_probe_leaves(View s, double *area_all, double *area_ne, double *count, uint32_t *level) {
    const uint32_t n = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (n >= s.nn) return;
    const rpt_bvh_node &node = s.nodes[n];
    count[n] = 1.0;
    if (node.triangle_count == 0u) { level[n] = LEVEL_UNSET; return; }
    double a = 0.0, ne = 0.0;
    for (uint32_t k = 0; k < node.triangle_count; ++k) {
        const uint32_t t = node.left_or_first + k;
        a += s.tri_area[t];
        if (!emissive(s, t)) ne += s.tri_area[t];
    }
    area_all[n] = a; area_ne[n] = ne; level[n] = 0u;
}
/* pass p: the inner nodes whose children were both finished by EARLIER launches (level < p: nothing read here is written by this launch) */
__global__ __launch_bounds__(RPT_BLOCK) void k_probe_inner(View s, double *area_all, double *area_ne, double *count, uint32_t *level, uint32_t pass) {
    const uint32_t n = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (n >= s.nn || level[n] != LEVEL_UNSET) return;
    const uint32_t L = s.nodes[n].left_or_first, R = L + 1u;
    if (level[L] >= pass || level[R] >= pass) return;
    area_all[n] = area_all[L] + area_all[R];
    area_ne[n] = area_ne[L] + area_ne[R];
    count[n] = 1.0 + count[L] + count[R];
    level[n] = pass;
}
__global__ __launch_bounds__(RPT_BLOCK) void k_probe_flips(View s, uint32_t n_pairs, uint8_t *flip1, uint8_t *flip2, uint8_t *flip3) {
    const uint32_t p = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (p >= n_pairs) return;
    flip1[p] = prefers_right(s, p, 1) ? 1 : 0;
    if (flip2) { flip2[p] = prefers_right(s, p, 2) ? 1 : 0; flip3[p] = prefers_right(s, p, 3) ? 1 : 0; }
}
/* PROBE_LANES adjacent lanes per (probe ray, order): they take every decision together (the same registers, redundantly) and split the triangles of a leaf — one thread per
 * walk spent 5.8 ms on the clustered stand-in's 64-triangle leaves, whatever the GPU's width.  Job 2i walks ray i near child first, job 2i + 1 in the fixed order;
 * counters: node visits near first, fixed, rays, occluded */
constexpr uint32_t PROBE_LANES = 8u;
__global__ __launch_bounds__(RPT_WAVE) void k_probe_shadow(View s, const uint8_t *flip, unsigned long long *counters) {
    __shared__ uint32_t stacks[ORDER_PROBE_STACK * RPT_WAVE];
    const uint32_t job = (blockIdx.x * RPT_WAVE + threadIdx.x) / PROBE_LANES, i = job >> 1;
    s.sub = threadIdx.x % PROBE_LANES;
    s.lanes = PROBE_LANES;
    const bool all = !(s.area_ne[0] > 0.0);
    if (i >= SHADOW_PROBE_RAYS || (all && !(s.area_all[0] > 0.0))) return;
    V o, d;
    float max_t;
    if (!shadow_probe_ray(s, all, i, o, d, max_t)) return;
    bool occluded = false;
    const DevStack stack{stacks + threadIdx.x};
    const uint32_t visits = (job & 1u) == 0u ? walk<false>(s, flip, o, d, max_t, stack, occluded) : walk<true>(s, flip, o, d, max_t, stack, occluded);
    if (s.sub != 0u) return;
    if ((job & 1u) == 0u) {
        atomicAdd(&counters[0], (unsigned long long)visits);
        atomicAdd(&counters[2], 1ull);
        if (occluded) atomicAdd(&counters[3], 1ull);
    } else {
        atomicAdd(&counters[1], (unsigned long long)visits);
    }
}
/* PROBE_LANES lanes per (probe ray, order 0..3); counters 4..7: node visits near first and under rules 1..3; 8: rays; 9: hits */
__global__ __launch_bounds__(RPT_WAVE) void k_probe_last(View s, const uint8_t *flip1, const uint8_t *flip2, const uint8_t *flip3, unsigned long long *counters) {
    __shared__ uint32_t stacks[ORDER_PROBE_STACK * RPT_WAVE];
    const uint32_t job = (blockIdx.x * RPT_WAVE + threadIdx.x) / PROBE_LANES, i = job >> 2, q = job & 3u;
    s.sub = threadIdx.x % PROBE_LANES;
    s.lanes = PROBE_LANES;
    if (i >= LAST_PROBE_RAYS || !(s.area_ne[0] > 0.0)) return;
    V o, d;
    if (!last_probe_ray(s, i, o, d)) return;
    bool hit = false;
    const DevStack stack{stacks + threadIdx.x};
    const uint8_t *flip = q == 1u ? flip1 : (q == 2u ? flip2 : flip3);
    const uint32_t visits = q == 0u ? walk<false>(s, nullptr, o, d, 1000000.0f, stack, hit) : walk<true>(s, flip, o, d, 1000000.0f, stack, hit);
    if (s.sub != 0u) return;
    atomicAdd(&counters[4u + q], (unsigned long long)visits);
    if (q == 3u) {                                        /* (the host loop reports the hit flag of its last walk: rule 3) */
        atomicAdd(&counters[8], 1ull);
        if (hit) atomicAdd(&counters[9], 1ull);
    }
}

}  // namespace order_probe

/* Both decisions of shadow_order.h for the scene just uploaded into `c`, on the device: the same rays, the same node visits and therefore the same decision
 * as choose_shadow_order / choose_last_order make on the host (tests/test_gpu_parity.py compares them to the last digit).  d_cross_sq: what k_derive_triangles
 * left.  Kernels on the null stream, like the other upload-time kernels. */
static int device_order_probes(rpt_ctx *c, const float *d_cross_sq, uint32_t depth, bool pair_shaped, bool lights, bool want_last, ShadowOrder &so, LastOrder &lo) {
    using namespace order_probe;
    const Clock clock;
    so = ShadowOrder();
    lo = LastOrder();
    const uint32_t nt = c->scene.n_triangles, nn = c->scene.n_nodes, P = nn >= 3u ? (nn - 1u) / 2u : 0u;
    if (nt == 0u || (!lights && !want_last)) { so.probe_ms = lo.probe_ms = clock.ms(); return RPT_OK; }
    if (!pair_shaped || nn < 3u) { if (lights) so.why = "node pool is not pair-shaped"; so.probe_ms = lo.probe_ms = clock.ms(); return RPT_OK; }
    /* ONE allocation, carved up (five hipMalloc / hipFree pairs were a third of the probe's 5 ms on a 1 M-triangle scene) */
    DevBuf<unsigned char> arena;
    auto pad = [](size_t bytes) { return (bytes + 255u) & ~(size_t)255u; };
    const size_t o_tri = 0, o_sums = o_tri + pad((size_t)nt * sizeof(double)), o_level = o_sums + pad(3 * (size_t)nn * sizeof(double)),
                 o_flips = o_level + pad((size_t)nn * sizeof(uint32_t)), o_counters = o_flips + pad(3 * (size_t)P), total = o_counters + pad(10 * sizeof(unsigned long long));
    auto release = [&]() { arena.release(); };
#define PROBE_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { release(); c->error = std::string("order probes: ") + hipGetErrorString(e_); return RPT_EHIP; } } while (0)
    PROBE_TRY(arena.alloc(total));
    struct { double *p; } tri_area{reinterpret_cast<double *>(arena.p + o_tri)}, sums{reinterpret_cast<double *>(arena.p + o_sums)};
    struct { uint32_t *p; } level{reinterpret_cast<uint32_t *>(arena.p + o_level)};
    struct { uint8_t *p; } flips{arena.p + o_flips};
    struct { unsigned long long *p; } counters{reinterpret_cast<unsigned long long *>(arena.p + o_counters)};
    PROBE_TRY(hipMemsetAsync(counters.p, 0, 10 * sizeof(unsigned long long), nullptr));
    double *area_all = sums.p, *area_ne = sums.p + nn, *count = sums.p + 2 * (size_t)nn;
    const View s{reinterpret_cast<const rpt_per_vertex_data *>(c->per_vertex.p), reinterpret_cast<const rpt_triangle *>(c->indices.p),
                 reinterpret_cast<const rpt_bvh_node *>(c->nodes.p), reinterpret_cast<const rpt_material_data *>(c->materials.p), c->light_pick.p, nt, nn,
                 c->scene.n_light_pick, tri_area.p, area_all, area_ne, count, 0u, 1u, reinterpret_cast<const float4_like *>(c->tri_geom.p)};
    const unsigned node_blocks = (nn + RPT_BLOCK - 1) / RPT_BLOCK;
    k_probe_tri_area<<<(nt + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK>>>(d_cross_sq, nt, tri_area.p);
    k_probe_leaves<<<node_blocks, RPT_BLOCK>>>(s, area_all, area_ne, count, level.p);
    for (uint32_t pass = 1; pass <= depth; ++pass) k_probe_inner<<<node_blocks, RPT_BLOCK>>>(s, area_all, area_ne, count, level.p, pass);
    uint8_t *flip1 = flips.p, *flip2 = want_last ? flips.p + P : nullptr, *flip3 = want_last ? flips.p + 2 * (size_t)P : nullptr;
    k_probe_flips<<<(P + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK>>>(s, P, flip1, flip2, flip3);
    if (lights) k_probe_shadow<<<2 * SHADOW_PROBE_RAYS * PROBE_LANES / RPT_WAVE, RPT_WAVE>>>(s, flip1, counters.p);
    if (want_last) k_probe_last<<<4 * LAST_PROBE_RAYS * PROBE_LANES / RPT_WAVE, RPT_WAVE>>>(s, flip1, flip2, flip3, counters.p);
    unsigned long long h[10];
    PROBE_TRY(hipMemcpy(h, counters.p, sizeof(h), hipMemcpyDeviceToHost));       /* (waits for the kernels) */
    PROBE_TRY(hipGetLastError());
    if (lights) {
        decide_shadow(so, h[0], h[1], (uint32_t)h[2], (uint32_t)h[3], c->knobs.shadow_order);
        so.flip.assign(P, 0);
        if (so.fixed) PROBE_TRY(hipMemcpy(so.flip.data(), flip1, P, hipMemcpyDeviceToHost));
    }
    if (want_last) {
        const uint64_t v[4] = {h[4], h[5], h[6], h[7]};
        decide_last(lo, v, (uint32_t)h[8], (uint32_t)h[9], c->knobs.last_order);
        if (lo.rule != 0) {
            lo.flip.assign(P, 0);
            PROBE_TRY(hipMemcpy(lo.flip.data(), flips.p + (size_t)(lo.rule - 1) * P, P, hipMemcpyDeviceToHost));
        }
    }
#undef PROBE_TRY
    release();
    so.probe_ms = lo.probe_ms = clock.ms();
    return RPT_OK;
}

extern "C" {

int rpt_abi_version(void) { return RPT_ABI_VERSION; }

#ifndef RPT_BUILD_FINGERPRINT
#define RPT_BUILD_FINGERPRINT "unknown"
#endif
const char *rpt_build_fingerprint(void) { return RPT_BUILD_FINGERPRINT; }

int rpt_device_info(int device_id, uint32_t *compute_units_out, uint32_t *clock_khz_out) {
    int cus = 0, khz = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess ||
        hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, device_id) != hipSuccess) {
        g_create_error = "rpt_device_info: no such HIP device";
        return RPT_ENODEV;
    }
    if (compute_units_out) *compute_units_out = (uint32_t)cus;
    if (clock_khz_out) *clock_khz_out = (uint32_t)khz;
    return RPT_OK;
}

int rpt_shadow_order(rpt_ctx *c, uint32_t *fixed_out, double *visits_near_out, double *visits_fixed_out, uint32_t *probe_rays_out, double *probe_ms_out) {
    if (!c) return RPT_EINVAL;
    if (!c->has_scene) { c->error = "rpt_shadow_order: no scene"; return RPT_EINVAL; }
    if (fixed_out) *fixed_out = c->scene.shadow_fixed;
    if (visits_near_out) *visits_near_out = c->shadow_order.visits_near;
    if (visits_fixed_out) *visits_fixed_out = c->shadow_order.visits_fixed;
    if (probe_rays_out) *probe_rays_out = c->shadow_order.probe_rays;
    if (probe_ms_out) *probe_ms_out = c->shadow_order.probe_ms;
    return RPT_OK;
}

/* the same decision without a device (tests: the probe is host code) */
int rpt_last_bounce_order(rpt_ctx *c, uint32_t *mode_out, uint32_t *n_emissive_out, double *visits_out, uint32_t *probe_rays_out, double *probe_ms_out) {
    if (!c) return RPT_EINVAL;
    if (!c->has_scene) { c->error = "rpt_last_bounce_order: no scene"; return RPT_EINVAL; }
    const bool on = c->scene.lds_scene != 0u && c->stack_cap == 16 && c->scene.last_emit_n <= RPT_LAST_EMIT_MAX;
    if (mode_out) *mode_out = !on ? 0u : 1u + (uint32_t)c->last_order.rule;
    if (n_emissive_out) *n_emissive_out = c->scene.last_emit_n;
    if (visits_out) for (int k = 0; k < 4; ++k) visits_out[k] = c->last_order.visits[k];
    if (probe_rays_out) *probe_rays_out = c->last_order.probe_rays;
    if (probe_ms_out) *probe_ms_out = c->last_order.probe_ms;
    return RPT_OK;
}

int rpt_debug_shadow_order_host(const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                                const rpt_material_data *mats, size_t nm, const rpt_light_pick_entry *lp, size_t nlp, uint32_t *fixed_out,
                                double *visits_near_out, double *visits_fixed_out, uint32_t *probe_rays_out, uint8_t *flip_out /* (nn - 1) / 2, nullable */) {
    if (!pv || !idx || !nodes || !mats || !lp || nn == 0) return RPT_EINVAL;
    {   /* the probes walk the pool: the same validation rpt_upload_scene applies first (a child link that points at an ancestor would never end) */
        rpt_ctx scratch;
        uint32_t depth = 0;
        const int rc = validate_scene(&scratch, pv, nv, idx, nt, nodes, nn, mats, nm, lp, nlp, depth);
        if (rc) { g_create_error = scratch.error; return rc; }
    }
    const bool pair_shaped = pool_is_pair_shaped(nodes, nn);
    const ShadowOrder so = choose_shadow_order(pv, idx, nt, nodes, nn, mats, lp, nlp, pair_shaped, nullptr, rpt_read_knobs().shadow_order);
    if (fixed_out) *fixed_out = so.fixed ? 1u : 0u;
    if (visits_near_out) *visits_near_out = so.visits_near;
    if (visits_fixed_out) *visits_fixed_out = so.visits_fixed;
    if (probe_rays_out) *probe_rays_out = so.probe_rays;
    if (flip_out && !so.flip.empty()) memcpy(flip_out, so.flip.data(), so.flip.size());
    return RPT_OK;
}

int rpt_debug_last_order_host(const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes, size_t nn,
                              const rpt_material_data *mats, size_t nm, uint32_t *rule_out, double *visits_out /* [4] */, uint32_t *probe_rays_out,
                              uint8_t *flip_out /* (nn - 1) / 2, nullable */) {
    if (!pv || !idx || !nodes || !mats || nn == 0) return RPT_EINVAL;
    {
        rpt_ctx scratch;
        uint32_t depth = 0;
        rpt_light_pick_entry none{};
        none.ratio = -1.0f;
        const int rc = validate_scene(&scratch, pv, nv, idx, nt, nodes, nn, mats, nm, &none, 1, depth);
        if (rc) { g_create_error = scratch.error; return rc; }
    }
    const bool pair_shaped = pool_is_pair_shaped(nodes, nn);
    const LastOrder lo = choose_last_order(pv, idx, nt, nodes, nn, mats, pair_shaped, rpt_read_knobs().last_order);
    if (rule_out) *rule_out = (uint32_t)lo.rule;
    if (visits_out) for (int k = 0; k < 4; ++k) visits_out[k] = lo.visits[k];
    if (probe_rays_out) *probe_rays_out = lo.probe_rays;
    if (flip_out && !lo.flip.empty()) memcpy(flip_out, lo.flip.data(), lo.flip.size());
    return RPT_OK;
}

const char *rpt_last_error(rpt_ctx *ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int rpt_create(int device_id, rpt_ctx **out) {
    if (!out) { g_create_error = "null out pointer"; return RPT_EINVAL; }
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0) {
        g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        return RPT_ENODEV;
    }
    if (device_id < 0 || device_id >= n_dev) { g_create_error = "device id out of range"; return RPT_EINVAL; }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return RPT_EHIP; }
    auto *c = new rpt_ctx();
    c->device = device_id;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return RPT_EHIP; }
    e = hipHostMalloc(reinterpret_cast<void **>(&c->host_ring), RING * sizeof(unsigned long long), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->host_ring_dev), c->host_ring, 0);
    if (e != hipSuccess) { g_create_error = std::string("hipHostMalloc(mapped): ") + hipGetErrorString(e); (void)hipStreamDestroy(c->stream); delete c; return RPT_EHIP; }
    memset(c->host_ring, 0, RING * sizeof(unsigned long long));
    if (c->dev_stats.alloc(1) != hipSuccess || hipMemsetAsync(c->dev_stats.p, 0, sizeof(DevStats), c->stream) != hipSuccess) {
        g_create_error = "device allocation failed";
        rpt_destroy(c);
        return RPT_ENOMEM;
    }
    c->knobs = rpt_read_knobs();
    c->timing_level = c->knobs.stage_timing;
    c->stage_timing = c->timing_level != 0;
    if (c->knobs.shade_compact >= 0) { c->shade_compact_mode = c->knobs.shade_compact; c->shade_compact = c->shade_compact_mode == 1; }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) c->stream_max_blocks = 2u * (uint32_t)prop.multiProcessorCount;
    }
    c->sky_blocks = 16u * c->stream_max_blocks / 2u;       /* 16 workgroups of 256 per CU: the sky stage strides over its queue */
    if (c->knobs.sky_strided >= 0) {
        c->sky_strided_mode = c->knobs.sky_strided != 0 ? 1 : 0;
        c->sky_strided = c->sky_strided_mode == 1;
        if (c->knobs.sky_strided > 1) c->sky_blocks = (uint32_t)c->knobs.sky_strided;
    }
    *out = c;
    return RPT_OK;
}

void rpt_destroy(rpt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rpt_comm_release(c);
    release_state(c);
    c->gpairs.release(); c->glinks.release(); c->lds_image_shadow.release(); c->lds_image_last.release(); c->gpairs_shadow.release(); c->glinks_shadow.release();
    c->nodes.release(); c->lds_image.release(); c->tri_geom.release(); c->tri_isect.release(); c->tri_shade.release(); c->tri_tangent.release(); c->mat_lite.release();
    c->per_vertex.release(); c->materials.release();
    c->indices.release(); c->light_pick.release(); c->light_rec.release(); c->atlas.release(); c->skybox.release();
    c->dev_stats.release();
    for (hipEvent_t e : c->timing_events) (void)hipEventDestroy(e);
    for (auto &b : c->timing_pending) for (hipEvent_t e : b.ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->timing_pool) (void)hipEventDestroy(e);
    if (c->host_ring) (void)hipHostFree(c->host_ring);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int rpt_set_partition(rpt_ctx *c, uint32_t rank, uint32_t world_size) {
    if (!c) return RPT_EINVAL;
    if (world_size == 0 || rank >= world_size) { c->error = "rank must be < world_size"; return RPT_EINVAL; }
    if (c->rank == rank && c->world == world_size) return RPT_OK;      /* nothing changes: keep the state */
    c->rank = rank;
    c->world = world_size;
    if (c->has_config) {   /* re-derive the slot order for the new partition */
        rpt_tracing_config cfg = c->cfg.c;
        c->has_config = false;
        return rpt_set_config(c, &cfg);
    }
    return RPT_OK;
}

int rpt_upload_scene(rpt_ctx *c, const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt,
                     const rpt_bvh_node *nodes, size_t nn, const rpt_material_data *mats, size_t nm,
                     const rpt_light_pick_entry *lp, size_t nlp, const uint8_t *atlas, uint32_t aw, uint32_t ah,
                     const float *skybox, uint32_t sw, uint32_t sh) {
    if (!c) return RPT_EINVAL;
    if (!pv || !idx || !nodes || !mats || !lp) { c->error = "null scene buffer"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    {   /* the knobs that act at upload (orders, leaf build, LDS residency) are read again: a test process changes them between scenes of one context */
        const rpt_knobs now = rpt_read_knobs();
        c->knobs.shadow_order = now.shadow_order; c->knobs.last_order = now.last_order; c->knobs.coop_leaves = now.coop_leaves; c->knobs.no_lds_scene = now.no_lds_scene;
    }
    SectionTimer sections("rpt_upload_scene");
    int rc = validate_scene_flat(c, pv, nv, idx, nt, nn, mats, nm, lp, nlp);
    if (rc) return rc;
    for (size_t i = 0; i < nm; ++i)
        if ((mats[i].has_albedo_texture | mats[i].has_metallic_texture | mats[i].has_roughness_texture | mats[i].has_normal_texture) &&
            (!atlas || !aw || !ah)) {
            c->error = "a material references the texture atlas but no atlas was supplied";
            return RPT_ESCENE;
        }
    /* texel indices are 32-bit on the device (k_shade.h sample_by_lod): the reference's atlas is 4096 x 4096 (src/asset.rs:177) */
    if ((atlas && (uint64_t)aw * ah > (1ull << 30)) || (skybox && (uint64_t)sw * sh > (1ull << 28))) {
        c->error = "atlas larger than 2^30 texels / skybox larger than 2^28 texels";
        return RPT_ESCENE;
    }
    sections.mark("validate_flat");
    /* the node pool goes up first, into a buffer of its own: it is checked on the device (a tree, in range, at most 31 levels: device_validate_tree) before the
     * context's scene is touched — a rejected upload leaves the previous scene in place */
    DevBuf<float4> new_nodes;
    struct NodesGuard { DevBuf<float4> &b; ~NodesGuard() { b.release(); } } nodes_guard{new_nodes};
    HIP_TRY(c, new_nodes.alloc(2 * nn));
    HIP_TRY(c, hipMemcpy(new_nodes.p, nodes, nn * sizeof(rpt_bvh_node), hipMemcpyHostToDevice));
    NodeFacts facts{};
    rc = device_validate_tree(c, reinterpret_cast<const rpt_bvh_node *>(new_nodes.p), nn, nt, facts);
    if (rc) return rc;
    const uint32_t depth = facts.max_depth;
    sections.mark("validate_tree_device");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->has_scene = false;
    c->fat_leaves = (facts.flags & 1u) != 0u;
    if (c->knobs.coop_leaves >= 0) c->fat_leaves = c->knobs.coop_leaves != 0;

    /* derived per-triangle records, computed with the very f32 operations the reference performs per hit:
     *   tri_geom : a, e1 = b - a, e2 = c - a (muller_trumbore, intersection.rs:13-14; barycentric v0, v1, util.rs:239-240)
     *              with d00 = e1.e1, d01 = e1.e2, d11 = e2.e2 (util.rs:242-244) in the .w lanes
     *   tri_shade: the three vertex normals, the three uv0 pairs and the material index in 64 contiguous bytes
     *   mat_lite : emissive / albedo colours + roughness.x / metallic.x in 32 bytes (untextured scenes) */
    /* Computed ON THE DEVICE from the uploaded vertices and indices (k_derive_triangles: the same IEEE single operations, no contraction — the
     * host loops over a million triangles, three scattered 64-byte vertices each, and the transfer of their 148 bytes per triangle were 90 ms of a
     * 1 M-triangle upload, profiles/r05_startup_sections.txt).  The host derives `geom` itself only for a scene small enough for the LDS image,
     * whose builder reads it. */
    std::vector<float4> geom, lite(2 * nm);
    const bool lds_candidate = nn * 50 + nt * 48 <= RPT_LDS_SCENE_BYTES && depth <= 15;
    if (lds_candidate) {
        auto dot = [](const float *u, const float *v) { return (u[0] * v[0]) + (u[1] * v[1]) + (u[2] * v[2]); };
        geom.resize(3 * nt);
        for (size_t i = 0; i < nt; ++i) {
            const float *a = pv[idx[i].v0].vertex, *b = pv[idx[i].v1].vertex, *cc = pv[idx[i].v2].vertex;
            float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
            float e2[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
            geom[3 * i + 0] = make_float4(a[0], a[1], a[2], dot(e1, e1));
            geom[3 * i + 1] = make_float4(e1[0], e1[1], e1[2], dot(e1, e2));
            geom[3 * i + 2] = make_float4(e2[0], e2[1], e2[2], dot(e2, e2));
        }
    }
    uint32_t textured = 0;
    bool normal_maps = false;
    for (size_t i = 0; i < nm; ++i) {
        if (mats[i].has_normal_texture) normal_maps = true;
        lite[2 * i + 0] = make_float4(mats[i].emissive[0], mats[i].emissive[1], mats[i].emissive[2], mats[i].roughness[0]);
        lite[2 * i + 1] = make_float4(mats[i].albedo[0], mats[i].albedo[1], mats[i].albedo[2], mats[i].metallic[0]);
        if (mats[i].has_albedo_texture | mats[i].has_metallic_texture | mats[i].has_roughness_texture | mats[i].has_normal_texture) textured = 1;
    }
    sections.mark("derive_host");
    c->nodes.release();
    std::swap(c->nodes.p, new_nodes.p);
    std::swap(c->nodes.n, new_nodes.n);
    HIP_TRY(c, c->tri_geom.alloc(3 * nt));
    HIP_TRY(c, c->tri_shade.alloc(4 * nt));
    HIP_TRY(c, c->tri_isect.alloc(9 * nt));
    HIP_TRY(c, c->tri_tangent.alloc(normal_maps ? 3 * nt : 0));
    HIP_TRY(c, c->mat_lite.alloc(2 * nm));
    HIP_TRY(c, c->per_vertex.alloc(4 * nv));
    HIP_TRY(c, c->materials.alloc(6 * nm));
    HIP_TRY(c, c->indices.alloc(nt));
    HIP_TRY(c, c->light_pick.alloc(nlp));
    DevBuf<float> d_cross_sq;              /* |e1 x e2|^2 per triangle: the order probes' triangle areas (released on every return path below: see CrossSqGuard) */
    struct CrossSqGuard { DevBuf<float> &b; ~CrossSqGuard() { b.release(); } } cross_sq_guard{d_cross_sq};
    HIP_TRY(c, d_cross_sq.alloc(nt));
    HIP_TRY(c, hipMemcpy(c->per_vertex.p, pv, nv * sizeof(rpt_per_vertex_data), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->indices.p, idx, nt * sizeof(rpt_triangle), hipMemcpyHostToDevice));
    if (nt) k_derive_triangles<<<(unsigned)((nt + RPT_BLOCK - 1) / RPT_BLOCK), RPT_BLOCK>>>(c->per_vertex.p, c->indices.p, (uint32_t)nt, c->tri_geom.p, c->tri_isect.p,
                                                                                         c->tri_shade.p, c->tri_tangent.p, d_cross_sq.p);
    HIP_TRY(c, hipMemcpy(c->mat_lite.p, lite.data(), lite.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->materials.p, mats, nm * sizeof(rpt_material_data), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->light_pick.p, lp, nlp * sizeof(rpt_light_pick_entry), hipMemcpyHostToDevice));
    HIP_TRY(c, hipGetLastError());
    sections.mark("h2d_derive_device");
    {
        /* per light-pick entry, for its two triangles: corners, the mean of the three vertex normals exactly as
         * sample_direct_lighting forms it ((na + nb + nc) / 3.0, light_pick.rs:129), and the material's emission */
        std::vector<float4> rec(8 * nlp, make_float4(0, 0, 0, 0));
        if (!(lp[0].ratio < 0.0f))
            for (size_t i = 0; i < nlp; ++i)
                for (int side = 0; side < 2; ++side) {
                    const uint32_t t = side ? lp[i].triangle_index_b : lp[i].triangle_index_a;
                    const rpt_per_vertex_data &A = pv[idx[t].v0], &B = pv[idx[t].v1], &C = pv[idx[t].v2];
                    float n[3];
                    for (int k = 0; k < 3; ++k) n[k] = ((A.normal[k] + B.normal[k]) + C.normal[k]) / 3.0f;
                    const float *em = mats[idx[t].material].emissive;
                    float4 *r = &rec[8 * i + 4 * side];
                    r[0] = make_float4(A.vertex[0], A.vertex[1], A.vertex[2], n[0]);
                    r[1] = make_float4(B.vertex[0], B.vertex[1], B.vertex[2], n[1]);
                    r[2] = make_float4(C.vertex[0], C.vertex[1], C.vertex[2], n[2]);
                    r[3] = make_float4(em[0], em[1], em[2], 0.0f);
                }
        HIP_TRY(c, c->light_rec.alloc(rec.size()));
        HIP_TRY(c, hipMemcpy(c->light_rec.p, rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
    }

    sections.mark("h2d_2_light_rec");
    static const uint8_t magenta_u8[16] = {255, 0, 255, 255, 255, 0, 255, 255, 255, 0, 255, 255, 255, 0, 255, 255};
    static const float magenta_f[16] = {1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1};   /* src/asset.rs:283-290 */
    if (!atlas || !aw || !ah) { atlas = magenta_u8; aw = ah = 2; }
    if (!skybox || !sw || !sh) { skybox = magenta_f; sw = sh = 2; }
    HIP_TRY(c, c->atlas.alloc((size_t)aw * ah));
    HIP_TRY(c, c->skybox.alloc((size_t)sw * sh));
    HIP_TRY(c, hipMemcpy(c->atlas.p, atlas, (size_t)aw * ah * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->skybox.p, skybox, (size_t)sw * sh * 16, hipMemcpyHostToDevice));

    DevScene &s = c->scene;
    s.nodes = c->nodes.p; s.tri_geom = c->tri_geom.p; s.tri_isect = c->tri_isect.p; s.tri_shade = c->tri_shade.p; s.tri_tangent = c->tri_tangent.p; s.mat_lite = c->mat_lite.p;
    s.textured = textured;
    s.indices = c->indices.p; s.per_vertex = c->per_vertex.p;
    s.materials = c->materials.p; s.light_pick = c->light_pick.p; s.light_rec = c->light_rec.p;
    s.n_light_pick = (uint32_t)nlp;
    s.n_nodes = (uint32_t)nn;
    s.n_triangles = (uint32_t)nt;
    s.lds_scene = 0u; s.lds_image = nullptr; s.lds_pairs = s.lds_vecs = s.lds_root = 0u;
    if (lds_candidate) {
        std::vector<float4> image;
        uint32_t pairs = 0, root = 0;
        if (build_lds_image(nodes, nn, geom, nt, image, pairs, root) && image.size() * sizeof(float4) <= RPT_LDS_SCENE_BYTES) {
            HIP_TRY(c, c->lds_image.alloc(std::max<size_t>(1, image.size())));
            if (!image.empty())
                HIP_TRY(c, hipMemcpy(c->lds_image.p, image.data(), image.size() * sizeof(float4), hipMemcpyHostToDevice));
            s.lds_scene = 1u; s.lds_image = c->lds_image.p;
            s.lds_pairs = pairs; s.lds_vecs = (uint32_t)image.size(); s.lds_root = root;
        }
    }
    sections.mark("atlas_lds_image");
    if (c->knobs.no_lds_scene) s.lds_scene = 0u;
    /* pair records for the streamed global-memory walks (k_traverse.h SceneViewPairsT); a pool they cannot express keeps the one-shot walks */
    s.gpairs = nullptr; s.glinks = nullptr;
    const bool pair_shaped = (nn & 1u) == 1u && nn >= 3 && nodes[0].triangle_count == 0u && (facts.flags & 2u) == 0u;      /* (= pool_is_pair_shaped(nodes, nn), its per-node conditions from k_node_flags) */
    const uint32_t n_pairs = pair_shaped ? (uint32_t)((nn - 1) / 2) : 0u;
    if (pair_shaped) {
        HIP_TRY(c, c->gpairs.alloc(std::max<size_t>(1, 4 * (size_t)n_pairs)));
        HIP_TRY(c, c->glinks.alloc(nn));
        k_build_pairs<<<(n_pairs + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK>>>(c->nodes.p, nullptr, n_pairs, c->gpairs.p, c->glinks.p);
        HIP_TRY(c, hipGetLastError());
        s.gpairs = c->gpairs.p; s.glinks = c->glinks.p;
    } else {
        /* a previous, pair-shaped scene's records are of no use to this one (36 bytes per node of the OLD scene otherwise stay until rpt_destroy) */
        c->gpairs.release();
        c->glinks.release();
    }
    /* The any-hit (shadow) walks may visit siblings in any order (shadow_order.h: only `.hit` is read, light_pick.rs:148).  Probe rays decide per
     * scene between the reference's near-first order and a fixed opaque-first order; the latter walks a copy of the tree whose pairs are flipped so
     * that the preferred child is the LEFT one: a second LDS image / pair array, read by the shadow kernels only. */
    sections.mark("pairs");
    s.shadow_fixed = 0u; s.lds_image_shadow = nullptr; s.gpairs_shadow = nullptr; s.glinks_shadow = nullptr;
    /* The last extension rays of a batch without NEE only have to say "hit or miss" unless they can end on an emitter (k_traverse.h
     * k_traverse_nearest_stream LAST): the triangles whose material emits (lib.rs:86: emissive.xyz() != 0, a NaN counts), if they are few enough to test
     * each ray against; and room behind the LDS image for the flipped copy's pair records (two 1 024-thread workgroups per CU). */
    s.last_emit_n = 0u;
    s.last_flip_vecs = 0u;
    for (uint32_t k = 0; k < RPT_LAST_EMIT_MAX; ++k) s.last_emit_tri[k] = 0u;
    for (size_t t = 0; t < nt && s.last_emit_n <= RPT_LAST_EMIT_MAX; ++t) {
        const float *e = mats[idx[t].material].emissive;
        if (!(e[0] == 0.0f && e[1] == 0.0f && e[2] == 0.0f)) {
            if (s.last_emit_n < RPT_LAST_EMIT_MAX) s.last_emit_tri[s.last_emit_n] = (uint32_t)t;
            s.last_emit_n += 1u;
        }
    }
    if (c->knobs.last_order == 4) s.last_emit_n = RPT_LAST_EMIT_MAX + 1u;      /* RPT_LAST_ORDER=off (A/B and tests): the plain launch */
    const bool want_last = s.lds_scene && s.last_emit_n <= RPT_LAST_EMIT_MAX;
    /* both decisions by probe rays, as kernels over the buffers just uploaded (shadow_order.h; round 5 walked the rays on the host: 17 - 40 ms of a 1 M-triangle upload) */
    rc = device_order_probes(c, d_cross_sq.p, depth, pair_shaped, !(lp[0].ratio < 0.0f), want_last, c->shadow_order, c->last_order);
    if (rc) return rc;
    d_cross_sq.release();
    if (c->shadow_order.fixed) {
        bool built = false;
        if (s.lds_scene) {
            const std::vector<rpt_bvh_node> pool = flipped_nodes(nodes, nn, c->shadow_order.flip);
            std::vector<float4> image;
            uint32_t pairs = 0, root = 0;
            if (build_lds_image(pool.data(), nn, geom, nt, image, pairs, root) && image.size() == (size_t)s.lds_vecs && pairs == s.lds_pairs && root == s.lds_root) {
                HIP_TRY(c, c->lds_image_shadow.alloc(std::max<size_t>(1, image.size())));
                HIP_TRY(c, hipMemcpy(c->lds_image_shadow.p, image.data(), image.size() * sizeof(float4), hipMemcpyHostToDevice));
                s.lds_image_shadow = c->lds_image_shadow.p;
                built = true;
            }
        }
        if (s.gpairs) {
            DevBuf<uint8_t> d_flip;
            HIP_TRY(c, d_flip.alloc(std::max<size_t>(1, c->shadow_order.flip.size())));
            hipError_t e_f = hipMemcpy(d_flip.p, c->shadow_order.flip.data(), c->shadow_order.flip.size(), hipMemcpyHostToDevice);
            if (e_f == hipSuccess) e_f = c->gpairs_shadow.alloc(std::max<size_t>(1, 4 * (size_t)n_pairs));
            if (e_f == hipSuccess) e_f = c->glinks_shadow.alloc(nn);
            if (e_f == hipSuccess) {
                k_build_pairs<<<(n_pairs + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK>>>(c->nodes.p, d_flip.p, n_pairs, c->gpairs_shadow.p, c->glinks_shadow.p);
                e_f = hipDeviceSynchronize();                  /* (d_flip goes out of scope) */
            }
            d_flip.release();
            HIP_TRY(c, e_f);
            s.gpairs_shadow = c->gpairs_shadow.p; s.glinks_shadow = c->glinks_shadow.p;
            built = true;
        }
        s.shadow_fixed = built ? 1u : 0u;
    }
    sections.mark("shadow_order");
    if (!s.lds_image_shadow) c->lds_image_shadow.release();
    if (!s.gpairs_shadow) { c->gpairs_shadow.release(); c->glinks_shadow.release(); }
    s.lds_image_last = nullptr;
    if (want_last) {
        /* the order those rays walk in (shadow_order.h): near child first over the primary image, or a fixed order over a copy whose pairs
         * are flipped by the rule that needed the fewest node visits on probe rays of their kind */
        const size_t flip_vecs = 6 * (size_t)s.lds_pairs + ((size_t)s.lds_pairs + 3) / 4;
        /* room: two such workgroups per CU (k_traverse.h), i.e. half of what THIS device's CU holds (160 KB on MI355X; a partitioned or older device
         * reports less and simply gets no flipped copy), minus the kernel's static LDS as the code object states it */
        size_t lds_room = 0;
        {
            hipDeviceProp_t prop;
            hipFuncAttributes fa;
            if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.maxSharedMemoryPerMultiProcessor != 0 &&
                rpt_last_walk_attributes(&fa) == hipSuccess) {
                const size_t per_wg = prop.maxSharedMemoryPerMultiProcessor / 2;
                lds_room = per_wg > fa.sharedSizeBytes ? per_wg - fa.sharedSizeBytes : 0;
            }
        }
        if (c->last_order.rule != 0 && ((size_t)s.lds_vecs + flip_vecs) * sizeof(float4) <= lds_room) {
            const std::vector<rpt_bvh_node> pool = flipped_nodes(nodes, nn, c->last_order.flip);
            std::vector<float4> image;
            uint32_t pairs = 0, root = 0;
            if (build_lds_image(pool.data(), nn, geom, nt, image, pairs, root) && image.size() == (size_t)s.lds_vecs && pairs == s.lds_pairs && root == s.lds_root) {
                HIP_TRY(c, c->lds_image_last.alloc(std::max<size_t>(1, flip_vecs)));
                HIP_TRY(c, hipMemcpy(c->lds_image_last.p, image.data(), flip_vecs * sizeof(float4), hipMemcpyHostToDevice));
                s.lds_image_last = c->lds_image_last.p;
                s.last_flip_vecs = (uint32_t)flip_vecs;
            }
        }
        if (!s.lds_image_last) c->last_order.rule = 0;
    }
    if (!s.lds_image_last) c->lds_image_last.release();
    sections.mark("last_order");
    s.no_lights = lp[0].ratio < 0.0f ? 1u : 0u;
    s.fastdiv_ok = (facts.flags & 4u) == 0u ? 1u : 0u;          /* every node bound is 0 or in [2^-60, 2^40) (k_node_flags) */
    s.atlas = DevImage{c->atlas.p, aw, ah};
    s.skybox = DevImage{c->skybox.p, sw, sh};
    sections.mark("fastdiv_check");
    HIP_TRY(c, hipStreamSynchronize(nullptr));                 /* the derive / pair kernels ran on the null stream; the context renders on its own */
    HIP_TRY(c, hipGetLastError());
    c->bvh_depth = depth;
    c->stack_cap = depth <= 15 ? 16 : (depth <= 23 ? 24 : 32);
    c->has_scene = true;
    return RPT_OK;
}

int rpt_set_config(rpt_ctx *c, const rpt_tracing_config *cfg) {
    if (!c || !cfg) return RPT_EINVAL;
    if (cfg->width == 0 || cfg->height == 0 || cfg->width > 65535u || cfg->height > 65535u) {
        c->error = "width/height must be in 1..65535";
        return RPT_EINVAL;
    }
    uint32_t nee_mode = cfg->nee <= 2u ? cfg->nee : 0u;
    if (cfg->max_bounces > 255u) { c->error = "max_bounces > 255"; return RPT_ECONFIG; }
    {
        /* LDS dimension budget (kernels/src/rng.rs:20-21,51-54): the CPU reference panics past 31 */
        uint64_t per_bounce = 3u + (nee_mode ? 4u : 0u);
        uint64_t rr = cfg->max_bounces > cfg->min_bounces + 1u ? cfg->max_bounces - 1u - cfg->min_bounces : 0u;
        uint64_t dims = 2u + (uint64_t)cfg->max_bounces * per_bounce + rr;
        if (dims > 31u) {
            c->error = "config needs " + std::to_string(dims) + " LDS dimensions; the reference's table has 31 usable";
            return RPT_ECONFIG;
        }
    }
    HIP_TRY(c, hipSetDevice(c->device));
    bool resized = !c->has_config || c->cfg.c.width != cfg->width || c->cfg.c.height != cfg->height;
    c->cfg.c = *cfg;
    c->cfg.nee_mode = nee_mode;
    float ry[9], rx[9];
    rotation_y(cfg->cam_rotation[1], ry);
    rotation_x(cfg->cam_rotation[0], rx);
    mat3_mul_host(ry, rx, c->cfg.euler);
    rotation_y(rptm::atan2r(cfg->sun_direction[2], cfg->sun_direction[0]), c->cfg.sky_rot);
    if (resized || !c->has_state) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        release_state(c);
        build_pixel_order(cfg->width, cfg->height, c->rank, c->world, c->pixel_xy_host);
        c->n_pixels = (uint32_t)c->pixel_xy_host.size();
        /* Samples of one pixel in flight.  The GPU holds 8 192 waves = 0.5 M paths at once and ray costs inside a
         * launch vary widely, so a launch of only 1-2 x that ends in a long half-empty tail (measured 4.3 of 8 waves
         * per SIMD resident on average with 1 M slots); up to 16 M slots (~3 GB of path state at 200 B/slot, nothing
         * on a 288 GB part) make the tail a small fraction and quarter the number of launches per batch:
         * DarkCornell 1024^2 4.84 -> 6.33 Grays/s for S = 1 -> 16.  The image does not depend on S. */
        /* This is the MOST a call may use (the arrays are sized for it); each rpt_render call keeps
         * min(this, next power of two >= its n_samples) slots per pixel busy — slots without a sample would only be
         * scanned (16 spp on 32 slots per pixel: 6.0 instead of 8.2 Grays/s).  Up to 32 slots per pixel and 32 M
         * slots: with the reference's default batch of 32 samples (sync_rate, src/trace.rs:75) a rank that owns 1/8
         * of a 1024^2 image then has 4 M paths in flight (7.2 instead of 6.6 Grays/s per GPU). */
        /* Up to 256 since round 6 (k_complete.h counts the finished slots of a pixel instead of keeping a 32-bit mask): a rank that owns 1/8
         * of a 1024^2 image runs a 256-sample batch as the same 33 M-slot launches as the whole image runs 32 — the fixed costs of a batch
         * (11 launches, drain tails) no longer weigh 8 x as much.  The arrays are allocated by the render call that needs them, at its size. */
        uint32_t S = 1;
        if (c->samples_in_flight_request > 0) {
            while (S < (uint32_t)c->samples_in_flight_request && S < RPT_MAX_SAMPLES_IN_FLIGHT) S <<= 1;
        } else {
            /* as many as fit in RPT_MAX_SLOTS (160 M slots x 70 - 150 B of path state and queues).  Round 1
             * stopped at 32 M and used S = 1 from 3 M pixels up, where more slots only cost: with one ray per lane the dead
             * slots of an open scene were walked as empty lanes.  The streamed walks skip them, and measured now (32-spp
             * batches): PBRTest 2048^2 4090 / 4354 / 4505 / 4537 / 4709 Mrays/s for S = 1 / 4 / 8 / 16 / 32, VeachMIS 1080p
             * 4388 / 4757 / 4892 for S = 8 / 16 / 32, the 1 M-triangle stand-in at 2048^2 994 / 1578 for S = 1 / 16. */
            while (S < RPT_MAX_SAMPLES_IN_FLIGHT && (uint64_t)c->n_pixels * S * 2u <= c->max_slots_budget) S <<= 1;
        }
        c->max_group_shift = 0;
        while ((1u << c->max_group_shift) < S) c->max_group_shift += 1;
        c->max_slots = padded_pixels(c->n_pixels) << c->max_group_shift;       /* chunks of 64 pixels x S slots (k_common.h, slot_pix) */
        c->group_shift = std::min(c->max_group_shift, 5u);                     /* (until the first render call says how many it needs) */
        c->n_slots = padded_pixels(c->n_pixels) << c->group_shift;
        int rc = alloc_pixel_state(c);
        if (rc) return rc;
        /* fresh accumulators; seeds must come from rpt_reset */
        if (c->n_pixels) {
            HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, c->n_pixels * sizeof(float4), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->rng.p, 0, c->n_pixels * sizeof(uint2), c->stream));
        }
        c->samples = 0;
    }
    c->has_config = true;
    return RPT_OK;
}

int rpt_reset(rpt_ctx *c, const rpt_rng_state *seed, const float *accum_init, uint32_t samples_init) {
    if (!c) return RPT_EINVAL;
    if (!c->has_config || !c->has_state) { c->error = "rpt_set_config must precede rpt_reset"; return RPT_EINVAL; }
    if (!seed) { c->error = "null seed buffer"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    /* The caller's row-major buffers go up as they are; a kernel gathers them into this rank's tile-major pixel order (rounds 1-5 gathered on the host:
     * two loops over every pixel and 24 bytes per pixel of staging, 20 ms of a 2048^2 reset).  No accum_init: the accumulators are simply zeroed. */
    const uint32_t W = c->cfg.c.width, H = c->cfg.c.height;
    const size_t n = c->n_pixels, whole = (size_t)W * H;
    if (n) {
        DevBuf<uint2> d_seed;
        DevBuf<float4> d_acc;
        HIP_TRY(c, d_seed.alloc(whole));
        hipError_t e = hipMemcpy(d_seed.p, seed, whole * sizeof(uint2), hipMemcpyHostToDevice);
        if (e == hipSuccess && accum_init) {
            e = d_acc.alloc(whole);
            if (e == hipSuccess) e = hipMemcpy(d_acc.p, accum_init, whole * sizeof(float4), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            k_reset_gather<<<(unsigned)((n + RPT_BLOCK - 1) / RPT_BLOCK), RPT_BLOCK, 0, c->stream>>>(c->pixel_xy.p, (uint32_t)n, W, d_seed.p, d_acc.p, c->rng.p, c->accum.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);             /* (the staging buffers go out of scope) */
        d_seed.release(); d_acc.release();
        HIP_TRY(c, e);
    }
    HIP_TRY(c, hipMemsetAsync(c->dev_stats.p, 0, sizeof(DevStats), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->ray_shards.p, 0, RPT_STAT_SHARDS * RPT_STAT_STRIDE * sizeof(unsigned long long), c->stream));
    c->samples = accum_init ? samples_init : 0u;
    c->stats = rpt_stats{};
    return RPT_OK;
}

/* Read the device counters after a synchronisation: reports samples a fixed-length batch left in flight, and picks the
 * shade-stage variant for the batches to come.  Packing the traversed slots per workgroup before shading (k_shade<..,
 * COMPACT>) pays when most slots of a pass are parked — measured: PBRTest 2048^2 (0.94 sky hits per sample) shade 86.6 ->
 * 68.5 ms per 4 batches — and costs on scenes whose paths stay alive (DarkCornell 31.4 -> 39.2; VeachMIS with 0.84: even):
 * so it is switched on when more than RPT_SHADE_COMPACT_AT (default 0.7) of the samples rendered since the last reset
 * ended in the sky.  Either variant produces the same image bit for bit. */
static int refresh_device_stats(rpt_ctx *c, const char *what) {
    DevStats ds;
    HIP_TRY(c, hipMemcpy(&ds, c->dev_stats.p, sizeof(ds), hipMemcpyDeviceToHost));
    if (ds.undrained != 0ull) {
        c->error = std::string(what) + ": " + std::to_string(ds.undrained) + " samples were still in flight after its iterations (internal error)";
        return RPT_EHIP;
    }
    if (c->shade_compact_mode < 0 && c->stats.samples != 0ull)
        c->shade_compact = (double)ds.sky_evals > c->shade_compact_at * (double)c->stats.samples;
    /* the sky stage walks its queue with a small fixed grid when misses are rare (k_sky<STRIDED>: see there) */
    if (c->sky_strided_mode < 0 && c->stats.samples != 0ull) c->sky_strided = (double)ds.sky_evals < 0.05 * (double)c->stats.samples;
    return RPT_OK;
}

/* rpt_wait: completes everything rpt_render_async enqueued (and folds its stage timing into the statistics). */
int rpt_wait(rpt_ctx *c) {
    if (!c) return RPT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    /* the last asynchronous batch must have left every slot idle — ALL the context's slots, not only the ones this call used: k_complete's
     * final pass looks at the slots of pixels it could not complete, and a slot beyond this call's n_slots (left by an earlier call with
     * more samples in flight) is outside its view.  One 8-byte read per slot, once per rpt_wait. */
    if (c->async_pending && c->has_state && c->hit.n) {
        const uint32_t all = (uint32_t)c->hit.n;
        k_check_drained<<<(all + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, c->stream>>>(c->hit.p, all, c->dev_stats.p);
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipGetLastError());
    if (c->async_pending) {
        /* An asynchronous batch runs a fixed number of iterations and never inspects a progress report.  Every batch
         * checks that its predecessor left all slots idle (k_generate_first), the last one is checked just above. */
        c->async_pending = false;
        int rc = refresh_device_stats(c, "asynchronous batch not drained");
        if (rc) return rc;
    }
    for (auto &b : c->timing_pending) {
        timing_accumulate(c, b.ev, b.iterations, b.complete_timed);
        c->timing_pool.insert(c->timing_pool.end(), b.ev.begin(), b.ev.end());
    }
    c->timing_pending.clear();
    c->async_pending = false;
    return RPT_OK;
}

int rpt_stream(rpt_ctx *c, void **stream_out) {
    if (!c || !stream_out) return RPT_EINVAL;
    *stream_out = reinterpret_cast<void *>(c->stream);
    return RPT_OK;
}

static int render_impl(rpt_ctx *c, uint32_t n_samples, bool allow_async) {
    if (!c) return RPT_EINVAL;
    if (!c->has_scene || !c->has_config || !c->has_state) { c->error = "scene, config and reset must precede rpt_render"; return RPT_EINVAL; }
    if (n_samples == 0 || c->n_pixels == 0) { c->samples += n_samples; return RPT_OK; }
    if ((uint64_t)n_samples + (1u << c->max_group_shift) >= 0x100000000ull) { c->error = "n_samples too large"; return RPT_EINVAL; }
    {   /* slots per pixel of THIS call: no more than it has samples for */
        uint32_t shift = 0;
        while (shift < c->max_group_shift && (1u << shift) < n_samples) shift += 1;
        c->group_shift = shift;
        c->n_slots = padded_pixels(c->n_pixels) << shift;
        c->state.group_shift = shift;
        /* samples of one pixel per wave (k_common.h, slot_pix): 1 unless the scene is a large one */
        const uint32_t qs = c->knobs.slot_q_shift >= 0 ? (uint32_t)c->knobs.slot_q_shift : (c->scene.n_triangles >= RPT_BIG_SCENE_TRIANGLES ? 5u : 0u);
        c->state.q_shift = qs < shift ? qs : shift;
        c->state.n_slots = c->n_slots;
        c->queues.sky_wide_limit = std::min(c->n_slots / 16u, c->sky_wide_cfg);   /* the wide sky pass spends 16 threads of the grid per miss */
        HIP_TRY(c, hipSetDevice(c->device));
        int rc = ensure_slot_state(c, c->n_slots, c->cfg.nee_mode != RPT_NEE_NONE, c->cfg.nee_mode == RPT_NEE_MIS);
        if (rc) return rc;
    }
    /* When no slot gets a second sample in this call (n_samples <= slots per pixel) nothing is regenerated: every path
     * ends within max_bounces iterations (lib.rs:62), its misses and shadow rays inside the iteration that produced
     * them (with several slots per pixel a path ended by a side stage is accumulated by the NEXT shade pass: one more
     * iteration) — so exactly that many iterations are enqueued and no progress report is awaited (saves the run-ahead's
     * surplus launches, 4 % of a 1.3 ms batch on 1/8 of an image). */
    const uint64_t known_iterations =
        (n_samples <= (1u << c->group_shift) && c->queues.sky_threshold <= 1u) ? (uint64_t)c->cfg.c.max_bounces : 0u;
    HIP_TRY(c, hipSetDevice(c->device));
    auto t0 = std::chrono::steady_clock::now();
    hipStream_t s = c->stream;
    const uint32_t blocks = (c->n_slots + RPT_BLOCK - 1) / RPT_BLOCK;
    /* asynchronous only when the iteration count is known up front: nothing has to be polled */
    const bool async = allow_async && known_iterations != 0;
    if (!async && c->async_pending) {                     /* the progress ring is about to be reused by the host */
        int rc = rpt_wait(c);
        if (rc) return rc;
    }

    if (!async) for (int k = 0; k < RING; ++k) __atomic_store_n(&c->host_ring[k], 0ull, __ATOMIC_RELAXED);
    c->call_samples = n_samples;
    /* A miss ends its path (lib.rs:79) and in a batch of known length nothing is started in its place: the misses of all iterations
     * wait in the queue for ONE sky launch after the last iteration (three launches less per batch) */
    c->queues.sky_at_end = known_iterations != 0 ? 1u : 0u;
    c->queues.known_length = known_iterations != 0 ? 1u : 0u;
    std::vector<hipEvent_t> async_events;
    std::vector<hipEvent_t> *ev = c->stage_timing ? (async ? &async_events : &c->timing_events) : nullptr;
    size_t ev_at = 0;
    if (ev && async) {
        /* this batch's own events, from the pool: they are read back by rpt_wait */
        const size_t need = timing_events_needed(c, known_iterations);
        while (c->timing_pool.size() < need) {
            hipEvent_t e;
            HIP_TRY(c, hipEventCreate(&e));
            c->timing_pool.push_back(e);
        }
        async_events.assign(c->timing_pool.end() - (long)need, c->timing_pool.end());
        c->timing_pool.resize(c->timing_pool.size() - need);
    }
    const bool time_stages = ev && c->timing_level == 1;
    if (time_stages) {
        while (ev->size() < (size_t)EVENTS_LEAD) { hipEvent_t e; HIP_TRY(c, hipEventCreate(&e)); ev->push_back(e); }
        HIP_TRY(c, hipEventRecord((*ev)[ev_at++], s));
    }
    k_generate_first<<<blocks, RPT_BLOCK, 0, s>>>(c->state, c->queues, c->cfg, n_samples, c->dev_stats.p);
    c->stats.kernel_launches[RPT_STAGE_GENERATE] += 1;
    if (time_stages) HIP_TRY(c, hipEventRecord((*ev)[ev_at++], s));
    bool complete_timed = false;

    uint64_t it = 0, full_iterations = 0;
    bool drained = c->cfg.c.max_bounces == 0u;
    /* rpt_debug_short_batch (test aid): enqueue one iteration too few in an asynchronous batch, to prove that the
     * completion checks of rpt_wait / k_generate_first notice */
    uint64_t short_batch = 0;
    if (async && known_iterations > 1 && c->test_short_batch) short_batch = 1;
    /* Run-ahead: it only has to cover the enqueue latency (tens of microseconds).  Launches over millions of slots
     * last far longer than that, and every surplus iteration still dispatches its (instantly returning) workgroups. */
    const int lag = c->n_slots >= (512u << 10) ? 2 : (c->n_slots >= (128u << 10) ? 3 : LAG);
    /* worst case: every sample needs max_bounces iterations, one after another */
    /* safety net against a stuck pipeline (a bug), far above what deferral of sky work can cost */
    const uint64_t it_limit = (uint64_t)n_samples * (uint64_t)(c->cfg.c.max_bounces + 2u) * 16u + 4096u;
    while (!drained) {
        /* only the synchronous call's vector grows as it goes (its iteration count is unknown); an asynchronous batch got exactly
           timing_events_needed() events from the pool above — 2 per iteration at level 2 — and hands exactly those back in rpt_wait */
        if (ev && !async) {
            const size_t per_iter = c->timing_level == 2 ? 2u : (size_t)(EVENTS_PER_ITER + EVENTS_TAIL);
            if (ev->size() < ev_at + per_iter) {
                size_t old = ev->size();
                ev->resize(ev_at + per_iter * 64);
                for (size_t k = old; k < ev->size(); ++k) HIP_TRY(c, hipEventCreate(&(*ev)[k]));
            }
        }
        const bool complete_each = known_iterations == 0 && c->group_shift != 0;
        const bool sky_now = it + 1 == known_iterations - short_batch;      /* (sky_at_end: the one sky launch of the batch) */
        switch (c->stack_cap) {
            case 16: launch_iteration_stack<16>(c, (uint32_t)it, blocks, ev, ev_at, complete_each, sky_now); break;
            case 24: launch_iteration_stack<24>(c, (uint32_t)it, blocks, ev, ev_at, complete_each, sky_now); break;
            default: launch_iteration_stack<32>(c, (uint32_t)it, blocks, ev, ev_at, complete_each, sky_now); break;
        }
        full_iterations += 1u;
        it += 1;
        if (it == known_iterations - short_batch) {             /* (no report needed: nothing can be left) */
            /* every path of the batch has ended (max_bounces iterations, side stages included): the one completion of the batch */
            if (c->group_shift != 0) {
                launch_complete(c, (uint32_t)it, 1u);
                c->stats.kernel_launches[RPT_STAGE_COMPLETE] += 1;
                if (time_stages) { HIP_TRY(c, hipEventRecord((*ev)[ev_at++], s)); complete_timed = true; }
            }
            break;
        }
        if (known_iterations == 0 && it >= (uint64_t)lag) {
            /* the sky kernel of iteration j published (j + 1) << 32 | "work remains after iteration j" */
            uint64_t j = it - lag;
            volatile unsigned long long *slot = &c->host_ring[j & (RING - 1)];
            unsigned long long v;
            uint64_t spins = 0;
            while (((v = *slot) >> 32) != ((j + 1) & 0xffffffffull)) {
                if (++spins > 2000000000ull || hipStreamQuery(s) == hipSuccess) {
                    v = *slot;
                    if ((v >> 32) == ((j + 1) & 0xffffffffull)) break;
                    HIP_TRY(c, hipGetLastError());
                    c->error = "wavefront progress report never arrived (internal error)";
                    return RPT_EHIP;
                }
            }
            if ((uint32_t)v == 0u) drained = true;       /* no ray traced, no sample started, no miss waiting: all later iterations are no-ops */
        }
        if (it > it_limit) { c->error = "wavefront did not drain (internal error)"; return RPT_EHIP; }
    }
    if (async) {
        c->async_pending = true;
        c->stats.iterations += it;
        c->stats.kernel_launches[RPT_STAGE_TRAVERSE] += full_iterations;
        c->stats.kernel_launches[RPT_STAGE_SHADE] += it;
        c->stats.kernel_launches[RPT_STAGE_SHADOW] += c->cfg.nee_mode != RPT_NEE_NONE ? full_iterations : 0;
        c->stats.kernel_launches[RPT_STAGE_SKY] += c->queues.sky_at_end ? 1u : full_iterations;
        if (ev) c->timing_pending.push_back(rpt_ctx::TimingBatch{async_events, it, complete_timed});
        c->samples += n_samples;
        c->stats.samples += (uint64_t)c->n_pixels * n_samples;
        c->stats.render_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        return RPT_OK;
    }
    /* a call that enqueued a fixed number of iterations must have left every slot idle: cross-check of that bound */
    if (known_iterations != 0 && c->n_slots && c->group_shift == 0)     /* (several slots per pixel: k_complete's final pass has counted) */
        k_check_drained<<<(c->n_slots + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, s>>>(c->hit.p, c->n_slots, c->dev_stats.p);
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, hipGetLastError());
    const bool nee = c->cfg.nee_mode != RPT_NEE_NONE;
    c->stats.iterations += it;
    c->stats.kernel_launches[RPT_STAGE_TRAVERSE] += full_iterations;
    c->stats.kernel_launches[RPT_STAGE_SHADE] += it;
    c->stats.kernel_launches[RPT_STAGE_SHADOW] += nee ? full_iterations : 0;
    c->stats.kernel_launches[RPT_STAGE_SKY] += c->queues.sky_at_end ? 1u : full_iterations;
    if (ev) timing_accumulate(c, *ev, it, complete_timed);
    c->samples += n_samples;
    c->stats.samples += (uint64_t)c->n_pixels * n_samples;
    c->stats.render_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {
        /* (an undrained count can only be non-zero for a call that enqueued a fixed number of iterations) */
        int rc = refresh_device_stats(c, "wavefront not drained after its known number of iterations");
        if (rc) return rc;
    }
    return RPT_OK;
}

int rpt_render(rpt_ctx *c, uint32_t n_samples) { return render_impl(c, n_samples, false); }

/* Like rpt_render, but returns as soon as the batch is enqueued when its iteration count is known up front (no slot
 * gets a second sample: n_samples <= slots per pixel); otherwise identical to rpt_render.  Every entry point that
 * reads results synchronises by itself; rpt_wait does so explicitly. */
int rpt_render_async(rpt_ctx *c, uint32_t n_samples) { return render_impl(c, n_samples, true); }

int rpt_read_rng(rpt_ctx *c, rpt_rng_state *out) {
    if (!c || !out) return RPT_EINVAL;
    if (!c->has_state) { c->error = "nothing to read: no config"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const uint32_t W = c->cfg.c.width, H = c->cfg.c.height;
    std::vector<uint2> rng(c->n_pixels);
    if (c->n_pixels) HIP_TRY(c, hipMemcpy(rng.data(), c->rng.p, c->n_pixels * sizeof(uint2), hipMemcpyDeviceToHost));
    memset(out, 0, (size_t)W * H * sizeof(rpt_rng_state));
    for (size_t s = 0; s < c->n_pixels; ++s) {
        uint32_t pxy = c->pixel_xy_host[s];
        out[(size_t)(pxy >> 16) * W + (pxy & 0xffffu)] = rpt_rng_state{rng[s].x, rng[s].y};
    }
    return RPT_OK;
}

int rpt_local_pixels(rpt_ctx *c, uint64_t *n) {
    if (!c || !n) return RPT_EINVAL;
    if (!c->has_config) { c->error = "no config"; return RPT_EINVAL; }
    *n = c->n_pixels;
    return RPT_OK;
}

int rpt_set_samples_in_flight(rpt_ctx *c, int s) {
    if (!c) return RPT_EINVAL;
    if (s < 0 || s > (int)RPT_MAX_SAMPLES_IN_FLIGHT) { c->error = "samples in flight must be 0 (automatic) or 1..256"; return RPT_EINVAL; }
    c->samples_in_flight_request = s;
    if (c->has_config) {   /* re-derive the slot count */
        rpt_tracing_config cfg = c->cfg.c;
        c->has_config = false;
        return rpt_set_config(c, &cfg);
    }
    return RPT_OK;
}

int rpt_rank_pixels(rpt_ctx *c, uint32_t rank, uint64_t *n) {
    if (!c || !n) return RPT_EINVAL;
    if (!c->has_config || rank >= c->world) { c->error = "no config / bad rank"; return RPT_EINVAL; }
    std::vector<uint32_t> order;
    build_pixel_order(c->cfg.c.width, c->cfg.c.height, rank, c->world, order);
    *n = order.size();
    return RPT_OK;
}

int rpt_tile_order(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size, uint32_t *out_xy, size_t capacity,
                   size_t *n) {
    if (!n || !width || !height || width > 65535u || height > 65535u || !world_size || rank >= world_size) return RPT_EINVAL;
    std::vector<uint32_t> order;
    build_pixel_order(width, height, rank, world_size, order);
    *n = order.size();
    if (out_xy) {
        if (capacity < order.size()) return RPT_EINVAL;
        memcpy(out_xy, order.data(), order.size() * sizeof(uint32_t));
    }
    return RPT_OK;
}

int rpt_resolve(rpt_ctx *c, uint32_t tonemap_op, float *out_rgb) {
    if (!c || !out_rgb) return RPT_EINVAL;
    if (!c->has_state) { c->error = "nothing to resolve: no config"; return RPT_EINVAL; }
    if (tonemap_op > 6u) { c->error = "tonemap operator must be 0..6"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_out = (size_t)c->cfg.c.width * c->cfg.c.height * 3;
    DevBuf<float> dev;
    HIP_TRY(c, dev.alloc(n_out));
    hipError_t e = hipMemsetAsync(dev.p, 0, n_out * sizeof(float), c->stream);
    if (e == hipSuccess && c->n_pixels) {
        k_resolve<<<(c->n_pixels + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, c->stream>>>(c->accum.p, c->pixel_xy.p, c->n_pixels, c->cfg.c.width,
                                                                                      (float)c->samples, tonemap_op, dev.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out_rgb, dev.p, n_out * sizeof(float), hipMemcpyDeviceToHost);
    dev.release();
    HIP_TRY(c, e);
    return RPT_OK;
}

int rpt_get_stats(rpt_ctx *c, rpt_stats *out) {
    if (!c || !out) return RPT_EINVAL;
    { int rc = rpt_wait(c); if (rc) return rc; }
    DevStats ds;
    HIP_TRY(c, hipMemcpy(&ds, c->dev_stats.p, sizeof(ds), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> shards(RPT_STAT_SHARDS * RPT_STAT_STRIDE, 0ull);
    if (c->ray_shards.p) HIP_TRY(c, hipMemcpy(shards.data(), c->ray_shards.p, shards.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    ds.extension_rays = 0;
    unsigned long long elided = 0ull;
    for (int k = 0; k < RPT_STAT_SHARDS; ++k) { ds.extension_rays += shards[(size_t)k * RPT_STAT_STRIDE]; elided += shards[(size_t)k * RPT_STAT_STRIDE + 1]; }
    c->stats.extension_rays = ds.extension_rays;
    c->stats.shadow_rays = ds.shadow_rays + elided;     /* as the reference counts: one per executed intersect_any (light_pick.rs:141) */
    c->stats.shadow_rays_elided = elided;               /* of those, not walked: their NEE term is zero whatever the walk finds (k_shade.h) */
    c->stats.sky_evals = ds.sky_evals;
    c->stats.light_index_clamped = ds.light_index_clamped;
    *out = c->stats;
    if (ds.undrained != 0ull) { c->error = "a render call found samples still in flight (internal error)"; return RPT_EHIP; }
    return RPT_OK;
}

/* ------------------------------------------------------------ test hooks (include/rpt/rpt_debug.h) -- */
int rpt_debug_short_batch(rpt_ctx *c, int on) {
    if (!c) return RPT_EINVAL;
    c->test_short_batch = on != 0;
    return RPT_OK;
}

__global__ void k_debug_math(int op, const float *x, const float *y, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
        case 0: r = rptm::sinr(x[i]); break;
        case 1: r = rptm::cosr(x[i]); break;
        case 2: r = rptm::acosr(x[i]); break;
        case 3: r = rptm::expr(x[i]); break;
        case 4: r = rptm::powr(x[i], y[i]); break;
        case 5: r = rptm::asinr(x[i]); break;
        case 6: r = rptm::atan2r(x[i], y[i]); break;
        case 7: r = rptm::sqrtr(x[i]); break;
        case 9: r = rptm::slab_quotient(x[i], 0.0f, y[i]); break;
        case 10: r = rptm::exp_sky(x[i]); break;
        case 11: r = rptm::unorm8(x[i]); break;
        default: r = x[i] / y[i]; break;
    }
    out[i] = r;
}

int rpt_debug_math_host(int op, const float *x, const float *y, float *out, size_t n) {
    if (op < 0 || op > 11 || !x || !y || !out) return RPT_EINVAL;
    for (size_t i = 0; i < n; ++i) {
        float r;
        switch (op) {
            case 0: r = rptm::sinr(x[i]); break;
            case 1: r = rptm::cosr(x[i]); break;
            case 2: r = rptm::acosr(x[i]); break;
            case 3: r = rptm::expr(x[i]); break;
            case 4: r = rptm::powr(x[i], y[i]); break;
            case 5: r = rptm::asinr(x[i]); break;
            case 6: r = rptm::atan2r(x[i], y[i]); break;
            case 7: r = rptm::sqrtr(x[i]); break;
            case 9: r = rptm::slab_quotient(x[i], 0.0f, y[i]); break;
            case 10: r = rptm::exp_sky(x[i]); break;
            case 11: r = rptm::unorm8(x[i]); break;
            default: r = x[i] / y[i]; break;
        }
        out[i] = r;
    }
    return RPT_OK;
}

int rpt_debug_math(rpt_ctx *c, int op, const float *x, const float *y, float *out, size_t n) {
    if (!c || op < 0 || op > 11 || !x || !y || !out) return RPT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<float> dx, dy, dout;
    HIP_TRY(c, dx.alloc(n)); HIP_TRY(c, dy.alloc(n)); HIP_TRY(c, dout.alloc(n));
    hipError_t e = hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dy.p, y, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_debug_math<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(op, dx.p, dy.p, dout.p, n);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost);
    dx.release(); dy.release(); dout.release();
    HIP_TRY(c, e);
    return RPT_OK;
}

/* Exhaustive check of a cheap exact operation against its IEEE form, over the bit patterns [lo_bits, lo_bits + count):
 * op 0: rptm::sqrtr == the compiler's correctly rounded sqrtf (trivially, today: the hook experiments with cheaper roots used); op 1: rptm::div_const_nontiny(x, y, RN(1 / y)) == x / y;
 * op 2: rptm::f2i32_sat (one v_cvt_i32_f32) == Rust's `f32 as i32` written out with its branches (results compared as bit patterns). */
__global__ void k_debug_math_sweep(int op, uint32_t lo_bits, unsigned long long count, float y, float ry, unsigned long long *out) {
    unsigned long long bad = 0ull;
    uint32_t first = 0xffffffffu;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = lo_bits + (uint32_t)i;
        const float x = rptm::u2f(bits);
        const float fast = op == 0 ? rptm::sqrtr(x) : (op == 1 ? rptm::div_const_nontiny(x, y, ry) : rptm::u2f((uint32_t)rptm::f2i32_sat(x)));
        const float ieee = op == 0 ? __builtin_sqrtf(x) : (op == 1 ? x / y : rptm::u2f((uint32_t)rptm::f2i32_sat_reference(x)));
        const bool same = rptm::f2u(fast) == rptm::f2u(ieee) || (op != 2 && fast != fast && ieee != ieee);     /* (op 2 carries integers: bit patterns only) */
        if (!same) { bad += 1ull; first = first < bits ? first : bits; }
    }
    if (bad != 0ull) {
        atomicAdd(&out[0], bad);
        atomicMin(&out[1], (unsigned long long)first);
    }
}

int rpt_debug_math_sweep(rpt_ctx *c, int op, uint32_t lo_bits, uint64_t count, float y, uint64_t *mismatches_out, uint32_t *first_bad_bits_out) {
    if (!c || op < 0 || op > 2 || !mismatches_out || count > 0x100000000ull) return RPT_EINVAL;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<unsigned long long> d;
    HIP_TRY(c, d.alloc(2));
    unsigned long long h[2] = {0ull, 0xffffffffull};
    hipError_t e = hipMemcpy(d.p, h, sizeof(h), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_debug_math_sweep<<<4096, 256, 0, c->stream>>>(op, lo_bits, (unsigned long long)count, y, 1.0f / y, d.p);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(h, d.p, sizeof(h), hipMemcpyDeviceToHost);
    d.release();
    HIP_TRY(c, e);
    *mismatches_out = h[0];
    if (first_bad_bits_out) *first_bad_bits_out = (uint32_t)h[1];
    return RPT_OK;
}

int rpt_debug_bsdf(rpt_ctx *c, int kind, size_t n, const float *in, float *out) {
    if (!c || kind < 0 || kind > 3 || !in || !out) return RPT_EINVAL;
    if (n == 0) return RPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<float> din, dout;
    HIP_TRY(c, din.alloc(16 * n)); HIP_TRY(c, dout.alloc(8 * n));
    hipError_t e = hipMemcpy(din.p, in, 64 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_debug_bsdf<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(kind, n, din.p, dout.p);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout.p, 32 * n, hipMemcpyDeviceToHost);
    din.release(); dout.release();
    HIP_TRY(c, e);
    return RPT_OK;
}

int rpt_debug_trace_rays(rpt_ctx *c, int any_hit, size_t n, const float *origins, const float *dirs, const float *max_t,
                         float *out_t, uint32_t *out_tri, uint32_t *out_flags) {
    if (!c || !origins || !dirs || !out_t || !out_tri || !out_flags || (any_hit && !max_t)) return RPT_EINVAL;
    if (!c->has_scene) { c->error = "no scene"; return RPT_EINVAL; }
    if (n == 0) return RPT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    DevBuf<float> d_o, d_d, d_m, d_t;
    DevBuf<uint32_t> d_tri, d_fl;
    HIP_TRY(c, d_o.alloc(3 * n)); HIP_TRY(c, d_d.alloc(3 * n)); HIP_TRY(c, d_m.alloc(n)); HIP_TRY(c, d_t.alloc(n));
    HIP_TRY(c, d_tri.alloc(n)); HIP_TRY(c, d_fl.alloc(n));
    hipError_t e = hipMemcpy(d_o.p, origins, 12 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d.p, dirs, 12 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess && max_t) e = hipMemcpy(d_m.p, max_t, 4 * n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipStream_t s = c->stream;
        rpt_launch_trace_debug(c, any_hit != 0, (uint32_t)n, d_o.p, d_d.p, d_m.p, d_t.p, d_tri.p, d_fl.p);
        e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess) e = hipMemcpy(out_t, d_t.p, 4 * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_tri, d_tri.p, 4 * n, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_flags, d_fl.p, 4 * n, hipMemcpyDeviceToHost);
    d_o.release(); d_d.release(); d_m.release(); d_t.release(); d_tri.release(); d_fl.release();
    HIP_TRY(c, e);
    return RPT_OK;
}

/* The same question through the PRODUCTION nearest-hit stage: the rays are written into the context's own slots as pending
 * extension rays, the traversal stage is launched exactly as an iteration of rpt_render launches it for this scene and state
 * (launch_nearest: persistent LDS stream / streamed global-memory walk with or without cooperative leaves / one-shot kernels,
 * per the developer knobs), and the hit records it wrote are read back.  Needs a configuration (the slots); leaves the
 * context as after rpt_reset with nothing rendered — call rpt_reset before rendering again. */
__global__ __launch_bounds__(RPT_BLOCK) void k_debug_load_rays(DevState st, uint32_t n, const float *origins, const float *dirs) {
    const uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (i >= n) return;
    st.ray_a[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], dirs[3 * i]);
    st.ray_b[i] = make_float2(dirs[3 * i + 1], dirs[3 * i + 2]);
    st.hit[i] = make_float2(0.0f, __uint_as_float(HIT_PENDING));
}

int rpt_debug_trace_rays_production(rpt_ctx *c, size_t n, const float *origins, const float *dirs, float *out_t, uint32_t *out_tri, uint32_t *out_flags) {
    if (!c || !origins || !dirs || !out_t || !out_tri || !out_flags) return RPT_EINVAL;
    if (!c->has_scene || !c->has_state) { c->error = "rpt_debug_trace_rays_production: needs a scene and a configuration"; return RPT_EINVAL; }
    if (n == 0) return RPT_OK;
    if (n > c->n_slots) { c->error = "rpt_debug_trace_rays_production: more rays than the context has slots (" + std::to_string(c->n_slots) + ")"; return RPT_EINVAL; }
    int rc = rpt_wait(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    rc = ensure_slot_state(c, c->n_slots, false, false);
    if (rc) return rc;
    DevBuf<float> d_o, d_d;
    HIP_TRY(c, d_o.alloc(3 * n)); HIP_TRY(c, d_d.alloc(3 * n));
    HIP_TRY(c, hipMemcpy(d_o.p, origins, 12 * n, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(d_d.p, dirs, 12 * n, hipMemcpyHostToDevice));
    hipStream_t s = c->stream;
    HIP_TRY(c, hipMemsetAsync(c->q_count.p, 0, Q_WORDS * sizeof(uint32_t), s));
    k_fill_idle<<<(c->n_slots + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, s>>>(c->hit.p, c->n_slots);
    k_debug_load_rays<<<(unsigned)((n + RPT_BLOCK - 1) / RPT_BLOCK), RPT_BLOCK, 0, s>>>(c->state, (uint32_t)n, d_o.p, d_d.p);
    rpt_launch_nearest(c, 0u, false, false);
    std::vector<float2> hits(n);
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(hits.data(), c->hit.p, n * sizeof(float2), hipMemcpyDeviceToHost);
    /* back to "nothing in flight" */
    if (e == hipSuccess) e = hipMemsetAsync(c->q_count.p, 0, Q_WORDS * sizeof(uint32_t), s);
    k_fill_idle<<<(c->n_slots + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, s>>>(c->hit.p, c->n_slots);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    d_o.release(); d_d.release();
    HIP_TRY(c, e);
    for (size_t i = 0; i < n; ++i) {
        uint32_t w;
        memcpy(&w, &hits[i].y, 4);
        if (w == HIT_PENDING || w == HIT_IDLE) { c->error = "rpt_debug_trace_rays_production: ray " + std::to_string(i) + " was not traversed"; return RPT_EHIP; }
        out_t[i] = hits[i].x;
        out_tri[i] = (w == HIT_MISS) ? 0u : (w & 0x7fffffffu);
        out_flags[i] = (w == HIT_MISS) ? 0u : (1u | ((w >> 31) << 1));
    }
    return RPT_OK;
}

}  // extern "C"
