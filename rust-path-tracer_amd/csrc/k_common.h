/*
 * k_common.h — device-side data layout of the wavefront path tracer.
 *
 * One "slot" per pixel owned by this GPU.  A slot carries one path at a time
 * (sample s of its pixel must finish before sample s+1 starts, so the f32
 * accumulation order per pixel equals the reference's sample order,
 * kernels/src/lib.rs:225 / src/trace.rs:295).  All per-slot state is SoA in
 * 16-byte records so a wave reads 1 KiB per load instruction.
 *
 * Several samples of one pixel may be in flight at once (S = 2^group_shift slots
 * per pixel, 64 slots apart: slot_pix / slot_k below) so that a GPU owning only a few
 * tiles still fills its 256 CUs.  Slot k takes samples k, k+S, k+2S, ...; the S samples
 * of a "generation" are summed into the accumulator in k order by the pixel's thread once
 * all of them have finished, which reproduces the reference's sample-order f32 sum
 * bit for bit whatever S is (k_complete.h).
 *
 * Extension rays need no queue: a finished path is regenerated in place, so
 * (except in the last few iterations of a render call) every slot always has
 * a ray in flight and thread i simply owns slot i — perfectly coalesced, no
 * index indirection, no atomics, and a wave is one 8x8 pixel block for ever.
 * The slot's stage is the hit word of its ray record (HIT_PENDING -> traverse,
 * a hit / HIT_MISS -> shade, HIT_PARKED -> nothing).
 * The two side stages that only SOME paths need — sky shading of misses and
 * shadow rays — are fed by compacted queues built with wave64 ballot + mbcnt
 * prefix sums, aggregated per workgroup through LDS so that one atomic serves
 * 256 lanes (a single queue counter sustains only ~90 returning atomics/us on
 * MI355X — MI355X_MICROARCH.md "dequeue" — which at one atomic per wave was
 * the measured bottleneck of the first version of the shade stage, and at one
 * per workgroup still cost a third of a late shade launch: the queues are
 * sharded, RPT_Q_SHARDS below).
 */
#ifndef RPT_K_COMMON_H
#define RPT_K_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rpt/shared_structs.h"
#include "rpt_math.h"

#define RPT_WAVE 64
#define RPT_BLOCK 256
#define RPT_MAX_SAMPLES_IN_FLIGHT 256u   /* most slots per pixel (k_complete.h counts a pixel's finished slots; rounds 1-5: a 32-bit mask) */

/* ---- glam-order float3 helpers (device side) ----------------------------- */
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 f3s(float s) { return F3{s, s, s}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
/* Vec3::dot = (x*x' + y*y') + z*z' ; cross, length, normalize = v * (1/len) */
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) {
    return F3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
__device__ __forceinline__ float len3(F3 a) { return rptm::sqrtr(dot3(a, a)); }
__device__ __forceinline__ F3 norm3(F3 a) { return a * (1.0f / len3(a)); }
__device__ __forceinline__ F3 lerp3(F3 a, F3 b, float s) { return a + ((b - a) * s); }
__device__ __forceinline__ bool finite3(F3 a) { return rptm::finiter(a.x) && rptm::finiter(a.y) && rptm::finiter(a.z); }
/* util.rs:271-277: v if finite, else zero — as ONE select of a bit mask and three ands: three selects on one condition compile to
 * back-to-back VOP2 v_cndmask on vcc, which gfx950 issues at ~30 cycles each (tools/microbench/valu_rates.hip) */
__device__ __forceinline__ F3 mask_nan3(F3 a) {
    uint32_t m = finite3(a) ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));          /* opaque: the optimiser would fold "x & (c ? ~0 : 0)" back into three selects */
    return F3{rptm::u2f(rptm::f2u(a.x) & m), rptm::u2f(rptm::f2u(a.y) & m), rptm::u2f(rptm::f2u(a.z) & m)};
}
__device__ __forceinline__ F3 xyz4(float4 v) { return F3{v.x, v.y, v.z}; }

/* ---- scene (read-only) ---------------------------------------------------- */
struct DevImage {
    const void *texels;    /* uchar4 (atlas) or float4 (skybox) */
    uint32_t width, height;
};

/* wave64 ballot of a bool that already lives in an SGPR mask: HIP's __ballot(int) first materialises the
 * predicate as 0/1 in a VGPR and compares it again (two extra VALU instructions per ballot). */
__device__ __forceinline__ unsigned long long rpt_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }

#define RPT_LAST_EMIT_MAX 4
struct DevScene {
    const float4 *nodes;           /* 2 x float4 per rpt_bvh_node, reference layout */
    const float4 *tri_geom;        /* 3 x float4 per triangle: (a | d00), (e1 = b-a | d01), (e2 = c-a | d11);
                                      dNN = the triangle-constant dot products of util::barycentric */
    const float *tri_isect;        /* 9 floats per triangle: e1, e2, a — what the intersection test reads, packed (36 instead of 48 bytes:
                                      the walk of a 1 M-triangle scene is bound by these bytes, profiles/r02_deepbvh_*) */
    const float4 *tri_shade;       /* 4 x float4 per triangle: (na | uva.x) (nb | uva.y) (nc | material) (uvb, uvc) */
    const float4 *tri_tangent;     /* 3 x float4 per triangle: the three vertex tangents (lib.rs:135-138), only where a material has a normal map — one 48-byte record
                                      instead of the index record + three 64-byte vertices */
    const float4 *mat_lite;        /* 2 x float4 per material: (emissive.rgb | roughness.x) (albedo.rgb | metallic.x) */
    const uint4 *indices;          /* rpt_triangle */
    const float4 *per_vertex;      /* 4 x float4 per rpt_per_vertex_data */
    const float4 *materials;       /* 6 x float4 per rpt_material_data */
    const rpt_light_pick_entry *light_pick;
    const float4 *light_rec;       /* 8 x float4 per light-pick entry: for its triangle a, then b: (A | n.x) (B | n.y) (C | n.z) (emission | -),
                                      n = the mean vertex normal of light_pick.rs:129 — what sample_direct_lighting gathers through the
                                      index buffer, three 64-byte vertices and the material, in one 64-byte record */
    uint32_t n_light_pick;
    uint32_t n_nodes, n_triangles;
    uint32_t lds_scene;            /* the traversal image fits in RPT_LDS_SCENE_BYTES: traverse out of LDS */
    const float4 *lds_image;       /* LDS-resident traversal image (k_traverse.h SceneViewLds), lds_vecs float4 */
    uint32_t lds_pairs, lds_vecs, lds_root;
    const float4 *gpairs;          /* streamed global-memory walks: one 64-byte record per child pair (k_traverse.h SceneViewPairsT), or null */
    const uint32_t *glinks;        /* triangle_count << 24 | left child / first triangle, per node */
    uint32_t shadow_fixed;         /* the any-hit walks use a fixed left-first order over the flipped copies below (shadow_order.h) */
    const float4 *lds_image_shadow;
    const float4 *gpairs_shadow;
    const uint32_t *glinks_shadow;
    /* the last extension rays of a batch without NEE (k_traverse.h k_traverse_nearest_stream LAST): the triangles whose material emits, when there are at most
     * RPT_LAST_EMIT_MAX of them (last_emit_n > RPT_LAST_EMIT_MAX: too many, the launch is the plain one), and the size of the flipped copy's pair records
     * staged behind the LDS image (0: not staged) */
    uint32_t last_emit_n, last_emit_tri[RPT_LAST_EMIT_MAX];
    uint32_t last_flip_vecs;
    const float4 *lds_image_last;  /* those pair records (6 x lds_pairs plane float4 + the child descriptors), flipped by the rule shadow_order.h choose_last_order picked */
    uint32_t no_lights;            /* light_pick[0].ratio < 0 */
    uint32_t fastdiv_ok;           /* every node bound is 0 or in [2^-60, 2^40): exact fast division allowed */
    uint32_t textured;             /* some material has a texture flag set */
    DevImage atlas, skybox;
};

/* constants of the traversal structures that the upload code (rpt_hip.hip) and the walk kernels (k_traverse.h, compiled in rpt_traverse.hip) share */
#define LDS_DESC_DEAD 0x4000u      /* 16-bit child descriptor of the LDS image (k_traverse.h SceneViewLds): pair index, or LEAF | count << 9 | first triangle */
#define LDS_DESC_LEAF 0x8000u
#define RPT_LDS_SCENE_BYTES 32768  /* a traversal image up to this size lives in LDS (+ 32 KB of 16-bit stacks = the 64 KB of one of two workgroups per CU) */
#define RPT_COOP_LEAF_MIN 6        /* leaves with more triangles than this are tested by the whole wave (global-memory scenes) */

/* ---- per-slot path state (SoA of float4 records) -------------------------- */
struct DevState {
    float4 *ray_a;        /* (ox, oy, oz, dx) */
    float2 *ray_b;        /* (dy, dz) */
    float2 *hit;          /* (hit_t, stage / hit word: triangle | backface << 31, or one of the HIT_* states below).  Its own dense
                             array: the traversal stage writes exactly these 8 bytes per ray, as full 512-byte wave stores
                             (as the upper half of a 16-byte ray record they were strided partial-sector writes: WRITE_SIZE 2.4 x) */
    float4 *thr;          /* (thr.r, thr.g, thr.b, flags bits): what every bounce reads and rewrites */
    float4 *rad;          /* (rad.r, rad.g, rad.b, todo bits): touched only where radiance is added or a path ends — the shade stage
                             moves 16 B of path state per slot and pass each way instead of 32 (it runs at ~5 TB/s) */
    float4 *mis_a;        /* (light-table entry * 2 + side bits, thr_pre.r, thr_pre.g, thr_pre.b)   nee == MIS only: last_light_sample */
    float4 *mis_b;        /* (bsdf_pdf, spec.r, spec.g, spec.b)                                       last_bsdf_sample          */
    /* per PIXEL (pixel = slot >> group_shift): */
    uint2 *rng;           /* (n, offset), the reference's rng buffer */
    float4 *accum;        /* (sum r, sum g, sum b, sum 1), tile-major == pixel order */
    const uint32_t *pixel_xy;   /* x | y << 16 */
    uint32_t n_slots;
    uint32_t n_pixels;
    uint32_t group_shift; /* log2 S, S = samples of one pixel in flight (slots per pixel) */
    uint32_t q_shift;     /* log2 Q <= group_shift, Q = samples of one pixel in ONE wave (a wave = 64 / Q pixels x Q samples) */
};

/* flags word: bits 0-7 bounce, bit 8 last sampled lobe (1 = specular), bits 16-21 LDS dimension */
#define FLAG_BOUNCE(f) ((f) & 0xffu)
#define FLAG_LOBE_SPEC(f) (((f) >> 8) & 1u)
#define FLAG_DIM(f) (((f) >> 16) & 0x3fu)
#define MAKE_FLAGS(bounce, spec, dim) (((bounce) & 0xffu) | ((uint32_t)(spec) << 8) | ((uint32_t)(dim) << 16))
#define HIT_MISS 0xffffffffu      /* traversed, nothing hit                          -> shade sends it to the sky queue */
#define HIT_PENDING 0xfffffffeu   /* a ray is waiting for the traversal stage                                          */
#define HIT_PARKED 0xfffffffdu    /* waiting in the sky / shadow queue                                                 */
#define HIT_DONE 0xfffffffcu      /* sample finished, radiance final in rad.xyz; waits for its siblings */
#define HIT_IDLE 0xfffffffbu      /* the slot has no sample left to take in this render call                           */

/* ---- queues ---------------------------------------------------------------- */
/* counter words: per iteration parity, "the traversal pass traced a ray" and
 * "the shade pass started a new sample" (together with a non-empty sky queue: work remains) */
/* each word sits in its own 128-byte line: queue counters take atomics while flags are polled and raised
 * by every wave, and sharing a line made the two serialise against each other in L2 */
#define Q_LINE 32
enum { /* lines 0 and 1: free (the side-queue counters are sharded, DevQueues::sky_cnt / shadow_cnt) */ Q_ALIVE0 = 2 * Q_LINE, Q_ALIVE1 = 3 * Q_LINE, Q_REGEN0 = 4 * Q_LINE,
       Q_REGEN1 = 5 * Q_LINE, Q_DRAINED = 6 * Q_LINE, Q_POOL0 = 7 * Q_LINE, Q_POOL1 = 8 * Q_LINE, Q_SPOOL = 9 * Q_LINE, Q_COUNT = 12 * Q_LINE };
/* Q_POOL0/1: per iteration parity, the next unclaimed slot of the streamed traversal launch (k_traverse_nearest_stream);
 * Q_SPOOL: the next unclaimed entry of the streamed shadow launch (zeroed by the shade stage that fills the queue) */
/* Q_DRAINED: set by the sky stage of the first iteration that found nothing left; the host runs a few iterations
 * ahead of the progress report, and every stage of those surplus launches returns on this word at once. */
/* The two side queues are SHARDED.  A queue reservation is a device-scope atomic with return on one word, and those execute one
 * after the other at the memory side — measured ~4 ns each (profiles/r03_shade_queue_atomics.txt): with one reservation per shade
 * workgroup, 96 k of them in the last shade launch of a DarkCornell batch cost 0.43 ms of a 1.2 ms launch.  So workgroup b
 * reserves in shard b % 16, whose counter lives in its own 4 KB of memory, and the shards are interleaved in chunks of 256
 * entries: entry e of shard s sits at position ((e / 256) * 16 + s) * 256 + e % 256.  As the shards fill evenly (neighbouring
 * workgroups alternate) the positions stay dense up to the tails of the shards; consumers sweep q_extent() positions and skip the
 * few that are not filled (q_filled: bit arithmetic + one cached counter load, uniform per wave). */
#define RPT_Q_SHARDS 16
#define RPT_Q_SHARD_STRIDE 1088  /* words between shard counters (4 352 bytes: different channels under 256 B and 4 KB interleaving alike) */
#define RPT_Q_SLACK (RPT_Q_SHARDS * (2048u + 256u))   /* positions beyond the slot count a queue array must hold (a shard's share of the
                                                        workgroups rounded up, by at most one workgroup of 2 048 slots, + its last chunk) */
__host__ __device__ __forceinline__ uint32_t q_position(uint32_t shard, uint32_t e) { return ((((e >> 8) * RPT_Q_SHARDS) | shard) << 8) | (e & 255u); }
__device__ __forceinline__ bool q_filled(const uint32_t *cnt, uint32_t p) {
    return ((((p >> 8) / RPT_Q_SHARDS) << 8) | (p & 255u)) < cnt[((p >> 8) % RPT_Q_SHARDS) * RPT_Q_SHARD_STRIDE];
}
/* positions a consumer has to sweep, and the entries among them */
__device__ __forceinline__ void q_extent(const uint32_t *cnt, uint32_t &positions, uint32_t &total) {
    uint32_t most = 0u, sum = 0u;
    for (uint32_t s = 0; s < RPT_Q_SHARDS; ++s) {
        const uint32_t c = cnt[s * RPT_Q_SHARD_STRIDE];
        most = c > most ? c : most;
        sum += c;
    }
    positions = ((most + 255u) >> 8) * (256u * RPT_Q_SHARDS);
    total = sum;
}
__device__ __forceinline__ void q_clear(uint32_t *cnt) {
    for (uint32_t s = 0; s < RPT_Q_SHARDS; ++s) cnt[s * RPT_Q_SHARD_STRIDE] = 0u;
}

#define Q_WORDS ((size_t)Q_COUNT + 2u * RPT_Q_SHARDS * RPT_Q_SHARD_STRIDE)   /* the counter allocation: Q_COUNT words, then the shard counters of both queues */
#define RPT_STAT_SHARDS 64        /* sharded 64-bit counters, one 128-byte line each */
#define RPT_STAT_STRIDE 16

struct DevQueues {
    uint32_t *sky;
    float4 *sh_o;      /* shadow ray (ox, oy, oz, max_t), indexed by shadow-queue position */
    float4 *sh_d;      /* (dx, dy, dz, slot bits | bit31 = path ends after this NEE) */
    float4 *sh_c;      /* (contribution r, g, b if unoccluded, unused) */
    uint32_t *count;   /* Q_COUNT words, see the enum above */
    uint32_t *sky_cnt, *shadow_cnt;   /* RPT_Q_SHARDS counters each, RPT_Q_SHARD_STRIDE words apart: entries reserved per shard */
    uint32_t sky_threshold;   /* the sky stage runs once this many misses are queued (or nothing else is left) */
    uint32_t known_length;    /* the call enqueues exactly max_bounces iterations (no slot takes a second sample): every traversed path of iteration i is at bounce i */
    uint32_t sky_at_end;      /* a batch of known length: misses only END paths, so they wait in the queue for ONE sky launch after the last iteration */
    uint32_t sky_wide_limit;  /* up to this many queued misses the sky march runs 16 lanes per miss */
    unsigned long long *ray_shards;  /* RPT_STAT_SHARDS x RPT_STAT_STRIDE: extension rays traced */
    unsigned long long *host_ring;   /* mapped pinned host memory: (iteration + 1) << 32 | extension-queue size */
    uint32_t ring_mask;
};

struct DevStats {
    unsigned long long extension_rays, shadow_rays, sky_evals, light_index_clamped, samples;
    unsigned long long undrained;   /* slots a render call found (or left) with a sample still in flight: must stay 0 (k_generate_first, k_check_drained) */
};

/* per-render constants derived from TracingConfig on the host */
struct DevConfig {
    rpt_tracing_config c;
    float euler[9];        /* column-major RotY(cam_rotation.y) * RotX(cam_rotation.x) (kernels/src/lib.rs:50) */
    float sky_rot[9];      /* RotY(atan2(sun.z, sun.x)) for the image skybox (lib.rs:72-73) */
    uint32_t nee_mode;     /* NextEventEstimation::from_u32 */
};

/* Slot <-> (pixel, k).  The slots of 64 consecutive pixels (an 8 x 8 pixel block of the tile-major pixel order) form a CHUNK of
 * 64 x S slots laid out sample-major: slot = chunk * 64 S + k * 64 + (pixel % 64).  A wave of 64 consecutive slots is 64
 * PIXELS at one sample index k (coherent primary rays), and the S samples of a pixel sit 64 slots apart — so the in-order sum of
 * a finished generation is a loop over k in ONE lane per pixel with coalesced loads (k_complete), not a chain across the lanes
 * of a wave.  Pixels are padded to a multiple of 64; the padding slots stay idle.  S = 1: slot == pixel.
 * General form (q_shift > 0): a wave holds 64 / Q pixels x Q consecutive samples of each (lane = pixel-in-wave * Q + k % Q; the
 * waves of a chunk are ordered by pixel group, then k / Q).  Q = 1 is the layout above and what every shipped scene uses; rays
 * of ONE pixel are the most coherent a wave can get, which only pays where a ray's walk is long: measured (profiles/
 * r03_slot_layout.txt) the 2 M-node stand-in's traversal 605 -> 563 ms per 2 batches at Q = 32 (deep BVH stand-in 388 -> 377),
 * VeachMIS / PBRTest +-0, DarkCornell's first launch 1.16 -> 1.12 ms but its later bounces 2.9 -> 3.0 — while k_complete's loads
 * stop being coalesced (0.2 -> 1.4 ms per 33 M slots).  So Q = 32 for scenes of RPT_BIG_SCENE_TRIANGLES and more, where a batch
 * takes hundreds of milliseconds, and 1 elsewhere; the image does not depend on it (tests: RPT_SLOT_Q_SHIFT). */
#define RPT_BIG_SCENE_TRIANGLES (1u << 19)
__device__ __forceinline__ uint32_t slot_k(const DevState &st, uint32_t slot) {
    const uint32_t gs = st.group_shift, qs = st.q_shift;
    const uint32_t wave = (slot >> 6) & ((1u << gs) - 1u);
    return ((wave & ((1u << (gs - qs)) - 1u)) << qs) | (slot & ((1u << qs) - 1u));
}
__device__ __forceinline__ uint32_t slot_pix(const DevState &st, uint32_t slot) {
    const uint32_t gs = st.group_shift, qs = st.q_shift;
    const uint32_t wave = (slot >> 6) & ((1u << gs) - 1u);
    return ((slot >> (6u + gs)) << 6) | ((wave >> (gs - qs)) << (6u - qs)) | ((slot & 63u) >> qs);
}
__device__ __forceinline__ uint32_t pix_slot(const DevState &st, uint32_t pix, uint32_t k) {
    const uint32_t gs = st.group_shift, qs = st.q_shift;
    const uint32_t p = pix & 63u, pg = p >> (6u - qs), pl = p & ((64u >> qs) - 1u);
    return ((pix >> 6) << (6u + gs)) | ((((pg << (gs - qs)) | (k >> qs))) << 6) | (pl << qs) | (k & ((1u << qs) - 1u));
}

/* wave64 ballot + prefix compaction: every lane calls it (converged); lanes with
 * pred get a dense index in the queue, one atomic per wave. */
__device__ __forceinline__ uint32_t wave_push(uint32_t *counter, bool pred) {
    unsigned long long mask = rpt_ballot(pred);
    if (mask == 0ull) return 0u;
    uint32_t lane = __lane_id();
    uint32_t total = (uint32_t)__popcll(mask);
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1u;
    uint32_t base = 0u;
    if (lane == leader) base = atomicAdd(counter, total);
    base = (uint32_t)__shfl((int)base, (int)leader, RPT_WAVE);
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    return base + prefix;
}

/* Raise a kernel-wide boolean.  Thousands of waves storing to ONE address serialise in L2 (measured: a
 * per-wave plain store of 1 to a single word cost 36..150 us per launch of the shade stage), so look first:
 * after the first few writers everybody reads 1 from L1/L2 and skips the store.  (L1 is invalidated at
 * kernel start, so a stale 0 only costs a redundant store, never a lost flag.) */
__device__ __forceinline__ void raise_flag(uint32_t *flag) {
    if (*flag == 0u) *flag = 1u;
}

/* Workgroup-aggregated variant: one atomic per 256 lanes.  Every thread of the
 * block must call it (two barriers inside).  `scratch` is RPT_BLOCK/RPT_WAVE + 1 words of LDS.
 * `counter` is the workgroup's shard of a side queue; the return value is the entry's index WITHIN the shard (q_position). */
__device__ __forceinline__ uint32_t block_push(uint32_t *counter, bool pred, uint32_t *scratch) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    constexpr uint32_t NW = RPT_BLOCK / RPT_WAVE;
    unsigned long long mask = rpt_ballot(pred);
    if (lane == 0u) scratch[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t total = 0u;
        for (uint32_t w = 0; w < NW; ++w) total += scratch[w];
        scratch[NW] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t base = scratch[NW];
    for (uint32_t w = 0; w < wave; ++w) base += scratch[w];
    uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    __syncthreads();     /* scratch may be reused by the next push */
    return base + prefix;
}

/* ---- stateless LDS sequence (kernels/src/rng.rs:20-32) --------------------- */
static __device__ __constant__ uint32_t c_lds_primes[32] = {
    0x6a09e667u, 0xbb67ae84u, 0x3c6ef372u, 0xa54ff539u, 0x510e527fu, 0x9b05688au, 0x1f83d9abu, 0x5be0cd18u,
    0xcbbb9d5cu, 0x629a2929u, 0x91590159u, 0x452fecd8u, 0x67332667u, 0x8eb44a86u, 0xdb0c2e0bu, 0x47b5481du,
    0xae5f9155u, 0xcf6c85d1u, 0x2f73477du, 0x6d1826cau, 0x8b43d455u, 0xe360b595u, 0x1c456002u, 0x6f196330u,
    0xd94ebeafu, 0x9cc4a611u, 0x261dc1f2u, 0x5815a7bdu, 0x70b7ed67u, 0xa1513c68u, 0x44f93634u, 0x720dcdfcu};

struct Rng {
    uint32_t key;     /* n + offset (wrapping) */
    uint32_t dim;
    __device__ __forceinline__ float next() {
        dim += 1u;
        uint32_t v = c_lds_primes[dim & 31u] * key;
        return (float)v * (1.0f / 4294967296.0f);
    }
};

#endif /* RPT_K_COMMON_H */
