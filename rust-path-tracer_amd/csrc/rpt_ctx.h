/*
 * rpt_ctx.h — the context object behind the C ABI of include/rpt/rpt.h, shared by the translation units of
 * librpt_hip.so (rpt_hip.hip: upload + wavefront scheduling; rpt_comm.hip: multi-GPU gather over RCCL, read-back).
 */
#ifndef RPT_CTX_H
#define RPT_CTX_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/rpt/rpt.h"
#include "k_common.h"
#include "shadow_order.h"

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                      \
            return RPT_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (!count) return hipSuccess;
        return hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

constexpr int RPT_RING_LAG = 6;   /* most iterations the host may run ahead of the progress report it inspects (small launches) */
constexpr int RPT_RING = 16;      /* power of two, > RPT_RING_LAG */

struct rpt_comm;                  /* rpt_comm.hip: RCCL communicator + gather buffers of one context */

struct rpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    uint32_t rank = 0, world = 1;
    bool lds_stream = true;
    bool first_presub = true;            /* iteration 0 walks plane records with the camera position already subtracted (k_traverse.h FIRST); RPT_FIRST_PRESUB=0: the plain launch */
    bool shade_compact = false;          /* shade stage variant in use: traversed slots packed per workgroup before shading (k_shade<.., COMPACT>) */
    int shade_compact_mode = -1;         /* -1 automatic (refresh_device_stats), 0 / 1 forced by RPT_SHADE_COMPACT */
    double shade_compact_at = 0.7;       /* automatic: on when more than this share of the samples ends in the sky */
    bool lds_shadow_stream = true;       /* LDS scenes: streamed shadow stage (k_traverse_shadow_stream + k_shadow_resolve) */
    bool gstream = true;                 /* scenes walked from global memory: streamed kernels (k_traverse_*_gstream) */
    bool fat_leaves = false;             /* some leaf holds more than RPT_COOP_LEAF_MIN triangles: the streamed walks' wave-cooperative build
                                            (RPT_COOP_LEAVES=0/1 forces the other build for tests; both give the same image) */
    uint32_t gstream_min_waves = 32768;
    uint32_t stream_max_blocks = 512;    /* persistent workgroups of the streamed LDS traversal: 2 per CU (each holds 32 KB of stacks + the scene image) */
    uint32_t stream_span = 0;            /* slots a workgroup fetches at a time; 0 = automatic */
    int stack_bits_min = 16;             /* RPT_STACK_BITS: narrowest stack entry the streamed global-memory walks may use (16 / 24 / 32) */
    uint32_t sky_blocks = 4096;          /* grid of the strided sky stage */
    bool sky_strided = false;            /* sky stage variant in use: fixed grid + grid stride (few misses) or one thread per entry */
    int sky_strided_mode = -1;           /* -1 automatic, 0 / 1 forced by RPT_SKY_STRIDED */

    /* scene */
    bool has_scene = false;
    DevBuf<float4> nodes, tri_geom, tri_shade, tri_tangent, mat_lite, per_vertex, materials, lds_image, light_rec;
    DevBuf<uint4> indices;
    DevBuf<float> tri_isect;
    DevBuf<float4> gpairs;                     /* pair records + links of the streamed global-memory walks (k_traverse.h SceneViewPairsT) */
    DevBuf<uint32_t> glinks;
    DevBuf<float4> lds_image_shadow, gpairs_shadow;   /* the same tree with its pairs flipped for the fixed-order any-hit walks (shadow_order.h) */
    DevBuf<uint32_t> glinks_shadow;
    ShadowOrder shadow_order;
    DevBuf<float4> lds_image_last;                    /* pair records of the copy flipped for the hit-or-miss lanes of the last extension rays (choose_last_order) */
    LastOrder last_order;
    DevBuf<rpt_light_pick_entry> light_pick;
    DevBuf<uchar4> atlas;
    DevBuf<float4> skybox;
    DevScene scene{};
    uint32_t bvh_depth = 0;
    int stack_cap = 16;

    /* config + partition */
    bool has_config = false;
    DevConfig cfg{};
    uint32_t n_slots = 0;       /* n_pixels << group_shift */
    uint32_t n_pixels = 0;      /* pixels of this rank's tiles */
    uint32_t group_shift = 0;   /* log2 of the samples of one pixel kept in flight (of the current / last rpt_render call) */
    uint32_t max_group_shift = 0, max_slots = 0;   /* what the state arrays are sized for */
    bool sky_at_end_ok = true;           /* batches of known length shade their misses once, after the last iteration (RPT_SKY_AT_END=0: in every iteration) */
    int slot_q_shift_mode = -1;          /* -1 automatic (by scene size, render_impl), else log2 of the samples of one pixel that share a wave (RPT_SLOT_Q_SHIFT) */
    uint32_t sky_wide_cfg = 32768;
    int samples_in_flight_request = 0;   /* 0 = automatic */
    uint64_t max_slots_budget = 160ull << 20;   /* automatic S: the most slots (pixels x samples in flight) a context allocates */
    std::vector<uint32_t> pixel_xy_host;
    DevBuf<uint32_t> pixel_xy;

    /* state */
    bool has_state = false;
    DevBuf<float2> ray_b, hit;
    DevBuf<float4> ray_a, thr, rad, mis_a, mis_b, accum;
    DevBuf<uint2> rng;
    DevBuf<uint32_t> q_sky, q_count;
    DevBuf<unsigned long long> ray_shards;
    DevBuf<float4> sh_o, sh_d, sh_c;
    DevBuf<DevStats> dev_stats;
    DevState state{};
    DevQueues queues{};
    uint32_t samples = 0;
    uint32_t call_samples = 0;  /* n_samples of the current / last rpt_render call (the shade stage of its first iteration derives what a slot owes) */

    /* scheduling: the traversal kernel reports each iteration's queue size into mapped pinned memory */
    unsigned long long *host_ring = nullptr;       /* host view, RING entries */
    unsigned long long *host_ring_dev = nullptr;   /* device view of the same memory */

    /* stats */
    rpt_stats stats{};
    bool stage_timing = false;
    int timing_level = 0;           /* RPT_STAGE_TIMING: 1 = an event after every stage kernel, 2 = only around the traversal kernel */
    std::vector<hipEvent_t> timing_events;
    /* batches enqueued by rpt_render_async whose stage timing has not been read back yet */
    struct TimingBatch { std::vector<hipEvent_t> ev; uint64_t iterations; bool complete_timed; };
    std::vector<TimingBatch> timing_pending;
    std::vector<hipEvent_t> timing_pool;
    bool async_pending = false;

    /* read-back and multi-GPU gather (rpt_comm.hip) */
    rpt_comm *comm = nullptr;
    DevBuf<float4> image;                 /* row-major W x H accumulator image (device), built by k_untile */
    float *host_image = nullptr;          /* pinned twin of it: rpt_read_accum is one DMA */
    size_t host_image_floats = 0;
    DevBuf<uint32_t> untile_map;          /* rpt_untile: destination map, rebuilt only when (W, H, world, stride) changes */
    uint64_t untile_key = 0;
    uint32_t untile_n = 0;
};

/* workgroup size of the LDS-resident-scene traversal kernels: 2 x (32 KB of 16-bit stacks + up to 32 KB of scene = the 64 KB a
 * workgroup may hold) per CU = 32 waves */
#ifndef RPT_LDS_THREADS
#define RPT_LDS_THREADS 1024
#endif
/* rank-local slot order (rpt_hip.hip) */
void rpt_build_pixel_order(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, std::vector<uint32_t> &out);
/* rpt_comm.hip: called by rpt_hip.hip when the context / its state goes away */
void rpt_comm_release(rpt_ctx *c);
void rpt_image_release(rpt_ctx *c);
std::string &rpt_create_error();

#endif /* RPT_CTX_H */
