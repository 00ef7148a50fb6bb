/*
 * rpt_ctx.h — the context object behind the C ABI of include/rpt/rpt.h, shared by the translation units of
 * librpt_hip.so (rpt_hip.hip: upload + wavefront scheduling; rpt_comm.hip: multi-GPU gather over RCCL, read-back).
 */
#ifndef RPT_CTX_H
#define RPT_CTX_H

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rpt/rpt.h"
#include "../../include/rpt/rpt_debug.h"
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

/* Every environment variable the library reads, read in ONE place (rpt_read_knobs, rpt_hip.hip; rpt_create copies them into the context).  None changes
 * a result (tests/test_gpu_parity.py::test_developer_knobs_do_not_change_the_image); they exist for tests that must reach a code path a shipped scene
 * does not take, and for the bench's stage timing.  Rounds 1-5 had thirty — every A/B of a tuning round left one behind, with the losing code path kept alive
 * behind it; round 6 removed the knobs whose alternative lost with kept evidence (one-shot walks instead of the streamed ones, per-iteration sky
 * launches in batches of known length, deferred sky threshold, span / grid sizes of the streamed walks, IEEE-only division, ...) together with that code. */
struct rpt_knobs {
    int stage_timing = 0;             /* RPT_STAGE_TIMING    1: HIP events after every stage kernel (rpt_stats.kernel_ms); 2: around the traversal kernel only */
    bool upload_timing = false;       /* RPT_UPLOAD_TIMING   1: section times of rpt_upload_scene / rpt_bvh_build_gpu on stderr */
    int slot_q_shift = -1;            /* RPT_SLOT_Q_SHIFT    0..5: log2 of the samples of a pixel that share a wave (automatic: 5 for scenes of 2^19 triangles and more) */
    int shadow_order = -1;            /* RPT_SHADOW_ORDER    near | fixed: the any-hit walks' order instead of the probe's choice (shadow_order.h) */
    int last_order = -1;              /* RPT_LAST_ORDER      off | near | opaque | small | ratio: the last extension rays' walk instead of the probe's choice */
    int shade_compact = -1;           /* RPT_SHADE_COMPACT   0 | 1: the packed shade stage off / on instead of by the scene's own miss share */
    int sky_strided = -1;             /* RPT_SKY_STRIDED     0: one thread per queued miss; N >= 1: the strided sky stage (N > 1: with a grid of N workgroups) */
    int stack_bits = 16;              /* RPT_STACK_BITS      16 | 21 | 24 | 32: narrowest stack entry of the streamed global-memory walks (deep trees only) */
    int coop_leaves = -1;             /* RPT_COOP_LEAVES     0 | 1: the wave-cooperative leaf build of the streamed walks off / on instead of by leaf size */
    bool no_lds_scene = false;        /* RPT_NO_LDS_SCENE    1: walk a scene that would fit in LDS from global memory */
    int bvh_team_min = 0;             /* RPT_BVH_TEAM_MIN    rpt_bvh_build_gpu: smallest node split by a team of workgroups (0: the built-in threshold) */
    /* (RPT_RCCL_LIBRARY — the collective library to dlopen instead of librccl.so, test stand-in only — is read by rpt_comm.hip when the first communicator is made) */
};
rpt_knobs rpt_read_knobs();

struct rpt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string error;
    uint32_t rank = 0, world = 1;
    rpt_knobs knobs;                     /* copy of rpt_read_knobs() at rpt_create */
    bool shade_compact = false;          /* shade stage variant in use: traversed slots packed per workgroup before shading (k_shade<.., COMPACT>) */
    int shade_compact_mode = -1;         /* -1 automatic (refresh_device_stats), 0 / 1 forced by RPT_SHADE_COMPACT */
    double shade_compact_at = 0.7;       /* automatic: on when more than this share of the samples ends in the sky */
    bool fat_leaves = false;             /* some leaf holds more than RPT_COOP_LEAF_MIN triangles: the streamed walks' wave-cooperative build
                                            (RPT_COOP_LEAVES=0/1 forces the other build for tests; both give the same image) */
    uint32_t gstream_min_waves = 32768;
    uint32_t stream_max_blocks = 512;    /* persistent workgroups of the streamed LDS traversal: 2 per CU (each holds 32 KB of stacks + the scene image) */
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
    uint32_t sky_wide_cfg = 32768;       /* up to this many queued misses the sky march runs 16 lanes per miss */
    bool test_short_batch = false;       /* rpt_debug_short_batch: enqueue one iteration too few in an asynchronous batch (proves that the completion checks notice) */
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
/* rpt_traverse.hip: the traversal stages (which walk kernel for the context's scene, on which grid) */
void rpt_launch_nearest(rpt_ctx *c, uint32_t iteration, bool last_without_nee /* the last extension rays of a batch of known length, no NEE */,
                        bool camera_rays /* iteration 0 of a render call: every ray leaves cfg.cam_position */);
void rpt_launch_shadow(rpt_ctx *c);
void rpt_launch_trace_debug(rpt_ctx *c, bool any_hit, uint32_t n, const float *origins, const float *dirs, const float *max_t, float *out_t, uint32_t *out_tri,
                            uint32_t *out_flags);
hipError_t rpt_last_walk_attributes(hipFuncAttributes *out);      /* of k_traverse_nearest_stream<.., LAST>: its static LDS decides whether the flipped copy fits */
/* rank-local slot order (rpt_hip.hip) */
void rpt_build_pixel_order(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, std::vector<uint32_t> &out);
/* rpt_comm.hip: called by rpt_hip.hip when the context / its state goes away */
void rpt_comm_release(rpt_ctx *c);
void rpt_image_release(rpt_ctx *c);
std::string &rpt_create_error();

/* RPT_UPLOAD_TIMING=1: host-side section times of rpt_upload_scene / rpt_bvh_build_gpu on stderr (where the start-up time of a large scene goes) */
struct SectionTimer {
    bool on;
    const char *title;
    std::chrono::steady_clock::time_point last;
    std::string line;
    explicit SectionTimer(const char *t) : on(false), title(t) {
        on = rpt_read_knobs().upload_timing;
        last = std::chrono::steady_clock::now();
    }
    void mark(const char *name) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[96];
        snprintf(buf, sizeof buf, " %s %.1f", name, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    ~SectionTimer() { if (on) fprintf(stderr, "%s (ms):%s\n", title, line.c_str()); }
};

#endif /* RPT_CTX_H */
