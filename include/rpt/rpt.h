/*
 * rpt.h — C ABI of librpt_hip.so, the MI355X (gfx950) wavefront path tracer
 * that drops in behind the reference's render dispatch.
 *
 * Boundary being replaced: `trace_gpu` in the reference, src/trace.rs:136-224,
 * which today drives one SPIR-V megakernel through gpgpu-rs/wgpu.  Each entry
 * point below cites the reference call(s) it stands in for.  All functions are
 * `extern "C"`, take plain pointers and sizes in the `shared_structs` layouts
 * (include/rpt/shared_structs.h), never throw, never abort, never call exit():
 * return 0 on success or a negative RPT_E* code, with a message available from
 * rpt_last_error().
 *
 * Ownership: the caller owns every host pointer passed in; the library copies
 * during the call.  Device memory belongs to the rpt_ctx and is released by
 * rpt_destroy().  A context is single-caller (the reference uses one render
 * thread, src/app.rs:157-164); distinct contexts may live on distinct threads.
 */
#ifndef RPT_H
#define RPT_H

#include <stddef.h>
#include <stdint.h>
#include "shared_structs.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RPT_ABI_VERSION 3

enum {
    RPT_OK = 0,
    RPT_EINVAL = -1,     /* bad argument / call order                             */
    RPT_ENODEV = -2,     /* no usable HIP device                                   */
    RPT_EHIP = -3,       /* a HIP runtime call failed (message has the HIP text)   */
    RPT_ECONFIG = -4,    /* config the reference itself would panic on (see below) */
    RPT_ESCENE = -5,     /* scene buffers inconsistent (index out of range, BVH too deep) */
    RPT_ENOMEM = -6
};

typedef struct rpt_ctx rpt_ctx;

/* Counters of one context since the last rpt_reset().  Ray counting rule
 * (SURVEY.md §8d): one extension ray per executed intersect_nearest
 * (kernels/src/lib.rs:63), one shadow ray per executed intersect_any
 * (kernels/src/light_pick.rs:141). */
typedef struct rpt_stats {
    uint64_t samples;          /* pixel-samples accumulated (W*H*spp on a full image)   */
    uint64_t extension_rays;
    uint64_t shadow_rays;
    uint64_t sky_evals;        /* misses shaded by the procedural / image sky             */
    uint64_t light_index_clamped; /* gen_r1()==1.0 alias-table overruns clamped (Appendix C) */
    uint64_t iterations;       /* wavefront iterations executed                           */
    double   render_ms;        /* wall time of rpt_render calls, host clock               */
    double   kernel_ms[8];     /* HIP-event time per stage: see RPT_STAGE_*               */
    uint64_t kernel_launches[8];
    uint64_t shadow_rays_elided; /* of shadow_rays: NEE evaluations whose shadow ray was not walked because the term it gates is zero
                                    whatever the walk finds (light_pdf = 0 or bsdf_pdf = 0, light_pick.rs:150-158); rays traced on the
                                    device = extension_rays + shadow_rays - shadow_rays_elided */
} rpt_stats;

enum {
    RPT_STAGE_GENERATE = 0,    /* camera rays + accumulate/regenerate */
    RPT_STAGE_TRAVERSE = 1,    /* nearest-hit BVH traversal           */
    RPT_STAGE_SHADE = 2,       /* material, BSDF sample, NEE setup    */
    RPT_STAGE_SHADOW = 3,      /* any-hit traversal + NEE resolve     */
    RPT_STAGE_SKY = 4,         /* miss shading                        */
    RPT_STAGE_COMPLETE = 5,    /* a batch's one completion pass: radiances of a generation added to the accumulator in sample order */
    RPT_STAGE_COUNT = 6
};

/* Replaces the lazy wgpu framework/adaptor creation, src/trace.rs:3-6,25-38.
 * device_id is a HIP ordinal. */
int rpt_create(int device_id, rpt_ctx **out);

/* Multi-GPU tile partition (no reference equivalent: the reference is single
 * device).  The W x H framebuffer is cut into RPT_TILE x RPT_TILE pixel tiles;
 * tile t belongs to rank (t mod world_size).  Must be called before
 * rpt_set_config; default is rank 0 of 1 (whole image). */
#define RPT_TILE 64
int rpt_set_partition(rpt_ctx *ctx, uint32_t rank, uint32_t world_size);

/* Tuning knob without reference equivalent: how many samples of one pixel are kept
 * in flight (rounded up to a power of two, <= 32).  0 = automatic (enough to give
 * the GPU ~0.75 M concurrent paths when this rank owns few pixels).  The result
 * does not depend on it: finished samples are added to a pixel's accumulator in
 * sample order whatever the value.  Invalidates the accumulator like a resize:
 * call before rpt_set_config or re-run rpt_reset afterwards. */
int rpt_set_samples_in_flight(rpt_ctx *ctx, int samples);

/* Replaces World::into_gpu (src/asset.rs:226-235), BVH::into_gpu
 * (src/bvh.rs:40-43) and the skybox upload (src/trace.rs:144).  Buffers are in
 * the order the reference binds them (kernels/src/lib.rs:195-202).  atlas and
 * skybox may be NULL (no textures / procedural sky).  atlas is RGBA8 as in
 * asset.rs:231 and is sampled with the CPU polyfill's semantics
 * (shared_structs/src/image_polyfill.rs:38-55; texel = u8/255, alpha = 1,
 * src/asset.rs:266-273).  skybox is float RGBA, same sampler. */
int rpt_upload_scene(rpt_ctx *ctx,
                     const rpt_per_vertex_data *per_vertex, size_t n_vertices,
                     const rpt_triangle *indices, size_t n_triangles,
                     const rpt_bvh_node *nodes, size_t n_nodes,
                     const rpt_material_data *materials, size_t n_materials,
                     const rpt_light_pick_entry *light_pick, size_t n_light_pick,
                     const uint8_t *atlas_rgba8, uint32_t atlas_w, uint32_t atlas_h,
                     const float *skybox_rgba32f, uint32_t sky_w, uint32_t sky_h);

/* Replaces GpuUniformBuffer::from_slice / config_buffer.write
 * (src/trace.rs:168,219).  Rejects with RPT_ECONFIG what the reference's CPU
 * path would panic on: more LDS dimensions than the 32-entry table
 * (kernels/src/rng.rs:20-21), i.e. 2 + max_bounces*(3 + 4*[nee!=0]) +
 * max(0, max_bounces-1-min_bounces) > 31. Changing width/height invalidates
 * the accumulator: call rpt_reset afterwards. */
int rpt_set_config(rpt_ctx *ctx, const rpt_tracing_config *config);

/* Replaces rng_buffer / output_buffer creation and the flush path
 * (src/trace.rs:164-170, 219-221).  rng_seed has width*height entries, pixel
 * index y*width + x; accum_init_rgba (nullable) is width*height float4 =
 * mean * samples for resume (src/trace.rs:163-164), with samples_init the
 * matching sample count. */
int rpt_reset(rpt_ctx *ctx, const rpt_rng_state *rng_seed,
              const float *accum_init_rgba, uint32_t samples_init);

/* Replaces the `for _ in 0..sync_rate { enqueue; poll_blocking }` loop
 * (src/trace.rs:182-194) — one call renders n_samples more samples for every
 * pixel of this rank's tiles with no per-sample host round trip.  Per pixel the
 * effect is exactly n_samples times `output[i] += (radiance, 1); rng[i].x += 1`
 * (kernels/src/lib.rs:225-226) in sample order. Synchronous on return. */
int rpt_render(rpt_ctx *ctx, uint32_t n_samples);

/* rpt_render that returns as soon as the batch is enqueued — possible when its iteration count is known up front,
 * i.e. when no slot gets a second sample in this call (n_samples <= slots per pixel, the usual batch); otherwise it
 * behaves exactly like rpt_render.  Results are read through the same entry points (each synchronises by itself).
 * rpt_wait blocks until everything enqueued so far has completed.  rpt_stream hands out the HIP stream
 * (hipStream_t) the library works on, so that a caller can order its own copies / collectives after a batch without
 * a host round trip (bench.py wraps it as a torch ExternalStream for the per-batch gather). */
int rpt_render_async(rpt_ctx *ctx, uint32_t n_samples);
int rpt_wait(rpt_ctx *ctx);
int rpt_stream(rpt_ctx *ctx, void **hip_stream_out);

/* Replaces output_buffer.read_blocking (src/trace.rs:198).  Writes the SUM
 * (not the mean) as width*height float4 (r, g, b, sample count) in row-major,
 * y-down order.  Pixels of tiles owned by other ranks are written as zeros.
 * The tile-major block is un-tiled on the device and leaves it as one DMA into
 * pinned memory owned by the context; rpt_map_accum hands out that buffer itself
 * (valid until the next rpt_read_accum / rpt_map_accum / rpt_set_config on the
 * context), rpt_read_accum copies it into the caller's. */
int rpt_read_accum(rpt_ctx *ctx, float *out_rgba, uint32_t *out_samples);
int rpt_map_accum(rpt_ctx *ctx, const float **out_rgba, uint32_t *out_samples);

/* Post-accumulation step on the device: mean = sum / samples
 * (src/trace.rs:199-204) then tonemap operator 0..6 exactly as the display shader
 * numbers them (src/resources/render.wgsl:131-153: 0 none, 1 Reinhard, 2 ACES
 * Narkowicz x0.6, 3 ACES Narkowicz, 4 ACES Hill, 5 Neutral, 6 Uncharted).  Writes
 * width*height*3 floats, row-major; other ranks' pixels are zero. */
int rpt_resolve(rpt_ctx *ctx, uint32_t tonemap_op, float *out_rgb);

/* Reads back rng[i] (n, offset) for every pixel (other ranks' pixels: zeros);
 * lets a caller verify `rng[i].x += 1` semantics (kernels/src/lib.rs:226). */
int rpt_read_rng(rpt_ctx *ctx, rpt_rng_state *out);

/* --- multi-GPU gather support ---------------------------- */
/* This rank's accumulators live in ONE contiguous tile-major device block of
 * rpt_local_pixels() float4: tiles in ascending tile id; inside a tile 8x8
 * pixel blocks row-major, pixels row-major inside a block; pixels outside the
 * image skipped (rpt_tile_order gives the exact order).  The caller (one
 * process per GPU) hands that pointer to its RCCL gather and gives the root
 * the concatenation. */
/* Pure function, no context / GPU needed: writes the (x | y << 16) pixel
 * coordinates of rank's block, in block order, into out_xy (capacity entries)
 * and the count into *n. out_xy may be NULL to query the count. */
int rpt_tile_order(uint32_t width, uint32_t height, uint32_t rank, uint32_t world_size,
                   uint32_t *out_xy, size_t capacity, size_t *n);
int rpt_local_pixels(rpt_ctx *ctx, uint64_t *n_pixels);
/* No synchronisation: order your work after the batch on rpt_stream().  The pointer is invalidated by
 * rpt_set_config (resize), rpt_set_partition and rpt_set_samples_in_flight. */
int rpt_local_block_device_ptr(rpt_ctx *ctx, void **dev_ptr);
/* Pixel count of any rank's block for the current config (for gather sizes). */
int rpt_rank_pixels(rpt_ctx *ctx, uint32_t rank, uint64_t *n_pixels);
/* For callers that run their own collective: scatter the `world_size` gathered tile-major blocks (device memory)
 * into a row-major width*height float4 image in device memory, on the context's stream — launch only (the destination
 * map is rebuilt when the configuration changes, never per batch); synchronise with rpt_wait.  Block r starts at
 * element r * block_stride_pixels (a gather into equal-sized padded slots), or the blocks are tightly concatenated
 * when block_stride_pixels == 0. */
int rpt_untile(rpt_ctx *ctx, const void *dev_gathered_blocks, uint64_t block_stride_pixels, void *dev_out_image);

/* The gather itself, inside the library: RCCL (ncclSend / ncclRecv, grouped) over xGMI, on a second HIP stream.
 *   one process per GPU:  rank 0 calls rpt_comm_unique_id and hands the 128 bytes to every rank by any means (file,
 *                         socket, torch.distributed); every rank calls rpt_comm_init (= ncclCommInitRank on the
 *                         context's device + rpt_set_partition(rank, world_size)).
 *   after a batch:        rpt_gather_async — stream-ordered after everything enqueued so far (rpt_render_async
 *                         included): the accumulators are snapshotted, the blocks travel to rank 0 and are un-tiled
 *                         there while the next batch renders.  Collective: every rank must call it, in the same order.
 *                         Nothing is allocated, copied from the host or synchronised per call.
 *   rank 0:               rpt_read_gathered (host, W*H float4, the whole image) or rpt_gathered_device_ptr;
 *                         rpt_gather_wait blocks until the last gather has completed. */
#define RPT_COMM_ID_BYTES 128
int rpt_comm_unique_id(uint8_t *id_out /* RPT_COMM_ID_BYTES */);
int rpt_comm_init(rpt_ctx *ctx, const uint8_t *unique_id, uint32_t rank, uint32_t world_size);
/* One GPU, no RCCL: the same snapshot / second stream / un-tile / pinned DMA for a single context, so that the reference's
 * loop (render a batch, read it back: src/trace.rs:182-204) reads batch k while batch k+1 renders:
 *   rpt_render_async ; rpt_gather_async ; rpt_render_async (next batch) ; rpt_read_gathered -> the image after the first. */
int rpt_comm_init_local(rpt_ctx *ctx);
int rpt_comm_world(rpt_ctx *ctx, uint32_t *rank_out, uint32_t *world_size_out);   /* as RCCL reports it (ncclCommCount) */
/* Which collective library the process resolved (dlopen): "librccl.so.1" ..., "" before the first communicator, or the path
 * given in RPT_RCCL_LIBRARY — an override that exists for tests/fake_rccl (N processes on a one-GPU test box); a
 * measurement made with it set is not a measurement of RCCL, and bench.py refuses to run then. */
const char *rpt_comm_library(void);
int rpt_gather_async(rpt_ctx *ctx);
int rpt_gather_wait(rpt_ctx *ctx);
int rpt_read_gathered(rpt_ctx *ctx, float *out_rgba, uint32_t *out_samples);
int rpt_gathered_device_ptr(rpt_ctx *ctx, void **dev_ptr);

/* ONE process driving every GPU of the node — the shape the reference's single render thread (src/trace.rs:136-224,
 * src/app.rs:157-164) can call: the same entry points as above, fanned out over n_devices contexts created with
 * ncclCommInitAll; the caller sees one W x H image.  rpt_multi_render = one batch on every GPU + the batch's single
 * gather, returns once enqueued; rpt_multi_read_accum waits and returns the whole image from rank 0.
 * RPT_MULTI_ALLOW_SHARED_DEVICE: test aid for boxes with one GPU — a device may be listed several times; ranks then
 * exchange their blocks by stream-ordered device copies instead of RCCL (which refuses two ranks on one device). */
typedef struct rpt_multi rpt_multi;
#define RPT_MULTI_ALLOW_SHARED_DEVICE 1u
int rpt_multi_create(const int *device_ids, int n_devices, uint32_t flags, rpt_multi **out);
int rpt_multi_size(rpt_multi *m);
rpt_ctx *rpt_multi_ctx(rpt_multi *m, int rank);        /* e.g. for rpt_set_samples_in_flight / rpt_get_stats per GPU */
int rpt_multi_upload_scene(rpt_multi *m,
                           const rpt_per_vertex_data *per_vertex, size_t n_vertices,
                           const rpt_triangle *indices, size_t n_triangles,
                           const rpt_bvh_node *nodes, size_t n_nodes,
                           const rpt_material_data *materials, size_t n_materials,
                           const rpt_light_pick_entry *light_pick, size_t n_light_pick,
                           const uint8_t *atlas_rgba8, uint32_t atlas_w, uint32_t atlas_h,
                           const float *skybox_rgba32f, uint32_t sky_w, uint32_t sky_h);
int rpt_multi_set_config(rpt_multi *m, const rpt_tracing_config *config);
int rpt_multi_reset(rpt_multi *m, const rpt_rng_state *rng_seed, const float *accum_init_rgba, uint32_t samples_init);
int rpt_multi_render(rpt_multi *m, uint32_t n_samples);
int rpt_multi_wait(rpt_multi *m);
int rpt_multi_read_accum(rpt_multi *m, float *out_rgba, uint32_t *out_samples);
int rpt_multi_get_stats(rpt_multi *m, rpt_stats *out);      /* counters summed, times = max over GPUs */
void rpt_multi_destroy(rpt_multi *m);
const char *rpt_multi_last_error(rpt_multi *m);

int rpt_get_stats(rpt_ctx *ctx, rpt_stats *out);
void rpt_destroy(rpt_ctx *ctx);
/* ctx may be NULL: returns the last error of a failed rpt_create on this thread. */
const char *rpt_last_error(rpt_ctx *ctx);
int rpt_abi_version(void);
/* tools/source_fingerprint.py of the device-side sources this library was built from ("unknown" for a build outside the Makefile):
 * profiles/traffic_*.json carry the same value, and bench.py reports counter-derived figures only when they match. */
const char *rpt_build_fingerprint(void);
/* compute units and peak engine clock of a HIP device, as the runtime reports them (bench.py: SIMD issue cycles available) */
int rpt_device_info(int device_id, uint32_t *compute_units_out, uint32_t *clock_khz_out);
/* Which visiting order the shadow (any-hit) walks of the uploaded scene use, and the probe that decided it.  The reference's shadow query
 * reads `.hit` only (kernels/src/light_pick.rs:148) and `.hit` does not depend on the order siblings are visited in (intersection.rs:191-213),
 * so the library may choose: fixed_out = 0 the reference's near-first order, 1 a fixed opaque-first order over a flipped copy of the tree —
 * chosen at rpt_upload_scene by the node visits of synthetic shadow rays under both (csrc/shadow_order.h).  The image is the same either way. */
int rpt_shadow_order(rpt_ctx *ctx, uint32_t *fixed_out, double *visits_near_out, double *visits_fixed_out, uint32_t *probe_rays_out, double *probe_ms_out);
/* How the LAST extension rays of a batch of known length are walked when the configuration has no NEE.  At bounce max_bounces - 1 the reference reads three
 * things off intersect_nearest (kernels/src/lib.rs:62-109): a miss adds the sky, a hit on the front of an emissive triangle adds its emission, any other hit
 * adds nothing — so a ray that passes the Moller-Trumbore test of no emissive triangle only has to answer "hit or miss", which the reference's own walk
 * answers at its FIRST accepted triangle (intersection.rs:195-203), in any visiting order.  mode_out: 0 the plain walk (more than four emissive triangles, a
 * scene that does not live in LDS, RPT_LAST_BOUNCE_HIT_OR_MISS=0); 1 those rays stop at their first accept, near child first; 2 / 3 / 4 fixed order over a
 * copy flipped so that the more opaque / the smaller / the more-opaque-per-node child comes first — chosen at rpt_upload_scene by the node visits of
 * synthetic rays (visits_out[4]: near first, then the three rules; csrc/shadow_order.h choose_last_order).  The image is the same in every mode. */
int rpt_last_bounce_order(rpt_ctx *ctx, uint32_t *mode_out, uint32_t *n_emissive_triangles_out, double *visits_out /* [4], nullable */, uint32_t *probe_rays_out,
                          double *probe_ms_out);

/* --- scene preparation on the device ---------------------- */
/* BVHBuilder::new(vertices, indices).sah_samples(n).build()  (reference src/bvh.rs:59-324, the call at
 * src/asset.rs:196) on the GPU: reorders `triangles` in place and writes the node pool exactly as the sequential
 * builder does (same nodes, same order, same bits; tests/test_gpu_bvh_build.py).  Host pointers, no context needed;
 * vertices are Vec4 (xyzw) as in bvh.rs:52; nodes_capacity >= 2 * n_triangles - 1; sah_samples <= 128.
 * device_ms_out (nullable) receives the device time of the build proper.  Errors: rpt_last_error(NULL). */
int rpt_bvh_build_gpu(int device_id, const float *vertices_xyzw, size_t n_vertices, rpt_triangle *triangles,
                      size_t n_triangles, uint32_t sah_samples, rpt_bvh_node *nodes_out, size_t nodes_capacity,
                      size_t *n_nodes_out, double *device_ms_out);

/* build_light_pick_table(vertices, indices, compute_emissive_mask(..), materials)  (reference src/light_pick.rs:13-122, the call at
 * src/asset.rs:197-198) with its parallel parts on the GPU — Heron areas and powers, probabilities, the stable sort, the table — and the
 * three order-dependent f32 chains (the two sums of :39-64 in index order, the robin-hood fill of :89-104) on the host between the device passes
 * (csrc/rpt_lights.hip: a question of latency per dependent operation, measured).  Same entries, same order, same bits as the sequential builder (tests/test_gpu_light_table.py).  Host pointers, no
 * context; entries_capacity >= the number of emissive triangles (>= 1: a scene without lights yields the one-entry sentinel, ratio = -1).
 * ms_out (nullable): 4 doubles — total, device passes, host chains, transfers.  A NaN pick probability — a NaN vertex, a total power of 0 (every emissive
 * triangle degenerate) or inf — is refused by name (RPT_ESCENE): the reference's answer there depends on its sequential sort, which rpt_light_table_build restates. */
int rpt_light_table_build_gpu(int device_id, const float *vertices_xyzw, size_t n_vertices, const rpt_triangle *triangles, size_t n_triangles,
                              const rpt_material_data *materials, size_t n_materials, rpt_light_pick_entry *entries_out, size_t entries_capacity,
                              size_t *n_entries_out, uint32_t *n_emissive_out, double *ms_out);

/* (The test hooks — rpt_debug_*: device math against the host build, ray parity through the production kernels, the order probes' host driver — are
 * declared in rpt/rpt_debug.h: exported by the same library, no part of the boundary a host binds.) */

#ifdef __cplusplus
}
#endif
#endif /* RPT_H */
