/*
 * rpt_debug.h — test hooks of librpt_hip.so.  NOT part of the drop-in boundary (include/rpt/rpt.h is what replaces the gpgpu-rs calls of
 * src/trace.rs:136-224); nothing here is needed by, or bound in, a host (ffi/rpt.rs).  They exist so that the tests can hold single pieces of the device
 * code — the shared math, the traversal kernels, the order probes, the gather's point-to-point calls — against the CPU oracle piece by piece.
 */
#ifndef RPT_DEBUG_H
#define RPT_DEBUG_H

#include "rpt.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Evaluate one shared-math function on the DEVICE over n floats so tests can
 * check bit-equality with the host build of the same header. op: 0 sin, 1 cos,
 * 2 acos, 3 exp, 4 pow(x,y), 5 asin, 6 atan2(y=x_in, x=y_in), 7 sqrt, 8 IEEE x/y,
 * 9 x/y through the traversal's guarded exact fast-division path (rpt_fastdiv.h), 10 the sky march's float-only exp
 * (rpt_math.h exp_sky), 11 an atlas texel channel u8 / 255 (rpt_math.h unorm8). */
int rpt_debug_math(rpt_ctx *ctx, int op, const float *x, const float *y, float *out, size_t n);
/* Same on the HOST build (no device needed, ctx may be NULL). */
int rpt_debug_math_host(int op, const float *x, const float *y, float *out, size_t n);
/* The decision rpt_upload_scene takes about the shadow walks' visiting order (rpt_shadow_order), without a device: the sequential driver of the probe
 * (shadow_order.h) — the checker of the kernels rpt_upload_scene runs.
 * flip_out (nullable): per child pair p = nodes (2p + 1, 2p + 2), 1 where the fixed order enters the right child first. */
int rpt_debug_shadow_order_host(const rpt_per_vertex_data *vertices, size_t n_vertices, const rpt_triangle *indices, size_t n_triangles,
                                const rpt_bvh_node *nodes, size_t n_nodes, const rpt_material_data *materials, size_t n_materials,
                                const rpt_light_pick_entry *light_pick, size_t n_light_pick, uint32_t *fixed_out, double *visits_near_out,
                                double *visits_fixed_out, uint32_t *probe_rays_out, uint8_t *flip_out);
/* The same for the order of the hit-or-miss lanes of the last extension rays (rpt_last_bounce_order): rule_out 0 near child first, 1 / 2 / 3 the fixed rules. */
int rpt_debug_last_order_host(const rpt_per_vertex_data *vertices, size_t n_vertices, const rpt_triangle *indices, size_t n_triangles,
                              const rpt_bvh_node *nodes, size_t n_nodes, const rpt_material_data *materials, size_t n_materials,
                              uint32_t *rule_out, double *visits_out /* [4] */, uint32_t *probe_rays_out, uint8_t *flip_out);
/* Exhaustive device-side check of a cheap exact operation of rpt_math.h against its IEEE form over the float bit patterns
 * [lo_bits, lo_bits + count): op 0 sqrtr vs the correctly rounded sqrtf, op 1 div_const_nontiny(x, y, RN(1/y)) vs x / y, op 2 the one-instruction
 * f32 -> i32 conversion vs Rust's `as i32` written out.  Returns the
 * number of arguments whose results differ in any bit (NaN == NaN) and the smallest such bit pattern (0xffffffff if none). */
int rpt_debug_math_sweep(rpt_ctx *ctx, int op, uint32_t lo_bits, uint64_t count, float y, uint64_t *mismatches_out,
                         uint32_t *first_bad_bits_out);
/* Trace n rays through the uploaded BVH on the device. any_hit = 0: nearest
 * (kernels/src/intersection.rs:169-171) ; 1: any-hit with max_t
 * (:173-175).  Outputs per ray: t, triangle_index, flags (bit0 hit, bit1 backface). */
int rpt_debug_trace_rays(rpt_ctx *ctx, int any_hit, size_t n,
                         const float *origins_xyz, const float *dirs_xyz, const float *max_t,
                         float *out_t, uint32_t *out_tri, uint32_t *out_flags);
/* Nearest hits of n rays through the PRODUCTION traversal stage — the kernel and grid an iteration of rpt_render uses for the
 * context's scene and state (persistent LDS stream, streamed global-memory walk, ... per scene and developer knobs), fed through
 * the context's own slots; same outputs as rpt_debug_trace_rays(any_hit = 0).  Needs a configuration with at least n slots;
 * afterwards the context is as after rpt_reset with nothing rendered (call rpt_reset before rendering again). */
int rpt_debug_trace_rays_production(rpt_ctx *ctx, size_t n, const float *origins_xyz, const float *dirs_xyz,
                                    float *out_t, uint32_t *out_tri, uint32_t *out_flags);

/* The two BSDFs of the reference's kernels crate that trace_pixel never instantiates (kernels/src/bsdf.rs:46-176,
 * SURVEY.md 8f N4), evaluated on the device.  One item = 16 floats in: view(3) normal(3) r(3) albedo(3) ior roughness
 * pad(2); 8 floats out: pdf, lobe (u32 bits), spectrum(3), direction(3).  kind 0 Lambertian::sample, 1 Glass::sample,
 * 2 Lambertian::{evaluate, pdf} (sample_direction = r), 3 Glass::{evaluate, pdf} (lobe = (u32) r.x). */
int rpt_debug_bsdf(rpt_ctx *ctx, int kind, size_t n, const float *in, float *out);
/* The gather's point-to-point calls against the collective library the process resolved (rpt_comm_library), without a second GPU:
 * inside one ncclGroupStart / ncclGroupEnd this rank posts ncclRecv from rank - 1 and ncclSend to rank + 1 (one rank: to and from
 * itself) on the communicator's second stream, ordered by the same events as rpt_gather_async; n_floats of a known pattern travel
 * as ncclFloat and are compared on the host.  Needs rpt_comm_init; collective (every rank calls it). */
int rpt_debug_comm_selftest(rpt_ctx *ctx, uint32_t n_floats, uint64_t *mismatches_out);

/* Test aid: the next asynchronous batches of this context enqueue one iteration too few — proves that the completion checks (rpt_wait, the next
 * batch's k_generate_first) notice a sample left in flight instead of losing it. */
int rpt_debug_short_batch(rpt_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif /* RPT_DEBUG_H */
