/*
 * rpt_host.h — C ABI of librpt_host.so: the host side of the render dispatch,
 * i.e. what sits ABOVE librpt_hip.so and mirrors the reference's own host
 * interface for this path (names and argument meaning follow the reference).
 *
 *   rpt_world_load        <->  World::from_path           src/asset.rs:55-224
 *   rpt_bvh_build         <->  BVHBuilder::build          src/bvh.rs:59-324
 *   rpt_light_table_build <->  build_light_pick_table     src/light_pick.rs:24-122
 *   rpt_blue_noise_seeds  <->  rng_data_blue setup        src/trace.rs:5,149-160
 *   rpt_tracing_state_*   <->  TracingState               src/trace.rs:40-92
 *   rpt_setup_trace       <->  setup_trace                src/trace.rs:331-344
 *   rpt_trace_gpu         <->  trace_gpu                  src/trace.rs:136-224
 *
 * The Rust host does not need this library (it already has all of the above
 * and only needs rpt.h); it exists because this image has no Rust toolchain,
 * so the host driver is restated in C++ to exercise the C ABI end to end.
 * Error behaviour follows rpt.h: 0 / negative code, rpt_host_last_error().
 */
#ifndef RPT_HOST_H
#define RPT_HOST_H

#include <stddef.h>
#include <stdint.h>
#include "shared_structs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Borrowed view of a loaded World (src/asset.rs:9-16). Pointers stay valid
 * until rpt_world_free. atlas is NULL when no material has a texture (the
 * reference always allocates 4096^2). */
typedef struct rpt_world_view {
    const rpt_per_vertex_data *per_vertex; size_t n_vertices;
    const rpt_triangle *indices; size_t n_triangles;
    const rpt_bvh_node *nodes; size_t n_nodes;
    const rpt_material_data *materials; size_t n_materials;
    const rpt_light_pick_entry *light_pick; size_t n_light_pick;
    const uint8_t *atlas_rgba8; uint32_t atlas_w, atlas_h;
    uint32_t bvh_max_depth;       /* root = depth 0 */
    uint32_t n_emissive_triangles;
} rpt_world_view;

typedef struct rpt_world rpt_world;

/* World::from_path for glTF-binary (.glb) files: node graph walk with the
 * reference's (x,z,y) axis swap and (i0,i2,i1) winding (asset.rs:78-128),
 * materials (135-175), BVH with sah_samples = 128 (196), light pick table on
 * the reordered indices (201-202), per-vertex packing (206-215).
 * Returns RPT_HOST_ELOAD when the file cannot be imported (the reference
 * returns None and trace_gpu silently returns, trace.rs:141-143). */
/* path: a .glb (glTF 2.0 binary; embedded PNG textures -> atlas) or a Wavefront .obj (+ .mtl: Kd, Ke, Pm, Pr). */
int rpt_world_load(const char *path, rpt_world **out);
/* rpt_world_load with options.  RPT_LOAD_EMISSIVE_STRENGTH: honour KHR_materials_emissive_strength
 * (emissive = factor * strength) instead of the reference's fixed x15 (src/asset.rs:163-166) for materials that carry
 * the extension — NOT the reference's behaviour, hence opt-in. */
#define RPT_LOAD_EMISSIVE_STRENGTH 1u
int rpt_world_load_ex(const char *path, uint32_t flags, rpt_world **out);
/* load_dynamic_image + dynamic_image_to_cpu_buffer (src/asset.rs:238-273): a skybox file (.png, Radiance .hdr) as the
 * reference's CPU path sees it — quantised to 8 bits per channel, alpha 1 — as width*height float RGBA, the layout
 * rpt_upload_scene takes.  Caller frees with rpt_host_free. */
int rpt_skybox_load(const char *path, float **rgba_out, uint32_t *width, uint32_t *height);
void rpt_host_free(void *p);
/* Build a World from caller-supplied geometry (already in the reference's
 * post-swap space): runs the same BVH + light-table + packing steps.  Used for
 * procedural scenes. normals/uvs may be NULL (zero-filled like asset.rs:209-212). */
int rpt_world_from_buffers(const float *vertices_xyz, const float *normals_xyz, const float *uvs,
                           size_t n_vertices, const uint32_t *triangles_v0v1v2_mat, size_t n_triangles,
                           const rpt_material_data *materials, size_t n_materials, rpt_world **out);
int rpt_world_view_get(const rpt_world *world, rpt_world_view *out);
/* ".rptscene": the World's five POD buffers (+ atlas) in one flat little-endian file, so the
 * Rust host (which has assimp) and this backend can exchange byte-identical inputs:
 *   header { "RPTSCN01", u64 n_vertices, n_triangles, n_nodes, n_materials, n_light_pick, u32 atlas_w, atlas_h }
 *   then PerVertexData[], UVec4 index[], BVHNode[], MaterialData[], LightPickEntry[], RGBA8 atlas. */
int rpt_world_save(const rpt_world *world, const char *path);
int rpt_world_load_cache(const char *path, rpt_world **out);
/* 8-bit RGBA PNG of a resolved frame (width*height*3 floats).  srgb_encode != 0 applies the sRGB transfer
 * function the reference's save path gets from its sRGB surface format (src/app.rs:759-845). */
int rpt_write_png(const char *path, const float *rgb, uint32_t width, uint32_t height, int srgb_encode);
void rpt_world_free(rpt_world *world);

/* BVHBuilder::new(vertices, indices).sah_samples(n).build(): reorders
 * `triangles` in place, writes up to 2*n_triangles-1 nodes, returns the node
 * count in *n_nodes_out. vertices are Vec4 (xyzw) as in bvh.rs:52. */
int rpt_bvh_build(const float *vertices_xyzw, size_t n_vertices, rpt_triangle *triangles, size_t n_triangles,
                  uint32_t sah_samples, rpt_bvh_node *nodes_out, size_t nodes_capacity, size_t *n_nodes_out);

/* Which builder rpt_world_load / rpt_world_from_buffers use for the BVH (src/asset.rs:196): 0 = the sequential host
 * restatement (default), 1 = rpt_bvh_build_gpu of librpt_hip.so (hip_library_path NULL = next to this library) on
 * device_id.  Both give the same node pool and triangle order bit for bit; a failing device build is an error, never
 * a silent fall-back. */
int rpt_host_set_bvh_builder(int use_gpu, const char *hip_library_path, int device_id);

/* compute_emissive_mask + build_light_pick_table (src/light_pick.rs:13-122).
 * table_out needs capacity >= max(1, n_triangles). */
int rpt_light_table_build(const float *vertices_xyzw, size_t n_vertices, const rpt_triangle *triangles,
                          size_t n_triangles, const rpt_material_data *materials, size_t n_materials,
                          rpt_light_pick_entry *table_out, size_t capacity, size_t *n_entries_out);

/* rng_data_blue (src/trace.rs:150-157): seed(x,y) = (0, u32(f32(b)/255 * 4294967295.0))
 * with b = 8-bit channel 0 of bluenoise.png at (x % 256, y % 256).
 * png_path NULL -> fixtures/bluenoise.png next to the library's repo root. */
int rpt_blue_noise_seeds(const char *png_path, uint32_t width, uint32_t height, rpt_rng_state *out);
/* The decoded 8-bit blue-noise tile itself (w*h bytes), for KATs. */
int rpt_blue_noise_tile(const char *png_path, uint8_t *out, size_t capacity, uint32_t *w, uint32_t *h);

/* TracingConfig::default() (shared_structs/src/lib.rs:27-42). */
void rpt_tracing_config_default(rpt_tracing_config *out);

/* TracingState (src/trace.rs:40-92). Plain struct + functions; `running`,
 * `samples`, `dirty`, `interacting` may be touched from another thread
 * (they are accessed with relaxed atomics inside the library). */
typedef struct rpt_tracing_state rpt_tracing_state;
rpt_tracing_state *rpt_tracing_state_new(uint32_t width, uint32_t height);   /* TracingState::new */
void rpt_tracing_state_free(rpt_tracing_state *s);
rpt_tracing_config *rpt_tracing_state_config(rpt_tracing_state *s);          /* state.config (caller mutates before tracing) */
const float *rpt_tracing_state_framebuffer(rpt_tracing_state *s, size_t *n_floats); /* W*H*3 mean RGB */
/* the same image copied under the state's lock — for a reader on another thread while rpt_trace_gpu runs (framebuffer.read()) */
int rpt_tracing_state_copy_framebuffer(rpt_tracing_state *s, float *out, size_t n_floats);
uint32_t rpt_tracing_state_samples(rpt_tracing_state *s);
void rpt_tracing_state_set_running(rpt_tracing_state *s, int running);
void rpt_tracing_state_set_sync_rate(rpt_tracing_state *s, uint32_t sync_rate);
void rpt_tracing_state_set_dirty(rpt_tracing_state *s, int dirty);
/* state.interacting (src/trace.rs:50): raised by the UI for as long as the camera is dragged; while it is up every batch is
 * rendered from zero samples, published, and discarded (src/trace.rs:187, 216-222). */
void rpt_tracing_state_set_interacting(rpt_tracing_state *s, int on);
/* state.config.write() from another thread while rpt_trace_gpu runs (taken under the state's lock); then set dirty: the
 * loop re-reads the configuration and restarts accumulation (src/trace.rs:216-222).  A new width / height ends the call. */
void rpt_tracing_state_set_config(rpt_tracing_state *s, const rpt_tracing_config *config);
/* DEFAULT ON: rpt_trace_gpu reads the image after batch k while batch k+1 renders (rpt_comm_init_local + rpt_gather_async /
 * rpt_read_gathered instead of rpt_read_accum): the framebuffer and sample count the caller sees run one batch behind the
 * device; final images, flush behaviour (a flushing iteration still publishes the batch it rendered) and sample totals are
 * the same.  on = 0 selects the blocking loop rpt_render ; rpt_read_accum, the literal shape of src/trace.rs:182-204
 * (8 % slower on DarkCornell 1024^2).  Set before rpt_trace_gpu is called. */
void rpt_tracing_state_set_overlap(rpt_tracing_state *s, int on);
/* setup_trace(width, height, samples) (src/trace.rs:331-344) — but exact: the
 * render stops after precisely `samples` samples (the reference's watcher
 * thread can overshoot). */
rpt_tracing_state *rpt_setup_trace(uint32_t width, uint32_t height, uint32_t samples);

/* trace_gpu(scene_path, skybox_path, state) (src/trace.rs:136-224) on top of
 * librpt_hip.so (loaded with dlopen from hip_library_path, or the default
 * location next to this library when NULL).  device_id selects the GPU. */
int rpt_trace_gpu(const char *scene_path, const char *skybox_path, rpt_tracing_state *state,
                  int device_id, const char *hip_library_path);

enum { RPT_HOST_ELOAD = -20, RPT_HOST_EPNG = -21, RPT_HOST_EDLOPEN = -22 };
const char *rpt_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RPT_HOST_H */
