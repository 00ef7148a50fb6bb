/*
 * shared_structs.h — C mirror of the reference's host<->kernel buffer contract.
 *
 * Every type here is the bit-exact `#[repr(C)]` layout of a struct in the
 * reference crate `shared_structs` (reference: shared_structs/src/lib.rs).
 * The Rust host keeps owning these buffers; librpt_hip.so only reads them.
 * Sizes/offsets are locked by static asserts (C11 / C++11).
 */
#ifndef RPT_SHARED_STRUCTS_H
#define RPT_SHARED_STRUCTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RPT_STATIC_ASSERT(c, m) static_assert(c, m)
extern "C" {
#else
#define RPT_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* reference: shared_structs/src/lib.rs:12-25 (TracingConfig), defaults :27-42 */
typedef struct rpt_tracing_config {
    float    cam_position[4];          /* xyz used                         @0  */
    float    cam_rotation[4];          /* x = pitch, y = yaw               @16 */
    uint32_t width;                    /*                                  @32 */
    uint32_t height;                   /*                                  @36 */
    uint32_t min_bounces;              /*                                  @40 */
    uint32_t max_bounces;              /*                                  @44 */
    float    sun_direction[4];         /* xyz direction, w intensity       @48 */
    uint32_t nee;                      /* 0 none, 1 MIS, 2 direct-only     @64 */
    uint32_t has_skybox;               /*                                  @68 */
    float    specular_weight_clamp[2]; /*                                  @72 */
} rpt_tracing_config;
RPT_STATIC_ASSERT(sizeof(rpt_tracing_config) == 80, "TracingConfig is 80 bytes");
RPT_STATIC_ASSERT(offsetof(rpt_tracing_config, width) == 32, "width@32");
RPT_STATIC_ASSERT(offsetof(rpt_tracing_config, sun_direction) == 48, "sun@48");
RPT_STATIC_ASSERT(offsetof(rpt_tracing_config, nee) == 64, "nee@64");
RPT_STATIC_ASSERT(offsetof(rpt_tracing_config, specular_weight_clamp) == 72, "clamp@72");

/* reference: shared_structs/src/lib.rs:44-56 (MaterialData). Each vec4 is a
 * colour / splat scalar, or an atlas rectangle (u0, v0, su, sv) when the
 * matching has_*_texture flag is non-zero. */
typedef struct rpt_material_data {
    float    emissive[4];              /* @0  */
    float    albedo[4];                /* @16 */
    float    roughness[4];             /* @32 */
    float    metallic[4];              /* @48 */
    float    normals[4];               /* @64 */
    uint32_t has_albedo_texture;       /* @80 */
    uint32_t has_metallic_texture;     /* @84 */
    uint32_t has_roughness_texture;    /* @88 */
    uint32_t has_normal_texture;       /* @92 */
} rpt_material_data;
RPT_STATIC_ASSERT(sizeof(rpt_material_data) == 96, "MaterialData is 96 bytes");
RPT_STATIC_ASSERT(offsetof(rpt_material_data, has_albedo_texture) == 80, "flags@80");

/* reference: shared_structs/src/lib.rs:92-100 (PerVertexData) */
typedef struct rpt_per_vertex_data {
    float vertex[4];                   /* w = 1  @0  */
    float normal[4];                   /* w = 0  @16 */
    float tangent[4];                  /*        @32 */
    float uv0[2];                      /*        @48 */
    float uv1[2];                      /* zero   @56 */
} rpt_per_vertex_data;
RPT_STATIC_ASSERT(sizeof(rpt_per_vertex_data) == 64, "PerVertexData is 64 bytes");
RPT_STATIC_ASSERT(offsetof(rpt_per_vertex_data, uv0) == 48, "uv0@48");

/* reference: shared_structs/src/lib.rs:102-119 (LightPickEntry);
 * ratio < 0 marks the "no lights" sentinel (table length 1). */
typedef struct rpt_light_pick_entry {
    uint32_t triangle_index_a;         /* @0  */
    float    triangle_area_a;          /* @4  */
    float    triangle_pick_pdf_a;      /* @8  */
    uint32_t triangle_index_b;         /* @12 */
    float    triangle_area_b;          /* @16 */
    float    triangle_pick_pdf_b;      /* @20 */
    float    ratio;                    /* @24 */
} rpt_light_pick_entry;
RPT_STATIC_ASSERT(sizeof(rpt_light_pick_entry) == 28, "LightPickEntry is 28 bytes");
RPT_STATIC_ASSERT(offsetof(rpt_light_pick_entry, ratio) == 24, "ratio@24");

/* reference: shared_structs/src/lib.rs:121-191 (BVHNode). The two .w lanes
 * carry u32 *bit patterns*: triangle_count and left_node_index (inner, count
 * == 0; right = left + 1) or first_triangle_index (leaf, count > 0). */
typedef struct rpt_bvh_node {
    float    aabb_min[3];              /* @0  */
    uint32_t triangle_count;           /* @12 */
    float    aabb_max[3];              /* @16 */
    uint32_t left_or_first;            /* @28 */
} rpt_bvh_node;
RPT_STATIC_ASSERT(sizeof(rpt_bvh_node) == 32, "BVHNode is 32 bytes");
RPT_STATIC_ASSERT(offsetof(rpt_bvh_node, triangle_count) == 12, "count@12");
RPT_STATIC_ASSERT(offsetof(rpt_bvh_node, left_or_first) == 28, "left@28");

/* index buffer element: UVec4(v0, v1, v2, material_index), post-BVH-reorder
 * (reference: src/asset.rs:106, src/bvh.rs:288). */
typedef struct rpt_triangle { uint32_t v0, v1, v2, material; } rpt_triangle;
RPT_STATIC_ASSERT(sizeof(rpt_triangle) == 16, "index element is 16 bytes");

/* rng buffer element: UVec2(n, offset) (reference: kernels/src/rng.rs:34-49,
 * src/trace.rs:150-159). */
typedef struct rpt_rng_state { uint32_t n, offset; } rpt_rng_state;
RPT_STATIC_ASSERT(sizeof(rpt_rng_state) == 8, "rng element is 8 bytes");

/* reference: shared_structs/src/lib.rs:193-236 (NextEventEstimation) */
enum { RPT_NEE_NONE = 0, RPT_NEE_MIS = 1, RPT_NEE_DIRECT = 2 };

#ifdef __cplusplus
}
#endif
#endif /* RPT_SHARED_STRUCTS_H */
