/*
 * k_path.h — begin / end of a path: camera ray generation and the accumulate +
 * regenerate step shared by every stage that can finish a path (shade, shadow,
 * sky).  Reference: kernels/src/lib.rs:36-60 (rng init, jitter, pinhole camera,
 * state init) and lib.rs:185,225-226 (`output[i] += (radiance, 1)`,
 * `rng[i].x += 1`).
 */
#ifndef RPT_K_PATH_H
#define RPT_K_PATH_H

#include "k_common.h"

__device__ __forceinline__ F3 mat3_mul(const float *m, F3 v) {   /* Mat3::mul_vec3, column-major */
    F3 r = f3(m[0], m[1], m[2]) * v.x;
    r = r + (f3(m[3], m[4], m[5]) * v.y);
    r = r + (f3(m[6], m[7], m[8]) * v.z);
    return r;
}

/* camera ray for the sample whose LDS key is `key` = n + offset (lib.rs:36-51) */
__device__ __forceinline__ void camera_ray(const DevConfig &cfg, uint32_t px, uint32_t py, uint32_t key, F3 &ro, F3 &rd) {
    Rng rng{key, 0u};
    float j1 = rng.next(), j2 = rng.next();
    float sx = (float)px + j1, sy = (float)py + j2;
    float ux = (sx / (float)cfg.c.width) * 2.0f - 1.0f;
    float uy = (1.0f - sy / (float)cfg.c.height) * 2.0f - 1.0f;
    uy *= (float)cfg.c.height / (float)cfg.c.width;
    ro = f3(cfg.c.cam_position[0], cfg.c.cam_position[1], cfg.c.cam_position[2]);
    rd = mat3_mul(cfg.euler, norm3(f3(ux, uy, 1.0f)));
}

/* write a fresh path into slot: throughput 1, radiance 0, bounce 0, LDS dimension 2 (jitter consumed).
 * `n` is the sample's own sequence number (pixel's rng.n + k), `offset` the pixel's LDS offset. */
__device__ __forceinline__ void start_path(const DevState &st, const DevConfig &cfg, uint32_t slot, uint32_t n, uint32_t offset,
                                           uint32_t todo_after) {
    uint32_t pxy = st.pixel_xy[slot_pix(st, slot)];
    F3 ro, rd;
    camera_ray(cfg, pxy & 0xffffu, pxy >> 16, n + offset, ro, rd);
    st.ray_a[slot] = make_float4(ro.x, ro.y, ro.z, rd.x);
    st.ray_b[slot] = make_float2(rd.y, rd.z);
    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PENDING));
    st.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(MAKE_FLAGS(0u, 0u, 2u)));
    st.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(todo_after));
}

/* The first paths of a render call (k_generate_first) store only their ray: throughput 1, radiance 0, bounce 0 and the samples
 * the slot still owes are functions of the slot, and the shade stage of iteration 0 — where EVERY traversed slot is such a path
 * — takes them as given instead of reading them (and writes the records for whoever reads them later: the continuing path, the
 * sky and shadow stages).  Saves a 32-byte write and a 16-byte read per slot and call: 1.6 GB of a DarkCornell batch. */
#define RPT_FRESH_FLAGS MAKE_FLAGS(0u, 0u, 2u)
__device__ __forceinline__ void start_first_path(const DevState &st, const DevConfig &cfg, uint32_t slot, uint32_t n, uint32_t offset) {
    uint32_t pxy = st.pixel_xy[slot_pix(st, slot)];
    F3 ro, rd;
    camera_ray(cfg, pxy & 0xffffu, pxy >> 16, n + offset, ro, rd);
    st.ray_a[slot] = make_float4(ro.x, ro.y, ro.z, rd.x);
    st.ray_b[slot] = make_float2(rd.y, rd.z);
    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PENDING));
}
/* samples slot `slot` owes after its first one in a call of n_samples (the slot has one: it was started) */
__device__ __forceinline__ uint32_t first_path_todo(const DevState &st, uint32_t slot, uint32_t n_samples) {
    const uint32_t S = 1u << st.group_shift;
    return (n_samples - slot_k(st, slot) + S - 1u) / S - 1u;
}

/* A path ended in a side stage (sky / shadow).  With one slot per pixel the sample is accumulated and the
 * next one started on the spot; with several, the finished radiance is parked (HIT_DONE) and k_complete, which sees all
 * the slots of a pixel, completes the generation.  Every stage that ends a path — shade, shadow, sky — ends it through here. */
__device__ __forceinline__ void finish_in_side_stage(const DevState &st, const DevConfig &cfg, uint32_t slot, F3 radiance, uint32_t todo) {
    if (st.group_shift == 0u) {
        float4 acc = st.accum[slot];
        acc.x += radiance.x; acc.y += radiance.y; acc.z += radiance.z; acc.w += 1.0f;
        st.accum[slot] = acc;
        uint2 rs = st.rng[slot];
        rs.x += 1u;
        st.rng[slot] = rs;
        if (todo == 0u) st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        else start_path(st, cfg, slot, rs.x, rs.y, todo - 1u);
    } else {
        st.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(todo));
        st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_DONE));
    }
}

#endif /* RPT_K_PATH_H */
