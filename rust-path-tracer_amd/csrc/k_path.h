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

/* write a fresh path into slot: throughput 1, radiance 0, bounce 0, LDS dimension 2 (jitter consumed) */
__device__ __forceinline__ void start_path(const DevState &st, const DevConfig &cfg, uint32_t slot, uint2 rs, uint32_t todo_after) {
    uint32_t pxy = st.pixel_xy[slot];
    F3 ro, rd;
    camera_ray(cfg, pxy & 0xffffu, pxy >> 16, rs.x + rs.y, ro, rd);
    st.ray_a[slot] = make_float4(ro.x, ro.y, ro.z, rd.x);
    st.ray_b[slot] = make_float4(rd.y, rd.z, 0.0f, __uint_as_float(HIT_PENDING));
    st.thr_rad[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    st.rad_misc[slot] = make_float4(0.0f, 0.0f, __uint_as_float(MAKE_FLAGS(0u, 0u, 2u)), __uint_as_float(todo_after));
}

/* A path of `slot` ended with `radiance`: accumulate it (sample order per pixel is
 * preserved because a slot carries one sample at a time), advance the pixel's rng
 * state, and start the pixel's next sample if this render call still owes one.
 * Returns true when a new path was written (its ray record is then marked
 * HIT_PENDING for the next traversal pass); otherwise the slot stays parked. */
__device__ __forceinline__ bool finish_and_regenerate(const DevState &st, const DevConfig &cfg, uint32_t slot, F3 radiance,
                                                      uint32_t todo) {
    float4 acc = st.accum[slot];
    acc.x += radiance.x; acc.y += radiance.y; acc.z += radiance.z; acc.w += 1.0f;
    st.accum[slot] = acc;
    uint2 rs = st.rng[slot];
    rs.x += 1u;
    st.rng[slot] = rs;
    if (todo == 0u) {
        reinterpret_cast<float2 *>(&st.ray_b[slot])[1] = make_float2(0.0f, __uint_as_float(HIT_PARKED));
        return false;
    }
    start_path(st, cfg, slot, rs, todo - 1u);
    return true;
}

#endif /* RPT_K_PATH_H */
