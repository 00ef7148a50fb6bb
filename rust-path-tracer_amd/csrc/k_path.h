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

/* Completion of generations: ONE THREAD PER PIXEL.  When all S slots of the pixel are finished (HIT_DONE) or have nothing left
 * (HIT_IDLE) and at least one is finished, their radiances are added to the accumulator IN SLOT ORDER (= sample order,
 * kernels/src/lib.rs:225, src/trace.rs:295: the f32 sum order is part of the result), the pixel's rng.n advances by the number
 * of samples (lib.rs:226), and every finished slot starts its next sample (lib.rs:36-60) or goes idle.  The slots of a pixel
 * are 64 apart (k_common.h, slot_pix; q_shift = 0), so every load of the loop is a coalesced wave access.
 * Round 2 did this inside the shade stage with the S slots of a pixel in adjacent lanes: a dependent chain across the lanes of
 * every wave (ds_bpermute, then a DPP shift chain: 2 of 64 lanes useful), 0.45 ms of the last shade launch of a DarkCornell
 * batch + a 0.19 ms completion pass.  A batch of known length (no slot takes a second sample) runs this once, after its last
 * iteration; otherwise it follows every shade stage. */
__global__ __launch_bounds__(RPT_BLOCK) void k_complete(DevState st, DevQueues q, DevConfig cfg, uint32_t iteration, uint32_t final_pass,
                                                        DevStats *stats) {
    /* a surplus launch of the run-ahead returns at once (grid-uniform) — but not the one completion of a batch of known length:
     * "drained" there only says that no RAY was left in an earlier iteration, the finished samples still wait to be added */
    if (!final_pass && q.count[Q_DRAINED] != 0u) return;
    const uint32_t pix = blockIdx.x * RPT_BLOCK + threadIdx.x;
    const uint32_t S = 1u << st.group_shift;
    bool started = false;
    if (pix < st.n_pixels) {
        uint32_t done_mask = 0u;
        bool all = true;
        for (uint32_t k = 0; k < S; ++k) {
            const uint32_t w = __float_as_uint(st.hit[pix_slot(st, pix, k)].y);
            if (w == HIT_DONE) done_mask |= 1u << k;
            else if (w != HIT_IDLE) { all = false; break; }
        }
        if (all && done_mask != 0u) {
            float4 acc = st.accum[pix];
            uint2 rs = st.rng[pix];
            const uint32_t new_n = rs.x + (uint32_t)__popc(done_mask);
            for (uint32_t k = 0; k < S; ++k) {
                if (((done_mask >> k) & 1u) == 0u) continue;
                const uint32_t slot = pix_slot(st, pix, k);
                const float4 r = st.rad[slot];
                acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += 1.0f;
                const uint32_t todo = __float_as_uint(r.w);
                if (todo == 0u) {
                    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
                } else {
                    start_path(st, cfg, slot, new_n + k, rs.y, todo - 1u);        /* slot k takes the samples k, k + S, ... */
                    started = true;
                }
            }
            st.accum[pix] = acc;
            rs.x = new_n;
            st.rng[pix] = rs;
        } else if (final_pass && !all) {
            /* the one completion of a batch of known length found a sample still in flight: the bound on its iterations was
             * wrong (must never happen; rpt_wait / rpt_render report it).  Counted like k_check_drained would: slots not idle. */
            uint32_t bad = 0u;
            for (uint32_t k = 0; k < S; ++k) bad += __float_as_uint(st.hit[pix_slot(st, pix, k)].y) != HIT_IDLE ? 1u : 0u;
            atomicAdd(&stats->undrained, (unsigned long long)bad);
        }
    }
    /* tell the host that new samples were started (one plain store per wave, every writer stores 1) */
    const unsigned long long any = rpt_ballot(started);
    if (any != 0ull && __lane_id() == (uint32_t)__ffsll((long long)any) - 1u) raise_flag(&q.count[Q_REGEN0 + (iteration & 1u) * Q_LINE]);
}

#endif /* RPT_K_PATH_H */
