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
    uint32_t pxy = st.pixel_xy[slot >> st.group_shift];
    F3 ro, rd;
    camera_ray(cfg, pxy & 0xffffu, pxy >> 16, n + offset, ro, rd);
    st.ray_a[slot] = make_float4(ro.x, ro.y, ro.z, rd.x);
    st.ray_b[slot] = make_float2(rd.y, rd.z);
    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PENDING));
    st.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(MAKE_FLAGS(0u, 0u, 2u)));
    st.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(todo_after));
}

/* A path ended in a side stage (sky / shadow).  With one slot per pixel the sample is accumulated and the
 * next one started on the spot; with several, the finished radiance is parked (HIT_DONE) and the shade
 * stage of the next iteration — where the pixel's slots sit in adjacent lanes — completes the generation. */
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

/* End of the shade stage, every lane of the wave (converged): lanes whose sample is finished (`done`,
 * radiance in registers) or whose slot has nothing left (`idle`) vote; when all S slots of a pixel are
 * done/idle the group's first lane adds the finished radiances to the accumulator IN SLOT ORDER (= sample
 * order, kernels/src/lib.rs:225, src/trace.rs:295), advances the pixel's rng.n by the number of samples
 * (lib.rs:226), and every done lane starts its next sample (lib.rs:36-60).  Lanes finished earlier than
 * their siblings write their radiance back and wait as HIT_DONE.  `fresh` = the sample finished in this
 * kernel (its state is not in memory yet); for a done lane that is not fresh the caller passes no radiance: it is read
 * from the slot's parked state here, and only when its generation completes. */
__device__ __forceinline__ void complete_generations(const DevState &st, const DevConfig &cfg, uint32_t *regen_flag, uint32_t slot,
                                                     bool done, bool idle, bool fresh, F3 radiance, uint32_t todo) {
    const uint32_t shift = st.group_shift, S = 1u << shift;
    const uint32_t lane = __lane_id();
    if (shift == 0u) {
        /* one slot per pixel: a finished sample is its own generation — no votes, no shuffles */
        bool started = false;
        if (done) {
            float4 acc = st.accum[slot];
            acc.x += radiance.x; acc.y += radiance.y; acc.z += radiance.z; acc.w += 1.0f;
            st.accum[slot] = acc;
            uint2 rs1 = st.rng[slot];
            rs1.x += 1u;
            st.rng[slot] = rs1;
            if (todo == 0u) {
                st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
            } else {
                start_path(st, cfg, slot, rs1.x, rs1.y, todo - 1u);
                started = true;
            }
        }
        unsigned long long any = rpt_ballot(started);
        if (any != 0ull && lane == (uint32_t)__ffsll((long long)any) - 1u) raise_flag(regen_flag);
        return;
    }
    const uint32_t g0 = lane & ~(S - 1u);
    const unsigned long long done_m = rpt_ballot(done), idle_m = rpt_ballot(idle);
    if (done_m == 0ull) return;
    const unsigned long long gm = (S >= 64u ? ~0ull : ((1ull << S) - 1ull)) << g0;
    const bool complete = (((done_m | idle_m) & gm) == gm) && ((done_m & gm) != 0ull);
    const bool leader = complete && lane == g0;
    const uint32_t pix = slot >> shift;
    if (rpt_ballot(complete) == 0ull) {
        /* No generation of this wave is complete — the common case on an open scene, where finished slots wait several
         * passes for the slowest sibling: skip the S-step exchange below (3 ds_bpermute per step: ~2 500 cycles per wave,
         * measured as ~half of the shade stage on PBRTest with 32 slots per pixel).  A lane that finished in this very
         * kernel still has to park its radiance. */
        if (done && fresh) {
            st.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(todo));
            st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_DONE));
        }
        return;
    }
    if (done && !fresh && complete) {
        /* parked earlier (HIT_DONE): only now is its radiance needed — a slot that waits several iterations for its
         * siblings (open scenes: most paths end in the sky after one bounce) costs one 8-byte look per pass, not 40 */
        const float4 parked = st.rad[slot];
        radiance = f3(parked.x, parked.y, parked.z);
        todo = __float_as_uint(parked.w);
    }
    /* The sequential f32 sum  acc = ((acc + r_0) + r_1) + ...  over the group's finished slots in slot order (= sample order:
     * part of the result).  The ACCUMULATOR travels along the lanes: the group's first lane loads it, and in every step each
     * lane takes its left neighbour's value (one DPP wave shift per component) and adds its own radiance if its sample
     * finished — after S - 1 shifts the group's last lane holds the sum and stores it.  (Round 2 had the first lane fetch one
     * slot per step through three ds_bpermute, ~26 cycles each: ~2 500 cycles per wave, 0.44 ms of the last shade launch of
     * a DarkCornell batch.)  */
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
    uint2 rs = make_uint2(0u, 0u);
    if (leader) {
        const float4 acc = st.accum[pix];
        ax = acc.x; ay = acc.y; az = acc.z; aw = acc.w;
        rs = st.rng[pix];
    }
    const bool adds = done && complete;
    if (adds) { ax += radiance.x; ay += radiance.y; az += radiance.z; }        /* (step 0: only the first lane's value is a real sum) */
    for (uint32_t k = 1; k < S; ++k) {                 /* wave-uniform trip count */
        ax = rpt_wave_shr1(ax); ay = rpt_wave_shr1(ay); az = rpt_wave_shr1(az);
        if (adds) { ax += radiance.x; ay += radiance.y; az += radiance.z; }
    }
    if (complete && lane == g0 + S - 1u) {
        float *out = reinterpret_cast<float *>(&st.accum[pix]);
        out[0] = ax; out[1] = ay; out[2] = az;
    }
    if (leader) {
        const uint32_t n_done = (uint32_t)__popcll(done_m & gm);
        /* .w += 1.0 per finished sample (lib.rs:185): for a whole count below 2^24 - 64 that IS + n_done in one step (every
           intermediate is exactly representable); anything else a caller resumed from takes the additions one by one */
        if (aw >= 0.0f && aw < 16777152.0f && aw == rptm::floorr(aw)) aw += (float)n_done;
        else for (uint32_t i = 0; i < n_done; ++i) aw += 1.0f;
        reinterpret_cast<float *>(&st.accum[pix])[3] = aw;
        rs.x += n_done;
        st.rng[pix] = rs;
    }
    const uint32_t new_n = (uint32_t)__shfl((int)rs.x, (int)g0, RPT_WAVE);
    const uint32_t offset = (uint32_t)__shfl((int)rs.y, (int)g0, RPT_WAVE);
    {   /* tell the host that new samples were started (one plain store per wave, every writer stores 1) */
        unsigned long long started = rpt_ballot(done && complete && todo != 0u);
        if (started != 0ull && lane == (uint32_t)__ffsll((long long)started) - 1u) raise_flag(regen_flag);
    }
    if (!done) return;
    if (complete) {
        if (todo == 0u) st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        else start_path(st, cfg, slot, new_n + (lane - g0), offset, todo - 1u);
    } else if (fresh) {
        st.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(todo));
        st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_DONE));
    }
}

#endif /* RPT_K_PATH_H */
