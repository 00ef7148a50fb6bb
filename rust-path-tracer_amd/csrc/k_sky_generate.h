/*
 * k_sky_generate.h — the two remaining stages of the wavefront pipeline.
 *
 *  k_sky       miss shading in its own queue, because the procedural sky is an
 *              ALU monster (12 march steps, ~108 exp, 4 pow per miss;
 *              kernels/src/skybox.rs:18-94) that must not share waves with
 *              cheap surface shading.  Also the image-skybox branch
 *              (kernels/src/lib.rs:70-78).
 *  k_generate_first  first camera ray of every slot at the start of rpt_render
 *              (lib.rs:36-60).  Finished paths are accumulated and regenerated
 *              in place by whichever stage ends them (k_path.h).
 */
#ifndef RPT_K_SKY_GENERATE_H
#define RPT_K_SKY_GENERATE_H

#include "k_shade.h"

/* ---- kernels/src/skybox.rs ------------------------------------------------- */
#define SKY_EARTH_RADIUS 6360e3f
#define SKY_ATMOSPHERE_RADIUS 6380e3f
#define SKY_H_RAY 8e3f
#define SKY_H_MIE 12e2f

__device__ __forceinline__ float sky_escape(F3 p, F3 d, float r) {
    F3 v = p - f3(0.0f, -SKY_EARTH_RADIUS, 0.0f);
    float b = dot3(v, d);
    float det = b * b - dot3(v, v) + r * r;
    if (det < 0.0f) return -1.0f;
    det = rptm::sqrtr(det);
    float t1 = -b - det;
    float t2 = -b + det;
    return t1 >= 0.0f ? t1 : t2;
}
__device__ __forceinline__ float2 sky_densities(F3 p) {
    float h = rptm::fmaxr(len3(p - f3(0.0f, -SKY_EARTH_RADIUS, 0.0f)) - SKY_EARTH_RADIUS, 0.0f);
    /* h is 0, NaN, or a multiple of ulp(SKY_EARTH_RADIUS) = 0.5 (a difference of two floats >= 6.36e6): never tiny */
    return make_float2(rptm::exp_sky(rptm::div_const_nontiny(-h, SKY_H_RAY, 1.0f / SKY_H_RAY)),
                       rptm::exp_sky(rptm::div_const_nontiny(-h, SKY_H_MIE, 1.0f / SKY_H_MIE)));
}

__device__ F3 sky_scatter(const float *sun4, F3 origin, F3 direction) {
    const F3 ray_coeff = f3(58e-7f, 135e-7f, 331e-7f);
    const F3 mie_scatter = f3(2e-5f, 2e-5f, 2e-5f);
    const F3 mie_effective = f3(2e-5f * 1.1f, 2e-5f * 1.1f, 2e-5f * 1.1f);
    const F3 sundir = f3(sun4[0], sun4[1], sun4[2]);

    float depth = sky_escape(origin, direction, SKY_ATMOSPHERE_RADIUS) / (float)12u;
    F3 i_r = f3s(0.0f), i_m = f3s(0.0f);
    float total_r = 0.0f, total_m = 0.0f;
    for (uint32_t i = 0; i < 12u; ++i) {
        F3 p = origin + direction * (depth * (float)i);
        float2 dens = sky_densities(p);
        float d_r = dens.x * depth, d_m = dens.y * depth;
        total_r = total_r + d_r;
        total_m = total_m + d_m;
        /* scatter_depth_int(p, sundir, escape(p, sundir, R_atm)) (skybox.rs:41-44) */
        float l = sky_escape(p, sundir, SKY_ATMOSPHERE_RADIUS);
        float2 da = sky_densities(p), db = sky_densities(p + sundir * l);
        float half_l = l / 2.0f;
        float sum_r = total_r + (da.x * half_l + db.x * half_l);
        float sum_m = total_m + (da.y * half_l + db.y * half_l);
        F3 e = (-ray_coeff) * sum_r - mie_effective * sum_m;
        F3 a = f3(rptm::exp_sky(e.x), rptm::exp_sky(e.y), rptm::exp_sky(e.z));
        i_r = i_r + a * d_r;
        i_m = i_m + a * d_m;
    }
    float mu = dot3(direction, sundir);
    F3 res = (sun4[3] * (1.0f + mu * mu)) *
             (i_r * ray_coeff * 0.0597f + i_m * mie_scatter * 0.0196f / rptm::powr(1.58f - 1.52f * mu, 1.5f));
    F3 g = mask_nan3(f3(rptm::sqrtr(res.x), rptm::sqrtr(res.y), rptm::sqrtr(res.z)));
    return f3(rptm::powr(g.x, 2.2f), rptm::powr(g.y, 2.2f), rptm::powr(g.z, 2.2f));
}

/* The same march with its 12 steps spread over 12 lanes of a 16-lane group (all lanes of a group hold the same
 * miss).  Only the two running sums are order dependent: they are rebuilt in exactly the sequential order
 * (total_k = ((0 + d_0) + d_1) + ... + d_k ; i_r = ((0 + t_0) + t_1) + ...) from lane-to-lane broadcasts, so
 * the result is bit-identical to sky_scatter.  Used when few misses are queued: the march is then latency
 * bound (one wave, 12 dependent steps of f64 exp/sqrt chains ~ 24 us) and this cuts the chain 12-fold. */
__device__ F3 sky_scatter_wide(const float *sun4, F3 origin, F3 direction, uint32_t j, uint32_t g0) {
    const F3 ray_coeff = f3(58e-7f, 135e-7f, 331e-7f);
    const F3 mie_scatter = f3(2e-5f, 2e-5f, 2e-5f);
    const F3 mie_effective = f3(2e-5f * 1.1f, 2e-5f * 1.1f, 2e-5f * 1.1f);
    const F3 sundir = f3(sun4[0], sun4[1], sun4[2]);

    float depth = sky_escape(origin, direction, SKY_ATMOSPHERE_RADIUS) / (float)12u;
    F3 p = origin + direction * (depth * (float)j);
    float2 dens = sky_densities(p);
    float d_r = dens.x * depth, d_m = dens.y * depth;
    float total_r = 0.0f, total_m = 0.0f;
    for (uint32_t k = 0; k < 12u; ++k) {
        float vr = __shfl(d_r, (int)(g0 + k), RPT_WAVE), vm = __shfl(d_m, (int)(g0 + k), RPT_WAVE);
        if (j >= k) {
            total_r = total_r + vr;
            total_m = total_m + vm;
        }
    }
    float l = sky_escape(p, sundir, SKY_ATMOSPHERE_RADIUS);
    float2 db = sky_densities(p + sundir * l);
    float half_l = l / 2.0f;
    float sum_r = total_r + (dens.x * half_l + db.x * half_l);
    float sum_m = total_m + (dens.y * half_l + db.y * half_l);
    F3 e = (-ray_coeff) * sum_r - mie_effective * sum_m;
    F3 a = f3(rptm::exp_sky(e.x), rptm::exp_sky(e.y), rptm::exp_sky(e.z));
    F3 t_r = a * d_r, t_m = a * d_m;
    F3 i_r = f3s(0.0f), i_m = f3s(0.0f);
    for (uint32_t k = 0; k < 12u; ++k) {
        int src = (int)(g0 + k);
        i_r = i_r + f3(__shfl(t_r.x, src, RPT_WAVE), __shfl(t_r.y, src, RPT_WAVE), __shfl(t_r.z, src, RPT_WAVE));
        i_m = i_m + f3(__shfl(t_m.x, src, RPT_WAVE), __shfl(t_m.y, src, RPT_WAVE), __shfl(t_m.z, src, RPT_WAVE));
    }
    float mu = dot3(direction, sundir);
    F3 res = (sun4[3] * (1.0f + mu * mu)) *
             (i_r * ray_coeff * 0.0597f + i_m * mie_scatter * 0.0196f / rptm::powr(1.58f - 1.52f * mu, 1.5f));
    F3 g = mask_nan3(f3(rptm::sqrtr(res.x), rptm::sqrtr(res.y), rptm::sqrtr(res.z)));
    return f3(rptm::powr(g.x, 2.2f), rptm::powr(g.y, 2.2f), rptm::powr(g.z, 2.2f));
}

/* The sky stage is LAZY: misses pile up in the queue over several iterations and are shaded once
 * q.sky_threshold of them are waiting, or when the traversal pass of this iteration found no ray at all
 * (so nothing else can make progress).  On closed scenes a handful of misses per iteration would otherwise
 * cost a full, latency-bound 12-step march every iteration (measured: 24 us x 280 launches = 10 % of the
 * DarkCornell run).  It is also the last kernel of an iteration, so its first thread reports progress to
 * the host through mapped pinned memory: work remains iff a ray was traced, a sample was started, or
 * misses are waiting. */
template <bool STRIDED>
__global__ __launch_bounds__(RPT_BLOCK) void k_sky(DevScene sc, DevState st, DevQueues q, DevConfig cfg, uint32_t iteration,
                                                   DevStats *stats) {
    uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    uint32_t positions, n;                                     /* queue positions to sweep (k_common.h: sharded queue), misses among them */
    q_extent(q.sky_cnt, positions, n);
    const uint32_t alive = q.count[Q_ALIVE0 + (iteration & 1u) * Q_LINE];
    const uint32_t drained = q.count[Q_DRAINED];
    if (i == 0u) {
        uint32_t busy = drained ? 0u : (alive | q.count[Q_REGEN0 + (iteration & 1u) * Q_LINE] | (n != 0u ? 1u : 0u));
        if (busy == 0u && drained == 0u) q.count[Q_DRAINED] = 1u;
        __hip_atomic_store(&q.host_ring[iteration & q.ring_mask], ((unsigned long long)(iteration + 1u) << 32) | busy,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (drained != 0u) return;                                 /* surplus launch (grid-uniform) */
    if (q.sky_at_end == 0u && n < q.sky_threshold && alive != 0u) return;          /* not worth a pass yet (same test as k_traverse_nearest) */
    if (i == 0u && n) atomicAdd(&stats->sky_evals, (unsigned long long)n);
    /* STRIDED: a fixed grid of a few thousand workgroups walks the queue with a grid stride.  On a closed scene the queue
     * holds a few thousand misses, and a launch of n_slots / 256 workgroups that all just look at the counter costs 54 us per
     * iteration (profiles/r02_darkcornell_kernel_stats.csv) — 8 % of a batch on 1/8 of an image.  As a loop the march needs
     * 133 instead of 92 VGPRs (its constants are hoisted), so scenes with many misses keep one thread per entry; the host
     * picks the variant from the share of samples that ended in the sky so far (refresh_device_stats). */
    const uint32_t stride = STRIDED ? gridDim.x * RPT_BLOCK : 0u;
    if (cfg.c.has_skybox == 0u && positions <= q.sky_wide_limit) {
        /* few misses: 16 lanes per miss (block-uniform branch) */
        for (uint32_t m = i >> 4; m < positions; m += stride >> 4) {
            if (!q_filled(q.sky_cnt, m)) {                     /* (uniform over the 16 lanes of the miss) */
                if (!STRIDED) break;
                continue;
            }
            const uint32_t slot = q.sky[m];
            const uint32_t lane = __lane_id(), g0 = lane & ~15u, j = lane & 15u;
            float4 ra = st.ray_a[slot];
            float2 rb = st.ray_b[slot];
            F3 ro = f3(ra.x, ra.y, ra.z), rd = f3(ra.w, rb.x, rb.y);
            F3 sky = sky_scatter_wide(cfg.c.sun_direction, ro, rd, j, g0);
            if (j == 0u) {
                const float4 tf = st.thr[slot], r4 = st.rad[slot];
                F3 throughput = f3(tf.x, tf.y, tf.z), radiance = f3(r4.x, r4.y, r4.z);
                radiance = radiance + throughput * sky;                                       /* lib.rs:69 */
                finish_in_side_stage(st, cfg, slot, radiance, __float_as_uint(r4.w));
            }
            if (!STRIDED) break;
        }
        return;
    }
    for (; i < positions; i += stride) {
        if (!q_filled(q.sky_cnt, i)) {
            if (!STRIDED) break;
            continue;
        }
        uint32_t slot = q.sky[i];
        float4 ra = st.ray_a[slot];
        float2 rb = st.ray_b[slot];
        F3 ro = f3(ra.x, ra.y, ra.z), rd = f3(ra.w, rb.x, rb.y);
        const float4 tf = st.thr[slot], r4 = st.rad[slot];
        F3 throughput = f3(tf.x, tf.y, tf.z), radiance = f3(r4.x, r4.y, r4.z);
        if (cfg.c.has_skybox == 0u) {
            radiance = radiance + throughput * sky_scatter(cfg.c.sun_direction, ro, rd);      /* lib.rs:69 */
        } else {                                                                             /* lib.rs:72-77 */
            F3 rotated = mat3_mul(cfg.sky_rot, rd);
            float u = 0.5f + rptm::atan2r(rotated.z, rotated.x) / (2.0f * RPT_PI_F);
            float v = 1.0f - (0.5f + rptm::asinr(rotated.y) / RPT_PI_F);
            float intensity = cfg.c.sun_direction[3] * (1.0f / 15.0f);
            float4 s = sample_by_lod<false>(sc.skybox, u, v);
            radiance = radiance + throughput * f3(s.x, s.y, s.z) * intensity;
        }
        /* a miss always ends the path (lib.rs:79) */
        finish_in_side_stage(st, cfg, slot, radiance, __float_as_uint(r4.w));
        if (!STRIDED) break;
    }
}

/* Start of an rpt_render call: slot k of every pixel begins sample k; it will take samples k, k+S, ... of
 * the n_samples this call owes the pixel. */
__global__ __launch_bounds__(RPT_BLOCK) void k_generate_first(DevState st, DevQueues q, DevConfig cfg, uint32_t n_samples, DevStats *stats) {
    uint32_t slot = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (blockIdx.x == 0u)                                     /* queue counters and flags of the new call (no kernel of this call has run yet) */
        for (uint32_t k = threadIdx.x; k < (uint32_t)Q_COUNT; k += RPT_BLOCK) q.count[k] = 0u;
    if (blockIdx.x == 0u && threadIdx.x < RPT_Q_SHARDS) q.sky_cnt[threadIdx.x * RPT_Q_SHARD_STRIDE] = q.shadow_cnt[threadIdx.x * RPT_Q_SHARD_STRIDE] = 0u;
    if (slot >= st.n_slots) return;
    /* Every slot must have been left idle by the previous render call: an asynchronous batch enqueues a fixed number of
     * iterations (max_bounces, + 1 with several slots per pixel) without ever looking at a progress report, so this is
     * where a wrong bound would show — a sample still in flight here would be overwritten and lost. */
    if (__float_as_uint(st.hit[slot].y) != HIT_IDLE) atomicAdd(&stats->undrained, 1ull);
    const uint32_t S = 1u << st.group_shift, k = slot_k(st, slot), pix = slot_pix(st, slot);
    if (pix >= st.n_pixels) {                                 /* padding of the last chunk of 64 pixels: never holds a path */
        st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        return;
    }
    uint2 rs = st.rng[pix];
    if (cfg.c.max_bounces == 0u) {
        /* the bounce loop never runs (lib.rs:62): every sample adds (0,0,0,1) */
        if (k == 0u) {
            float4 acc = st.accum[pix];
            for (uint32_t s = 0; s < n_samples; ++s) acc.w += 1.0f;
            st.accum[pix] = acc;
            rs.x += n_samples;
            st.rng[pix] = rs;
        }
        st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        return;
    }
    uint32_t count = n_samples > k ? (n_samples - k + S - 1u) / S : 0u;
    if (count == 0u) st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
    else start_first_path(st, cfg, slot, rs.x + k, rs.y);      /* (owes count - 1 more: first_path_todo) */
}

/* all slots idle (state after allocation: nothing in flight) */
__global__ __launch_bounds__(RPT_BLOCK) void k_fill_idle(float2 *hit, uint32_t n) {
    uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (i < n) hit[i] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
}
/* counts the slots of the last render call that are not idle (rpt_wait: must be 0 once the stream has drained) */
__global__ __launch_bounds__(RPT_BLOCK) void k_check_drained(const float2 *hit, uint32_t n, DevStats *stats) {
    uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    const bool busy = i < n && __float_as_uint(hit[i].y) != HIT_IDLE;
    const unsigned long long m = rpt_ballot(busy);
    if (m != 0ull && __lane_id() == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(&stats->undrained, (unsigned long long)__popcll(m));
}

/* ---- post-accumulation step --------------------------------------------------------
 * mean = sum / sample_count (src/trace.rs:199-204) followed by one of the display tonemappers of
 * src/resources/render.wgsl:36-117 (operator selection :131-153).  Pure f32 rational curves, written in
 * the shader's operation order; input = the tile-major accumulator block, output = row-major RGB. */
__device__ __forceinline__ float tm_clamp01(float x) { return rptm::fminr(rptm::fmaxr(x, 0.0f), 1.0f); }
__device__ __forceinline__ F3 tm_aces_narkowicz(F3 x) {                       /* render.wgsl:36-43 */
    const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    F3 num = x * (a * x + f3s(b));
    F3 den = x * (c * x + f3s(d)) + f3s(e);
    return f3(tm_clamp01(num.x / den.x), tm_clamp01(num.y / den.y), tm_clamp01(num.z / den.z));
}
__device__ __forceinline__ F3 tm_aces_hill(F3 x) {                            /* render.wgsl:46-69 */
    /* transpose(mat3x3(rows)) * v  ==  rows dotted with v, accumulated column by column (x, then y, then z) */
    F3 color = f3(0.59719f, 0.07600f, 0.02840f) * x.x + f3(0.35458f, 0.90834f, 0.13383f) * x.y + f3(0.04823f, 0.01566f, 0.83777f) * x.z;
    F3 a = color * (color + f3s(0.0245786f)) - f3s(0.000090537f);
    F3 b = color * (0.983729f * color + f3s(0.4329510f)) + f3s(0.238081f);
    color = f3(a.x / b.x, a.y / b.y, a.z / b.z);
    color = f3(1.60475f, -0.10208f, -0.00327f) * color.x + f3(-0.53108f, 1.10813f, -0.07276f) * color.y +
            f3(-0.07367f, -0.00605f, 1.07602f) * color.z;
    return f3(tm_clamp01(color.x), tm_clamp01(color.y), tm_clamp01(color.z));
}
__device__ __forceinline__ F3 tm_curve(F3 x, float a, float b, float c, float d, float e, float f) {   /* :75-77, :103-111 */
    F3 num = x * (a * x + f3s(c * b)) + f3s(d * e);
    F3 den = x * (a * x + f3s(b)) + f3s(d * f);
    return f3(num.x / den.x, num.y / den.y, num.z / den.z) - f3s(e / f);
}
__device__ __forceinline__ F3 tonemap(uint32_t op, F3 x) {
    switch (op) {
        case 1u: return f3(x.x / (x.x + 1.0f), x.y / (x.y + 1.0f), x.z / (x.z + 1.0f));     /* Reinhard :71-73 */
        case 2u: return tm_aces_narkowicz(x * 0.6f);
        case 3u: return tm_aces_narkowicz(x);
        case 4u: return tm_aces_hill(x);
        case 5u: {                                                                           /* Neutral :79-101 */
            F3 w = tm_curve(f3s(5.3f), 0.2f, 0.29f, 0.24f, 0.272f, 0.02f, 0.3f);
            F3 white_scale = f3(1.0f / w.x, 1.0f / w.y, 1.0f / w.z);
            F3 y = tm_curve(x * white_scale, 0.2f, 0.29f, 0.24f, 0.272f, 0.02f, 0.3f) * white_scale;
            return f3(y.x / 1.0f, y.y / 1.0f, y.z / 1.0f);
        }
        case 6u: {                                                                           /* Uncharted :113-121 */
            F3 curr = tm_curve(x * 2.0f, 0.15f, 0.50f, 0.10f, 0.20f, 0.02f, 0.30f);
            F3 w = tm_curve(f3s(11.2f), 0.15f, 0.50f, 0.10f, 0.20f, 0.02f, 0.30f);
            return curr * f3(1.0f / w.x, 1.0f / w.y, 1.0f / w.z);
        }
        default: return x;
    }
}
__global__ __launch_bounds__(RPT_BLOCK) void k_resolve(const float4 *accum, const uint32_t *pixel_xy, uint32_t n_pixels, uint32_t width,
                                                       float sample_count, uint32_t op, float *out_rgb) {
    uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = accum[i];
    F3 mean = f3(a.x / sample_count, a.y / sample_count, a.z / sample_count);
    F3 c = tonemap(op, mean);
    uint32_t pxy = pixel_xy[i];
    size_t at = ((size_t)(pxy >> 16) * width + (pxy & 0xffffu)) * 3u;
    out_rgb[at] = c.x; out_rgb[at + 1] = c.y; out_rgb[at + 2] = c.z;
}

#endif /* RPT_K_SKY_GENERATE_H */
