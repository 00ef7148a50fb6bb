/*
 * k_shade.h — surface stage of the wavefront pipeline: material fetch, emissive
 * rules, vertex interpolation, PBR (Lambert + GGX) BSDF sampling, next-event
 * estimation set-up, throughput / ray update and Russian roulette.
 *
 * Reference semantics reproduced (bit-exact f32 operation order):
 *   hit handling sequence            kernels/src/lib.rs:80-181
 *   barycentric / interpolation      kernels/src/util.rs:238-251, lib.rs:112-129
 *   normal map                       lib.rs:132-141
 *   get_pbr_bsdf / PBR::sample       kernels/src/bsdf.rs:354-387, 272-334
 *   PBR::evaluate / pdf (NEE side)   bsdf.rs:237-270, 336-351
 *   sample_direct_lighting           kernels/src/light_pick.rs:100-173
 *   calculate_bsdf_mis_contribution  light_pick.rs:179-199
 *   CPU image sampler                shared_structs/src/image_polyfill.rs:32-55
 *
 * Wavefront-specific structure: the shadow ray of light_pick.rs:141 is NOT traced
 * here.  The stage computes the contribution the light sample WOULD add if it
 * is unoccluded (everything in light_pick.rs:148-160 except the visibility
 * test depends only on data available here) and emits {ray, max_t,
 * contribution, slot} into the shadow queue; k_traverse_shadow adds it to the
 * path's radiance when the any-hit traversal finds nothing — before the next
 * bounce is shaded, which preserves the reference's f32 summation order of
 * `radiance` (lib.rs:98,106,164).
 */
#ifndef RPT_K_SHADE_H
#define RPT_K_SHADE_H

#include "k_common.h"
#include "k_path.h"

#define RPT_PI_F 3.14159265358979323846f
#define RPT_EPS 0.001f   /* util.rs:5 */

/* ---- CPU-polyfill image sampling (shared_structs/src/image_polyfill.rs:32-55) ----------------------------------------------------------
 * sample_raw: `coord.x as usize % width as usize` — the i32 sign-extends to 64 bits, then an UNSIGNED 64-bit remainder.  Written out that
 * is a 64-bit division by a run-time value, ~120 instructions, eight of them per lookup and four lookups per textured hit: rounds 1-5 spent
 * HALF the textured shade stage there (profiles/r06_texture_sampler.txt).  The same value, cheaply:
 *   a power-of-two extent (the reference's atlas is always 4096 x 4096, src/asset.rs:177): the remainder of the sign-extended value is its
 *     low bits — `coord & (extent - 1)`, for negative coordinates too;
 *   any extent: a coordinate in [0, extent] (every lookup of a uv inside its atlas rectangle: floor / ceil of uv * extent) wraps to itself,
 *     or to 0 at `extent`; only what is left over divides. */
__device__ __forceinline__ uint32_t image_wrap(int32_t c, uint32_t extent, bool pow2) {
    if (pow2) return (uint32_t)c & (extent - 1u);
    const uint32_t u = (uint32_t)c;
    if (u < extent) return u;
    if (u == extent) return 0u;
    return (uint32_t)((uint64_t)(int64_t)c % (uint64_t)extent);
}
__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float s) {
    return make_float4(a.x + ((b.x - a.x) * s), a.y + ((b.y - a.y) * s), a.z + ((b.z - a.z) * s), a.w + ((b.w - a.w) * s));
}
/* texel = Vec4(r, g, b, 255) / 255.0 (src/asset.rs:270); rptm::unorm8 is that division, exactly */
__device__ __forceinline__ float4 texel_of_u8(uint32_t t) {
    return make_float4(rptm::unorm8((float)(t & 0xffu)), rptm::unorm8((float)((t >> 8) & 0xffu)), rptm::unorm8((float)((t >> 16) & 0xffu)), 1.0f);
}
/* The two texels (x0, y) and (x1, y) of a footprint row.  x1 is x0 or its right neighbour (ceil vs floor of one number) unless the footprint wraps, so
 * both come from ONE 8-byte load of the texel pair that holds x0 — (x0, x0 + 1), or (x0 - 1, x0) in the last column — and only a wrapping footprint
 * issues a second load: 8 instead of 16 lane-divergent loads per textured hit. */
__device__ __forceinline__ void u8_row_pair(const uint32_t *texels, uint32_t row /* texel index of the row's first texel: an image has at most 2^30 texels (rpt_upload_scene) */, uint32_t x0, uint32_t x1, uint32_t width, uint32_t &t0, uint32_t &t1) {
    const bool edge = x0 + 1u >= width;
    uint2 pair;
    __builtin_memcpy(&pair, texels + (row + (edge ? x0 - 1u : x0)), sizeof(pair));
    t0 = edge ? pair.y : pair.x;
    t1 = x1 == x0 ? t0 : pair.y;
    if (x1 != x0 && (edge || x1 != x0 + 1u)) t1 = texels[row + x1];
}
template <bool IS_U8>
__device__ __forceinline__ float4 sample_by_lod(const DevImage &img, float u, float v) {
    float sx = u * (float)img.width, sy = v * (float)img.height;
    float fx = rptm::floorr(sx), fy = rptm::floorr(sy);
    float tx = sx - fx, ty = sy - fy;
    int32_t ix0 = rptm::f2i32_sat(fx), iy0 = rptm::f2i32_sat(fy);
    int32_t ix1 = rptm::f2i32_sat(rptm::ceilr(sx)), iy1 = rptm::f2i32_sat(rptm::ceilr(sy));
    const bool pow2 = ((img.width & (img.width - 1u)) | (img.height & (img.height - 1u))) == 0u;       /* (uniform) */
    const uint32_t x0 = image_wrap(ix0, img.width, pow2), x1 = image_wrap(ix1, img.width, pow2);
    const uint32_t y0 = image_wrap(iy0, img.height, pow2), y1 = image_wrap(iy1, img.height, pow2);
    float4 c00, c01, c10, c11;
    if (IS_U8) {
        const uint32_t *texels = reinterpret_cast<const uint32_t *>(img.texels);
        uint32_t t00, t10, t01, t11;
        if (img.width >= 2u) {                                                                         /* (uniform) */
            u8_row_pair(texels, y0 * img.width, x0, x1, img.width, t00, t10);
            u8_row_pair(texels, y1 * img.width, x0, x1, img.width, t01, t11);
        } else {
            t00 = t10 = texels[y0];
            t01 = t11 = texels[y1];
        }
        c00 = texel_of_u8(t00); c10 = texel_of_u8(t10); c01 = texel_of_u8(t01); c11 = texel_of_u8(t11);
    } else {
        const float4 *texels = reinterpret_cast<const float4 *>(img.texels);
        c00 = texels[y0 * img.width + x0]; c10 = texels[y0 * img.width + x1];
        c01 = texels[y1 * img.width + x0]; c11 = texels[y1 * img.width + x1];
    }
    float4 a = lerp4(c00, c10, tx);
    float4 b = lerp4(c01, c11, tx);
    return lerp4(a, b, ty);
}

/* ---- microfacet helpers (util.rs live subset) ------------------------------ */
__device__ __forceinline__ float ggx_d(F3 n, F3 h, float roughness) {
    float numerator = roughness * roughness;
    float ndh = rptm::fmaxr(dot3(n, h), 0.0f);
    float den = (ndh * ndh) * (numerator - 1.0f) + 1.0f;
    den = rptm::fmaxr(RPT_PI_F * (den * den), RPT_EPS);
    return numerator / den;
}
__device__ __forceinline__ float schlick_g1(F3 n, F3 v, float roughness) {
    float numerator = rptm::fmaxr(dot3(n, v), 0.0f);
    float r = (roughness * roughness) / 8.0f;
    float den = numerator * (1.0f - r) + r;
    return numerator / den;
}
__device__ __forceinline__ F3 fresnel_schlick3(float cos_theta, F3 f0) {
    return f0 + (f3s(1.0f) - f0) * rptm::powi5(1.0f - cos_theta);
}

struct Pbr {
    F3 albedo;
    float roughness, metallic;
    float clamp_lo, clamp_hi;

    /* bsdf.rs:275-280 / 244-248 */
    __device__ __forceinline__ float specular_weight(F3 view, F3 n) const {
        float c = rptm::fmaxr(dot3(n, view), 0.0f);
        float f0 = rptm::powi2((1.0f - 1.5f) / (1.0f + 1.5f));
        float approx_fresnel = f0 + (1.0f - f0) * rptm::powi5(1.0f - c);
        float w = approx_fresnel * (1.0f - metallic) + 1.0f * metallic;   /* util::lerp(a, 1.0, t) */
        if (w != 0.0f && w != 1.0f) {
            if (w < clamp_lo) w = clamp_lo;
            if (w > clamp_hi) w = clamp_hi;
        }
        return w;
    }
    __device__ __forceinline__ F3 ks_of(F3 view, F3 halfway) const {
        const float F0_SQRT = (1.5f - 1.0f) / (1.5f + 1.0f);
        F3 f0 = lerp3(f3s(F0_SQRT * F0_SQRT), albedo, metallic);
        return fresnel_schlick3(rptm::fmaxr(dot3(halfway, view), 0.0f), f0);
    }
    /* bsdf.rs:193-202 */
    __device__ __forceinline__ F3 diffuse_term(float cos_theta, float w, F3 ks) const {
        F3 kd = (f3s(1.0f) - ks) * (1.0f - metallic);
        F3 diffuse = kd * albedo / RPT_PI_F;
        return diffuse * cos_theta / (1.0f - w);
    }
    /* bsdf.rs:204-219 */
    __device__ __forceinline__ F3 specular_term(F3 view, F3 n, F3 l, float cos_theta, float d_term, float w, F3 ks) const {
        float g_term = schlick_g1(n, view, roughness) * schlick_g1(n, l, roughness);
        F3 num = (d_term * g_term) * ks;
        float den = 4.0f * rptm::fmaxr(dot3(n, view), 0.0f) * cos_theta;
        F3 specular = num / rptm::fmaxr(den, RPT_EPS);
        return specular * cos_theta / w;
    }
};

/* material record: 6 float4 (shared_structs/src/lib.rs:44-56) */
struct Mat {
    float4 emissive, albedo, roughness, metallic, normals;
    uint4 has;   /* albedo, metallic, roughness, normal texture flags */
};
template <bool TEXTURED>
__device__ __forceinline__ Mat load_material(const DevScene &sc, uint32_t index) {
    Mat m;
    if (TEXTURED) {
        const float4 *p = sc.materials + 6u * index;
        m.emissive = p[0]; m.albedo = p[1]; m.roughness = p[2]; m.metallic = p[3]; m.normals = p[4];
        float4 f = p[5];
        m.has = make_uint4(__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z), __float_as_uint(f.w));
    } else {
        /* untextured scene: everything the BSDF needs sits in 32 bytes */
        const float4 *p = sc.mat_lite + 2u * index;
        float4 a = p[0], b = p[1];
        m.emissive = make_float4(a.x, a.y, a.z, 0.0f);
        m.albedo = make_float4(b.x, b.y, b.z, 0.0f);
        m.roughness = make_float4(a.w, 0, 0, 0);
        m.metallic = make_float4(b.w, 0, 0, 0);
        m.normals = make_float4(0, 0, 0, 0);
        m.has = make_uint4(0u, 0u, 0u, 0u);
    }
    return m;
}


/* NEE = NextEventEstimation mode (0 none, 1 MIS, 2 direct only), TEXTURED = the
 * scene has at least one texture flag.  Specialising removes the dead halves of
 * the stage (and their registers) for the common untextured / no-NEE case. */
/* What the stage does for ONE traversed slot (hit word `hw`): everything of lib.rs:64-181 after the intersection.  Outputs:
 * to_sky (a miss: queued for k_sky), emit_shadow + the shadow-queue entry; a path that ends here with nothing pending is
 * finished through finish_in_side_stage (k_path.h). */
template <int NEE, bool TEXTURED, bool COMPACT>
__device__ __forceinline__ void shade_slot(const DevScene &sc, const DevState &st, const DevQueues &q, const DevConfig &cfg, DevStats *stats,
                                           uint32_t slot, float2 hw, bool active, bool &to_sky, bool &emit_shadow, float4 &sh_o, float4 &sh_d,
                                           float4 &sh_c, bool first /* iteration 0 of the call: every path is a first path (k_path.h) */,
                                           uint32_t n_samples, bool &elided /* an NEE evaluation whose shadow ray decides nothing: not queued */,
                                           bool last_iteration /* wave-uniform: every path of this launch is at its last bounce and has a radiance record */) {
    const uint32_t hit_tri = __float_as_uint(hw.y);
    if (NEE == RPT_NEE_NONE && last_iteration && active && hit_tri != HIT_MISS) {
        /* The last bounce of every path of this launch, without NEE (see `last` below): all the stage can still do for a hit is add the emission of a
         * front-facing emitter (lib.rs:86-100) and end the path — so it reads the hit word it was handed, the triangle's material index and emission, and
         * nothing else: no ray, no throughput unless the surface emits, and where nothing is added the slot's radiance record (radiance, samples owed) already
         * holds what finish_in_side_stage would write into it. */
        const uint32_t tri_index = hit_tri & 0x7fffffffu;
        const uint32_t m = __float_as_uint(sc.tri_shade[4u * tri_index + 2u].w);
        const float4 e4 = TEXTURED ? sc.materials[6u * m] : sc.mat_lite[2u * m];
        const bool emits = (e4.x != 0.0f || e4.y != 0.0f || e4.z != 0.0f) && (hit_tri >> 31) == 0u;
        if (emits || st.group_shift == 0u) {
            const float4 r4 = st.rad[slot];
            F3 radiance = f3(r4.x, r4.y, r4.z);
            if (emits) {
                const float4 tf = st.thr[slot];
                radiance = radiance + mask_nan3(f3(tf.x, tf.y, tf.z) * f3(e4.x, e4.y, e4.z));
            }
            finish_in_side_stage(st, cfg, slot, radiance, __float_as_uint(r4.w));
        } else {
            st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_DONE));
        }
        return;
    }
    if (active) {
        const float4 ra = st.ray_a[slot];
        const float2 rb = st.ray_b[slot];
        const F3 ro = f3(ra.x, ra.y, ra.z), rd = f3(ra.w, rb.x, rb.y);
        const float hit_t = hw.x;
        if (hit_tri == HIT_MISS) {
            to_sky = true;                           /* lib.rs:66-79: shaded by k_sky, which also ends the path */
            st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PARKED));
            if (first) {                             /* k_sky reads the path state */
                st.thr[slot] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(RPT_FRESH_FLAGS));
                st.rad[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(first_path_todo(st, slot, n_samples)));
            }
        } else {
            const float4 tf = first ? make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(RPT_FRESH_FLAGS)) : st.thr[slot];
            F3 throughput = f3(tf.x, tf.y, tf.z);
            /* radiance + samples still owed: read only by a lane whose path adds emission or ends here (never written back by
             * this stage mid-path: the NEE terms are added by the shadow stage, everything else ends the path) */
            F3 radiance = f3s(0.0f);
            uint32_t todo = 0u;
            bool rad_loaded = false;
            auto load_rad = [&]() {
                if (!rad_loaded) {
                    if (first) {
                        todo = first_path_todo(st, slot, n_samples);     /* radiance 0 */
                    } else {
                        const float4 r4 = st.rad[slot];
                        radiance = f3(r4.x, r4.y, r4.z);
                        todo = __float_as_uint(r4.w);
                    }
                    rad_loaded = true;
                }
            };
            const uint32_t flags = __float_as_uint(tf.w);
            const uint32_t bounce = FLAG_BOUNCE(flags);
            const bool last_spec = FLAG_LOBE_SPEC(flags) != 0u;
            constexpr bool nee = NEE != RPT_NEE_NONE;
            const bool backface = (hit_tri >> 31) != 0u;
            const uint32_t tri_index = hit_tri & 0x7fffffffu;
            /* per-triangle shading record: normals, uvs, material in 64 contiguous bytes */
            const float4 *ts = sc.tri_shade + 4u * tri_index;
            const float4 s0 = ts[0], s1 = ts[1], s2 = ts[2];
            const Mat mat = load_material<TEXTURED>(sc, __float_as_uint(s2.w));
            const F3 hit = ro + rd * hit_t;

            bool done = false;
            const F3 emissive = xyz4(mat.emissive);
            if (emissive.x != 0.0f || emissive.y != 0.0f || emissive.z != 0.0f) {       /* lib.rs:86 */
                if (backface) {
                    done = true;                                                         /* :88-90 */
                } else if (!nee || bounce == 0u || last_spec) {                          /* :97-100 */
                    load_rad();
                    radiance = radiance + mask_nan3(throughput * emissive);
                    done = true;
                } else if (NEE == RPT_NEE_MIS) {                                         /* :104-108, last lobe is diffuse here */
                    /* last_light_sample / last_bsdf_sample (lib.rs:59-60): the light sample is carried as the table entry it
                     * came from (its area, pick pdf, normal and emission are functions of the entry) + the throughput before
                     * that bounce; the BSDF sample as pdf + spectrum — 32 bytes per slot instead of 64 */
                    const float4 ma = st.mis_a[slot], mb = st.mis_b[slot];
                    const uint32_t code = __float_as_uint(ma.x);
                    F3 contribution = f3s(0.0f);
                    if (code != 0xffffffffu) {                                           /* (0xffffffff: no light table, DirectLightSample::default()) */
                        const rpt_light_pick_entry e = sc.light_pick[code >> 1];
                        const bool side_b = (code & 1u) != 0u;
                        const uint32_t light_tri = side_b ? e.triangle_index_b : e.triangle_index_a;
                        if (tri_index == light_tri) {                                    /* light_pick.rs:185 */
                            const float4 *lr = sc.light_rec + 8u * (code >> 1) + (side_b ? 4u : 0u);
                            const float light_area = side_b ? e.triangle_area_b : e.triangle_area_a;
                            const float light_pick_pdf = side_b ? e.triangle_pick_pdf_b : e.triangle_pick_pdf_a;
                            F3 light_normal = f3(lr[0].w, lr[1].w, lr[2].w);
                            float cos_theta = dot3(light_normal, -rd);                  /* last sampled_direction == rd */
                            float light_pdf = cos_theta <= 0.0f ? 0.0f : rptm::powi2(hit_t) / (light_area * cos_theta);
                            if (light_pdf > 0.0f) {
                                float bsdf_pdf = mb.x;
                                float p1 = bsdf_pdf * bsdf_pdf;
                                float weight = p1 / (p1 + light_pdf * light_pdf);
                                F3 spectrum = f3(mb.y, mb.z, mb.w), emission = xyz4(lr[3]);
                                F3 direct = (spectrum * emission * weight / bsdf_pdf) / light_pick_pdf;
                                contribution = f3(ma.y, ma.z, ma.w) * direct;
                            }
                        }
                    }
                    load_rad();
                    radiance = radiance + mask_nan3(contribution);
                    done = true;
                }
                /* nee == direct-only, diffuse bounce > 0: fall through, shade the light as a surface */
            }

            uint32_t new_flags = flags;
            F3 new_o = ro, new_d = rd;
            /* The LAST bounce of a path (bounce + 1 == max_bounces: the loop of lib.rs:62 ends after it whatever the roulette says).  The reference still
             * samples the BSDF there (lib.rs:144-146) and updates throughput and ray (:168-172), and then drops all of it: nothing reads the sampled direction,
             * pdf or spectrum, last_bsdf_sample or last_light_sample again.  What the sample's radiance can still receive is the emission handled above and the
             * NEE term of this hit — which needs the hit point, the normal, the BSDF's parameters, the three draws of PBR::sample (for the lobe choice and to
             * keep the later draws on their dimensions) and nothing of the sampled lobe.  Without NEE the whole surface body is dead: the path just ends.
             * In a batch of known length the last bounce is the last ITERATION — wave-uniform: the fourth shade launch of a DarkCornell batch sheds its
             * ~1 300-instruction surface body (0.70 -> 0.3 ms). */
            const bool last = bounce + 1u >= cfg.c.max_bounces;
            if (!nee && last) done = true;
            if (!done) {
                /* ---- interpolate vertex data (lib.rs:112-129); d00/d01/d11 are triangle constants ---- */
                const float4 *tg = sc.tri_geom + 3u * tri_index;
                const float4 g0 = tg[0], g1 = tg[1], g2 = tg[2];
                F3 bary;
                {
                    F3 v0 = xyz4(g1), v1 = xyz4(g2), v2 = hit - xyz4(g0);
                    float d00 = g0.w, d01 = g1.w, d11 = g2.w;
                    float d20 = dot3(v2, v0), d21 = dot3(v2, v1);
                    float denom = d00 * d11 - d01 * d01;
                    float v = (d11 * d20 - d01 * d21) / denom;
                    float w = (d00 * d21 - d01 * d20) / denom;
                    bary = f3(1.0f - v - w, v, w);
                }
                F3 normal = bary.x * xyz4(s0) + bary.y * xyz4(s1) + bary.z * xyz4(s2);
                float uv_x = 0.0f, uv_y = 0.0f;
                if (TEXTURED) {
                    const float4 s3 = ts[3];     /* (uvb.x, uvb.y, uvc.x, uvc.y); uva in s0.w, s1.w */
                    uv_x = (bary.x * s0.w + bary.y * s3.x) + bary.z * s3.z;
                    uv_y = (bary.x * s1.w + bary.y * s3.y) + bary.z * s3.w;
                    float cx = rptm::fminr(rptm::fmaxr(uv_x, 0.0f), 1.0f), cy = rptm::fminr(rptm::fmaxr(uv_y, 0.0f), 1.0f);
                    if (cx != uv_x || cy != uv_y) {
                        uv_x = uv_x - rptm::floorr(uv_x);
                        uv_y = uv_y - rptm::floorr(uv_y);
                    }
                    if (mat.has.w != 0u) {                                               /* lib.rs:132-141 */
                        float su = mat.normals.x + uv_x * mat.normals.z, sv = mat.normals.y + uv_y * mat.normals.w;
                        float4 s = sample_by_lod<true>(sc.atlas, su, sv);
                        F3 nm = f3(s.x * 2.0f - 1.0f, s.y * 2.0f - 1.0f, s.z * 2.0f - 1.0f);
                        const float4 *tt = sc.tri_tangent + 3u * tri_index;                  /* the three vertex tangents, gathered at upload */
                        F3 tangent = bary.x * xyz4(tt[0]) + bary.y * xyz4(tt[1]) + bary.z * xyz4(tt[2]);
                        F3 bitangent = cross3(tangent, normal);
                        F3 r = tangent * nm.x;
                        r = r + (bitangent * nm.y);
                        r = r + (normal * nm.z);
                        normal = norm3(r);
                    }
                }

                /* ---- get_pbr_bsdf (bsdf.rs:354-387) ---- */
                Pbr bsdf;
                bsdf.albedo = xyz4(mat.albedo);
                float roughness = mat.roughness.x, metallic = mat.metallic.x;
                if (TEXTURED) {
                    if (mat.has.x != 0u) {
                        float4 s = sample_by_lod<true>(sc.atlas, mat.albedo.x + uv_x * mat.albedo.z, mat.albedo.y + uv_y * mat.albedo.w);
                        bsdf.albedo = f3(s.x, s.y, s.z);
                    }
                    if (mat.has.z != 0u)
                        roughness = sample_by_lod<true>(sc.atlas, mat.roughness.x + uv_x * mat.roughness.z, mat.roughness.y + uv_y * mat.roughness.w).x;
                    if (mat.has.y != 0u)
                        metallic = sample_by_lod<true>(sc.atlas, mat.metallic.x + uv_x * mat.metallic.z, mat.metallic.y + uv_y * mat.metallic.w).x;
                }
                bsdf.roughness = rptm::fmaxr(roughness, RPT_EPS);
                bsdf.metallic = rptm::fminr(metallic, 1.0f - RPT_EPS);
                bsdf.clamp_lo = cfg.c.specular_weight_clamp[0];
                bsdf.clamp_hi = cfg.c.specular_weight_clamp[1];

                /* ---- PBR::sample (bsdf.rs:272-334) ---- */
                const uint2 rs = st.rng[slot_pix(st, slot)];
                Rng rng{rs.x + slot_k(st, slot) + rs.y, FLAG_DIM(flags)};
                const float r1 = rng.next(), r2 = rng.next(), r3 = rng.next();
                const F3 view = -rd;
                const float w_spec = bsdf.specular_weight(view, normal);
                F3 sdir = f3s(0.0f);
                const bool spec = !(r3 >= w_spec);
                float pdf = 1.0f;
                F3 spectrum = f3s(0.0f);
                if (!last) {
                /* Both lobes turn one random number into an azimuth and take its sine and cosine (util.rs:27-28, 70), and
                 * both normalise their final direction (util.rs:31, 84): in a wave that holds both kinds of lanes — nearly
                 * every wave — the divergent branches would each issue those ~130 instructions.  They are issued once,
                 * on the lane's own operand. */
                float sin_p, cos_p;
                rptm::sincosr(2.0f * RPT_PI_F * (spec ? r1 : r2), sin_p, cos_p);
                if (!spec) {
                    /* create_cartesian(normal) (util.rs:34-40) */
                    F3 temp_vec = norm3(cross3(normal, f3(0.1f, 0.5f, 0.9f)));
                    F3 nt = norm3(cross3(temp_vec, normal));       /* right   */
                    F3 nb = norm3(cross3(normal, nt));             /* forward */
                    /* cosine_sample_hemisphere (util.rs:24-32) */
                    float theta = rptm::acosr(rptm::sqrtr(r1));
                    float sin_t, cos_t;
                    rptm::sincosr(theta, sin_t, cos_t);
                    F3 s = f3(sin_t * cos_p, cos_t, sin_t * sin_p);
                    sdir = f3(s.x * nb.x + s.y * normal.x + s.z * nt.x,
                              s.x * nb.y + s.y * normal.y + s.z * nt.y,
                              s.x * nb.z + s.y * normal.z + s.z * nt.z);
                } else {
                    /* reflect(-view, n) then sample_ggx (util.rs:42-44, 67-85) */
                    F3 inc = -view;
                    F3 refl = inc - normal * 2.0f * dot3(inc, normal);
                    float a = bsdf.roughness * bsdf.roughness;
                    float cos_theta = rptm::sqrtr((1.0f - r2) / (r2 * (a * a - 1.0f) + 1.0f));
                    float sin_theta = rptm::sqrtr(1.0f - cos_theta * cos_theta);
                    F3 h = f3(cos_p * sin_theta, sin_p * sin_theta, cos_theta);
                    F3 up = rptm::absr(refl.z) < 0.999f ? f3(0.0f, 0.0f, 1.0f) : f3(1.0f, 0.0f, 0.0f);
                    F3 tangent = norm3(cross3(up, refl));
                    F3 bitangent = cross3(refl, tangent);
                    sdir = tangent * h.x + bitangent * h.y + refl * h.z;
                }
                sdir = norm3(sdir);
                const float cos_theta = rptm::fmaxr(dot3(normal, sdir), RPT_EPS);
                const F3 halfway = norm3(view + sdir);
                const F3 ks = bsdf.ks_of(view, halfway);
                if (!spec) {
                    pdf = cos_theta / RPT_PI_F;
                    spectrum = bsdf.diffuse_term(cos_theta, w_spec, ks);
                } else {
                    float d_term = ggx_d(normal, halfway, bsdf.roughness);
                    pdf = (d_term * dot3(normal, halfway)) / (4.0f * dot3(view, halfway));
                    spectrum = bsdf.specular_term(view, normal, sdir, cos_theta, d_term, w_spec, ks);
                }
                }       /* !last */

                /* ---- next-event estimation set-up (light_pick.rs:100-173) ---- */
                if (nee && !spec) {
                    if (sc.no_lights) {
                        /* sentinel: DirectLightSample::default() — zero contribution, zeroed carry */
                        if (NEE == RPT_NEE_MIS && !last) st.mis_a[slot] = make_float4(__uint_as_float(0xffffffffu), 0, 0, 0);
                    } else {
                        const float l1 = rng.next(), l2 = rng.next();
                        uint32_t idx = rptm::f2u32_sat(l1 * (float)sc.n_light_pick);
                        if (idx >= sc.n_light_pick) {                       /* gen_r1() == 1.0: the reference panics (Appendix C) */
                            idx = sc.n_light_pick - 1u;
                            atomicAdd(&stats->light_index_clamped, 1ull);
                        }
                        const rpt_light_pick_entry e = sc.light_pick[idx];
                        const bool pick_a = l2 < e.ratio;
                        const float light_area = pick_a ? e.triangle_area_a : e.triangle_area_b;
                        const float light_pick_pdf = pick_a ? e.triangle_pick_pdf_a : e.triangle_pick_pdf_b;
                        /* the light triangle's corners, mean normal and emission: one 64-byte record built at upload
                         * (was: index buffer -> three 64-byte vertices + the material, four dependent scattered loads) */
                        const float4 *lr = sc.light_rec + 8u * idx + (pick_a ? 0u : 4u);
                        const float4 lra = lr[0], lrb = lr[1], lrc = lr[2], lre = lr[3];
                        const F3 light_normal = f3(lra.w, lrb.w, lrc.w);
                        const F3 light_emission = xyz4(lre);
                        const float p1 = rng.next(), p2 = rng.next();
                        const float r1_sqrt = rptm::sqrtr(p1);
                        const F3 light_point = (1.0f - r1_sqrt) * xyz4(lra) + (r1_sqrt * (1.0f - p2)) * xyz4(lrb) +
                                               (r1_sqrt * p2) * xyz4(lrc);
                        const F3 unorm = light_point - hit;
                        const float light_distance = len3(unorm);
                        const F3 light_direction = unorm / light_distance;

                        /* everything after the visibility test, assuming it passes */
                        F3 direct = f3s(0.0f);
                        {
                            float cos_l = dot3(light_normal, -light_direction);
                            float light_pdf = cos_l <= 0.0f ? 0.0f : rptm::powi2(light_distance) / (light_area * cos_l);
                            if (light_pdf > 0.0f) {
                                /* PBR::evaluate(view, n, L, Diffuse) and PBR::pdf(.., Diffuse) */
                                float w_e = bsdf.specular_weight(view, normal);
                                float cos_e = rptm::fmaxr(dot3(normal, light_direction), 0.0f);
                                F3 h_e = norm3(view + light_direction);
                                F3 ks_e = bsdf.ks_of(view, h_e);
                                F3 attenuation = bsdf.diffuse_term(cos_e, w_e, ks_e);
                                float bsdf_pdf = cos_e / RPT_PI_F;
                                if (bsdf_pdf > 0.0f) {
                                    float weight = 1.0f;
                                    if (NEE == RPT_NEE_MIS) {
                                        float q1 = light_pdf * light_pdf;
                                        weight = q1 / (q1 + bsdf_pdf * bsdf_pdf);
                                    }
                                    direct = (attenuation * light_emission * weight / light_pdf) / light_pick_pdf;
                                }
                            }
                        }
                        const F3 contribution = throughput * direct;
                        /* The reference traces the shadow ray first and looks at light_pdf / bsdf_pdf afterwards (light_pick.rs:141-158).  Where the term
                         * an UNOCCLUDED ray adds is zero — the light point faces away (light_pdf = 0), lies below this surface's horizon (bsdf_pdf = 0), or
                         * the product is masked as non-finite — `radiance += mask_nan(term)` (lib.rs:164) leaves every bit of radiance as it was whatever the
                         * walk finds (radiance starts at +0.0 and x + y is -0.0 only for two negative zeros: it is never -0.0, so x + (+-0.0) == x), and
                         * nothing else reads `.hit`: the ray is not queued.  48 % of DarkCornell's shadow rays, 57 % of VeachMIS's (tools/dead_shadow_rays.py);
                         * counted in rpt_stats.shadow_rays_elided, and rpt_stats.shadow_rays keeps counting what the reference executes. */
                        const F3 term = mask_nan3(contribution);
                        const bool decides = term.x != 0.0f || term.y != 0.0f || term.z != 0.0f;
                        elided = !decides;
                        const F3 so = hit + light_direction * RPT_EPS;
                        emit_shadow = decides;
                        sh_o = make_float4(so.x, so.y, so.z, light_distance - RPT_EPS * 2.0f);
                        sh_d = make_float4(light_direction.x, light_direction.y, light_direction.z, 0.0f);
                        sh_c = make_float4(contribution.x, contribution.y, contribution.z, 0.0f);
                        if (NEE == RPT_NEE_MIS && !last)      /* (the carry of a last bounce has no reader) */
                            st.mis_a[slot] = make_float4(__uint_as_float(2u * idx + (pick_a ? 0u : 1u)), throughput.x, throughput.y, throughput.z);
                    }
                }
                if (last) {
                    done = true;
                } else {
                    if (NEE == RPT_NEE_MIS) st.mis_b[slot] = make_float4(pdf, spectrum.x, spectrum.y, spectrum.z);

                    /* ---- attenuate, respawn, roulette (lib.rs:168-181) ---- */
                    throughput = throughput * (spectrum / pdf);
                    new_d = sdir;
                    new_o = hit + sdir * RPT_EPS;
                    if (bounce > cfg.c.min_bounces) {
                        float prob = rptm::fmaxr(throughput.x, rptm::fmaxr(throughput.y, throughput.z));
                        if (rng.next() > prob) {
                            done = true;
                        } else {
                            throughput = throughput * (1.0f / prob);
                        }
                    }
                    new_flags = MAKE_FLAGS(bounce + 1u, spec ? 1u : 0u, rng.dim);
                }
            }

            if (done && !emit_shadow) {
                /* the path ends here with nothing pending: one slot per pixel — accumulated and restarted on the spot; several —
                 * parked as HIT_DONE for k_complete */
                load_rad();
                finish_in_side_stage(st, cfg, slot, radiance, todo);
            } else {
                if (first) {                         /* the path goes on (or waits for its shadow ray): its radiance record begins here */
                    load_rad();
                    st.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, __uint_as_float(todo));
                }
                if (!done) {
                    st.thr[slot] = make_float4(throughput.x, throughput.y, throughput.z, __uint_as_float(new_flags));
                    st.ray_a[slot] = make_float4(new_o.x, new_o.y, new_o.z, new_d.x);
                    st.ray_b[slot] = make_float2(new_d.y, new_d.z);
                    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PENDING));
                    sh_d.w = __uint_as_float(slot);
                } else {
                    /* the shadow stage adds the NEE term and then finishes the path */
                    st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_PARKED));
                    sh_d.w = __uint_as_float(slot | 0x80000000u);
                }
            }
        }
    }

}

/* COMPACT: a workgroup owns RPT_SHADE_ROUNDS x 256 consecutive slots; it first walks them in identity layout to pack the slots
 * that were traversed in this iteration into an LDS list, then shades the list 256 at a time.
 * Why (profiles/r02_pbrtest_pmc_sq.txt, r02_veachmis): on an open scene most slots are parked after the first bounce
 * (their paths ended in the sky and wait for the siblings of their generation).  One thread per slot, the stage ran 2.1 M
 * waves per pass on PBRTest with 32 slots per pixel, two thirds of their cycles waiting (a dependent load or two, then
 * four workgroup barriers for side queues nothing is pushed to), 40 % of the lanes live in what was issued — the fixed
 * cost per wave, not the shading, was the stage.  Packed, a workgroup does the bookkeeping once per 2 048 slots with
 * eight independent loads in flight per thread, and only full waves shade.  (Generations are completed by k_complete.) */
#ifndef RPT_SHADE_ROUNDS
#define RPT_SHADE_ROUNDS 8
#endif
__device__ __forceinline__ uint32_t block_rank(bool pred, uint32_t *scratch, uint32_t &total) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    constexpr uint32_t NW = RPT_BLOCK / RPT_WAVE;
    const unsigned long long mask = rpt_ballot(pred);
    if (lane == 0u) scratch[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0u, sum = 0u;
    for (uint32_t w = 0; w < NW; ++w) {
        if (w == wave) base = sum;
        sum += scratch[w];
    }
    total = sum;
    __syncthreads();     /* scratch may be reused */
    return base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

/* the same for two disjoint predicates in one pass (one pair of barriers): ranks and totals of both */
__device__ __forceinline__ void block_rank2(bool a, bool b, uint32_t *scratch, uint32_t &rank_a, uint32_t &rank_b, uint32_t &total_a, uint32_t &total_b) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    constexpr uint32_t NW = RPT_BLOCK / RPT_WAVE;
    const unsigned long long ma = rpt_ballot(a), mb = rpt_ballot(b);
    if (lane == 0u) scratch[wave] = (uint32_t)__popcll(ma) | ((uint32_t)__popcll(mb) << 16);
    __syncthreads();
    uint32_t base = 0u, sum = 0u;
    for (uint32_t w = 0; w < NW; ++w) {
        if (w == wave) base = sum;
        sum += scratch[w];                   /* (each half stays below 2^16: at most RPT_BLOCK entries) */
    }
    total_a = sum & 0xffffu; total_b = sum >> 16;
    __syncthreads();     /* scratch may be reused */
    rank_a = (base & 0xffffu) + __builtin_amdgcn_mbcnt_hi((uint32_t)(ma >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ma, 0u));
    rank_b = (base >> 16) + __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
}

/* side-queue emission of one pass over up to 256 slots: wave64 ballot + mbcnt prefix, one atomic per workgroup
 * (block-uniform early outs keep the barriers inside block_push legal) */
template <int NEE>
__device__ __forceinline__ void shade_emit(const DevQueues &q, uint32_t *push_scratch, uint32_t slot, bool to_sky, bool emit_shadow,
                                           float4 sh_o, float4 sh_d, float4 sh_c) {
    const uint32_t shard = blockIdx.x % RPT_Q_SHARDS;
    uint32_t at;
    if (__syncthreads_or(to_sky)) {
        at = q_position(shard, block_push(&q.sky_cnt[shard * RPT_Q_SHARD_STRIDE], to_sky, push_scratch));
        if (to_sky) q.sky[at] = slot;
    }
    if (NEE != RPT_NEE_NONE && __syncthreads_or(emit_shadow)) {
        at = q_position(shard, block_push(&q.shadow_cnt[shard * RPT_Q_SHARD_STRIDE], emit_shadow, push_scratch));
        if (emit_shadow) {
            q.sh_o[at] = sh_o;
            q.sh_d[at] = sh_d;
            q.sh_c[at] = sh_c;
        }
    }
}

/* wave-aggregated count into the workgroup's LDS word */
__device__ __forceinline__ void count_elided(uint32_t *lds_counter, bool elided) {
    const unsigned long long m = rpt_ballot(elided);
    if (m != 0ull && __lane_id() == (uint32_t)__ffsll((long long)m) - 1u) atomicAdd(lds_counter, (uint32_t)__popcll(m));
}

template <int NEE, bool TEXTURED, bool COMPACT>
/* Occupancy asked of the compiler where it costs no spill (left alone it stops at 68 and 104 VGPRs): the plain nee = 0
 * variant runs at 8 waves per SIMD in 64 VGPRs (DarkCornell shade 31.7 -> 31.0 ms per 8 batches), its packed form at 5 in 96
 * (PBRTest 69.3 -> 67.4 per 4).  The NEE variants spill when pushed (72 VGPRs: 20-44 bytes of scratch; packed: VeachMIS shade 44.0 -> 48.6) and are left alone. */
#ifndef RPT_SHADE_WAVES_PLAIN
#define RPT_SHADE_WAVES_PLAIN 8
#endif
#ifndef RPT_SHADE_WAVES_PACKED
#define RPT_SHADE_WAVES_PACKED 5
#endif
__attribute__((amdgpu_waves_per_eu((NEE == RPT_NEE_NONE && !TEXTURED) ? (COMPACT ? RPT_SHADE_WAVES_PACKED : RPT_SHADE_WAVES_PLAIN) : 1, 8)))
__global__ __launch_bounds__(RPT_BLOCK) void k_shade(DevScene sc, DevState st, DevQueues q, DevConfig cfg, uint32_t iteration,
                                                     DevStats *stats, uint32_t n_samples /* of this render call */) {
    const bool first_paths = iteration == 0u;                  /* every traversed slot holds the first path of its call (k_path.h) */
    /* a batch of known length: iteration i shades bounce i of every path, so its last iteration is every path's last bounce (not the first one: a first
       path has no radiance record yet) */
    const bool last_iteration = q.known_length != 0u && iteration != 0u && iteration + 1u >= cfg.c.max_bounces;
    __shared__ uint32_t push_scratch[RPT_BLOCK / RPT_WAVE + 1];
    __shared__ uint32_t c_slot[COMPACT ? RPT_BLOCK * RPT_SHADE_ROUNDS : 1];
    __shared__ uint32_t n_elided;                              /* NEE evaluations of this workgroup whose shadow ray was not queued (shade_slot) */
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    if (NEE != RPT_NEE_NONE) {
        if (threadIdx.x == 0u) n_elided = 0u;
        __syncthreads();                                       /* before any wave's count_elided (one barrier per workgroup; block-uniform: the returns above are) */
    }
    if (NEE != RPT_NEE_NONE && blockIdx.x == 0u && threadIdx.x == 0u) q.count[Q_SPOOL] = 0u;   /* the shadow stage that follows starts its pool at entry 0 */
    if (COMPACT) {
        const uint32_t base = blockIdx.x * (RPT_BLOCK * RPT_SHADE_ROUNDS);
        if (base >= st.n_slots) return;                        /* block-uniform */
        float2 hws[RPT_SHADE_ROUNDS];
#pragma unroll
        for (int r = 0; r < RPT_SHADE_ROUNDS; ++r) {           /* all looks in flight before the first is used */
            const uint32_t s0 = base + (uint32_t)r * RPT_BLOCK + threadIdx.x;
            hws[r] = s0 < st.n_slots ? st.hit[s0] : make_float2(0.0f, __uint_as_float(HIT_PARKED));
        }
        /* Binned by what the stage will do with the slot (lib.rs:64-79 vs :80-181): HITS fill the list from the front, MISSES from the back, and the
         * hits are padded to whole waves — so no wave holds both.  A miss only parks its slot and queues it for the sky stage (~25 instructions); in
         * slot order it sat beside lanes running the ~1 400-instruction surface body: from the second bounce on 40-80 % of the traversed slots of an
         * open scene are misses, the stage ran at 45 % lanes (profiles/r04_{veachmis,pbrtest}_pmc_sq.txt; tools/shade_bin_sim.py replays it:
         * 26-31 % fewer wave-instructions).  The order inside the list is no part of any result: every entry touches only its own slot. */
        constexpr uint32_t CAP = RPT_BLOCK * RPT_SHADE_ROUNDS;
        uint32_t total_h = 0u, total_m = 0u;                   /* block-uniform */
#pragma unroll
        for (int r = 0; r < RPT_SHADE_ROUNDS; ++r) {
            const uint32_t s0 = base + (uint32_t)r * RPT_BLOCK + threadIdx.x;
            const uint32_t word = __float_as_uint(hws[r].y);
            const bool is_hit = word < HIT_IDLE, is_miss = word == HIT_MISS;
            uint32_t ih, im, nh, nm;
            block_rank2(is_hit, is_miss, push_scratch, ih, im, nh, nm);
            if (is_hit) c_slot[total_h + ih] = s0;
            if (is_miss) c_slot[CAP - 1u - (total_m + im)] = s0;
            total_h += nh; total_m += nm;
        }
        __syncthreads();
        const uint32_t hits_padded = (total_h + RPT_WAVE - 1u) & ~(uint32_t)(RPT_WAVE - 1u);
        const uint32_t total = hits_padded + total_m;
        for (uint32_t first = 0u; first < total; first += RPT_BLOCK) {      /* block-uniform trip count */
            const uint32_t v = first + threadIdx.x;
            const bool a_hit = v < total_h, a_miss = v >= hits_padded && v < total;
            const bool active = a_hit || a_miss;
            const uint32_t slot = a_hit ? c_slot[v] : (a_miss ? c_slot[CAP - 1u - (v - hits_padded)] : 0u);
            const float2 hw = a_hit ? st.hit[slot] : make_float2(0.0f, __uint_as_float(a_miss ? HIT_MISS : HIT_PARKED));
            bool to_sky = false, emit_shadow = false, elided = false;
            float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, sh_c = sh_o;
            shade_slot<NEE, TEXTURED, true>(sc, st, q, cfg, stats, slot, hw, active, to_sky, emit_shadow, sh_o, sh_d, sh_c, first_paths, n_samples, elided, last_iteration);
            if (NEE != RPT_NEE_NONE) count_elided(&n_elided, elided);
            shade_emit<NEE>(q, push_scratch, slot, to_sky, emit_shadow, sh_o, sh_d, sh_c);
        }
    } else {
        const uint32_t slot = blockIdx.x * RPT_BLOCK + threadIdx.x;
        float2 hw = make_float2(0.0f, __uint_as_float(HIT_PARKED));
        if (slot < st.n_slots) hw = st.hit[slot];
        const uint32_t hit_tri = __float_as_uint(hw.y);
        const bool active = hit_tri < HIT_IDLE || hit_tri == HIT_MISS;        /* traversed in this iteration */
        bool to_sky = false, emit_shadow = false, elided = false;
        float4 sh_o = make_float4(0, 0, 0, 0), sh_d = sh_o, sh_c = sh_o;
        shade_slot<NEE, TEXTURED, false>(sc, st, q, cfg, stats, slot, hw, active, to_sky, emit_shadow, sh_o, sh_d, sh_c, first_paths, n_samples, elided, last_iteration);
        if (NEE != RPT_NEE_NONE) count_elided(&n_elided, elided);
        shade_emit<NEE>(q, push_scratch, slot, to_sky, emit_shadow, sh_o, sh_d, sh_c);
    }
    if (NEE != RPT_NEE_NONE) {
        /* one sharded, non-returning atomic per workgroup (element 1 of the shard's line: element 0 counts extension rays) */
        __syncthreads();
        if (threadIdx.x == 0u && n_elided != 0u) atomicAdd(&q.ray_shards[(blockIdx.x % RPT_STAT_SHARDS) * RPT_STAT_STRIDE + 1u], (unsigned long long)n_elided);
    }
}

#endif /* RPT_K_SHADE_H */
