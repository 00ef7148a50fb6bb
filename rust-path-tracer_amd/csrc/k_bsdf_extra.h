/*
 * k_bsdf_extra.h — the two BSDFs of the reference's kernels crate that trace_pixel never instantiates:
 * Lambertian (kernels/src/bsdf.rs:46-105) and Glass (bsdf.rs:107-176, with util.rs:117-142 sample_ggx_microsurface_normal
 * and util.rs:233-236 fresnel_schlick_scalar).  SURVEY.md 8f N4: dead code in the reference (get_pbr_bsdf is the only
 * constructor trace_pixel calls, lib.rs:144), kept here in the reference's operation order so that a material flag can
 * select them the day the reference does; reachable today through the rpt_debug_bsdf test hook only, which
 * tests/test_gpu_bsdf_extra.py compares with the oracle's restatement bit for bit.
 */
#ifndef RPT_K_BSDF_EXTRA_H
#define RPT_K_BSDF_EXTRA_H

#include "k_shade.h"

struct BsdfSample {
    float pdf;
    uint32_t lobe;        /* LobeType: 0 diffuse reflection, 1 specular reflection, 2 diffuse transmission, 3 specular transmission */
    F3 spectrum, direction;
};

__device__ __forceinline__ void create_cartesian(F3 up, F3 &right, F3 &forward) {        /* util.rs:34-40 */
    F3 temp_vec = norm3(cross3(up, f3(0.1f, 0.5f, 0.9f)));
    right = norm3(cross3(temp_vec, up));
    forward = norm3(cross3(up, right));
}

__device__ __forceinline__ BsdfSample lambertian_sample(F3 albedo, F3 normal, F3 r) {    /* bsdf.rs:71-92 */
    F3 nt, nb;
    create_cartesian(normal, nt, nb);
    float theta = rptm::acosr(rptm::sqrtr(r.x)), phi = 2.0f * RPT_PI_F * r.y;
    F3 s = f3(rptm::sinr(theta) * rptm::cosr(phi), rptm::cosr(theta), rptm::sinr(theta) * rptm::sinr(phi));
    BsdfSample o;
    o.direction = norm3(f3(s.x * nb.x + s.y * normal.x + s.z * nt.x, s.x * nb.y + s.y * normal.y + s.z * nt.y,
                           s.x * nb.z + s.y * normal.z + s.z * nt.z));
    float cos_theta = rptm::fmaxr(dot3(normal, o.direction), 0.0f);
    o.pdf = cos_theta / RPT_PI_F;
    o.spectrum = albedo / RPT_PI_F * cos_theta;
    o.lobe = 0u;
    return o;
}
__device__ __forceinline__ void lambertian_evaluate(F3 albedo, F3 normal, F3 sample_direction, F3 &spectrum, float &pdf) {
    float cos_theta = rptm::fmaxr(dot3(normal, sample_direction), 0.0f);                  /* bsdf.rs:59-69, 94-104 */
    pdf = cos_theta / RPT_PI_F;
    spectrum = albedo / RPT_PI_F * cos_theta;
}

__device__ __forceinline__ BsdfSample glass_sample(F3 albedo, float ior, float roughness, F3 view, F3 normal, F3 r) {   /* bsdf.rs:130-168 */
    const bool inside = dot3(normal, view) < 0.0f;
    const F3 nrm = inside ? -normal : normal;
    const float in_ior = inside ? ior : 1.0f, out_ior = inside ? 1.0f : ior;
    const float a_g = roughness * roughness;
    const float theta_m = rptm::atanr((a_g * rptm::sqrtr(r.x)) / rptm::sqrtr(1.0f - r.x));
    const float phi_m = 2.0f * RPT_PI_F * r.y;
    const F3 m = f3(rptm::sinr(theta_m) * rptm::cosr(phi_m), rptm::cosr(theta_m), rptm::sinr(theta_m) * rptm::sinr(phi_m));
    F3 nt, nb;
    create_cartesian(nrm, nt, nb);
    const F3 mn = norm3(f3(m.x * nb.x + m.y * nrm.x + m.z * nt.x, m.x * nb.y + m.y * nrm.y + m.z * nt.y,
                           m.x * nb.z + m.y * nrm.z + m.z * nt.z));
    float f0 = (in_ior - out_ior) / (in_ior + out_ior);
    f0 = f0 * f0;
    const float fresnel = f0 + (1.0f - f0) * rptm::powi5(1.0f - rptm::fmaxr(dot3(mn, view), 0.0f));
    BsdfSample o;
    o.pdf = 1.0f;
    if (r.z <= fresnel) {
        o.direction = norm3(2.0f * rptm::absr(dot3(view, mn)) * mn - view);
        o.lobe = 1u;
        o.spectrum = f3s(1.0f);
    } else {
        const float eta = in_ior / out_ior;
        const float c = dot3(view, mn);
        const float d = dot3(view, nrm);
        const float sg = d != d ? d : ((rptm::f2u(d) >> 31) ? -1.0f : 1.0f);               /* f32::signum */
        o.direction = norm3((eta * c - sg * rptm::sqrtr(rptm::fmaxr(1.0f + eta * (c * c - 1.0f), 0.0f))) * mn - eta * view);
        o.lobe = 3u;
        o.spectrum = albedo;
    }
    return o;
}

/* test hook: see oracle_bsdf for the item layout */
__global__ void k_debug_bsdf(int kind, size_t n, const float *in, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *p = in + 16 * i;
    F3 view = f3(p[0], p[1], p[2]), normal = f3(p[3], p[4], p[5]), r = f3(p[6], p[7], p[8]), albedo = f3(p[9], p[10], p[11]);
    BsdfSample o;
    o.pdf = 0.0f; o.lobe = 0u; o.spectrum = f3s(0.0f); o.direction = f3s(0.0f);
    if (kind == 0) o = lambertian_sample(albedo, normal, r);
    else if (kind == 1) o = glass_sample(albedo, p[12], p[13], view, normal, r);
    else if (kind == 2) lambertian_evaluate(albedo, normal, r, o.spectrum, o.pdf);
    else { o.lobe = rptm::f2u32_sat(r.x); o.spectrum = o.lobe == 1u ? f3s(1.0f) : albedo; o.pdf = 1.0f; }   /* bsdf.rs:115-128, 170-176 */
    float *q = out + 8 * i;
    q[0] = o.pdf; q[1] = __uint_as_float(o.lobe);
    q[2] = o.spectrum.x; q[3] = o.spectrum.y; q[4] = o.spectrum.z;
    q[5] = o.direction.x; q[6] = o.direction.y; q[7] = o.direction.z;
}

#endif /* RPT_K_BSDF_EXTRA_H */
