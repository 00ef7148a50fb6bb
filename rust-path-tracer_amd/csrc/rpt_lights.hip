/*
 * rpt_lights.hip — build_light_pick_table (reference src/light_pick.rs:24-122) on the GPU: the second half of "Startup time"
 * (benches/benchmark.rs:11-16) that SURVEY.md 8f N1 names next to the BVH build.  Bit-identical to the sequential builder
 * (csrc/host/light_table.cpp, oracle/bvh_oracle.cpp oracle_light_table): same entries, same order, same f32 values.
 *
 * What is parallel and what is not.  The reference computes, in this order,
 *   1  per emissive triangle: Heron area (5 sqrt) and power = emissive . (1,1,1) * area              independent per triangle
 *   2  total_power = the f32 sum of the powers IN INDEX ORDER                                        a rounding chain: order is the result
 *   3  probability_i = power_i / total_power                                                         independent
 *   4  average = (f32 sum of the probabilities in index order) / #emissive                           chain
 *   5  bins = the triangles with probability != 0, STABLY sorted ascending by probability           parallel: LSD radix sort is stable
 *   6  "robin hood": walk the bins upwards, top each up to `average` from the current most probable bin, which steps down when it is used up
 *                                                                                                    a two-cursor f32 recurrence
 *   7  the table: areas / pdfs gathered through index_a / index_b, ratio = p_a / (p_a + p_b)         independent
 * 1, 3, 5, 7 are kernels (5 = a stable LSD radix sort on order-preserving keys: lt_radix_sort_pairs below; rounds 4-5 used rocPRIM's, whose 156 kernel
 * instantiations were more than half of librpt_hip.so).  The three chains — 2, 4, 6 — run on the HOST between the
 * device passes: each of their steps waits for the rounded result of the one before, so what runs them is a question of latency per
 * dependent operation, not of width.  Measured on MI355X (profiles/r05_light_table.txt): one wave adding a million floats in index order —
 * values through the scalar cache, one v_add_f32 per element — takes 10.5 ms, 22 cycles per element on a GPU that idles around that one wave
 * (the first version, v_readlane per element behind vector loads: 9 ms); the host core does it in 1.2 ms, and the robin-hood walk over a
 * million bins in 0.6 ms.  What the chains need crosses PCIe in less time than that: the powers down (4 bytes per triangle), two scalars up,
 * the sorted bins down and their fill up (8 + 12 bytes per bin).  No part of the RESULT is computed differently: the same f32 operations in the
 * same order, wherever they run (the probabilities the host sums are the quotients the device computes: IEEE division on both sides).
 *
 * A NaN probability (a NaN vertex, or total_power = 0 / inf / NaN) has no place in the order-preserving keys — Rust's sort_by treats it as
 * equal to everything, which is not an order — and is refused (RPT_ESCENE), like a NaN coordinate by the BVH builder.
 */
#include <cstring>

#include <hip/hip_runtime.h>

#include <chrono>
#include <string>
#include <vector>

#include "rpt_ctx.h"

namespace {

constexpr int LT_BLOCK = 256;

/* the device memory of one call: ONE allocation per phase (hipMalloc / hipFree cost 0.1-0.3 ms each — seventeen of them were a fifth of a
 * 1 M-triangle build), carved up 256-byte aligned, released on every way out */
struct Arena {
    char *base = nullptr;
    size_t size = 0, used = 0;
    Arena() = default;
    Arena(const Arena &) = delete;
    Arena &operator=(const Arena &) = delete;
    ~Arena() { if (base) (void)hipFree(base); }
    static size_t pad(size_t bytes) { return (bytes + 255u) & ~(size_t)255u; }
    hipError_t reserve(size_t bytes) { size = bytes; used = 0; return hipMalloc(reinterpret_cast<void **>(&base), bytes ? bytes : 256); }
    template <typename T> T *take(size_t count) { T *p = reinterpret_cast<T *>(base + used); used += pad(count * sizeof(T)); return p; }
};

/* ---- stable LSD radix sort of (u32 key, u32 value) pairs: four passes over 8-bit digits, 1 024 pairs per workgroup ----------------------------------------
 * pass: k_rs_count (digits of a workgroup's tile) -> k_rs_scan (exclusive scan over [digit][workgroup]: where every tile's pairs of every digit go) -> k_rs_scatter
 * (tile in order, four rounds of 256: a pair's rank among the equal digits before it = lanes before it in its wave (ballots per digit bit), waves before it in the
 * round, rounds before it in the tile).  Stable by construction: equal probabilities keep their triangle-index order, as Rust's sort_by does (light_pick.rs:84-88).
 * A few hundred microseconds per million pairs; the table build is scene preparation. */
constexpr uint32_t RS_TILE = 1024u, RS_DIGITS = 256u;
__global__ __launch_bounds__(256) void k_rs_count(const uint32_t *keys, uint32_t n, uint32_t shift, uint32_t n_tiles, uint32_t *hist /* [digit][tile] */) {
    __shared__ uint32_t cnt[RS_DIGITS];
    cnt[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t base = blockIdx.x * RS_TILE;
    for (uint32_t r = 0; r < RS_TILE / 256u; ++r) {
        const uint32_t i = base + r * 256u + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_tiles + blockIdx.x] = cnt[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t *hist, uint32_t count) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (count + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < count ? lo + per : count;
    uint32_t s = 0u;
    for (uint32_t k = lo; k < hi; ++k) s += hist[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t run = 0u;
        for (int k = 0; k < 1024; ++k) { const uint32_t v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = hist[k]; hist[k] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t *keys, const uint32_t *vals, uint32_t n, uint32_t shift, uint32_t n_tiles, const uint32_t *offsets /* scanned hist */,
                                                    uint32_t *keys_out, uint32_t *vals_out) {
    __shared__ uint32_t next[RS_DIGITS];                 /* where the tile's next pair of every digit goes */
    __shared__ uint32_t wave_cnt[4][RS_DIGITS];          /* pairs of every digit in every wave of the current round */
    next[threadIdx.x] = offsets[threadIdx.x * n_tiles + blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, base = blockIdx.x * RS_TILE;
    for (uint32_t r = 0; r < RS_TILE / 256u; ++r) {
        for (uint32_t w = 0; w < 4u; ++w) wave_cnt[w][threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t i = base + r * 256u + threadIdx.x;
        const bool have = i < n;
        const uint32_t key = have ? keys[i] : 0u, val = have ? vals[i] : 0u, digit = (key >> shift) & 255u;
        /* the lanes of this wave that hold a pair of the same digit */
        unsigned long long same = __ballot(have);
        for (uint32_t b = 0; b < 8u; ++b) {
            const unsigned long long with_bit = __ballot(have && ((digit >> b) & 1u) != 0u);
            same &= ((digit >> b) & 1u) != 0u ? with_bit : ~with_bit;
        }
        const uint32_t before = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (have && before == 0u) wave_cnt[wave][digit] = (uint32_t)__popcll(same);
        __syncthreads();
        if (have) {
            uint32_t at = next[digit] + before;
            for (uint32_t w = 0; w < wave; ++w) at += wave_cnt[w][digit];
            keys_out[at] = key;
            vals_out[at] = val;
        }
        __syncthreads();
        next[threadIdx.x] += (wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x]) + (wave_cnt[2][threadIdx.x] + wave_cnt[3][threadIdx.x]);
        __syncthreads();
    }
}
size_t lt_sort_tmp_bytes(uint32_t n) { return (size_t)RS_DIGITS * ((n + RS_TILE - 1u) / RS_TILE) * sizeof(uint32_t); }
/* ascending by key, stable; the sorted pairs end in (keys, vals) — four passes, ping-pong through (keys_tmp, vals_tmp) */
hipError_t lt_radix_sort_pairs(uint32_t *hist, uint32_t *keys, uint32_t *keys_tmp, uint32_t *vals, uint32_t *vals_tmp, uint32_t n) {
    const uint32_t n_tiles = (n + RS_TILE - 1u) / RS_TILE;
    for (uint32_t pass = 0; pass < 4u; ++pass) {
        const uint32_t *ki = pass & 1u ? keys_tmp : keys, *vi = pass & 1u ? vals_tmp : vals;
        uint32_t *ko = pass & 1u ? keys : keys_tmp, *vo = pass & 1u ? vals : vals_tmp;
        k_rs_count<<<n_tiles, 256>>>(ki, n, pass * 8u, n_tiles, hist);
        k_rs_scan<<<1, 1024>>>(hist, RS_DIGITS * n_tiles);
        k_rs_scatter<<<n_tiles, 256>>>(ki, vi, n, pass * 8u, n_tiles, hist, ko, vo);
    }
    return hipGetLastError();
}

struct LtScalars {
    float total_power, average, prob_sum;
    uint32_t total_tris, n_bins, has_nan;
};

__device__ __forceinline__ float lt_len(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }      /* glam Vec3::length: dot(self).sqrt() */

/* step 1 (light_pick.rs:5-11, 34-51) */
__global__ __launch_bounds__(LT_BLOCK) void k_lt_power(const float4 *vertices, const uint4 *triangles, const float4 *emissive /* per material */, uint32_t nt,
                                                       float *area, float *power, LtScalars *sc) {
    const uint32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    bool mask = false;
    if (i < nt) {
        const uint4 t = triangles[i];
        const float4 e = emissive[t.w];
        mask = e.x != 0.0f || e.y != 0.0f || e.z != 0.0f;                     /* compute_emissive_mask, :13-21 */
        float a_out = 0.0f, p_out = 0.0f;
        if (mask) {
            const float4 a = vertices[t.x], b = vertices[t.y], c = vertices[t.z];
            const float la = lt_len(b.x - a.x, b.y - a.y, b.z - a.z), lb = lt_len(c.x - b.x, c.y - b.y, c.z - b.z), lc = lt_len(a.x - c.x, a.y - c.y, a.z - c.z);
            const float s = ((la + lb) + lc) / 2.0f;
            a_out = sqrtf(((s * (s - la)) * (s - lb)) * (s - lc));
            p_out = ((e.x * 1.0f + e.y * 1.0f) + e.z * 1.0f) * a_out;          /* emissive.xyz().dot(Vec3::ONE) * area */
        }
        area[i] = a_out;
        power[i] = p_out;
    }
    const unsigned long long m = __ballot(mask);
    if (m != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&sc->total_tris, (uint32_t)__popcll(m));
}

/* step 3 (:59-62) + the bin predicate of :74-83 */
__global__ __launch_bounds__(LT_BLOCK) void k_lt_prob(const float *power, uint32_t nt, float *prob, LtScalars *sc, uint32_t *block_counts) {
    const uint32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    bool bin = false, nan = false;
    if (i < nt) {
        const float p = power[i] / sc->total_power;
        prob[i] = p;
        nan = p != p;
        bin = p != 0.0f;                                                       /* (true for a NaN, as in the reference) */
    }
    __shared__ uint32_t cnt;
    if (threadIdx.x == 0u) cnt = 0u;
    __syncthreads();
    const unsigned long long m = __ballot(bin);
    if (m != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(&cnt, (uint32_t)__popcll(m));
    if (__ballot(nan) != 0ull && (threadIdx.x & 63u) == 0u) sc->has_nan = 1u;
    __syncthreads();
    if (threadIdx.x == 0u) block_counts[blockIdx.x] = cnt;
}

/* exclusive scan of the per-block bin counts (one workgroup; a few thousand values) */
__global__ __launch_bounds__(1024) void k_lt_scan(uint32_t *block_counts, uint32_t nb, LtScalars *sc) {
    __shared__ uint32_t part[1024];
    const uint32_t per = (nb + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    uint32_t s = 0u;
    for (uint32_t k = lo; k < hi; ++k) s += block_counts[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t run = 0u;
        for (int k = 0; k < 1024; ++k) { const uint32_t v = part[k]; part[k] = run; run += v; }
        sc->n_bins = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = block_counts[k]; block_counts[k] = run; run += v; }
}

/* the bins in index order: (order-preserving key of the probability, triangle index) */
__device__ __forceinline__ uint32_t lt_key(float p) { const uint32_t u = __float_as_uint(p); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lt_unkey(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ __launch_bounds__(LT_BLOCK) void k_lt_bins(const float *prob, uint32_t nt, const uint32_t *block_offsets, uint32_t *keys, uint32_t *vals) {
    const uint32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    const bool bin = i < nt && prob[i] != 0.0f;
    __shared__ uint32_t wave_cnt[LT_BLOCK / 64];
    const unsigned long long m = __ballot(bin);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) wave_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t base = block_offsets[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) base += wave_cnt[w];
    if (bin) {
        const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        keys[at] = lt_key(prob[i]);
        vals[at] = i;
    }
}

__global__ __launch_bounds__(LT_BLOCK) void k_lt_unkey(const uint32_t *keys, uint32_t n, float *pa) {
    const uint32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    if (i < n) pa[i] = lt_unkey(keys[i]);
}

/* step 7 (:106-119) */
__global__ __launch_bounds__(LT_BLOCK) void k_lt_table(const float *pa, const float *pb, const uint32_t *ia, const uint32_t *ib, const float *area, const float *prob,
                                                       uint32_t n, rpt_light_pick_entry *out) {
    const uint32_t i = blockIdx.x * LT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = ia[i], b = ib[i];
    rpt_light_pick_entry e;
    e.triangle_index_a = a;
    e.triangle_area_a = area[a];
    e.triangle_pick_pdf_a = prob[a];
    e.triangle_index_b = b;
    e.triangle_area_b = area[b];
    e.triangle_pick_pdf_b = prob[b];
    e.ratio = pa[i] / (pa[i] + pb[i]);
    out[i] = e;
}

}  // namespace

extern "C" int rpt_light_table_build_gpu(int device_id, const float *vertices_xyzw, size_t n_vertices, const rpt_triangle *triangles, size_t n_triangles,
                                         const rpt_material_data *materials, size_t n_materials, rpt_light_pick_entry *entries_out, size_t entries_capacity,
                                         size_t *n_entries_out, uint32_t *n_emissive_out, double *ms_out /* nullable, 4 doubles: total, device passes, host chains, transfers */) {
    std::string &err = rpt_create_error();
    if (!vertices_xyzw || !triangles || !materials || !entries_out || !n_entries_out || n_vertices == 0 || n_materials == 0 || entries_capacity == 0) {
        err = "rpt_light_table_build_gpu: null or empty argument";
        return RPT_EINVAL;
    }
    if (n_triangles >= (1ull << 31)) { err = "rpt_light_table_build_gpu: too many triangles"; return RPT_EINVAL; }
    for (size_t i = 0; i < n_triangles; ++i)
        if (triangles[i].v0 >= n_vertices || triangles[i].v1 >= n_vertices || triangles[i].v2 >= n_vertices || triangles[i].material >= n_materials) {
            err = "rpt_light_table_build_gpu: vertex or material index out of range";
            return RPT_ESCENE;
        }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { err = "no HIP device"; return RPT_ENODEV; }
    if (device_id < 0 || device_id >= n_dev) { err = "device id out of range"; return RPT_EINVAL; }
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    double ms_transfer = 0.0, ms_fill = 0.0;
    auto sentinel = [&](uint32_t emissive) {
        rpt_light_pick_entry s{};
        s.ratio = -1.0f;                                                       /* :52-58 */
        entries_out[0] = s;
        *n_entries_out = 1;
        if (n_emissive_out) *n_emissive_out = emissive;
        if (ms_out) { ms_out[0] = since(t_begin); ms_out[1] = ms_out[0] - ms_transfer; ms_out[2] = 0.0; ms_out[3] = ms_transfer; }
        return RPT_OK;
    };
    if (n_triangles == 0) return sentinel(0u);

#define LT_TRY(x)                                                                                       \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) { err = std::string("rpt_light_table_build_gpu: ") + hipGetErrorString(e_); return RPT_EHIP; } \
    } while (0)
    const uint32_t nt = (uint32_t)n_triangles, nb = (nt + LT_BLOCK - 1) / LT_BLOCK;
    Arena phase1, phase2;
    LT_TRY(hipSetDevice(device_id));
    std::vector<float4> emissive(n_materials);
    for (size_t m = 0; m < n_materials; ++m) emissive[m] = make_float4(materials[m].emissive[0], materials[m].emissive[1], materials[m].emissive[2], 0.0f);
    LT_TRY(phase1.reserve(Arena::pad(n_vertices * sizeof(float4)) + Arena::pad((size_t)nt * sizeof(uint4)) + Arena::pad(n_materials * sizeof(float4)) +
                          3 * Arena::pad((size_t)nt * sizeof(float)) + Arena::pad((size_t)nb * sizeof(uint32_t)) + Arena::pad(sizeof(LtScalars))));
    struct { float4 *p; } d_verts{phase1.take<float4>(n_vertices)}, d_emissive{nullptr};
    struct { uint4 *p; } d_tris{phase1.take<uint4>(nt)};
    d_emissive.p = phase1.take<float4>(n_materials);
    struct { float *p; } d_area{phase1.take<float>(nt)}, d_power{phase1.take<float>(nt)}, d_prob{phase1.take<float>(nt)}, d_pa{nullptr}, d_pb{nullptr};
    struct { uint32_t *p; } d_counts{phase1.take<uint32_t>(nb)}, d_keys{nullptr}, d_keys2{nullptr}, d_vals{nullptr}, d_vals2{nullptr}, d_ib{nullptr};
    struct { LtScalars *p; } d_sc{phase1.take<LtScalars>(1)};
    struct { rpt_light_pick_entry *p; } d_out{nullptr};
    struct { char *p; } d_tmp{nullptr};
    auto t0 = std::chrono::steady_clock::now();
    LT_TRY(hipMemcpy(d_verts.p, vertices_xyzw, n_vertices * sizeof(float4), hipMemcpyHostToDevice));
    LT_TRY(hipMemcpy(d_tris.p, triangles, (size_t)nt * sizeof(uint4), hipMemcpyHostToDevice));
    LT_TRY(hipMemcpy(d_emissive.p, emissive.data(), n_materials * sizeof(float4), hipMemcpyHostToDevice));
    LT_TRY(hipMemset(d_sc.p, 0, sizeof(LtScalars)));
    ms_transfer += since(t0);

    k_lt_power<<<nb, LT_BLOCK>>>(d_verts.p, d_tris.p, d_emissive.p, nt, d_area.p, d_power.p, d_sc.p);
    LtScalars sc;
    std::vector<float> power(nt);
    t0 = std::chrono::steady_clock::now();
    LT_TRY(hipMemcpy(&sc, d_sc.p, sizeof(sc), hipMemcpyDeviceToHost));
    if (sc.total_tris == 0u) return sentinel(0u);
    LT_TRY(hipMemcpy(power.data(), d_power.p, (size_t)nt * sizeof(float), hipMemcpyDeviceToHost));
    ms_transfer += since(t0);
    /* steps 2 and 4 (:39-51, :59-64): both sums in index order (the 0.0 of a non-emissive triangle changes nothing: the sums are never -0.0) */
    t0 = std::chrono::steady_clock::now();
    {
        float total_power = 0.0f;
        for (uint32_t i = 0; i < nt; ++i) total_power += power[i];
        float prob_sum = 0.0f;
        for (uint32_t i = 0; i < nt; ++i) prob_sum += power[i] / total_power;
        sc.total_power = total_power;
        sc.prob_sum = prob_sum;
        sc.average = prob_sum / (float)sc.total_tris;
    }
    ms_fill += since(t0);
    LT_TRY(hipMemcpy(d_sc.p, &sc, sizeof(sc), hipMemcpyHostToDevice));
    k_lt_prob<<<nb, LT_BLOCK>>>(d_power.p, nt, d_prob.p, d_sc.p, d_counts.p);
    k_lt_scan<<<1, 1024>>>(d_counts.p, nb, d_sc.p);
    LT_TRY(hipMemcpy(&sc, d_sc.p, sizeof(sc), hipMemcpyDeviceToHost));
    if (sc.has_nan) { err = "rpt_light_table_build_gpu: a pick probability is NaN (NaN vertex, or a total power of 0 / inf): such a scene must be built by the host builder"; return RPT_ESCENE; }
    const uint32_t n = sc.n_bins;
    if (n == 0u) return sentinel(sc.total_tris);       /* every emissive triangle is degenerate: the reference would index bins[usize::MAX] and panic */
    if ((size_t)n > entries_capacity) { err = "rpt_light_table_build_gpu: the table needs " + std::to_string(n) + " entries"; return RPT_EINVAL; }
    const size_t tmp_bytes = lt_sort_tmp_bytes(n);
    LT_TRY(phase2.reserve(7 * Arena::pad((size_t)n * sizeof(uint32_t)) + Arena::pad((size_t)n * sizeof(rpt_light_pick_entry)) + Arena::pad(tmp_bytes ? tmp_bytes : 1)));
    d_keys.p = phase2.take<uint32_t>(n); d_keys2.p = phase2.take<uint32_t>(n); d_vals.p = phase2.take<uint32_t>(n); d_vals2.p = phase2.take<uint32_t>(n);
    d_pa.p = phase2.take<float>(n); d_pb.p = phase2.take<float>(n); d_ib.p = phase2.take<uint32_t>(n);
    d_out.p = phase2.take<rpt_light_pick_entry>(n);
    d_tmp.p = phase2.take<char>(tmp_bytes ? tmp_bytes : 1);
    k_lt_bins<<<nb, LT_BLOCK>>>(d_prob.p, nt, d_counts.p, d_keys.p, d_vals.p);
    LT_TRY(lt_radix_sort_pairs(reinterpret_cast<uint32_t *>(d_tmp.p), d_keys.p, d_keys2.p, d_vals.p, d_vals2.p, n));   /* stable; sorted pairs in (d_keys, d_vals) */
    const uint32_t nbn = (n + LT_BLOCK - 1) / LT_BLOCK;
    k_lt_unkey<<<nbn, LT_BLOCK>>>(d_keys.p, n, d_pa.p);

    /* step 6, :89-104, on the host: (pa, index_a) of the sorted bins down, (pa', index_b, pb) up */
    std::vector<float> pa(n), pb(n, 0.0f);
    std::vector<uint32_t> ia(n), ib(n, 0u);
    t0 = std::chrono::steady_clock::now();
    LT_TRY(hipMemcpy(pa.data(), d_pa.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    LT_TRY(hipMemcpy(ia.data(), d_vals.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    ms_transfer += since(t0);
    t0 = std::chrono::steady_clock::now();
    {
        const float average = sc.average;
        size_t most_probable = (size_t)n - 1;
        for (size_t i = 0; i < n; ++i) {
            const float needed = average - pa[i];
            if (needed <= 0.0f) break;
            ib[i] = ia[most_probable];
            pb[i] = needed;
            pa[most_probable] -= needed;
            if (pa[most_probable] <= average) {
                if (most_probable == 0) break;                                 /* (the reference: usize underflow) */
                most_probable -= 1;
            }
        }
    }
    ms_fill += since(t0);
    t0 = std::chrono::steady_clock::now();
    LT_TRY(hipMemcpy(d_pa.p, pa.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    LT_TRY(hipMemcpy(d_pb.p, pb.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
    LT_TRY(hipMemcpy(d_ib.p, ib.data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
    ms_transfer += since(t0);
    k_lt_table<<<nbn, LT_BLOCK>>>(d_pa.p, d_pb.p, d_vals.p, d_ib.p, d_area.p, d_prob.p, n, d_out.p);
    LT_TRY(hipGetLastError());
    t0 = std::chrono::steady_clock::now();
    LT_TRY(hipMemcpy(entries_out, d_out.p, (size_t)n * sizeof(rpt_light_pick_entry), hipMemcpyDeviceToHost));
    ms_transfer += since(t0);           /* (includes the wait for k_lt_table) */
    *n_entries_out = n;
    if (n_emissive_out) *n_emissive_out = sc.total_tris;
    if (ms_out) { ms_out[0] = since(t_begin); ms_out[2] = ms_fill; ms_out[3] = ms_transfer; ms_out[1] = ms_out[0] - ms_fill - ms_transfer; }
#undef LT_TRY
    return RPT_OK;
}
