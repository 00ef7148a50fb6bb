/*
 * k_bvh_build.h — the reference's binned-SAH BVH build (src/bvh.rs:59-324) on the GPU, level by level, producing
 * the SAME node pool and the SAME triangle order as the sequential builder, bit for bit.
 *
 * What makes that possible:
 *   - A node's split depends only on the triangles in its range and on their ORDER (the order decides the sign of a
 *     zero bound and the next partition), never on when the node is processed.  So the explicit-stack, depth-first
 *     build of bvh.rs:257-324 can run breadth-first — one workgroup per node, all nodes of a level in one launch —
 *     and the nodes are renumbered afterwards to the order in which the reference would have split them
 *     (children get the next two indices at split time, left subtree first: bvh.rs:296-320).
 *   - f32::min / f32::max folds (bvh.rs:85-103, 9-33) are order-independent except for the sign of a zero result:
 *     `a < b ? a : b` keeps the LATER operand on a tie, and -0 == +0.  Every reduction here therefore runs on 64-bit
 *     keys  ord(value) << 32 | tie-break(sequence position) | sign-of-zero  so that the winner is the element the
 *     sequential fold would have kept.
 *   - The in-place two-pointer partition (bvh.rs:281-292) is a data-dependent walk, but its result has a closed
 *     form: with nl = #(centroid < split), left-side elements already in [first, first+nl) stay; the i-th "hole"
 *     (right-side element in that prefix, ascending) is filled by the i-th left-side element of the suffix taken in
 *     DESCENDING order; right-side elements land at last - r where r is their discovery rank: hole i is discovered
 *     after i earlier holes and after every suffix right-side element above the (i-1)-th suffix left-side element; a
 *     suffix right-side element q after min(m+1, H) holes (m = suffix left-side elements above q, H = #holes) and
 *     after the suffix right-side elements above it — except in the tail below the lowest suffix left-side element,
 *     where the element AT first+nl is met first.  (Checked against the sequential loop on 200 000 random inputs
 *     before it was written down here; tests/test_gpu_bvh_build.py compares whole builds.)
 *
 * One workgroup walks its node's range with a stride loop (k_bvb_level); nodes of 65 536+ triangles — the top of a
 * large tree, where a level is only as fast as its largest node — are split by a team of 64 workgroups (k_bvb_team,
 * end of this file).
 */
#ifndef RPT_K_BVH_BUILD_H
#define RPT_K_BVH_BUILD_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define BVB_THREADS 256
#define BVB_MAX_BINS 128
#define BVB_NONE 0xffffffffu

struct BvbNode {
    float mn[3];
    uint32_t first;
    float mx[3];
    uint32_t count;
    uint32_t left;        /* build-order id of the left child (right = left + 1), BVB_NONE for a leaf */
    uint32_t pad[3];
};

struct BvbArgs {
    const float4 *verts;
    const uint4 *tris;          /* original order */
    float4 *centroid;           /* per original triangle */
    uint32_t *order;            /* position -> original triangle */
    uint32_t *order_tmp, *tmp_a, *tmp_b;
    BvbNode *nodes;
    uint32_t *node_count;
    uint32_t n_tris, bins;
};

/* monotone float -> u32 with both zeros on the same code; the sign of a zero travels separately */
__device__ __forceinline__ uint32_t bvb_ord(float v, uint32_t &neg_zero) {
    uint32_t u = __float_as_uint(v);
    neg_zero = u == 0x80000000u ? 1u : 0u;
    if (neg_zero) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float bvb_unord(uint32_t o, uint32_t neg_zero) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    if (u == 0u && neg_zero) u = 0x80000000u;
    return __uint_as_float(u);
}
/* keys: a sequential `cur = (cur < x) ? cur : x` keeps the later operand on ties */
__device__ __forceinline__ unsigned long long bvb_min_key(float v, uint32_t seq) {
    uint32_t nz;
    uint32_t o = bvb_ord(v, nz);
    return ((unsigned long long)o << 32) | (unsigned long long)(((0x7fffffffu - seq) << 1) | nz);
}
__device__ __forceinline__ unsigned long long bvb_max_key(float v, uint32_t seq) {
    uint32_t nz;
    uint32_t o = bvb_ord(v, nz);
    return ((unsigned long long)o << 32) | (unsigned long long)((seq << 1) | nz);
}
__device__ __forceinline__ float bvb_key_value(unsigned long long k) { return bvb_unord((uint32_t)(k >> 32), (uint32_t)k & 1u); }

#define BVB_MIN_IDENT 0xffffffffffffffffull
#define BVB_MAX_IDENT 0ull

__global__ __launch_bounds__(BVB_THREADS) void k_bvb_init(BvbArgs a) {
    uint32_t i = blockIdx.x * BVB_THREADS + threadIdx.x;
    if (i >= a.n_tris) return;
    uint4 t = a.tris[i];
    float4 v0 = a.verts[t.x], v1 = a.verts[t.y], v2 = a.verts[t.z];
    /* (v0 + v1 + v2) / 3.0 (bvh.rs:66-69) */
    a.centroid[i] = make_float4(((v0.x + v1.x) + v2.x) / 3.0f, ((v0.y + v1.y) + v2.y) / 3.0f, ((v0.z + v1.z) + v2.z) / 3.0f, 0.0f);
    a.order[i] = i;
}

struct BvbBox {
    float mn[3], mx[3];
};
__device__ __forceinline__ float bvb_min(float a, float b) { return (a < b || b != b) ? a : b; }
__device__ __forceinline__ float bvb_max(float a, float b) { return (a > b || b != b) ? a : b; }
__device__ __forceinline__ float bvb_area(const BvbBox &b) {
    float ex = b.mx[0] - b.mn[0], ey = b.mx[1] - b.mn[1], ez = b.mx[2] - b.mn[2];
    return ex * ey + ey * ez + ez * ex;
}

/* exclusive prefix of a per-thread 0/1 flag over the workgroup (+ running base), in thread order; every thread calls */
template <int THREADS>
__device__ __forceinline__ uint32_t bvb_block_rank(bool flag, uint32_t *wave_tot, uint32_t &block_total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned long long m = __builtin_amdgcn_ballot_w64(flag);
    uint32_t within = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    __syncthreads();                       /* wave_tot free again */
    if (lane == 0u) wave_tot[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0u, total = 0u;
    for (uint32_t w = 0; w < (uint32_t)THREADS / 64u; ++w) {
        uint32_t n = wave_tot[w];
        before += w < wave ? n : 0u;
        total += n;
    }
    block_total = total;
    return before + within;
}

/* ---- the sweep over the bins of one axis (bvh.rs:214-253) as two wave scans ---------------------------------------------------------------------------
 * The sequential sweep folds the bins' boxes upwards (left boxes) and downwards (right boxes) with encapsulate = (f32::min, f32::max) per component, skipping empty
 * bins, and takes the first candidate of least cost.  `bvb_min(a, b)` keeps the LATER operand on a tie (only the sign of a zero can differ), which is an associative
 * rule — the fold of a run is "the last of its least elements" however it is bracketed — and an empty bin behaves like the identity (+inf, -inf): so the left boxes are
 * an inclusive scan in ascending bin order and the right boxes one in descending order, two bins per lane, earlier operand first.  Round 6: the three-thread serial sweep
 * (127 dependent steps per axis) was most of the 60 us a workgroup spends on a small node (profiles/r06_bvh_levels_scatter.txt). */
__device__ __forceinline__ BvbBox bvb_box_identity() {
    BvbBox b;
    for (int j = 0; j < 3; ++j) { b.mn[j] = __builtin_inff(); b.mx[j] = -__builtin_inff(); }
    return b;
}
__device__ __forceinline__ BvbBox bvb_box_join(const BvbBox &earlier, const BvbBox &later) {
    BvbBox b;
    for (int j = 0; j < 3; ++j) { b.mn[j] = bvb_min(earlier.mn[j], later.mn[j]); b.mx[j] = bvb_max(earlier.mx[j], later.mx[j]); }
    return b;
}
__device__ __forceinline__ BvbBox bvb_box_shfl(const BvbBox &b, int from) {
    BvbBox o;
    for (int j = 0; j < 3; ++j) { o.mn[j] = __shfl(b.mn[j], from, 64); o.mx[j] = __shfl(b.mx[j], from, 64); }
    return o;
}
/* one wave, one axis: best candidate (least cost, lowest index on ties; none: +inf, 0) over candidates 0 .. S - 2 */
__device__ __forceinline__ void bvb_wave_sweep(const unsigned long long (*key)[6], const uint32_t *cnt, uint32_t S, float &best_cost, uint32_t &best_i) {
    const int lane = (int)(threadIdx.x & 63u);
    BvbBox e[2];
    uint32_t c[2];
    for (int k = 0; k < 2; ++k) {
        const uint32_t b = 2u * (uint32_t)lane + (uint32_t)k;
        e[k] = bvb_box_identity();
        c[k] = 0u;
        if (b < S) {
            c[k] = cnt[b];
            if (c[k] != 0u && bvb_key_value(key[b][0]) != __builtin_inff())                /* encapsulate_node skips an empty box */
                for (int j = 0; j < 3; ++j) { e[k].mn[j] = bvb_key_value(key[b][j]); e[k].mx[j] = bvb_key_value(key[b][3 + j]); }
        }
    }
    /* left: bins ascending.  up = join of the bins of all lanes below */
    BvbBox up = bvb_box_join(e[0], e[1]);
    uint32_t up_c = c[0] + c[1];
    for (int d = 1; d < 64; d <<= 1) {
        const BvbBox o = bvb_box_shfl(up, lane - d < 0 ? lane : lane - d);
        const uint32_t oc = (uint32_t)__shfl((int)up_c, lane - d < 0 ? lane : lane - d, 64);
        if (lane >= d) { up = bvb_box_join(o, up); up_c += oc; }
    }
    BvbBox below = bvb_box_shfl(up, lane == 0 ? 0 : lane - 1);
    uint32_t below_c = (uint32_t)__shfl((int)up_c, lane == 0 ? 0 : lane - 1, 64);
    if (lane == 0) { below = bvb_box_identity(); below_c = 0u; }
    const BvbBox left0 = bvb_box_join(below, e[0]), left1 = bvb_box_join(left0, e[1]);
    const uint32_t lc0 = below_c + c[0], lc1 = lc0 + c[1];
    /* right: bins descending.  dn = join of the bins of all lanes above, highest first */
    BvbBox dn = bvb_box_join(e[1], e[0]);
    uint32_t dn_c = c[0] + c[1];
    for (int d = 1; d < 64; d <<= 1) {
        const BvbBox o = bvb_box_shfl(dn, lane + d > 63 ? lane : lane + d);
        const uint32_t oc = (uint32_t)__shfl((int)dn_c, lane + d > 63 ? lane : lane + d, 64);
        if (lane + d <= 63) { dn = bvb_box_join(o, dn); dn_c += oc; }
    }
    BvbBox above = bvb_box_shfl(dn, lane == 63 ? 63 : lane + 1);                          /* bins 2 lane + 2 and up */
    uint32_t above_c = (uint32_t)__shfl((int)dn_c, lane == 63 ? 63 : lane + 1, 64);
    if (lane == 63) { above = bvb_box_identity(); above_c = 0u; }
    const BvbBox right_of_0 = bvb_box_join(above, e[1]);                                   /* bins 2 lane + 1 and up: right side of candidate 2 lane */
    const uint32_t rc_of_0 = above_c + c[1];
    /* candidates 2 lane (right side: bins 2 lane + 1 ...) and 2 lane + 1 (right side: bins 2 lane + 2 ...), in index order */
    float cost = __builtin_inff();
    uint32_t at = 0u;
    {
        const uint32_t i0 = 2u * (uint32_t)lane, i1 = i0 + 1u;
        if (i0 + 1u < S) {
            const float v = (float)lc0 * bvb_area(left0) + (float)rc_of_0 * bvb_area(right_of_0);
            if (v < cost) { cost = v; at = i0; }
        }
        if (i1 + 1u < S) {
            const float v = (float)lc1 * bvb_area(left1) + (float)above_c * bvb_area(above);
            if (v < cost) { cost = v; at = i1; }
        }
    }
    /* the first least cost of the wave (`cost < best` in index order: a NaN or +inf candidate is never taken) */
    for (int d = 32; d >= 1; d >>= 1) {
        const float oc = __shfl_xor(cost, d, 64);
        const uint32_t oa = (uint32_t)__shfl_xor((int)at, d, 64);
        if (oc < cost || (oc == cost && oc != __builtin_inff() && oa < at)) { cost = oc; at = oa; }
    }
    best_cost = cost;
    best_i = cost == __builtin_inff() ? 0u : at;
}

/* THREADS = 1024 for the few huge nodes at the top of the tree (a workgroup walks its node's whole range), 256 below, 64 — one wave per node — for levels whose
 * largest node has at most 64 triangles (the deep levels of a tree with small leaves: 0.4 M nodes of two to eight triangles each on the scattered stand-in) */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_bvb_level(BvbArgs a, uint32_t level_begin) {
    constexpr uint32_t AXES = THREADS >= 192 ? 3u : 1u;            /* axes binned at once */
    __shared__ unsigned long long s_key[AXES][BVB_MAX_BINS][6];   /* [axis][bin]: min x,y,z  max x,y,z */
    __shared__ uint32_t s_cnt[AXES][BVB_MAX_BINS];
    __shared__ unsigned long long s_red[6];
    __shared__ uint32_t s_cb[6];                                  /* centroid bounds: ord(min) x3, ord(max) x3 */
    __shared__ float s_best_cost[3];
    __shared__ uint32_t s_best_i[3];
    __shared__ uint32_t s_wave_tot[THREADS / 64];
    __shared__ uint32_t s_misc[4];
    __shared__ float s_split;
    __shared__ int s_axis;

    const uint32_t tid = threadIdx.x;
    const uint32_t node_id = level_begin + blockIdx.x;
    BvbNode &node = a.nodes[node_id];
    if (node.pad[0] != 0u) return;                 /* already split at this level by a team (k_bvb_team) */
    const uint32_t first = node.first, count = node.count, S = a.bins;
    const uint32_t last = first + count - 1u;

    /* ---- update_node_aabb (bvh.rs:85-103), sequential tie-breaking reproduced by the keys */
    if (tid < 6u) s_red[tid] = tid < 3u ? BVB_MIN_IDENT : BVB_MAX_IDENT;
    if (tid < 3u) { s_cb[tid] = 0xffffffffu; s_cb[3u + tid] = 0u; }
    __syncthreads();
    {
        unsigned long long kmin[3] = {BVB_MIN_IDENT, BVB_MIN_IDENT, BVB_MIN_IDENT}, kmax[3] = {BVB_MAX_IDENT, BVB_MAX_IDENT, BVB_MAX_IDENT};
        uint32_t cmin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, cmax[3] = {0u, 0u, 0u};
        for (uint32_t i = tid; i < count; i += THREADS) {
            const uint32_t tri = a.order[first + i];
            const uint4 t = a.tris[tri];
            const float4 v[3] = {a.verts[t.x], a.verts[t.y], a.verts[t.z]};
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t seq = i * 3u + k;
                const float c[3] = {v[k].x, v[k].y, v[k].z};
                for (int j = 0; j < 3; ++j) {
                    unsigned long long lo = bvb_min_key(c[j], seq), hi = bvb_max_key(c[j], seq);
                    kmin[j] = lo < kmin[j] ? lo : kmin[j];
                    kmax[j] = hi > kmax[j] ? hi : kmax[j];
                }
            }
            const float4 ce = a.centroid[tri];
            const float cc[3] = {ce.x, ce.y, ce.z};
            for (int j = 0; j < 3; ++j) {
                uint32_t nz;
                uint32_t o = bvb_ord(cc[j], nz);
                cmin[j] = o < cmin[j] ? o : cmin[j];
                cmax[j] = o > cmax[j] ? o : cmax[j];
            }
        }
        for (int j = 0; j < 3; ++j) {
            atomicMin(&s_red[j], kmin[j]);
            atomicMax(&s_red[3 + j], kmax[j]);
            atomicMin(&s_cb[j], cmin[j]);
            atomicMax(&s_cb[3 + j], cmax[j]);
        }
    }
    __syncthreads();
    if (tid < 3u) {
        node.mn[tid] = bvb_key_value(s_red[tid]);
        node.mx[tid] = bvb_key_value(s_red[3u + tid]);
    }
    float bmin[3], bmax[3];
    for (int j = 0; j < 3; ++j) { bmin[j] = bvb_unord(s_cb[j], 0u); bmax[j] = bvb_unord(s_cb[3 + j], 0u); }
    const float nmn[3] = {bvb_key_value(s_red[0]), bvb_key_value(s_red[1]), bvb_key_value(s_red[2])};
    const float nmx[3] = {bvb_key_value(s_red[3]), bvb_key_value(s_red[4]), bvb_key_value(s_red[5])};

    /* ---- find_best_split_segmented (bvh.rs:178-255): all three axes binned in one pass by a wide workgroup; the one-wave build bins and sweeps one axis at a
     * time through ONE set of bins (6 KB instead of 18 KB of LDS: more than twice the nodes in flight per CU, and a small node is all latency) */
    float scale[3];
    bool axis_on[3];
    for (int j = 0; j < 3; ++j) {
        axis_on[j] = !(bmin[j] == bmax[j]);
        scale[j] = (float)S / (bmax[j] - bmin[j]);
    }
    for (uint32_t pass = 0; pass < 3u / AXES; ++pass) {
        for (uint32_t k = tid; k < AXES * BVB_MAX_BINS; k += THREADS) {
            const uint32_t slot = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
            for (int j = 0; j < 6; ++j) s_key[slot][b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
            s_cnt[slot][b] = 0u;
        }
        __syncthreads();
        for (uint32_t i = tid; i < count; i += THREADS) {
            const uint32_t tri = a.order[first + i];
            const uint4 t = a.tris[tri];
            const float4 v[3] = {a.verts[t.x], a.verts[t.y], a.verts[t.z]};
            const float4 ce = a.centroid[tri];
            const float cc[3] = {ce.x, ce.y, ce.z};
            for (uint32_t slot = 0; slot < AXES; ++slot) {
                const uint32_t ax = AXES == 3u ? slot : pass;
                if (!axis_on[ax]) continue;
                const float x = (cc[ax] - bmin[ax]) * scale[ax];
                uint32_t si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;      /* `as usize` then min(S-1) */
                if (si > S - 1u) si = S - 1u;
                for (uint32_t k = 0; k < 3u; ++k) {
                    const uint32_t seq = i * 3u + k;
                    const float c[3] = {v[k].x, v[k].y, v[k].z};
                    for (int j = 0; j < 3; ++j) {
                        atomicMin(&s_key[slot][si][j], bvb_min_key(c[j], seq));
                        atomicMax(&s_key[slot][si][3 + j], bvb_max_key(c[j], seq));
                    }
                }
                atomicAdd(&s_cnt[slot][si], 1u);
            }
        }
        __syncthreads();
        if (AXES == 3u) {
            if (tid < 192u) {                                  /* waves 0, 1, 2: one axis each (bvb_wave_sweep) */
                const uint32_t ax = tid >> 6;
                float best_cost = __builtin_inff();
                uint32_t best_i = 0u;
                if (axis_on[ax]) bvb_wave_sweep(s_key[ax], s_cnt[ax], S, best_cost, best_i);
                if ((tid & 63u) == 0u) { s_best_cost[ax] = best_cost; s_best_i[ax] = best_i; }
            }
        } else {
            float best_cost = __builtin_inff();
            uint32_t best_i = 0u;
            if (axis_on[pass]) bvb_wave_sweep(s_key[0], s_cnt[0], S, best_cost, best_i);
            if (tid == 0u) { s_best_cost[pass] = best_cost; s_best_i[pass] = best_i; }
            __syncthreads();                                   /* the bins are about to be cleared for the next axis */
        }
    }
    __syncthreads();
    if (tid == 0u) {
        int axis = 0;
        float split = 0.0f, cost = __builtin_inff();
        for (int ax = 0; ax < 3; ++ax) {
            if (s_best_cost[ax] < cost) {
                cost = s_best_cost[ax];
                axis = ax;
                const float scale2 = (bmax[ax] - bmin[ax]) / (float)S;
                split = bmin[ax] + scale2 * (float)(s_best_i[ax] + 1u);
            }
        }
        BvbBox nb;
        for (int j = 0; j < 3; ++j) { nb.mn[j] = nmn[j]; nb.mx[j] = nmx[j]; }
        const float parent_cost = bvb_area(nb) * (float)count;
        s_axis = parent_cost <= cost ? -1 : axis;                 /* bvh.rs:272-277 */
        s_split = split;
    }
    __syncthreads();
    const int axis = s_axis;
    if (axis < 0) {
        if (tid == 0u) node.left = BVB_NONE;
        return;
    }
    const float split = s_split;

    /* ---- the partition (bvh.rs:281-292) in closed form (file header) */
    auto is_left = [&](uint32_t pos) {
        const float4 ce = a.centroid[a.order[pos]];
        const float c = axis == 0 ? ce.x : (axis == 1 ? ce.y : ce.z);
        return c < split;
    };
    uint32_t nl;
    {
        uint32_t mine = 0u;
        for (uint32_t i = tid; i < count; i += THREADS) mine += is_left(first + i) ? 1u : 0u;
        if (tid == 0u) s_misc[0] = 0u;
        __syncthreads();
        atomicAdd(&s_misc[0], mine);
        __syncthreads();
        nl = s_misc[0];
    }
    const uint32_t back_n = count - nl;                 /* suffix positions first+nl .. last */
    /* pass B1: suffix, descending — rb_at_L[m] = right-side elements above the m-th left-side element */
    uint32_t H = 0u;
    {
        uint32_t run_l = 0u, run_r = 0u;
        for (uint32_t base = 0; base < back_n; base += THREADS) {
            const uint32_t i = base + tid;
            const bool valid = i < back_n;
            const uint32_t q = last - i;
            const bool L = valid && is_left(q);
            const bool R = valid && !L;
            uint32_t tot_l, tot_r;
            const uint32_t m = run_l + bvb_block_rank<THREADS>(L, s_wave_tot, tot_l);
            const uint32_t rb = run_r + bvb_block_rank<THREADS>(R, s_wave_tot, tot_r);
            if (L) a.tmp_b[first + m] = rb;
            run_l += tot_l;
            run_r += tot_r;
        }
        H = run_l;
    }
    __syncthreads();
    __threadfence_block();
    const uint32_t base_rb = H >= 1u ? a.tmp_b[first + H - 1u] : 0u;
    /* pass F: prefix, ascending */
    {
        uint32_t run = 0u;
        for (uint32_t base = 0; base < nl; base += THREADS) {
            const uint32_t i = base + tid;
            const bool valid = i < nl;
            const uint32_t p = first + i;
            const bool L = valid && is_left(p);
            const bool R = valid && !L;
            uint32_t tot;
            const uint32_t hole = run + bvb_block_rank<THREADS>(R, s_wave_tot, tot);
            if (L) a.order_tmp[p] = a.order[p];
            if (R) {
                const uint32_t rank = hole + (hole >= 1u ? a.tmp_b[first + hole - 1u] : 0u);
                a.order_tmp[last - rank] = a.order[p];
                a.tmp_a[first + hole] = p;
            }
            run += tot;
        }
    }
    __syncthreads();
    __threadfence_block();
    /* pass B2: suffix again, now every destination is known */
    {
        uint32_t run_l = 0u, run_r = 0u;
        for (uint32_t base = 0; base < back_n; base += THREADS) {
            const uint32_t i = base + tid;
            const bool valid = i < back_n;
            const uint32_t q = last - i;
            const bool L = valid && is_left(q);
            const bool R = valid && !L;
            uint32_t tot_l, tot_r;
            const uint32_t m = run_l + bvb_block_rank<THREADS>(L, s_wave_tot, tot_l);
            const uint32_t rb = run_r + bvb_block_rank<THREADS>(R, s_wave_tot, tot_r);
            if (L) a.order_tmp[a.tmp_a[first + m]] = a.order[q];
            if (R) {
                uint32_t rank;
                if (m == H) rank = (q == first + nl) ? H + base_rb : H + rb + 1u;
                else rank = (m + 1u) + rb;
                a.order_tmp[last - rank] = a.order[q];
            }
            run_l += tot_l;
            run_r += tot_r;
        }
    }
    __syncthreads();
    __threadfence_block();
    for (uint32_t i = tid; i < count; i += THREADS) a.order[first + i] = a.order_tmp[first + i];

    if (tid == 0u) {
        if (nl == 0u || nl == count) {
            node.left = BVB_NONE;                               /* bvh.rs:294-296: stays a leaf, triangles already permuted */
        } else {
            const uint32_t id = atomicAdd(a.node_count, 2u);
            atomicMax(a.node_count + 1, nl > count - nl ? nl : count - nl);      /* the largest node of the next level: the host picks that level's workgroup size by it */
            node.left = id;
            a.nodes[id].first = first;          a.nodes[id].count = nl;              a.nodes[id].left = BVB_NONE;   a.nodes[id].pad[0] = 0u;
            a.nodes[id + 1u].first = first + nl; a.nodes[id + 1u].count = count - nl; a.nodes[id + 1u].left = BVB_NONE; a.nodes[id + 1u].pad[0] = 0u;
        }
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * The few HUGE nodes at the top of a large tree: a TEAM of BVB_TEAM workgroups per node instead of one.
 * Same arithmetic, same keys, same closed-form partition — the passes are cut into contiguous chunks of the node's
 * range, partial results meet in global memory (64-bit atomic min/max on the same keys, per-chunk left counts), and
 * the workgroups of a team meet at a counter barrier between passes (all of them are resident: the host launches at
 * most BVB_MAX_TEAMS teams).  With L(p) = number of left-side elements at positions < p (chunk prefix + local rank)
 * every quantity of the closed form is local:  hole rank of p = (p - first) - L(p);  for a suffix position q:
 * m(q) = nl - L(q+1) left-side elements above it, rb(q) = (last - q) - m(q) right-side elements above it.
 * Measured on the 1 M-triangle stand-in: the levels that hold a 0.5-1 M-triangle node took 7-15 ms each with one
 * workgroup per node. */
#define BVB_TEAM 64
#define BVB_TEAM_THREADS 256
#define BVB_MAX_TEAMS 16          /* x 64 workgroups of ~26 KB LDS: all resident at once (the barrier needs that) */
#define BVB_TEAM_MIN_COUNT 65536u

struct BvbTeamScratch {
    unsigned long long red[6];
    unsigned long long key[3][BVB_MAX_BINS][6];
    uint32_t cb[6];
    uint32_t cnt[3][BVB_MAX_BINS];
    uint32_t chunk_l[BVB_TEAM], chunk_lf[BVB_TEAM];
    uint32_t barrier;
    int axis;
    float split;
    uint32_t node_id;
};

__global__ __launch_bounds__(BVB_TEAM_THREADS) void k_bvb_team_init(BvbTeamScratch *scratch, const uint32_t *team_nodes) {
    BvbTeamScratch &t = scratch[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (tid < 6u) { t.red[tid] = tid < 3u ? BVB_MIN_IDENT : BVB_MAX_IDENT; t.cb[tid] = tid < 3u ? 0xffffffffu : 0u; }
    for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += BVB_TEAM_THREADS) {
        const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
        for (int j = 0; j < 6; ++j) t.key[ax][b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
        t.cnt[ax][b] = 0u;
    }
    if (tid < BVB_TEAM) { t.chunk_l[tid] = 0u; t.chunk_lf[tid] = 0u; }
    if (tid == 0u) { t.barrier = 0u; t.axis = -1; t.split = 0.0f; t.node_id = team_nodes[blockIdx.x]; }
}

/* all BVB_TEAM workgroups of a team arrive; `phase` counts the barriers passed so far */
__device__ __forceinline__ void bvb_team_sync(uint32_t *counter, uint32_t &phase) {
    __syncthreads();
    phase += 1u;
    if (threadIdx.x == 0u) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase * BVB_TEAM) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(BVB_TEAM_THREADS) void k_bvb_team(BvbArgs a, BvbTeamScratch *scratch) {
    constexpr int THREADS = BVB_TEAM_THREADS;
    __shared__ unsigned long long s_key[3][BVB_MAX_BINS][6];
    __shared__ uint32_t s_cnt[3][BVB_MAX_BINS];
    __shared__ float s_la[3][BVB_MAX_BINS], s_ra[3][BVB_MAX_BINS];
    __shared__ uint32_t s_lc[3][BVB_MAX_BINS], s_rc[3][BVB_MAX_BINS];
    __shared__ unsigned long long s_red[6];
    __shared__ uint32_t s_cb[6];
    __shared__ float s_best_cost[3];
    __shared__ uint32_t s_best_i[3];
    __shared__ uint32_t s_wave_tot[THREADS / 64];
    __shared__ uint32_t s_misc[4];

    const uint32_t tid = threadIdx.x;
    const uint32_t team = blockIdx.x / BVB_TEAM, member = blockIdx.x % BVB_TEAM;
    BvbTeamScratch &T = scratch[team];
    BvbNode &node = a.nodes[T.node_id];
    const uint32_t first = node.first, count = node.count, S = a.bins;
    const uint32_t last = first + count - 1u;
    const uint32_t chunk = (count + BVB_TEAM - 1u) / BVB_TEAM;
    const uint32_t c_begin = first + member * chunk < first + count ? first + member * chunk : first + count;
    const uint32_t c_end = c_begin + chunk < first + count ? c_begin + chunk : first + count;      /* positions [c_begin, c_end) */
    uint32_t phase = 0u;

    /* ---- pass 1: update_node_aabb keys + centroid bounds over the chunk */
    if (tid < 6u) { s_red[tid] = tid < 3u ? BVB_MIN_IDENT : BVB_MAX_IDENT; s_cb[tid] = tid < 3u ? 0xffffffffu : 0u; }
    __syncthreads();
    {
        unsigned long long kmin[3] = {BVB_MIN_IDENT, BVB_MIN_IDENT, BVB_MIN_IDENT}, kmax[3] = {BVB_MAX_IDENT, BVB_MAX_IDENT, BVB_MAX_IDENT};
        uint32_t cmin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, cmax[3] = {0u, 0u, 0u};
        for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS) {
            const uint32_t i = pos - first;
            const uint32_t tri = a.order[pos];
            const uint4 t = a.tris[tri];
            const float4 v[3] = {a.verts[t.x], a.verts[t.y], a.verts[t.z]};
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t seq = i * 3u + k;
                const float c[3] = {v[k].x, v[k].y, v[k].z};
                for (int j = 0; j < 3; ++j) {
                    unsigned long long lo = bvb_min_key(c[j], seq), hi = bvb_max_key(c[j], seq);
                    kmin[j] = lo < kmin[j] ? lo : kmin[j];
                    kmax[j] = hi > kmax[j] ? hi : kmax[j];
                }
            }
            const float4 ce = a.centroid[tri];
            const float cc[3] = {ce.x, ce.y, ce.z};
            for (int j = 0; j < 3; ++j) {
                uint32_t nz;
                uint32_t o = bvb_ord(cc[j], nz);
                cmin[j] = o < cmin[j] ? o : cmin[j];
                cmax[j] = o > cmax[j] ? o : cmax[j];
            }
        }
        for (int j = 0; j < 3; ++j) {
            atomicMin(&s_red[j], kmin[j]);
            atomicMax(&s_red[3 + j], kmax[j]);
            atomicMin(&s_cb[j], cmin[j]);
            atomicMax(&s_cb[3 + j], cmax[j]);
        }
    }
    __syncthreads();
    if (tid < 3u) {
        atomicMin(&T.red[tid], s_red[tid]);
        atomicMax(&T.red[3u + tid], s_red[3u + tid]);
        atomicMin(&T.cb[tid], s_cb[tid]);
        atomicMax(&T.cb[3u + tid], s_cb[3u + tid]);
    }
    bvb_team_sync(&T.barrier, phase);
    float bmin[3], bmax[3], nmn[3], nmx[3];
    for (int j = 0; j < 3; ++j) {
        bmin[j] = bvb_unord(__hip_atomic_load(&T.cb[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0u);
        bmax[j] = bvb_unord(__hip_atomic_load(&T.cb[3 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), 0u);
        nmn[j] = bvb_key_value(__hip_atomic_load(&T.red[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        nmx[j] = bvb_key_value(__hip_atomic_load(&T.red[3 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    if (member == 0u && tid < 3u) { node.mn[tid] = nmn[tid]; node.mx[tid] = nmx[tid]; }

    /* ---- pass 2: bins of the chunk in LDS, merged into the team's bins */
    for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += THREADS) {
        const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
        for (int j = 0; j < 6; ++j) s_key[ax][b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
        s_cnt[ax][b] = 0u;
    }
    __syncthreads();
    float scale[3];
    bool axis_on[3];
    for (int j = 0; j < 3; ++j) {
        axis_on[j] = !(bmin[j] == bmax[j]);
        scale[j] = (float)S / (bmax[j] - bmin[j]);
    }
    for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS) {
        const uint32_t i = pos - first;
        const uint32_t tri = a.order[pos];
        const uint4 t = a.tris[tri];
        const float4 v[3] = {a.verts[t.x], a.verts[t.y], a.verts[t.z]};
        const float4 ce = a.centroid[tri];
        const float cc[3] = {ce.x, ce.y, ce.z};
        for (int ax = 0; ax < 3; ++ax) {
            if (!axis_on[ax]) continue;
            const float x = (cc[ax] - bmin[ax]) * scale[ax];
            uint32_t si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;
            if (si > S - 1u) si = S - 1u;
            for (uint32_t k = 0; k < 3u; ++k) {
                const uint32_t seq = i * 3u + k;
                const float c[3] = {v[k].x, v[k].y, v[k].z};
                for (int j = 0; j < 3; ++j) {
                    atomicMin(&s_key[ax][si][j], bvb_min_key(c[j], seq));
                    atomicMax(&s_key[ax][si][3 + j], bvb_max_key(c[j], seq));
                }
            }
            atomicAdd(&s_cnt[ax][si], 1u);
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += THREADS) {
        const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
        const uint32_t n = s_cnt[ax][b];
        if (n != 0u) {
            atomicAdd(&T.cnt[ax][b], n);
            for (int j = 0; j < 3; ++j) {
                atomicMin(&T.key[ax][b][j], s_key[ax][b][j]);
                atomicMax(&T.key[ax][b][3 + j], s_key[ax][b][3 + j]);
            }
        }
    }
    bvb_team_sync(&T.barrier, phase);

    /* ---- pass 3: member 0 evaluates the splits exactly as the single-workgroup kernel does */
    if (member == 0u) {
        for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += THREADS) {
            const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
            for (int j = 0; j < 6; ++j) s_key[ax][b][j] = __hip_atomic_load(&T.key[ax][b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_cnt[ax][b] = __hip_atomic_load(&T.cnt[ax][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 3u) {
            const uint32_t ax = tid;
            float best_cost = __builtin_inff();
            uint32_t best_i = 0u;
            if (axis_on[ax]) {
                BvbBox lb, rb;
                for (int j = 0; j < 3; ++j) { lb.mn[j] = rb.mn[j] = __builtin_inff(); lb.mx[j] = rb.mx[j] = -__builtin_inff(); }
                uint32_t lsum = 0u, rsum = 0u;
                for (uint32_t i = 0; i + 1u < S; ++i) {
                    lsum += s_cnt[ax][i];
                    s_lc[ax][i] = lsum;
                    if (s_cnt[ax][i] != 0u && bvb_key_value(s_key[ax][i][0]) != __builtin_inff()) {
                        for (int j = 0; j < 3; ++j) {
                            lb.mn[j] = bvb_min(lb.mn[j], bvb_key_value(s_key[ax][i][j]));
                            lb.mx[j] = bvb_max(lb.mx[j], bvb_key_value(s_key[ax][i][3 + j]));
                        }
                    }
                    s_la[ax][i] = bvb_area(lb);
                    const uint32_t r = S - 1u - i;
                    rsum += s_cnt[ax][r];
                    s_rc[ax][S - 2u - i] = rsum;
                    if (s_cnt[ax][r] != 0u && bvb_key_value(s_key[ax][r][0]) != __builtin_inff()) {
                        for (int j = 0; j < 3; ++j) {
                            rb.mn[j] = bvb_min(rb.mn[j], bvb_key_value(s_key[ax][r][j]));
                            rb.mx[j] = bvb_max(rb.mx[j], bvb_key_value(s_key[ax][r][3 + j]));
                        }
                    }
                    s_ra[ax][S - 2u - i] = bvb_area(rb);
                }
                for (uint32_t i = 0; i + 1u < S; ++i) {
                    const float cost = (float)s_lc[ax][i] * s_la[ax][i] + (float)s_rc[ax][i] * s_ra[ax][i];
                    if (cost < best_cost) { best_cost = cost; best_i = i; }
                }
            }
            s_best_cost[ax] = best_cost;
            s_best_i[ax] = best_i;
        }
        __syncthreads();
        if (tid == 0u) {
            int axis = 0;
            float split = 0.0f, cost = __builtin_inff();
            for (int ax = 0; ax < 3; ++ax) {
                if (s_best_cost[ax] < cost) {
                    cost = s_best_cost[ax];
                    axis = ax;
                    const float scale2 = (bmax[ax] - bmin[ax]) / (float)S;
                    split = bmin[ax] + scale2 * (float)(s_best_i[ax] + 1u);
                }
            }
            BvbBox nb;
            for (int j = 0; j < 3; ++j) { nb.mn[j] = nmn[j]; nb.mx[j] = nmx[j]; }
            const float parent_cost = bvb_area(nb) * (float)count;
            T.axis = parent_cost <= cost ? -1 : axis;
            T.split = split;
        }
    }
    bvb_team_sync(&T.barrier, phase);
    const int axis = T.axis;
    if (axis < 0) {
        if (member == 0u && tid == 0u) { node.left = BVB_NONE; node.pad[0] = 1u; }
        return;                                                   /* team-uniform */
    }
    const float split = T.split;
    auto is_left = [&](uint32_t pos) {
        const float4 ce = a.centroid[a.order[pos]];
        const float c = axis == 0 ? ce.x : (axis == 1 ? ce.y : ce.z);
        return c < split;
    };

    /* ---- pass 4: left counts per chunk -> nl and this chunk's prefix */
    {
        uint32_t mine = 0u;
        for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS) mine += is_left(pos) ? 1u : 0u;
        if (tid == 0u) s_misc[0] = 0u;
        __syncthreads();
        atomicAdd(&s_misc[0], mine);
        __syncthreads();
        if (tid == 0u) T.chunk_l[member] = s_misc[0];
    }
    bvb_team_sync(&T.barrier, phase);
    uint32_t nl = 0u, pref_l = 0u;
    for (uint32_t b = 0; b < BVB_TEAM; ++b) {
        const uint32_t n = __hip_atomic_load(&T.chunk_l[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pref_l += b < member ? n : 0u;
        nl += n;
    }
    const uint32_t split_pos = first + nl;                       /* prefix = [first, split_pos), suffix = [split_pos, last] */
    /* ---- pass 5: left-side elements inside the prefix -> H (holes) */
    {
        uint32_t mine = 0u;
        for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS) mine += (pos < split_pos && is_left(pos)) ? 1u : 0u;
        __syncthreads();
        if (tid == 0u) s_misc[0] = 0u;
        __syncthreads();
        atomicAdd(&s_misc[0], mine);
        __syncthreads();
        if (tid == 0u) T.chunk_lf[member] = s_misc[0];
    }
    bvb_team_sync(&T.barrier, phase);
    uint32_t l_in_prefix = 0u;
    for (uint32_t b = 0; b < BVB_TEAM; ++b) l_in_prefix += __hip_atomic_load(&T.chunk_lf[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t H = nl - l_in_prefix;

    /* ---- pass 6 (B1): rb_at_L for the suffix left-side elements of this chunk */
    {
        uint32_t run = pref_l;                                    /* left-side elements before the current tile */
        for (uint32_t base = c_begin; base < c_end; base += THREADS) {
            const uint32_t q = base + tid;
            const bool valid = q < c_end;
            const bool L = valid && is_left(q);
            uint32_t tot;
            const uint32_t l_before = run + bvb_block_rank<THREADS>(L, s_wave_tot, tot);
            if (L && q >= split_pos) {
                const uint32_t m = nl - (l_before + 1u);
                a.tmp_b[first + m] = (last - q) - m;
            }
            run += tot;
        }
    }
    bvb_team_sync(&T.barrier, phase);
    const uint32_t base_rb = H >= 1u ? __hip_atomic_load(&a.tmp_b[first + H - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    /* ---- pass 7 (F): the prefix */
    {
        uint32_t run = pref_l;
        for (uint32_t base = c_begin; base < c_end; base += THREADS) {
            const uint32_t p = base + tid;
            const bool valid = p < c_end;
            const bool L = valid && is_left(p);
            uint32_t tot;
            const uint32_t l_before = run + bvb_block_rank<THREADS>(L, s_wave_tot, tot);
            if (valid && p < split_pos) {
                if (L) {
                    a.order_tmp[p] = a.order[p];
                } else {
                    const uint32_t hole = (p - first) - l_before;
                    const uint32_t rank = hole + (hole >= 1u ? __hip_atomic_load(&a.tmp_b[first + hole - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
                    a.order_tmp[last - rank] = a.order[p];
                    a.tmp_a[first + hole] = p;
                }
            }
            run += tot;
        }
    }
    bvb_team_sync(&T.barrier, phase);
    /* ---- pass 8 (B2): the suffix */
    {
        uint32_t run = pref_l;
        for (uint32_t base = c_begin; base < c_end; base += THREADS) {
            const uint32_t q = base + tid;
            const bool valid = q < c_end;
            const bool L = valid && is_left(q);
            uint32_t tot;
            const uint32_t l_before = run + bvb_block_rank<THREADS>(L, s_wave_tot, tot);
            if (valid && q >= split_pos) {
                if (L) {
                    const uint32_t m = nl - (l_before + 1u);
                    a.order_tmp[__hip_atomic_load(&a.tmp_a[first + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = a.order[q];
                } else {
                    const uint32_t m = nl - l_before;
                    const uint32_t rb = (last - q) - m;
                    uint32_t rank;
                    if (m == H) rank = (q == split_pos) ? H + base_rb : H + rb + 1u;
                    else rank = (m + 1u) + rb;
                    a.order_tmp[last - rank] = a.order[q];
                }
            }
            run += tot;
        }
    }
    bvb_team_sync(&T.barrier, phase);
    for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS)
        a.order[pos] = __hip_atomic_load(&a.order_tmp[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (member == 0u && tid == 0u) {
        node.pad[0] = 1u;
        if (nl == 0u || nl == count) {
            node.left = BVB_NONE;
        } else {
            const uint32_t id = atomicAdd(a.node_count, 2u);
            node.left = id;
            a.nodes[id].first = first;          a.nodes[id].count = nl;              a.nodes[id].left = BVB_NONE;   a.nodes[id].pad[0] = 0u;
            a.nodes[id + 1u].first = first + nl; a.nodes[id + 1u].count = count - nl; a.nodes[id + 1u].left = BVB_NONE; a.nodes[id + 1u].pad[0] = 0u;
        }
    }
}

/* ---- renumbering to the reference's node order, on the device -----------------------------------------------------------
 * The build above numbers nodes level by level; the reference (src/bvh.rs:296-320) gives the two children of a node the next two free indices at the
 * moment the node is popped from its stack, left subtree first: children(b) = 1 + 2 r(b), 2 + 2 r(b) with r(b) = the number of INNER nodes popped
 * before b = b's rank among the inner nodes in left-first pre-order.  With I(b) = the inner nodes of b's subtree (bottom-up, one launch per level):
 * r(left) = r(b) + 1, r(right) = r(b) + 1 + I(left) (top-down, one launch per level), and every node is written straight to its place in the
 * reference's 32-byte layout.  (Round 4 read the build-order pool back and renumbered 2 M nodes on the host: 128 ms + 23 ms of read-back, more
 * than the build itself took; profiles/r05_startup_sections.txt.) */
__global__ __launch_bounds__(BVB_THREADS) void k_bvb_inner_count(const BvbNode *nodes, uint32_t *inner, uint32_t begin, uint32_t end) {
    const uint32_t b = begin + blockIdx.x * BVB_THREADS + threadIdx.x;
    if (b >= end) return;
    const uint32_t l = nodes[b].left;
    inner[b] = l == BVB_NONE ? 0u : 1u + inner[l] + inner[l + 1u];
}
__global__ __launch_bounds__(BVB_THREADS) void k_bvb_place(const BvbNode *nodes, const uint32_t *inner, uint32_t *rank, uint32_t *oidx, rpt_bvh_node *out,
                                                           uint32_t begin, uint32_t end) {
    const uint32_t b = begin + blockIdx.x * BVB_THREADS + threadIdx.x;
    if (b >= end) return;
    const BvbNode n = nodes[b];
    rpt_bvh_node o;
    o.aabb_min[0] = n.mn[0]; o.aabb_min[1] = n.mn[1]; o.aabb_min[2] = n.mn[2];
    o.aabb_max[0] = n.mx[0]; o.aabb_max[1] = n.mx[1]; o.aabb_max[2] = n.mx[2];
    if (n.left == BVB_NONE) {
        o.triangle_count = n.count;
        o.left_or_first = n.first;
    } else {
        const uint32_t r = rank[b], c = 1u + 2u * r;
        o.triangle_count = 0u;
        o.left_or_first = c;
        oidx[n.left] = c;          oidx[n.left + 1u] = c + 1u;
        rank[n.left] = r + 1u;     rank[n.left + 1u] = r + 1u + inner[n.left];
    }
    out[oidx[b]] = o;
}

#endif /* RPT_K_BVH_BUILD_H */
