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
 * Who splits a node, by its size (a level is one launch of each kernel that has work on it, then k_bvb_children):
 *   16 384+ triangles   k_bvb_team: a team of 2 ... 256 workgroups, one per 8 192 triangles — the top of a large tree, where a level is only as
 *                       fast as its largest node;
 *   above 256           k_bvb_level<1024 | 256>: one workgroup walks the node's range with a stride loop;
 *   65 ... 256          k_bvb_level<64>: one wave, one axis at a time through one set of bins;
 *   9 ... 64            k_bvb_small: one wave, the node's triangles fetched once and kept in registers;
 *   up to 8             k_bvb_tiny: eight nodes per wave, no bins.
 * The children of a level are numbered by k_bvb_children (one atomic per 1 024 nodes: the two hot words — next free id, largest child — cost 11 ns per
 * wave-atomic device-wide, which was most of a deep level's time).  Round 6, 1 M triangles: 40 -> 5.8 ms (64-triangle leaves, input in spatial order)
 * and 38 -> 6.0 ms (2 M nodes); profiles/r06_bvh_levels_*.txt, r06_bvh_input_order.txt (step by step).
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

/* What the build reads of a triangle — its box, which of its three vertices the sequential fold of the bounds would have kept per component (the tie-break of the
 * keys: the LAST vertex that attains the minimum / maximum, -0 == +0), its centroid — gathered once by k_bvb_init into one 64-byte line: every pass over a node is
 * position -> triangle -> this line instead of position -> triangle -> vertex indices -> three vertices (+ the centroid): five random 16-byte reads over 48 MB became
 * one, three dependent round trips two (a big node's bounds and bin passes are nothing else). */
struct BvbRec {
    float4 mn;                  /* least x, y, z over the vertices (as the winner holds it: the sign of a zero is the winner's); w: 2 bits per key = the winner's vertex, keys 0-5 */
    float4 mx;
    float4 ce;                  /* (v0 + v1 + v2) / 3.0 (bvh.rs:66-69) */
    float4 pad;
};

struct BvbArgs {
    const float4 *verts;
    const uint4 *tris;          /* original order */
    struct BvbRec *recs;        /* per original triangle: everything the build reads of it, in one 64-byte line (k_bvb_init) */
    uint32_t *order;            /* position -> original triangle */
    uint32_t *order_tmp, *tmp_a, *tmp_b;
    uint32_t *lpre;             /* per position of a node being partitioned: (left-side elements before it << 1) | it is one — written by the partition's counting pass;
                                   every later quantity of the closed form follows from it without another scan (teams: counted inside the workgroup's chunk) */
    BvbNode *nodes;
    uint32_t *node_count;       /* [0] nodes so far, [1] the largest child the level just built has made, [2] bits: 1 a kernel met a node it was not built for, 2 a vertex index out of range, 4 a NaN coordinate */
    uint32_t n_tris, bins, n_verts;
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
    a.order[i] = i;
    if (t.x >= a.n_verts || t.y >= a.n_verts || t.z >= a.n_verts) {      /* said to the host (node_count[2]); the build runs on over a point at the origin and is thrown away */
        atomicOr(a.node_count + 2, 2u);
        BvbRec z;
        z.mn = z.mx = z.ce = z.pad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        a.recs[i] = z;
        return;
    }
    const float4 v[3] = {a.verts[t.x], a.verts[t.y], a.verts[t.z]};
    {   /* the ordered keys have no place for a NaN (which f32::min / max skip): refused, see rpt_bvh_build_gpu */
        bool nan = false;
        for (int k = 0; k < 3; ++k) nan = nan || v[k].x != v[k].x || v[k].y != v[k].y || v[k].z != v[k].z;
        if (nan) atomicOr(a.node_count + 2, 4u);
    }
    BvbRec r;
    /* (v0 + v1 + v2) / 3.0 (bvh.rs:66-69) */
    r.ce = make_float4(((v[0].x + v[1].x) + v[2].x) / 3.0f, ((v[0].y + v[1].y) + v[2].y) / 3.0f, ((v[0].z + v[1].z) + v[2].z) / 3.0f, 0.0f);
    /* the fold of the three vertices under the keys' order: value first, then the later vertex (bvb_min_key / bvb_max_key with seq = 3 i + k) */
    float lo[3], hi[3];
    uint32_t who = 0u;
    for (int j = 0; j < 3; ++j) {
        const float c[3] = {j == 0 ? v[0].x : (j == 1 ? v[0].y : v[0].z), j == 0 ? v[1].x : (j == 1 ? v[1].y : v[1].z), j == 0 ? v[2].x : (j == 1 ? v[2].y : v[2].z)};
        uint32_t kl = 0u, kh = 0u;
        for (uint32_t k = 1; k < 3u; ++k) {
            if (bvb_min_key(c[k], k) < bvb_min_key(c[kl], kl)) kl = k;
            if (bvb_max_key(c[k], k) > bvb_max_key(c[kh], kh)) kh = k;
        }
        lo[j] = c[kl]; hi[j] = c[kh];
        who |= (kl << (2 * j)) | (kh << (2 * (3 + j)));
    }
    r.mn = make_float4(lo[0], lo[1], lo[2], __uint_as_float(who));
    r.mx = make_float4(hi[0], hi[1], hi[2], 0.0f);
    r.pad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    a.recs[i] = r;
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

/* exclusive prefix, in thread order, of a small per-thread count over the workgroup (+ the workgroup's total); every thread calls */
template <int THREADS>
__device__ __forceinline__ uint32_t bvb_block_scan(uint32_t v, uint32_t *wave_tot, uint32_t &block_total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= (uint32_t)d) inc += o;
    }
    __syncthreads();                       /* wave_tot free again */
    if (lane == 63u) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = 0u, total = 0u;
    for (uint32_t w = 0; w < (uint32_t)THREADS / 64u; ++w) {
        const uint32_t n = wave_tot[w];
        before += w < wave ? n : 0u;
        total += n;
    }
    block_total = total;
    return before + inc - v;
}

/* The partition's counting pass over positions [begin, end): each thread takes BVB_RUN consecutive positions per trip (their gathers — position -> triangle ->
 * centroid — in flight together), one workgroup scan per trip; leaves (left-side elements before the position, counted from `begin`) << 1 | is-left in a.lpre
 * and returns the number of left-side elements (bvh.rs:283: centroid[axis] < split).  Everything the closed-form partition needs afterwards is a function of
 * that word (the first form ranked the elements again in each of the three passes, two workgroup barriers per trip; one scan is simpler and 3 % faster —
 * a big node's time is in its bounds and bin passes, 0.10 ms each of the 0.24 ms a 16 k-triangle node takes under 1 024 threads: random gathers, position ->
 * triangle -> vertices, over 48 MB). */
#define BVB_RUN 4
template <int THREADS>
__device__ __forceinline__ uint32_t bvb_count_left(const BvbArgs &a, uint32_t begin, uint32_t end, int axis, float split, uint32_t *wave_tot) {
    uint32_t run = 0u;
    for (uint32_t base = begin; base < end; base += (uint32_t)THREADS * BVB_RUN) {
        uint32_t tri[BVB_RUN];
        bool have[BVB_RUN], left[BVB_RUN];
        const uint32_t p0 = base + threadIdx.x * BVB_RUN;
        _Pragma("unroll") for (int e = 0; e < BVB_RUN; ++e) { have[e] = p0 + (uint32_t)e < end; tri[e] = a.order[have[e] ? p0 + (uint32_t)e : begin]; }
        uint32_t mine = 0u;
        _Pragma("unroll") for (int e = 0; e < BVB_RUN; ++e) {
            const float4 ce = a.recs[tri[e]].ce;
            const float c = axis == 0 ? ce.x : (axis == 1 ? ce.y : ce.z);
            left[e] = have[e] && c < split;
            mine += left[e] ? 1u : 0u;
        }
        uint32_t total;
        uint32_t before = run + bvb_block_scan<THREADS>(mine, wave_tot, total);
        _Pragma("unroll") for (int e = 0; e < BVB_RUN; ++e)
            if (have[e]) { a.lpre[p0 + (uint32_t)e] = (before << 1) | (left[e] ? 1u : 0u); before += left[e] ? 1u : 0u; }
        run += total;
    }
    return run;
}

/* ---- a triangle into the bins --------------------------------------------------------------------------------------------------------------------------------
 * bvh.rs:214-229 grows the bin's box by the triangle's three vertices: nine coordinates, eighteen min / max folds.  The fold over a triangle's own three vertices
 * is taken once, by k_bvb_init (BvbRec; bvb_rec_keys: six keys per triangle, the same for every axis): 7 LDS atomics per triangle and axis instead of 19.  What
 * they cost is decided by how many lanes of a wave meet on one address — an LDS atomic is served lane by lane —, hence bvb_scatter_index below, and the wave's
 * own fold (bvb_wave_fold) before anything that every thread of a workgroup would add to the same word (the node's bounds). */
__device__ __forceinline__ void bvb_rec_keys(const float4 &mn, const float4 &mx, uint32_t i, bool valid, unsigned long long k6[6]) {
    for (int j = 0; j < 3; ++j) { k6[j] = BVB_MIN_IDENT; k6[3 + j] = BVB_MAX_IDENT; }
    if (!valid) return;
    const uint32_t who = __float_as_uint(mn.w);
    const float lo[3] = {mn.x, mn.y, mn.z}, hi[3] = {mx.x, mx.y, mx.z};
    for (int j = 0; j < 3; ++j) {
        k6[j] = bvb_min_key(lo[j], i * 3u + ((who >> (2 * j)) & 3u));
        k6[3 + j] = bvb_max_key(hi[j], i * 3u + ((who >> (2 * (3 + j))) & 3u));
    }
}
__device__ __forceinline__ unsigned long long bvb_shfl_xor_u64(unsigned long long v, int d) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d, 64);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ void bvb_wave_fold(const unsigned long long k6[6], unsigned long long out[6]) {
    for (int j = 0; j < 6; ++j) out[j] = k6[j];
    for (int d = 32; d >= 1; d >>= 1)
        for (int j = 0; j < 3; ++j) {
            const unsigned long long a = bvb_shfl_xor_u64(out[j], d), b = bvb_shfl_xor_u64(out[3 + j], d);
            out[j] = a < out[j] ? a : out[j];
            out[3 + j] = b > out[3 + j] ? b : out[3 + j];
        }
}
/* the element a thread takes in the bin pass: NOT its neighbour's neighbour.  In spatially ordered input (a scene file's meshes, an earlier build's leaf order)
 * the 64 triangles at consecutive positions fall into one or two bins of every axis, and an LDS atomic on one address is served lane by lane: the bin pass took
 * twice as long as on shuffled input.  An affine bijection of [0, 2^k) (odd multiplier), cycle-walked into [0, n), deals the node's positions to the lanes
 * like a shuffle; the keys carry the element's own position, so who folds an element into its bin is immaterial. */
__device__ __forceinline__ uint32_t bvb_scatter_index(uint32_t i, uint32_t n, uint32_t mask /* 2^k - 1 >= n - 1 */) {
    uint32_t j = i;
    do { j = (j * 0x9E3779B1u + 0x7F4A7C15u) & mask; } while (j >= n);
    return j;
}
__device__ __forceinline__ uint32_t bvb_mask_for(uint32_t n) { return n <= 1u ? 0u : (0xffffffffu >> __builtin_clz(n - 1u)); }

/* a triangle (six keys) into bin si of one axis */
__device__ __forceinline__ void bvb_bin_add(unsigned long long (*key)[6], uint32_t *cnt, bool valid, uint32_t si, const unsigned long long k6[6]) {
    if (!valid) return;
    for (int j = 0; j < 3; ++j) { atomicMin(&key[si][j], k6[j]); atomicMax(&key[si][3 + j], k6[3 + j]); }
    atomicAdd(&cnt[si], 1u);
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

/* THREADS = 1024 for levels whose largest node has 16 384+ triangles or that have fewer than 64 nodes (a workgroup walks its node's whole range), 256 below, 64 — one
 * wave per node — for levels whose largest node has at most 256 triangles (nodes of up to 64 go to k_bvb_small, of up to 8 to k_bvb_tiny) */
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_bvb_level(BvbArgs a, uint32_t level_begin, uint32_t skip_upto) {
    constexpr uint32_t AXES = THREADS >= 192 ? 3u : 1u;            /* axes binned at once */
    __shared__ unsigned long long s_key[AXES][BVB_MAX_BINS][6];   /* [axis][bin]: min x,y,z  max x,y,z */
    __shared__ uint32_t s_cnt[AXES][BVB_MAX_BINS];
    __shared__ unsigned long long s_red[6];
    __shared__ uint32_t s_cb[6];                                  /* centroid bounds: ord(min) x3, ord(max) x3 */
    __shared__ float s_best_cost[3];
    __shared__ uint32_t s_best_i[3];
    __shared__ uint32_t s_wave_tot[THREADS / 64];
    __shared__ float s_split;
    __shared__ int s_axis;

    const uint32_t tid = threadIdx.x;
    const uint32_t node_id = level_begin + blockIdx.x;
    BvbNode &node = a.nodes[node_id];
    if (node.pad[0] != 0u) return;                 /* already split at this level by a team (k_bvb_team) */
    const uint32_t first = node.first, count = node.count, S = a.bins;
    const uint32_t last = first + count - 1u;
    if (count <= skip_upto) return;                /* a node k_bvb_tiny has taken (0: none) */

    /* ---- update_node_aabb (bvh.rs:85-103), sequential tie-breaking reproduced by the keys */
    if (tid < 6u) s_red[tid] = tid < 3u ? BVB_MIN_IDENT : BVB_MAX_IDENT;
    if (tid < 3u) { s_cb[tid] = 0xffffffffu; s_cb[3u + tid] = 0u; }
    __syncthreads();
    {
        unsigned long long kmin[3] = {BVB_MIN_IDENT, BVB_MIN_IDENT, BVB_MIN_IDENT}, kmax[3] = {BVB_MAX_IDENT, BVB_MAX_IDENT, BVB_MAX_IDENT};
        uint32_t cmin[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, cmax[3] = {0u, 0u, 0u};
        for (uint32_t i = tid; i < count; i += THREADS) {
            const uint32_t tri = a.order[first + i];
            const float4 r_mn = a.recs[tri].mn, r_mx = a.recs[tri].mx, ce = a.recs[tri].ce;
            unsigned long long k6[6];
            bvb_rec_keys(r_mn, r_mx, i, true, k6);
            for (int j = 0; j < 3; ++j) {
                kmin[j] = k6[j] < kmin[j] ? k6[j] : kmin[j];
                kmax[j] = k6[3 + j] > kmax[j] ? k6[3 + j] : kmax[j];
            }
            const float cc[3] = {ce.x, ce.y, ce.z};
            for (int j = 0; j < 3; ++j) {
                uint32_t nz;
                uint32_t o = bvb_ord(cc[j], nz);
                cmin[j] = o < cmin[j] ? o : cmin[j];
                cmax[j] = o > cmax[j] ? o : cmax[j];
            }
        }
        /* the wave's fold by shuffles, then ONE lane per wave into the workgroup's words: an LDS atomic on one address is served lane by lane — 1 024 threads
         * x 12 of them were 0.10 of the 0.24 ms a 16 k-triangle node took */
        for (int d = 32; d >= 1; d >>= 1)
            for (int j = 0; j < 3; ++j) {
                const unsigned long long x = bvb_shfl_xor_u64(kmin[j], d), y = bvb_shfl_xor_u64(kmax[j], d);
                kmin[j] = x < kmin[j] ? x : kmin[j];
                kmax[j] = y > kmax[j] ? y : kmax[j];
                const uint32_t lo = (uint32_t)__shfl_xor((int)cmin[j], d, 64), hi = (uint32_t)__shfl_xor((int)cmax[j], d, 64);
                cmin[j] = lo < cmin[j] ? lo : cmin[j];
                cmax[j] = hi > cmax[j] ? hi : cmax[j];
            }
        if ((tid & 63u) == 0u)
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
    const uint32_t deal_mask = bvb_mask_for(count);
    for (uint32_t pass = 0; pass < 3u / AXES; ++pass) {
        for (uint32_t k = tid; k < AXES * BVB_MAX_BINS; k += THREADS) {
            const uint32_t slot = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
            for (int j = 0; j < 6; ++j) s_key[slot][b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
            s_cnt[slot][b] = 0u;
        }
        __syncthreads();
        for (uint32_t base = 0; base < count; base += THREADS) {
            const bool have = base + tid < count;
            const uint32_t i = have ? bvb_scatter_index(base + tid, count, deal_mask) : 0u;
            float4 r_mn = make_float4(0, 0, 0, 0), r_mx = make_float4(0, 0, 0, 0);
            float cc[3] = {0.0f, 0.0f, 0.0f};
            if (have) {
                const uint32_t tri = a.order[first + i];
                r_mn = a.recs[tri].mn; r_mx = a.recs[tri].mx;
                const float4 ce = a.recs[tri].ce;
                cc[0] = ce.x; cc[1] = ce.y; cc[2] = ce.z;
            }
            unsigned long long k6[6];
            bvb_rec_keys(r_mn, r_mx, i, have, k6);
            for (uint32_t slot = 0; slot < AXES; ++slot) {
                const uint32_t ax = AXES == 3u ? slot : pass;
                if (!axis_on[ax]) continue;
                const float x = (cc[ax] - bmin[ax]) * scale[ax];
                uint32_t si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;      /* `as usize` then min(S-1) */
                if (si > S - 1u) si = S - 1u;
                bvb_bin_add(s_key[slot], s_cnt[slot], have, si, k6);
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

    /* ---- the partition (bvh.rs:281-292) in closed form (file header), every quantity from L(p) = left-side elements before p (bvb_count_left):
     * hole rank of a prefix position p = (p - first) - L(p);  for a suffix position q: m(q) = nl - L(q + 1) left-side elements above it, rb(q) = (last - q) - m(q)
     * right-side elements above it */
    const uint32_t nl = bvb_count_left<THREADS>(a, first, first + count, axis, split, s_wave_tot);
    __syncthreads();
    __threadfence_block();
    const uint32_t split_pos = first + nl;                         /* prefix = [first, split_pos), suffix = [split_pos, last] */
    const uint32_t H = nl - (nl < count ? (a.lpre[split_pos] >> 1) : nl);      /* holes: right-side elements in the prefix */
    /* pass B1: rb_at_L[m] = right-side elements above the m-th left-side element of the suffix (counted from the top) */
    for (uint32_t q = split_pos + tid; q <= last; q += THREADS) {
        const uint32_t w = a.lpre[q];
        if (w & 1u) {
            const uint32_t m = nl - ((w >> 1) + 1u);
            a.tmp_b[first + m] = (last - q) - m;
        }
    }
    __syncthreads();
    __threadfence_block();
    const uint32_t base_rb = H >= 1u ? a.tmp_b[first + H - 1u] : 0u;
    /* pass F: the prefix */
    for (uint32_t p = first + tid; p < split_pos; p += THREADS) {
        const uint32_t w = a.lpre[p];
        if (w & 1u) {
            a.order_tmp[p] = a.order[p];
        } else {
            const uint32_t hole = (p - first) - (w >> 1);
            const uint32_t rank = hole + (hole >= 1u ? a.tmp_b[first + hole - 1u] : 0u);
            a.order_tmp[last - rank] = a.order[p];
            a.tmp_a[first + hole] = p;
        }
    }
    __syncthreads();
    __threadfence_block();
    /* pass B2: the suffix, now every destination is known */
    for (uint32_t q = split_pos + tid; q <= last; q += THREADS) {
        const uint32_t w = a.lpre[q];
        if (w & 1u) {
            const uint32_t m = nl - ((w >> 1) + 1u);
            a.order_tmp[a.tmp_a[first + m]] = a.order[q];
        } else {
            const uint32_t m = nl - (w >> 1);
            const uint32_t rb = (last - q) - m;
            uint32_t rank;
            if (m == H) rank = (q == split_pos) ? H + base_rb : H + rb + 1u;
            else rank = (m + 1u) + rb;
            a.order_tmp[last - rank] = a.order[q];
        }
    }
    __syncthreads();
    __threadfence_block();
    for (uint32_t i = tid; i < count; i += THREADS) a.order[first + i] = a.order_tmp[first + i];

    /* bvh.rs:294-296: with an empty side the node stays a leaf, its triangles already permuted; otherwise k_bvb_children makes the two children */
    if (tid == 0u) node.pad[1] = (nl == 0u || nl == count) ? 0u : nl;
}

/* ---- the children of a level's nodes (bvh.rs:296-320) -----------------------------------------------------------------------------------------------------
 * The split kernels leave `nl` (triangles on the left side) in pad[1] of a node that gets children; this pass numbers them: a workgroup ranks its 1 024 nodes and
 * takes its ids with ONE atomic.  (An atomic on one address costs 11.3 ns device-wide however many lanes of the wave take part — tools/atomic_probe.hip — so the
 * 0.2-0.4 M nodes of a deep level, each fetching its own pair of ids and raising the level's largest-child word, spent 2-5 ms per level on those two words.) */
__global__ __launch_bounds__(1024) void k_bvb_children(BvbArgs a, uint32_t level_begin, uint32_t level_end) {
    __shared__ uint32_t s_wave_tot[16], s_wave_big[16], s_base;
    const uint32_t tid = threadIdx.x;
    const uint32_t node_id = level_begin + blockIdx.x * 1024u + tid;
    uint32_t nl = 0u, first = 0u, count = 0u;
    if (node_id < level_end) {
        nl = a.nodes[node_id].pad[1];
        first = a.nodes[node_id].first; count = a.nodes[node_id].count;
    }
    uint32_t total;
    const uint32_t rank = bvb_block_rank<1024>(nl != 0u, s_wave_tot, total);
    if (total == 0u) return;
    uint32_t big = nl != 0u ? (nl > count - nl ? nl : count - nl) : 0u;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)big, d, 64); big = o > big ? o : big; }
    if ((tid & 63u) == 0u) s_wave_big[tid >> 6] = big;
    __syncthreads();
    if (tid == 0u) {
        s_base = atomicAdd(a.node_count, 2u * total);
        for (int w = 1; w < 16; ++w) big = s_wave_big[w] > big ? s_wave_big[w] : big;
        /* the largest node of the next level: the host picks that level's kernels by it */
        if (big > __hip_atomic_load(a.node_count + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.node_count + 1, big);
    }
    __syncthreads();
    if (nl != 0u) {
        const uint32_t id = s_base + 2u * rank;
        a.nodes[node_id].left = id;
        a.nodes[node_id].pad[1] = 0u;
        BvbNode l{}, r{};
        l.first = first;      l.count = nl;         l.left = BVB_NONE;
        r.first = first + nl; r.count = count - nl; r.left = BVB_NONE;
        a.nodes[id] = l;
        a.nodes[id + 1u] = r;
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * The deep levels of a tree with small leaves: every node of the level holds at most 64 triangles (0.2-0.4 M nodes of one to eight triangles on each of six levels of
 * the scattered stand-in).  One wave per node as in k_bvb_level<64>, but the node's triangles are fetched ONCE — lane i keeps triangle first + i: index, six keys,
 * centroid — and every pass of k_bvb_level (bounds, three binnings, the partition's count and its three ranking passes, the copy back) works out of registers: a small
 * node is nothing but latency, and the stride-loop kernel walks order -> triangle -> vertices eight times per node (5.4 ms for a level of 0.4 M nodes).  Same keys,
 * same bins, same sweep, same closed-form partition (its prefix counts are ballots here): the same tree. */
__global__ __launch_bounds__(64) void k_bvb_small(BvbArgs a, uint32_t level_begin, uint32_t skip_upto) {
    __shared__ unsigned long long s_key[BVB_MAX_BINS][6];
    __shared__ uint32_t s_cnt[BVB_MAX_BINS];
    __shared__ uint32_t s_rb[64], s_hole_at[64];

    const uint32_t lane = threadIdx.x;
    const uint32_t node_id = level_begin + blockIdx.x;
    BvbNode &node = a.nodes[node_id];
    if (node.pad[0] != 0u) return;
    const uint32_t first = node.first, count = node.count, S = a.bins;       /* count <= 64: the host launches this kernel only then */
    if (count <= skip_upto) return;
    if (count > 64u) {                             /* (cannot happen: the host picks the kernel by the level's largest node — said aloud rather than built wrong) */
        if (lane == 0u) atomicOr(a.node_count + 2, 1u);
        return;
    }
    const bool have = lane < count;
    uint32_t tri = 0u;
    float4 r_mn = make_float4(0, 0, 0, 0), r_mx = make_float4(0, 0, 0, 0);
    float cc[3] = {0.0f, 0.0f, 0.0f};
    if (have) {
        tri = a.order[first + lane];
        r_mn = a.recs[tri].mn; r_mx = a.recs[tri].mx;
        const float4 ce = a.recs[tri].ce;
        cc[0] = ce.x; cc[1] = ce.y; cc[2] = ce.z;
    }
    /* ---- update_node_aabb: the fold of the wave's keys */
    unsigned long long k6[6], red[6];
    bvb_rec_keys(r_mn, r_mx, lane, have, k6);
    bvb_wave_fold(k6, red);
    uint32_t cb[6];
    for (int j = 0; j < 3; ++j) {
        uint32_t nz;
        const uint32_t o = bvb_ord(cc[j], nz);
        cb[j] = have ? o : 0xffffffffu;
        cb[3 + j] = have ? o : 0u;
    }
    for (int d = 32; d >= 1; d >>= 1)
        for (int j = 0; j < 3; ++j) {
            const uint32_t lo = (uint32_t)__shfl_xor((int)cb[j], d, 64), hi = (uint32_t)__shfl_xor((int)cb[3 + j], d, 64);
            cb[j] = lo < cb[j] ? lo : cb[j];
            cb[3 + j] = hi > cb[3 + j] ? hi : cb[3 + j];
        }
    if (lane < 3u) {
        node.mn[lane] = bvb_key_value(lane == 0u ? red[0] : (lane == 1u ? red[1] : red[2]));
        node.mx[lane] = bvb_key_value(lane == 0u ? red[3] : (lane == 1u ? red[4] : red[5]));
    }
    float bmin[3], bmax[3], nmn[3], nmx[3];
    for (int j = 0; j < 3; ++j) {
        bmin[j] = bvb_unord(cb[j], 0u); bmax[j] = bvb_unord(cb[3 + j], 0u);
        nmn[j] = bvb_key_value(red[j]); nmx[j] = bvb_key_value(red[3 + j]);
    }
    /* ---- find_best_split_segmented, one axis at a time through one set of bins */
    int axis = 0;
    float split = 0.0f, cost = __builtin_inff();
    for (int ax = 0; ax < 3; ++ax) {
        if (bmin[ax] == bmax[ax]) continue;                          /* (uniform: every lane holds the same bounds) */
        for (uint32_t b = lane; b < BVB_MAX_BINS; b += 64u) {
            for (int j = 0; j < 6; ++j) s_key[b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
            s_cnt[b] = 0u;
        }
        __syncthreads();
        if (have) {
            const float x = (cc[ax] - bmin[ax]) * ((float)S / (bmax[ax] - bmin[ax]));
            uint32_t si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;
            if (si > S - 1u) si = S - 1u;
            for (int j = 0; j < 3; ++j) { atomicMin(&s_key[si][j], k6[j]); atomicMax(&s_key[si][3 + j], k6[3 + j]); }
            atomicAdd(&s_cnt[si], 1u);
        }
        __syncthreads();
        float best_cost;
        uint32_t best_i;
        bvb_wave_sweep(s_key, s_cnt, S, best_cost, best_i);
        __syncthreads();
        if (best_cost < cost) {
            cost = best_cost;
            axis = ax;
            const float scale2 = (bmax[ax] - bmin[ax]) / (float)S;
            split = bmin[ax] + scale2 * (float)(best_i + 1u);
        }
    }
    {
        BvbBox nb;
        for (int j = 0; j < 3; ++j) { nb.mn[j] = nmn[j]; nb.mx[j] = nmx[j]; }
        if (bvb_area(nb) * (float)count <= cost) {                   /* bvh.rs:272-277 */
            if (lane == 0u) node.left = BVB_NONE;
            return;
        }
    }
    /* ---- the partition in closed form (file header), its running counts as ballots */
    const float c = axis == 0 ? cc[0] : (axis == 1 ? cc[1] : cc[2]);
    const bool L = have && c < split;
    const unsigned long long all_l = __builtin_amdgcn_ballot_w64(L), all = __builtin_amdgcn_ballot_w64(have);
    const uint32_t nl = (uint32_t)__popcll(all_l);
    const unsigned long long below_nl = nl >= 64u ? ~0ull : ((1ull << nl) - 1ull);
    const unsigned long long pre_r = all & ~all_l & below_nl, suf_l = all_l & ~below_nl, suf_r = all & ~all_l & ~below_nl;
    const unsigned long long above = lane >= 63u ? 0ull : (~0ull << (lane + 1u)), under = (1ull << lane) - 1ull;
    const uint32_t H = (uint32_t)__popcll(suf_l);
    const bool in_prefix = lane < nl;
    const uint32_t m = (uint32_t)__popcll(suf_l & above), rb = (uint32_t)__popcll(suf_r & above);        /* suffix lanes: left / right elements above */
    const uint32_t hole = (uint32_t)__popcll(pre_r & under);                                             /* prefix lanes: holes below */
    if (have && !in_prefix && L) s_rb[m] = rb;
    if (have && in_prefix && !L) s_hole_at[hole] = lane;
    __syncthreads();
    uint32_t dest = lane;                                             /* a left-side element of the prefix stays */
    if (have) {
        const uint32_t last_i = count - 1u;
        if (in_prefix && !L) {
            dest = last_i - (hole + (hole >= 1u ? s_rb[hole - 1u] : 0u));
        } else if (!in_prefix && L) {
            dest = s_hole_at[m];
        } else if (!in_prefix) {
            const uint32_t base_rb = H >= 1u ? s_rb[H - 1u] : 0u;
            uint32_t rank;
            if (m == H) rank = lane == nl ? H + base_rb : H + rb + 1u;
            else rank = (m + 1u) + rb;
            dest = last_i - rank;
        }
        a.order[first + dest] = tri;
    }
    if (lane == 0u) node.pad[1] = (nl == 0u || nl == count) ? 0u : nl;          /* -> k_bvb_children */
}

/* ---------------------------------------------------------------------------------------------------------------
 * Nodes of at most BVB_TINY triangles — 87 % of the nodes of the scattered stand-in's level 18, all of them from level 21 down, two triangles on average: EIGHT nodes
 * per wave, eight lanes each, and no bins at all.  With n <= 8 triangles at most 8 of the S bins hold anything, and the sweep's candidates between two occupied bins
 * cost the same (same left set, same right set: the first of them is the candidate AT the lower occupied bin), those before the first occupied bin or from the last
 * one on have an empty side (0 x area(empty) = NaN: never taken, bvh.rs:246-250 as k_bvb_level's sweep).  So lane t prices candidate i = bin(t) by folding the keys of
 * the group's members u with bin(u) <= bin(t) into the left box and the others into the right box: n steps of one shuffle round each.  The sequential fold is by
 * triangle inside a bin, then by bin — ascending for the left boxes, descending for the right ones, the LATER operand winning a tie (which only the sign of a zero can
 * tell) — so the tie-break word of a key gets the bin on top of the triangle's sequence number, for the left boxes the other way round than for the right boxes.  A
 * bin whose least x is +inf is skipped by the sweep (encapsulate_node of an "empty" box) though its triangles count: `excluded`.  Nodes of more than BVB_TINY
 * triangles are left to the launch that follows (k_bvb_small / k_bvb_level<64> with skip_upto = BVB_TINY). */
#define BVB_TINY 8u
__device__ __forceinline__ float bvb_select3(const float v[3], uint32_t j) { return j == 0u ? v[0] : (j == 1u ? v[1] : v[2]); }
__global__ __launch_bounds__(64) void k_bvb_tiny(BvbArgs a, uint32_t level_begin, uint32_t level_end) {
    __shared__ uint32_t s_rb[64], s_hole_at[64];
    const uint32_t lane = threadIdx.x, g0 = lane & ~7u, l = lane & 7u;
    const uint32_t node_id = level_begin + blockIdx.x * 8u + (lane >> 3);
    BvbNode *node = a.nodes + (node_id < level_end ? node_id : level_begin);
    uint32_t first = 0u, count = 0u;
    bool mine = false;
    if (node_id < level_end && node->pad[0] == 0u) {
        first = node->first; count = node->count;
        mine = count <= BVB_TINY;
    }
    const uint32_t S = a.bins;
    const bool have = mine && l < count;
    uint32_t tri = 0u;
    float4 r_mn = make_float4(0, 0, 0, 0), r_mx = make_float4(0, 0, 0, 0);
    float cc[3] = {0.0f, 0.0f, 0.0f};
    if (have) {
        tri = a.order[first + l];
        r_mn = a.recs[tri].mn; r_mx = a.recs[tri].mx;
        const float4 ce = a.recs[tri].ce;
        cc[0] = ce.x; cc[1] = ce.y; cc[2] = ce.z;
    }
    /* ---- update_node_aabb over the group */
    unsigned long long k6[6], red[6];
    bvb_rec_keys(r_mn, r_mx, l, have, k6);
    uint32_t cb[6];
    for (int j = 0; j < 3; ++j) {
        uint32_t nz;
        const uint32_t o = bvb_ord(cc[j], nz);
        cb[j] = have ? o : 0xffffffffu;
        cb[3 + j] = have ? o : 0u;
    }
    for (int j = 0; j < 6; ++j) red[j] = k6[j];
    for (int d = 4; d >= 1; d >>= 1)
        for (int j = 0; j < 3; ++j) {
            const unsigned long long x = bvb_shfl_xor_u64(red[j], d), y = bvb_shfl_xor_u64(red[3 + j], d);
            red[j] = x < red[j] ? x : red[j];
            red[3 + j] = y > red[3 + j] ? y : red[3 + j];
            const uint32_t lo = (uint32_t)__shfl_xor((int)cb[j], d, 64), hi = (uint32_t)__shfl_xor((int)cb[3 + j], d, 64);
            cb[j] = lo < cb[j] ? lo : cb[j];
            cb[3 + j] = hi > cb[3 + j] ? hi : cb[3 + j];
        }
    float bmin[3], bmax[3], nmn[3], nmx[3];
    for (int j = 0; j < 3; ++j) {
        bmin[j] = bvb_unord(cb[j], 0u); bmax[j] = bvb_unord(cb[3 + j], 0u);
        nmn[j] = bvb_key_value(red[j]); nmx[j] = bvb_key_value(red[3 + j]);
    }
    if (mine && l < 3u) { node->mn[l] = bvb_select3(nmn, l); node->mx[l] = bvb_select3(nmx, l); }
    /* the longest group of the wave bounds the member loops */
    uint32_t n_max = mine ? count : 0u;
    for (int d = 32; d >= 8; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)n_max, d, 64); n_max = o > n_max ? o : n_max; }
    n_max = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_max);
    /* ---- find_best_split_segmented without bins */
    int axis = 0;
    float split = 0.0f, cost = __builtin_inff();
    for (int ax = 0; ax < 3; ++ax) {
        const bool on = mine && !(bmin[ax] == bmax[ax]);
        if (__builtin_amdgcn_ballot_w64(on) == 0ull) continue;
        uint32_t si = 0u;
        if (on && have) {
            const float x = (cc[ax] - bmin[ax]) * ((float)S / (bmax[ax] - bmin[ax]));
            si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;
            if (si > S - 1u) si = S - 1u;
        }
        /* bins the sweep skips: least x of the bin's triangles is +inf */
        bool excluded = false;
        if (__builtin_amdgcn_ballot_w64(on && have && bvb_key_value(k6[0]) == __builtin_inff()) != 0ull) {
            unsigned long long bin_min_x = k6[0];
            for (uint32_t u = 0; u < n_max; ++u) {
                const uint32_t su = (uint32_t)__shfl((int)si, (int)(g0 + u), 64);
                const unsigned long long ku = (unsigned long long)(uint32_t)__shfl((int)(uint32_t)k6[0], (int)(g0 + u), 64) |
                                              ((unsigned long long)(uint32_t)__shfl((int)(uint32_t)(k6[0] >> 32), (int)(g0 + u), 64) << 32);
                if (u < count && su == si && ku < bin_min_x) bin_min_x = ku;
            }
            excluded = bvb_key_value(bin_min_x) == __builtin_inff();
        }
        const uint32_t tag = si | (excluded ? 0x80000000u : 0u);
        unsigned long long Lk[6], Rk[6];
        for (int j = 0; j < 3; ++j) { Lk[j] = Rk[j] = BVB_MIN_IDENT; Lk[3 + j] = Rk[3 + j] = BVB_MAX_IDENT; }
        uint32_t lc = 0u, rc = 0u;
        for (uint32_t u = 0; u < n_max; ++u) {
            const uint32_t tu = (uint32_t)__shfl((int)tag, (int)(g0 + u), 64);
            uint32_t hi[6], lo[6];
            for (int j = 0; j < 6; ++j) {
                lo[j] = (uint32_t)__shfl((int)(uint32_t)k6[j], (int)(g0 + u), 64);
                hi[j] = (uint32_t)__shfl((int)(uint32_t)(k6[j] >> 32), (int)(g0 + u), 64);
            }
            const uint32_t su = tu & 0x7fffffffu;
            const bool member = u < count, is_left = su <= si;
            lc += member && is_left ? 1u : 0u;
            rc += member && !is_left ? 1u : 0u;
            const bool to_l = member && !(tu & 0x80000000u) && is_left, to_r = member && !(tu & 0x80000000u) && !is_left;
            /* the later operand of the fold wins a tie: a higher bin for the left boxes (ascending), a lower one for the right boxes (descending) */
            const uint32_t up = su << 24, down = (127u - su) << 24;
            for (int j = 0; j < 3; ++j) {
                const unsigned long long kmin = ((unsigned long long)hi[j] << 32) | ((is_left ? down : up) | (lo[j] & 0xffffffu));
                const unsigned long long kmax = ((unsigned long long)hi[3 + j] << 32) | ((is_left ? up : down) | (lo[3 + j] & 0xffffffu));
                Lk[j] = to_l && kmin < Lk[j] ? kmin : Lk[j];
                Lk[3 + j] = to_l && kmax > Lk[3 + j] ? kmax : Lk[3 + j];
                Rk[j] = to_r && kmin < Rk[j] ? kmin : Rk[j];
                Rk[3 + j] = to_r && kmax > Rk[3 + j] ? kmax : Rk[3 + j];
            }
        }
        BvbBox lb = bvb_box_identity(), rb = bvb_box_identity();
        if (Lk[0] != BVB_MIN_IDENT) for (int j = 0; j < 3; ++j) { lb.mn[j] = bvb_key_value(Lk[j]); lb.mx[j] = bvb_key_value(Lk[3 + j]); }
        if (Rk[0] != BVB_MIN_IDENT) for (int j = 0; j < 3; ++j) { rb.mn[j] = bvb_key_value(Rk[j]); rb.mx[j] = bvb_key_value(Rk[3 + j]); }
        float c_here = __builtin_inff();
        uint32_t at = 0u;
        if (on && have && si + 1u < S) {
            const float v = (float)lc * bvb_area(lb) + (float)rc * bvb_area(rb);
            if (v < c_here) { c_here = v; at = si; }
        }
        for (int d = 4; d >= 1; d >>= 1) {
            const float oc = __shfl_xor(c_here, d, 64);
            const uint32_t oa = (uint32_t)__shfl_xor((int)at, d, 64);
            if (oc < c_here || (oc == c_here && oc != __builtin_inff() && oa < at)) { c_here = oc; at = oa; }
        }
        if (c_here < cost) {
            cost = c_here;
            axis = ax;
            const float scale2 = (bmax[ax] - bmin[ax]) / (float)S;
            split = bmin[ax] + scale2 * (float)(at + 1u);
        }
    }
    bool splits = false;
    if (mine) {
        BvbBox nb;
        for (int j = 0; j < 3; ++j) { nb.mn[j] = nmn[j]; nb.mx[j] = nmx[j]; }
        splits = !(bvb_area(nb) * (float)count <= cost);             /* bvh.rs:272-277 */
        if (!splits && l == 0u) node->left = BVB_NONE;
    }
    /* ---- the partition in closed form, the running counts as bits of a ballot */
    const float c = axis == 0 ? cc[0] : (axis == 1 ? cc[1] : cc[2]);
    const bool L = splits && have && c < split;
    const uint32_t grp_l = (uint32_t)(__builtin_amdgcn_ballot_w64(L) >> g0) & 0xffu;
    const uint32_t grp = (uint32_t)(__builtin_amdgcn_ballot_w64(splits && have) >> g0) & 0xffu;
    const uint32_t nl = (uint32_t)__popc(grp_l);
    const uint32_t below_nl = (1u << nl) - 1u;
    const uint32_t pre_r = grp & ~grp_l & below_nl, suf_l = grp_l & ~below_nl, suf_r = grp & ~grp_l & ~below_nl;
    const uint32_t above = (~0u << (l + 1u)) & 0xffu, under = (1u << l) - 1u;
    const uint32_t H = (uint32_t)__popc(suf_l);
    const bool in_prefix = l < nl;
    const uint32_t m = (uint32_t)__popc(suf_l & above), rb_above = (uint32_t)__popc(suf_r & above);
    const uint32_t hole = (uint32_t)__popc(pre_r & under);
    if (splits && have && !in_prefix && L) s_rb[g0 + m] = rb_above;
    if (splits && have && in_prefix && !L) s_hole_at[g0 + hole] = l;
    __syncthreads();
    if (splits && have) {
        const uint32_t last_i = count - 1u;
        uint32_t dest = l;
        if (in_prefix && !L) {
            dest = last_i - (hole + (hole >= 1u ? s_rb[g0 + hole - 1u] : 0u));
        } else if (!in_prefix && L) {
            dest = s_hole_at[g0 + m];
        } else if (!in_prefix) {
            const uint32_t base_rb = H >= 1u ? s_rb[g0 + H - 1u] : 0u;
            uint32_t rank;
            if (m == H) rank = l == nl ? H + base_rb : H + rb_above + 1u;
            else rank = (m + 1u) + rb_above;
            dest = last_i - rank;
        }
        a.order[first + dest] = tri;
    }
    if (splits && l == 0u) node->pad[1] = (nl == 0u || nl == count) ? 0u : nl;          /* -> k_bvb_children (bvh.rs:294-296: an empty side, and the node stays a leaf) */
}

/* ---------------------------------------------------------------------------------------------------------------
 * The big nodes at the top of a large tree: a TEAM of 2 ... BVB_TEAM workgroups per node (one per BVB_TEAM_CHUNK triangles) instead of one.
 * Same arithmetic, same keys, same closed-form partition — the passes are cut into contiguous chunks of the node's
 * range, partial results meet in global memory (64-bit atomic min/max on the same keys, per-chunk left counts), and
 * the workgroups of a team meet at a counter barrier between passes (all of them are resident: the host launches no
 * more workgroups than the device holds at once).  With L(p) = number of left-side elements at positions < p (chunk prefix + local rank)
 * every quantity of the closed form is local:  hole rank of p = (p - first) - L(p);  for a suffix position q:
 * m(q) = nl - L(q+1) left-side elements above it, rb(q) = (last - q) - m(q) right-side elements above it.
 * Measured on the 1 M-triangle stand-in: the levels that hold a 0.5-1 M-triangle node took 7-15 ms each with one
 * workgroup per node. */
#define BVB_TEAM 256               /* the largest team */
#define BVB_TEAM_THREADS 256
#define BVB_TEAM_CHUNK 8192u       /* triangles per workgroup a team is sized for (a power of two of workgroups, 2 ... BVB_TEAM): measured 1 024 ... 32 768 */
#define BVB_MAX_TEAMS 128          /* all workgroups of all teams of a launch are resident at once (the barrier needs that): the host counts them */
#define BVB_TEAM_MIN_COUNT 16384u  /* measured on both 1 M-triangle stand-ins, 4 096 ... 32 768 (profiles/r06_bvh_team_sweep.txt) */
#define BVB_WIDE_MIN_COUNT 16384u  /* a level whose largest node is at least this big gets 1 024 threads per node (4 waves per SIMD walk a 30 k-triangle node 4 x faster) */

struct BvbTeamRef {
    uint32_t node_id, size, first_block;
};
struct BvbTeamScratch {
    unsigned long long red[6];
    unsigned long long key[3][BVB_MAX_BINS][6];
    uint32_t cb[6];
    uint32_t cnt[3][BVB_MAX_BINS];
    uint32_t chunk_l[BVB_TEAM];
    uint32_t barrier;
    int axis;
    float split;
    uint32_t node_id, size, first_block;
};

__global__ __launch_bounds__(BVB_TEAM_THREADS) void k_bvb_team_init(BvbTeamScratch *scratch, const BvbTeamRef *refs) {
    BvbTeamScratch &t = scratch[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    if (tid < 6u) { t.red[tid] = tid < 3u ? BVB_MIN_IDENT : BVB_MAX_IDENT; t.cb[tid] = tid < 3u ? 0xffffffffu : 0u; }
    for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += BVB_TEAM_THREADS) {
        const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
        for (int j = 0; j < 6; ++j) t.key[ax][b][j] = j < 3 ? BVB_MIN_IDENT : BVB_MAX_IDENT;
        t.cnt[ax][b] = 0u;
    }
    for (uint32_t k = tid; k < BVB_TEAM; k += BVB_TEAM_THREADS) t.chunk_l[k] = 0u;
    if (tid == 0u) {
        t.barrier = 0u; t.axis = -1; t.split = 0.0f;
        t.node_id = refs[blockIdx.x].node_id; t.size = refs[blockIdx.x].size; t.first_block = refs[blockIdx.x].first_block;
    }
}

/* all workgroups of a team arrive; `phase` counts the barriers passed so far */
__device__ __forceinline__ void bvb_team_sync(uint32_t *counter, uint32_t &phase, uint32_t team_size) {
    __syncthreads();
    phase += 1u;
    if (threadIdx.x == 0u) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase * team_size) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(BVB_TEAM_THREADS) void k_bvb_team(BvbArgs a, BvbTeamScratch *scratch, const uint16_t *block_team) {
    constexpr int THREADS = BVB_TEAM_THREADS;
    __shared__ unsigned long long s_key[3][BVB_MAX_BINS][6];
    __shared__ uint32_t s_cnt[3][BVB_MAX_BINS];
    __shared__ unsigned long long s_red[6];
    __shared__ uint32_t s_cb[6];
    __shared__ float s_best_cost[3];
    __shared__ uint32_t s_best_i[3];
    __shared__ uint32_t s_wave_tot[THREADS / 64];

    const uint32_t tid = threadIdx.x;
    BvbTeamScratch &T = scratch[block_team[blockIdx.x]];
    const uint32_t team_size = T.size, member = blockIdx.x - T.first_block;
    BvbNode &node = a.nodes[T.node_id];
    const uint32_t first = node.first, count = node.count, S = a.bins;
    const uint32_t last = first + count - 1u;
    const uint32_t chunk = (count + team_size - 1u) / team_size;
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
            const float4 r_mn = a.recs[tri].mn, r_mx = a.recs[tri].mx, ce = a.recs[tri].ce;
            unsigned long long k6[6];
            bvb_rec_keys(r_mn, r_mx, i, true, k6);
            for (int j = 0; j < 3; ++j) {
                kmin[j] = k6[j] < kmin[j] ? k6[j] : kmin[j];
                kmax[j] = k6[3 + j] > kmax[j] ? k6[3 + j] : kmax[j];
            }
            const float cc[3] = {ce.x, ce.y, ce.z};
            for (int j = 0; j < 3; ++j) {
                uint32_t nz;
                uint32_t o = bvb_ord(cc[j], nz);
                cmin[j] = o < cmin[j] ? o : cmin[j];
                cmax[j] = o > cmax[j] ? o : cmax[j];
            }
        }
        /* the wave's fold by shuffles, then ONE lane per wave into the workgroup's words: an LDS atomic on one address is served lane by lane — 1 024 threads
         * x 12 of them were 0.10 of the 0.24 ms a 16 k-triangle node took */
        for (int d = 32; d >= 1; d >>= 1)
            for (int j = 0; j < 3; ++j) {
                const unsigned long long x = bvb_shfl_xor_u64(kmin[j], d), y = bvb_shfl_xor_u64(kmax[j], d);
                kmin[j] = x < kmin[j] ? x : kmin[j];
                kmax[j] = y > kmax[j] ? y : kmax[j];
                const uint32_t lo = (uint32_t)__shfl_xor((int)cmin[j], d, 64), hi = (uint32_t)__shfl_xor((int)cmax[j], d, 64);
                cmin[j] = lo < cmin[j] ? lo : cmin[j];
                cmax[j] = hi > cmax[j] ? hi : cmax[j];
            }
        if ((tid & 63u) == 0u)
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
    bvb_team_sync(&T.barrier, phase, team_size);
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
    const uint32_t c_n = c_end - c_begin, deal_mask = bvb_mask_for(c_n);
    for (uint32_t base = 0; base < c_n; base += THREADS) {
        const bool have = base + tid < c_n;
        const uint32_t pos = c_begin + (have ? bvb_scatter_index(base + tid, c_n, deal_mask) : 0u);
        const uint32_t i = pos - first;
        float4 r_mn = make_float4(0, 0, 0, 0), r_mx = make_float4(0, 0, 0, 0);
        float cc[3] = {0.0f, 0.0f, 0.0f};
        if (have) {
            const uint32_t tri = a.order[pos];
            r_mn = a.recs[tri].mn; r_mx = a.recs[tri].mx;
            const float4 ce = a.recs[tri].ce;
            cc[0] = ce.x; cc[1] = ce.y; cc[2] = ce.z;
        }
        unsigned long long k6[6];
        bvb_rec_keys(r_mn, r_mx, i, have, k6);
        for (int ax = 0; ax < 3; ++ax) {
            if (!axis_on[ax]) continue;
            const float x = (cc[ax] - bmin[ax]) * scale[ax];
            uint32_t si = x > 0.0f ? (x >= (float)S ? S - 1u : (uint32_t)x) : 0u;
            if (si > S - 1u) si = S - 1u;
            bvb_bin_add(s_key[ax], s_cnt[ax], have, si, k6);
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
    bvb_team_sync(&T.barrier, phase, team_size);

    /* ---- pass 3: member 0 evaluates the splits exactly as the single-workgroup kernel does */
    if (member == 0u) {
        for (uint32_t k = tid; k < 3u * BVB_MAX_BINS; k += THREADS) {
            const uint32_t ax = k / BVB_MAX_BINS, b = k % BVB_MAX_BINS;
            for (int j = 0; j < 6; ++j) s_key[ax][b][j] = __hip_atomic_load(&T.key[ax][b][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_cnt[ax][b] = __hip_atomic_load(&T.cnt[ax][b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 192u) {                                      /* waves 0, 1, 2: one axis each (bvb_wave_sweep) */
            const uint32_t ax = tid >> 6;
            float best_cost = __builtin_inff();
            uint32_t best_i = 0u;
            if (axis_on[ax]) bvb_wave_sweep(s_key[ax], s_cnt[ax], S, best_cost, best_i);
            if ((tid & 63u) == 0u) { s_best_cost[ax] = best_cost; s_best_i[ax] = best_i; }
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
    bvb_team_sync(&T.barrier, phase, team_size);
    const int axis = T.axis;
    if (axis < 0) {
        if (member == 0u && tid == 0u) { node.left = BVB_NONE; node.pad[0] = 1u; }
        return;                                                   /* team-uniform */
    }
    const float split = T.split;

    /* ---- pass 4: left-side elements before every position of the chunk (bvb_count_left) -> this chunk's count */
    {
        const uint32_t mine = bvb_count_left<THREADS>(a, c_begin, c_end, axis, split, s_wave_tot);
        if (tid == 0u) T.chunk_l[member] = mine;
    }
    bvb_team_sync(&T.barrier, phase, team_size);
    /* nl, this chunk's prefix, and L(split position) = the left-side elements inside the prefix: the count before the chunk that holds the split position
     * + that position's own word (written by that chunk's workgroup before the barrier) */
    uint32_t nl = 0u, pref_l = 0u;
    for (uint32_t b = 0; b < team_size; ++b) {
        const uint32_t n = __hip_atomic_load(&T.chunk_l[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pref_l += b < member ? n : 0u;
        nl += n;
    }
    const uint32_t split_pos = first + nl;                       /* prefix = [first, split_pos), suffix = [split_pos, last] */
    uint32_t l_in_prefix = nl;
    if (nl < count) {
        const uint32_t c_split = nl / chunk;
        l_in_prefix = __hip_atomic_load(&a.lpre[split_pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 1;
        for (uint32_t b = 0; b < c_split; ++b) l_in_prefix += __hip_atomic_load(&T.chunk_l[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t H = nl - l_in_prefix;

    /* ---- pass 6 (B1): rb_at_L for the suffix left-side elements of this chunk */
    for (uint32_t q = (c_begin > split_pos ? c_begin : split_pos) + tid; q < c_end; q += THREADS) {
        const uint32_t w = a.lpre[q];
        if (w & 1u) {
            const uint32_t m = nl - (pref_l + (w >> 1) + 1u);
            a.tmp_b[first + m] = (last - q) - m;
        }
    }
    bvb_team_sync(&T.barrier, phase, team_size);
    const uint32_t base_rb = H >= 1u ? __hip_atomic_load(&a.tmp_b[first + H - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    /* ---- pass 7 (F): the prefix */
    for (uint32_t p = c_begin + tid; p < c_end && p < split_pos; p += THREADS) {
        const uint32_t w = a.lpre[p];
        if (w & 1u) {
            a.order_tmp[p] = a.order[p];
        } else {
            const uint32_t hole = (p - first) - (pref_l + (w >> 1));
            const uint32_t rank = hole + (hole >= 1u ? __hip_atomic_load(&a.tmp_b[first + hole - 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
            a.order_tmp[last - rank] = a.order[p];
            a.tmp_a[first + hole] = p;
        }
    }
    bvb_team_sync(&T.barrier, phase, team_size);
    /* ---- pass 8 (B2): the suffix */
    for (uint32_t q = (c_begin > split_pos ? c_begin : split_pos) + tid; q < c_end; q += THREADS) {
        const uint32_t w = a.lpre[q];
        const uint32_t l_before = pref_l + (w >> 1);
        if (w & 1u) {
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
    bvb_team_sync(&T.barrier, phase, team_size);
    for (uint32_t pos = c_begin + tid; pos < c_end; pos += THREADS)
        a.order[pos] = __hip_atomic_load(&a.order_tmp[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (member == 0u && tid == 0u) {
        node.pad[0] = 1u;
        if (nl == 0u || nl == count) {
            node.left = BVB_NONE;
        } else {
            const uint32_t id = atomicAdd(a.node_count, 2u);
            node.left = id;
            a.nodes[id].first = first;          a.nodes[id].count = nl;              a.nodes[id].left = BVB_NONE;   a.nodes[id].pad[0] = 0u;   a.nodes[id].pad[1] = 0u;
            a.nodes[id + 1u].first = first + nl; a.nodes[id + 1u].count = count - nl; a.nodes[id + 1u].left = BVB_NONE; a.nodes[id + 1u].pad[0] = 0u; a.nodes[id + 1u].pad[1] = 0u;
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
