/*
 * k_traverse_top.h — streamed walks for MID-SIZE scenes: the TOP of the tree in LDS, the rest behind deferred global bodies.
 *
 * Scenes whose traversal image does not fit the 32 KB of the LDS walk (VeachMIS: 2 887 child pairs, FurnaceTest 10 k, PBRTest 24 k)
 * read every node through the CU's one texture-address path — 64 bytes per lane and visit, 16 cycles per wave load instruction — and
 * wait for L1 / L2 two thirds of their cycles (profiles/r03_{veachmis,pbrtest}_pmc_sq.txt).  gfx950 has 160 KB of LDS per CU, and the
 * visits are concentrated (tools/node_visit_share.py, profiles/r04_node_visit_share.txt): the 1 000 largest pairs of VeachMIS take
 * 99.9 % of the pair tests of its nearest-hit walks and 94 % of those of its shadow walks.
 *
 * So ONE 1 024-thread workgroup per CU holds
 *   - the plane records + child ids of the K "top" child pairs (the image format of the LDS walk, k_traverse.h SceneViewLds: sign-selected
 *     plane records, one max3 + one min3 per box) — chosen at upload by a best-first descent on the surface area of the parent box (the
 *     SAH's visit probability: camera independent, closed under "parent of");
 *   - the 16-bit stacks of its 16 waves,
 * and walks with ONE body per trip, the one most lanes wait for:
 *   image body    the pair is in LDS: the LDS walk's inner step, nothing else in it (the two attempts of round 2 / 3 at a 127- / 192-pair
 *                 cache put a second load path and its branch INSIDE the inner step and lost 20 %)
 *   global body   a pair below the image: four 16-byte loads from a pair array in global memory, the generic slab test
 *   leaf body     leaf record (first, count) + triangle records from global memory, the reference's sequential triangle loop
 * A lane whose next step needs another body than the one issued PARKS on its node, exactly as leaf lanes always have (k_traverse.h,
 * lds_walk_run).  Per ray the visiting order and every comparison are the reference's (intersection.rs:177-234): hit records are
 * bit-identical whichever body ran where.
 *
 * Nodes are renumbered for this walk ("tid", 16 bits): image pairs' parents first [0, K), then the other inner nodes [K, K2), then the
 * leaves [K2, n) — the kind of a node is a range test on its id, and a stack entry is the id.
 */
#ifndef RPT_K_TRAVERSE_TOP_H
#define RPT_K_TRAVERSE_TOP_H

#include "k_traverse.h"

#define TOP_DEAD 0xffffu
#ifndef RPT_TOP_THREADS
#define RPT_TOP_THREADS 1024
#endif
/* body choice: a global / leaf step waits for memory, so it should not be issued for a handful of lanes while many could take an image
 * step: the image body runs unless `pct` % of its lane count is exceeded by the larger of the two others (100 = plain majority) */
#ifndef RPT_TOP_GLOBAL_PCT
#define RPT_TOP_GLOBAL_PCT 100
#endif
#ifndef RPT_TOP_TRIPS
#define RPT_TOP_TRIPS 16
#endif
#ifndef RPT_TOP_REFILL
#define RPT_TOP_REFILL 16
#endif

struct SceneViewTop {
    const float4 *img;            /* LDS: 6 K plane records, then K child-id words (l | r << 16) */
    uint32_t K, K2;               /* ids < K: pair in the image; < K2: pair in `gpairs`; else leaf */
    const float4 *gpairs;         /* global: 4 x float4 per inner node K <= id < K2: (L.lo | id L) (L.hi | -) (R.lo | id R) (R.hi | -) */
    const uint32_t *leaves;       /* global: first triangle | count << 24 per leaf id - K2 */
    uint32_t leaf_lds_vecs;       /* != 0: the table is also in LDS, that many float4 behind `img` (a dependent global load less per leaf step) */
    const float *tri_isect;
    __device__ __forceinline__ void edges(uint32_t ti, F3 &e1, F3 &e2) const {
        const float *p = tri_isect + 9u * (size_t)ti;
        e1 = f3(p[0], p[1], p[2]); e2 = f3(p[3], p[4], p[5]);
    }
    __device__ __forceinline__ F3 corner(uint32_t ti) const {
        const float *p = tri_isect + 9u * (size_t)ti + 6u;
        return f3(p[0], p[1], p[2]);
    }
};

struct TopWalk {
    uint32_t cur;          /* id of the node the ray stands on; TOP_DEAD when finished / no ray */
    int sp;
    HitRecord res;
};
__device__ __forceinline__ void top_walk_begin(TopWalk &w) {
    w.cur = 0u;            /* the root is an inner node and the first pair of the image */
    w.sp = 0;
    w.res.t = 1000000.0f;
    w.res.tri = HIT_MISS;
}

/* At most `budget` trips for the lanes of this wave; rays inside the exact-division guard only (the caller walks the others alone). */
template <int STACK, bool ANY_HIT, bool LEAF_LDS>
__device__ __forceinline__ void top_walk_run(const SceneViewTop &view, TopWalk &w, F3 ro, F3 rd, F3 ird, float max_t, uint16_t *stack, int budget) {
    const uint32_t K = view.K, K2 = view.K2;
    const float4 *px = view.img + (rd.x < 0.0f ? K : 0u);
    const float4 *py = view.img + 2u * K + (rd.y < 0.0f ? K : 0u);
    const float4 *pz = view.img + 4u * K + (rd.z < 0.0f ? K : 0u);
    const uint32_t *ids = reinterpret_cast<const uint32_t *>(view.img + 6u * K);
    uint32_t cur = w.cur;
    int sp = w.sp;
    HitRecord res = w.res;
    for (int trip = 0; trip < budget; ++trip) {
        const bool at_img = cur < K;
        const bool at_glb = cur >= K && cur < K2;
        const bool at_leaf = cur >= K2 && cur < TOP_DEAD;
        const unsigned long long img_m = rpt_ballot(at_img), glb_m = rpt_ballot(at_glb), leaf_m = rpt_ballot(at_leaf);
        if ((img_m | glb_m | leaf_m) == 0ull) break;
        const uint32_t n_img = (uint32_t)__popcll(img_m), n_glb = (uint32_t)__popcll(glb_m), n_leaf = (uint32_t)__popcll(leaf_m);
        const uint32_t n_other = n_glb > n_leaf ? n_glb : n_leaf;
        const bool do_img = n_img * (uint32_t)RPT_TOP_GLOBAL_PCT >= n_other * 100u && n_img != 0u;
        const bool do_leaf = !do_img && n_leaf >= n_glb;
        if (do_img) {
            if (at_img) {
                const float4 X = px[cur], Y = py[cur], Z = pz[cur];     /* (L.near, R.near, L.far, R.far) per axis */
                const uint32_t d = ids[cur];
                float tl, tr;
                const bool hit_l = slab_pair_lds<true>(X.x, Y.x, Z.x, X.z, Y.z, Z.z, ro, rd, ird, res.t, tl);
                const bool hit_r = slab_pair_lds<true>(X.y, Y.y, Z.y, X.w, Y.w, Z.w, ro, rd, ird, res.t, tr);
                const bool swap = hit_r && (!hit_l || tl > tr);     /* strict: ties keep left first */
                if (hit_l || hit_r) {
                    const uint32_t nf = __builtin_amdgcn_alignbit(d, d, swap ? 16u : 0u);    /* near | far << 16 */
                    if (hit_l && hit_r && sp < STACK) {
                        stack[sp * RPT_WAVE] = (uint16_t)(nf >> 16);
                        sp += 1;
                    }
                    cur = nf & 0xffffu;
                } else if (sp == 0) {
                    cur = TOP_DEAD;
                } else {
                    sp -= 1;
                    cur = stack[sp * RPT_WAVE];
                }
            }
        } else if (!do_leaf) {
            if (at_glb) {
                const float4 *ch = view.gpairs + 4u * (cur - K);
                const float4 lmin = ch[0], lmax = ch[1], rmin = ch[2], rmax = ch[3];
                float tl, tr;
                const bool hit_l = slab_test<true>(lmin, lmax, ro, rd, ird, res.t, tl);
                const bool hit_r = slab_test<true>(rmin, rmax, ro, rd, ird, res.t, tr);
                const bool swap = hit_r && (!hit_l || tl > tr);
                const uint32_t il = __float_as_uint(lmin.w), ir = __float_as_uint(rmin.w);
                if (hit_l || hit_r) {
                    if (hit_l && hit_r && sp < STACK) {
                        stack[sp * RPT_WAVE] = (uint16_t)(swap ? il : ir);
                        sp += 1;
                    }
                    cur = swap ? ir : il;
                } else if (sp == 0) {
                    cur = TOP_DEAD;
                } else {
                    sp -= 1;
                    cur = stack[sp * RPT_WAVE];
                }
            }
        } else if (at_leaf) {
            bool accepted = false;
            /* (a template parameter, not a run-time choice: a select between an LDS and a global address becomes a FLAT load) */
            const uint32_t rec = LEAF_LDS ? reinterpret_cast<const uint32_t *>(view.img + view.leaf_lds_vecs)[cur - K2] : view.leaves[cur - K2];
            const uint32_t count = rec >> 24, first = rec & 0xffffffu;
            for (uint32_t i = 0; i < count; ++i) {
                uint32_t ti = first + i;
                float t = 0.0f;
                bool bf = false;
                /* the whole 36-byte record at once: with 4 waves per SIMD a second dependent load (the corner, which the LDS walk fetches
                 * only past the determinant test) is a memory round trip nobody hides */
                const float *p = view.tri_isect + 9u * (size_t)ti;
                const F3 e1 = f3(p[0], p[1], p[2]), e2 = f3(p[3], p[4], p[5]), a = f3(p[6], p[7], p[8]);
                if (moller_trumbore_regs(e1, e2, a, ro, rd, t, bf) && t > 0.001f && t < res.t && (!ANY_HIT || t <= max_t)) {
                    asm volatile("" ::: "memory");          /* (a real branch: see lds_walk_run) */
                    res.t = t;
                    res.tri = ti | (bf ? 0x80000000u : 0u);
                    if (ANY_HIT) { accepted = true; break; }
                }
            }
            if ((ANY_HIT && accepted) || sp == 0) {
                cur = TOP_DEAD;
            } else {
                sp -= 1;
                cur = stack[sp * RPT_WAVE];
            }
        }
    }
    w.cur = cur;
    w.sp = sp;
    w.res = res;
}

/* all LDS of these kernels is dynamic: [ image (top_vecs float4) | leaf table (if it fits) | stacks NW x entries x 64 x u16 | per-wave refill
 * scratch NW x 64 x u32 | pool ] — together up to 160 KB.  `entries` = the tree's depth + 1 (<= STACK), not the template's cap: 16 waves
 * pay 2 KB per stack level. */
template <int THREADS>
__device__ __forceinline__ SceneViewTop stage_scene_top(const DevScene &sc, float4 *lds, uint16_t *&stacks, uint32_t *&scratch, WgPool *&pool) {
    uint32_t *leaf_lds = reinterpret_cast<uint32_t *>(lds + sc.top_vecs);
    stacks = reinterpret_cast<uint16_t *>(leaf_lds + sc.top_leaf_words);
    scratch = reinterpret_cast<uint32_t *>(stacks + (size_t)(THREADS / RPT_WAVE) * sc.top_stack * RPT_WAVE);
    pool = reinterpret_cast<WgPool *>(scratch + (size_t)(THREADS / RPT_WAVE) * RPT_WAVE);
    return SceneViewTop{lds, sc.top_pairs, sc.top_k2, sc.top_gpairs, sc.top_leaves, sc.top_leaf_words ? sc.top_vecs : 0u, sc.tri_isect};
}
/* copy the image (and the leaf table) in; the caller synchronises */
template <int THREADS>
__device__ __forceinline__ void copy_scene_top(const DevScene &sc, float4 *lds) {
    for (uint32_t k = threadIdx.x; k < sc.top_vecs; k += THREADS) lds[k] = sc.top_image[k];
    uint32_t *leaf_lds = reinterpret_cast<uint32_t *>(lds + sc.top_vecs);
    for (uint32_t k = threadIdx.x; k < sc.top_leaf_words; k += THREADS) leaf_lds[k] = sc.top_leaves[k];
}

/* a ray outside the exact-division guard (a zero / denormal-small direction component): walked alone through the reference's own node
 * array by the generic loop, on the lane's stack column (node indices < 65 536 fit its 16 bits) */
template <int STACK, bool ANY_HIT>
__device__ __forceinline__ HitRecord top_walk_slow(const DevScene &sc, F3 ro, F3 rd, float max_t, uint16_t *stack) {
    const SceneViewGlobalT<false> g{sc.nodes, sc.tri_isect};
    return traverse_loop<STACK, ANY_HIT, false>(g, ro, rd, rd, max_t, stack);
}

template <int STACK, int THREADS, bool LEAF_LDS>
__global__ __launch_bounds__(THREADS) void k_traverse_nearest_tstream(DevScene sc, DevState st, DevQueues q, uint32_t iteration, uint32_t SPAN) {
    float4 *lds = rpt_lds_dyn;
    uint16_t *stacks;
    uint32_t *scratch_all;
    WgPool *poolp;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    SceneViewTop view = stage_scene_top<THREADS>(sc, lds, stacks, scratch_all, poolp);
    WgPool &pool = *poolp;
    uint32_t *global_next = &q.count[Q_POOL0 + (iteration & 1u) * Q_LINE];
    if (blockIdx.x == 0u && threadIdx.x == 0u) {
        iteration_bookkeeping(q, iteration);
        q.count[Q_POOL0 + ((iteration + 1u) & 1u) * Q_LINE] = 0u;
    }
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    if (threadIdx.x == 0u) {
        const uint32_t g = atomicAdd(global_next, SPAN);
        pool.word = g < st.n_slots ? ((unsigned long long)(g + SPAN < st.n_slots ? g + SPAN : st.n_slots) << 32) | g : 0x00000000f0000000ull;
        pool.lock = 0u;
    }
    __syncthreads();
    if ((uint32_t)(pool.word >> 32) == 0u) return;             /* block-uniform: a late workgroup, nothing left */
    copy_scene_top<THREADS>(sc, lds);
    __syncthreads();
    uint16_t *stack = stacks + (size_t)wave * sc.top_stack * RPT_WAVE + lane;
    volatile uint32_t *scratch = scratch_all + wave * RPT_WAVE;
    F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
    TopWalk w;
    top_walk_begin(w);
    w.cur = TOP_DEAD;
    uint32_t slot = 0u;
    bool have = false;                                         /* this lane holds a ray whose result is not written yet */
    bool pool_open = true;                                     /* wave-uniform: the launch may still have slots */
    uint32_t traced = 0u;                                      /* wave-uniform */
    /* An open scene leaves most slots without a pending ray after the first bounce: the wave LOOKS at 64 consecutive slots at a time
     * (one coalesced load of their stage words) and keeps the pending ones of that chunk as a mask; idle lanes take them in order. */
    unsigned long long left_m = 0ull;                          /* wave-uniform: pending, not yet taken slots of the chunk at scan_base */
    uint32_t scan_base = 0u;
    for (;;) {
        const unsigned long long idle_m = rpt_ballot(w.cur == TOP_DEAD);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if ((pool_open || left_m != 0ull) && n_idle >= (uint32_t)RPT_TOP_REFILL) {
            if (w.cur == TOP_DEAD && have) {
                st.hit[slot] = make_float2(w.res.t, __uint_as_float(w.res.tri));
                have = false;
            }
            if (left_m == 0ull) {
                uint32_t base = 0u, got = 0u;
                bool finished = false;
                if (lane == 0u) base = wg_pool_take(&pool, global_next, st.n_slots, SPAN, RPT_WAVE, got, finished);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
                pool_open = __builtin_amdgcn_readfirstlane((int)finished) == 0;
                if (got == 0u) {
                    if (pool_open && idle_m == ~0ull) __builtin_amdgcn_s_sleep(8);   /* another wave is fetching the next span */
                    if (!pool_open && idle_m == ~0ull) break;
                    if (idle_m == ~0ull) continue;
                } else {
                    left_m = rpt_ballot(lane < got && __float_as_uint(st.hit[base + lane].y) == HIT_PENDING);
                    scan_base = base;
                }
            }
            if (left_m != 0ull) {
                const bool mine = ((left_m >> lane) & 1ull) != 0ull;
                const uint32_t rank_p = __builtin_amdgcn_mbcnt_hi((uint32_t)(left_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)left_m, 0u));
                const uint32_t n_p = (uint32_t)__popcll(left_m), n_take = n_p < n_idle ? n_p : n_idle;
                if (mine) scratch[rank_p] = scan_base + lane;
                __builtin_amdgcn_wave_barrier();
                const uint32_t rank_i = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                bool took = false;
                if (w.cur == TOP_DEAD && rank_i < n_take) {
                    const uint32_t cand = scratch[rank_i];
                    const float4 ra = st.ray_a[cand];
                    const float2 rb = st.ray_b[cand];
                    ro = f3(ra.x, ra.y, ra.z); rd = f3(ra.w, rb.x, rb.y);
                    slot = cand;
                    took = true;
                    if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                        ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        top_walk_begin(w);
                        have = true;
                    } else {
                        HitRecord h = top_walk_slow<STACK, false>(sc, ro, rd, 0.0f, stack);
                        st.hit[cand] = make_float2(h.t, __uint_as_float(h.tri));
                    }
                }
                __builtin_amdgcn_wave_barrier();
                traced += (uint32_t)__popcll(rpt_ballot(took));
                left_m &= ~rpt_ballot(mine && rank_p < n_take);
                continue;                                      /* look again: lanes may still be idle, the chunk or the pool may hold more */
            }
        }
        if (idle_m == ~0ull) {
            if (!pool_open && left_m == 0ull) break;           /* nothing in flight and nothing left to hand out */
            continue;
        }
        top_walk_run<STACK, false, LEAF_LDS>(view, w, ro, rd, ird, 0.0f, stack, (pool_open || left_m != 0ull) ? RPT_TOP_TRIPS : 0x7fffffff);
    }
    if (have) st.hit[slot] = make_float2(w.res.t, __uint_as_float(w.res.tri));
    if (lane == 0u && traced != 0u) {
        raise_flag(&q.count[Q_ALIVE0 + (iteration & 1u) * Q_LINE]);
        atomicAdd(&q.ray_shards[(blockIdx.x % RPT_STAT_SHARDS) * RPT_STAT_STRIDE], (unsigned long long)traced);
    }
}

/* shadow rays, as k_traverse_shadow_stream: lanes note "occluded" in the entry's contribution record, k_shadow_resolve adds the NEE terms */
template <int STACK, int THREADS, bool LEAF_LDS>
__global__ __launch_bounds__(THREADS) void k_traverse_shadow_tstream(DevScene sc, DevState st, DevQueues q, DevStats *stats, uint32_t SPAN) {
    float4 *lds = rpt_lds_dyn;
    uint16_t *stacks;
    uint32_t *scratch_all;
    WgPool *poolp;
    if (q.count[Q_DRAINED] != 0u) return;                      /* surplus launch (grid-uniform) */
    SceneViewTop view = stage_scene_top<THREADS>(sc, lds, stacks, scratch_all, poolp);
    WgPool &pool = *poolp;
    uint32_t n, n_entries;                                     /* n: queue positions to hand out */
    q_extent(q.shadow_cnt, n, n_entries);
    uint32_t *global_next = &q.count[Q_SPOOL];                 /* zeroed by the shade stage of this iteration */
    if (blockIdx.x == 0u && threadIdx.x == 0u && n_entries) atomicAdd(&stats->shadow_rays, (unsigned long long)n_entries);
    const uint32_t lane = __lane_id(), wave = threadIdx.x / RPT_WAVE;
    if (threadIdx.x == 0u) {
        const uint32_t g = n ? atomicAdd(global_next, SPAN) : 0u;
        pool.word = g < n ? ((unsigned long long)(g + SPAN < n ? g + SPAN : n) << 32) | g : 0x00000000f0000000ull;
        pool.lock = 0u;
    }
    __syncthreads();
    if ((uint32_t)(pool.word >> 32) == 0u) return;             /* block-uniform: nothing (left) to trace */
    copy_scene_top<THREADS>(sc, lds);
    __syncthreads();
    uint16_t *stack = stacks + (size_t)wave * sc.top_stack * RPT_WAVE + lane;
    F3 ro = f3(0, 0, 0), rd = f3(1, 1, 1), ird = f3(1, 1, 1);
    float max_t = 0.0f;
    TopWalk w;
    top_walk_begin(w);
    w.cur = TOP_DEAD;
    uint32_t entry = 0u;
    bool have = false;
    bool pool_open = true;                                     /* wave-uniform */
    for (;;) {
        const unsigned long long idle_m = rpt_ballot(w.cur == TOP_DEAD);
        const uint32_t n_idle = (uint32_t)__popcll(idle_m);
        if (pool_open && n_idle >= (uint32_t)RPT_TOP_REFILL) {
            uint32_t base = 0u, got = 0u;
            bool finished = false;
            if (lane == 0u) base = wg_pool_take(&pool, global_next, n, SPAN, n_idle, got, finished);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            got = (uint32_t)__builtin_amdgcn_readfirstlane((int)got);
            pool_open = __builtin_amdgcn_readfirstlane((int)finished) == 0;
            if (w.cur == TOP_DEAD) {
                if (have) {
                    q.sh_c[entry].w = w.res.tri == HIT_MISS ? 0.0f : 1.0f;
                    have = false;
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
                if (rank < got && q_filled(q.shadow_cnt, base + rank)) {   /* (a position in the tail of a shard may be empty) */
                    entry = base + rank;
                    const float4 o = q.sh_o[entry], d = q.sh_d[entry];
                    ro = f3(o.x, o.y, o.z); rd = f3(d.x, d.y, d.z);
                    max_t = o.w;
                    have = true;
                    if (fastdiv_ray_ok(sc.fastdiv_ok, ro, rd)) {
                        ird = f3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                        top_walk_begin(w);
                    } else {
                        w.res = top_walk_slow<STACK, true>(sc, ro, rd, max_t, stack);   /* alone; recorded at the next refill */
                    }
                }
            }
            if (got != 0u || !pool_open) continue;
            if (idle_m == ~0ull) { __builtin_amdgcn_s_sleep(8); continue; }   /* another wave is fetching the next span */
        }
        if (idle_m == ~0ull) {
            if (!pool_open) break;                             /* nothing in flight and nothing left to hand out */
            continue;
        }
        top_walk_run<STACK, true, LEAF_LDS>(view, w, ro, rd, ird, max_t, stack, pool_open ? RPT_TOP_TRIPS : 0x7fffffff);
    }
    if (have) q.sh_c[entry].w = w.res.tri == HIT_MISS ? 0.0f : 1.0f;
}

#endif /* RPT_K_TRAVERSE_TOP_H */
