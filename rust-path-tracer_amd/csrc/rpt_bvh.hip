/*
 * rpt_bvh.hip — rpt_bvh_build_gpu: the host side of the device BVH build (kernels: k_bvh_build.h).  Its own translation unit: nothing here touches a context.
 * Reference: BVHBuilder::build (src/bvh.rs:59-324); SURVEY.md section 8(f) N1.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "rpt_ctx.h"
#include "k_bvh_build.h"

extern "C" {

/* BVHBuilder::new(vertices, indices).sah_samples(n).build() (src/bvh.rs:59-324) on the device: same node pool, same
 * triangle order as the sequential build (k_bvh_build.h).  Host pointers in and out, like rpt_bvh_build of
 * rpt_host.h; needs no context. */
int rpt_bvh_build_gpu(int device_id, const float *vertices_xyzw, size_t n_vertices, rpt_triangle *triangles, size_t n_triangles,
                      uint32_t sah_samples, rpt_bvh_node *nodes_out, size_t nodes_capacity, size_t *n_nodes_out, double *device_ms_out) {
    if (!vertices_xyzw || !triangles || !nodes_out || !n_nodes_out || n_triangles == 0 || n_vertices == 0) {
        rpt_create_error() = "rpt_bvh_build_gpu: null or empty argument";
        return RPT_EINVAL;
    }
    SectionTimer sections("rpt_bvh_build_gpu");
    if (sah_samples < 2) sah_samples = 2;
    if (sah_samples > BVB_MAX_BINS) { rpt_create_error() = "rpt_bvh_build_gpu: at most 128 SAH bins"; return RPT_EINVAL; }
    if (n_triangles >= (1u << 28)) { rpt_create_error() = "rpt_bvh_build_gpu: too many triangles"; return RPT_EINVAL; }
    if (n_vertices > 0xffffffffull) { rpt_create_error() = "rpt_bvh_build_gpu: too many vertices"; return RPT_EINVAL; }
    if (nodes_capacity < 2 * n_triangles - 1) { rpt_create_error() = "rpt_bvh_build_gpu: node buffer needs 2N-1 entries"; return RPT_EINVAL; }
    /* (vertex indices in range, no NaN coordinate: checked by k_bvb_init where the triangles are gathered anyway — two host loops over the scene were 1.8 ms of a
     *  1 M-triangle build.  The ordered 64-bit keys that reproduce the builder's f32::min / max folds (k_bvh_build.h) have no place for a NaN, which those folds
     *  SKIP (src/bvh.rs via f32::min): with a NaN coordinate this build and the sequential one part ways (found by tools/bvh_nan_probe.py; infinities, denormals
     *  and coincident points are fine).  Said, not built around: such a scene is refused.) */
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0) { rpt_create_error() = "no HIP device"; return RPT_ENODEV; }
    if (device_id < 0 || device_id >= n_dev) { rpt_create_error() = "device id out of range"; return RPT_EINVAL; }
#define BVB_TRY(x)                                                                                      \
    do {                                                                                                \
        hipError_t e_ = (x);                                                                            \
        if (e_ != hipSuccess) {                                                                         \
            rpt_create_error() = std::string("rpt_bvh_build_gpu: ") + hipGetErrorString(e_);                \
            goto fail;                                                                                  \
        }                                                                                               \
    } while (0)
    DevBuf<float4> d_verts;
    DevBuf<BvbRec> d_recs;
    DevBuf<uint4> d_tris;
    DevBuf<uint32_t> d_order, d_order_tmp, d_tmp_a, d_tmp_b, d_count;
    DevBuf<uint32_t> d_lpre;
    DevBuf<BvbNode> d_nodes;
    DevBuf<BvbTeamScratch> d_scratch;
    DevBuf<BvbTeamRef> d_team_refs;
    DevBuf<uint16_t> d_block_team;
    DevBuf<uint32_t> d_inner, d_rank, d_oidx;
    DevBuf<rpt_bvh_node> d_out;
    std::vector<std::pair<uint32_t, uint32_t>> levels;          /* build-order id ranges, root level first */
    const bool use_teams = true;
    uint32_t team_min = BVB_TEAM_MIN_COUNT;           /* RPT_BVH_TEAM_MIN: test aid, lets small nodes take the team path */
    if (const int forced = rpt_read_knobs().bvh_team_min) team_min = (uint32_t)forced;
    const uint32_t team_chunk = BVB_TEAM_CHUNK;
    std::vector<BvbNode> bn;
    std::vector<uint32_t> order;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int fail_code = RPT_EHIP;
    sections.mark("validate");
    const uint32_t nt = (uint32_t)n_triangles;
    {
        BVB_TRY(hipSetDevice(device_id));
        BVB_TRY(d_verts.alloc(n_vertices));
        BVB_TRY(d_tris.alloc(nt));
        BVB_TRY(d_recs.alloc(nt));
        BVB_TRY(d_order.alloc(nt));
        BVB_TRY(d_order_tmp.alloc(nt));
        BVB_TRY(d_tmp_a.alloc(nt));
        BVB_TRY(d_tmp_b.alloc(nt));
        BVB_TRY(d_lpre.alloc(nt));
        BVB_TRY(d_count.alloc(3));                       /* BvbArgs::node_count */
        BVB_TRY(d_nodes.alloc(2 * (size_t)nt - 1));
        BVB_TRY(d_scratch.alloc(BVB_MAX_TEAMS));
        BVB_TRY(d_team_refs.alloc(BVB_MAX_TEAMS));
        BVB_TRY(d_block_team.alloc(BVB_MAX_TEAMS * BVB_TEAM));
        BVB_TRY(hipMemcpy(d_verts.p, vertices_xyzw, n_vertices * sizeof(float4), hipMemcpyHostToDevice));
        BVB_TRY(hipMemcpy(d_tris.p, triangles, nt * sizeof(uint4), hipMemcpyHostToDevice));
        BvbNode root{};
        root.first = 0; root.count = nt; root.left = BVB_NONE;
        BVB_TRY(hipMemcpy(d_nodes.p, &root, sizeof(root), hipMemcpyHostToDevice));
        const uint32_t count_init[3] = {1u, 0u, 0u};
        BVB_TRY(hipMemcpy(d_count.p, count_init, sizeof count_init, hipMemcpyHostToDevice));
        BVB_TRY(hipEventCreate(&ev0));
        BVB_TRY(hipEventCreate(&ev1));
        sections.mark("alloc_h2d");
        BvbArgs a{d_verts.p, d_tris.p, d_recs.p, d_order.p, d_order_tmp.p, d_tmp_a.p, d_tmp_b.p, d_lpre.p, d_nodes.p, d_count.p, nt, sah_samples, (uint32_t)n_vertices};
        BVB_TRY(hipEventRecord(ev0, nullptr));
        k_bvb_init<<<(nt + BVB_THREADS - 1) / BVB_THREADS, BVB_THREADS>>>(a);
        /* a team's workgroups meet at counter barriers (bvb_team_sync): every workgroup of the launch must be resident at
         * once, so the teams of a launch are capped by what the device can hold (a CU-masked or partitioned device holds fewer;
         * with room for none the big nodes simply take the one-workgroup path) */
        uint32_t resident = 0u;
        {
            int per_cu = 0;
            hipDeviceProp_t prop;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bvb_team, BVB_TEAM_THREADS, 0) == hipSuccess &&
                hipGetDeviceProperties(&prop, device_id) == hipSuccess && per_cu > 0 && prop.multiProcessorCount > 0) {
                /* one block per CU less than the API says: it over-reports by one at some register counts (MI355X_MICROARCH.md) */
                resident = std::min<uint32_t>((uint32_t)std::max(0, per_cu - 1) * (uint32_t)prop.multiProcessorCount, BVB_MAX_TEAMS * BVB_TEAM);
            }
        }
        uint32_t begin = 0, end = 1, level_max = nt;        /* level_max: the largest node of the level about to be built */
        std::vector<BvbNode> level_nodes;
        std::vector<BvbTeamRef> team_refs;
        std::vector<uint16_t> block_team;
        while (begin < end) {                               /* one launch per tree level */
            bool teams_here = false;
            /* the big nodes of this level (if any) are split by teams of workgroups first: one workgroup per BVB_TEAM_CHUNK triangles */
            if (use_teams && resident >= 2u && end - begin <= 4096u && level_max >= team_min) {
                level_nodes.resize(end - begin);
                BVB_TRY(hipMemcpy(level_nodes.data(), d_nodes.p + begin, (size_t)(end - begin) * sizeof(BvbNode), hipMemcpyDeviceToHost));
                team_refs.clear();
                block_team.clear();
                for (uint32_t k = 0; k < end - begin && team_refs.size() < BVB_MAX_TEAMS; ++k) {
                    if (level_nodes[k].count < team_min) continue;
                    uint32_t size = 2u;
                    while (size < BVB_TEAM && (size_t)size * team_chunk < level_nodes[k].count) size *= 2u;
                    while (size > 2u && block_team.size() + size > resident) size /= 2u;
                    if (block_team.size() + size > resident) break;
                    team_refs.push_back(BvbTeamRef{begin + k, size, (uint32_t)block_team.size()});
                    block_team.insert(block_team.end(), size, (uint16_t)(team_refs.size() - 1));
                }
                if (!team_refs.empty()) {
                    BVB_TRY(hipMemcpy(d_team_refs.p, team_refs.data(), team_refs.size() * sizeof(BvbTeamRef), hipMemcpyHostToDevice));
                    BVB_TRY(hipMemcpy(d_block_team.p, block_team.data(), block_team.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                    k_bvb_team_init<<<(unsigned)team_refs.size(), BVB_TEAM_THREADS>>>(d_scratch.p, d_team_refs.p);
                    k_bvb_team<<<(unsigned)block_team.size(), BVB_TEAM_THREADS>>>(a, d_scratch.p, d_block_team.p);
                    teams_here = true;                      /* (their children are not in level_max: the next level keeps the wide workgroups) */
                }
            }
            if (end - begin < 64u || level_max >= BVB_WIDE_MIN_COUNT) k_bvb_level<1024><<<end - begin, 1024>>>(a, begin, 0u);    /* few nodes, or big ones */
            else if (level_max <= 256u) {
                /* small nodes: those of up to 8 triangles eight to a wave, the others one wave each — out of registers up to 64 triangles
                 * (measured: 64 / 256 / 1024 as the limit of the one-wave kernel) */
                k_bvb_tiny<<<(end - begin + 7u) / 8u, 64>>>(a, begin, end);
                if (level_max > BVB_TINY) {
                    if (level_max <= 64u) k_bvb_small<<<end - begin, 64>>>(a, begin, BVB_TINY);
                    else k_bvb_level<64><<<end - begin, 64>>>(a, begin, BVB_TINY);
                }
            } else k_bvb_level<BVB_THREADS><<<end - begin, BVB_THREADS>>>(a, begin, 0u);
            k_bvb_children<<<(end - begin + 1023u) / 1024u, 1024>>>(a, begin, end);
            uint32_t total[3] = {0u, 0u, 0u};
            BVB_TRY(hipMemcpy(total, d_count.p, sizeof total, hipMemcpyDeviceToHost));
            if (total[2] != 0u) {
                if (total[2] & 2u) rpt_create_error() = "rpt_bvh_build_gpu: vertex index out of range";
                else if (total[2] & 4u) rpt_create_error() = "rpt_bvh_build_gpu: a vertex coordinate is NaN — such a scene must be built by the host builder (rpt_bvh_build)";
                else rpt_create_error() = "rpt_bvh_build_gpu: internal error, a level's largest node was misjudged";
                fail_code = (total[2] & 6u) ? RPT_ESCENE : RPT_EHIP;
                goto fail;
            }
            BVB_TRY(hipMemsetAsync(d_count.p + 1, 0, 4, nullptr));
            level_max = teams_here ? nt : total[1];
            levels.push_back({begin, end});
            begin = end;
            end = total[0];
        }
        BVB_TRY(hipEventRecord(ev1, nullptr));
        BVB_TRY(hipEventSynchronize(ev1));
        float ms = 0.0f;
        BVB_TRY(hipEventElapsedTime(&ms, ev0, ev1));
        if (device_ms_out) *device_ms_out = ms;
        sections.mark("device_build");
        /* renumber to the order in which the reference splits nodes (bvh.rs:296-320), on the device: k_bvh_build.h, k_bvb_inner_count / k_bvb_place */
        BVB_TRY(d_inner.alloc(end)); BVB_TRY(d_rank.alloc(end)); BVB_TRY(d_oidx.alloc(end)); BVB_TRY(d_out.alloc(end));
        BVB_TRY(hipMemset(d_rank.p, 0, 4));
        BVB_TRY(hipMemset(d_oidx.p, 0, 4));
        for (size_t l = levels.size(); l-- > 0;) {
            const uint32_t lb = levels[l].first, le = levels[l].second;
            k_bvb_inner_count<<<(le - lb + BVB_THREADS - 1) / BVB_THREADS, BVB_THREADS>>>(d_nodes.p, d_inner.p, lb, le);
        }
        for (size_t l = 0; l < levels.size(); ++l) {
            const uint32_t lb = levels[l].first, le = levels[l].second;
            k_bvb_place<<<(le - lb + BVB_THREADS - 1) / BVB_THREADS, BVB_THREADS>>>(d_nodes.p, d_inner.p, d_rank.p, d_oidx.p, d_out.p, lb, le);
        }
        sections.mark("renumber_device");
        order.resize(nt);
        BVB_TRY(hipMemcpy(nodes_out, d_out.p, (size_t)end * sizeof(rpt_bvh_node), hipMemcpyDeviceToHost));
        BVB_TRY(hipMemcpy(order.data(), d_order.p, (size_t)nt * 4, hipMemcpyDeviceToHost));
        *n_nodes_out = end;
    }
    sections.mark("d2h");
    {
        std::vector<rpt_triangle> src(triangles, triangles + nt);
        for (uint32_t i = 0; i < nt; ++i) triangles[i] = src[order[i]];
        sections.mark("reorder_triangles");
    }
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    d_verts.release(); d_tris.release(); d_recs.release(); d_order.release(); d_order_tmp.release();
    d_tmp_a.release(); d_tmp_b.release(); d_lpre.release(); d_count.release(); d_nodes.release(); d_scratch.release(); d_team_refs.release(); d_block_team.release();
    d_inner.release(); d_rank.release(); d_oidx.release(); d_out.release();
    return RPT_OK;
fail:
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    d_verts.release(); d_tris.release(); d_recs.release(); d_order.release(); d_order_tmp.release();
    d_tmp_a.release(); d_tmp_b.release(); d_lpre.release(); d_count.release(); d_nodes.release(); d_scratch.release(); d_team_refs.release(); d_block_team.release();
    d_inner.release(); d_rank.release(); d_oidx.release(); d_out.release();
    return fail_code;
#undef BVB_TRY
}

}  // extern "C"
