/*
 * rpt_comm.hip — what happens to the accumulators after a sample batch, behind the C ABI:
 *
 *   read-back   rpt_read_accum / rpt_map_accum (≙ output_buffer.read_blocking, reference src/trace.rs:198): the
 *               tile-major accumulator block is un-tiled ON THE DEVICE into a row-major W x H image and leaves the GPU
 *               as ONE DMA into pinned host memory.
 *   gather      the single collective of the multi-GPU path: after a batch every rank's contiguous
 *               block travels to rank 0 with grouped ncclSend / ncclRecv over xGMI (RCCL), on a second HIP stream so
 *               that batch k+1 renders while the blocks of batch k travel; the root un-tiles with a map built once per
 *               configuration.  Nothing here allocates, copies from the host or synchronises per batch.
 *   two drivers one process per GPU (rpt_comm_init: ncclCommInitRank from a caller-distributed unique id — bench.py /
 *               torchrun), or ONE process driving every GPU of the node (rpt_multi_*: ncclCommInitAll) — the shape the
 *               reference's single render thread (src/trace.rs:136-224, src/app.rs:157-164) can call directly.
 *
 * RCCL is resolved with dlopen when a communicator is first requested: single-GPU users never load it.
 */
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <set>

#include "rpt_ctx.h"

namespace {

/* ---- RCCL entry points, resolved lazily ------------------------------------------------------------------ */
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    std::string library;             /* what was loaded: a soname of RCCL, or the RPT_RCCL_LIBRARY override */

    std::mutex mutex;                /* contexts on different threads may ask for their first communicator at the same time */

    bool load() {
        std::lock_guard<std::mutex> lock(mutex);
        if (handle) return true;
        error.clear();
        /* RPT_RCCL_LIBRARY: load THIS library instead of RCCL.  It exists for tests/fake_rccl (N processes on the one GPU of a
           test box, where RCCL refuses a second rank per device); rpt_comm_library() reports it and bench.py refuses to run with it. */
        const char *override_path = getenv("RPT_RCCL_LIBRARY");
        if (override_path && *override_path) {
            handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
            if (!handle) { error = std::string("RPT_RCCL_LIBRARY: cannot load ") + override_path + ": " + dlerror(); return false; }
            library = override_path;
        } else {
            /* by soname: if the process already holds an RCCL (torch ships one) the loader hands back that very copy */
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (handle) { library = name; break; }
            }
        }
        if (!handle) { error = std::string("cannot load RCCL: ") + dlerror(); return false; }
        auto sym = [&](const char *n) { void *p = dlsym(handle, n); if (!p && error.empty()) error = std::string("RCCL lacks ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(sym("ncclCommCount"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!error.empty()) { dlclose(handle); handle = nullptr; library.clear(); return false; }
        return true;
    }
};
RcclApi &rccl() {
    static RcclApi api;
    return api;
}

#define NCCL_TRY(ctx, expr)                                                                        \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            (ctx)->error = std::string(#expr) + ": " + rccl().GetErrorString(r_);                  \
            return RPT_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

static_assert(sizeof(ncclUniqueId) == RPT_COMM_ID_BYTES, "rpt.h: RPT_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

/* scatter of tile-major accumulator blocks into a row-major image; map[i] = y << 16 | x of element i, 0xffffffff = padding */
__global__ __launch_bounds__(RPT_BLOCK) void k_untile(const float4 *blocks, const uint32_t *map, uint32_t n_total, uint32_t width,
                                                      float4 *image) {
    uint32_t i = blockIdx.x * RPT_BLOCK + threadIdx.x;
    if (i >= n_total) return;
    uint32_t pxy = map[i];
    if (pxy == 0xffffffffu) return;
    image[(size_t)(pxy >> 16) * width + (pxy & 0xffffu)] = blocks[i];
}

}  // namespace

struct rpt_comm {
    ncclComm_t comm = nullptr;       /* null: the ranks of an rpt_multi that share ONE device exchange blocks by stream-ordered copies
                                        (RCCL refuses two ranks on a device) — only reachable through RPT_MULTI_ALLOW_SHARED_DEVICE */
    bool owns_comm = false;
    uint32_t rank = 0, world = 1;
    hipStream_t stream = nullptr;    /* the gather runs here and overlaps the next batch on ctx->stream */
    hipEvent_t staged = nullptr;     /* ctx->stream: this batch's accumulators are snapshotted in `send` */
    hipEvent_t sent = nullptr;       /* comm stream: `send` may be overwritten (and, root, the full image is complete) */
    DevBuf<float4> send;             /* stride float4: the collective never touches memory the renderer writes */
    DevBuf<float4> gathered;         /* root: world x stride */
    DevBuf<uint32_t> map;            /* root: destination pixel of every element of `gathered` */
    DevBuf<float4> full_image;       /* root: row-major W x H */
    float *host_full = nullptr;      /* root: pinned staging for rpt_read_gathered */
    size_t host_full_floats = 0;
    uint64_t stride = 0;
    std::vector<uint64_t> sizes;
    uint32_t conf_w = 0, conf_h = 0;
    uint32_t gathered_samples = 0;
    bool started = false;
};

void rpt_image_release(rpt_ctx *c) {
    c->image.release();
    if (c->host_image) (void)hipHostFree(c->host_image);
    c->host_image = nullptr;
    c->host_image_floats = 0;
    c->untile_map.release();
    c->untile_key = 0;
}

void rpt_comm_release(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    if (!cm) return;
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    if (cm->comm && cm->owns_comm) (void)rccl().CommDestroy(cm->comm);
    cm->send.release(); cm->gathered.release(); cm->map.release(); cm->full_image.release();
    if (cm->host_full) (void)hipHostFree(cm->host_full);
    if (cm->staged) (void)hipEventDestroy(cm->staged);
    if (cm->sent) (void)hipEventDestroy(cm->sent);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
    c->comm = nullptr;
}

namespace {

/* the row-major device image + its pinned host twin, sized for the current configuration */
int ensure_image(rpt_ctx *c) {
    const size_t n = (size_t)c->cfg.c.width * c->cfg.c.height;
    if (c->image.n != n) {
        HIP_TRY(c, c->image.alloc(n));
        HIP_TRY(c, hipMemsetAsync(c->image.p, 0, n * sizeof(float4), c->stream));   /* other ranks' pixels stay zero for ever */
    }
    if (c->host_image_floats != n * 4) {
        if (c->host_image) (void)hipHostFree(c->host_image);
        c->host_image = nullptr;
        c->host_image_floats = 0;
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&c->host_image), n * sizeof(float4), hipHostMallocDefault));
        c->host_image_floats = n * 4;
    }
    return RPT_OK;
}

/* un-tile this rank's block into c->image and start its DMA into the pinned buffer; both on the render stream */
int enqueue_readback(rpt_ctx *c) {
    int rc = ensure_image(c);
    if (rc) return rc;
    if (c->n_pixels)
        k_untile<<<(c->n_pixels + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, c->stream>>>(c->accum.p, c->pixel_xy.p, c->n_pixels, c->cfg.c.width, c->image.p);
    HIP_TRY(c, hipMemcpyAsync(c->host_image, c->image.p, c->host_image_floats * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    return RPT_OK;
}

int comm_check_partition(rpt_ctx *c);

/* (re)size the gather buffers for the current configuration: block sizes of every rank, the padded stride, and — root —
 * the destination map, the landing buffer and the full image.  Runs when the configuration changed, never per batch. */
int comm_configure(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    const uint32_t W = c->cfg.c.width, H = c->cfg.c.height;
    if (cm->conf_w == W && cm->conf_h == H && cm->send.p) return RPT_OK;
    HIP_TRY(c, hipStreamSynchronize(cm->stream));
    std::vector<std::vector<uint32_t>> orders(cm->world);
    cm->sizes.assign(cm->world, 0);
    cm->stride = 0;
    for (uint32_t r = 0; r < cm->world; ++r) {
        rpt_build_pixel_order(W, H, r, cm->world, orders[r]);
        cm->sizes[r] = orders[r].size();
        cm->stride = std::max<uint64_t>(cm->stride, orders[r].size());
    }
    int prc = comm_check_partition(c);
    if (prc) return prc;
    cm->started = false;             /* no gathered image exists for this configuration yet (rpt_read_gathered refuses until one does) */
    HIP_TRY(c, cm->send.alloc(std::max<uint64_t>(cm->stride, 1)));
    if (cm->rank == 0) {
        std::vector<uint32_t> map((size_t)cm->world * cm->stride, 0xffffffffu);
        for (uint32_t r = 0; r < cm->world; ++r) std::copy(orders[r].begin(), orders[r].end(), map.begin() + (size_t)r * cm->stride);
        HIP_TRY(c, cm->map.alloc(std::max<size_t>(map.size(), 1)));
        HIP_TRY(c, cm->gathered.alloc(std::max<size_t>(map.size(), 1)));
        HIP_TRY(c, cm->full_image.alloc((size_t)W * H));
        if (!map.empty()) HIP_TRY(c, hipMemcpy(cm->map.p, map.data(), map.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemsetAsync(cm->full_image.p, 0, (size_t)W * H * sizeof(float4), cm->stream));   /* (on the stream that un-tiles into it: rpt_hip.hip, stream discipline) */
        if (cm->host_full_floats != (size_t)W * H * 4) {
            if (cm->host_full) (void)hipHostFree(cm->host_full);
            cm->host_full = nullptr;
            cm->host_full_floats = 0;
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void **>(&cm->host_full), (size_t)W * H * sizeof(float4), hipHostMallocDefault));
            cm->host_full_floats = (size_t)W * H * 4;
        }
    }
    cm->conf_w = W;
    cm->conf_h = H;
    return RPT_OK;
}

/* the context renders the rank the gather's map assumes, at the size the map was built for */
int comm_check_partition(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    if (c->rank != cm->rank || c->world != cm->world || (!cm->sizes.empty() && cm->sizes[cm->rank] != c->n_pixels)) {
        c->error = "gather: partition of the context and of the communicator differ";
        return RPT_EINVAL;
    }
    return RPT_OK;
}

int comm_attach(rpt_ctx *c, ncclComm_t comm, bool owns, uint32_t rank, uint32_t world) {
    rpt_comm_release(c);
    auto *cm = new rpt_comm();
    c->comm = cm;
    cm->comm = comm;
    cm->owns_comm = owns;
    cm->rank = rank;
    cm->world = world;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&cm->staged, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&cm->sent, hipEventDisableTiming));
    return rpt_set_partition(c, rank, world);
}

/* Part 1 of a gather, on the render stream: wait until the previous gather has released `send`, snapshot the
 * accumulators into it, and make the comm stream wait for the snapshot. */
int gather_stage(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    if (!c->has_state || !c->has_config) { c->error = "gather: no config / state"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = comm_configure(c);
    if (rc) return rc;
    /* a caller that re-partitioned a context behind the communicator's back must neither overrun the snapshot buffer nor —
       same block size, other rank — have its block un-tiled through another rank's map */
    rc = comm_check_partition(c);
    if (rc) return rc;
    /* on the render stream: after everything it has enqueued (the batch) and after the previous gather released `send` */
    if (cm->started) HIP_TRY(c, hipStreamWaitEvent(c->stream, cm->sent, 0));
    if (c->n_pixels) HIP_TRY(c, hipMemcpyAsync(cm->send.p, c->accum.p, (size_t)c->n_pixels * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(cm->staged, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(cm->stream, cm->staged, 0));
    cm->gathered_samples = c->samples;
    return RPT_OK;
}

/* Part 2, on the comm stream, inside the caller's ncclGroupStart/End: the point-to-point calls of this rank. */
int gather_exchange(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    HIP_TRY(c, hipSetDevice(c->device));
    if (cm->rank == 0) {
        for (uint32_t r = 1; r < cm->world; ++r)
            if (cm->sizes[r])
                NCCL_TRY(c, rccl().Recv(cm->gathered.p + (size_t)r * cm->stride, cm->sizes[r] * 4, ncclFloat, (int)r, cm->comm, cm->stream));
    } else if (cm->sizes[cm->rank]) {
        NCCL_TRY(c, rccl().Send(cm->send.p, cm->sizes[cm->rank] * 4, ncclFloat, 0, cm->comm, cm->stream));
    }
    return RPT_OK;
}

/* Part 3, on the comm stream: the root un-tiles; every rank marks `send` reusable. */
int gather_finish(rpt_ctx *c) {
    rpt_comm *cm = c->comm;
    HIP_TRY(c, hipSetDevice(c->device));
    if (cm->rank == 0) {
        if (cm->sizes[0]) HIP_TRY(c, hipMemcpyAsync(cm->gathered.p, cm->send.p, cm->sizes[0] * sizeof(float4), hipMemcpyDeviceToDevice, cm->stream));
        const uint32_t n = (uint32_t)((size_t)cm->world * cm->stride);
        if (n) k_untile<<<(n + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, cm->stream>>>(cm->gathered.p, cm->map.p, n, c->cfg.c.width, cm->full_image.p);
    }
    HIP_TRY(c, hipEventRecord(cm->sent, cm->stream));
    cm->started = true;
    return RPT_OK;
}

}  // namespace

extern "C" {

/* ------------------------------------------------------------------------------------------- read-back -- */

int rpt_read_accum(rpt_ctx *c, float *out, uint32_t *out_samples) {
    const float *mapped = nullptr;
    if (!out) return RPT_EINVAL;
    int rc = rpt_map_accum(c, &mapped, out_samples);
    if (rc) return rc;
    memcpy(out, mapped, c->host_image_floats * sizeof(float));
    return RPT_OK;
}

int rpt_map_accum(rpt_ctx *c, const float **out, uint32_t *out_samples) {
    if (!c || !out) return RPT_EINVAL;
    if (!c->has_state) { c->error = "nothing to read: no config"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = enqueue_readback(c);
    if (rc) return rc;
    rc = rpt_wait(c);                    /* one synchronisation; also verifies that asynchronous batches drained */
    if (rc) return rc;
    *out = c->host_image;
    if (out_samples) *out_samples = c->samples;
    return RPT_OK;
}

int rpt_local_block_device_ptr(rpt_ctx *c, void **p) {
    if (!c || !p) return RPT_EINVAL;
    if (!c->has_state) { c->error = "no config"; return RPT_EINVAL; }
    *p = c->accum.p;                     /* no synchronisation: order work after the batch on rpt_stream() */
    return RPT_OK;
}

/* Legacy root-side un-tile for callers that run their own collective (tiles.py over torch.distributed): launch only.
 * The destination map is built when (width, height, world, stride) changes, never per batch. */
int rpt_untile(rpt_ctx *c, const void *dev_blocks, uint64_t block_stride_pixels, void *dev_out_image) {
    if (!c || !dev_blocks || !dev_out_image) return RPT_EINVAL;
    if (!c->has_config) { c->error = "no config"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    const uint64_t key = ((uint64_t)c->cfg.c.width << 48) ^ ((uint64_t)c->cfg.c.height << 32) ^ ((uint64_t)c->world << 24) ^ block_stride_pixels ^ (1ull << 63);
    if (c->untile_key != key) {
        std::vector<uint32_t> all, order;
        for (uint32_t r = 0; r < c->world; ++r) {
            rpt_build_pixel_order(c->cfg.c.width, c->cfg.c.height, r, c->world, order);
            if (block_stride_pixels) {
                if (order.size() > block_stride_pixels) { c->error = "block stride smaller than a rank's block"; return RPT_EINVAL; }
                all.resize((size_t)r * block_stride_pixels, 0xffffffffu);
            }
            all.insert(all.end(), order.begin(), order.end());
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->untile_map.alloc(std::max<size_t>(all.size(), 1)));
        if (!all.empty()) HIP_TRY(c, hipMemcpy(c->untile_map.p, all.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        c->untile_n = (uint32_t)all.size();
        c->untile_key = key;
    }
    if (c->untile_n)
        k_untile<<<(c->untile_n + RPT_BLOCK - 1) / RPT_BLOCK, RPT_BLOCK, 0, c->stream>>>(reinterpret_cast<const float4 *>(dev_blocks), c->untile_map.p,
                                                                                      c->untile_n, c->cfg.c.width, reinterpret_cast<float4 *>(dev_out_image));
    HIP_TRY(c, hipGetLastError());
    return RPT_OK;
}

/* ---------------------------------------------------------------------------- gather, one process per GPU -- */

int rpt_comm_unique_id(uint8_t *id_out) {
    if (!id_out) return RPT_EINVAL;
    if (!rccl().load()) { rpt_create_error() = rccl().error; return RPT_ENODEV; }
    ncclUniqueId id;
    ncclResult_t r = rccl().GetUniqueId(&id);
    if (r != ncclSuccess) { rpt_create_error() = std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r); return RPT_EHIP; }
    memcpy(id_out, &id, sizeof(id));
    return RPT_OK;
}

const char *rpt_comm_library(void) { return rccl().library.c_str(); }

int rpt_comm_init(rpt_ctx *c, const uint8_t *unique_id, uint32_t rank, uint32_t world_size) {
    if (!c || !unique_id) return RPT_EINVAL;
    if (world_size == 0 || rank >= world_size) { c->error = "rank must be < world_size"; return RPT_EINVAL; }
    if (!rccl().load()) { c->error = rccl().error; return RPT_ENODEV; }
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(c, rccl().CommInitRank(&comm, (int)world_size, id, (int)rank));
    int count = 0;
    NCCL_TRY(c, rccl().CommCount(comm, &count));
    if (count != (int)world_size) {
        (void)rccl().CommDestroy(comm);
        c->error = "RCCL communicator has " + std::to_string(count) + " ranks, expected " + std::to_string(world_size);
        return RPT_EHIP;
    }
    return comm_attach(c, comm, true, rank, world_size);
}

/* A communicator of ONE rank without RCCL: the snapshot / second-stream / un-tile / pinned-DMA machinery of the gather for
 * a single GPU, so that the reference's loop (render a batch, read it back: src/trace.rs:182-204) can read batch k while batch
 * k+1 renders:  rpt_render_async(k) ; rpt_gather_async ; rpt_render_async(k+1) ; rpt_read_gathered -> image after batch k. */
int rpt_comm_init_local(rpt_ctx *c) {
    if (!c) return RPT_EINVAL;
    if (c->world != 1u) { c->error = "rpt_comm_init_local: the context is one rank of several (rpt_set_partition); use rpt_comm_init"; return RPT_EINVAL; }
    return comm_attach(c, nullptr, false, 0u, 1u);
}

int rpt_comm_world(rpt_ctx *c, uint32_t *rank, uint32_t *world_size) {
    if (!c || !c->comm) return RPT_EINVAL;
    if (c->comm->comm) {
        int count = 0;
        NCCL_TRY(c, rccl().CommCount(c->comm->comm, &count));      /* what RCCL itself believes */
        if (world_size) *world_size = (uint32_t)count;
    } else if (world_size) {
        *world_size = c->comm->world;
    }
    if (rank) *rank = c->comm->rank;
    return RPT_OK;
}

int rpt_gather_async(rpt_ctx *c) {
    if (!c) return RPT_EINVAL;
    if (!c->comm || (!c->comm->comm && c->comm->world != 1u)) { c->error = "rpt_gather_async: no communicator (rpt_comm_init / rpt_comm_init_local)"; return RPT_EINVAL; }
    int rc = gather_stage(c);
    if (rc) return rc;
    if (c->comm->world > 1u) {                                   /* (one rank: nothing travels) */
        NCCL_TRY(c, rccl().GroupStart());
        rc = gather_exchange(c);
        ncclResult_t ge = rccl().GroupEnd();
        if (rc) return rc;
        NCCL_TRY(c, ge);
    }
    return gather_finish(c);
}

/* The library's own point-to-point calls, executed: through the RcclApi function table, on the communicator's comm stream, ordered by the
 * `staged` / `sent` events exactly as a gather orders them, inside ONE ncclGroupStart / ncclGroupEnd — this rank posts ncclRecv from
 * rank - 1 and ncclSend to rank + 1 (a ring; with one rank: a send to itself and a receive from itself, which RCCL accepts inside a
 * group).  n_floats of a known pattern travel as ncclFloat and are compared word for word on the host.  A wrong argument order, count
 * unit or dtype enum in gather_exchange's calls shows here, on one GPU, before an 8-GPU run. */
int rpt_debug_comm_selftest(rpt_ctx *c, uint32_t n_floats, uint64_t *mismatches_out) {
    if (!c || !mismatches_out || n_floats == 0) return RPT_EINVAL;
    rpt_comm *cm = c->comm;
    if (!cm || !cm->comm) { c->error = "rpt_debug_comm_selftest: needs an RCCL communicator (rpt_comm_init)"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    const int next = (int)((cm->rank + 1u) % cm->world), prev = (int)((cm->rank + cm->world - 1u) % cm->world);
    std::vector<float> host(n_floats), back(n_floats, -1.0f);
    /* what `prev` sends is a function of ITS rank: every rank can check what it received */
    auto pattern = [](uint32_t rank, uint32_t i) { return (float)(int32_t)((i * 2654435761u) ^ (rank * 0x9e3779b9u)) * (1.0f / 65536.0f); };
    for (uint32_t i = 0; i < n_floats; ++i) host[i] = pattern(cm->rank, i);
    DevBuf<float> src, dst;
    HIP_TRY(c, src.alloc(n_floats));
    if (dst.alloc(n_floats) != hipSuccess) { src.release(); c->error = "rpt_debug_comm_selftest: out of device memory"; return RPT_ENOMEM; }
    int rc = RPT_OK;
    auto body = [&]() -> int {
        /* render stream: the payload is staged (as gather_stage's snapshot is), the comm stream waits for it */
        if (cm->started) HIP_TRY(c, hipStreamWaitEvent(c->stream, cm->sent, 0));
        HIP_TRY(c, hipMemcpyAsync(src.p, host.data(), (size_t)n_floats * sizeof(float), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(dst.p, 0xff, (size_t)n_floats * sizeof(float), c->stream));
        HIP_TRY(c, hipEventRecord(cm->staged, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(cm->stream, cm->staged, 0));
        NCCL_TRY(c, rccl().GroupStart());
        ncclResult_t r1 = rccl().Send(src.p, n_floats, ncclFloat, next, cm->comm, cm->stream);
        ncclResult_t r2 = rccl().Recv(dst.p, n_floats, ncclFloat, prev, cm->comm, cm->stream);
        ncclResult_t ge = rccl().GroupEnd();
        NCCL_TRY(c, r1);
        NCCL_TRY(c, r2);
        NCCL_TRY(c, ge);
        HIP_TRY(c, hipEventRecord(cm->sent, cm->stream));
        HIP_TRY(c, hipMemcpyAsync(back.data(), dst.p, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToHost, cm->stream));
        HIP_TRY(c, hipStreamSynchronize(cm->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return RPT_OK;
    };
    rc = body();
    src.release();
    dst.release();
    if (rc) return rc;
    uint64_t bad = 0;
    for (uint32_t i = 0; i < n_floats; ++i) {
        const float want = pattern((uint32_t)prev, i);
        bad += memcmp(&want, &back[i], sizeof(float)) != 0;
    }
    *mismatches_out = bad;
    return RPT_OK;
}

int rpt_gather_wait(rpt_ctx *c) {
    if (!c) return RPT_EINVAL;
    if (!c->comm) { c->error = "no communicator"; return RPT_EINVAL; }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->comm->stream));
    HIP_TRY(c, hipGetLastError());
    return RPT_OK;
}

int rpt_gathered_device_ptr(rpt_ctx *c, void **p) {
    if (!c || !p) return RPT_EINVAL;
    if (!c->comm || c->comm->rank != 0 || !c->comm->full_image.p) { c->error = "gathered image exists on rank 0 after the first gather"; return RPT_EINVAL; }
    if (!c->has_config || c->comm->conf_w != c->cfg.c.width || c->comm->conf_h != c->cfg.c.height) {
        c->error = "rpt_gathered_device_ptr: the configuration was resized since the last gather";
        return RPT_EINVAL;
    }
    *p = c->comm->full_image.p;
    return RPT_OK;
}

int rpt_read_gathered(rpt_ctx *c, float *out, uint32_t *out_samples) {
    if (!c || !out) return RPT_EINVAL;
    rpt_comm *cm = c->comm;
    if (!cm || cm->rank != 0 || !cm->full_image.p || !cm->started) { c->error = "gathered image exists on rank 0 after the first gather"; return RPT_EINVAL; }
    /* `out` is sized by the caller for the CURRENT configuration; the gathered image has the size of the configuration it was
       gathered under — after a resize there is no image to hand out until the next gather */
    if (!c->has_config || cm->conf_w != c->cfg.c.width || cm->conf_h != c->cfg.c.height) {
        c->error = "rpt_read_gathered: the configuration was resized since the last gather (" + std::to_string(cm->conf_w) + "x" + std::to_string(cm->conf_h) +
                   " gathered): call rpt_gather_async first";
        return RPT_EINVAL;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(cm->host_full, cm->full_image.p, cm->host_full_floats * sizeof(float), hipMemcpyDeviceToHost, cm->stream));
    HIP_TRY(c, hipStreamSynchronize(cm->stream));
    memcpy(out, cm->host_full, cm->host_full_floats * sizeof(float));
    if (out_samples) *out_samples = cm->gathered_samples;
    return RPT_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------- one process, every GPU of the node -- */

struct rpt_multi {
    std::vector<rpt_ctx *> ctx;
    bool shared_device = false;      /* ranks share a device: copy transport (test aid) */
    bool gathered = false;           /* rank 0 holds the image of everything rendered so far */
    std::string error;
};

namespace {

int multi_fail(rpt_multi *m, rpt_ctx *c, int rc) {
    m->error = c->error;
    return rc;
}

int multi_gather(rpt_multi *m) {
    int rc;
    for (rpt_ctx *c : m->ctx)
        if ((rc = gather_stage(c))) return multi_fail(m, c, rc);
    if (!m->shared_device) {
        rpt_ctx *c0 = m->ctx[0];
        if (rccl().GroupStart() != ncclSuccess) { m->error = "ncclGroupStart failed"; return RPT_EHIP; }
        rc = RPT_OK;
        rpt_ctx *failed = nullptr;
        for (rpt_ctx *c : m->ctx)
            if (!rc && (rc = gather_exchange(c))) failed = c;
        ncclResult_t ge = rccl().GroupEnd();
        if (rc) return multi_fail(m, failed, rc);
        if (ge != ncclSuccess) { m->error = std::string("ncclGroupEnd: ") + rccl().GetErrorString(ge); return RPT_EHIP; }
        (void)c0;
    } else {
        /* ranks on one device: the root's comm stream copies every peer's snapshot once that peer has staged it */
        rpt_ctx *root = m->ctx[0];
        rpt_comm *rcm = root->comm;
        for (size_t r = 1; r < m->ctx.size(); ++r) {
            rpt_comm *pcm = m->ctx[r]->comm;
            if (hipStreamWaitEvent(rcm->stream, pcm->staged, 0) != hipSuccess) { m->error = "hipStreamWaitEvent failed"; return RPT_EHIP; }
            if (pcm->sizes[r] && hipMemcpyAsync(rcm->gathered.p + r * rcm->stride, pcm->send.p, pcm->sizes[r] * sizeof(float4), hipMemcpyDeviceToDevice,
                                                rcm->stream) != hipSuccess) { m->error = "peer copy failed"; return RPT_EHIP; }
        }
    }
    /* the root finishes first: in copy mode the peers' `sent` must follow the root's copies */
    if ((rc = gather_finish(m->ctx[0]))) return multi_fail(m, m->ctx[0], rc);
    for (size_t r = 1; r < m->ctx.size(); ++r) {
        rpt_ctx *c = m->ctx[r];
        if (m->shared_device && hipStreamWaitEvent(c->comm->stream, m->ctx[0]->comm->sent, 0) != hipSuccess) { m->error = "hipStreamWaitEvent failed"; return RPT_EHIP; }
        if ((rc = gather_finish(c))) return multi_fail(m, c, rc);
    }
    m->gathered = true;
    return RPT_OK;
}

}  // namespace

extern "C" {

const char *rpt_multi_last_error(rpt_multi *m) { return m ? m->error.c_str() : rpt_create_error().c_str(); }

void rpt_multi_destroy(rpt_multi *m) {
    if (!m) return;
    for (rpt_ctx *c : m->ctx) rpt_destroy(c);
    delete m;
}

int rpt_multi_create(const int *device_ids, int n_devices, uint32_t flags, rpt_multi **out) {
    if (!out || !device_ids || n_devices < 1 || n_devices > 64) { rpt_create_error() = "rpt_multi_create: bad arguments"; return RPT_EINVAL; }
    *out = nullptr;
    std::set<int> distinct(device_ids, device_ids + n_devices);
    const bool shared = (int)distinct.size() != n_devices;
    if (shared && !(flags & RPT_MULTI_ALLOW_SHARED_DEVICE)) {
        rpt_create_error() = "rpt_multi_create: a device is listed twice (RCCL needs one device per rank; RPT_MULTI_ALLOW_SHARED_DEVICE is the one-GPU test aid)";
        return RPT_EINVAL;
    }
    auto *m = new rpt_multi();
    m->shared_device = shared;
    for (int r = 0; r < n_devices; ++r) {
        rpt_ctx *c = nullptr;
        int rc = rpt_create(device_ids[r], &c);
        if (rc) { rpt_multi_destroy(m); return rc; }
        m->ctx.push_back(c);
    }
    std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
    if (!shared) {
        if (!rccl().load()) { rpt_create_error() = rccl().error; rpt_multi_destroy(m); return RPT_ENODEV; }
        ncclResult_t r = rccl().CommInitAll(comms.data(), n_devices, device_ids);
        if (r != ncclSuccess) { rpt_create_error() = std::string("ncclCommInitAll: ") + rccl().GetErrorString(r); rpt_multi_destroy(m); return RPT_EHIP; }
    }
    for (int r = 0; r < n_devices; ++r) {
        int rc = comm_attach(m->ctx[(size_t)r], comms[(size_t)r], !shared, (uint32_t)r, (uint32_t)n_devices);
        if (rc) {
            rpt_create_error() = m->ctx[(size_t)r]->error;
            /* communicators of the ranks after r belong to no context yet: rpt_multi_destroy cannot reach them */
            if (!shared)
                for (int k = r + 1; k < n_devices; ++k)
                    if (comms[(size_t)k]) (void)rccl().CommDestroy(comms[(size_t)k]);
            if (!shared && !m->ctx[(size_t)r]->comm && comms[(size_t)r]) (void)rccl().CommDestroy(comms[(size_t)r]);
            rpt_multi_destroy(m);
            return rc;
        }
    }
    *out = m;
    return RPT_OK;
}

int rpt_multi_size(rpt_multi *m) { return m ? (int)m->ctx.size() : 0; }
rpt_ctx *rpt_multi_ctx(rpt_multi *m, int rank) { return (m && rank >= 0 && rank < (int)m->ctx.size()) ? m->ctx[(size_t)rank] : nullptr; }

int rpt_multi_upload_scene(rpt_multi *m, const rpt_per_vertex_data *pv, size_t nv, const rpt_triangle *idx, size_t nt, const rpt_bvh_node *nodes,
                           size_t nn, const rpt_material_data *mats, size_t nm, const rpt_light_pick_entry *lp, size_t nlp,
                           const uint8_t *atlas, uint32_t aw, uint32_t ah, const float *skybox, uint32_t sw, uint32_t sh) {
    if (!m) return RPT_EINVAL;
    for (rpt_ctx *c : m->ctx) {       /* the scene is small and read-only: replicated */
        int rc = rpt_upload_scene(c, pv, nv, idx, nt, nodes, nn, mats, nm, lp, nlp, atlas, aw, ah, skybox, sw, sh);
        if (rc) return multi_fail(m, c, rc);
    }
    return RPT_OK;
}

int rpt_multi_set_config(rpt_multi *m, const rpt_tracing_config *cfg) {
    if (!m || !cfg) return RPT_EINVAL;
    rpt_ctx *root = m->ctx[0];
    if (!root->has_config || root->cfg.c.width != cfg->width || root->cfg.c.height != cfg->height)
        m->gathered = false;         /* a resize: the root's gathered image belongs to the old size (rpt_multi_read_accum gathers afresh) */
    for (rpt_ctx *c : m->ctx) {
        int rc = rpt_set_config(c, cfg);
        if (rc) return multi_fail(m, c, rc);
    }
    return RPT_OK;
}

int rpt_multi_reset(rpt_multi *m, const rpt_rng_state *seed, const float *accum_init, uint32_t samples_init) {
    if (!m) return RPT_EINVAL;
    for (rpt_ctx *c : m->ctx) {
        int rc = rpt_reset(c, seed, accum_init, samples_init);
        if (rc) return multi_fail(m, c, rc);
    }
    m->gathered = false;
    return RPT_OK;
}

/* One sample batch on every GPU + the single gather of the batch; returns once everything is enqueued (when the
 * iteration count of the batch is known, see rpt_render_async). */
int rpt_multi_render(rpt_multi *m, uint32_t n_samples) {
    if (!m) return RPT_EINVAL;
    for (rpt_ctx *c : m->ctx) {
        int rc = rpt_render_async(c, n_samples);
        if (rc) return multi_fail(m, c, rc);
    }
    return multi_gather(m);
}

int rpt_multi_wait(rpt_multi *m) {
    if (!m) return RPT_EINVAL;
    for (rpt_ctx *c : m->ctx) {
        int rc = rpt_wait(c);
        if (!rc) rc = rpt_gather_wait(c);
        if (rc) return multi_fail(m, c, rc);
    }
    return RPT_OK;
}

int rpt_multi_read_accum(rpt_multi *m, float *out, uint32_t *out_samples) {
    if (!m || !out) return RPT_EINVAL;
    int rc = rpt_multi_wait(m);
    if (rc) return rc;
    rpt_ctx *root = m->ctx[0];
    if (!m->gathered) {              /* nothing rendered since the last reset: gather what there is (zeros / the resume image) */
        if ((rc = multi_gather(m))) return rc;
        if ((rc = rpt_multi_wait(m))) return rc;
    }
    rc = rpt_read_gathered(root, out, out_samples);
    return rc ? multi_fail(m, root, rc) : RPT_OK;
}

int rpt_multi_get_stats(rpt_multi *m, rpt_stats *out) {
    if (!m || !out) return RPT_EINVAL;
    rpt_stats sum{};
    for (rpt_ctx *c : m->ctx) {
        rpt_stats s;
        int rc = rpt_get_stats(c, &s);
        if (rc) return multi_fail(m, c, rc);
        sum.samples += s.samples; sum.extension_rays += s.extension_rays; sum.shadow_rays += s.shadow_rays; sum.shadow_rays_elided += s.shadow_rays_elided; sum.sky_evals += s.sky_evals;
        sum.light_index_clamped += s.light_index_clamped;
        sum.iterations = std::max(sum.iterations, s.iterations);
        sum.render_ms = std::max(sum.render_ms, s.render_ms);
        for (int k = 0; k < 8; ++k) { sum.kernel_ms[k] = std::max(sum.kernel_ms[k], s.kernel_ms[k]); sum.kernel_launches[k] = std::max(sum.kernel_launches[k], s.kernel_launches[k]); }
    }
    *out = sum;
    return RPT_OK;
}

}  // extern "C"
