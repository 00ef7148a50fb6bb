/*
 * host_api.cpp — C entry points of librpt_host.so: World loading, TracingState,
 * setup_trace and the trace_gpu render loop driven through the rpt.h C ABI.
 *
 * Mirrors (reference): src/trace.rs:40-92 (TracingState), :136-224 (trace_gpu),
 * :331-344 (setup_trace); src/asset.rs:55-224 (World::from_path);
 * shared_structs/src/lib.rs:27-42 (TracingConfig::default).
 */
#include <dlfcn.h>

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>
#include <mutex>

#include "../../../include/rpt/rpt.h"
#include "host_internal.h"

namespace rpth {
static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }
}  // namespace rpth

using namespace rpth;

struct rpt_tracing_state {
    std::mutex lock;                       /* RwLock<..> stand-in for framebuffer + config */
    std::vector<float> framebuffer;        /* W*H*3 mean RGB */
    rpt_tracing_config config;
    std::atomic<bool> running{false};
    std::atomic<uint32_t> samples{0};
    std::atomic<bool> denoise{false};      /* kept for interface parity; OIDN is out of scope */
    std::atomic<uint32_t> sync_rate{32};
    std::atomic<bool> use_blue_noise{true};
    std::atomic<bool> interacting{false};
    std::atomic<bool> dirty{false};
    uint32_t target_samples = 0;           /* setup_trace: stop after exactly this many (0 = run until !running) */
    std::atomic<bool> overlap{true};       /* rpt_trace_gpu reads batch k back while batch k+1 renders (rpt_comm_init_local): the default */
};

extern "C" {

const char *rpt_host_last_error(void) { return g_error.c_str(); }

void rpt_tracing_config_default(rpt_tracing_config *c) {
    memset(c, 0, sizeof(*c));
    c->cam_position[0] = 0.0f; c->cam_position[1] = 1.0f; c->cam_position[2] = -5.0f; c->cam_position[3] = 0.0f;
    c->width = 1280; c->height = 720;
    c->min_bounces = 3; c->max_bounces = 4;
    /* Vec3::new(0.5, 1.3, 1.0).normalize().extend(15.0) : v * (1 / sqrt(dot)) */
    float x = 0.5f, y = 1.3f, z = 1.0f;
    float inv = 1.0f / std::sqrt((x * x) + (y * y) + (z * z));
    c->sun_direction[0] = x * inv; c->sun_direction[1] = y * inv; c->sun_direction[2] = z * inv; c->sun_direction[3] = 15.0f;
    c->nee = 0; c->has_skybox = 0;
    c->specular_weight_clamp[0] = 0.1f; c->specular_weight_clamp[1] = 0.9f;
}

/* ------------------------------------------------------------- World ---- */
int rpt_world_load(const char *path, rpt_world **out) { return rpt_world_load_ex(path, 0u, out); }

void rpt_host_free(void *p) { free(p); }

int rpt_skybox_load(const char *path, float **rgba_out, uint32_t *width, uint32_t *height) {
    if (!path || !rgba_out || !width || !height) { set_error("null argument"); return RPT_EINVAL; }
    try {
        std::vector<float> rgba;
        if (!load_skybox_file(path, rgba, *width, *height)) return RPT_HOST_ELOAD;
        float *p = (float *)malloc(rgba.size() * sizeof(float));
        if (!p) { set_error("out of memory"); return RPT_ENOMEM; }
        memcpy(p, rgba.data(), rgba.size() * sizeof(float));
        *rgba_out = p;
    } catch (const std::exception &e) {
        set_error(std::string("skybox load failed: ") + e.what());
        return RPT_HOST_ELOAD;
    }
    return 0;
}

int rpt_world_load_ex(const char *path, uint32_t flags, rpt_world **out) {
    if (!path || !out) { set_error("null argument"); return RPT_EINVAL; }
    /* nothing may unwind across the C ABI: a hostile file can still ask for more memory than there is */
    rpt_world *w = nullptr;
    try {
        w = new rpt_world();
        const size_t len = strlen(path);
        const bool is_obj = len >= 4 && (!strcmp(path + len - 4, ".obj") || !strcmp(path + len - 4, ".OBJ"));
        if (!(is_obj ? load_obj(path, w->w) : load_glb(path, w->w, flags))) { delete w; return RPT_HOST_ELOAD; }
    } catch (const std::bad_alloc &) {
        delete w;
        set_error("out of memory while loading the scene");
        return RPT_ENOMEM;
    } catch (const std::exception &e) {
        delete w;
        set_error(std::string("scene load failed: ") + e.what());
        return RPT_HOST_ELOAD;
    }
    *out = w;
    return 0;
}

int rpt_world_from_buffers(const float *vertices_xyz, const float *normals_xyz, const float *uvs, size_t n_vertices,
                           const uint32_t *tris, size_t n_triangles, const rpt_material_data *materials,
                           size_t n_materials, rpt_world **out) {
    if (!vertices_xyz || !tris || !materials || !out || !n_vertices || !n_triangles || !n_materials) {
        set_error("null/empty argument");
        return RPT_EINVAL;
    }
    auto *w = new rpt_world();
    std::vector<Vec4f> vertices(n_vertices), normals, tangents;
    std::vector<float> uv;
    for (size_t i = 0; i < n_vertices; ++i) vertices[i] = Vec4f{vertices_xyz[3 * i], vertices_xyz[3 * i + 1], vertices_xyz[3 * i + 2], 1.0f};
    if (normals_xyz) {
        normals.resize(n_vertices);
        for (size_t i = 0; i < n_vertices; ++i) normals[i] = Vec4f{normals_xyz[3 * i], normals_xyz[3 * i + 1], normals_xyz[3 * i + 2], 0.0f};
    }
    if (uvs) uv.assign(uvs, uvs + 2 * n_vertices);
    w->w.indices.resize(n_triangles);
    for (size_t i = 0; i < n_triangles; ++i) {
        rpt_triangle t{tris[4 * i], tris[4 * i + 1], tris[4 * i + 2], tris[4 * i + 3]};
        if (t.v0 >= n_vertices || t.v1 >= n_vertices || t.v2 >= n_vertices || t.material >= n_materials) {
            delete w;
            set_error("triangle index out of range");
            return RPT_ESCENE;
        }
        w->w.indices[i] = t;
    }
    w->w.materials.assign(materials, materials + n_materials);
    if (!finish_world(w->w, vertices, normals, tangents, uv)) { delete w; return RPT_EHIP; }
    *out = w;
    return 0;
}

int rpt_world_view_get(const rpt_world *world, rpt_world_view *v) {
    if (!world || !v) { set_error("null argument"); return RPT_EINVAL; }
    const World &w = world->w;
    v->per_vertex = w.per_vertex.data(); v->n_vertices = w.per_vertex.size();
    v->indices = w.indices.data(); v->n_triangles = w.indices.size();
    v->nodes = w.nodes.data(); v->n_nodes = w.nodes.size();
    v->materials = w.materials.data(); v->n_materials = w.materials.size();
    v->light_pick = w.light_pick.data(); v->n_light_pick = w.light_pick.size();
    v->atlas_rgba8 = w.atlas.empty() ? nullptr : w.atlas.data();
    v->atlas_w = w.atlas_w; v->atlas_h = w.atlas_h;
    v->bvh_max_depth = w.max_depth;
    v->n_emissive_triangles = w.n_emissive;
    return 0;
}

void rpt_world_free(rpt_world *world) { delete world; }

int rpt_bvh_build(const float *vertices_xyzw, size_t n_vertices, rpt_triangle *triangles, size_t n_triangles,
                  uint32_t sah_samples, rpt_bvh_node *nodes_out, size_t nodes_capacity, size_t *n_nodes_out) {
    if (!vertices_xyzw || !triangles || !nodes_out || !n_nodes_out || !n_triangles) { set_error("null/empty argument"); return RPT_EINVAL; }
    for (size_t i = 0; i < n_triangles; ++i)
        if (triangles[i].v0 >= n_vertices || triangles[i].v1 >= n_vertices || triangles[i].v2 >= n_vertices) {
            set_error("triangle index out of range");
            return RPT_ESCENE;
        }
    std::vector<rpt_bvh_node> nodes;
    size_t n = bvh_build(reinterpret_cast<const Vec4f *>(vertices_xyzw), triangles, n_triangles, sah_samples, nodes);
    if (n > nodes_capacity) { set_error("node buffer too small"); return RPT_EINVAL; }
    memcpy(nodes_out, nodes.data(), n * sizeof(rpt_bvh_node));
    *n_nodes_out = n;
    return 0;
}

int rpt_light_table_build(const float *vertices_xyzw, size_t n_vertices, const rpt_triangle *triangles,
                          size_t n_triangles, const rpt_material_data *materials, size_t n_materials,
                          rpt_light_pick_entry *table_out, size_t capacity, size_t *n_entries_out) {
    if (!vertices_xyzw || !triangles || !materials || !table_out || !n_entries_out) { set_error("null argument"); return RPT_EINVAL; }
    for (size_t i = 0; i < n_triangles; ++i)
        if (triangles[i].v0 >= n_vertices || triangles[i].v1 >= n_vertices || triangles[i].v2 >= n_vertices ||
            triangles[i].material >= n_materials) {
            set_error("triangle index out of range");
            return RPT_ESCENE;
        }
    auto table = build_light_pick_table(reinterpret_cast<const Vec4f *>(vertices_xyzw), triangles, n_triangles, materials, nullptr);
    if (table.size() > capacity) { set_error("table buffer too small"); return RPT_EINVAL; }
    memcpy(table_out, table.data(), table.size() * sizeof(rpt_light_pick_entry));
    *n_entries_out = table.size();
    return 0;
}

/* ----------------------------------------------------------- seeds ------ */
int rpt_blue_noise_tile(const char *png_path, uint8_t *out, size_t capacity, uint32_t *w, uint32_t *h) {
    std::vector<uint8_t> tile;
    uint32_t tw, th;
    std::string p = png_path ? std::string(png_path) : default_fixture_path("bluenoise.png");
    if (!load_blue_noise(p.c_str(), tile, tw, th)) return RPT_HOST_EPNG;
    if (w) *w = tw;
    if (h) *h = th;
    if (out) {
        if (capacity < tile.size()) { set_error("tile buffer too small"); return RPT_EINVAL; }
        memcpy(out, tile.data(), tile.size());
    }
    return 0;
}

int rpt_blue_noise_seeds(const char *png_path, uint32_t width, uint32_t height, rpt_rng_state *out) {
    if (!out) { set_error("null argument"); return RPT_EINVAL; }
    std::vector<uint8_t> tile;
    uint32_t tw, th;
    std::string p = png_path ? std::string(png_path) : default_fixture_path("bluenoise.png");
    if (!load_blue_noise(p.c_str(), tile, tw, th)) return RPT_HOST_EPNG;
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            float pixel = (float)tile[(size_t)(y % th) * tw + (x % tw)] / 255.0f;
            float scaled = pixel * 4294967295.0f;             /* literal rounds to 2^32 in f32 */
            uint32_t seed = !(scaled > 0.0f) ? 0u : (scaled >= 4294967296.0f ? 0xffffffffu : (uint32_t)scaled);   /* `as u32` */
            out[(size_t)y * width + x] = rpt_rng_state{0u, seed};
        }
    return 0;
}

/* ---------------------------------------------------- TracingState ------ */
rpt_tracing_state *rpt_tracing_state_new(uint32_t width, uint32_t height) {
    auto *s = new rpt_tracing_state();
    rpt_tracing_config_default(&s->config);
    s->config.width = width;
    s->config.height = height;
    s->framebuffer.assign((size_t)width * height * 3, 0.0f);
    return s;
}
void rpt_tracing_state_free(rpt_tracing_state *s) { delete s; }
rpt_tracing_config *rpt_tracing_state_config(rpt_tracing_state *s) { return &s->config; }
const float *rpt_tracing_state_framebuffer(rpt_tracing_state *s, size_t *n) {
    if (n) *n = s->framebuffer.size();
    return s->framebuffer.data();
}
/* state.framebuffer.read() from another thread while rpt_trace_gpu runs: a copy taken under the state's lock */
int rpt_tracing_state_copy_framebuffer(rpt_tracing_state *s, float *out, size_t n_floats) {
    if (!s || !out) return RPT_EINVAL;
    std::lock_guard<std::mutex> g(s->lock);
    if (n_floats != s->framebuffer.size()) return RPT_EINVAL;
    memcpy(out, s->framebuffer.data(), n_floats * sizeof(float));
    return 0;
}
uint32_t rpt_tracing_state_samples(rpt_tracing_state *s) { return s->samples.load(std::memory_order_relaxed); }
void rpt_tracing_state_set_running(rpt_tracing_state *s, int r) { s->running.store(r != 0, std::memory_order_relaxed); }
void rpt_tracing_state_set_sync_rate(rpt_tracing_state *s, uint32_t r) { s->sync_rate.store(r ? r : 1, std::memory_order_relaxed); }
void rpt_tracing_state_set_dirty(rpt_tracing_state *s, int d) { s->dirty.store(d != 0, std::memory_order_relaxed); }
/* state.interacting (src/trace.rs:50; the UI raises it while the camera is dragged, src/app.rs): every batch flushes while it is up */
void rpt_tracing_state_set_interacting(rpt_tracing_state *s, int on) { if (s) s->interacting.store(on != 0, std::memory_order_relaxed); }
/* state.config.write() of the UI thread (src/app.rs) while trace_gpu runs: under the lock the render loop takes when it
 * re-reads the configuration on a flush (trace.rs:216-222); follow with rpt_tracing_state_set_dirty(s, 1) */
void rpt_tracing_state_set_overlap(rpt_tracing_state *s, int on) { if (s) s->overlap.store(on != 0, std::memory_order_relaxed); }
void rpt_tracing_state_set_config(rpt_tracing_state *s, const rpt_tracing_config *c) {
    if (!s || !c) return;
    std::lock_guard<std::mutex> g(s->lock);
    s->config = *c;
}

rpt_tracing_state *rpt_setup_trace(uint32_t width, uint32_t height, uint32_t samples) {
    rpt_tracing_state *s = rpt_tracing_state_new(width, height);
    s->running.store(true, std::memory_order_relaxed);
    s->target_samples = samples;
    if (samples == 0) s->running.store(false, std::memory_order_relaxed);   /* "Startup time" benches: 0 samples */
    return s;
}

}  // extern "C"

/* ------------------------------------------------ BVH builder choice ---- */
namespace {
typedef int (*bvh_build_gpu_fn)(int, const float *, size_t, rpt_triangle *, size_t, uint32_t, rpt_bvh_node *, size_t, size_t *, double *);
typedef const char *(*last_error_fn)(void *);
std::mutex g_builder_lock;
bvh_build_gpu_fn g_gpu_builder = nullptr;
last_error_fn g_gpu_last_error = nullptr;
void *g_gpu_builder_lib = nullptr;
int g_gpu_builder_device = 0;
std::string hip_library_default_path() {
    Dl_info info;
    std::string p = "librpt_hip.so";
    if (dladdr((void *)&hip_library_default_path, &info) && info.dli_fname) {
        std::string self(info.dli_fname);
        size_t s = self.find_last_of('/');
        if (s != std::string::npos) p = self.substr(0, s) + "/librpt_hip.so";
    }
    return p;
}
}  // namespace

namespace rpth {
bool build_world_bvh(const Vec4f *vertices, size_t n_vertices, std::vector<rpt_triangle> &triangles, std::vector<rpt_bvh_node> &nodes) {
    bvh_build_gpu_fn gpu;
    int device;
    { std::lock_guard<std::mutex> g(g_builder_lock); gpu = g_gpu_builder; device = g_gpu_builder_device; }
    if (!gpu || triangles.empty()) {
        bvh_build(vertices, triangles.data(), triangles.size(), 128, nodes);
        return true;
    }
    nodes.assign(triangles.size() * 2 - 1, rpt_bvh_node{});
    size_t n = 0;
    int rc = gpu(device, reinterpret_cast<const float *>(vertices), n_vertices, triangles.data(), triangles.size(), 128, nodes.data(),
                 nodes.size(), &n, nullptr);
    if (rc) {
        set_error(std::string("GPU BVH build failed: ") + (g_gpu_last_error ? g_gpu_last_error(nullptr) : "?"));
        return false;                                    /* no silent fall-back to the host builder */
    }
    nodes.resize(n);
    return true;
}
}  // namespace rpth

extern "C" int rpt_host_set_bvh_builder(int use_gpu, const char *hip_library_path, int device_id) {
    std::lock_guard<std::mutex> g(g_builder_lock);
    if (!use_gpu) { g_gpu_builder = nullptr; return 0; }
    if (!g_gpu_builder_lib) {
        std::string p = hip_library_path ? std::string(hip_library_path) : hip_library_default_path();
        g_gpu_builder_lib = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!g_gpu_builder_lib) { set_error(std::string("dlopen failed: ") + dlerror()); return RPT_HOST_EDLOPEN; }
    }
    auto fn = reinterpret_cast<bvh_build_gpu_fn>(dlsym(g_gpu_builder_lib, "rpt_bvh_build_gpu"));
    g_gpu_last_error = reinterpret_cast<last_error_fn>(dlsym(g_gpu_builder_lib, "rpt_last_error"));
    if (!fn) { set_error("librpt_hip.so has no rpt_bvh_build_gpu"); return RPT_HOST_EDLOPEN; }
    g_gpu_builder = fn;
    g_gpu_builder_device = device_id;
    return 0;
}

/* ------------------------------------------------------- trace_gpu ------ */
namespace {
struct HipApi {
    void *handle = nullptr;
    decltype(&rpt_create) create;
    decltype(&rpt_upload_scene) upload_scene;
    decltype(&rpt_set_config) set_config;
    decltype(&rpt_reset) reset;
    decltype(&rpt_render) render;
    decltype(&rpt_read_accum) read_accum;
    decltype(&rpt_render_async) render_async;
    decltype(&rpt_comm_init_local) comm_init_local;
    decltype(&rpt_gather_async) gather_async;
    decltype(&rpt_read_gathered) read_gathered;
    decltype(&rpt_destroy) destroy;
    decltype(&rpt_last_error) last_error;
};
template <typename T> bool sym(void *h, const char *name, T &fn) {
    fn = reinterpret_cast<T>(dlsym(h, name));
    if (!fn) set_error(std::string("missing symbol ") + name);
    return fn != nullptr;
}
bool load_hip_api(const char *path, HipApi &api) {
    std::string p;
    if (path) {
        p = path;
    } else {
        Dl_info info;
        p = "librpt_hip.so";
        if (dladdr((void *)&load_hip_api, &info) && info.dli_fname) {
            std::string self(info.dli_fname);
            size_t s = self.find_last_of('/');
            if (s != std::string::npos) p = self.substr(0, s) + "/librpt_hip.so";
        }
    }
    api.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!api.handle) { set_error(std::string("dlopen failed: ") + dlerror()); return false; }
    return sym(api.handle, "rpt_create", api.create) && sym(api.handle, "rpt_upload_scene", api.upload_scene) &&
           sym(api.handle, "rpt_set_config", api.set_config) && sym(api.handle, "rpt_reset", api.reset) &&
           sym(api.handle, "rpt_render", api.render) && sym(api.handle, "rpt_read_accum", api.read_accum) &&
           sym(api.handle, "rpt_render_async", api.render_async) && sym(api.handle, "rpt_comm_init_local", api.comm_init_local) &&
           sym(api.handle, "rpt_gather_async", api.gather_async) && sym(api.handle, "rpt_read_gathered", api.read_gathered) &&
           sym(api.handle, "rpt_destroy", api.destroy) && sym(api.handle, "rpt_last_error", api.last_error);
}
}  // namespace

extern "C" {

int rpt_trace_gpu(const char *scene_path, const char *skybox_path, rpt_tracing_state *state, int device_id,
                  const char *hip_library_path) {
    if (!scene_path || !state) { set_error("null argument"); return RPT_EINVAL; }
    /* skybox_path.and_then(load_dynamic_image) ... unwrap_or_else(fallback) (trace.rs:144): a file that cannot be
     * read falls back to the 2x2 magenta image, as the reference does */
    std::vector<float> skybox;
    uint32_t sky_w = 0, sky_h = 0;
    if (skybox_path && !load_skybox_file(skybox_path, skybox, sky_w, sky_h)) { skybox.clear(); sky_w = sky_h = 0; }

    rpt_world *world = nullptr;
    int rc = rpt_world_load(scene_path, &world);
    if (rc) return rc;                                   /* reference: silent return (trace.rs:141-143) */

    HipApi api;
    if (!load_hip_api(hip_library_path, api)) { rpt_world_free(world); return RPT_HOST_EDLOPEN; }

    rpt_tracing_config config;
    { std::lock_guard<std::mutex> g(state->lock); config = state->config; }
    const uint32_t W = config.width, H = config.height;
    const size_t pixel_count = (size_t)W * H;

    /* seeds (trace.rs:149-160). Uniform mode uses thread_rng() in the reference
     * (non-deterministic); here a fixed SplitMix64 stream so runs are repeatable. */
    std::vector<rpt_rng_state> rng_blue(pixel_count), rng_uniform(pixel_count);
    if (rpt_blue_noise_seeds(nullptr, W, H, rng_blue.data())) { rpt_world_free(world); dlclose(api.handle); return RPT_HOST_EPNG; }
    uint64_t sm = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < pixel_count; ++i) {
        uint64_t z = (sm += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        rng_uniform[i] = rpt_rng_state{(uint32_t)((z ^ (z >> 31)) >> 32), 0u};
    }
    auto seeds = [&]() { return state->use_blue_noise.load(std::memory_order_relaxed) ? rng_blue.data() : rng_uniform.data(); };

    /* restore previous state (trace.rs:163-164) */
    uint32_t samples_init = state->samples.load(std::memory_order_relaxed);
    std::vector<float> accum_init(pixel_count * 4);
    {
        std::lock_guard<std::mutex> g(state->lock);
        float si = (float)samples_init;
        for (size_t i = 0; i < pixel_count; ++i) {
            accum_init[4 * i + 0] = state->framebuffer[3 * i + 0] * si;
            accum_init[4 * i + 1] = state->framebuffer[3 * i + 1] * si;
            accum_init[4 * i + 2] = state->framebuffer[3 * i + 2] * si;
            accum_init[4 * i + 3] = 1.0f * si;
        }
    }

    rpt_ctx *ctx = nullptr;
    auto fail = [&](int code) {
        set_error(std::string("librpt_hip: ") + api.last_error(ctx));
        if (ctx) api.destroy(ctx);
        rpt_world_free(world);
        dlclose(api.handle);
        return code;
    };
    if ((rc = api.create(device_id, &ctx))) return fail(rc);
    rpt_world_view v;
    rpt_world_view_get(world, &v);
    if ((rc = api.upload_scene(ctx, v.per_vertex, v.n_vertices, v.indices, v.n_triangles, v.nodes, v.n_nodes, v.materials,
                               v.n_materials, v.light_pick, v.n_light_pick, v.atlas_rgba8, v.atlas_w, v.atlas_h,
                               skybox.empty() ? nullptr : skybox.data(), sky_w, sky_h)))
        return fail(rc);
    if ((rc = api.set_config(ctx, &config))) return fail(rc);
    if ((rc = api.reset(ctx, seeds(), accum_init.data(), samples_init))) return fail(rc);

    std::vector<float> image_raw(pixel_count * 4), image(pixel_count * 3);
    /* Overlapped form (the default): batch k+1 is enqueued before the image after batch k is read — the
     * read-back (device un-tile, DMA, host copy) hides behind the rendering; the framebuffer and `samples` the UI sees are
     * those of the image just read, one batch behind the device.  Same images, same flush semantics: an iteration that
     * flushes (`interacting | dirty`) reads and publishes what it has just enqueued before the reset, as the reference
     * publishes on every iteration (trace.rs:198-213 run before 216-222) — so a camera drag, which holds `interacting` for many
     * iterations, keeps updating the framebuffer, with 1-sample images as in the reference (the flag is seen at the top of the
     * iteration here, after the first sample there; a flag raised DURING a batch takes effect after that batch, up to sync_rate - 1
     * samples later than in the reference, whose loop polls it per sample).  rpt_tracing_state_set_overlap(state, 0) selects the
     * blocking loop (rpt_render ; rpt_read_accum), the reference's literal shape. */
    const bool overlap = state->overlap.load(std::memory_order_relaxed);
    if (overlap && (rc = api.comm_init_local(ctx))) return fail(rc);
    uint32_t in_flight = 0;                              /* overlap: samples of the batch that was enqueued but not read yet */
    auto publish = [&](uint32_t samples_of_image) {
        float sample_count = (float)samples_of_image;
        for (size_t i = 0; i < pixel_count; ++i) {
            image[3 * i + 0] = image_raw[4 * i + 0] / sample_count;
            image[3 * i + 1] = image_raw[4 * i + 1] / sample_count;
            image[3 * i + 2] = image_raw[4 * i + 2] / sample_count;
        }
        std::lock_guard<std::mutex> g(state->lock);
        state->framebuffer = image;
    };
    while (state->running.load(std::memory_order_relaxed)) {
        uint32_t n = state->sync_rate.load(std::memory_order_relaxed);
        if (state->target_samples) {
            uint32_t done = state->samples.load(std::memory_order_relaxed) + in_flight;
            uint32_t left = state->target_samples > done ? state->target_samples - done : 0;
            if (n > left) n = left;
        }
        /* the reference polls `interacting | dirty` after every sample
         * (trace.rs:187); one rpt_render call covers the whole batch, so the
         * poll happens once per batch */
        bool flush = state->interacting.load(std::memory_order_relaxed) || state->dirty.load(std::memory_order_relaxed);
        /* a flag that is ALREADY up ends the reference's inner loop after its first sample (trace.rs:181-189): while a camera drag
         * holds `interacting`, every published frame is a 1-sample image — not a whole sync_rate batch that the reset then discards */
        if (flush && n > 1u) n = 1u;
        if (!overlap) {
            if (n) {
                if ((rc = api.render(ctx, n))) return fail(rc);
                state->samples.fetch_add(n, std::memory_order_relaxed);
            }
            uint32_t device_samples = 0;
            if ((rc = api.read_accum(ctx, image_raw.data(), &device_samples))) return fail(rc);
            publish(state->samples.load(std::memory_order_relaxed));
        } else {
            if (n && (rc = api.render_async(ctx, n))) return fail(rc);          /* batch k+1 ... */
            if (in_flight) {                                                    /* ... while the image after batch k comes back */
                uint32_t device_samples = 0;
                if ((rc = api.read_gathered(ctx, image_raw.data(), &device_samples))) return fail(rc);
                state->samples.store(device_samples, std::memory_order_relaxed);
                publish(device_samples);
            }
            if (n && (rc = api.gather_async(ctx))) return fail(rc);             /* snapshot after batch k+1 */
            in_flight = n;
            if (flush && in_flight) {                                           /* the reset below discards device state: this image is shown first */
                uint32_t device_samples = 0;
                if ((rc = api.read_gathered(ctx, image_raw.data(), &device_samples))) return fail(rc);
                state->samples.store(device_samples, std::memory_order_relaxed);
                publish(device_samples);
            }
            if (!n && !flush) {                                                 /* nothing left to enqueue: the last image is in */
                if (state->target_samples && state->samples.load(std::memory_order_relaxed) >= state->target_samples)
                    state->running.store(false, std::memory_order_relaxed);
                continue;
            }
        }
        if (flush) {
            state->dirty.store(false, std::memory_order_relaxed);
            state->samples.store(0, std::memory_order_relaxed);
            in_flight = 0;                                                      /* (overlap: already read and published above) */
            { std::lock_guard<std::mutex> g(state->lock); config = state->config; }
            if (config.width != W || config.height != H) {
                /* every buffer of this call (seeds, read-back image, the state's framebuffer) is sized for the
                 * resolution the call started with, as in the reference (trace.rs:146-148): a resize needs a new call */
                set_error("resolution changed while rendering: stop and call rpt_trace_gpu again");
                api.destroy(ctx);
                rpt_world_free(world);
                dlclose(api.handle);
                return RPT_EINVAL;
            }
            if ((rc = api.set_config(ctx, &config))) return fail(rc);
            if ((rc = api.reset(ctx, seeds(), nullptr, 0))) return fail(rc);
        }
        if (!overlap && state->target_samples && state->samples.load(std::memory_order_relaxed) >= state->target_samples)
            state->running.store(false, std::memory_order_relaxed);
    }
    if (overlap && in_flight) {                          /* stopped from outside: the batch still in flight is read too */
        uint32_t device_samples = 0;
        if ((rc = api.read_gathered(ctx, image_raw.data(), &device_samples))) return fail(rc);
        state->samples.store(device_samples, std::memory_order_relaxed);
        publish(device_samples);
    }
    api.destroy(ctx);
    rpt_world_free(world);
    dlclose(api.handle);
    return 0;
}

}  // extern "C"
