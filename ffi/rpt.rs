// ffi/rpt.rs — Rust binding of librpt_hip.so (C ABI: include/rpt/rpt.h), to be added to the reference as
// `src/rpt_ffi.rs`.  NOT compiled in this repository's image (no Rust toolchain there); INTEGRATION.md shows the patch of
// `trace_gpu` (reference src/trace.rs:136-224) that calls it, and rust-path-tracer_amd/csrc/host/host_api.cpp runs the
// same call sequence from C++ (rpt_trace_gpu), which the GPU tests exercise.
//
// Link:   build.rs:  println!("cargo:rustc-link-search=native=<dir of librpt_hip.so>");
//                    println!("cargo:rustc-link-lib=dylib=rpt_hip");
// Every buffer argument is one of the reference's own `shared_structs` types (#[repr(C)], Pod) passed by pointer:
// nothing is converted on the way in.  All functions return 0 or a negative RPT_E* code; the text is rpt_last_error().

#![allow(non_camel_case_types, dead_code)]

use glam::{UVec2, UVec4, Vec4};
use shared_structs::{BVHNode, LightPickEntry, MaterialData, PerVertexData, TracingConfig};
use std::os::raw::{c_char, c_int, c_void};

pub const RPT_OK: c_int = 0;
pub const RPT_EINVAL: c_int = -1;
pub const RPT_ENODEV: c_int = -2;
pub const RPT_EHIP: c_int = -3;
pub const RPT_ECONFIG: c_int = -4;
pub const RPT_ESCENE: c_int = -5;
pub const RPT_ENOMEM: c_int = -6;
pub const RPT_COMM_ID_BYTES: usize = 128;
pub const RPT_MULTI_ALLOW_SHARED_DEVICE: u32 = 1;

#[repr(C)] pub struct rpt_ctx { _private: [u8; 0] }
#[repr(C)] pub struct rpt_multi { _private: [u8; 0] }

#[repr(C)]
#[derive(Clone, Copy, Default, Debug)]
pub struct rpt_stats {
    pub samples: u64,
    pub extension_rays: u64,
    pub shadow_rays: u64,
    pub sky_evals: u64,
    pub light_index_clamped: u64,
    pub iterations: u64,
    pub render_ms: f64,
    pub kernel_ms: [f64; 8],
    pub kernel_launches: [u64; 8],
    pub shadow_rays_elided: u64,   // of shadow_rays: not walked, their NEE term is zero whatever the walk finds
}

extern "C" {
    pub fn rpt_abi_version() -> c_int;                                              // == 3
    pub fn rpt_build_fingerprint() -> *const c_char;                                // fingerprint of the kernel sources the library was built from
    pub fn rpt_last_error(ctx: *mut rpt_ctx) -> *const c_char;
    pub fn rpt_device_info(device_id: c_int, compute_units_out: *mut u32, clock_khz_out: *mut u32) -> c_int;   // compute units / peak clock of a HIP device
    pub fn rpt_light_table_build_gpu(device_id: c_int, vertices_xyzw: *const f32, n_vertices: usize, triangles: *const [u32; 4], n_triangles: usize,
                                     materials: *const MaterialData, n_materials: usize, entries_out: *mut LightPickEntry, entries_capacity: usize,
                                     n_entries_out: *mut usize, n_emissive_out: *mut u32, ms_out: *mut f64) -> c_int;   // build_light_pick_table, src/light_pick.rs:24-122
    pub fn rpt_shadow_order(ctx: *mut rpt_ctx, fixed_out: *mut u32, visits_near_out: *mut f64, visits_fixed_out: *mut f64, probe_rays_out: *mut u32, probe_ms_out: *mut f64) -> c_int;   // which (bit-exact) order the shadow walks use
    pub fn rpt_last_bounce_order(ctx: *mut rpt_ctx, mode_out: *mut u32, n_emissive_triangles_out: *mut u32, visits_out: *mut f64, probe_rays_out: *mut u32, probe_ms_out: *mut f64) -> c_int;   // how the last extension rays of a batch without NEE are walked

    // --- one GPU: what trace_gpu needs (each line: the reference call it replaces) -------------------------------
    pub fn rpt_create(device_id: c_int, out: *mut *mut rpt_ctx) -> c_int;          // FW / adaptor creation, trace.rs:3-6,25-38
    pub fn rpt_upload_scene(ctx: *mut rpt_ctx,                                      // World::into_gpu asset.rs:226-235, bvh.rs:40-43, skybox trace.rs:144
        per_vertex: *const PerVertexData, n_vertices: usize,
        indices: *const UVec4, n_triangles: usize,
        nodes: *const BVHNode, n_nodes: usize,
        materials: *const MaterialData, n_materials: usize,
        light_pick: *const LightPickEntry, n_light_pick: usize,
        atlas_rgba8: *const u8, atlas_w: u32, atlas_h: u32,                         // world.atlas.to_rgba8(); null = no textures
        skybox_rgba32f: *const f32, sky_w: u32, sky_h: u32) -> c_int;               // Vec4 texels; null = procedural sky
    pub fn rpt_set_config(ctx: *mut rpt_ctx, config: *const TracingConfig) -> c_int;                // config_buffer write, trace.rs:168,219
    pub fn rpt_reset(ctx: *mut rpt_ctx, rng_seed: *const UVec2, accum_init: *const Vec4, samples_init: u32) -> c_int;   // trace.rs:164-170,219-221
    pub fn rpt_render(ctx: *mut rpt_ctx, n_samples: u32) -> c_int;                  // the enqueue / poll_blocking loop, trace.rs:182-194
    pub fn rpt_render_async(ctx: *mut rpt_ctx, n_samples: u32) -> c_int;            // same, returns once enqueued
    pub fn rpt_wait(ctx: *mut rpt_ctx) -> c_int;
    pub fn rpt_read_accum(ctx: *mut rpt_ctx, out: *mut Vec4, out_samples: *mut u32) -> c_int;      // output_buffer.read_blocking, trace.rs:198
    pub fn rpt_map_accum(ctx: *mut rpt_ctx, out: *mut *const Vec4, out_samples: *mut u32) -> c_int; // same without the copy (library-owned pinned buffer)
    pub fn rpt_resolve(ctx: *mut rpt_ctx, tonemap_op: u32, out_rgb: *mut f32) -> c_int;             // sum / samples + render.wgsl tonemappers
    pub fn rpt_get_stats(ctx: *mut rpt_ctx, out: *mut rpt_stats) -> c_int;
    pub fn rpt_destroy(ctx: *mut rpt_ctx);

    // --- every GPU of the node from the one render thread (ncclCommInitAll inside) -------------------------------
    pub fn rpt_multi_create(device_ids: *const c_int, n_devices: c_int, flags: u32, out: *mut *mut rpt_multi) -> c_int;
    pub fn rpt_multi_size(m: *mut rpt_multi) -> c_int;
    pub fn rpt_multi_ctx(m: *mut rpt_multi, rank: c_int) -> *mut rpt_ctx;
    pub fn rpt_multi_upload_scene(m: *mut rpt_multi,
        per_vertex: *const PerVertexData, n_vertices: usize,
        indices: *const UVec4, n_triangles: usize,
        nodes: *const BVHNode, n_nodes: usize,
        materials: *const MaterialData, n_materials: usize,
        light_pick: *const LightPickEntry, n_light_pick: usize,
        atlas_rgba8: *const u8, atlas_w: u32, atlas_h: u32,
        skybox_rgba32f: *const f32, sky_w: u32, sky_h: u32) -> c_int;
    pub fn rpt_multi_set_config(m: *mut rpt_multi, config: *const TracingConfig) -> c_int;
    pub fn rpt_multi_reset(m: *mut rpt_multi, rng_seed: *const UVec2, accum_init: *const Vec4, samples_init: u32) -> c_int;
    pub fn rpt_multi_render(m: *mut rpt_multi, n_samples: u32) -> c_int;            // one batch on every GPU + the batch's single RCCL gather
    pub fn rpt_multi_wait(m: *mut rpt_multi) -> c_int;
    pub fn rpt_multi_read_accum(m: *mut rpt_multi, out: *mut Vec4, out_samples: *mut u32) -> c_int;   // the whole W x H image, from rank 0
    pub fn rpt_multi_get_stats(m: *mut rpt_multi, out: *mut rpt_stats) -> c_int;
    pub fn rpt_multi_last_error(m: *mut rpt_multi) -> *const c_char;
    pub fn rpt_multi_destroy(m: *mut rpt_multi);

    // --- one process per GPU (MPI-style launch): the same gather, communicator from a shared unique id -----------
    pub fn rpt_comm_unique_id(id_out: *mut u8) -> c_int;                            // RPT_COMM_ID_BYTES, rank 0
    pub fn rpt_comm_init(ctx: *mut rpt_ctx, unique_id: *const u8, rank: u32, world_size: u32) -> c_int;
    pub fn rpt_comm_init_local(ctx: *mut rpt_ctx) -> c_int;                         // one GPU, no RCCL: overlapped read-back
    pub fn rpt_gather_async(ctx: *mut rpt_ctx) -> c_int;
    pub fn rpt_gather_wait(ctx: *mut rpt_ctx) -> c_int;
    pub fn rpt_read_gathered(ctx: *mut rpt_ctx, out: *mut Vec4, out_samples: *mut u32) -> c_int;   // rank 0
    pub fn rpt_gathered_device_ptr(ctx: *mut rpt_ctx, dev_ptr: *mut *mut c_void) -> c_int;

    // --- optional: BVHBuilder::new(&vertices, &mut indices).sah_samples(n).build() on the GPU (asset.rs:196) ------
    pub fn rpt_bvh_build_gpu(device_id: c_int, vertices: *const Vec4, n_vertices: usize, indices: *mut UVec4, n_triangles: usize,
        sah_samples: u32, nodes: *mut BVHNode, nodes_capacity: usize, n_nodes: *mut usize, device_ms: *mut f64) -> c_int;
}

/// `Err(message)` for a non-zero return code.
pub unsafe fn check(ctx: *mut rpt_ctx, rc: c_int) -> Result<(), String> {
    if rc == RPT_OK {
        Ok(())
    } else {
        Err(format!("librpt_hip error {}: {}", rc, std::ffi::CStr::from_ptr(rpt_last_error(ctx)).to_string_lossy()))
    }
}
