//! parity_test.rs — run the REAL `kernels::trace_pixel` (kernels/src/lib.rs:21-186) on the byte-identical inputs of the parity
//! kit and compare with the committed accumulators of the MI355X backend's CPU oracle (which the HIP kernels equal bit for bit).
//!
//! NOT compiled in the backend's build image (no rustc there).  For a maintainer of rust-path-tracer:
//!   1. copy this file to `tests/parity_kit.rs` of the reference checkout (it uses only crates the package already depends
//!      on: kernels, shared_structs, glam, bytemuck, rayon);
//!   2. `RPT_PARITY_KIT=<backend checkout>/tests/golden/parity_kit cargo test --release --test parity_kit -- --nocapture`
//!
//! Five cases (FurnaceTest and DarkCornell with nee 0 / MIS, VeachMIS with MIS).  Per case the test replays exactly what `trace_cpu` does per sample (src/trace.rs:278-298: `trace_pixel(UVec3(x, y, 1), ...)`,
//! `output[x] += radiance`, `rng[x] = rng_state`) `spp` times and asserts  rel-L2(accumulators) <= 1e-4  — BASELINE's bar.
//! It also REPORTS how many 32-bit words differ from `<case>.accum.bin` (the oracle with correctly rounded shared
//! transcendentals) and from `<case>.accum_libm.bin` (the oracle calling glibc, as the Rust std f32 functions do on Linux):
//! zero differing words against the latter on a machine with the same glibc means the oracle restates the reference exactly.
//!
//! File layouts: tools/export_parity_kit.py (docstring) and include/rpt/rpt_host.h (".rptscene").

use glam::{UVec2, UVec3, UVec4, Vec4};
use rayon::prelude::*;
use shared_structs::{BVHNode, CpuImage, LightPickEntry, MaterialData, PerVertexData, TracingConfig};
use std::path::{Path, PathBuf};

struct Scene {
    per_vertex: Vec<PerVertexData>,
    index: Vec<UVec4>,
    nodes: Vec<BVHNode>,
    materials: Vec<MaterialData>,
    light_pick: Vec<LightPickEntry>,
    atlas: Vec<Vec4>,
    atlas_size: (u32, u32),
}

fn pod_vec<T: bytemuck::Pod>(bytes: &[u8]) -> Vec<T> {
    bytemuck::pod_collect_to_vec::<u8, T>(bytes) // (copies: the file buffer need not be aligned for T)
}

/// ".rptscene": header { "RPTSCN01", u64 n_vertices, n_triangles, n_nodes, n_materials, n_light_pick, u32 atlas_w, atlas_h },
/// then PerVertexData[], UVec4[], BVHNode[], MaterialData[], LightPickEntry[], RGBA8 atlas — all little-endian, no padding.
fn load_scene(path: &Path) -> Scene {
    let b = std::fs::read(path).unwrap_or_else(|e| panic!("{}: {e}", path.display()));
    assert_eq!(&b[0..8], b"RPTSCN01", "not an .rptscene file");
    let u64_at = |o: usize| u64::from_le_bytes(b[o..o + 8].try_into().unwrap()) as usize;
    let u32_at = |o: usize| u32::from_le_bytes(b[o..o + 4].try_into().unwrap());
    let (nv, nt, nn, nm, nl) = (u64_at(8), u64_at(16), u64_at(24), u64_at(32), u64_at(40));
    let (aw, ah) = (u32_at(48), u32_at(52));
    let mut at = 56usize;
    let mut take = |bytes: usize| {
        let s = &b[at..at + bytes];
        at += bytes;
        s
    };
    let per_vertex = pod_vec::<PerVertexData>(take(nv * std::mem::size_of::<PerVertexData>()));
    let index = pod_vec::<UVec4>(take(nt * 16));
    let nodes = pod_vec::<BVHNode>(take(nn * std::mem::size_of::<BVHNode>()));
    let materials = pod_vec::<MaterialData>(take(nm * std::mem::size_of::<MaterialData>()));
    let light_pick = pod_vec::<LightPickEntry>(take(nl * std::mem::size_of::<LightPickEntry>()));
    // CPU atlas texel = (r, g, b, 255) / 255 (dynamic_image_to_cpu_buffer, src/asset.rs:266-273); no atlas in the file: the
    // kit's file scenes have no textures and the image is never sampled — the reference's 2x2 fallback stands in (asset.rs:283-290);
    // Textured.rptscene carries a 64 x 64 atlas
    let (atlas, atlas_size) = if aw > 0 && ah > 0 {
        let px = take(aw as usize * ah as usize * 4);
        (px.chunks(4).map(|p| Vec4::new(p[0] as f32, p[1] as f32, p[2] as f32, 255.0) / 255.0).collect(), (aw, ah))
    } else {
        (vec![Vec4::new(1.0, 0.0, 1.0, 1.0); 4], (2, 2))
    };
    Scene { per_vertex, index, nodes, materials, light_pick, atlas, atlas_size }
}

fn kit_dir() -> PathBuf {
    PathBuf::from(std::env::var("RPT_PARITY_KIT").expect("set RPT_PARITY_KIT to <backend checkout>/tests/golden/parity_kit"))
}

fn differing_words(a: &[Vec4], b: &[Vec4]) -> usize {
    a.iter().zip(b).map(|(x, y)| x.to_array().iter().zip(y.to_array()).filter(|(p, q)| p.to_bits() != q.to_bits()).count()).sum()
}

fn rel_l2(a: &[Vec4], b: &[Vec4]) -> f64 {
    let (mut num, mut den) = (0f64, 0f64);
    for (x, y) in a.iter().zip(b) {
        for k in 0..3 {
            let (p, q) = (x[k] as f64, y[k] as f64);
            num += (p - q) * (p - q);
            den += q * q;
        }
    }
    (num / den).sqrt()
}

fn run_case(name: &str, scene_file: &str, width: u32, height: u32, spp: u32) {
    let dir = kit_dir();
    let scene = load_scene(&dir.join(scene_file));
    let config: TracingConfig = *bytemuck::from_bytes(&std::fs::read(dir.join(format!("{name}.config.bin"))).unwrap()[..std::mem::size_of::<TracingConfig>()]);
    assert_eq!((config.width, config.height), (width, height));
    let mut rng: Vec<UVec2> = pod_vec(&std::fs::read(dir.join(format!("seeds_{width}x{height}.bin"))).unwrap());
    let expected: Vec<Vec4> = pod_vec(&std::fs::read(dir.join(format!("{name}.accum.bin"))).unwrap());
    let expected_libm: Vec<Vec4> = pod_vec(&std::fs::read(dir.join(format!("{name}.accum_libm.bin"))).unwrap());
    let pixel_count = (width * height) as usize;
    assert_eq!(rng.len(), pixel_count);
    assert_eq!(expected.len(), pixel_count);

    let atlas_image = CpuImage::new(&scene.atlas, scene.atlas_size.0, scene.atlas_size.1);
    let skybox_buffer = vec![Vec4::new(1.0, 0.0, 1.0, 1.0); 4]; // fallback_cpu_buffer(): has_skybox = 0 in every kit config
    let skybox_image = CpuImage::new(&skybox_buffer, 2, 2);
    let mut output = vec![Vec4::ZERO; pixel_count];
    for _sample in 0..spp {
        // src/trace.rs:278-298, verbatim in structure: rows in parallel, pixels of a row in order
        let outputs = output.par_chunks_mut(width as usize).enumerate();
        let rngs = rng.par_chunks_mut(width as usize);
        outputs.zip(rngs).for_each(|((y, out_row), rng_row)| {
            for x in 0..width {
                let (radiance, rng_state) = kernels::trace_pixel(
                    UVec3::new(x, y as u32, 1),
                    &config,
                    rng_row[x as usize],
                    &scene.per_vertex,
                    &scene.index,
                    &scene.nodes,
                    &scene.materials,
                    &scene.light_pick,
                    &shared_structs::Sampler,
                    &atlas_image,
                    &skybox_image,
                );
                out_row[x as usize] += radiance;
                rng_row[x as usize] = rng_state;
            }
        });
    }

    assert!(rng.iter().all(|r| r.x == spp), "rng[i].x advances by one per sample (kernels/src/rng.rs:47-49)");
    assert!(output.iter().all(|o| o.w == spp as f32), "the w lane counts samples (kernels/src/lib.rs:185)");
    let err = rel_l2(&output, &expected);
    let words = differing_words(&output, &expected);
    let words_libm = differing_words(&output, &expected_libm);
    println!(
        "{name}: rel-L2 vs oracle {err:.3e}; differing 32-bit words: {words} of {} vs the shared-math oracle, {words_libm} vs the libm oracle{}",
        pixel_count * 4,
        if words_libm == 0 { "  (BITWISE: the oracle restates trace_pixel exactly)" } else { "" }
    );
    assert!(err <= 1e-4, "{name}: rel-L2 {err} exceeds 1e-4");
}

#[test]
fn furnace_nee0() {
    run_case("furnace_nee0", "FurnaceTest.rptscene", 128, 128, 32);
}

#[test]
fn furnace_mis() {
    run_case("furnace_mis", "FurnaceTest.rptscene", 128, 128, 32);
}

#[test]
fn darkcornell_nee0() {
    run_case("darkcornell_nee0", "DarkCornell.rptscene", 128, 128, 32);
}

#[test]
fn darkcornell_mis() {
    run_case("darkcornell_mis", "DarkCornell.rptscene", 128, 128, 32);
}

/// An open scene: most paths end in the procedural sky (skybox.rs:18-94), glossy plates, NEE + MIS.
#[test]
fn veachmis_mis() {
    run_case("veachmis_mis", "VeachMIS.rptscene", 128, 128, 32);
}

/// The texture path (round 6): a procedural scene with a 64 x 64 RGBA8 atlas — albedo / metallic / roughness / normal-map lookups through
/// `CpuImage::sample_by_lod` (shared_structs/src/image_polyfill.rs:38-55), uv wrap (kernels/src/lib.rs:127-129), the tangent frame (:132-141).
/// No shipped scene file carries a texture; the `.rptscene` holds the atlas bytes, `load_scene` turns them into the CPU texels.
#[test]
fn textured_mis() {
    run_case("textured_mis", "Textured.rptscene", 128, 128, 32);
}

/// The buffers themselves: does the reference's own importer + BVH builder + light table produce the kit's `.rptscene`?
/// (assimp's vertex joining may legitimately reorder vertices; a mismatch here localises a later image difference to the
/// INPUT side instead of the kernels.)  Needs `World` to be reachable from tests (`rustic::asset::World`).
#[test]
#[ignore = "informational: compares World::from_path buffers with the kit's; run with --ignored"]
fn world_buffers_match_the_kit() {
    use rustic::asset::World;
    for (glb, file) in [("scenes/FurnaceTest.glb", "FurnaceTest.rptscene"), ("scenes/DarkCornell.glb", "DarkCornell.rptscene"), ("scenes/VeachMIS.glb", "VeachMIS.rptscene")] {
        let world = World::from_path(glb).expect("scene");
        let kit = load_scene(&kit_dir().join(file));
        let same = |a: &[u8], b: &[u8]| a == b;
        println!(
            "{glb}: per_vertex {} index {} nodes {} materials {} light_pick {}",
            same(bytemuck::cast_slice(&world.per_vertex_buffer), bytemuck::cast_slice(&kit.per_vertex)),
            same(bytemuck::cast_slice(&world.index_buffer), bytemuck::cast_slice(&kit.index)),
            same(bytemuck::cast_slice(&world.bvh.nodes), bytemuck::cast_slice(&kit.nodes)),
            same(bytemuck::cast_slice(&world.material_data_buffer), bytemuck::cast_slice(&kit.materials)),
            same(bytemuck::cast_slice(&world.light_pick_buffer), bytemuck::cast_slice(&kit.light_pick)),
        );
    }
}
