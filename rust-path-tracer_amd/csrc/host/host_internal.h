/* Internal declarations shared by the librpt_host.so translation units. */
#ifndef RPT_HOST_INTERNAL_H
#define RPT_HOST_INTERNAL_H

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/rpt/rpt_host.h"

namespace rpth {

struct Vec4f { float x, y, z, w; };

void set_error(const std::string &msg);

/* src/bvh.rs */
size_t bvh_build(const Vec4f *vertices, rpt_triangle *triangles, size_t n_triangles, uint32_t sah_samples,
                 std::vector<rpt_bvh_node> &nodes);
uint32_t bvh_max_depth(const std::vector<rpt_bvh_node> &nodes);

/* src/light_pick.rs */
std::vector<rpt_light_pick_entry> build_light_pick_table(const Vec4f *vertices, const rpt_triangle *triangles,
                                                         size_t n_triangles, const rpt_material_data *materials,
                                                         uint32_t *n_emissive);

/* textures.cpp: decoders, the texture atlas (src/atlas.rs), the skybox as the CPU path sees it (src/asset.rs:238-273) */
struct Image8 { uint32_t w = 0, h = 0; std::vector<uint8_t> rgba; };
bool read_file(const char *path, std::vector<uint8_t> &data);
bool decode_png(const uint8_t *data, size_t size, Image8 &out);
bool decode_jpeg(const uint8_t *data, size_t size, Image8 &out);     /* jpeg_decode.cpp: baseline + progressive, grey / YCbCr, as jpeg-decoder 0.3 */
bool decode_image(const uint8_t *data, size_t size, Image8 &out);    /* PNG or JPEG by signature (image::load_from_memory) */
void albedo_gamma_to_linear(Image8 &img);
void resize_lanczos3(const Image8 &src, uint32_t dw, uint32_t dh, Image8 &dst);
void pack_textures(const std::vector<Image8> &textures, uint32_t atlas_w, uint32_t atlas_h, std::vector<uint8_t> &atlas,
                   std::vector<std::array<float, 4>> &sts);
bool load_skybox_file(const char *path, std::vector<float> &rgba, uint32_t &w, uint32_t &h);

/* bluenoise.png -> 8-bit tile */
bool load_blue_noise(const char *path, std::vector<uint8_t> &tile, uint32_t &w, uint32_t &h);
std::string default_fixture_path(const char *name);

struct World {
    std::vector<rpt_per_vertex_data> per_vertex;
    std::vector<rpt_triangle> indices;
    std::vector<rpt_bvh_node> nodes;
    std::vector<rpt_material_data> materials;
    std::vector<rpt_light_pick_entry> light_pick;
    std::vector<uint8_t> atlas;
    uint32_t atlas_w = 0, atlas_h = 0;
    uint32_t max_depth = 0, n_emissive = 0;
};

/* shared tail of World::from_path (src/asset.rs:194-223) */
bool finish_world(World &w, std::vector<Vec4f> &vertices, std::vector<Vec4f> &normals, std::vector<Vec4f> &tangents,
                  std::vector<float> &uvs);

/* BVH builder used by finish_world: the sequential restatement (default) or the device build of librpt_hip.so
 * (rpt_host_set_bvh_builder).  Both produce the same node pool and triangle order. */
bool build_world_bvh(const Vec4f *vertices, size_t n_vertices, std::vector<rpt_triangle> &triangles, std::vector<rpt_bvh_node> &nodes);

/* flags: RPT_LOAD_* of rpt_host.h */
bool load_glb(const char *path, World &out, uint32_t flags = 0);
/* Wavefront OBJ + MTL (obj_scene.cpp) */
bool load_obj(const char *path, World &out);

}  // namespace rpth

struct rpt_world { rpth::World w; };

#endif
