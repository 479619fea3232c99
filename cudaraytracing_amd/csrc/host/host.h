// Host-side scene model and helpers behind the C ABI in include/crt_amd.h.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../../include/crt_amd.h"
#include "../crt_vec.h"

namespace crt {

struct HostTexture {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

struct HostMesh {
    std::string name;
    std::vector<float> v, vn, vt;          // xyz triples
    std::vector<uint32_t> f, fn, ft;       // index triples
    float box_min[3], box_max[3];
    uint32_t nv() const { return (uint32_t)(v.size() / 3); }
    uint32_t nf() const { return (uint32_t)(f.size() / 3); }
    uint32_t nvn() const { return (uint32_t)(vn.size() / 3); }
    uint32_t nvt() const { return (uint32_t)(vt.size() / 3); }
};

struct HostScene {
    crt_camera camera{};
    std::vector<crt_node> nodes;
    std::vector<crt_material> materials;
    std::vector<crt_light> lights;
    std::vector<HostTexture> textures;
    std::vector<HostMesh> meshes;
    int32_t env_texture = -1;
    // views handed out by crt_scene_get_desc
    std::vector<crt_texture> texture_views;
    std::vector<crt_mesh> mesh_views;
    void refresh_views();
};

// --- PNG (png.cpp)
bool png_read_rgb8(const char* path, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err);
bool png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height, std::string& err);

// --- OBJ (obj.cpp)
bool obj_load(const char* path, HostMesh& mesh, std::string& err);

// --- scene loader (scene_load.cpp)
bool scene_load_xml(const char* xml_path, const char* asset_root, HostScene& scene, std::string& err);
void camera_derive(crt_camera& cam);
mat34 mat_inverse(const mat34& m);           // Matrix::GetInverse operation order (math/matrix.cuh:25-56)
void scene_compute_boxes(HostScene& scene);  // Node::CalculateBoundingBox over the flat node list

// --- synthetic instanced scene (scene_synth.cpp)
bool scene_make_instanced(const HostScene& source, uint32_t mesh_index, uint32_t copies, uint32_t seed,
                          uint32_t width, uint32_t height, HostScene& out, std::string& err);

// --- flat scene file (scene_file.cpp)
bool scene_save_file(const HostScene& scene, const char* path, std::string& err);
bool scene_load_file(const char* path, HostScene& scene, std::string& err);

// --- BVH (bvh_build.cpp)
// Field-per-array form of the reference's BVHNode list (renderer/bvh.cuh:13-35).
struct HostBvh {
    std::vector<uint8_t> leaf;
    std::vector<uint32_t> elem_start, elem_count, left, right, next;
    std::vector<float> box;                 // 6 per node
    std::vector<uint32_t> elements;
    uint32_t depth = 0;
    uint32_t node_count() const { return (uint32_t)leaf.size(); }
};
// Returns false (err set) when the tree is deeper than 64 levels (bvh.cu:29-34).
bool bvh_build(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
               const float* box_min, const float* box_max, HostBvh& out, std::string& err);

void set_host_error(const std::string& msg);
// numeric environment override: the value of `text` if it is a whole decimal number in [lo, hi], else `fallback`
uint32_t knob_value(const char* text, uint32_t fallback, uint32_t lo, uint32_t hi);

}  // namespace crt
