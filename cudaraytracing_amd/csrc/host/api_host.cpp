// Host half of the C ABI (scene model, BVH builder, PNG).
#include <cstdlib>
#include <cstring>
#include <new>

#include "host.h"

struct crt_scene {
    crt::HostScene s;
};

struct crt_bvh {
    crt::HostBvh b;
};

namespace crt {

static thread_local std::string g_host_error;
void set_host_error(const std::string& msg) { g_host_error = msg; }

uint32_t knob_value(const char* text, uint32_t fallback, uint32_t lo, uint32_t hi) {
    if (!text || !*text) return fallback;
    uint64_t v = 0;
    for (const char* c = text; *c; c++) {
        if (*c < '0' || *c > '9') return fallback;
        v = v * 10 + (uint64_t)(*c - '0');
        if (v > 0xffffffffull) return fallback;
    }
    return (v < lo || v > hi) ? fallback : (uint32_t)v;
}

void HostScene::refresh_views() {
    texture_views.resize(textures.size());
    for (size_t i = 0; i < textures.size(); i++) {
        texture_views[i].width = textures[i].width;
        texture_views[i].height = textures[i].height;
        texture_views[i].rgb = textures[i].rgb.data();
    }
    mesh_views.resize(meshes.size());
    for (size_t i = 0; i < meshes.size(); i++) {
        const HostMesh& m = meshes[i];
        crt_mesh& v = mesh_views[i];
        v.nv = m.nv(); v.nf = m.nf(); v.nvn = m.nvn(); v.nvt = m.nvt();
        v.v = m.v.data(); v.f = m.f.data();
        v.vn = m.vn.empty() ? nullptr : m.vn.data();
        v.fn = m.fn.empty() ? nullptr : m.fn.data();
        v.vt = m.vt.empty() ? nullptr : m.vt.data();
        v.ft = m.ft.empty() ? nullptr : m.ft.data();
        memcpy(v.box_min, m.box_min, 12); memcpy(v.box_max, m.box_max, 12);
    }
}

}  // namespace crt

extern "C" {

const char* crt_host_last_error(void) { return crt::g_host_error.c_str(); }

int crt_scene_load_xml(const char* xml_path, const char* asset_root, crt_scene** out) {
    if (!xml_path || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    crt_scene* sc = new (std::nothrow) crt_scene();
    if (!sc) { crt::set_host_error("out of memory"); return CRT_E_INVALID; }
    std::string err;
    if (!crt::scene_load_xml(xml_path, asset_root, sc->s, err)) {
        delete sc; crt::set_host_error(err); return CRT_E_IO;
    }
    *out = sc;
    return CRT_OK;
}

int crt_scene_load_file(const char* path, crt_scene** out) {
    if (!path || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    crt_scene* sc = new (std::nothrow) crt_scene();
    if (!sc) { crt::set_host_error("out of memory"); return CRT_E_INVALID; }
    std::string err;
    if (!crt::scene_load_file(path, sc->s, err)) {
        delete sc; crt::set_host_error(err); return CRT_E_IO;
    }
    *out = sc;
    return CRT_OK;
}

int crt_scene_save_file(const crt_scene* scene, const char* path) {
    if (!scene || !path) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    std::string err;
    if (!crt::scene_save_file(scene->s, path, err)) { crt::set_host_error(err); return CRT_E_IO; }
    return CRT_OK;
}

void crt_scene_free(crt_scene* scene) { delete scene; }

int crt_scene_get_desc(const crt_scene* scene, crt_scene_desc* out) {
    if (!scene || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    const crt::HostScene& s = scene->s;
    out->camera = s.camera;
    out->nodes = s.nodes.data(); out->node_count = (uint32_t)s.nodes.size();
    out->materials = s.materials.data(); out->material_count = (uint32_t)s.materials.size();
    out->lights = s.lights.data(); out->light_count = (uint32_t)s.lights.size();
    out->textures = s.texture_views.data(); out->texture_count = (uint32_t)s.texture_views.size();
    out->meshes = s.mesh_views.data(); out->mesh_count = (uint32_t)s.mesh_views.size();
    out->env_texture = s.env_texture;
    return CRT_OK;
}

int crt_camera_derive(crt_camera* cam) {
    if (!cam || cam->width == 0 || cam->height == 0) { crt::set_host_error("camera needs a positive size"); return CRT_E_INVALID; }
    crt::camera_derive(*cam);
    return CRT_OK;
}

int crt_scene_set_resolution(crt_scene* scene, uint32_t width, uint32_t height) {
    if (!scene || width == 0 || height == 0) { crt::set_host_error("bad resolution"); return CRT_E_INVALID; }
    scene->s.camera.width = width;
    scene->s.camera.height = height;
    crt::camera_derive(scene->s.camera);
    return CRT_OK;
}

int crt_scene_make_instanced(const crt_scene* source, uint32_t mesh_index, uint32_t copies, uint32_t seed,
                             uint32_t width, uint32_t height, crt_scene** out) {
    if (!source || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    crt_scene* sc = new (std::nothrow) crt_scene();
    if (!sc) { crt::set_host_error("out of memory"); return CRT_E_INVALID; }
    std::string err;
    if (!crt::scene_make_instanced(source->s, mesh_index, copies, seed, width, height, sc->s, err)) {
        delete sc; crt::set_host_error(err); return CRT_E_INVALID;
    }
    *out = sc;
    return CRT_OK;
}

int crt_bvh_build(const float* v, uint32_t nv, const uint32_t* f, uint32_t nf,
                  const float* mesh_box_min, const float* mesh_box_max, crt_bvh** out) {
    if ((!v && nv) || (!f && nf) || !mesh_box_min || !mesh_box_max || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    crt_bvh* b = new (std::nothrow) crt_bvh();
    if (!b) { crt::set_host_error("out of memory"); return CRT_E_INVALID; }
    std::string err;
    if (!crt::bvh_build(v, nv, f, nf, mesh_box_min, mesh_box_max, b->b, err)) {
        delete b; crt::set_host_error(err);
        return err.find("depth") != std::string::npos ? CRT_E_LIMIT : CRT_E_INVALID;
    }
    *out = b;
    return CRT_OK;
}

int crt_bvh_get_arrays(const crt_bvh* bvh, crt_bvh_arrays* out) {
    if (!bvh || !out) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    const crt::HostBvh& b = bvh->b;
    out->node_count = b.node_count();
    out->elem_count = (uint32_t)b.elements.size();
    out->depth = b.depth;
    out->leaf = b.leaf.data();
    out->elem_start = b.elem_start.data();
    out->elem_count_per_node = b.elem_count.data();
    out->left = b.left.data(); out->right = b.right.data(); out->next = b.next.data();
    out->box = b.box.data();
    out->elements = b.elements.data();
    return CRT_OK;
}

void crt_bvh_free(crt_bvh* bvh) { delete bvh; }

int crt_png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height) {
    std::string err;
    if (!path || !crt::png_write_rgb8(path, rgb, width, height, err)) { crt::set_host_error(path ? err : "null path"); return CRT_E_IO; }
    return CRT_OK;
}

int crt_png_read_rgb8(const char* path, uint8_t** rgb, uint32_t* width, uint32_t* height) {
    if (!path || !rgb || !width || !height) { crt::set_host_error("null argument"); return CRT_E_INVALID; }
    std::vector<uint8_t> data;
    std::string err;
    if (!crt::png_read_rgb8(path, data, *width, *height, err)) { crt::set_host_error(err); return CRT_E_IO; }
    *rgb = (uint8_t*)malloc(data.size());
    if (!*rgb) { crt::set_host_error("out of memory"); return CRT_E_INVALID; }
    memcpy(*rgb, data.data(), data.size());
    return CRT_OK;
}

void crt_free(void* p) { free(p); }

uint32_t crt_knob_value(const char* text, uint32_t fallback, uint32_t lo, uint32_t hi) { return crt::knob_value(text, fallback, lo, hi); }

}  // extern "C"
