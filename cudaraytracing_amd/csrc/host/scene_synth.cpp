// Synthetic instanced scene: many transformed copies of one mesh flattened into
// a single mesh (one BVH), a plane floor and one point light.  This is the
// bandwidth-bound workload of the benchmark plan (about ten million triangles
// from 1582 copies of the 6320-triangle teapot); the reference ships no such
// scene, so its definition lives here and is deterministic:
//
//   * a 32-bit LCG (x <- 1664525 x + 1013904223) seeded with `seed`; a draw is
//     the top 24 bits times 2^-24;
//   * copy i sits in cell (i mod n, (i / n) mod n, i / n^2) of an n^3 grid,
//     n = ceil(cbrt(copies)), cell pitch 1.25 x the mesh's largest extent;
//   * five draws per copy, in this order: uniform scale 0.5 + 0.5 u, rotation
//     angle 2 pi u about z, then a jitter of (u - 0.5) / 2 cells along x, y, z
//     (z is lifted by a quarter cell so that every copy stays above the floor);
//   * a vertex p becomes  centre + Rz(angle) * (scale * (p - anchor)),  anchor =
//     the centre of the mesh box in x, y and its floor in z; normals are rotated
//     only, texture coordinates are shared by all copies.
//
// All arithmetic is float with the pinned sine / cosine, so the scene is the
// same bytes on every host.
#include <cmath>
#include <cstring>

#include "host.h"

namespace crt {

namespace {

struct Lcg {
    uint32_t x;
    float next() {
        x = x * 1664525u + 1013904223u;
        return (float)(x >> 8) * (1.0f / 16777216.0f);
    }
};

uint8_t quantise8(float c) {
    const int v = (int)(c * 256);
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

int add_colour(HostScene& sc, float r, float g, float b) {
    HostTexture t;
    t.width = t.height = 1;
    t.rgb = {quantise8(r), quantise8(g), quantise8(b)};
    sc.textures.push_back(std::move(t));
    return (int)sc.textures.size() - 1;
}

crt_material make_material(HostScene& sc, const float d[3], const float s[3], float glossiness) {
    crt_material m{};
    m.diffuse = add_colour(sc, d[0], d[1], d[2]);
    m.specular = add_colour(sc, s[0], s[1], s[2]);
    m.refraction = add_colour(sc, 0, 0, 0);
    m.emission = add_colour(sc, 0, 0, 0);
    m.glossiness = glossiness;
    m.ior = 1;
    return m;
}

crt_node make_node(int kind, int mesh_id, int material_id, const mat34& tm) {
    crt_node n;
    memset(&n, 0, sizeof n);
    n.object_kind = kind; n.mesh_id = mesh_id; n.material_id = material_id;
    const mat34 itm = mat_inverse(tm);
    memcpy(n.tm, tm.c, sizeof tm.c);
    memcpy(n.itm, itm.c, sizeof itm.c);
    return n;
}

}  // namespace

bool scene_make_instanced(const HostScene& source, uint32_t mesh_index, uint32_t copies, uint32_t seed,
                          uint32_t width, uint32_t height, HostScene& out, std::string& err) {
    if (mesh_index >= source.meshes.size()) { err = "mesh index out of range"; return false; }
    const HostMesh& src = source.meshes[mesh_index];
    if (copies == 0 || src.nf() == 0) { err = "nothing to instance"; return false; }
    if ((uint64_t)copies * src.nf() >= (1ull << 27) || (uint64_t)copies * src.nv() >= (1ull << 31)) {
        err = "instanced mesh exceeds 2^27 triangles"; return false;
    }
    if (width == 0 || height == 0) { err = "bad resolution"; return false; }

    out = HostScene();
    uint32_t n = 1;
    while ((uint64_t)n * n * n < copies) n++;
    const float ext_x = src.box_max[0] - src.box_min[0], ext_y = src.box_max[1] - src.box_min[1], ext_z = src.box_max[2] - src.box_min[2];
    const float extent = fmaxf(ext_x, fmaxf(ext_y, ext_z));
    const float pitch = 1.25f * extent;
    const float half = 0.5f * (float)n * pitch;
    const f3 anchor = mk3(0.5f * (src.box_min[0] + src.box_max[0]), 0.5f * (src.box_min[1] + src.box_max[1]), src.box_min[2]);

    HostMesh mesh;
    mesh.name = "instanced:" + src.name;
    const uint32_t nv = src.nv(), nf = src.nf(), nvn = src.nvn();
    const bool has_n = nvn > 0 && src.fn.size() == src.f.size();
    const bool has_t = src.nvt() > 0 && src.ft.size() == src.f.size();
    mesh.v.resize((size_t)copies * nv * 3);
    mesh.f.resize((size_t)copies * nf * 3);
    if (has_n) { mesh.vn.resize((size_t)copies * nvn * 3); mesh.fn.resize((size_t)copies * nf * 3); }
    if (has_t) { mesh.vt = src.vt; mesh.ft.resize((size_t)copies * nf * 3); }

    Lcg rng{seed};
    f3 lo = mk3(CRT_BIGFLOAT, CRT_BIGFLOAT, CRT_BIGFLOAT), hi = mk3(-CRT_BIGFLOAT, -CRT_BIGFLOAT, -CRT_BIGFLOAT);
    for (uint32_t i = 0; i < copies; i++) {
        const float scale = 0.5f + 0.5f * rng.next();
        const float angle = 6.2831853071795864769f * rng.next();
        const float jx = (rng.next() - 0.5f) * 0.5f, jy = (rng.next() - 0.5f) * 0.5f, jz = (rng.next() - 0.5f) * 0.5f;
        const uint32_t cx = i % n, cy = (i / n) % n, cz = i / (n * n);
        const f3 centre = mk3(((float)cx + 0.5f + jx) * pitch - half, ((float)cy + 0.5f + jy) * pitch - half,
                              ((float)cz + (jz + 0.25f)) * pitch);
        float sn, cs;
        crtm::det_sincosf(angle, sn, cs);
        float* v = mesh.v.data() + (size_t)i * nv * 3;
        for (uint32_t k = 0; k < nv; k++) {
            const float qx = (src.v[3 * k] - anchor.x) * scale, qy = (src.v[3 * k + 1] - anchor.y) * scale, qz = (src.v[3 * k + 2] - anchor.z) * scale;
            const float px = (cs * qx - sn * qy) + centre.x, py = (sn * qx + cs * qy) + centre.y, pz = qz + centre.z;
            v[3 * k] = px; v[3 * k + 1] = py; v[3 * k + 2] = pz;
            if (lo.x > px) lo.x = px; if (lo.y > py) lo.y = py; if (lo.z > pz) lo.z = pz;
            if (hi.x < px) hi.x = px; if (hi.y < py) hi.y = py; if (hi.z < pz) hi.z = pz;
        }
        uint32_t* f = mesh.f.data() + (size_t)i * nf * 3;
        for (uint32_t k = 0; k < nf * 3; k++) f[k] = src.f[k] + i * nv;
        if (has_n) {
            float* vn = mesh.vn.data() + (size_t)i * nvn * 3;
            for (uint32_t k = 0; k < nvn; k++) {
                const float nx = src.vn[3 * k], ny = src.vn[3 * k + 1];
                vn[3 * k] = cs * nx - sn * ny; vn[3 * k + 1] = sn * nx + cs * ny; vn[3 * k + 2] = src.vn[3 * k + 2];
            }
            uint32_t* fn = mesh.fn.data() + (size_t)i * nf * 3;
            for (uint32_t k = 0; k < nf * 3; k++) fn[k] = src.fn[k] + i * nvn;
        }
        if (has_t) memcpy(mesh.ft.data() + (size_t)i * nf * 3, src.ft.data(), (size_t)nf * 3 * sizeof(uint32_t));
    }
    mesh.box_min[0] = lo.x; mesh.box_min[1] = lo.y; mesh.box_min[2] = lo.z;
    mesh.box_max[0] = hi.x; mesh.box_max[1] = hi.y; mesh.box_max[2] = hi.z;
    out.meshes.push_back(std::move(mesh));

    // materials: 0 = default white, 1 = the copies, 2 = the floor
    const float white[3] = {1, 1, 1}, body_d[3] = {0.8f, 0.55f, 0.25f}, body_s[3] = {0.25f, 0.25f, 0.25f};
    const float floor_d[3] = {0.55f, 0.55f, 0.6f}, floor_s[3] = {0.05f, 0.05f, 0.05f};
    out.materials.push_back(make_material(out, white, white, 20));
    out.materials.push_back(make_material(out, body_d, body_s, 50));
    out.materials.push_back(make_material(out, floor_d, floor_s, 20));
    out.env_texture = add_colour(out, 0.35f, 0.45f, 0.6f);

    // nodes: root, floor plane (scaled unit square at z = 0), the instanced mesh (identity transform)
    const mat34 ident = mat_identity();
    crt_node root = make_node(CRT_OBJ_NONE, -1, 0, ident);
    root.child_count = 2;
    mat34 floor_tm = mat_identity();
    floor_tm.c[0] = 1.5f * half; floor_tm.c[4] = 1.5f * half; floor_tm.c[8] = 1.5f * half;
    out.nodes.push_back(root);
    out.nodes.push_back(make_node(CRT_OBJ_PLANE, -1, 2, floor_tm));
    out.nodes.push_back(make_node(CRT_OBJ_MESH, 0, 1, ident));
    scene_compute_boxes(out);

    const float top = hi.z;
    crt_light light{};
    light.kind = CRT_LIGHT_POINT;
    light.intensity[0] = light.intensity[1] = light.intensity[2] = 1.0f;
    light.vec[0] = 0.4f * half; light.vec[1] = -0.6f * half; light.vec[2] = top + 1.5f * half;
    light.size = 0.15f * half;
    out.lights.push_back(light);

    crt_camera& cam = out.camera;
    memset(&cam, 0, sizeof cam);
    cam.width = width; cam.height = height;
    cam.position[0] = 1.9f * half; cam.position[1] = -2.3f * half; cam.position[2] = top + 0.6f * half;
    cam.target[0] = 0; cam.target[1] = 0; cam.target[2] = 0.45f * top;
    cam.up[0] = 0; cam.up[1] = 0; cam.up[2] = 1;
    cam.fov = 40;
    cam.dof = 0;
    cam.srgb = 1;
    camera_derive(cam);
    out.refresh_views();
    return true;
}

}  // namespace crt
