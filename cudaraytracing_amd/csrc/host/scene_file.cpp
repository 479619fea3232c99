// Flat scene container ("CRTSCN01"): the arrays of a crt_scene_desc, each stored
// as a zlib-compressed blob.  Layout, all little endian:
//
//   char     magic[8] = "CRTSCN01"
//   u32      node_count, material_count, light_count, texture_count, mesh_count
//   i32      env_texture
//   blob     camera    (one crt_camera)
//   blob     nodes     (node_count crt_node)
//   blob     materials (material_count crt_material)
//   blob     lights    (light_count crt_light)
//   per texture: u32 width, height; blob rgb (width*height*3 bytes)
//   per mesh:    u32 nv, nf, nvn, nvt, nfn, nft; f32 box_min[3], box_max[3];
//                blob v, f, vn, fn, vt, ft
//
//   blob = u64 raw_bytes, u64 stored_bytes, stored_bytes of zlib stream
#include <zlib.h>

#include <cstdio>
#include <cstring>

#include "host.h"

namespace crt {

namespace {

bool write_blob(FILE* fp, const void* data, size_t bytes) {
    uLongf clen = compressBound((uLong)bytes);
    std::vector<uint8_t> comp(clen ? clen : 1);
    if (bytes && compress2(comp.data(), &clen, (const Bytef*)data, (uLong)bytes, 6) != Z_OK) return false;
    if (!bytes) clen = 0;
    const uint64_t hdr[2] = {(uint64_t)bytes, (uint64_t)clen};
    if (fwrite(hdr, sizeof hdr, 1, fp) != 1) return false;
    return clen == 0 || fwrite(comp.data(), 1, clen, fp) == clen;
}

bool read_blob(FILE* fp, std::vector<uint8_t>& out, size_t expect_multiple_of) {
    uint64_t hdr[2];
    if (fread(hdr, sizeof hdr, 1, fp) != 1) return false;
    if (hdr[0] > (1ull << 36) || hdr[1] > (1ull << 36)) return false;
    if (expect_multiple_of && hdr[0] % expect_multiple_of) return false;
    std::vector<uint8_t> comp(hdr[1]);
    if (hdr[1] && fread(comp.data(), 1, hdr[1], fp) != hdr[1]) return false;
    out.resize(hdr[0]);
    if (hdr[0]) {
        uLongf rawlen = (uLongf)hdr[0];
        if (uncompress(out.data(), &rawlen, comp.data(), (uLong)hdr[1]) != Z_OK || rawlen != hdr[0]) return false;
    }
    return true;
}

template <typename T>
bool read_array(FILE* fp, std::vector<T>& out, size_t count) {
    std::vector<uint8_t> raw;
    if (!read_blob(fp, raw, sizeof(T))) return false;
    if (raw.size() != count * sizeof(T)) return false;
    out.resize(count);
    if (count) memcpy(out.data(), raw.data(), raw.size());
    return true;
}

}  // namespace

bool scene_save_file(const HostScene& sc, const char* path, std::string& err) {
    FILE* fp = fopen(path, "wb");
    if (!fp) { err = std::string("cannot write ") + path; return false; }
    bool ok = fwrite("CRTSCN01", 8, 1, fp) == 1;
    const uint32_t counts[5] = {(uint32_t)sc.nodes.size(), (uint32_t)sc.materials.size(), (uint32_t)sc.lights.size(),
                                (uint32_t)sc.textures.size(), (uint32_t)sc.meshes.size()};
    ok = ok && fwrite(counts, sizeof counts, 1, fp) == 1;
    ok = ok && fwrite(&sc.env_texture, 4, 1, fp) == 1;
    ok = ok && write_blob(fp, &sc.camera, sizeof sc.camera);
    ok = ok && write_blob(fp, sc.nodes.data(), sc.nodes.size() * sizeof(crt_node));
    ok = ok && write_blob(fp, sc.materials.data(), sc.materials.size() * sizeof(crt_material));
    ok = ok && write_blob(fp, sc.lights.data(), sc.lights.size() * sizeof(crt_light));
    for (const HostTexture& t : sc.textures) {
        const uint32_t wh[2] = {t.width, t.height};
        ok = ok && fwrite(wh, sizeof wh, 1, fp) == 1;
        ok = ok && write_blob(fp, t.rgb.data(), t.rgb.size());
    }
    for (const HostMesh& m : sc.meshes) {
        const uint32_t n[6] = {m.nv(), m.nf(), m.nvn(), m.nvt(), (uint32_t)(m.fn.size() / 3), (uint32_t)(m.ft.size() / 3)};
        ok = ok && fwrite(n, sizeof n, 1, fp) == 1;
        ok = ok && fwrite(m.box_min, sizeof m.box_min, 1, fp) == 1;
        ok = ok && fwrite(m.box_max, sizeof m.box_max, 1, fp) == 1;
        ok = ok && write_blob(fp, m.v.data(), m.v.size() * 4);
        ok = ok && write_blob(fp, m.f.data(), m.f.size() * 4);
        ok = ok && write_blob(fp, m.vn.data(), m.vn.size() * 4);
        ok = ok && write_blob(fp, m.fn.data(), m.fn.size() * 4);
        ok = ok && write_blob(fp, m.vt.data(), m.vt.size() * 4);
        ok = ok && write_blob(fp, m.ft.data(), m.ft.size() * 4);
    }
    ok = (fclose(fp) == 0) && ok;
    if (!ok) err = std::string("write error on ") + path;
    return ok;
}

bool scene_load_file(const char* path, HostScene& sc, std::string& err) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot open ") + path; return false; }
    sc = HostScene();
    auto fail = [&](const char* what) { err = std::string(path) + ": " + what; fclose(fp); return false; };
    char magic[8];
    if (fread(magic, 8, 1, fp) != 1 || memcmp(magic, "CRTSCN01", 8) != 0) return fail("not a flat scene file");
    uint32_t counts[5];
    if (fread(counts, sizeof counts, 1, fp) != 1) return fail("truncated header");
    if (fread(&sc.env_texture, 4, 1, fp) != 1) return fail("truncated header");
    std::vector<crt_camera> cam;
    if (!read_array(fp, cam, 1)) return fail("bad camera block");
    sc.camera = cam[0];
    if (!read_array(fp, sc.nodes, counts[0])) return fail("bad node block");
    if (!read_array(fp, sc.materials, counts[1])) return fail("bad material block");
    if (!read_array(fp, sc.lights, counts[2])) return fail("bad light block");
    sc.textures.resize(counts[3]);
    for (HostTexture& t : sc.textures) {
        uint32_t wh[2];
        if (fread(wh, sizeof wh, 1, fp) != 1) return fail("truncated texture");
        t.width = wh[0]; t.height = wh[1];
        if (!read_array(fp, t.rgb, (size_t)t.width * t.height * 3)) return fail("bad texture block");
    }
    sc.meshes.resize(counts[4]);
    for (HostMesh& m : sc.meshes) {
        uint32_t n[6];
        if (fread(n, sizeof n, 1, fp) != 1) return fail("truncated mesh");
        if (fread(m.box_min, sizeof m.box_min, 1, fp) != 1 || fread(m.box_max, sizeof m.box_max, 1, fp) != 1) return fail("truncated mesh");
        if (!read_array(fp, m.v, (size_t)n[0] * 3) || !read_array(fp, m.f, (size_t)n[1] * 3) ||
            !read_array(fp, m.vn, (size_t)n[2] * 3) || !read_array(fp, m.fn, (size_t)n[4] * 3) ||
            !read_array(fp, m.vt, (size_t)n[3] * 3) || !read_array(fp, m.ft, (size_t)n[5] * 3))
            return fail("bad mesh block");
    }
    fclose(fp);

    // validate every index the renderer will follow unchecked
    const int nt = (int)sc.textures.size(), nm = (int)sc.meshes.size(), nmat = (int)sc.materials.size();
    if (sc.env_texture < 0 || sc.env_texture >= nt) { err = "environment texture index out of range"; return false; }
    for (const crt_material& m : sc.materials)
        for (int id : {m.diffuse, m.specular, m.refraction, m.emission})
            if (id < 0 || id >= nt) { err = "material texture index out of range"; return false; }
    for (const crt_node& n : sc.nodes) {
        if (n.material_id < 0 || n.material_id >= nmat) { err = "node material index out of range"; return false; }
        if (n.object_kind == CRT_OBJ_MESH && (n.mesh_id < 0 || n.mesh_id >= nm)) { err = "node mesh index out of range"; return false; }
    }
    for (const HostMesh& m : sc.meshes) {
        for (uint32_t x : m.f) if (x >= m.nv()) { err = "mesh vertex index out of range"; return false; }
        for (uint32_t x : m.fn) if (x >= m.nvn()) { err = "mesh normal index out of range"; return false; }
        for (uint32_t x : m.ft) if (x >= m.nvt()) { err = "mesh texture index out of range"; return false; }
        if (m.nvn() && m.fn.size() != m.f.size()) { err = "mesh normal faces missing"; return false; }
        if (m.nvt() && m.ft.size() != m.f.size()) { err = "mesh texture faces missing"; return false; }
    }
    sc.refresh_views();
    return true;
}

}  // namespace crt
