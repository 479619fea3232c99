// Scene XML -> flat scene.  Own implementation of the schema the reference
// reads (behaviour of lib/xmlload.cu:153-406 and lib/xmlload.cuh:59-127):
//
//   <xml><scene> object* material* light* environment? </scene><camera/></xml>
//   object[type=sphere|plane|obj, name, material]{scale, rotate[angle], translate, object*}
//   material[name]{diffuse, specular, glossiness, refraction[index], emission, absorption}
//   light[type=ambient|direct|point]{intensity, direction, position, size}
//   camera[gamma]{position, target, up, fov, width, height, dof}
//
// Vector elements read x/y/z (colours r/g/b) and a "value" attribute scales all
// three.  Per object the last scale / rotate / translate child wins and the
// node matrix is parent * T * R * S whatever their order in the file.  Every
// colour or texture of a material becomes an 8-bit texture.
#include <cctype>
#include <cstdio>
#include <cstring>
#include <map>

#include "host.h"
#include "xml.h"

namespace crt {

namespace {

constexpr float kDeg2Rad = 0.017453292519943296f;

bool ieq(const std::string& a, const char* b) {
    size_t i = 0;
    for (; i < a.size() && b[i]; i++)
        if (tolower((unsigned char)a[i]) != tolower((unsigned char)b[i])) return false;
    return i == a.size() && b[i] == '\0';
}

// element may be null: then the value is left untouched (Loader::ReadFloat3, xmlload.cuh:61)
void read_float3(const XmlElement* e, f3& v, f3 def = mk3(0, 0, 0)) {
    if (!e) return;
    v = def;
    e->float_attr("x", v.x); e->float_attr("y", v.y); e->float_attr("z", v.z);
    float f = 1;
    if (e->float_attr("value", f)) v *= f;
}

void read_color(const XmlElement* e, f3& c, f3 def = mk3(1, 1, 1)) {
    if (!e) return;
    c = def;
    e->float_attr("r", c.x); e->float_attr("g", c.y); e->float_attr("b", c.z);
    float f = 1;
    if (e->float_attr("value", f)) c *= f;
}

// float -> int as the host conversion the reference's texture constructor runs
// under (Color24, math/color.cuh:31-42, called from scene/texture.cu:5-8):
// out-of-range and NaN give INT_MIN, which then clamps to 0.
uint8_t quantise(float c) {
    const float f = c * 256;
    int v;
    if (!(f > -2147483648.0f && f < 2147483648.0f)) v = (int)0x80000000;
    else v = (int)f;
    return v < 0 ? 0 : (v > 255 ? 255 : (uint8_t)v);
}

std::string join_path(const char* root, const char* rel) {
    if (!root || !root[0] || rel[0] == '/') return rel;
    std::string p = root;
    if (p.back() != '/') p += '/';
    return p + rel;
}

struct Loader {
    HostScene& sc;
    const char* asset_root;
    std::string& err;
    std::map<std::string, int> material_ids;   // first definition of a name wins
    std::map<std::string, int> mesh_ids;

    // Loader::ReadTexture (xmlload.cuh:94-127)
    int make_texture(const XmlElement* e, f3 def) {
        HostTexture t;
        if (!e) {
            t.width = t.height = 1;
            t.rgb = {quantise(def.x), quantise(def.y), quantise(def.z)};
        } else {
            std::vector<uint8_t> bytes;
            uint32_t w = 0, h = 0;
            bool loaded = false;
            const char* name = e->attr("texture");
            if (name && strlen(name) >= 3) {
                const size_t len = strlen(name);
                const char ext[3] = {(char)tolower(name[len - 3]), (char)tolower(name[len - 2]), (char)tolower(name[len - 1])};
                if (strncmp(ext, "png", 3) == 0) {
                    std::string perr;
                    loaded = png_read_rgb8(join_path(asset_root, name).c_str(), bytes, w, h, perr);
                }
            }
            if (!loaded) { w = h = 1; bytes = {0, 0, 0}; }
            f3 mult = mk3(1, 1, 1);
            e->float_attr("r", mult.x); e->float_attr("g", mult.y); e->float_attr("b", mult.z);
            float f = 1;
            if (e->float_attr("value", f)) mult *= f;
            t.width = w; t.height = h;
            t.rgb.resize((size_t)w * h * 3);
            constexpr float over256 = 1.0f / 256.0f;
            for (size_t i = 0; i < (size_t)w * h; i++) {
                f3 c = loaded ? mk3((float)bytes[3 * i], (float)bytes[3 * i + 1], (float)bytes[3 * i + 2]) * over256
                              : mk3(1, 1, 1);
                c *= mult;
                t.rgb[3 * i] = quantise(c.x); t.rgb[3 * i + 1] = quantise(c.y); t.rgb[3 * i + 2] = quantise(c.z);
            }
        }
        sc.textures.push_back(std::move(t));
        return (int)sc.textures.size() - 1;
    }

    void load_material(const XmlElement& e) {
        crt_material m{};
        m.glossiness = 20; m.ior = 1;
        m.absorption[0] = m.absorption[1] = m.absorption[2] = 0;
        m.diffuse = make_texture(e.child("diffuse"), mk3(1, 1, 1));
        m.specular = make_texture(e.child("specular"), mk3(1, 1, 1));
        if (const XmlElement* g = e.child("glossiness")) g->float_attr("value", m.glossiness);
        m.refraction = make_texture(e.child("refraction"), mk3(0, 0, 0));
        if (const XmlElement* r = e.child("refraction")) r->float_attr("index", m.ior);
        m.emission = make_texture(e.child("emission"), mk3(0, 0, 0));
        f3 ab = mk3(0, 0, 0);
        read_color(e.child("absorption"), ab);
        m.absorption[0] = ab.x; m.absorption[1] = ab.y; m.absorption[2] = ab.z;
        const char* name = e.attr("name");
        sc.materials.push_back(m);
        material_ids.emplace(name ? name : "", (int)sc.materials.size() - 1);
    }

    void load_light(const XmlElement& e) {
        crt_light l{};
        l.kind = CRT_LIGHT_AMBIENT;
        l.size = 1;
        const char* type = e.attr("type");
        const std::string ty = type ? type : "";
        f3 intensity = mk3(0, 0, 0);
        if (ieq(ty, "ambient")) {
            l.kind = CRT_LIGHT_AMBIENT;
            read_color(e.child("intensity"), intensity);
        } else if (ieq(ty, "direct")) {
            l.kind = CRT_LIGHT_DIRECT;
            read_color(e.child("intensity"), intensity);
            f3 d = mk3(0, 1, 0);
            read_float3(e.child("direction"), d);
            d = normalized(d);
            l.vec[0] = d.x; l.vec[1] = d.y; l.vec[2] = d.z;
        } else if (ieq(ty, "point")) {
            l.kind = CRT_LIGHT_POINT;
            read_color(e.child("intensity"), intensity);
            f3 p = mk3(0, 0, 0);
            read_float3(e.child("position"), p);
            l.vec[0] = p.x; l.vec[1] = p.y; l.vec[2] = p.z;
            if (const XmlElement* s = e.child("size")) s->float_attr("value", l.size);
        } else {
            fprintf(stderr, "crt: unknown light type %s (kept as a black ambient light)\n", ty.c_str());
        }
        l.intensity[0] = intensity.x; l.intensity[1] = intensity.y; l.intensity[2] = intensity.z;
        sc.lights.push_back(l);
    }

    static void mat_mul(const mat34& L, const mat34& R, mat34& P) {
        const float* l = L.c; const float* r = R.c; float* p = P.c;
        for (int col = 0; col < 3; col++) {
            p[3 * col + 0] = l[0] * r[3 * col] + l[3] * r[3 * col + 1] + l[6] * r[3 * col + 2];
            p[3 * col + 1] = l[1] * r[3 * col] + l[4] * r[3 * col + 1] + l[7] * r[3 * col + 2];
            p[3 * col + 2] = l[2] * r[3 * col] + l[5] * r[3 * col + 1] + l[8] * r[3 * col + 2];
        }
        p[9] = l[0] * r[9] + l[3] * r[10] + l[6] * r[11] + l[9];
        p[10] = l[1] * r[9] + l[4] * r[10] + l[7] * r[11] + l[10];
        p[11] = l[2] * r[9] + l[5] * r[10] + l[8] * r[11] + l[11];
    }

    // axis-angle rotation (Matrix::Rotation, math/matrix.cuh:138-164)
    static mat34 mat_rotation(f3 axis, float degrees) {
        mat34 m = mat_identity();
        const float rad = kDeg2Rad * degrees;
        // host code in the reference: the C library's sinf / cosf, not the pinned device replacements
        const float sina = sinf(rad);
        const float cosa = cosf(rad);
        const float t = 1.0f - cosa;
        const f3 a = t * axis;
        const float txy = a.x * axis.y, txz = a.x * axis.z, tyz = a.y * axis.z;
        const f3 s = sina * axis;
        m.c[0] = a.x * axis.x + cosa; m.c[1] = txy + s.z; m.c[2] = txz - s.y;
        m.c[3] = txy - s.z; m.c[4] = a.y * axis.y + cosa; m.c[5] = tyz + s.x;
        m.c[6] = txz + s.y; m.c[7] = tyz - s.x; m.c[8] = a.z * axis.z + cosa;
        return m;
    }

    size_t count_nodes(const XmlElement& e) {
        size_t total = 1;
        for (const auto& c : e.children)
            if (ieq(c->tag, "object")) total += count_nodes(*c);
        return total;
    }

    bool assign_nodes(const XmlElement& e, const mat34& parent) {
        const size_t my = sc.nodes.size();
        sc.nodes.emplace_back();
        {
            crt_node& n = sc.nodes[my];
            memset(&n, 0, sizeof n);
            n.object_kind = CRT_OBJ_NONE; n.mesh_id = -1; n.material_id = 0;
        }
        const char* type = e.attr("type");
        if (type) {
            const std::string ty = type;
            if (ieq(ty, "sphere")) sc.nodes[my].object_kind = CRT_OBJ_SPHERE;
            else if (ieq(ty, "plane")) sc.nodes[my].object_kind = CRT_OBJ_PLANE;
            else if (ieq(ty, "obj")) {
                const char* name = e.attr("name");
                const std::string key = name ? name : "";
                auto it = mesh_ids.find(key);
                if (it != mesh_ids.end()) {
                    sc.nodes[my].object_kind = CRT_OBJ_MESH; sc.nodes[my].mesh_id = it->second;
                } else {
                    HostMesh mesh;
                    std::string merr;
                    if (!name || !obj_load(join_path(asset_root, name).c_str(), mesh, merr)) {
                        fprintf(stderr, "crt: cannot load mesh \"%s\": %s (node left without an object)\n", key.c_str(), merr.c_str());
                    } else {
                        mesh.name = key;
                        sc.meshes.push_back(std::move(mesh));
                        const int id = (int)sc.meshes.size() - 1;
                        mesh_ids.emplace(key, id);
                        sc.nodes[my].object_kind = CRT_OBJ_MESH; sc.nodes[my].mesh_id = id;
                    }
                }
            } else {
                fprintf(stderr, "crt: unknown object type %s\n", type);
            }
        }
        if (const char* mname = e.attr("material")) {
            auto it = material_ids.find(mname);
            if (it == material_ids.end()) { err = std::string("material \"") + mname + "\" is not defined"; return false; }
            sc.nodes[my].material_id = it->second;
        }

        // Local transform: the node's last <translate>, last <rotate> and last <scale> element, always composed as
        // parent x T x R x S whatever their order in the file (lib/xmlload.cu:274-294).
        enum { kTranslate, kRotate, kScale };
        mat34 part[3] = {mat_identity(), mat_identity(), mat_identity()};
        for (const auto& c : e.children) {
            f3 value;
            if (ieq(c->tag, "translate")) {
                read_float3(c.get(), value);
                part[kTranslate] = mat_identity();
                part[kTranslate].c[9] = value.x; part[kTranslate].c[10] = value.y; part[kTranslate].c[11] = value.z;
            } else if (ieq(c->tag, "rotate")) {
                read_float3(c.get(), value);                      // the axis
                float degrees = 0.0f;
                c->float_attr("angle", degrees);
                part[kRotate] = mat_rotation(normalized(value), degrees);
            } else if (ieq(c->tag, "scale")) {
                read_float3(c.get(), value, mk3(1, 1, 1));
                part[kScale] = mat_identity();
                part[kScale].c[0] = value.x; part[kScale].c[4] = value.y; part[kScale].c[8] = value.z;
            }
        }
        mat34 tm = parent;
        for (const mat34& m : part) { mat34 product; mat_mul(tm, m, product); tm = product; }
        const mat34 itm = mat_inverse(tm);
        memcpy(sc.nodes[my].tm, tm.c, sizeof tm.c);
        memcpy(sc.nodes[my].itm, itm.c, sizeof itm.c);

        for (const auto& c : e.children)
            if (ieq(c->tag, "object")) {
                sc.nodes[my].child_count++;
                if (!assign_nodes(*c, tm)) return false;
            }
        return true;
    }

    bool load_scene(const XmlElement& xscene) {
        // material 0 is the default: white diffuse / specular, black refraction / emission
        crt_material def{};
        def.glossiness = 20; def.ior = 1;
        def.diffuse = make_texture(nullptr, mk3(1, 1, 1));
        def.specular = make_texture(nullptr, mk3(1, 1, 1));
        def.refraction = make_texture(nullptr, mk3(0, 0, 0));
        def.emission = make_texture(nullptr, mk3(0, 0, 0));
        sc.materials.push_back(def);
        material_ids.emplace("__DEFAULT_MTL__", 0);

        for (const auto& c : xscene.children) {
            if (ieq(c->tag, "material")) load_material(*c);
            else if (ieq(c->tag, "light")) load_light(*c);
        }

        // node 0: object-less root with an identity transform
        sc.nodes.emplace_back();
        memset(&sc.nodes[0], 0, sizeof(crt_node));
        sc.nodes[0].mesh_id = -1;
        const mat34 ident = mat_identity();
        memcpy(sc.nodes[0].tm, ident.c, sizeof ident.c);
        memcpy(sc.nodes[0].itm, ident.c, sizeof ident.c);
        for (const auto& c : xscene.children)
            if (ieq(c->tag, "object")) {
                sc.nodes[0].child_count++;
                if (!assign_nodes(*c, ident)) return false;
            }
        scene_compute_boxes(sc);
        sc.env_texture = make_texture(xscene.child("environment"), mk3(0, 0, 0));
        return true;
    }

    bool load_camera(const XmlElement& xcam) {
        crt_camera& cam = sc.camera;
        memset(&cam, 0, sizeof cam);
        f3 pos = mk3(0, 0, 0), target = mk3(0, 0, 0), up = mk3(0, 0, 0);
        read_float3(xcam.child("position"), pos);
        read_float3(xcam.child("target"), target);
        read_float3(xcam.child("up"), up);
        if (const XmlElement* e = xcam.child("fov")) e->float_attr("value", cam.fov);
        if (const XmlElement* e = xcam.child("dof")) e->float_attr("value", cam.dof);
        const char* gamma = xcam.attr("gamma");
        cam.srgb = (gamma && strcmp(gamma, "sRGB") == 0) ? 1 : 0;
        int w = 0, h = 0;
        const XmlElement* ew = xcam.child("width");
        const XmlElement* eh = xcam.child("height");
        if (!ew || !eh || !ew->int_attr("value", w) || !eh->int_attr("value", h) || w <= 0 || h <= 0) {
            err = "camera needs positive width and height"; return false;
        }
        cam.width = (uint32_t)w; cam.height = (uint32_t)h;
        cam.position[0] = pos.x; cam.position[1] = pos.y; cam.position[2] = pos.z;
        cam.target[0] = target.x; cam.target[1] = target.y; cam.target[2] = target.z;
        cam.up[0] = up.x; cam.up[1] = up.y; cam.up[2] = up.z;
        camera_derive(cam);
        return true;
    }
};

}  // namespace

// adjugate / determinant inverse with the operation order of Matrix::GetInverse
// (math/matrix.cuh:25-56)
mat34 mat_inverse(const mat34& M) {
    const float* c = M.c;
    mat34 inv;
    float* o = inv.c;
    const float d_a = c[8] * c[10] - c[7] * c[11];
    const float d_b = c[6] * c[11] - c[8] * c[9];
    const float d_c = c[7] * c[9] - c[6] * c[10];
    o[0] = c[4] * c[8] - c[5] * c[7];
    o[1] = c[2] * c[7] - c[1] * c[8];
    o[2] = c[1] * c[5] - c[2] * c[4];
    o[3] = c[5] * c[6] - c[3] * c[8];
    o[4] = c[0] * c[8] - c[2] * c[6];
    o[5] = c[2] * c[3] - c[0] * c[5];
    o[6] = c[3] * c[7] - c[4] * c[6];
    o[7] = c[1] * c[6] - c[0] * c[7];
    o[8] = c[0] * c[4] - c[1] * c[3];
    o[9] = c[3] * d_a + c[4] * d_b + c[5] * d_c;
    o[10] = c[0] * -d_a + c[1] * -d_b + c[2] * -d_c;
    o[11] = c[0] * (c[5] * c[10] - c[4] * c[11]) + c[1] * (c[3] * c[11] - c[5] * c[9]) +
            c[2] * (c[4] * c[9] - c[3] * c[10]);
    const float det = c[0] * o[0] + c[1] * o[3] + c[2] * o[6];
    const float s = 1.0f / det;
    for (int i = 0; i < 12; i++) o[i] = o[i] * s;
    return inv;
}


struct Box { f3 lo, hi; };
static void box_init(Box& b) { b.lo = mk3(CRT_BIGFLOAT, CRT_BIGFLOAT, CRT_BIGFLOAT); b.hi = mk3(-CRT_BIGFLOAT, -CRT_BIGFLOAT, -CRT_BIGFLOAT); }
static void box_add(Box& b, f3 p) {
    if (b.lo.x > p.x) b.lo.x = p.x; if (b.lo.y > p.y) b.lo.y = p.y; if (b.lo.z > p.z) b.lo.z = p.z;
    if (b.hi.x < p.x) b.hi.x = p.x; if (b.hi.y < p.y) b.hi.y = p.y; if (b.hi.z < p.z) b.hi.z = p.z;
}

// Node::CalculateBoundingBox (scene/scene.cuh:191-251), run from the last node to the first
void scene_compute_boxes(HostScene& sc) {
    std::vector<Box> boxes(sc.nodes.size());
    for (int i = (int)sc.nodes.size() - 1; i >= 0; i--) {
        crt_node& n = sc.nodes[i];
        Box b; box_init(b);
        if (n.object_kind != CRT_OBJ_NONE) {
            f3 lo, hi;
            if (n.object_kind == CRT_OBJ_SPHERE) { lo = mk3(-1, -1, -1); hi = mk3(1, 1, 1); }
            else if (n.object_kind == CRT_OBJ_PLANE) { lo = mk3(-1, -1, 0); hi = mk3(1, 1, 0); }
            else {
                const HostMesh& m = sc.meshes[n.mesh_id];
                lo = mk3(m.box_min[0], m.box_min[1], m.box_min[2]);
                hi = mk3(m.box_max[0], m.box_max[1], m.box_max[2]);
            }
            Box t; box_init(t);
            for (int k = 0; k < 8; k++) {
                const f3 corner = mk3((k & 4) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 1) ? hi.z : lo.z);
                box_add(t, xform_point(n.tm, corner));
            }
            box_add(b, t.lo); box_add(b, t.hi);
        }
        unsigned children = n.child_count, skip = 0;
        for (unsigned j = (unsigned)i + 1; children > 0 && j < sc.nodes.size(); j++) {
            if (skip > 0) { skip--; skip += sc.nodes[j].child_count; }
            else { box_add(b, boxes[j].lo); box_add(b, boxes[j].hi); skip = sc.nodes[j].child_count; children--; }
        }
        boxes[i] = b;
        n.box_min[0] = b.lo.x; n.box_min[1] = b.lo.y; n.box_min[2] = b.lo.z;
        n.box_max[0] = b.hi.x; n.box_max[1] = b.hi.y; n.box_max[2] = b.hi.z;
    }
}


// RenderInfo::Load (lib/xmlload.cu:324-354)
void camera_derive(crt_camera& cam) {
    const f3 pos = mk3(cam.position[0], cam.position[1], cam.position[2]);
    const f3 target = mk3(cam.target[0], cam.target[1], cam.target[2]);
    const f3 up = mk3(cam.up[0], cam.up[1], cam.up[2]);
    const float aspect = (float)cam.width / (float)cam.height;
    const float fov_rad = cam.fov * kDeg2Rad;
    const float focal = length(target - pos);
    const f3 cZ = normalized(target - pos);
    const f3 cX = cross_literal(cZ, normalized(up));
    const f3 cY = cross_literal(cX, cZ);
    const float plane_h = 2 * focal * tanf(fov_rad * 0.5f);      // host code in the reference: the C library's tanf
    const float plane_w = plane_h * aspect;
    cam.pixel_size = plane_h / (float)cam.height;
    const f3 centre = pos + focal * cZ;
    const f3 corner = centre - (plane_w * 0.5f * cX) + (plane_h * 0.5f * cY);
    const f3 tl = corner + cam.pixel_size * 0.5f * (cX - cY);
    cam.cX[0] = cX.x; cam.cX[1] = cX.y; cam.cX[2] = cX.z;
    cam.cY[0] = cY.x; cam.cY[1] = cY.y; cam.cY[2] = cY.z;
    cam.cZ[0] = cZ.x; cam.cZ[1] = cZ.y; cam.cZ[2] = cZ.z;
    cam.top_left_pixel[0] = tl.x; cam.top_left_pixel[1] = tl.y; cam.top_left_pixel[2] = tl.z;
}

bool scene_load_xml(const char* xml_path, const char* asset_root, HostScene& scene, std::string& err) {
    std::string perr;
    auto doc = xml_parse_file(xml_path, perr);
    if (!doc) { err = std::string("cannot read ") + xml_path + ": " + perr; return false; }
    const XmlElement* xml = doc->child("xml");
    if (!xml) { err = std::string(xml_path) + ": the document has no <xml> root element"; return false; }
    const XmlElement* xscene = xml->child("scene");
    if (!xscene) { err = std::string(xml_path) + ": <xml> holds no <scene> element"; return false; }
    const XmlElement* xcam = xml->child("camera");
    if (!xcam) { err = std::string(xml_path) + ": <xml> holds no <camera> element"; return false; }
    scene = HostScene();
    Loader ld{scene, asset_root, err, {}, {}};
    if (!ld.load_scene(*xscene)) return false;
    if (!ld.load_camera(*xcam)) return false;
    scene.refresh_views();
    return true;
}

}  // namespace crt
