// Wavefront OBJ reader with the reference loader's behaviour
// (Mesh::LoadFromFileObj, scene/mesh.cu:14-200): commands v / vt / vn / f only,
// whitespace runs collapse to one blank, '#' lines are skipped, polygons are
// fan-triangulated, negative indices count from the end of the list read so
// far, and index triples for vt / vn are recorded for a face only once a vt /
// vn record or index has been seen.
#include "host.h"

#include <cctype>
#include <cstdio>
#include <cstring>

namespace crt {

namespace {

struct LineReader {
    const std::vector<uint8_t>& d;
    size_t pos = 0;
    bool eof = false;            // set once a read went past the end (feof semantics)
    char line[1100];
    explicit LineReader(const std::vector<uint8_t>& data) : d(data) {}

    int get() {
        if (pos < d.size()) return d[pos++];
        eof = true;
        return -1;
    }

    // Returns the length of the next non-empty, non-comment line (0 at end of file).
    int read_line() {
        int c = get();
        while (!eof) {
            while (c >= 0 && isspace(c)) c = get();
            if (c == '#') { while (!eof && c != '\n' && c != '\r' && c != '\0') c = get(); }
            else break;
        }
        int i = 0;
        bool inspace = false;
        while (i < 1024 - 1) {
            if (eof || c == '\n' || c == '\r' || c == '\0') break;
            if (isspace(c)) inspace = true;
            else {
                if (inspace) line[i++] = ' ';
                inspace = false;
                line[i++] = (char)c;
            }
            c = get();
        }
        line[i] = '\0';
        return i;
    }

    bool is_command(const char* cmd) const {
        int i = 0;
        while (cmd[i] != '\0') { if (cmd[i] != line[i]) return false; i++; }
        return line[i] == '\0' || line[i] == ' ';
    }

    // Components the line does not supply are zero (the reference leaves them
    // uninitialised, mesh.cu:54; only the unused third texture coordinate is affected
    // in the shipped models).
    void read_float3(float out[3]) const {
        out[0] = out[1] = out[2] = 0.0f;
        if (strlen(line) >= 2) sscanf(line + 2, "%f %f %f", &out[0], &out[1], &out[2]);
    }
};

}  // namespace

bool obj_load(const char* path, HostMesh& mesh, std::string& err) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot open ") + path; return false; }
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(fp);

    mesh = HostMesh();
    mesh.name = path;
    bool has_normals = false, has_tex = false;
    LineReader rd(data);
    while (int rb = rd.read_line()) {
        if (rd.is_command("v")) {
            float p[3]; rd.read_float3(p);
            mesh.v.insert(mesh.v.end(), p, p + 3);
        } else if (rd.is_command("vt")) {
            float p[3]; rd.read_float3(p);
            mesh.vt.insert(mesh.vt.end(), p, p + 3);
            has_tex = true;
        } else if (rd.is_command("vn")) {
            float p[3]; rd.read_float3(p);
            mesh.vn.insert(mesh.vn.end(), p, p + 3);
            has_normals = true;
        } else if (rd.is_command("f")) {
            int facevert = -1;
            bool inspace = true, negative = false;
            int type = 0;
            unsigned index = 0;
            uint32_t face[3] = {0, 0, 0}, tface[3] = {0, 0, 0}, nface[3] = {0, 0, 0};
            auto push = [&]() {
                mesh.f.insert(mesh.f.end(), face, face + 3);
                if (has_tex) mesh.ft.insert(mesh.ft.end(), tface, tface + 3);
                if (has_normals) mesh.fn.insert(mesh.fn.end(), nface, nface + 3);
            };
            for (int i = 2; i < rb; i++) {
                const char ch = rd.line[i];
                if (ch == ' ') { inspace = true; continue; }
                if (inspace) {
                    inspace = false; negative = false; type = 0; index = 0;
                    if (facevert < 2) {
                        facevert++;
                    } else {
                        // fourth and later corner: emit the previous triangle, keep corner 0 and
                        // slide the last corner into slot 1 (fan triangulation)
                        push();
                        face[1] = face[2]; tface[1] = tface[2]; nface[1] = nface[2];
                    }
                }
                if (ch == '/') { type++; index = 0; }
                if (ch == '-') negative = true;
                if (ch >= '0' && ch <= '9') {
                    index = index * 10 + (unsigned)(ch - '0');
                    switch (type) {
                        case 0: face[facevert] = negative ? mesh.nv() - index : index - 1; break;
                        case 1: tface[facevert] = negative ? mesh.nvt() - index : index - 1; has_tex = true; break;
                        case 2: nface[facevert] = negative ? mesh.nvn() - index : index - 1; has_normals = true; break;
                        default: break;
                    }
                }
            }
            push();
        }
        if (rd.eof) break;
    }

    // every index must address its list; the GPU reads them unchecked
    const uint32_t nv = mesh.nv(), nvn = mesh.nvn(), nvt = mesh.nvt();
    for (uint32_t x : mesh.f) if (x >= nv) { err = std::string("vertex index out of range in ") + path; return false; }
    if (nvn) {
        if (mesh.fn.size() != mesh.f.size()) { err = std::string("normal index count differs from face count in ") + path; return false; }
        for (uint32_t x : mesh.fn) if (x >= nvn) { err = std::string("normal index out of range in ") + path; return false; }
    }
    if (nvt) {
        if (mesh.ft.size() != mesh.f.size()) { err = std::string("texture index count differs from face count in ") + path; return false; }
        for (uint32_t x : mesh.ft) if (x >= nvt) { err = std::string("texture index out of range in ") + path; return false; }
    }

    // bounding box over every vertex (Mesh::ComputeBoundingBox, mesh.cu:7-11)
    for (int k = 0; k < 3; k++) { mesh.box_min[k] = CRT_BIGFLOAT; mesh.box_max[k] = -CRT_BIGFLOAT; }
    for (uint32_t i = 0; i < nv; i++)
        for (int k = 0; k < 3; k++) {
            const float x = mesh.v[3 * i + k];
            if (mesh.box_min[k] > x) mesh.box_min[k] = x;
            if (mesh.box_max[k] < x) mesh.box_max[k] = x;
        }
    return true;
}

}  // namespace crt
