// Wavefront OBJ reader.  Reads what Mesh::LoadFromFileObj reads (scene/mesh.cu:14-200) and arrives at the same
// arrays: records v / vt / vn / f only, everything else ignored; '#' starts a comment line; blanks of any kind and
// number separate fields; polygons become triangle fans around their first corner; a negative index counts back
// from the end of the list as read so far; a face gets a texture / normal index triple only once the file has
// shown a vt / vn record or a corner with that field.
//
// One quirk of the reference decides real data: it holds a line in 1024 bytes, so a record whose text (blank runs
// counted as one character) exceeds 1023 characters is cut there, the cut may fall inside a number, and the rest of
// the line is read as a record of its own, which no keyword matches.  models/final/body_cell_018.obj has such a
// polygon, so the cut is reproduced (next_record).
//
// The file is held in memory; next_record yields one record at a time with its blank runs already reduced, a token
// cursor walks it, and numbers go through strtof.  Deliberate differences, none of which a shipped model meets: a
// record's missing components are 0 (the reference leaves them unset, mesh.cu:54) and every index is range-checked
// at the end because the GPU reads the arrays unchecked.
#include "host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace crt {

namespace {

constexpr size_t kRecordLimit = 1023;     // characters of one record (the reference's line buffer less its terminator)

// End of a line: newline, carriage return, NUL or the end of the buffer.
inline bool ends_line(const char* p, const char* end) { return p == end || *p == '\n' || *p == '\r' || *p == '\0'; }
inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }

// The next record of the file into `rec`: leading white space, empty lines and '#' lines are passed over, every run of
// blanks inside the record becomes one space and trailing blanks vanish.  A record stops at the end of its line or at
// kRecordLimit characters, whichever comes first; in the second case the character after the cut is lost (the
// reference has already taken it from the stream) and the remainder of the line comes back as the next record.
// False once the file has nothing more (a NUL byte also ends the file, as an empty record does in the reference).
bool next_record(const char*& p, const char* end, std::string& rec) {
    for (;;) {
        while (p < end && (is_blank(*p) || *p == '\n' || *p == '\r')) p++;
        if (p < end && *p == '#') { while (!ends_line(p, end)) p++; }
        else break;
    }
    rec.clear();
    bool gap = false;
    while (rec.size() < kRecordLimit) {
        if (ends_line(p, end)) { if (p < end) p++; return !rec.empty(); }
        const char c = *p++;
        if (is_blank(c)) { gap = true; continue; }
        if (gap) rec.push_back(' ');
        gap = false;
        rec.push_back(c);
    }
    if (p < end) p++;
    return true;
}

// The next blank-separated token of the line at *cursor; false when the line has no more.
bool next_token(const char*& cursor, const char* end, const char*& tok, size_t& len) {
    while (!ends_line(cursor, end) && is_blank(*cursor)) cursor++;
    if (ends_line(cursor, end)) return false;
    tok = cursor;
    while (!ends_line(cursor, end) && !is_blank(*cursor)) cursor++;
    len = (size_t)(cursor - tok);
    return true;
}

// Up to three numbers after the record's keyword; parsing stops at the first token that is not a number.
void read_triple(const char*& cursor, const char* end, float out[3]) {
    out[0] = out[1] = out[2] = 0.0f;
    for (int k = 0; k < 3; k++) {
        const char* tok; size_t len;
        if (!next_token(cursor, end, tok, len)) return;
        char text[64];
        if (len >= sizeof text) return;
        memcpy(text, tok, len); text[len] = '\0';
        char* stop = nullptr;
        const float v = strtof(text, &stop);
        if (stop == text) return;
        out[k] = v;
    }
}

// One corner of a face, "v", "v/vt", "v//vn" or "v/vt/vn".  A field counts as present once it has a digit; its value is
// the decimal number made of the field's digits (other characters are skipped), one-based, or counted back from
// `counts[field]` when a '-' has appeared anywhere earlier in the corner (the sign carries over to later fields).
struct Corner {
    bool present[3] = {false, false, false};
    uint32_t index[3] = {0, 0, 0};
};

Corner parse_corner(const char* tok, size_t len, const uint32_t counts[3]) {
    Corner c;
    int field = 0;
    bool from_end = false;
    uint32_t number = 0;
    for (size_t i = 0; i < len; i++) {
        const char ch = tok[i];
        if (ch == '/') { field++; number = 0; }
        else if (ch == '-') from_end = true;
        else if (ch >= '0' && ch <= '9') {
            number = number * 10u + (uint32_t)(ch - '0');
            if (field < 3) {
                c.present[field] = true;
                c.index[field] = from_end ? counts[field] - number : number - 1u;      // (unsigned wrap on 0: caught by the range check)
            }
        }
    }
    return c;
}

}  // namespace

bool obj_load(const char* path, HostMesh& mesh, std::string& err) {
    std::vector<char> text;
    {
        FILE* fp = fopen(path, "rb");
        if (!fp) { err = std::string("cannot open ") + path; return false; }
        char chunk[1 << 16];
        size_t got;
        while ((got = fread(chunk, 1, sizeof chunk, fp)) > 0) text.insert(text.end(), chunk, chunk + got);
        fclose(fp);
    }
    mesh = HostMesh();
    mesh.name = path;

    // whether faces carry texture / normal index triples yet
    bool with_tex = false, with_normals = false;
    const char* file_cursor = text.data();
    const char* const file_end = text.data() + text.size();
    std::string record;
    while (next_record(file_cursor, file_end, record)) {
        const char* cursor = record.data();
        const char* const end = record.data() + record.size();
        const char* key; size_t key_len;
        if (next_token(cursor, end, key, key_len)) {
            const bool is_v = key_len == 1 && key[0] == 'v';
            const bool is_vt = key_len == 2 && key[0] == 'v' && key[1] == 't';
            const bool is_vn = key_len == 2 && key[0] == 'v' && key[1] == 'n';
            const bool is_f = key_len == 1 && key[0] == 'f';
            if (is_v || is_vt || is_vn) {
                float p[3];
                read_triple(cursor, end, p);
                std::vector<float>& list = is_v ? mesh.v : (is_vt ? mesh.vt : mesh.vn);
                list.insert(list.end(), p, p + 3);
                with_tex = with_tex || is_vt;
                with_normals = with_normals || is_vn;
            } else if (is_f) {
                // tri[field][slot]: the triangle being assembled, per index kind (0 vertex, 1 texture, 2 normal)
                uint32_t tri[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                auto emit = [&]() {
                    mesh.f.insert(mesh.f.end(), tri[0], tri[0] + 3);
                    if (with_tex) mesh.ft.insert(mesh.ft.end(), tri[1], tri[1] + 3);
                    if (with_normals) mesh.fn.insert(mesh.fn.end(), tri[2], tri[2] + 3);
                };
                const uint32_t counts[3] = {mesh.nv(), mesh.nvt(), mesh.nvn()};
                int corners = 0;
                const char* tok; size_t len;
                while (next_token(cursor, end, tok, len)) {
                    int slot = corners;
                    if (corners >= 3) {
                        // a further corner of a polygon: the triangle so far is done; the fan keeps corner 0 and the
                        // last corner becomes the middle one
                        emit();
                        for (auto& kind : tri) kind[1] = kind[2];
                        slot = 2;
                    }
                    corners++;
                    const Corner c = parse_corner(tok, len, counts);
                    for (int kind = 0; kind < 3; kind++)
                        if (c.present[kind]) tri[kind][slot] = c.index[kind];      // (an absent field keeps what the slot held)
                    with_tex = with_tex || c.present[1];
                    with_normals = with_normals || c.present[2];
                }
                emit();
            }
        }
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
