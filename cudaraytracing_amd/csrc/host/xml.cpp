#include "xml.h"

#include <cctype>
#include <cstdio>
#include <cstring>

namespace crt {

const char* XmlElement::attr(const char* name) const {
    for (const auto& a : attrs)
        if (a.first == name) return a.second.c_str();
    return nullptr;
}

const XmlElement* XmlElement::child(const char* name) const {
    for (const auto& c : children)
        if (c->tag == name) return c.get();
    return nullptr;
}

// The reference reads numbers through tinyxml2's QueryFloatAttribute /
// QueryIntAttribute (lib/xmlload.cuh:59-60), i.e. a scanf conversion that
// accepts a numeric prefix.
bool XmlElement::float_attr(const char* name, float& out) const {
    const char* s = attr(name);
    if (!s) return false;
    float v;
    if (sscanf(s, "%f", &v) != 1) return false;
    out = v;
    return true;
}

bool XmlElement::int_attr(const char* name, int& out) const {
    const char* s = attr(name);
    if (!s) return false;
    const char* p = s;
    while (*p && isspace((unsigned char)*p)) p++;
    int v;
    if (p[0] == '0' && (p[1] == 'x' || p[1] == 'X')) {
        unsigned u;
        if (sscanf(p + 2, "%x", &u) != 1) return false;
        v = (int)u;
    } else if (sscanf(p, "%d", &v) != 1) {
        return false;
    }
    out = v;
    return true;
}

namespace {

struct Parser {
    const std::string& s;
    size_t i = 0;
    std::string err;
    explicit Parser(const std::string& text) : s(text) {}

    bool starts(const char* lit) const { return s.compare(i, strlen(lit), lit) == 0; }
    void skip_ws() { while (i < s.size() && isspace((unsigned char)s[i])) i++; }

    bool fail(const std::string& msg) {
        if (err.empty()) {
            size_t line = 1;
            for (size_t k = 0; k < i && k < s.size(); k++) if (s[k] == '\n') line++;
            err = msg + " (line " + std::to_string(line) + ")";
        }
        return false;
    }

    static bool name_char(char c) {
        return isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.';
    }

    std::string decode(const std::string& raw) {
        std::string out;
        for (size_t k = 0; k < raw.size(); k++) {
            if (raw[k] != '&') { out += raw[k]; continue; }
            const size_t semi = raw.find(';', k);
            if (semi == std::string::npos) { out += raw[k]; continue; }
            const std::string ent = raw.substr(k + 1, semi - k - 1);
            if (ent == "amp") out += '&';
            else if (ent == "lt") out += '<';
            else if (ent == "gt") out += '>';
            else if (ent == "quot") out += '"';
            else if (ent == "apos") out += '\'';
            else if (!ent.empty() && ent[0] == '#') {
                const int code = (ent.size() > 1 && (ent[1] == 'x' || ent[1] == 'X'))
                                     ? (int)strtol(ent.c_str() + 2, nullptr, 16)
                                     : (int)strtol(ent.c_str() + 1, nullptr, 10);
                if (code > 0 && code < 128) out += (char)code;
            } else { out += raw.substr(k, semi - k + 1); }
            k = semi;
        }
        return out;
    }

    // Skips comments, declarations, doctype and text; stops at '<' of an element
    // start or end tag, or at end of input.
    bool skip_misc() {
        for (;;) {
            while (i < s.size() && s[i] != '<') i++;   // text content is ignored
            if (i >= s.size()) return true;
            if (starts("<!--")) {
                const size_t e = s.find("-->", i + 4);
                if (e == std::string::npos) return fail("unterminated comment");
                i = e + 3;
            } else if (starts("<?")) {
                const size_t e = s.find("?>", i + 2);
                if (e == std::string::npos) return fail("unterminated declaration");
                i = e + 2;
            } else if (starts("<![CDATA[")) {
                const size_t e = s.find("]]>", i + 9);
                if (e == std::string::npos) return fail("unterminated CDATA");
                i = e + 3;
            } else if (starts("<!")) {
                const size_t e = s.find('>', i + 2);
                if (e == std::string::npos) return fail("unterminated <! block");
                i = e + 1;
            } else {
                return true;
            }
        }
    }

    bool parse_element(XmlElement& parent) {
        // s[i] == '<' and next is a name char
        i++;
        auto el = std::make_unique<XmlElement>();
        const size_t n0 = i;
        while (i < s.size() && name_char(s[i])) i++;
        if (i == n0) return fail("expected element name");
        el->tag = s.substr(n0, i - n0);
        for (;;) {
            skip_ws();
            if (i >= s.size()) return fail("unterminated start tag");
            if (s[i] == '/') {
                if (i + 1 < s.size() && s[i + 1] == '>') { i += 2; parent.children.push_back(std::move(el)); return true; }
                return fail("stray '/' in tag");
            }
            if (s[i] == '>') { i++; break; }
            const size_t a0 = i;
            while (i < s.size() && name_char(s[i])) i++;
            if (i == a0) return fail("expected attribute name");
            std::string aname = s.substr(a0, i - a0);
            skip_ws();
            if (i >= s.size() || s[i] != '=') return fail("expected '=' after attribute name");
            i++;
            skip_ws();
            if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) return fail("expected quoted attribute value");
            const char q = s[i++];
            const size_t v0 = i;
            while (i < s.size() && s[i] != q) i++;
            if (i >= s.size()) return fail("unterminated attribute value");
            el->attrs.emplace_back(std::move(aname), decode(s.substr(v0, i - v0)));
            i++;
        }
        // children until the matching end tag
        for (;;) {
            if (!skip_misc()) return false;
            if (i >= s.size()) return fail("missing end tag for <" + el->tag + ">");
            if (starts("</")) {
                i += 2;
                const size_t e0 = i;
                while (i < s.size() && name_char(s[i])) i++;
                if (s.substr(e0, i - e0) != el->tag) return fail("mismatched end tag for <" + el->tag + ">");
                skip_ws();
                if (i >= s.size() || s[i] != '>') return fail("malformed end tag");
                i++;
                parent.children.push_back(std::move(el));
                return true;
            }
            if (!parse_element(*el)) return false;
        }
    }
};

}  // namespace

std::unique_ptr<XmlElement> xml_parse_string(const std::string& text, std::string& err) {
    Parser p(text);
    auto root = std::make_unique<XmlElement>();
    for (;;) {
        if (!p.skip_misc()) { err = p.err; return nullptr; }
        if (p.i >= text.size()) break;
        if (p.starts("</")) { p.fail("unexpected end tag"); err = p.err; return nullptr; }
        if (!p.parse_element(*root)) { err = p.err; return nullptr; }
    }
    return root;
}

std::unique_ptr<XmlElement> xml_parse_file(const char* path, std::string& err) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot open ") + path; return nullptr; }
    std::string text;
    char buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) text.append(buf, n);
    fclose(fp);
    return xml_parse_string(text, err);
}

}  // namespace crt
