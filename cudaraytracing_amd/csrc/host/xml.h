// Minimal XML DOM: elements, attributes, comments, declarations.  Text content
// is skipped because the scene schema keeps every value in attributes.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace crt {

struct XmlElement {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<XmlElement>> children;

    const char* attr(const char* name) const;                 // nullptr when absent
    const XmlElement* child(const char* name) const;          // first child with this tag
    bool float_attr(const char* name, float& out) const;      // false when absent / unparsable
    bool int_attr(const char* name, int& out) const;
};

// Parses a whole document; returns a synthetic root whose children are the
// top-level elements.  On failure returns nullptr and fills err.
std::unique_ptr<XmlElement> xml_parse_file(const char* path, std::string& err);
std::unique_ptr<XmlElement> xml_parse_string(const std::string& text, std::string& err);

}  // namespace crt
