// yaml_lite.hpp -- the YAML subset libpointmatcher chain files use.
//
// pgslam hands three YAML files to libpointmatcher (reference
// src/pgslam/PoseGraphSlam.hpp:43-51 -> Localizer.hpp:54-78, LoopCloser.hpp:58-74);
// yaml-cpp is not available in this image, so the drop-in layer parses the
// subset those files need: block mappings, block sequences of
// "- ModuleName:" / "- ModuleName" entries, scalar values, '#' comments.
#pragma once
#include <istream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pgslam_amd {
namespace yaml_lite {

struct Module {                       // "Name: {param: value, ...}" or a bare "Name"
    std::string name;
    std::map<std::string, std::string> params;
};
struct Chain {                        // top-level key -> list of modules (one entry for non-list keys)
    std::map<std::string, std::vector<Module>> sections;
    bool has(const std::string &k) const { return sections.count(k) != 0; }
};

inline std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline std::string strip_comment(const std::string &s)
{
    bool in_s = false, in_d = false;
    for (size_t i = 0; i < s.size(); i++) {
        if (s[i] == '\'' && !in_d) in_s = !in_s;
        else if (s[i] == '"' && !in_s) in_d = !in_d;
        else if (s[i] == '#' && !in_s && !in_d) return s.substr(0, i);
    }
    return s;
}

inline std::string unquote(const std::string &s)
{
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

inline Chain parse(std::istream &in)
{
    Chain out;
    std::string line, section;
    Module *cur = nullptr;
    int section_indent = -1, module_indent = -1;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        line = strip_comment(line);
        if (trim(line).empty() || trim(line) == "---") continue;
        if (line.find('\t') != std::string::npos) throw std::runtime_error("yaml_lite: tab in indentation, line " + std::to_string(lineno));
        const int indent = (int)line.find_first_not_of(' ');
        std::string body = trim(line);
        bool item = false;
        if (body.size() >= 2 && body[0] == '-' && body[1] == ' ') { item = true; body = trim(body.substr(2)); }
        else if (body == "-") throw std::runtime_error("yaml_lite: empty sequence item, line " + std::to_string(lineno));
        std::string key = body, val;
        const size_t colon = body.find(':');
        if (colon != std::string::npos) { key = trim(body.substr(0, colon)); val = unquote(trim(body.substr(colon + 1))); }
        key = unquote(key);
        if (indent == 0 && !item) {                       // top-level section
            section = key; section_indent = 0; cur = nullptr; module_indent = -1;
            out.sections[section];
            if (!val.empty()) { out.sections[section].push_back(Module{val, {}}); }   // "inspector: NullInspector"
            continue;
        }
        if (section.empty()) throw std::runtime_error("yaml_lite: entry outside a section, line " + std::to_string(lineno));
        if (item || module_indent < 0 || indent <= module_indent) {  // a module of the current section
            out.sections[section].push_back(Module{key, {}});
            cur = &out.sections[section].back();
            module_indent = indent;
            // inline flow mapping "Name: {a: 1, b: 2}" is not used by libpointmatcher examples; reject loudly
            if (!val.empty()) {
                if (val.front() == '{') throw std::runtime_error("yaml_lite: flow mappings are not supported, line " + std::to_string(lineno));
                // "- Name: value" has no meaning for modules
                throw std::runtime_error("yaml_lite: unexpected scalar after module name, line " + std::to_string(lineno));
            }
            continue;
        }
        if (!cur) throw std::runtime_error("yaml_lite: parameter without module, line " + std::to_string(lineno));
        cur->params[key] = val;
    }
    (void)section_indent;
    return out;
}

inline Chain parse_string(const std::string &s)
{
    std::istringstream iss(s);
    return parse(iss);
}

inline double to_double(const std::string &s, const std::string &what)
{
    if (s == "inf" || s == ".inf" || s == "+inf" || s == "Inf") return 1.0 / 0.0;
    try {
        size_t pos = 0;
        const double v = std::stod(s, &pos);
        if (pos != s.size()) throw std::invalid_argument(s);
        return v;
    } catch (const std::exception &) {
        throw std::runtime_error("yaml_lite: parameter " + what + " is not a number: '" + s + "'");
    }
}

}  // namespace yaml_lite
}  // namespace pgslam_amd
