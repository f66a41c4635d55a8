// Wavefront OBJ -> GroupShape tree for the C++ mirror (rtc.hpp): lib/src/obj_parser.rs, the input format of the
// reference's mesh demo (demos/src/bin/here_be_dragons.rs).  Same semantics as ray_tracer_challenge_amd/obj_parser.py,
// quirks included, because they decide what the renderer is fed:
//   * every line counts as "ignored" (obj_parser.rs:203);
//   * all `v` records must precede the first `f`; at the first face (or at end of input) the vertices are normalised
//     into the [-1, 1] cube around their bounding box centre, in f32 (:250-263);
//   * faces are fan-triangulated (:265-291); a face whose FIRST vertex carries a normal index yields SmoothTriangles,
//     whose normals are looked up with the VERTEX index (:281-283);
//   * `g name` starts a group; faces before any `g` go to the group named "" (:167-176).
#ifndef RTC_OBJ_HPP
#define RTC_OBJ_HPP
#include <cmath>
#include <cstdlib>
#include <istream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "rtc.hpp"

namespace rtc {

struct ParseError : std::runtime_error {  // obj_parser.rs:56-66; `kind` names the enum variant
    std::string kind;
    ParseError(const std::string& k, const std::string& what) : std::runtime_error(what), kind(k) {}
};

struct ObjParseResults {
    size_t num_ignored_lines = 0;
    std::vector<Tuple> vertices{point(0, 0, 0)}, normals{point(0, 0, 0)};  // 1-based, as in the file format (:104-107)
    std::vector<std::pair<std::string, GroupShape>> groups;               // in file order
    bool taken = false;

    GroupShape* get_group(const std::string& name) {
        if (taken) return nullptr;
        for (auto& g : groups)
            if (g.first == name) return &g.second;
        return nullptr;
    }
    GroupShape* get_default_group() { return get_group(""); }
    // obj_parser.rs:33-53 (the reference drains a HashMap, i.e. in unspecified order; here: file order)
    GroupShape take_all_as_group() {
        if (taken) throw ParseError("AlreadyTaken", "the groups were taken before");
        taken = true;
        if (groups.size() == 1) return std::move(groups[0].second);
        GroupShape all;
        for (auto& g : groups) all.add_child(std::move(g.second));
        return all;
    }
};

namespace obj_detail {
inline float parse_f32(const std::string& tok) {  // f32::from_str: correctly rounded, like glibc's strtof
    bool ok = !tok.empty();
    for (char ch : tok) ok = ok && ((ch >= '0' && ch <= '9') || ch == '+' || ch == '-' || ch == '.' || ch == 'e' || ch == 'E');
    char* end = nullptr;
    float v = ok ? std::strtof(tok.c_str(), &end) : 0.0f;
    if (!ok || end != tok.c_str() + tok.size()) throw ParseError("ParseFloatError", "invalid float literal");
    return v;
}
struct FaceVertex {
    long vertex, texture, normal;  // -1: absent
};
inline FaceVertex parse_face(const std::string& s) {  // obj_parser.rs:223-247
    long e[3] = {-1, -1, -1};
    size_t n = 0, start = 0;
    for (;;) {
        size_t slash = s.find('/', start);
        std::string x = s.substr(start, slash == std::string::npos ? std::string::npos : slash - start);
        if (!x.empty()) {
            for (char ch : x)
                if (ch < '0' || ch > '9') throw ParseError("ParseIntError", "invalid digit found in string");
            if (n < 3) e[n] = std::strtol(x.c_str(), nullptr, 10);
        }
        n++;
        if (slash == std::string::npos) break;
        start = slash + 1;
    }
    if (e[0] < 0) throw ParseError("MalformedFace", "Missing vertex index");
    return {e[0], e[1], e[2]};
}
inline void normalize_vertices(std::vector<Tuple>& v) {  // obj_parser.rs:250-263, all in f32
    if (v.size() < 2) return;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t i = 1; i < v.size(); i++) {
        const float* p = v[i].data();
        for (int k = 0; k < 3; k++) mn[k] = std::fmin(mn[k], p[k]), mx[k] = std::fmax(mx[k], p[k]);
    }
    const float sx = mx[0] - mn[0], sy = mx[1] - mn[1], sz = mx[2] - mn[2];
    const float scale = std::fmax(sx, std::fmax(sy, sz)) / 2.0f;
    for (size_t i = 1; i < v.size(); i++) {
        v[i].x = (v[i].x - (mn[0] + sx / 2.0f)) / scale;
        v[i].y = (v[i].y - (mn[1] + sy / 2.0f)) / scale;
        v[i].z = (v[i].z - (mn[2] + sz / 2.0f)) / scale;
    }
}
}  // namespace obj_detail

inline ObjParseResults parse_obj(std::istream& in) {  // obj_parser.rs:100-215
    using namespace obj_detail;
    ObjParseResults r;
    GroupShape* current = nullptr;
    std::string current_name;
    bool have_current = false, normalization_finished = false;
    std::string line;
    size_t index = 0;
    auto at = [](const std::vector<Tuple>& a, long i, const char* what) -> const Tuple& {
        if (i < 0 || (size_t)i >= a.size()) throw ParseError("MalformedFace", std::string(what) + " index out of range");
        return a[(size_t)i];
    };
    for (; std::getline(in, line); index++) {
        std::istringstream ss(line);
        std::vector<std::string> el;
        for (std::string tok; ss >> tok;) el.push_back(tok);
        const std::string head = el.empty() ? "" : el[0];
        if (head == "v" || head == "vn") {
            if (head == "v" && normalization_finished)
                throw ParseError("UnexpectedSymbol", "Found vertex at line " + std::to_string(index) +
                                                         "; vertices must all be specified before any faces are specified");
            std::vector<float> c;
            for (size_t k = 1; k < el.size(); k++) c.push_back(parse_f32(el[k]));
            if (c.size() != 3)
                throw ParseError(head == "v" ? "MalformedVertex" : "MalformedNormal",
                                 "Wrong number of coordinates at line " + std::to_string(index) + "; expected 3, found " +
                                     std::to_string(c.size()));
            if (head == "v") r.vertices.push_back(point(c[0], c[1], c[2]));
            else r.normals.push_back(vector(c[0], c[1], c[2]));
        } else if (head == "f") {
            if (!normalization_finished) {
                normalize_vertices(r.vertices);
                normalization_finished = true;
            }
            std::vector<FaceVertex> fs;
            for (size_t k = 1; k < el.size(); k++) fs.push_back(parse_face(el[k]));
            if (fs.size() < 3)
                throw ParseError("MalformedFace", "Not enough vertices to form a face at line " + std::to_string(index) +
                                                      "; expected 3, found " + std::to_string(fs.size()));
            if (!have_current) {
                r.groups.emplace_back("", GroupShape());
                current_name = "";
                have_current = true;
            }
            for (auto& g : r.groups)
                if (g.first == current_name) current = &g.second;
            const bool smooth = fs[0].normal >= 0;  // fan triangulation, :265-291
            for (size_t i = 1; i + 1 < fs.size(); i++) {
                const Tuple &v1 = at(r.vertices, fs[0].vertex, "vertex"), &v2 = at(r.vertices, fs[i].vertex, "vertex"),
                            &v3 = at(r.vertices, fs[i + 1].vertex, "vertex");
                if (smooth)  // sic: the normals are indexed by the VERTEX index
                    current->add_child(SmoothTriangle(v1, v2, v3, at(r.normals, fs[0].vertex, "normal"),
                                                      at(r.normals, fs[i].vertex, "normal"), at(r.normals, fs[i + 1].vertex, "normal")));
                else
                    current->add_child(Triangle(v1, v2, v3));
            }
        } else if (head == "g") {
            if (el.size() < 2) throw ParseError("MalformedGroupDeclaration", "Missing group name on line " + std::to_string(index));
            bool found = false;
            for (auto& g : r.groups)
                if (g.first == el[1]) g.second = GroupShape(), found = true;  // a repeated name replaces the group (HashMap::insert)
            if (!found) r.groups.emplace_back(el[1], GroupShape());
            current_name = el[1];
            have_current = true;
        }
        r.num_ignored_lines++;
    }
    if (!normalization_finished) normalize_vertices(r.vertices);
    return r;
}

inline ObjParseResults parse_obj(const std::string& text) {
    std::istringstream in(text);
    return parse_obj(in);
}

}  // namespace rtc
#endif
