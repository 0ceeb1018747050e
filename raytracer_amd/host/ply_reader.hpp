// PLY triangle-mesh reader (ascii, binary_little_endian, binary_big_endian).
//
// Delivers what the reference takes from a PLY file (scene.cpp:699-733): the vertex
// element's x, y, z widened to double, and the first three indices of every face of
// the list property `vertex_indices` (or `vertex_index`).  Other elements and
// properties are parsed only far enough to be skipped.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtxhost {

struct PlyMesh {
    std::vector<double> xyz;        // 3 per vertex
    std::vector<int64_t> tri;       // 3 per face (first three indices of each face)
};

namespace plydetail {

enum Scalar { I8, U8, I16, U16, I32, U32, F32, F64, BAD };

inline Scalar scalar_of(const std::string& t) {
    if (t == "char" || t == "int8") return I8;
    if (t == "uchar" || t == "uint8") return U8;
    if (t == "short" || t == "int16") return I16;
    if (t == "ushort" || t == "uint16") return U16;
    if (t == "int" || t == "int32") return I32;
    if (t == "uint" || t == "uint32") return U32;
    if (t == "float" || t == "float32") return F32;
    if (t == "double" || t == "float64") return F64;
    return BAD;
}
inline size_t size_of(Scalar s) {
    switch (s) { case I8: case U8: return 1; case I16: case U16: return 2;
                 case I32: case U32: case F32: return 4; case F64: return 8; default: return 0; }
}
struct Prop { std::string name; bool is_list = false; Scalar type = BAD, count_type = BAD; };
struct Elem { std::string name; size_t count = 0; std::vector<Prop> props; };

inline double read_bin(const unsigned char*& p, const unsigned char* end, Scalar t, bool big) {
    size_t n = size_of(t);
    if (p + n > end) throw std::runtime_error("PLY: unexpected end of data");
    unsigned char b[8];
    for (size_t i = 0; i < n; i++) b[i] = big ? p[n - 1 - i] : p[i];
    p += n;
    switch (t) {
        case I8: { int8_t v; memcpy(&v, b, 1); return v; }
        case U8: { uint8_t v; memcpy(&v, b, 1); return v; }
        case I16: { int16_t v; memcpy(&v, b, 2); return v; }
        case U16: { uint16_t v; memcpy(&v, b, 2); return v; }
        case I32: { int32_t v; memcpy(&v, b, 4); return v; }
        case U32: { uint32_t v; memcpy(&v, b, 4); return v; }
        case F32: { float v; memcpy(&v, b, 4); return v; }
        case F64: { double v; memcpy(&v, b, 8); return v; }
        default: return 0;
    }
}

}  // namespace plydetail

inline PlyMesh read_ply(const std::string& path) {
    using namespace plydetail;
    std::ifstream in(path, std::ios::binary);
    if (!in.good()) throw std::runtime_error("Error: PLY file does not exist: " + path);
    std::string data((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    size_t pos = 0;
    auto next_line = [&](std::string& line) -> bool {
        if (pos >= data.size()) return false;
        size_t e = data.find('\n', pos);
        if (e == std::string::npos) e = data.size();
        line = data.substr(pos, e - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = e + 1;
        return true;
    };
    std::string line;
    if (!next_line(line) || line.compare(0, 3, "ply") != 0) throw std::runtime_error("PLY: bad magic in " + path);
    int fmt = -1;  // 0 ascii, 1 little, 2 big
    std::vector<Elem> elems;
    bool header_done = false;
    while (next_line(line)) {
        std::istringstream ls(line);
        std::string tok;
        if (!(ls >> tok)) continue;
        if (tok == "format") {
            std::string f; ls >> f;
            fmt = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
        } else if (tok == "element") {
            Elem e; ls >> e.name >> e.count; elems.push_back(e);
        } else if (tok == "property") {
            if (elems.empty()) throw std::runtime_error("PLY: property before element");
            Prop p; std::string t; ls >> t;
            if (t == "list") {
                std::string ct, it; ls >> ct >> it >> p.name;
                p.is_list = true; p.count_type = scalar_of(ct); p.type = scalar_of(it);
            } else { p.type = scalar_of(t); ls >> p.name; }
            if (p.type == BAD || (p.is_list && p.count_type == BAD)) throw std::runtime_error("PLY: unknown property type");
            elems.back().props.push_back(p);
        } else if (tok == "end_header") { header_done = true; break; }
    }
    if (!header_done || fmt < 0) throw std::runtime_error("PLY: malformed header in " + path);

    PlyMesh mesh;
    const unsigned char* p = (const unsigned char*)data.data() + pos;
    const unsigned char* end = (const unsigned char*)data.data() + data.size();
    const char* ap = data.c_str() + pos;   // ascii cursor
    auto read_ascii = [&]() -> double {
        char* e = nullptr;
        double v = strtod(ap, &e);
        if (e == ap) throw std::runtime_error("PLY: unexpected end of ascii data");
        ap = e;
        return v;
    };
    for (const Elem& e : elems) {
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        int ix = -1, iy = -1, iz = -1, il = -1;
        for (size_t k = 0; k < e.props.size(); k++) {
            const Prop& pr = e.props[k];
            if (is_vertex && !pr.is_list) { if (pr.name == "x") ix = (int)k; else if (pr.name == "y") iy = (int)k; else if (pr.name == "z") iz = (int)k; }
            if (is_face && pr.is_list && (pr.name == "vertex_indices" || pr.name == "vertex_index") && il < 0) il = (int)k;
        }
        if (is_vertex) { if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error("PLY: vertex element lacks x/y/z"); mesh.xyz.resize(e.count * 3); }
        if (is_face) { if (il < 0) throw std::runtime_error("PLY: face element lacks vertex_indices"); mesh.tri.resize(e.count * 3); }
        for (size_t i = 0; i < e.count; i++) {
            for (size_t k = 0; k < e.props.size(); k++) {
                const Prop& pr = e.props[k];
                if (!pr.is_list) {
                    double v = fmt == 0 ? read_ascii() : read_bin(p, end, pr.type, fmt == 2);
                    if (is_vertex) {
                        if ((int)k == ix) mesh.xyz[i * 3 + 0] = v;
                        else if ((int)k == iy) mesh.xyz[i * 3 + 1] = v;
                        else if ((int)k == iz) mesh.xyz[i * 3 + 2] = v;
                    }
                } else {
                    size_t n = (size_t)(fmt == 0 ? read_ascii() : read_bin(p, end, pr.count_type, fmt == 2));
                    for (size_t q = 0; q < n; q++) {
                        double v = fmt == 0 ? read_ascii() : read_bin(p, end, pr.type, fmt == 2);
                        if (is_face && (int)k == il && q < 3) mesh.tri[i * 3 + q] = (int64_t)v;
                    }
                    if (is_face && (int)k == il && n < 3) throw std::runtime_error("PLY: face with fewer than 3 indices");
                }
            }
        }
    }
    return mesh;
}

}  // namespace rtxhost
