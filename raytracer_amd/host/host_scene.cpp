// Host-side scene pipeline: scene file -> finished rtx_scene_desc.
//
// Produces the same data, bit for bit, that the reference holds when it starts to
// render (raytracer.cpp:216-246): the tests compare a sectioned dump of this pipeline's
// result with a dump of the reference's own Scene object.  Build with
// -ffp-contract=off: the reference binary contains no fused multiply-adds, and matrix
// entries / BVH boxes / vertex normals feed bit-exact hit selection on the device.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "rtx_host.h"
#include "json_min.hpp"
#include "ply_reader.hpp"

namespace {

using rtxhost::Json;

thread_local std::string g_error;

struct V3 { double x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) {                                            // overloads.cpp:18-21
    double len = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return {a.x / len, a.y / len, a.z / len};
}

struct Mat4 { double m[16]; };

Mat4 identity() {
    Mat4 r;
    for (double& v : r.m) v = 0.0;
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0;
    return r;
}

// utils.cpp:1237-1248: each entry starts at 0.0 and accumulates k = 0..3 in order.
Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += a.m[i * 4 + k] * b.m[k * 4 + j];
            r.m[i * 4 + j] = acc;
        }
    return r;
}

Mat4 transpose(const Mat4& a) {
    Mat4 r;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i * 4 + j] = a.m[j * 4 + i];
    return r;
}

// Cofactor inverse (utils.cpp:1260-1294).  Every adjugate entry is a sum of six signed
// triple products accumulated left to right; the table below lists, per output entry,
// the six (sign, i, j, k) terms in the reference's order so that each entry rounds
// identically.  |det| < 1e-9 yields the identity (utils.cpp:1284).
Mat4 invert(const Mat4& mat) {
    struct Term { signed char s; unsigned char a, b, c; };
    static const Term T[16][6] = {
        /* 0*/ {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}},
        /* 1*/ {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}},
        /* 2*/ {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}},
        /* 3*/ {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}},
        /* 4*/ {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}},
        /* 5*/ {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}},
        /* 6*/ {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}},
        /* 7*/ {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}},
        /* 8*/ {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}},
        /* 9*/ {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}},
        /*10*/ {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}},
        /*11*/ {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}},
        /*12*/ {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}},
        /*13*/ {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}},
        /*14*/ {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}},
        /*15*/ {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}},
    };
    const double* m = mat.m;
    Mat4 inv;
    for (int e = 0; e < 16; e++) {
        double acc = 0.0;
        for (int t = 0; t < 6; t++) {
            const Term& q = T[e][t];
            double p = (q.s > 0 ? m[q.a] : -m[q.a]) * m[q.b] * m[q.c];
            acc = t == 0 ? p : acc + p;
        }
        inv.m[e] = acc;
    }
    double det = m[0] * inv.m[0] + m[1] * inv.m[4] + m[2] * inv.m[8] + m[3] * inv.m[12];
    if (std::fabs(det) < 1e-9) return identity();
    det = 1.0 / det;
    for (double& v : inv.m) v *= det;
    return inv;
}

// utils.cpp:96-106: perspective divide when w leaves (0.99999f, 1.00001f).
V3 xf_point(const Mat4& M, V3 p) {
    const double* m = M.m;
    double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w > (double)1.00001f || w < (double)0.99999f) {
        double iw = 1.0 / w;
        return {x * iw, y * iw, z * iw};
    }
    return {x, y, z};
}
V3 xf_dir(const Mat4& M, V3 d) {
    const double* m = M.m;
    return {m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
            m[8] * d.x + m[9] * d.y + m[10] * d.z};
}

// utils.cpp:1353-1361
bool negative_scale(const Mat4& M) {
    const double* a = M.m;
    double det = a[0] * (a[5] * a[10] - a[6] * a[9]) - a[1] * (a[4] * a[10] - a[6] * a[8]) +
                 a[2] * (a[4] * a[9] - a[5] * a[8]);
    return det < 0.0;
}

struct Box {
    double mn[3], mx[3];
    Box() { for (int i = 0; i < 3; i++) { mn[i] = (double)1e30f; mx[i] = (double)-1e30f; } }   // bvh.h:13
    void expand(const double* p) {
        for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); }
    }
    void expand(V3 p) { double q[3] = {p.x, p.y, p.z}; expand(q); }
};

// bvh.cpp:58-78
Box transform_box(const Box& b, const Mat4& M) {
    Box r;
    for (int c = 0; c < 8; c++) {
        V3 p = {(c & 4) ? b.mx[0] : b.mn[0], (c & 2) ? b.mx[1] : b.mn[1], (c & 1) ? b.mx[2] : b.mn[2]};
        r.expand(xf_point(M, p));
    }
    return r;
}

// --- string field parsing: operator>> semantics on whitespace-separated tokens ----------
struct Tokens {
    const char* p;
    explicit Tokens(const std::string& s) : p(s.c_str()) {}
    bool next_double(double& out) {
        char* e = nullptr;
        double v = std::strtod(p, &e);
        if (e == p) return false;
        p = e; out = v; return true;
    }
    bool next_long(long& out) {
        char* e = nullptr;
        long v = std::strtol(p, &e, 10);
        if (e == p) return false;
        p = e; out = v; return true;
    }
};
double parse_double(const std::string& s) { Tokens t(s); double v = 0; t.next_double(v); return v; }
long parse_long(const std::string& s) { Tokens t(s); long v = 0; t.next_long(v); return v; }
V3 parse_v3(const std::string& s) { Tokens t(s); V3 v{0, 0, 0}; t.next_double(v.x); t.next_double(v.y); t.next_double(v.z); return v; }

struct TRef { char type; unsigned id; };
std::vector<TRef> parse_trefs(const std::string& s) {                   // scene.cpp:772-797
    std::vector<TRef> out;
    std::istringstream in(s);
    std::string tok;
    while (in >> tok) {
        if (tok.empty()) continue;
        TRef r;
        r.type = (char)std::tolower((unsigned char)tok[0]);
        r.id = (unsigned)parse_long(tok.substr(1));
        out.push_back(r);
    }
    return out;
}

struct HostCamera { rtx_camera cam; std::string name; std::vector<TRef> tf; };

struct HostObject {
    unsigned id = 0;
    int material = -1;
    std::vector<TRef> tf;
    int xform = -1;
    bool has_tf = false;
    Mat4 M;                 // valid when has_tf
};
struct HostMesh : HostObject {
    bool smooth = false;
    std::vector<int32_t> faces;
    std::vector<rtx_bvh_node> nodes;
    std::vector<int32_t> face_indices;
    std::vector<double> vnormals;
};
struct HostInstance : HostObject {
    unsigned base_id = 0;
    bool reset = false;
    int base_mesh = -1;
    bool resolved = false;
};

}  // namespace

struct rtx_host_scene {
    rtx_scene_desc desc{};
    std::vector<double> vertices;
    std::vector<rtx_material> materials;
    std::map<unsigned, size_t> material_index;
    std::vector<rtx_light> lights;
    std::vector<std::vector<TRef>> light_tf;
    std::vector<HostCamera> cameras;
    std::map<unsigned, V3> translations, scalings;
    struct Rot { double angle; V3 axis; };
    std::map<unsigned, Rot> rotations;
    std::map<unsigned, Mat4> composites;
    std::vector<HostMesh> meshes;
    std::vector<HostInstance> instances;
    std::vector<HostObject> tri_obj, sph_obj, pln_obj;
    std::vector<rtx_triangle> triangles;
    std::vector<rtx_sphere> spheres;
    std::vector<rtx_plane> planes;
    std::vector<rtx_xform> xforms;
    std::vector<rtx_mesh> mesh_desc;
    std::vector<rtx_instance> inst_desc;
    std::string base_dir;
    double parse_ms = 0, preprocess_ms = 0;
    bool last_mesh_smooth = false;   // see parse_mesh
    // What the geometry was built from, kept so that the next frame of an animation can tell
    // whether it shares it: the scene document minus its cameras and point lights, the load
    // flags, and size + modification time of every PLY file read.
    Json geometry_doc;
    bool use_bvh = true, cull = false;
    struct FileStamp { std::string path; long long size; long long mtime_ns; };
    std::vector<FileStamp> files;

    // utils.cpp:1308-1351: result = I; for i = n-1 .. 0: result = result * T_i
    Mat4 object_matrix(const std::vector<TRef>& refs) const {
        Mat4 result = identity();
        for (int i = (int)refs.size() - 1; i >= 0; i--) {
            const TRef& r = refs[(size_t)i];
            Mat4 t = identity();
            if (r.type == 't') {
                auto it = translations.find(r.id);
                if (it == translations.end()) continue;
                t.m[3] = it->second.x; t.m[7] = it->second.y; t.m[11] = it->second.z;
            } else if (r.type == 's') {
                auto it = scalings.find(r.id);
                if (it == scalings.end()) continue;
                t.m[0] = it->second.x; t.m[5] = it->second.y; t.m[10] = it->second.z;
            } else if (r.type == 'r') {                                  // utils.cpp:1207-1227
                auto it = rotations.find(r.id);
                if (it == rotations.end()) continue;
                double angle = it->second.angle * M_PI / 180.0;
                double c = std::cos(angle), s = std::sin(angle);
                V3 u = normalize(it->second.axis);
                t.m[0] = u.x * u.x * (1 - c) + c;
                t.m[1] = u.x * u.y * (1 - c) - u.z * s;
                t.m[2] = u.x * u.z * (1 - c) + u.y * s;
                t.m[4] = u.y * u.x * (1 - c) + u.z * s;
                t.m[5] = u.y * u.y * (1 - c) + c;
                t.m[6] = u.y * u.z * (1 - c) - u.x * s;
                t.m[8] = u.z * u.x * (1 - c) - u.y * s;
                t.m[9] = u.z * u.y * (1 - c) + u.x * s;
                t.m[10] = u.z * u.z * (1 - c) + c;
            } else if (r.type == 'c') {
                auto it = composites.find(r.id);
                if (it == composites.end()) continue;
                t = it->second;
            } else {
                continue;
            }
            result = mul(result, t);
        }
        return result;
    }

    int add_xform(const Mat4& M, const Box* world_bounds) {
        rtx_xform x{};
        Mat4 inv = invert(M);
        Mat4 nrm = transpose(inv);
        memcpy(x.transform, M.m, sizeof M.m);
        memcpy(x.inverse, inv.m, sizeof inv.m);
        memcpy(x.normal, nrm.m, sizeof nrm.m);
        if (world_bounds) {
            memcpy(x.bounds_min, world_bounds->mn, sizeof x.bounds_min);
            memcpy(x.bounds_max, world_bounds->mx, sizeof x.bounds_max);
            x.has_bounds = 1;
        }
        x.negative_scale = negative_scale(M);
        xforms.push_back(x);
        return (int)xforms.size() - 1;
    }
};

namespace {

int material_lookup(const rtx_host_scene& s, unsigned id) {
    auto it = s.material_index.find(id);
    return it == s.material_index.end() ? -1 : (int)it->second;
}

template <typename F>
void for_each_item(const Json& j, F&& fn) {     // "single object or array" duality
    if (j.is_array()) { for (const Json& e : j.items) fn(e); }
    else if (!j.is_null()) fn(j);
}

rtx_material parse_material(const Json& j, unsigned& id) {              // scene.cpp:433-472
    rtx_material m{};
    id = (unsigned)parse_long(j["_id"].str());
    auto put = [](double* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    put(m.ambient, parse_v3(j["AmbientReflectance"].str()));
    put(m.diffuse, parse_v3(j["DiffuseReflectance"].str()));
    put(m.specular, parse_v3(j["SpecularReflectance"].str()));
    m.phong_exponent = j.contains("PhongExponent") ? parse_double(j["PhongExponent"].str()) : 0.0;
    m.refraction_index = 1.0;
    if (j.has("MirrorReflectance")) {
        V3 k = parse_v3(j["MirrorReflectance"].str());
        put(m.mirror, k);
        if (k.x > 0.0 || k.y > 0.0 || k.z > 0.0) m.is_mirror = 1;
    }
    if (j.has("_type")) {
        const std::string& t = j["_type"].str();
        m.type = t == "mirror" ? RTX_MAT_MIRROR : t == "conductor" ? RTX_MAT_CONDUCTOR
               : t == "dielectric" ? RTX_MAT_DIELECTRIC : t.empty() ? RTX_MAT_PLAIN : RTX_MAT_OTHER;
        if (m.type == RTX_MAT_MIRROR) m.is_mirror = 1;
        if (j.contains("RefractionIndex")) m.refraction_index = parse_double(j["RefractionIndex"].str());
        if (j.contains("AbsorptionIndex")) m.absorption_index = parse_double(j["AbsorptionIndex"].str());
        if (j.contains("AbsorptionCoefficient")) put(m.absorption_coefficient, parse_v3(j["AbsorptionCoefficient"].str()));
    }
    return m;
}

HostCamera parse_camera(const Json& j, unsigned& id) {                   // scene.cpp:343-415
    HostCamera hc{};
    rtx_camera& c = hc.cam;
    auto put = [](double* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    id = (unsigned)parse_long(j["_id"].str());
    bool look_at = j.has("_type") && j["_type"].str() == "lookAt";
    V3 pos = parse_v3(j["Position"].str());
    V3 up = parse_v3(j["Up"].str());
    c.near_distance = parse_double(j["NearDistance"].str());
    {
        Tokens t(j["ImageResolution"].str());
        long w = 0, h = 0; t.next_long(w); t.next_long(h);
        c.width = (int32_t)w; c.height = (int32_t)h;
    }
    hc.name = j["ImageName"].str();
    V3 gaze;
    if (look_at) {
        V3 gp = parse_v3(j["GazePoint"].str());
        gaze = normalize(sub(gp, pos));
        double fovY = parse_double(j["FovY"].str()) * M_PI / 180.0;
        double height = 2.0 * std::tan(fovY / 2.0) * c.near_distance;
        double aspect = (double)c.width / (double)c.height;
        double width = height * aspect;
        c.near_plane[0] = -width / 2.0; c.near_plane[1] = width / 2.0;
        c.near_plane[2] = -height / 2.0; c.near_plane[3] = height / 2.0;
    } else {
        gaze = parse_v3(j["Gaze"].str());
        Tokens t(j["NearPlane"].str());
        for (int i = 0; i < 4; i++) t.next_double(c.near_plane[i]);
    }
    if (dot(gaze, up) != 0) {                                            // scene.cpp:398-405
        V3 w = normalize(neg(gaze));
        V3 vp = normalize(up);
        V3 u = cross(vp, w);
        up = cross(w, u);
    }
    put(c.position, pos); put(c.gaze, gaze); put(c.up, up);
    if (j.has("Transformations")) hc.tf = parse_trefs(j["Transformations"].str());
    return hc;
}

void parse_common(const rtx_host_scene& s, const Json& j, HostObject& o) {   // scene.cpp:512-567
    o.id = (unsigned)parse_long(j["_id"].str());
    o.material = j.has("Material") ? material_lookup(s, (unsigned)parse_long(j["Material"].str())) : -1;
    if (j.has("Transformations")) o.tf = parse_trefs(j["Transformations"].str());
}

rtx_host_scene::FileStamp file_stamp(const std::string& path) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return {path, -1, -1};
    return {path, (long long)st.st_size, (long long)st.st_mtim.tv_sec * 1000000000LL + (long long)st.st_mtim.tv_nsec};
}

// Cameras and point lights of a scene document, with their transformations applied
// (scene.cpp:343-415, 971-987).  Replaces what the scene held.
void parse_view(rtx_host_scene& s, const Json& sc) {
    auto put = [](double* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    s.cameras.clear(); s.lights.clear(); s.light_tf.clear();
    if (sc.has("Cameras"))
        for_each_item(sc["Cameras"]["Camera"], [&](const Json& j) {
            unsigned id; HostCamera c = parse_camera(j, id);
            if (id != 0) s.cameras.push_back(c);
        });
    if (sc.has("Lights")) {
        const Json& L = sc["Lights"];
        if (L.has("PointLight"))
            for_each_item(L["PointLight"], [&](const Json& j) {
                unsigned id = (unsigned)parse_long(j["_id"].str());
                rtx_light l{};
                put(l.position, parse_v3(j["Position"].str()));
                put(l.intensity, parse_v3(j["Intensity"].str()));
                if (id != 0) {
                    s.lights.push_back(l);
                    s.light_tf.push_back(j.has("Transformations") ? parse_trefs(j["Transformations"].str()) : std::vector<TRef>());
                }
            });
    }
}
void transform_view(rtx_host_scene& s) {
    auto put = [](double* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
    for (HostCamera& hc : s.cameras)                                     // scene.cpp:971-979
        if (!hc.tf.empty()) {
            Mat4 M = s.object_matrix(hc.tf);
            rtx_camera& c = hc.cam;
            V3 p = xf_point(M, {c.position[0], c.position[1], c.position[2]});
            V3 g = normalize(xf_dir(M, {c.gaze[0], c.gaze[1], c.gaze[2]}));
            V3 u = normalize(xf_dir(M, {c.up[0], c.up[1], c.up[2]}));
            put(c.position, p); put(c.gaze, g); put(c.up, u);
        }
    for (size_t i = 0; i < s.lights.size(); i++)                         // scene.cpp:981-987
        if (!s.light_tf[i].empty()) {
            Mat4 M = s.object_matrix(s.light_tf[i]);
            rtx_light& l = s.lights[i];
            put(l.position, xf_point(M, {l.position[0], l.position[1], l.position[2]}));
        }
}

// The scene document without the parts a frame of an animation may change freely.
Json geometry_part(const Json& sc) {
    Json g = sc;
    for (size_t i = 0; i < g.members.size();) {
        if (g.members[i].first == "Cameras") g.members.erase(g.members.begin() + (long)i);
        else i++;
    }
    for (auto& m : g.members) {
        if (m.first != "Lights" || !m.second.is_object()) continue;
        auto& lm = m.second.members;
        for (size_t i = 0; i < lm.size();) {
            if (lm[i].first == "PointLight") lm.erase(lm.begin() + (long)i);
            else i++;
        }
    }
    return g;
}
bool same_json(const Json& a, const Json& b) {
    if (a.type != b.type || a.text != b.text || a.items.size() != b.items.size() || a.members.size() != b.members.size()) return false;
    for (size_t i = 0; i < a.items.size(); i++) if (!same_json(a.items[i], b.items[i])) return false;
    for (size_t i = 0; i < a.members.size(); i++)
        if (a.members[i].first != b.members[i].first || !same_json(a.members[i].second, b.members[i].second)) return false;
    return true;
}

void parse_faces(rtx_host_scene& s, const Json& j, std::vector<int32_t>& faces) {   // scene.cpp:485-510
    if (j.contains("_data")) {
        Tokens t(j["_data"].str());
        long a, b, c;
        while (t.next_long(a) && t.next_long(b) && t.next_long(c)) {
            faces.push_back((int32_t)(a - 1)); faces.push_back((int32_t)(b - 1)); faces.push_back((int32_t)(c - 1));
        }
    } else if (j.contains("_plyFile")) {
        const std::string ply_path = s.base_dir + j["_plyFile"].str();
        s.files.push_back(file_stamp(ply_path));
        rtxhost::PlyMesh pm = rtxhost::read_ply(ply_path);                            // scene.cpp:699-733
        size_t base = s.vertices.size() / 3;
        faces.resize(pm.tri.size());
        for (size_t i = 0; i < pm.tri.size(); i++) faces[i] = (int32_t)(base + (size_t)pm.tri[i]);
        s.vertices.insert(s.vertices.end(), pm.xyz.begin(), pm.xyz.end());
    }
}

// bvh.cpp:80-102, 187-295.  Median split (std::nth_element) on the largest centroid-extent
// axis, <= 4 faces per leaf, depth <= 64; nodes in DFS pre-order.
struct BvhBuilder {
    const std::vector<double>& verts;
    const std::vector<int32_t>& faces;
    std::vector<rtx_bvh_node>& nodes;
    std::vector<int32_t>& idx;
    std::vector<V3> centroid;

    BvhBuilder(const std::vector<double>& v, const std::vector<int32_t>& f, std::vector<rtx_bvh_node>& n,
               std::vector<int32_t>& i) : verts(v), faces(f), nodes(n), idx(i) {}

    void run() {
        size_t nf = faces.size() / 3;
        nodes.clear(); idx.clear();
        if (nf == 0) return;
        nodes.reserve(nf * 2);
        idx.resize(nf);
        std::iota(idx.begin(), idx.end(), 0);
        centroid.resize(nf);
        const double inv3 = 1.0 / 3.0;
        for (size_t i = 0; i < nf; i++) {
            const double* a = &verts[3 * (size_t)faces[3 * i]];
            const double* b = &verts[3 * (size_t)faces[3 * i + 1]];
            const double* c = &verts[3 * (size_t)faces[3 * i + 2]];
            centroid[i] = {(a[0] + b[0] + c[0]) * inv3, (a[1] + b[1] + c[1]) * inv3, (a[2] + b[2] + c[2]) * inv3};
        }
        build(0, (int)nf, 0);
    }

    int build(int start, int end, int depth) {
        if (start >= end) return -1;
        const int self = (int)nodes.size();
        nodes.emplace_back();
        {
            Box b;
            for (int i = start; i < end; i++) {
                int f = idx[(size_t)i];
                b.expand(&verts[3 * (size_t)faces[3 * (size_t)f]]);
                b.expand(&verts[3 * (size_t)faces[3 * (size_t)f + 1]]);
                b.expand(&verts[3 * (size_t)faces[3 * (size_t)f + 2]]);
            }
            rtx_bvh_node& n = nodes[(size_t)self];
            memcpy(n.bounds_min, b.mn, sizeof b.mn);
            memcpy(n.bounds_max, b.mx, sizeof b.mx);
            n.first_face = start; n.face_count = end - start;
            n.left = -1; n.right = -1; n.split_axis = 0; n.pad = 0;
        }
        const int count = end - start;
        if (count <= 4 || depth >= 64) return self;
        Box cb;
        for (int i = start; i < end; i++) cb.expand(centroid[(size_t)idx[(size_t)i]]);
        double ext[3] = {cb.mx[0] - cb.mn[0], cb.mx[1] - cb.mn[1], cb.mx[2] - cb.mn[2]};
        int axis = 0;
        double best = ext[0];
        if (ext[1] > best) { axis = 1; best = ext[1]; }
        if (ext[2] > best) { axis = 2; best = ext[2]; }
        if (best < (double)1e-5f) return self;
        const int mid = start + count / 2;
        const V3* cen = centroid.data();
        auto key = [cen, axis](int f) { return axis == 0 ? cen[f].x : axis == 1 ? cen[f].y : cen[f].z; };
        std::nth_element(idx.begin() + start, idx.begin() + mid, idx.begin() + end,
                         [&key](int l, int r) { return key(l) < key(r); });
        nodes[(size_t)self].split_axis = axis;
        nodes[(size_t)self].first_face = -1;
        nodes[(size_t)self].face_count = 0;
        int l = build(start, mid, depth + 1);
        int r = build(mid, end, depth + 1);
        nodes[(size_t)self].left = l;
        nodes[(size_t)self].right = r;
        return self;
    }
};

Box mesh_box(const rtx_host_scene& s, const HostMesh& m) {               // bvh.cpp:297-305
    Box b;
    for (int32_t v : m.faces) b.expand(&s.vertices[3 * (size_t)v]);
    return b;
}

// precompute.cpp:9-44: area-weighted vertex normals, accumulated in face order.
void vertex_normals(const rtx_host_scene& s, HostMesh& m) {
    size_t nv = s.vertices.size() / 3;
    std::vector<double>& acc = m.vnormals;
    acc.assign(nv * 3, 0.0);
    std::vector<unsigned char> used(nv, 0);
    for (size_t f = 0; f + 2 < m.faces.size(); f += 3) {
        int i0 = m.faces[f], i1 = m.faces[f + 1], i2 = m.faces[f + 2];
        const double* a = &s.vertices[3 * (size_t)i0];
        const double* b = &s.vertices[3 * (size_t)i1];
        const double* c = &s.vertices[3 * (size_t)i2];
        V3 n = cross({b[0] - a[0], b[1] - a[1], b[2] - a[2]}, {c[0] - a[0], c[1] - a[1], c[2] - a[2]});
        for (int i : {i0, i1, i2}) {
            acc[3 * (size_t)i] = acc[3 * (size_t)i] + n.x;
            acc[3 * (size_t)i + 1] = acc[3 * (size_t)i + 1] + n.y;
            acc[3 * (size_t)i + 2] = acc[3 * (size_t)i + 2] + n.z;
            used[(size_t)i] = 1;
        }
    }
    for (size_t i = 0; i < nv; i++) {
        double x = acc[3 * i], y = acc[3 * i + 1], z = acc[3 * i + 2];
        double len = std::sqrt(x * x + y * y + z * z);
        if (len > 0.0) { double il = 1.0 / len; acc[3 * i] = x * il; acc[3 * i + 1] = y * il; acc[3 * i + 2] = z * il; }
        else { acc[3 * i] = 0.0; acc[3 * i + 1] = 0.0; acc[3 * i + 2] = 1.0; }
    }
}

void load(rtx_host_scene& s, const std::string& path, bool use_bvh, bool cull) {
    auto t0 = std::chrono::high_resolution_clock::now();
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) throw std::runtime_error("Error: The json file cannot be loaded.");
    size_t slash = path.find_last_of("/\\");
    s.base_dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
    std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    Json root = Json::parse(text);
    if (!root.contains("Scene")) throw std::runtime_error("Error: Scene root is not found.");
    const Json& sc = root["Scene"];
    rtx_scene_desc& d = s.desc;
    d.abi_version = RTX_ABI_VERSION;
    d.enable_back_face_culling = cull ? 1 : 0;
    auto put = [](double* dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };

    if (sc.has("BackgroundColor")) put(d.background, parse_v3(sc["BackgroundColor"].str()));
    d.max_recursion_depth = sc.has("MaxRecursionDepth") ? (int32_t)parse_long(sc["MaxRecursionDepth"].str()) : 0;
    d.shadow_ray_epsilon = sc.has("ShadowRayEpsilon") ? parse_double(sc["ShadowRayEpsilon"].str()) : (double)0.001f;
    d.intersection_test_epsilon = sc.has("IntersectionTestEpsilon") ? parse_double(sc["IntersectionTestEpsilon"].str()) : (double)1e-10f;

    s.use_bvh = use_bvh; s.cull = cull;
    parse_view(s, sc);
    if (sc.has("Lights") && sc["Lights"].has("AmbientLight")) put(d.ambient, parse_v3(sc["Lights"]["AmbientLight"].str()));
    if (sc.has("Materials"))
        for_each_item(sc["Materials"]["Material"], [&](const Json& j) {
            unsigned id; rtx_material m = parse_material(j, id);
            if (id != 0) { s.materials.push_back(m); s.material_index[id] = s.materials.size() - 1; }
        });
    if (sc.has("Transformations")) {
        const Json& T = sc["Transformations"];
        if (T.has("Translation")) for_each_item(T["Translation"], [&](const Json& j) {
            s.translations[(unsigned)parse_long(j["_id"].str())] = parse_v3(j["_data"].str()); });
        if (T.has("Scaling")) for_each_item(T["Scaling"], [&](const Json& j) {
            s.scalings[(unsigned)parse_long(j["_id"].str())] = parse_v3(j["_data"].str()); });
        if (T.has("Rotation")) for_each_item(T["Rotation"], [&](const Json& j) {
            Tokens t(j["_data"].str());
            rtx_host_scene::Rot r{0, {0, 0, 0}};
            t.next_double(r.angle); t.next_double(r.axis.x); t.next_double(r.axis.y); t.next_double(r.axis.z);
            s.rotations[(unsigned)parse_long(j["_id"].str())] = r; });
        if (T.has("Composite")) for_each_item(T["Composite"], [&](const Json& j) {
            Tokens t(j["_data"].str());
            Mat4 m{};
            for (double& v : m.m) t.next_double(v);
            s.composites[(unsigned)parse_long(j["_id"].str())] = m; });
    }
    if (sc.has("VertexData"))
        for_each_item(sc["VertexData"]["_data"], [&](const Json& j) {
            Tokens t(j.str());
            double x, y, z;
            while (t.next_double(x) && t.next_double(y) && t.next_double(z)) {
                s.vertices.push_back(x); s.vertices.push_back(y); s.vertices.push_back(z);
            }
        });
    if (sc.has("Objects")) {
        const Json& O = sc["Objects"];
        for_each_item(O["Mesh"], [&](const Json& j) {
            HostMesh m;
            parse_common(s, j, m);
            // The reference assigns Mesh::shadingMode only when the key is present
            // (scene.cpp:572-575) and otherwise leaves it indeterminate; in practice the
            // previous mesh's value shows through.  Reproduce that: inherit, starting flat.
            if (j.has("_shadingMode")) {
                const std::string& sm = j["_shadingMode"].str();
                s.last_mesh_smooth = !sm.empty() && std::tolower((unsigned char)sm[0]) == 's';
            }
            m.smooth = s.last_mesh_smooth;
            if (j.has("Faces")) parse_faces(s, j["Faces"], m.faces);
            s.meshes.push_back(std::move(m));
        });
        for_each_item(O["Triangle"], [&](const Json& j) {
            HostObject o; parse_common(s, j, o);
            rtx_triangle t{};
            if (j.contains("Indices")) {
                Tokens tk(j["Indices"].str());
                long a = 0, b = 0, c = 0; tk.next_long(a); tk.next_long(b); tk.next_long(c);
                t.v[0] = (int32_t)a - 1; t.v[1] = (int32_t)b - 1; t.v[2] = (int32_t)c - 1;
            }
            t.material = o.material; t.xform = -1;
            s.tri_obj.push_back(o); s.triangles.push_back(t);
        });
        for_each_item(O["Sphere"], [&](const Json& j) {
            HostObject o; parse_common(s, j, o);
            rtx_sphere sp{};
            if (j.contains("Center")) sp.center = (int32_t)parse_long(j["Center"].str()) - 1;
            if (j.contains("Radius")) sp.radius = parse_double(j["Radius"].str());
            sp.material = o.material; sp.xform = -1;
            s.sph_obj.push_back(o); s.spheres.push_back(sp);
        });
        for_each_item(O["Plane"], [&](const Json& j) {
            HostObject o; parse_common(s, j, o);
            rtx_plane p{};
            if (j.contains("Point")) p.point = (int32_t)parse_long(j["Point"].str()) - 1;
            if (j.contains("Normal")) put(p.normal, parse_v3(j["Normal"].str()));
            p.material = o.material; p.xform = -1;
            s.pln_obj.push_back(o); s.planes.push_back(p);
        });
        for_each_item(O["MeshInstance"], [&](const Json& j) {
            HostInstance in; parse_common(s, j, in);
            if (j.contains("_baseMeshId")) in.base_id = (unsigned)parse_long(j["_baseMeshId"].str());
            if (j.has("_resetTransform")) {
                const std::string& r = j["_resetTransform"].str();
                in.reset = r == "true" || r == "True" || r == "TRUE" || r == "1";
            }
            s.instances.push_back(in);
        });
    }
    auto t1 = std::chrono::high_resolution_clock::now();

    // ---- preprocessing: raytracer.cpp:229-246 -------------------------------------------
    size_t nverts = s.vertices.size() / 3;
    for (HostMesh& m : s.meshes)
        for (int32_t v : m.faces)
            if (v < 0 || (size_t)v >= nverts) throw std::runtime_error("mesh face references a vertex outside the pool");
    if (use_bvh)
        for (HostMesh& m : s.meshes) { BvhBuilder b(s.vertices, m.faces, m.nodes, m.face_indices); b.run(); }

    auto plain_xform = [&](HostObject& o) {                              // scene.cpp:827-849
        if (o.tf.empty()) return;
        o.M = s.object_matrix(o.tf);
        o.has_tf = true;
    };
    for (HostMesh& m : s.meshes) {
        plain_xform(m);
        if (m.has_tf) {                                                  // scene.cpp:867-879
            Box local;
            if (!m.nodes.empty()) { memcpy(local.mn, m.nodes[0].bounds_min, sizeof local.mn); memcpy(local.mx, m.nodes[0].bounds_max, sizeof local.mx); }
            else local = mesh_box(s, m);
            Box world = transform_box(local, m.M);
            m.xform = s.add_xform(m.M, &world);
        }
    }
    for (size_t i = 0; i < s.triangles.size(); i++) { plain_xform(s.tri_obj[i]); if (s.tri_obj[i].has_tf) s.triangles[i].xform = s.add_xform(s.tri_obj[i].M, nullptr); }
    for (size_t i = 0; i < s.spheres.size(); i++) { plain_xform(s.sph_obj[i]); if (s.sph_obj[i].has_tf) s.spheres[i].xform = s.add_xform(s.sph_obj[i].M, nullptr); }
    for (size_t i = 0; i < s.planes.size(); i++) { plain_xform(s.pln_obj[i]); if (s.pln_obj[i].has_tf) s.planes[i].xform = s.add_xform(s.pln_obj[i].M, nullptr); }

    for (size_t k = 0; k < s.instances.size(); k++) {                    // scene.cpp:881-969
        HostInstance& in = s.instances[k];
        // findMeshOrInstanceById / findBaseMeshIndex (scene.cpp:799-825): meshes first, then
        // instances; an instance whose own base is still unresolved does not resolve.
        int base = -1;
        for (size_t i = 0; i < s.meshes.size(); i++) if (s.meshes[i].id == in.base_id) { base = (int)i; break; }
        HostInstance* parent = nullptr;
        if (base < 0)
            for (HostInstance& other : s.instances)
                if (other.id == in.base_id) { if (other.resolved) base = other.base_mesh; break; }
        in.base_mesh = base;
        in.resolved = base >= 0;
        if (!in.resolved) continue;
        const HostMesh& bm = s.meshes[(size_t)base];
        if (in.material < 0) in.material = bm.material;
        Mat4 fin = identity();
        bool actual = false;
        if (in.reset) {
            if (!in.tf.empty()) { fin = s.object_matrix(in.tf); actual = true; }
        } else {
            for (HostInstance& other : s.instances) if (other.id == in.base_id) { parent = &other; break; }
            bool base_has = false;
            Mat4 base_m = identity();
            if (parent) { base_has = parent->has_tf; if (base_has) base_m = parent->M; }
            else { base_has = bm.has_tf; if (base_has) base_m = bm.M; }
            bool own = !in.tf.empty();
            if (base_has || own) {
                Mat4 im = own ? s.object_matrix(in.tf) : identity();
                fin = mul(im, base_m);
                actual = true;
            }
        }
        if (actual) {
            in.M = fin; in.has_tf = true;
            Box local;
            if (!bm.nodes.empty()) { memcpy(local.mn, bm.nodes[0].bounds_min, sizeof local.mn); memcpy(local.mx, bm.nodes[0].bounds_max, sizeof local.mx); }
            else local = mesh_box(s, bm);
            Box world = transform_box(local, fin);
            in.xform = s.add_xform(fin, &world);
        }
    }
    transform_view(s);
    for (HostMesh& m : s.meshes) if (m.smooth) vertex_normals(s, m);
    auto t2 = std::chrono::high_resolution_clock::now();
    s.parse_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    s.preprocess_ms = std::chrono::duration<double, std::milli>(t2 - t1).count();

    // ---- description -------------------------------------------------------------------
    for (const HostMesh& m : s.meshes) {
        rtx_mesh r{};
        r.faces = m.faces.data(); r.n_faces = (int32_t)(m.faces.size() / 3);
        r.nodes = m.nodes.empty() ? nullptr : m.nodes.data(); r.n_nodes = (int32_t)m.nodes.size();
        r.face_indices = m.face_indices.empty() ? nullptr : m.face_indices.data();
        r.vertex_normals = m.vnormals.empty() ? nullptr : m.vnormals.data();
        r.material = m.material; r.smooth = m.smooth; r.xform = m.xform;
        s.mesh_desc.push_back(r);
    }
    for (const HostInstance& in : s.instances) {
        rtx_instance r{};
        r.base_mesh = in.resolved ? in.base_mesh : -1; r.material = in.material; r.xform = in.xform;
        s.inst_desc.push_back(r);
    }
    d.vertices = s.vertices.data(); d.n_vertices = (int64_t)nverts;
    d.materials = s.materials.data(); d.n_materials = (int32_t)s.materials.size();
    d.lights = s.lights.data(); d.n_lights = (int32_t)s.lights.size();
    d.xforms = s.xforms.data(); d.n_xforms = (int32_t)s.xforms.size();
    d.planes = s.planes.data(); d.n_planes = (int32_t)s.planes.size();
    d.spheres = s.spheres.data(); d.n_spheres = (int32_t)s.spheres.size();
    d.triangles = s.triangles.data(); d.n_triangles = (int32_t)s.triangles.size();
    d.meshes = s.mesh_desc.data(); d.n_meshes = (int32_t)s.mesh_desc.size();
    d.instances = s.inst_desc.data(); d.n_instances = (int32_t)s.inst_desc.size();
    s.geometry_doc = geometry_part(sc);
}

// Next frame of an animation: when `path` describes the geometry `s` already holds (same
// document apart from cameras and point lights, same directory, PLY files untouched), only
// its cameras and lights are taken.  Returns false, leaving `s` alone, when it does not.
bool load_view_only(rtx_host_scene& s, const std::string& path) {
    auto t0 = std::chrono::high_resolution_clock::now();
    std::ifstream file(path, std::ios::binary);
    if (!file.is_open()) throw std::runtime_error("Error: The json file cannot be loaded.");
    size_t slash = path.find_last_of("/\\");
    if ((slash == std::string::npos ? "" : path.substr(0, slash + 1)) != s.base_dir) return false;
    std::string text((std::istreambuf_iterator<char>(file)), std::istreambuf_iterator<char>());
    Json root = Json::parse(text);
    if (!root.contains("Scene")) throw std::runtime_error("Error: Scene root is not found.");
    const Json& sc = root["Scene"];
    if (!same_json(geometry_part(sc), s.geometry_doc)) return false;
    for (const rtx_host_scene::FileStamp& f : s.files) {
        const rtx_host_scene::FileStamp now = file_stamp(f.path);
        if (now.size != f.size || now.mtime_ns != f.mtime_ns) return false;
    }
    parse_view(s, sc);
    transform_view(s);
    s.desc.lights = s.lights.data(); s.desc.n_lights = (int32_t)s.lights.size();
    s.parse_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    s.preprocess_ms = 0.0;
    return true;
}

// ---- dump: same sections as the reference-side dumper used by the tests ----------------
void put_section(FILE* f, const std::string& name, const void* data, size_t nbytes) {
    char hdr[24];
    memset(hdr, 0, sizeof hdr);
    strncpy(hdr, name.c_str(), sizeof hdr - 1);
    fwrite(hdr, 1, sizeof hdr, f);
    unsigned long long n = nbytes;
    fwrite(&n, sizeof n, 1, f);
    if (nbytes) fwrite(data, 1, nbytes, f);
}
template <typename T> void put_vec(FILE* f, const std::string& name, const std::vector<T>& v) {
    put_section(f, name, v.data(), v.size() * sizeof(T));
}
void put_xform(FILE* f, const std::string& name, const rtx_xform& x) {
    std::vector<double> v;
    v.insert(v.end(), x.transform, x.transform + 16);
    v.insert(v.end(), x.inverse, x.inverse + 16);
    v.insert(v.end(), x.normal, x.normal + 16);
    if (x.has_bounds) { v.insert(v.end(), x.bounds_min, x.bounds_min + 3); v.insert(v.end(), x.bounds_max, x.bounds_max + 3); }
    put_vec(f, name, v);
}

void dump(const rtx_host_scene& s, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot open dump file " + path);
    const rtx_scene_desc& d = s.desc;
    put_vec(f, "scalars", std::vector<double>{d.background[0], d.background[1], d.background[2], d.shadow_ray_epsilon,
            d.intersection_test_epsilon, (double)d.max_recursion_depth, d.ambient[0], d.ambient[1], d.ambient[2],
            d.enable_back_face_culling ? 1.0 : 0.0});
    {
        std::vector<double> v;
        for (const HostCamera& hc : s.cameras) {
            const rtx_camera& c = hc.cam;
            v.insert(v.end(), c.position, c.position + 3); v.insert(v.end(), c.gaze, c.gaze + 3);
            v.insert(v.end(), c.up, c.up + 3); v.insert(v.end(), c.near_plane, c.near_plane + 4);
            v.push_back(c.near_distance); v.push_back(c.width); v.push_back(c.height);
        }
        put_vec(f, "cameras", v);
    }
    {
        std::vector<double> v;
        for (const rtx_light& l : s.lights) { v.insert(v.end(), l.position, l.position + 3); v.insert(v.end(), l.intensity, l.intensity + 3); }
        put_vec(f, "lights", v);
    }
    {
        std::vector<double> v;
        for (const rtx_material& m : s.materials) {
            v.insert(v.end(), m.ambient, m.ambient + 3); v.insert(v.end(), m.diffuse, m.diffuse + 3);
            v.insert(v.end(), m.specular, m.specular + 3);
            if (m.is_mirror || m.type == RTX_MAT_CONDUCTOR) v.insert(v.end(), m.mirror, m.mirror + 3);
            else { v.push_back(0); v.push_back(0); v.push_back(0); }
            v.push_back(m.phong_exponent); v.push_back(m.refraction_index); v.push_back(m.absorption_index);
            v.insert(v.end(), m.absorption_coefficient, m.absorption_coefficient + 3);
            v.push_back(m.type); v.push_back(m.is_mirror ? 1.0 : 0.0);
        }
        put_vec(f, "materials", v);
    }
    put_vec(f, "vertices", s.vertices);
    for (size_t i = 0; i < s.meshes.size(); i++) {
        const HostMesh& m = s.meshes[i];
        std::string p = "mesh" + std::to_string(i);
        put_vec(f, p + ".hdr", std::vector<int>{(int)(m.faces.size() / 3), (int)m.smooth, m.material, (int)m.has_tf,
                (int)(m.has_tf ? s.xforms[(size_t)m.xform].negative_scale : 0)});
        put_vec(f, p + ".faces", m.faces);
        if (m.has_tf) put_xform(f, p + ".xf", s.xforms[(size_t)m.xform]);
        if (!m.nodes.empty()) {
            std::vector<double> boxes; std::vector<int> ints;
            for (const rtx_bvh_node& n : m.nodes) {
                boxes.insert(boxes.end(), n.bounds_min, n.bounds_min + 3); boxes.insert(boxes.end(), n.bounds_max, n.bounds_max + 3);
                ints.push_back(n.left); ints.push_back(n.right); ints.push_back(n.first_face); ints.push_back(n.face_count); ints.push_back(n.split_axis);
            }
            put_vec(f, p + ".bvhbox", boxes); put_vec(f, p + ".bvhint", ints); put_vec(f, p + ".faceidx", m.face_indices);
        }
        if (!m.vnormals.empty()) {
            std::vector<double> v;
            for (int32_t vi : m.faces) v.insert(v.end(), &m.vnormals[3 * (size_t)vi], &m.vnormals[3 * (size_t)vi] + 3);
            put_vec(f, p + ".cornernrm", v);
        }
    }
    {
        std::vector<int> v;
        for (size_t i = 0; i < s.triangles.size(); i++) {
            const rtx_triangle& t = s.triangles[i];
            v.push_back(t.v[0]); v.push_back(t.v[1]); v.push_back(t.v[2]); v.push_back(t.material); v.push_back(t.xform >= 0);
            if (t.xform >= 0) put_xform(f, "tri" + std::to_string(i) + ".xf", s.xforms[(size_t)t.xform]);
        }
        put_vec(f, "triangles", v);
    }
    {
        std::vector<double> v;
        for (size_t i = 0; i < s.spheres.size(); i++) {
            const rtx_sphere& t = s.spheres[i];
            v.push_back(t.center); v.push_back(t.radius); v.push_back(t.material); v.push_back(t.xform >= 0);
            if (t.xform >= 0) put_xform(f, "sph" + std::to_string(i) + ".xf", s.xforms[(size_t)t.xform]);
        }
        put_vec(f, "spheres", v);
    }
    {
        std::vector<double> v;
        for (size_t i = 0; i < s.planes.size(); i++) {
            const rtx_plane& t = s.planes[i];
            v.push_back(t.point); v.insert(v.end(), t.normal, t.normal + 3); v.push_back(t.material); v.push_back(t.xform >= 0);
            if (t.xform >= 0) put_xform(f, "pln" + std::to_string(i) + ".xf", s.xforms[(size_t)t.xform]);
        }
        put_vec(f, "planes", v);
    }
    {
        std::vector<int> v;
        for (size_t i = 0; i < s.inst_desc.size(); i++) {
            const rtx_instance& t = s.inst_desc[i];
            v.push_back(t.base_mesh); v.push_back(t.material); v.push_back(t.xform >= 0);
            v.push_back(t.xform >= 0 ? s.xforms[(size_t)t.xform].negative_scale : 0);
            if (t.base_mesh >= 0 && t.xform >= 0) put_xform(f, "inst" + std::to_string(i) + ".xf", s.xforms[(size_t)t.xform]);
        }
        put_vec(f, "instances", v);
    }
    fclose(f);
}

// CRC-32 of PNG chunks, eight bytes per step (slicing tables built on first use).
struct CrcTables {
    uint32_t t[8][256];
    CrcTables() {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = t[0][t[k - 1][i] & 0xFF] ^ (t[k - 1][i] >> 8);
    }
};

uint32_t crc32_update(uint32_t crc, const unsigned char* p, size_t n) {
    static const CrcTables T;
    const uint32_t (*t)[256] = T.t;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);         // (little-endian host)
        lo ^= crc;
        crc = t[7][lo & 0xFF] ^ t[6][(lo >> 8) & 0xFF] ^ t[5][(lo >> 16) & 0xFF] ^ t[4][lo >> 24] ^
              t[3][hi & 0xFF] ^ t[2][(hi >> 8) & 0xFF] ^ t[1][(hi >> 16) & 0xFF] ^ t[0][hi >> 24];
        p += 8; n -= 8;
    }
    for (size_t i = 0; i < n; i++) crc = t[0][(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}

}  // namespace

extern "C" {

const char* rtx_host_last_error(void) { return g_error.c_str(); }

int rtx_host_load(const char* json_path, int use_bvh, int enable_cull, rtx_host_scene** out) {
    if (!json_path || !out) { g_error = "null argument"; return -1; }
    rtx_host_scene* s = nullptr;
    try {
        s = new rtx_host_scene();
        load(*s, json_path, use_bvh != 0, enable_cull != 0);
        *out = s;
        return 0;
    } catch (const std::exception& e) {
        g_error = e.what();
        delete s;
        return -1;
    }
}

int rtx_host_load_next(const char* json_path, rtx_host_scene* prev, rtx_host_scene** out, int* reused) {
    if (!json_path || !out) { g_error = "rtx_host_load_next: null argument"; return -1; }
    if (reused) *reused = 0;
    if (prev) {
        try {
            if (load_view_only(*prev, json_path)) {
                *out = prev;
                if (reused) *reused = 1;
                return 0;
            }
        } catch (const std::exception& e) {
            g_error = e.what();
            return -1;
        }
        const int use_bvh = prev->use_bvh, cull = prev->cull;
        rtx_host_scene* fresh = nullptr;
        if (rtx_host_load(json_path, use_bvh, cull, &fresh) != 0) return -1;   // prev stays valid
        delete prev;
        *out = fresh;
        return 0;
    }
    g_error = "rtx_host_load_next: no previous scene (use rtx_host_load for the first frame)";
    return -1;
}

const rtx_scene_desc* rtx_host_desc(const rtx_host_scene* s) { return s ? &s->desc : nullptr; }
int rtx_host_camera_count(const rtx_host_scene* s) { return s ? (int)s->cameras.size() : 0; }

int rtx_host_camera(const rtx_host_scene* s, int index, rtx_camera* out, char* image_name, int name_cap) {
    if (!s || !out || index < 0 || index >= (int)s->cameras.size()) { g_error = "bad camera index"; return -1; }
    *out = s->cameras[(size_t)index].cam;
    if (image_name && name_cap > 0) {
        strncpy(image_name, s->cameras[(size_t)index].name.c_str(), (size_t)name_cap - 1);
        image_name[name_cap - 1] = 0;
    }
    return 0;
}

int rtx_host_attach_bvh(rtx_host_scene* s, int mesh, const rtx_bvh_node* nodes, int n_nodes, const int32_t* face_indices) {
    if (!s || mesh < 0 || (size_t)mesh >= s->meshes.size() || n_nodes < 0 || (n_nodes > 0 && (!nodes || !face_indices))) {
        g_error = "rtx_host_attach_bvh: bad argument";
        return -1;
    }
    HostMesh& m = s->meshes[(size_t)mesh];
    const size_t nf = m.faces.size() / 3;
    // the tree must be a well-formed one over this mesh's faces: it will be walked on the device
    std::vector<char> seen(nf, 0);
    for (int i = 0; i < n_nodes; i++) {
        const rtx_bvh_node& n = nodes[i];
        const bool leaf = n.left < 0;
        bool ok = leaf ? (n.right < 0 && n.first_face >= 0 && n.face_count >= 0 && (size_t)n.first_face + (size_t)n.face_count <= nf)
                       : (n.left > i && n.left < n_nodes && n.right > i && n.right < n_nodes);
        if (!ok) { g_error = "rtx_host_attach_bvh: malformed node " + std::to_string(i); return -1; }
    }
    for (size_t i = 0; i < (n_nodes > 0 ? nf : 0); i++) {
        const int32_t f = face_indices[i];
        if (f < 0 || (size_t)f >= nf || seen[(size_t)f]) { g_error = "rtx_host_attach_bvh: face_indices is not a permutation of the faces"; return -1; }
        seen[(size_t)f] = 1;
    }
    m.nodes.assign(nodes, nodes + n_nodes);
    m.face_indices.assign(face_indices, face_indices + (n_nodes > 0 ? nf : 0));
    rtx_mesh& r = s->mesh_desc[(size_t)mesh];
    r.nodes = m.nodes.empty() ? nullptr : m.nodes.data();
    r.n_nodes = (int32_t)m.nodes.size();
    r.face_indices = m.face_indices.empty() ? nullptr : m.face_indices.data();
    return 0;
}

int rtx_host_timings(const rtx_host_scene* s, double* parse_ms, double* preprocess_ms) {
    if (!s) return -1;
    if (parse_ms) *parse_ms = s->parse_ms;
    if (preprocess_ms) *preprocess_ms = s->preprocess_ms;
    return 0;
}

int rtx_host_dump(const rtx_host_scene* s, const char* path) {
    if (!s || !path) { g_error = "null argument"; return -1; }
    try { dump(*s, path); return 0; } catch (const std::exception& e) { g_error = e.what(); return -1; }
}

void rtx_host_free(rtx_host_scene* s) { delete s; }

int rtx_host_write_ppm(const char* path, const uint8_t* rgb, int width, int height) {   // scene.cpp:675-697
    if (!path || !rgb) { g_error = "null argument"; return -1; }
    std::string fn = path;
    size_t dot = fn.find_last_of('.');
    if (dot != std::string::npos) fn = fn.substr(0, dot) + ".ppm";
    else fn += ".ppm";
    FILE* f = fopen(fn.c_str(), "w");
    if (!f) { g_error = "Error: The ppm file cannot be opened for writing: " + std::string(path); return -1; }
    fprintf(f, "P3\n%d %d\n255\n", width, height);
    // "<value> " per channel, one text row per image row; the 256 possible pieces are
    // prepared once (a 4K frame is 25 million of them)
    char piece[256][4];
    unsigned char piece_len[256];
    for (int v = 0; v < 256; v++) {
        int n = 0;
        if (v >= 100) piece[v][n++] = (char)('0' + v / 100);
        if (v >= 10) piece[v][n++] = (char)('0' + v / 10 % 10);
        piece[v][n++] = (char)('0' + v % 10);
        piece[v][n++] = ' ';
        for (int k = n; k < 4; k++) piece[v][k] = ' ';
        piece_len[v] = (unsigned char)n;
    }
    std::vector<char> row((size_t)width * 3 * 4 + 4);
    for (int y = 0; y < height; y++) {
        char* o = row.data();
        const uint8_t* src = rgb + (size_t)y * width * 3;
        for (int x = 0; x < width * 3; x++) {
            const unsigned v = src[x];
            memcpy(o, piece[v], 4);
            o += piece_len[v];
        }
        *o++ = '\n';
        fwrite(row.data(), 1, (size_t)(o - row.data()), f);
    }
    fputc('\n', f);
    fclose(f);
    return 0;
}

int rtx_host_write_png(const char* path, const uint8_t* rgb, int width, int height) {
    if (!path || !rgb || width <= 0 || height <= 0) { g_error = "bad argument"; return -1; }
    FILE* f = fopen(path, "wb");
    if (!f) { g_error = std::string("cannot open ") + path; return -1; }
    auto be32 = [](unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char* tag, const std::vector<unsigned char>& data) {
        unsigned char len[4]; be32(len, (uint32_t)data.size());
        fwrite(len, 1, 4, f);
        uint32_t crc = crc32_update(0xFFFFFFFFu, (const unsigned char*)tag, 4);
        fwrite(tag, 1, 4, f);
        if (!data.empty()) { fwrite(data.data(), 1, data.size(), f); crc = crc32_update(crc, data.data(), data.size()); }
        unsigned char c[4]; be32(c, crc ^ 0xFFFFFFFFu);
        fwrite(c, 1, 4, f);
    };
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    std::vector<unsigned char> ihdr(13);
    be32(&ihdr[0], (uint32_t)width); be32(&ihdr[4], (uint32_t)height);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr);
    // raw scanlines (filter byte 0), wrapped in stored deflate blocks inside a zlib stream
    size_t stride = (size_t)width * 3 + 1;
    std::vector<unsigned char> raw(stride * (size_t)height);
    for (int y = 0; y < height; y++) {
        raw[(size_t)y * stride] = 0;
        memcpy(&raw[(size_t)y * stride + 1], rgb + (size_t)y * width * 3, (size_t)width * 3);
    }
    std::vector<unsigned char> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    size_t off = 0;
    while (off < raw.size() || raw.empty()) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        bool last = off + n >= raw.size();
        z.push_back(last ? 1 : 0);
        z.push_back((unsigned char)(n & 0xFF)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xFF)); z.push_back((unsigned char)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + n));
        for (size_t i = 0; i < n;) {                      // Adler-32, reduced once per 5552 bytes (no overflow before)
            const size_t end = std::min(n, i + 5552);
            for (; i < end; i++) { a += raw[off + i]; b += a; }
            a %= 65521u; b %= 65521u;
        }
        off += n;
        if (last) break;
    }
    unsigned char ad[4]; be32(ad, (b << 16) | a);
    z.insert(z.end(), ad, ad + 4);
    chunk("IDAT", z);
    chunk("IEND", {});
    fclose(f);
    return 0;
}

}  // extern "C"
