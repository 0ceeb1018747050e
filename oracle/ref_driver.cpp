// TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product.
//
// Driver translation unit that is linked against the *unmodified* reference
// sources where they lie under /root/reference (everything except raytracer.cpp,
// which owns main()).  It repeats the reference's own set-up sequence
// (raytracer.cpp:216-246: load -> buildBVH -> precomputeTransformations ->
// precompute normals/determinants) and then, for every camera, drives the
// reference's own castRay / intersect / computePixelColor / clamp exactly as
// __compute does (raytracer.cpp:45-51, quantisation raytracer.cpp:78-80).
//
// Outputs (all little-endian binary, written to <out_prefix>.*):
//   <prefix>.cam<k>.rgb   u8[h][w][3]        the rendered image
//   <prefix>.cam<k>.hits  per pixel 4 x int32 {kind, container, face, src} + f64 distance
//                         (kind uses the reference enum order None,Plane,Sphere,Triangle,Mesh;
//                          src = index of the object that produced a Mesh-kind hit:
//                          mesh index for the meshes loop, n_meshes + instance index for the
//                          instances loop, -1 if it cannot be attributed; for other kinds
//                          src = container)
//   <prefix>.scene        sectioned dump of the finished Scene object (used to check the
//                         product's own loader / BVH builder / transform precompute
//                         bit-for-bit): repeated { char name[24]; u64 nbytes; payload }
//
// Usage: ref_dump <scene.json> <out_prefix> [--cull 0|1] [--bvh 0|1] [--no-render]
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <cfloat>
#include <string>
#include <vector>
#include <limits>
#include <chrono>

#include "scene.h"
#include "utils.h"
#include "precompute.h"
#include "bvh.h"
#include "overloads.h"

using namespace scene;

static void put_section(FILE* f, const std::string& name, const void* data, size_t nbytes) {
    char hdr[24];
    memset(hdr, 0, sizeof hdr);
    strncpy(hdr, name.c_str(), sizeof hdr - 1);
    fwrite(hdr, 1, sizeof hdr, f);
    unsigned long long n = nbytes;
    fwrite(&n, sizeof n, 1, f);
    if (nbytes) fwrite(data, 1, nbytes, f);
}

template <typename T>
static void put_vec(FILE* f, const std::string& name, const std::vector<T>& v) {
    put_section(f, name, v.data(), v.size() * sizeof(T));
}

static int material_index(const Scene& s, const Material* m) {
    if (!m) return -1;
    return (int)(m - s.materials.data());
}

static void push3(std::vector<double>& v, const VectorFloatTriplet& t) {
    v.push_back(t.x); v.push_back(t.y); v.push_back(t.z);
}

static void put_xform(FILE* f, const std::string& name, const Object& o) {
    std::vector<double> v;
    for (int i = 0; i < 16; i++) v.push_back(o.transformMatrix->m[i]);
    for (int i = 0; i < 16; i++) v.push_back(o.inverseTransformMatrix->m[i]);
    for (int i = 0; i < 16; i++) v.push_back(o.normalMatrix->m[i]);
    if (o.worldSpaceBounds) {
        push3(v, o.worldSpaceBounds->min);
        push3(v, o.worldSpaceBounds->max);
    }
    put_vec(f, name, v);
}

static int type_code(const std::string& t) {
    if (t == "mirror") return 1;
    if (t == "conductor") return 2;
    if (t == "dielectric") return 3;
    if (t.empty()) return 0;
    return 4;  // unknown type string: shaded by the reference like a plain material
}

static void dump_scene(const Scene& s, const std::string& path) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { perror("dump"); exit(2); }
    {
        std::vector<double> v;
        push3(v, s.backgroundColor);
        v.push_back(s.shadowRayEpsilon);
        v.push_back(s.intersectionTestEpsilon);
        v.push_back((double)s.maxRecursionDepth);
        push3(v, s.ambientLight.intensity);
        v.push_back(s.enableBackFaceCulling ? 1.0 : 0.0);
        put_vec(f, "scalars", v);
    }
    {
        std::vector<double> v;
        for (const Camera& c : s.cameras) {
            push3(v, c.position); push3(v, c.gaze); push3(v, c.up);
            v.push_back(c.nearPlane.x); v.push_back(c.nearPlane.y);
            v.push_back(c.nearPlane.z); v.push_back(c.nearPlane.w);
            v.push_back(c.nearDistance);
            v.push_back(c.imageResolution.x); v.push_back(c.imageResolution.y);
        }
        put_vec(f, "cameras", v);
    }
    {
        std::vector<double> v;
        for (const PointLight& l : s.pointLights) { push3(v, l.position); push3(v, l.intensity); }
        put_vec(f, "lights", v);
    }
    {
        std::vector<double> v;
        for (const Material& m : s.materials) {
            push3(v, m.ambientReflectance); push3(v, m.diffuseReflectance);
            push3(v, m.specularReflectance);
            // mirrorReflectance is left uninitialised by the reference when the key is absent
            // and is then never read unless isMirror / conductor: dump zeros in that case.
            if (m.isMirror || m.type == "conductor") push3(v, m.mirrorReflectance);
            else { v.push_back(0); v.push_back(0); v.push_back(0); }
            v.push_back(m.phongExponent); v.push_back(m.refractionIndex);
            v.push_back(m.absorptionIndex); push3(v, m.absorptionCoefficient);
            v.push_back(type_code(m.type)); v.push_back(m.isMirror ? 1.0 : 0.0);
        }
        put_vec(f, "materials", v);
    }
    put_section(f, "vertices", s.vertices.data(), s.vertices.size() * sizeof(VectorFloatTriplet));
    for (size_t i = 0; i < s.meshes.size(); i++) {
        const Mesh& m = s.meshes[i];
        std::string p = "mesh" + std::to_string(i);
        std::vector<int> hdr = {(int)m.faces.size(), (int)(m.shadingMode == 's'),
                                material_index(s, m.material), (int)m.hasTransformation,
                                (int)m.hasNegativeScale};
        put_vec(f, p + ".hdr", hdr);
        put_section(f, p + ".faces", m.faces.data(), m.faces.size() * sizeof(VectorIntTriplet));
        if (m.hasTransformation) put_xform(f, p + ".xf", m);
        const MeshBVH* b = i < s.meshBVHs.size() ? s.meshBVHs[i] : nullptr;
        if (b) {
            // BVHNode has 4 bytes of tail padding; dump the fields explicitly.
            std::vector<double> boxes; std::vector<int> ints;
            for (const BVHNode& n : b->nodes) {
                push3(boxes, n.bounds.min); push3(boxes, n.bounds.max);
                ints.push_back(n.leftChild); ints.push_back(n.rightChild);
                ints.push_back(n.firstFace); ints.push_back(n.faceCount);
                ints.push_back(n.splitAxis);
            }
            put_vec(f, p + ".bvhbox", boxes);
            put_vec(f, p + ".bvhint", ints);
            put_vec(f, p + ".faceidx", b->faceIndices);
        }
        if (i < s.meshVertexNormals.size()) {
            // Only the normals of vertices this mesh references are meaningful; dump
            // them per face corner (9 doubles per face) to keep the file small.
            std::vector<double> v;
            const auto& vn = s.meshVertexNormals[i];
            for (const VectorIntTriplet& fc : m.faces) {
                push3(v, vn[fc.x]); push3(v, vn[fc.y]); push3(v, vn[fc.z]);
            }
            put_vec(f, p + ".cornernrm", v);
        }
    }
    {
        std::vector<int> v;
        for (size_t i = 0; i < s.triangles.size(); i++) {
            const Triangle& t = s.triangles[i];
            v.push_back(t.indices.x); v.push_back(t.indices.y); v.push_back(t.indices.z);
            v.push_back(material_index(s, t.material)); v.push_back(t.hasTransformation);
            if (t.hasTransformation) put_xform(f, "tri" + std::to_string(i) + ".xf", t);
        }
        put_vec(f, "triangles", v);
    }
    {
        std::vector<double> v;
        for (size_t i = 0; i < s.spheres.size(); i++) {
            const Sphere& t = s.spheres[i];
            v.push_back(t.center); v.push_back(t.radius);
            v.push_back(material_index(s, t.material)); v.push_back(t.hasTransformation);
            if (t.hasTransformation) put_xform(f, "sph" + std::to_string(i) + ".xf", t);
        }
        put_vec(f, "spheres", v);
    }
    {
        std::vector<double> v;
        for (size_t i = 0; i < s.planes.size(); i++) {
            const Plane& t = s.planes[i];
            v.push_back(t.point); push3(v, t.normal);
            v.push_back(material_index(s, t.material)); v.push_back(t.hasTransformation);
            if (t.hasTransformation) put_xform(f, "pln" + std::to_string(i) + ".xf", t);
        }
        put_vec(f, "planes", v);
    }
    {
        std::vector<int> v;
        for (size_t i = 0; i < s.meshInstances.size(); i++) {
            const MeshInstance& t = s.meshInstances[i];
            v.push_back(t.baseMesh ? t.baseMeshIndex : -1);
            v.push_back(material_index(s, t.material));
            v.push_back(t.hasTransformation); v.push_back(t.hasNegativeScale);
            if (t.baseMesh && t.hasTransformation)
                put_xform(f, "inst" + std::to_string(i) + ".xf", t);
        }
        put_vec(f, "instances", v);
    }
    fclose(f);
}

struct HitRec { int kind, container, face, src; double distance; };

// Attribute a Mesh-kind hit to the object that produced it by re-running the
// reference's own per-object test with a fresh bound and matching the hit point
// bit-for-bit (the Intersection record does not keep the instance index,
// scene.h:248-264).
static int attribute_mesh_hit(const Scene& s, Ray ray, const Intersection& fin) {
    const std::vector<double> nodet;
    for (int pass = 0; pass < 2; pass++) {
        int n = pass == 0 ? (int)s.meshes.size() : (int)s.meshInstances.size();
        for (int i = 0; i < n; i++) {
            double t_min = std::numeric_limits<double>::max();
            Intersection tmp;
            bool hit;
            if (pass == 0) {
                MeshBVH* bvh = (i < (int)s.meshBVHs.size()) ? s.meshBVHs[i] : nullptr;
                hit = rayHitsMesh(ray, s.meshes[i], s.vertices,
                                  s.cameraMeshDeterminant[s.currentCameraIndex][i], t_min, tmp,
                                  s.intersectionTestEpsilon, bvh, s.enableBackFaceCulling, i,
                                  nullptr, nullptr, nullptr, &s);
            } else {
                const MeshInstance& in = s.meshInstances[i];
                if (!(in.baseMesh && in.baseMeshIndex >= 0)) continue;
                int b = in.baseMeshIndex;
                MeshBVH* bvh = (b < (int)s.meshBVHs.size()) ? s.meshBVHs[b] : nullptr;
                hit = rayHitsMesh(ray, *in.baseMesh, s.vertices,
                                  s.cameraMeshDeterminant[s.currentCameraIndex][b], t_min, tmp,
                                  s.intersectionTestEpsilon, bvh, s.enableBackFaceCulling, b,
                                  in.transformMatrix, in.inverseTransformMatrix, in.normalMatrix,
                                  &s, in.material, in.worldSpaceBounds);
            }
            if (!hit) continue;
            if (tmp.faceIndex == fin.faceIndex && tmp.material == fin.material &&
                memcmp(&tmp.point, &fin.point, sizeof tmp.point) == 0)
                return pass == 0 ? i : (int)s.meshes.size() + i;
        }
    }
    return -1;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_dump scene.json out_prefix [--cull 0|1] [--bvh 0|1] [--no-render]\n"); return 2; }
    std::string sceneFile = argv[1], prefix = argv[2];
    bool cull = false, useBVH = true, render = true, attribute = true;
    for (int i = 3; i < argc; i++) {
        if (!strcmp(argv[i], "--cull") && i + 1 < argc) cull = atoi(argv[++i]) != 0;
        else if (!strcmp(argv[i], "--bvh") && i + 1 < argc) useBVH = atoi(argv[++i]) != 0;
        else if (!strcmp(argv[i], "--no-render")) render = false;
        else if (!strcmp(argv[i], "--no-attribute")) attribute = false;
    }
    Scene scene;
    scene.loadSceneFromFile(sceneFile);
    scene.enableBackFaceCulling = cull;                       // raytracer.cpp:220
    if (useBVH) scene.buildBVH();                             // raytracer.cpp:232-234
    scene.precomputeTransformations();                        // raytracer.cpp:236
    std::vector<std::vector<VectorFloatTriplet>> meshVertexNormals;
    std::vector<VectorFloatTriplet> triangleNormals;
    std::vector<std::vector<double>> camTriDet;
    std::vector<std::vector<std::vector<double>>> camMeshDet;
    precomputeMeshNormals(scene.meshes, meshVertexNormals, scene.vertices);
    precomputeTriangleNormals(scene.triangles, triangleNormals, scene.vertices);
    precomputeCameraTriangleDeterminant(scene, camTriDet);
    precomputeCameraMeshDeterminant(scene, camMeshDet);
    scene.cameraTriangleDeterminant = camTriDet;
    scene.cameraMeshDeterminant = camMeshDet;
    scene.meshVertexNormals = meshVertexNormals;
    scene.triangleNormals = triangleNormals;

    dump_scene(scene, prefix + ".scene");
    if (!render) return 0;

    for (size_t k = 0; k < scene.cameras.size(); k++) {
        Camera camera = scene.cameras[k];
        scene.currentCameraIndex = (unsigned char)k;
        int w = camera.imageResolution.x, h = camera.imageResolution.y;
        std::vector<unsigned char> img((size_t)w * h * 3);
        std::vector<HitRec> hits((size_t)w * h);
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                Ray ray = castRay(camera, x, y, w, h);
                Intersection isect = intersect(scene, ray);
                VectorFloatTriplet c = computePixelColor(scene, ray, isect);
                clamp(c, 0, 255);
                size_t p = (size_t)y * w + x;
                img[p * 3 + 0] = (unsigned char)round(c.x);
                img[p * 3 + 1] = (unsigned char)round(c.y);
                img[p * 3 + 2] = (unsigned char)round(c.z);
                HitRec& hr = hits[p];
                hr.kind = isect.hit ? (int)isect.kind : 0;
                hr.container = isect.hit ? isect.containerIndex : -1;
                hr.face = isect.hit ? isect.faceIndex : -1;
                hr.distance = isect.hit ? isect.distance : 0.0;
                hr.src = hr.container;
                if (isect.hit && isect.kind == Intersection::Kind::Mesh)
                    hr.src = attribute ? attribute_mesh_hit(scene, ray, isect) : -1;
                if (isect.hit && isect.kind != Intersection::Kind::Mesh &&
                    isect.kind != Intersection::Kind::Triangle) hr.face = -1;
            }
        }
        auto t1 = std::chrono::high_resolution_clock::now();
        fprintf(stderr, "[ref_dump] camera %zu: %dx%d single-thread %.1f ms\n", k, w, h,
                std::chrono::duration<double, std::milli>(t1 - t0).count());
        std::string base = prefix + ".cam" + std::to_string(k);
        FILE* f = fopen((base + ".rgb").c_str(), "wb");
        fwrite(img.data(), 1, img.size(), f); fclose(f);
        f = fopen((base + ".hits").c_str(), "wb");
        for (const HitRec& hr : hits) {
            fwrite(&hr.kind, 4, 1, f); fwrite(&hr.container, 4, 1, f);
            fwrite(&hr.face, 4, 1, f); fwrite(&hr.src, 4, 1, f);
            fwrite(&hr.distance, 8, 1, f);
        }
        fclose(f);
    }
    return 0;
}
