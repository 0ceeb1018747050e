// rtx_bridge.cpp -- see rtx_bridge.h.  Added to the reference's build:
//   g++ -std=c++11 -I<reference> -I<repo>/include ... rtx_bridge.cpp -L<repo>/raytracer_amd/lib -lrtx_amd
#include "rtx_bridge.h"

#include <cstring>
#include <stdexcept>

using namespace scene;

namespace rtx_bridge {

// The core takes MeshBVH::nodes.data() as it stands (bvh.h:50-61).
static_assert(sizeof(BVHNode) == sizeof(rtx_bvh_node), "BVHNode and rtx_bvh_node must have the same layout");
static_assert(sizeof(VectorFloatTriplet) == 24 && sizeof(VectorIntTriplet) == 12, "triplets are passed as plain arrays");

static int material_index(const Scene& s, const Material* m) {           // pointers come from getMaterialById,
    return m ? int(m - s.materials.data()) : -1;                          // scene.cpp:629-635
}

static int push_xform(std::vector<rtx_xform>& out, const Object& o) {     // scene.h:119-129
    if (!o.transformMatrix) return -1;
    rtx_xform x;
    memset(&x, 0, sizeof x);
    memcpy(x.transform, o.transformMatrix->m, 128);
    memcpy(x.inverse, o.inverseTransformMatrix->m, 128);
    memcpy(x.normal, o.normalMatrix->m, 128);
    if (o.worldSpaceBounds) {
        x.has_bounds = 1;
        memcpy(x.bounds_min, &o.worldSpaceBounds->min, 24);
        memcpy(x.bounds_max, &o.worldSpaceBounds->max, 24);
    }
    x.negative_scale = o.hasNegativeScale;
    out.push_back(x);
    return int(out.size()) - 1;
}

void describe(const Scene& s, Description& D) {
    D.materials.clear(); D.lights.clear(); D.xforms.clear(); D.planes.clear(); D.spheres.clear();
    D.triangles.clear(); D.meshes.clear(); D.instances.clear();
    for (const Material& m : s.materials) {
        rtx_material r;
        memset(&r, 0, sizeof r);
        memcpy(r.ambient, &m.ambientReflectance, 24);
        memcpy(r.diffuse, &m.diffuseReflectance, 24);
        memcpy(r.specular, &m.specularReflectance, 24);
        // (mirrorReflectance is only assigned when the key is present, scene.cpp:446-460)
        if (m.isMirror || m.type == "conductor") memcpy(r.mirror, &m.mirrorReflectance, 24);
        r.phong_exponent = m.phongExponent;
        r.refraction_index = m.refractionIndex;
        r.absorption_index = m.absorptionIndex;
        memcpy(r.absorption_coefficient, &m.absorptionCoefficient, 24);
        r.type = m.type == "mirror" ? RTX_MAT_MIRROR : m.type == "conductor" ? RTX_MAT_CONDUCTOR
               : m.type == "dielectric" ? RTX_MAT_DIELECTRIC : m.type.empty() ? RTX_MAT_PLAIN : RTX_MAT_OTHER;
        r.is_mirror = m.isMirror;
        D.materials.push_back(r);
    }
    for (const PointLight& l : s.pointLights) {
        rtx_light r;
        memcpy(r.position, &l.position, 24);
        memcpy(r.intensity, &l.intensity, 24);
        D.lights.push_back(r);
    }
    for (const Plane& p : s.planes) {
        rtx_plane r;
        memset(&r, 0, sizeof r);
        memcpy(r.normal, &p.normal, 24);
        r.point = p.point; r.material = material_index(s, p.material); r.xform = push_xform(D.xforms, p);
        D.planes.push_back(r);
    }
    for (const Sphere& p : s.spheres) {
        rtx_sphere r;
        memset(&r, 0, sizeof r);
        r.radius = p.radius; r.center = p.center; r.material = material_index(s, p.material); r.xform = push_xform(D.xforms, p);
        D.spheres.push_back(r);
    }
    for (const Triangle& p : s.triangles) {
        rtx_triangle r;
        memset(&r, 0, sizeof r);
        r.v[0] = p.indices.x; r.v[1] = p.indices.y; r.v[2] = p.indices.z;
        r.material = material_index(s, p.material); r.xform = push_xform(D.xforms, p);
        D.triangles.push_back(r);
    }
    for (size_t i = 0; i < s.meshes.size(); i++) {
        const Mesh& m = s.meshes[i];
        const MeshBVH* b = i < s.meshBVHs.size() ? s.meshBVHs[i] : nullptr;   // absent with -b 0
        rtx_mesh r;
        memset(&r, 0, sizeof r);
        r.faces = reinterpret_cast<const int32_t*>(m.faces.data());
        r.n_faces = int(m.faces.size());
        if (b && !b->nodes.empty()) {
            r.nodes = reinterpret_cast<const rtx_bvh_node*>(b->nodes.data());
            r.n_nodes = int(b->nodes.size());
            r.face_indices = b->faceIndices.data();
        }
        r.smooth = m.shadingMode == 's';
        r.vertex_normals = (r.smooth && i < s.meshVertexNormals.size() && !s.meshVertexNormals[i].empty())
                         ? reinterpret_cast<const double*>(s.meshVertexNormals[i].data()) : nullptr;
        r.material = material_index(s, m.material);
        r.xform = push_xform(D.xforms, m);
        D.meshes.push_back(r);
    }
    for (const MeshInstance& in : s.meshInstances) {
        rtx_instance r;
        memset(&r, 0, sizeof r);
        r.base_mesh = (in.baseMesh && in.baseMeshIndex >= 0) ? in.baseMeshIndex : -1;   // utils.cpp:758
        r.material = material_index(s, in.material);
        r.xform = push_xform(D.xforms, in);
        D.instances.push_back(r);
    }
    rtx_scene_desc& d = D.desc;
    memset(&d, 0, sizeof d);
    d.abi_version = RTX_ABI_VERSION;
    d.max_recursion_depth = s.maxRecursionDepth;
    d.enable_back_face_culling = s.enableBackFaceCulling;                 // the -c flag, raytracer.cpp:220
    memcpy(d.background, &s.backgroundColor, 24);
    memcpy(d.ambient, &s.ambientLight.intensity, 24);
    d.shadow_ray_epsilon = s.shadowRayEpsilon;
    d.intersection_test_epsilon = s.intersectionTestEpsilon;
    d.vertices = reinterpret_cast<const double*>(s.vertices.data());
    d.n_vertices = (int64_t)s.vertices.size();
    d.materials = D.materials.data();   d.n_materials = int(D.materials.size());
    d.lights = D.lights.data();         d.n_lights = int(D.lights.size());
    d.xforms = D.xforms.data();         d.n_xforms = int(D.xforms.size());
    d.planes = D.planes.data();         d.n_planes = int(D.planes.size());
    d.spheres = D.spheres.data();       d.n_spheres = int(D.spheres.size());
    d.triangles = D.triangles.data();   d.n_triangles = int(D.triangles.size());
    d.meshes = D.meshes.data();         d.n_meshes = int(D.meshes.size());
    d.instances = D.instances.data();   d.n_instances = int(D.instances.size());
}

rtx_scene* upload(const Scene& s, int device) {
    Description D;
    describe(s, D);
    rtx_scene* out = nullptr;
    if (rtx_scene_create(&D.desc, device, &out) != RTX_OK) throw std::runtime_error(rtx_last_error());
    return out;                                   // the description may be dropped now
}

rtx_camera camera(const Camera& c, int width, int height) {
    rtx_camera cam;
    memset(&cam, 0, sizeof cam);
    memcpy(cam.position, &c.position, 24);
    memcpy(cam.gaze, &c.gaze, 24);
    memcpy(cam.up, &c.up, 24);
    cam.near_plane[0] = c.nearPlane.x; cam.near_plane[1] = c.nearPlane.y;
    cam.near_plane[2] = c.nearPlane.z; cam.near_plane[3] = c.nearPlane.w;
    cam.near_distance = c.nearDistance;
    cam.width = width; cam.height = height;
    return cam;
}

double deviceRayTracing(rtx_scene* dev, const Camera& c, int width, int height, unsigned char* image) {
    const rtx_camera cam = camera(c, width, height);
    rtx_stats st;
    if (rtx_render(dev, &cam, width, height, 0, 1, image, 0, nullptr, &st) != RTX_OK)
        throw std::runtime_error(rtx_last_error());
    return st.total_ms;
}

}  // namespace rtx_bridge
