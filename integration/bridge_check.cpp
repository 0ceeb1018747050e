// bridge_check -- TEST PROGRAM for integration/rtx_bridge.cpp (not shipped): loads a scene file
// with the reference's own loader, BVH builder and precompute (the sequence of
// raytracer.cpp:216-246), describes it through the bridge, loads the same file through this
// repository's host library, and compares the two descriptions field by field.  Where a GPU
// is present (--render) it also uploads through the bridge and renders camera 0.
//   bridge_check <scene.json> [--cull 0|1] [--bvh 0|1] [--render out.rgb]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rtx_bridge.h"
#include "precompute.h"
#include "utils.h"
#include "rtx_host.h"

using namespace scene;

static int g_bad = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (g_bad++ < 20) { fprintf(stderr, "MISMATCH: " __VA_ARGS__); fputc('\n', stderr); } } } while (0)

static bool same_xform(const rtx_scene_desc& a, int xa, const rtx_scene_desc& b, int xb) {
    if ((xa < 0) != (xb < 0)) return false;
    if (xa < 0) return true;
    const rtx_xform &p = a.xforms[xa], &q = b.xforms[xb];
    if (memcmp(p.transform, q.transform, 128) || memcmp(p.inverse, q.inverse, 128) || memcmp(p.normal, q.normal, 128)) return false;
    if (p.has_bounds != q.has_bounds || p.negative_scale != q.negative_scale) return false;
    if (p.has_bounds && (memcmp(p.bounds_min, q.bounds_min, 24) || memcmp(p.bounds_max, q.bounds_max, 24))) return false;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: bridge_check scene.json [--cull 0|1] [--bvh 0|1] [--render out.rgb]\n"); return 2; }
    bool cull = false, use_bvh = true;
    const char* render_to = nullptr;
    for (int i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "--cull") && i + 1 < argc) cull = atoi(argv[++i]) != 0;
        else if (!strcmp(argv[i], "--bvh") && i + 1 < argc) use_bvh = atoi(argv[++i]) != 0;
        else if (!strcmp(argv[i], "--render") && i + 1 < argc) render_to = argv[++i];
    }
    // the reference's own pipeline, raytracer.cpp:216-246
    Scene sc;
    sc.loadSceneFromFile(argv[1]);
    sc.enableBackFaceCulling = cull;
    if (use_bvh) sc.buildBVH();
    sc.precomputeTransformations();
    std::vector<std::vector<VectorFloatTriplet>> meshVertexNormals;
    std::vector<VectorFloatTriplet> triangleNormals;
    precomputeMeshNormals(sc.meshes, meshVertexNormals, sc.vertices);
    precomputeTriangleNormals(sc.triangles, triangleNormals, sc.vertices);
    sc.meshVertexNormals = meshVertexNormals;
    sc.triangleNormals = triangleNormals;

    rtx_bridge::Description D;
    rtx_bridge::describe(sc, D);
    const rtx_scene_desc& a = D.desc;

    rtx_host_scene* hs = nullptr;
    if (rtx_host_load(argv[1], use_bvh ? 1 : 0, cull ? 1 : 0, &hs) != 0) { fprintf(stderr, "host: %s\n", rtx_host_last_error()); return 2; }
    const rtx_scene_desc& b = *rtx_host_desc(hs);

    CHECK(a.abi_version == b.abi_version && a.max_recursion_depth == b.max_recursion_depth &&
          a.enable_back_face_culling == b.enable_back_face_culling, "header scalars");
    CHECK(!memcmp(a.background, b.background, 24) && !memcmp(a.ambient, b.ambient, 24), "background / ambient");
    CHECK(a.shadow_ray_epsilon == b.shadow_ray_epsilon && a.intersection_test_epsilon == b.intersection_test_epsilon, "epsilons");
    CHECK(a.n_vertices == b.n_vertices && !memcmp(a.vertices, b.vertices, (size_t)a.n_vertices * 24), "vertices (%lld vs %lld)", (long long)a.n_vertices, (long long)b.n_vertices);
    CHECK(a.n_materials == b.n_materials, "material count");
    for (int i = 0; i < a.n_materials && i < b.n_materials; i++)
        CHECK(!memcmp(&a.materials[i], &b.materials[i], sizeof(rtx_material)), "material %d", i);
    CHECK(a.n_lights == b.n_lights && (a.n_lights == 0 || !memcmp(a.lights, b.lights, (size_t)a.n_lights * sizeof(rtx_light))), "lights");
    CHECK(a.n_planes == b.n_planes, "plane count");
    for (int i = 0; i < a.n_planes && i < b.n_planes; i++)
        CHECK(!memcmp(a.planes[i].normal, b.planes[i].normal, 24) && a.planes[i].point == b.planes[i].point &&
              a.planes[i].material == b.planes[i].material && same_xform(a, a.planes[i].xform, b, b.planes[i].xform), "plane %d", i);
    CHECK(a.n_spheres == b.n_spheres, "sphere count");
    for (int i = 0; i < a.n_spheres && i < b.n_spheres; i++)
        CHECK(a.spheres[i].radius == b.spheres[i].radius && a.spheres[i].center == b.spheres[i].center &&
              a.spheres[i].material == b.spheres[i].material && same_xform(a, a.spheres[i].xform, b, b.spheres[i].xform), "sphere %d", i);
    CHECK(a.n_triangles == b.n_triangles, "triangle count");
    for (int i = 0; i < a.n_triangles && i < b.n_triangles; i++)
        CHECK(!memcmp(a.triangles[i].v, b.triangles[i].v, 12) && a.triangles[i].material == b.triangles[i].material &&
              same_xform(a, a.triangles[i].xform, b, b.triangles[i].xform), "triangle %d", i);
    CHECK(a.n_meshes == b.n_meshes, "mesh count");
    for (int i = 0; i < a.n_meshes && i < b.n_meshes; i++) {
        const rtx_mesh &p = a.meshes[i], &q = b.meshes[i];
        CHECK(p.n_faces == q.n_faces && p.n_nodes == q.n_nodes && p.material == q.material && p.smooth == q.smooth, "mesh %d header", i);
        if (p.n_faces != q.n_faces || p.n_nodes != q.n_nodes) continue;
        CHECK(p.n_faces == 0 || !memcmp(p.faces, q.faces, (size_t)p.n_faces * 12), "mesh %d faces", i);
        for (int n = 0; n < p.n_nodes; n++) {
            // (field by field: the last four bytes of the 72-byte record are padding)
            const rtx_bvh_node &x = p.nodes[n], &y = q.nodes[n];
            const bool same = !memcmp(x.bounds_min, y.bounds_min, 48) && x.left == y.left && x.right == y.right &&
                              x.first_face == y.first_face && x.face_count == y.face_count && x.split_axis == y.split_axis;
            if (!same) { CHECK(false, "mesh %d BVH node %d", i, n); break; }
        }
        CHECK(p.n_nodes == 0 || !memcmp(p.face_indices, q.face_indices, (size_t)p.n_faces * 4), "mesh %d face order", i);
        CHECK((p.vertex_normals != nullptr) == (q.vertex_normals != nullptr), "mesh %d vertex normals present", i);
        if (p.vertex_normals && q.vertex_normals) {
            // only the normals of vertices the mesh uses are defined (precompute.cpp:9-44)
            for (int f = 0; f < p.n_faces * 3; f++) {
                const int v = p.faces[f];
                if (memcmp(p.vertex_normals + 3 * (size_t)v, q.vertex_normals + 3 * (size_t)v, 24)) { CHECK(false, "mesh %d vertex normal %d", i, v); break; }
            }
        }
        CHECK(same_xform(a, p.xform, b, q.xform), "mesh %d transform", i);
    }
    CHECK(a.n_instances == b.n_instances, "instance count");
    for (int i = 0; i < a.n_instances && i < b.n_instances; i++)
        CHECK(a.instances[i].base_mesh == b.instances[i].base_mesh && a.instances[i].material == b.instances[i].material &&
              same_xform(a, a.instances[i].xform, b, b.instances[i].xform), "instance %d", i);
    // cameras
    CHECK((int)sc.cameras.size() == rtx_host_camera_count(hs), "camera count");
    for (int k = 0; k < (int)sc.cameras.size() && k < rtx_host_camera_count(hs); k++) {
        rtx_camera hc;
        char name[512];
        rtx_host_camera(hs, k, &hc, name, sizeof name);
        const rtx_camera bc = rtx_bridge::camera(sc.cameras[k], sc.cameras[k].imageResolution.x, sc.cameras[k].imageResolution.y);
        CHECK(!memcmp(&hc, &bc, sizeof hc), "camera %d", k);
    }
    printf("bridge_check: %d meshes, %d instances, %d xforms (bridge) / %d (host), %d mismatches\n", a.n_meshes, a.n_instances,
           a.n_xforms, b.n_xforms, g_bad);

    if (render_to && g_bad == 0) {
        rtx_scene* dev = rtx_bridge::upload(sc, 0);
        const Camera& c = sc.cameras[0];
        const int w = c.imageResolution.x, h = c.imageResolution.y;
        std::vector<unsigned char> img((size_t)w * h * 3);
        const double ms = rtx_bridge::deviceRayTracing(dev, c, w, h, img.data());
        rtx_scene_destroy(dev);
        FILE* f = fopen(render_to, "wb");
        fwrite(img.data(), 1, img.size(), f);
        fclose(f);
        printf("bridge_check: rendered %dx%d through the bridge in %.3f ms\n", w, h, ms);
    }
    rtx_host_free(hs);
    return g_bad ? 1 : 0;
}
