// rtx_bridge.h -- the reference-side binding of the MI355X core: what a maintainer of the
// reference adds to its build (together with rtx_bridge.cpp and -lrtx_amd) to render on the
// device instead of in multiThreadedRayTracing (raytracer.cpp:93-141).  C++11, like the reference.
#pragma once
#include <vector>

#include "scene.h"      // the reference's own headers
#include "bvh.h"
#include "rtx_amd.h"

namespace rtx_bridge {

// A finished scene::Scene (after raytracer.cpp:246) as the plain description the core takes.
// The arrays of structs live here; vertices, faces, BVH nodes, face order and vertex normals are
// NOT copied -- the description points straight into the Scene, which must outlive it.
struct Description {
    rtx_scene_desc desc;
    std::vector<rtx_material> materials;
    std::vector<rtx_light> lights;
    std::vector<rtx_xform> xforms;
    std::vector<rtx_plane> planes;
    std::vector<rtx_sphere> spheres;
    std::vector<rtx_triangle> triangles;
    std::vector<rtx_mesh> meshes;
    std::vector<rtx_instance> instances;
};

void describe(const scene::Scene& s, Description& out);

// Upload once per scene, after raytracer.cpp:246.  Throws std::runtime_error on failure (the
// reference has no other error channel at this seam).
rtx_scene* upload(const scene::Scene& s, int device);

rtx_camera camera(const scene::Camera& c, int width, int height);

// Replacement for multiThreadedRayTracing / singleThreadedRayTracing (raytracer.cpp:93,143):
// same arguments after the scene handle, same `image` layout (row-major RGB8, row 0 on top),
// returns the elapsed milliseconds.
double deviceRayTracing(rtx_scene* dev, const scene::Camera& c, int width, int height, unsigned char* image);

}  // namespace rtx_bridge
