// Device-side (HBM) layout of a flattened scene.  Host code in rtx_core.hip fills these
// from an rtx_scene_desc; the kernels read nothing else.
//
// Two-level layout:
//   BLAS  -- per mesh, the reference's own binary BVH (bvh.cpp:187-258) re-packed so that
//            one 128-byte record holds BOTH children's boxes of an inner node; leaves are
//            ranges of a triangle array stored in leaf order (the order of
//            MeshBVH::faceIndices), each triangle pre-gathered as (a, b-a, c-a).
//   TLAS  -- a spatial BVH over the world boxes of the mesh instances, built here (the
//            reference scans instances linearly, utils.cpp:756-790).  It only *selects
//            candidates*; candidates are then processed in ascending instance index with
//            the reference's own per-instance logic, because the result of the reference's
//            loop depends on that order (utils.cpp:483-489,510 and 567-570).
#pragma once
#include <stdint.h>

namespace rtxdev {

// Inner BLAS node.  A child with count == 0 is another inner node (ref = node index);
// count > 0 is a leaf owning triangles [ref, ref + count) of the global triangle array.
struct Node {
    double lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t lref, lcount, rref, rcount;
    int32_t axis;          // BVHNode::splitAxis: near child = left iff dir[axis] >= 0
    int32_t pad[3];
};
static_assert(sizeof(Node) == 128, "Node must be 128 bytes");

// Hot triangle record: vertex a and the two edges (b - a), (c - a), plus the original face
// index; padded to 80 bytes so that records start on 16-byte boundaries (five 16-byte pieces).
struct Tri { double a[3], e1[3], e2[3]; int32_t face, pad; };
static_assert(sizeof(Tri) == 80, "Tri must be 80 bytes");

struct Mesh {
    double rmin[3], rmax[3];     // root box (tested before anything else, bvh.cpp:132 on node 0)
    int32_t root_ref, root_count;
    int32_t material, smooth, xform, n_faces;
    int32_t has_vn, pad;
};

// TLAS inner node: child boxes + the index range of the instances below each child.
struct TlasNode {
    double lmin[3], lmax[3], rmin[3], rmax[3];
    int32_t lref, lcount, rref, rcount;      // count > 0: leaf over tlas_order[ref .. ref+count)
    int32_t lmin_idx, lmax_idx, rmin_idx, rmax_idx;
};
static_assert(sizeof(TlasNode) == 128, "TlasNode must be 128 bytes");

struct Inst {
    int32_t base_mesh, material;
    int32_t xform;        // effective transform: the instance's own, else the base mesh's, else -1
    int32_t pad;
};

struct Plane  { double n[3], a[3]; int32_t material, xform; };
struct Sphere { double c[3], r;    int32_t material, xform; };
struct FreeTri { double a[3], e1[3], e2[3], gn[3]; int32_t material, xform; };

struct Material {
    double ambient[3], diffuse[3], specular[3], mirror[3];
    double phong, n, k;
    double sigma[3];
    int32_t type, is_mirror;
};

struct Light { double p[3], i[3]; };

// Row-major 3x4 parts are what the kernels need; the 4th row is kept for the perspective
// divide of transformPointFast (utils.cpp:100-104).
struct Xform {
    double m[16], inv[16], nrm[16];
    double bmin[3], bmax[3];
    int32_t has_bounds, pad;
};

struct Scene {
    double bg[3], ambient[3];
    double shadow_eps, isect_eps;
    int32_t max_depth, cull;
    int32_t n_lights, n_planes, n_spheres, n_ftris, n_meshes, n_insts;
    const Material* materials;
    const Light* lights;
    const Xform* xforms;
    const Plane* planes;
    const Sphere* spheres;
    const FreeTri* ftris;
    const Mesh* meshes;
    const Inst* insts;
    const Node* nodes;
    const Tri* tris;
    const double* tri_gn;     // 3 per triangle: normalize((b-a) x (c-a))
    const double* tri_vn;     // 9 per triangle (vertex normals of its corners) or null
    const TlasNode* tlas;
    const int32_t* tlas_order;
    // TLAS root (may itself be a leaf)
    double tl_min[3], tl_max[3];
    int32_t tl_ref, tl_count, tl_min_idx, tl_max_idx;
};

// Per-frame camera basis, precomputed on the host with the arithmetic of utils.cpp:135-146.
struct Frame {
    double e[3], u[3], v[3], q[3];
    double r_minus_l, t_minus_b;
    int32_t width, height;
    int32_t tile_w, tile_h, tiles_x, tile_first, tile_stride;
    int32_t n_items0;       // generation-0 work items: selected tiles x 8x8 blocks x 64
};

// Ray queue (structure of arrays).  Radiance rays occupy slots 0, 1, ... and shadow rays
// slots capacity-1, capacity-2, ...; meta = depth | (beer_material + 1) << 8.
struct Queue {
    double* ox; double* oy; double* oz;
    double* dx; double* dy; double* dz;
    double* wx; double* wy; double* wz;
    double* aux;            // shadow rays: distance to the light
    uint32_t* pixel;
    uint32_t* meta;
    uint32_t capacity;
};

// Closest-hit records of the radiance rays of one generation (structure of arrays, indexed
// by queue slot; by work item for generation 0).  During traversal the slot IS the ray's
// Intersection record (scene.h:248-264): every event that writes the reference's record stores
// the same fields here.  The normals are stored as the reference's object loop leaves them; the
// tail of intersect() (normalisation, vertex-normal interpolation, utils.cpp:807-853) is
// applied by the shading kernel.  code: -2 no ray (padding item), -1 miss, 1 hit.
struct HitBuf {
    double* dist;
    double* px; double* py; double* pz;
    double* gx; double* gy; double* gz;     // geometric normal
    double* sx; double* sy; double* sz;     // shading normal
    double* beta; double* gamma;
    int32_t* code;                          // completion status, written last
    int32_t* mat; int32_t* kind; int32_t* container; int32_t* face; int32_t* tri; int32_t* src;
};

// Control block of one generation.  counts packs the number of radiance rays (low half) and
// shadow rays (high half) so that one 64-bit atomic reserves slots at either end of the
// queue and observes the other end's fill level at the same instant.
struct GenCtrl {
    unsigned head_trace, head_shade;
    unsigned long long counts;
    unsigned overflow, pad[3];
};

// Everything the kernels of one frame read besides the bulk arrays, kept in device memory so
// that kernels take one pointer (scalar loads) instead of a few hundred bytes of by-value
// arguments held in SGPRs.  q[g & 1] receives the rays spawned by generation g.
struct Launch {
    Scene S;
    Frame F;
    Queue q[2];
    HitBuf hb;
    // Traversal stacks of the trace kernel: RTX_STACK 16-byte entries per lane, each lane's
    // entries contiguous (a push or pop touches one cache line per lane), indexed by
    // (global wave number * 64 + lane).
    void* stacks;
};

struct Counters {
    unsigned long long rays, primary, shadow, secondary;
    unsigned long long node_visits, tri_tests, world_box_tests, object_traversals, tlas_visits;
    unsigned long long inner_steps, inner_lanes, leaf_steps, leaf_lanes;
    // time split of the trace kernel's waves, in shader-clock ticks summed over waves
    unsigned long long service_passes, clk_traverse, clk_service, clk_total, waves, clk_fetch, clk_pop;
    unsigned long long section[8];   // service sections: ticks summed over lanes (begin, finalize, next job, complete), then call counts
};

}  // namespace rtxdev
