// Device-side (HBM) layout of a flattened scene.  Host code in rtx_core.hip fills these
// from an rtx_scene_desc; the kernels read nothing else.
//
// Two-level layout:
//   BLAS  -- per mesh, the reference's own binary BVH (bvh.cpp:187-258) re-packed two levels
//            at a time: one 256-byte record holds the boxes of up to four grandchildren of a
//            binary node; leaves are ranges of a triangle array stored in leaf order (the
//            order of MeshBVH::faceIndices), each triangle pre-gathered as (a, b-a, c-a).
//   TLAS  -- a spatial BVH over the world boxes of the mesh instances, built here in the same
//            four-wide record format (the reference scans instances linearly,
//            utils.cpp:756-790).  It only *selects candidates* (marks them in a per-lane
//            bitmap); candidates are then processed in ascending instance index with the
//            reference's own per-instance logic, because the result of the reference's loop
//            depends on that order (utils.cpp:483-489,510 and 567-570).
#pragma once
#include <stdint.h>

namespace rtxdev {

// words of the per-lane instance bitmap and of its summary level, for n instances
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned rtx_bitmap_words(int n) { return ((unsigned)(n > 0 ? n : 1) + 31u) / 32u; }
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned rtx_summary_words(int n) { return (rtx_bitmap_words(n) + 31u) / 32u; }

// Inner BLAS node: two levels of the reference's binary BVH collapsed into one 256-byte
// record (sixteen 16-byte pieces), so that a traversal step descends two binary levels.
// Slots 0,1 belong to the left child of the binary node the record stands for, slots 2,3 to its
// right child: an inner child contributes its own two children, a leaf child sits in the first
// slot of its pair and leaves the second empty.  cnt: 0 = inner (ref = Node4 index),
// > 0 = leaf owning the cnt triangle records from ref on, -1 = empty slot, -2 (TLAS records
// only) = mesh instance `ref`.  ref is a record key
// (offset in 128-byte units, see Scene::blas).
// axis0 / axisL / axisR are BVHNode::splitAxis of the binary node, its left and its right child:
// the near side of a split is the left one iff dir[axis] >= 0 (bvh.cpp:139-147).
struct alignas(16) Node4 {
    double box[4][6];              // per slot: min xyz, max xyz
    int32_t ref[4];
    int32_t cnt[4];
    int32_t axis0, axisL, axisR, pad0;
    int32_t pad1[4];
};
static_assert(sizeof(Node4) == 256, "Node4 must be 256 bytes");

// Triangle record, 128 bytes = eight 16-byte pieces.  Pieces 0-4 are what a traversal step
// needs (vertex a, the edges b - a and c - a, the original face index); pieces 5-6 hold the
// geometric normal normalize((b-a) x (c-a)), needed by back-face culling and when a hit is
// finalised, so that neither needs a second array.
struct alignas(16) Tri { double a[3], e1[3], e2[3]; int32_t face, pad; double gn[3]; double pad2[3]; };
static_assert(sizeof(Tri) == 128, "Tri must be 128 bytes");
#define RTX_VN_STRIDE 10      // doubles per triangle in tri_vn (9 used; keeps records 16-byte aligned)

// The eight-wide BLAS ("BLAS8") of the default and consistent modes' main kernels.  Three levels of
// the reference's binary BVH in one 256-byte record; the boxes are the reference's own, stored as
// floats rounded OUTWARD, so a box test on them can only let more through than the reference's
// (bvh.cpp:23-56) does.  Slot s = the 3-bit path (left/right at each of the three levels) from the
// node the record stands for; a leaf met above the third level sits in the first slot of its range.
// The inner children of a record are consecutive records and the triangles of its leaf children are
// consecutive in a triangle array of its own (Tri records in this order, Tri::pad = the triangle's
// index in the reference-order array): a child is named by its slot alone, and the stack holds
// ONE entry per record (the children still to visit), not one per child.
// Exactness does not rest on these boxes: the kernel visits a superset of the reference's leaves in
// the reference's own order, every triangle test is the reference's FP64 test, and at a job's end
// the winning triangle is checked against the exact box of its reference leaf (TriAux): the
// reference visits that leaf whenever its box test passes with a bound no smaller than the hit.
struct alignas(16) Node8 {
    float lo[3][8];              // pieces 0-5: min x (slots 0-7), min y, min z
    float hi[3][8];              // pieces 6-11: max x, max y, max z
    uint32_t child_base;         // key of the first inner child's record
    uint32_t tri_base;           // key of the first leaf child's first triangle (fast-order triangle array)
    uint32_t counts;             // four bits per slot: 0 empty, 1..14 triangles of a leaf, 15 inner
    uint32_t axes;               // two bits each: split axis of the node, of its two children (left, right), of its four grandchildren
    uint32_t pad[12];
};
static_assert(sizeof(Node8) == 256, "Node8 must be 256 bytes");
#define RTX_B8_MAX_LEAF 14
// Per triangle of the eight-wide BLAS's triangle array: the exact (FP64) box of the reference BVH leaf that holds it.
struct alignas(16) TriAux { double mn[3], mx[3]; double pad[2]; };
static_assert(sizeof(TriAux) == 64, "TriAux must be 64 bytes");

struct Mesh {
    double rmin[3], rmax[3];     // root box (tested before anything else, bvh.cpp:132 on node 0)
    int32_t root_ref, root_count;
    int32_t material, smooth, xform, n_faces;
    int32_t has_vn, pad;
};

struct Inst {
    int32_t base_mesh, material;
    int32_t xform;        // effective transform: the instance's own, else the base mesh's, else -1
    int32_t pad;
};

// Everything the trace kernel needs about one traversable object -- mesh i is object i,
// instance j is object n_meshes + j -- gathered into one record so that setting up or
// finishing a traversal job is a few wide loads from one place instead of a chain of
// dependent lookups (instance -> mesh -> transform).  First 256 bytes: job set-up
// (utils.cpp:418-494); second 256 bytes: what a found hit needs (utils.cpp:533-668).
enum { OBJ_XF = 1, OBJ_BOUNDS = 2, OBJ_SKIP = 4, OBJ_SMOOTH_VN = 8, OBJ_AFFINE = 16 };   // OBJ_AFFINE: last matrix rows are (0, 0, 0, 1)
// Bits 8-15 of Obj::flags: 0, or 1 + the LDS slot of a *tiny* mesh -- an untransformed mesh among the
// scene's first objects whose BVH is a single leaf of one or two triangles (a ground quad).  The
// trace kernel keeps those triangles in LDS and tests them where it would otherwise set up a
// traversal job: the job would be one step between two service passes.
#define RTX_LDS_TINY 4
struct alignas(16) Obj {
    double inv[16];                              // world -> object (Object::inverseTransformation)
    double bmin[3], bmax[3];                     // world-space AABB (valid with OBJ_BOUNDS)
    double rmin[3], rmax[3];                     // root box of the mesh's BVH (object space)
    int32_t root_ref, root_count, flags, mesh;   // mesh: index of the (base) mesh
    int32_t material, root8_ref, pad0[2];        // root8_ref: the root in the eight-wide BLAS (record key, or its triangles' key for a leaf root)
    double m[16];                                // object -> world
    double nrm[12];                              // rows 0-2 of the normal matrix
    int32_t pad1[8];
};
static_assert(sizeof(Obj) == 512, "Obj must be 512 bytes");

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
    const Obj* objs;          // n_meshes + n_insts records (they live at the end of the BLAS allocation)
    // BLAS records: all Node4 records, then all Tri records, then the Obj records, in one
    // allocation.  A record is named by its offset in 128-byte units ("key"): node i has key 2i,
    // triangle j has key tri_key0 + j, object o has key obj_key0 + 4 o.  Child refs in Node4 and
    // root refs in Obj are keys.
    const char* blas;
    int32_t tri_key0, obj_key0;
    const double* tri_vn;     // RTX_VN_STRIDE per triangle (vertex normals of its corners) or null
    const TriAux* tri_aux;    // per triangle of the eight-wide BLAS (key - t8_key0): exact box of its reference leaf
    int32_t t8_key0, pad_b8;  // key of the eight-wide BLAS's first triangle
    // TLAS: Node4 records among the BLAS records (a slot with cnt == -2 is instance `ref`);
    // box of everything and key of the root record
    double tl_min[3], tl_max[3];
    int32_t tl_key, pad_tl;
    // Where the traversal reports a full stack or frontier (GenCtrl::limit of the frame's first
    // control block; set per frame).  Read only at the place that reports, so it costs no register.
    // The other words of that block (search_back, tally) are reached from it.
    unsigned* limit_flag;
};

// Per-frame camera basis, precomputed on the host with the arithmetic of utils.cpp:135-146.
struct Frame {
    double e[3], u[3], v[3], q[3];
    double r_minus_l, t_minus_b;
    int32_t width, height;
    int32_t tile_w, tile_h, tiles_x, tile_first, tile_stride;
    int32_t n_items0;       // generation-0 work items: selected tiles x 8x8 blocks x 64
    // 8x8 pixel blocks are handed out in a scattered order (block of items i = i * block_mul mod
    // n_blocks, block_mul coprime to n_blocks): neighbouring blocks have rays of similar length,
    // and a wave that takes several of them in a row is left with nothing but long rays at the end.
    uint32_t block_mul, n_blocks;
};

// One queued ray, 96 bytes = six 16-byte pieces.  meta = depth | (beer_material + 1) << 8;
// aux = distance to the light (shadow rays).
struct alignas(16) Ray { double o[3], d[3], w[3], aux; uint32_t pixel, meta; double pad; };
static_assert(sizeof(Ray) == 96, "Ray must be 96 bytes");

// Ray queue of RTX_NSUB sub-queues of sub_capacity slots each (see GenCtrl): in a sub-queue,
// radiance rays occupy slots 0, 1, ... and shadow rays slots sub_capacity-1, sub_capacity-2, ...
struct Queue { Ray* rays; uint32_t sub_capacity, pad; };

// Closest-hit record of a radiance ray, 128 bytes, indexed by queue slot (by work item for
// generation 0).  During traversal the record IS the ray's Intersection (scene.h:248-264):
// every event that writes the reference's record stores the same fields here.  The normals
// are stored as the reference's object loop leaves them; the tail of intersect()
// (normalisation, vertex-normal interpolation, utils.cpp:807-853) is applied by the shading
// kernel.  code: -2 no ray (padding item), -1 miss, 1 hit.
struct alignas(16) Hit {
    double dist, p[3], g[3], s[3], beta, gamma;            // g geometric, s shading normal
    int32_t face, tri, mat, container;
    int32_t kind, src, code, pad;                          // code: completion status, written last
};
static_assert(sizeof(Hit) == 128, "Hit must be 128 bytes");
struct HitBuf { Hit* recs; };

// Control block of one generation.  The queue a generation's rays are appended to is split
// into RTX_NSUB sub-queues of capacity/RTX_NSUB slots, each with its own counter in its own
// 128-byte line: appending waves pick a sub-queue by wave number, so reservations do not all
// serialise on one address.  A counter packs the number of radiance rays (low half) and shadow
// rays (high half) of its sub-queue, so that one 64-bit atomic reserves slots at either end of
// the sub-queue and observes the other end's fill level at the same instant.
#define RTX_NSUB 16
// early: shadow rays of this sub-queue's appending waves that were settled where they would have been
// emitted (k_shade, shadow_ray_is_free) and never queued -- they count as rays all the same.
struct alignas(128) SubCount { unsigned long long counts; unsigned long long early; unsigned long long pad[14]; };
// overflow: a reservation in one of this generation's sub-queues did not fit -- the generation's
// rays are incomplete, its kernels and those of every later generation return at once (the flag is
// handed on from control block to control block) and the host renders the frame again with
// larger queues.  limit: a structural limit of the traversal was hit on the device (a lane's
// stack or a wave's frontier was full and an entry would have been dropped); the host turns it
// into RTX_ELIMIT instead of returning a picture.
// search_back (first control block of a frame): rays that exact mode's search of the instances handed
// back to the index-order replay, by cause (rtx_kernels.h, search_service); `tally`: scratch of
// diagnostic builds.  records: 256-byte records the generation's trace kernel fetched (one per
// traversing lane and step).
struct alignas(128) GenCtrl {
    SubCount sub[RTX_NSUB];
    unsigned head_trace, head_shade, overflow, limit;
    unsigned search_back[5];
    // inexact (first control block): a job of the eight-wide BLAS ended with a triangle whose reference
    // leaf the reference's own box test would not have let it into; the host renders the frame again
    // with the four-wide kernels
    unsigned inexact;
    unsigned pad[2];
    unsigned tally[16];
    unsigned pad2[2];
    unsigned long long records;
};
static_assert(sizeof(GenCtrl) == (RTX_NSUB + 1) * 128, "GenCtrl: the sub-queue counters and one more line");

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
    // Instance bitmaps of the trace kernel, two levels: per lane rtx_summary_words(n) summary words
    // (bit w = bitmap word w holds a candidate) followed by ceil(n/32) bitmap words (one bit per
    // instance); inst_words is their sum.  All clear between rays; indexed like the stacks.
    unsigned* inst_bits;
    uint32_t inst_words;
    // Lanes waiting for service that make a wave of the trace kernel leave its traversal loop.
    uint32_t service_threshold;
    // The records are shallow enough (<= 26 levels) for the wave-wide traversal's visiting-order keys.
    uint32_t wide_ok;
    // In generations of size, drained waves finish their long-running jobs wave-wide as well.
    uint32_t wide_old;
    // Largest number of rays a wave of the trace kernel claims at once.
    uint32_t chunk_max;
    // The shading kernel settles a shadow ray that every object turns down at its first test (no queue, no trace).
    uint32_t early_shadow;
};

struct Counters {
    unsigned long long rays, primary, shadow, secondary;
    unsigned long long node_visits, tri_tests, world_box_tests, object_traversals, tlas_visits;
    unsigned long long inner_steps, inner_lanes, leaf_steps, leaf_lanes;
    // time split of the trace kernel's waves, in shader-clock ticks summed over waves
    unsigned long long service_passes, clk_traverse, clk_service, clk_total, waves, clk_fetch, clk_pop;
    unsigned long long max_wave_ticks, max_wave_steps;   // longest-lived wave of the launch and its step count (atomic max)
    unsigned long long section[8];   // service sections: ticks summed over lanes (begin, finalize, next job, complete), then call counts
    unsigned long long ray_steps_hist[8];   // rays by traversal steps: < 16, < 32, ..., < 1024, >= 1024
    unsigned long long shadow_steps_hist[8]; // the shadow rays among them
    unsigned long long longest_ray;          // (steps << 32) | pixel of the ray with the most traversal steps (atomic max)
};

}  // namespace rtxdev
