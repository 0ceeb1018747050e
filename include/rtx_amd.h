/*
 * rtx_amd.h -- C-ABI of the MI355X ray-casting core.
 *
 * The core replaces, as one unit, what the reference does between
 * `multiThreadedRayTracing(Scene&, Camera&, w, h, unsigned char* image)`
 * (raytracer.cpp:93-141; single-thread twin raytracer.cpp:143-171) and the bytes
 * it leaves in `image`: castRay (utils.cpp:134), intersect (utils.cpp:671),
 * computePixelColor / computeShading (utils.cpp:1046 / 868), clamp + round
 * (utils.cpp:125, raytracer.cpp:78-80).  Everything before that call -- JSON/PLY
 * parsing, BVH build, transform and normal precompute -- stays on the host; its
 * results cross this boundary as the plain-old-data `rtx_scene_desc` below.
 *
 * Every struct is plain C, pointers + counts only.  Field layouts mirror the
 * reference's in-memory records so that a host that already holds a finished
 * `scene::Scene` can point straight into it:
 *   vertices        <- Scene::vertices.data()                 (scene.h:25-27,167)
 *   rtx_bvh_node    <- MeshBVH::nodes.data()  (72-byte BVHNode, bvh.h:50-61)
 *   face_indices    <- MeshBVH::faceIndices.data()            (bvh.h:66)
 *   faces           <- Mesh::faces.data()     (VectorIntTriplet, scene.h:29-31,133)
 *   vertex_normals  <- Scene::meshVertexNormals[i].data()     (scene.h:186)
 *
 * All functions return 0 on success and a negative RTX_E* code on failure;
 * rtx_last_error() gives the message of the calling thread's last failure.
 * There is no CPU fallback: without a usable HIP device every entry point
 * that needs one fails with RTX_ENODEVICE.
 */
#ifndef RTX_AMD_H
#define RTX_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_ABI_VERSION 1

enum {
    RTX_OK = 0,
    RTX_EINVAL = -1,     /* malformed description / argument            */
    RTX_ENODEVICE = -2,  /* no HIP device, or device index out of range */
    RTX_EHIP = -3,       /* a HIP runtime call failed                   */
    RTX_ENOMEM = -4,     /* ray queues would exceed the configured cap  */
    RTX_ELIMIT = -5      /* a structural limit was exceeded: BVH or instance hierarchy deeper than the
                            traversal stack (at creation, or detected on the device while rendering --
                            no picture is returned then), frame too large for one call, fixed scratch
                            of rtx_render_noalloc too small */
};

/* Intersection::Kind, in the reference's enum order (scene.h:256). */
enum { RTX_KIND_NONE = 0, RTX_KIND_PLANE = 1, RTX_KIND_SPHERE = 2, RTX_KIND_TRIANGLE = 3,
       RTX_KIND_MESH = 4 };

/* Material::type string (scene.h:110) as an enum; any other string shades like PLAIN. */
enum { RTX_MAT_PLAIN = 0, RTX_MAT_MIRROR = 1, RTX_MAT_CONDUCTOR = 2, RTX_MAT_DIELECTRIC = 3,
       RTX_MAT_OTHER = 4 };

/* scene::Material (scene.h:102-114). */
typedef struct rtx_material {
    double ambient[3], diffuse[3], specular[3], mirror[3];
    double phong_exponent, refraction_index, absorption_index;
    double absorption_coefficient[3];
    int32_t type;       /* RTX_MAT_*                                         */
    int32_t is_mirror;  /* Material::isMirror (scene.cpp:446-460)            */
} rtx_material;

/* scene::PointLight after Scene::precomputeTransformations (scene.cpp:981-987). */
typedef struct rtx_light { double position[3], intensity[3]; } rtx_light;

/* scene::Camera after precomputeTransformations (scene.h:78-88, scene.cpp:971-979). */
typedef struct rtx_camera {
    double position[3], gaze[3], up[3];
    double near_plane[4];   /* l, r, b, t */
    double near_distance;
    int32_t width, height;
} rtx_camera;

/* The three matrices of a transformed Object (scene.h:123-128), row-major, and its
 * world-space AABB (Object::worldSpaceBounds, scene.cpp:867-879,947-957). */
typedef struct rtx_xform {
    double transform[16], inverse[16], normal[16];
    double bounds_min[3], bounds_max[3];
    int32_t has_bounds;       /* 0 for planes / spheres / free triangles              */
    int32_t negative_scale;   /* Object::hasNegativeScale (informational)             */
} rtx_xform;

typedef struct rtx_plane    { double normal[3]; int32_t point, material, xform, pad; } rtx_plane;
typedef struct rtx_sphere   { double radius;    int32_t center, material, xform, pad; } rtx_sphere;
typedef struct rtx_triangle { int32_t v[3];     int32_t material, xform, pad; } rtx_triangle;

/* scene::BVHNode (bvh.h:50-61): same 72-byte layout, so MeshBVH::nodes.data() can be
 * passed as is.  left == -1 marks a leaf owning face_indices[first_face .. +face_count). */
typedef struct rtx_bvh_node {
    double bounds_min[3], bounds_max[3];
    int32_t left, right, first_face, face_count, split_axis;
    int32_t pad;
} rtx_bvh_node;

/* scene::Mesh (scene.h:131-134) + its MeshBVH (bvh.h:63-66) + its vertex normals. */
typedef struct rtx_mesh {
    const int32_t* faces;           /* n_faces x 3 indices into the global vertex pool  */
    const rtx_bvh_node* nodes;      /* DFS pre-order as built by bvh.cpp:187-258        */
    const int32_t* face_indices;    /* n_faces entries                                   */
    const double* vertex_normals;   /* 3 doubles per *global* vertex id, or NULL         */
    int32_t n_faces, n_nodes;
    int32_t material;               /* index into materials, -1 = none (shades black)    */
    int32_t smooth;                 /* Mesh::shadingMode == 's'                          */
    int32_t xform;                  /* index into xforms, -1 = untransformed             */
    int32_t pad;
} rtx_mesh;

/* scene::MeshInstance after precomputeTransformations (scene.h:150-155, scene.cpp:881-969).
 * base_mesh < 0 marks an instance whose base could not be resolved (skipped, utils.cpp:758). */
typedef struct rtx_instance { int32_t base_mesh, material, xform, pad; } rtx_instance;

/* rtx_scene_desc::build_flags (no counterpart in the reference).
 * RTX_SCENE_WIDE_BLAS: also build and upload the eight-wide float-box records that
 * rtx_scene_set_wide_blas switches the mesh traversal to (the record memory about doubles -- the
 * triangles are held a second time in that tree's leaf order -- plus 64 bytes per triangle). */
enum { RTX_SCENE_WIDE_BLAS = 1 };

typedef struct rtx_scene_desc {
    int32_t abi_version;                 /* RTX_ABI_VERSION */
    int32_t max_recursion_depth;
    int32_t enable_back_face_culling;    /* the -c flag (raytracer.cpp:199-207)          */
    int32_t build_flags;                 /* RTX_SCENE_* below; 0 = the default device scene */
    double background[3], ambient[3];
    double shadow_ray_epsilon, intersection_test_epsilon;
    const double* vertices;        int64_t n_vertices;   /* 3 doubles each */
    const rtx_material* materials; int32_t n_materials;
    const rtx_light* lights;       int32_t n_lights;
    const rtx_xform* xforms;       int32_t n_xforms;
    const rtx_plane* planes;       int32_t n_planes;
    const rtx_sphere* spheres;     int32_t n_spheres;
    const rtx_triangle* triangles; int32_t n_triangles;
    const rtx_mesh* meshes;        int32_t n_meshes;
    const rtx_instance* instances; int32_t n_instances;
} rtx_scene_desc;

/* Per-pixel record of the primary ray's closest hit, for parity checks.
 * src: for MESH hits the object that produced the hit -- mesh index (meshes loop,
 * utils.cpp:736-739) or n_meshes + instance index (instances loop, utils.cpp:756-790);
 * for other kinds the container index.  distance is Intersection::distance. */
typedef struct rtx_hit_id { int32_t kind, container, face, src; double distance; } rtx_hit_id;

/* Work counters of one rtx_render call.  A "ray" is one closest-hit query the reference
 * semantics require (one intersect() call, utils.cpp:671), excluding the calls whose result
 * computePixelColor discards because depth > maxRecursionDepth (utils.cpp:1047).  The four ray
 * counts, the times and generations / queue_overflows are always filled; the traversal counts
 * (node_visits ... wave_leaf_lanes) only with rtx_scene_set_counters on. */
typedef struct rtx_stats {
    uint64_t rays, primary_rays, shadow_rays, secondary_rays;
    uint64_t node_visits;      /* BLAS record visits (one record = two binary BVH levels) */
    uint64_t tri_tests;        /* rayHitsTriangle evaluations                             */
    uint64_t world_box_tests;  /* per-object world-AABB tests (utils.cpp:451,761)         */
    uint64_t object_space_traversals;
    uint64_t tlas_node_visits;
    double kernel_ms;          /* HIP-event time of all kernels of the frame              */
    double total_ms;           /* kernels + download of the image                         */
    uint32_t generations, queue_overflows;
    /* SIMT occupancy of the traversal (counters on): wave-level inner-node / triangle steps
     * and the number of lanes that did useful work in them (64 per step is the ideal). */
    uint64_t wave_inner_steps, wave_inner_lanes, wave_leaf_steps, wave_leaf_lanes;
    /* The dominant kernel (closest-hit traversal, one launch per ray generation): summed HIP-event
     * duration of its launches and their number (a frame redone after a queue overflow counts
     * both attempts). */
    double trace_ms;
    uint32_t trace_launches;
    /* How the frame took the scene's mesh instances: 0 = candidates from the TLAS replayed in index
     * order (the reference's loop, utils.cpp:756-790, without the instances its world-box test turns
     * down); 1 = the default mode's nearest-first search with verification (scenes of 32 instances
     * or more; same records and pictures, fewer steps); 2 = consistent mode's nearest-first walk. */
    uint32_t instance_path;
    /* 256-byte records the traversal fetched (collapsed nodes, triangle pairs, instance set-up
     * records; one per traversing lane and step of the trace kernel's main loop): always filled,
     * in every mode -- the device's own count of what it moved, whatever path the rays took. */
    uint64_t record_fetches;
    /* rtx_scene_set_wide_blas: 1 when the frame's mesh traversal ran over the eight-wide records;
     * wide_blas_redone: 1 when its end-of-job check failed somewhere and the frame was rendered again
     * with the four-wide kernels (the picture returned is that second one). */
    uint32_t wide_blas, wide_blas_redone;
    /* Of shadow_rays: those that no object of the scene could hold up -- every mesh and instance turns
     * them down at its first box test -- and that the shading kernel therefore settled on the spot
     * (isInShadow, utils.cpp:1066-1084, answers "no") instead of queueing them for the trace kernel.
     * Default and consistent mode, scenes without planes, spheres and free triangles; 0 otherwise
     * (and with work counters on, when every query is carried out). */
    uint64_t shadow_rays_settled_early;
} rtx_stats;

typedef struct rtx_scene rtx_scene;   /* opaque: owns the device copy of the scene */

/* Flatten `desc` into the device layout and upload it to HIP device `device`
 * (replaces nothing in the reference; it is the "device SoA upload" the seam needs).
 * The description may be freed after the call returns. */
int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out);

/* Replace the point lights (animated lights; same count not required). */
int rtx_scene_update_lights(rtx_scene* scene, const rtx_light* lights, int32_t n_lights);

/* Render one camera: the replacement for multiThreadedRayTracing (raytracer.cpp:93-141).
 *  tile_w/tile_h/tile_first/tile_stride select an interleaved subset of tile_w x tile_h
 *  pixel tiles (row-major tile numbering): tiles tile_first, tile_first+tile_stride, ...
 *  The whole frame in the reference's image layout (row-major, RGB interleaved, row 0 on top,
 *  raytracer.cpp:78-80; width*height*3 bytes) is ONE tile: tile_w = width, tile_h = height,
 *  tile_first = 0, tile_stride = 1.
 *  Any other tiling -- also tile_first = 0, tile_stride = 1 with smaller tiles, which selects
 *  every tile -- writes the selected tiles PACKED one after another, each tile_h rows of
 *  tile_w pixels; a tile clipped by the image edge still occupies a full tile_w*tile_h*3
 *  slot, so the buffer can be larger than width*height*3: size it with
 *  rtx_render_output_bytes and bring it into image order with rtx_unpack_tiles.
 *  rgb_is_device != 0: rgb_out is a device pointer on the scene's device.
 *  hit_ids (host pointer or NULL): w*h records, one-tile (whole-frame) renders only.
 *  stats may be NULL.  Blocking. */
int rtx_render(rtx_scene* scene, const rtx_camera* camera,
               int32_t tile_w, int32_t tile_h, int32_t tile_first, int32_t tile_stride,
               uint8_t* rgb_out, int32_t rgb_is_device,
               rtx_hit_id* hit_ids, rtx_stats* stats);

/* rtx_render for callers that cannot allocate device memory: tight serving loops, harnesses that
 * forbid hipMalloc.  It has no counterpart in the reference.  Same arguments and results as
 * rtx_render (no hit_ids), but every byte of device memory the frame needs beyond the scene and
 * rgb_out -- ray queues, hit records, traversal stacks, instance bitmaps, accumulators, control
 * blocks -- comes from one fixed array of rtx_noalloc_scratch_bytes() bytes (64 MiB) that the
 * library holds on every GPU it runs on until the process ends; the call makes no hipMalloc /
 * hipFree and its trace grid is 32 blocks, so it is meant for tiles, not for large frames.  The call
 * is blocking (it records events, copies its control blocks back and waits for its stream), so it
 * cannot itself be captured into a HIP graph.  What fits: roughly pixels * (lights + 2) <= 150 000
 * queued rays (e.g. a 192 x 192 tile with two lights); a larger request fails with RTX_ELIMIT and a
 * message that states the bytes needed, a queue overflow at run time (deep dielectric branching)
 * fails with RTX_ENOMEM instead of growing the queues.  Pass page-locked or device memory as
 * rgb_out (the runtime stages copies into pageable memory itself).  Every call on a GPU shares that
 * GPU's array: calls that overlap on one GPU, from any scene or thread, race. */
int rtx_render_noalloc(rtx_scene* scene, const rtx_camera* camera,
                       int32_t tile_w, int32_t tile_h, int32_t tile_first, int32_t tile_stride,
                       uint8_t* rgb_out, int32_t rgb_is_device, rtx_stats* stats);
int64_t rtx_noalloc_scratch_bytes(void);

/* Several cameras of ONE uploaded scene in flight at once.  The reference renders an animation as one
 * process per frame file (runner.py:105-141) and the cameras of a file one after the other
 * (raytracer.cpp:252-264); here a frame context owns everything a render needs besides the scene
 * -- a stream, ray queues, hit records, traversal stacks, pixel sums, control blocks -- while
 * geometry, instance hierarchy and object records are shared read-only, and the lights are
 * versioned (a render keeps the lights that were current when it was submitted).  The last ray
 * generations of a frame are a few long rays in an otherwise idle GPU; a neighbouring frame's
 * kernels fill that time.
 *   rtx_frame_create   a new frame context of `scene` (about pixels x (lights + 2) x 320 bytes of
 *                      device memory once used, plus 210 MB of traversal stacks)
 *   rtx_render_async   rtx_render's arguments (same meaning); returns once the work is queued.
 *                      rgb_out / hit_ids must stay valid until the wait; page-locked or device
 *                      memory keeps the final copy asynchronous.  One render per frame at a time.
 *   rtx_frame_wait     waits for the render, redoes it with larger queues if they overflowed, or
 *                      with the four-wide records if the eight-wide check failed (as rtx_render
 *                      does), fills `stats` (may be NULL).  The bytes are the same as rtx_render's
 *                      for the same camera.
 *   rtx_frame_destroy  waits for the frame's stream and frees the context.  Destroy a scene's
 *                      frames before the scene (rtx_scene_destroy frees the ones left).
 * A frame is used by one thread at a time; different frames of one scene may be used from
 * different threads.  The rtx_scene_set_* switches apply to renders submitted after the call.
 * rtx_render(scene, ...) is rtx_render_async + rtx_frame_wait on a frame the scene owns. */
typedef struct rtx_frame rtx_frame;
int rtx_frame_create(rtx_scene* scene, rtx_frame** out);
int rtx_render_async(rtx_frame* frame, const rtx_camera* camera,
                     int32_t tile_w, int32_t tile_h, int32_t tile_first, int32_t tile_stride,
                     uint8_t* rgb_out, int32_t rgb_is_device, rtx_hit_id* hit_ids);
int rtx_frame_wait(rtx_frame* frame, rtx_stats* stats);
void rtx_frame_destroy(rtx_frame* frame);

/* Number of bytes rtx_render writes to rgb_out for the given tiling. */
int64_t rtx_render_output_bytes(const rtx_camera* camera, int32_t tile_w, int32_t tile_h,
                                int32_t tile_first, int32_t tile_stride);

/* Scatter packed tiles (as written by rtx_render for each rank) back into a full
 * row-major frame.  Host-side helper for the multi-GPU tile gather. */
int rtx_unpack_tiles(const uint8_t* packed, int32_t width, int32_t height,
                     int32_t tile_w, int32_t tile_h, int32_t tile_first, int32_t tile_stride,
                     uint8_t* frame);

/* The root's side of a frame rendered as interleaved tiles by n_parts ranks (the counterpart of
 * the join of the reference's row bands, raytracer.cpp:124-130): parts[r] is rank r's packed
 * buffer -- tiles r, r + n_parts, ... as rtx_render writes them with tile_first = r,
 * tile_stride = n_parts -- in DEVICE memory on `device` (e.g. where an RCCL gather has put
 * them; a rank without tiles may pass NULL).  The tiles are scattered into one row-major frame
 * on the device and the frame is copied to frame_out (host memory, or device memory on `device`
 * when frame_is_device != 0): one kernel and one copy instead of a copy and a host-side
 * rtx_unpack_tiles per rank.  The buffers must be complete when the call is made (it does not
 * wait for other streams).  Blocking; at most 64 parts. */
int rtx_assemble_tiles(int32_t device, const uint8_t* const* parts, int32_t n_parts,
                       int32_t width, int32_t height, int32_t tile_w, int32_t tile_h,
                       uint8_t* frame_out, int32_t frame_is_device);

/* MeshBVH::build (bvh.cpp:80-102, 187-258) on the device: the reference's tree for one mesh --
 * same boxes, topology, split axes, pre-order numbering and leaf face sets -- from a
 * level-synchronous build (segmented reduction + segmented sort per level) instead of the
 * recursive nth_element build.  Faces inside a leaf come out in sorted order (the reference:
 * whatever nth_element left), and faces whose centroids tie exactly at a median may change sides.
 * `faces`: n_faces x 3 indices into `vertices` (3 doubles each); `nodes_out`: room for
 * 2 * n_faces nodes; `face_indices_out`: n_faces entries; `build_ms` (optional): wall time
 * including the transfers.  The result goes to rtx_mesh.nodes / face_indices (or to
 * rtx_host_attach_bvh). */
int rtx_build_bvh(int32_t device, const double* vertices, int64_t n_vertices, const int32_t* faces, int32_t n_faces,
                  rtx_bvh_node* nodes_out, int32_t* n_nodes_out, int32_t* face_indices_out, double* build_ms);

/* Collect per-ray work counters in rtx_stats (slower kernels); default off.
 * With counters on, the device carries out every closest-hit query to its end, so that ray
 * and triangle-test counts equal those of the reference's algorithm (utils.cpp:671-865).
 * With counters off, a shadow ray (isInShadow, utils.cpp:1066-1084) is retired at the first
 * accepted hit that lies in front of its light: the answer `hit && distance < |L - P|` can
 * no longer change after that, so pictures and hit records are the same in both modes. */
int rtx_scene_set_counters(rtx_scene* scene, int32_t enable);

/* Consistent mode; default off.  It has no counterpart in the reference: it departs from the
 * reference's results where those depend on two properties of its transformed-object code,
 * and is not covered by the parity guarantee of the default mode.
 *  - rayHitsMesh searches a transformed mesh or instance with the *world-space* bound against
 *    *object-space* distances (utils.cpp:476-531), and a hit that turns out farther than the
 *    bound is dropped but has already overwritten the record (utils.cpp:559-570): scaled
 *    instances lose or gain hits.  In consistent mode the bound is carried into object space
 *    with the ray and a dropped hit leaves no trace.
 *  - isInShadow (utils.cpp:1066-1084) runs a closest-hit query.  In consistent mode a shadow ray
 *    starts with the light's distance as its bound and ends at the first hit it accepts.
 * Everything else (object order, tests, shading, roundings) is unchanged; scenes without
 * transformed meshes and instances give the same picture in both modes.  Work counters
 * (rtx_scene_set_counters) cannot be combined with it: rtx_render returns RTX_EINVAL. */
int rtx_scene_set_consistent(rtx_scene* scene, int32_t enable);

/* Eight-wide BLAS; default off.  The part of the non-parity "fast traversal" row that concerns the
 * boxes -- FP32 conservative boxes in a wide BVH -- built so that it is exact all the same: the
 * meshes' traversal runs over 256-byte records that hold three levels of the reference's tree
 * (bvh.cpp:187-258) with float boxes rounded outward, one stack entry per record, children in the
 * reference's near-to-far order (bvh.cpp:139-147).  Float boxes can only let a ray into MORE leaves
 * than the reference's FP64 test (bvh.cpp:23-56); every triangle test is the reference's
 * (utils.cpp:327-370, strict t < t_min), and at a job's end the winning triangle's leaf is checked
 * against the reference's exact box test -- if that fails anywhere in a frame (a hit within rounding
 * of its leaf's faces), the frame is rendered again with the four-wide kernels
 * (rtx_stats.wide_blas_redone).  Pictures and hit records are those of the default path; the tests
 * assert it.  The records exist only in a scene created with RTX_SCENE_WIDE_BLAS in
 * rtx_scene_desc::build_flags (RTX_EINVAL otherwise).  Used for scenes with at most four instances and
 * leaves of at most 14 triangles, in the default and the consistent mode, never with work counters on
 * (rtx_stats.wide_blas says whether a frame used it).  On an MI355X it fetches a fifth fewer records
 * and takes the same time -- a step's time follows its instruction count, which eight float boxes
 * do not lower against four double ones -- which is why it is off by default. */
int rtx_scene_set_wide_blas(rtx_scene* scene, int32_t enable);

/* rtx_stats.trace_ms is measured with a pair of events round every ray generation's trace
 * launches (on by default).  The markers cost a frame about 50 microseconds -- 3-4 % of an
 * 800x800 frame: callers that do not read trace_ms (the animation pipeline of the command line)
 * turn them off; trace_ms is then 0.  No counterpart in the reference. */
int rtx_scene_set_generation_timing(rtx_scene* scene, int32_t enable);

/* Cap, in bytes, of the two ray queues together (default 32 GiB). */
int rtx_scene_set_queue_limit(rtx_scene* scene, int64_t bytes);

/* Bytes of the records the traversal reads -- collapsed BVH nodes, triangles, object records -- in
 * device memory: the working set of the closest-hit kernel (a measurement aid; no counterpart in
 * the reference). */
int64_t rtx_scene_record_bytes(const rtx_scene* scene);

/* Development switches, for tests and measurements (no counterpart in the reference); every one
 * leaves pictures and hit records unchanged.  A scene starts with the switches named in the
 * process environment when the process creates its first scene (RTX_AMD_NO_SEARCH=1 ... for the
 * tools under tools/); this call replaces them, for renders submitted after it. */
enum {
    RTX_DEBUG_NO_SCATTER = 1,        /* generation 0 hands out pixel blocks in image order            */
    RTX_DEBUG_POISON_QUEUES = 2,     /* ray queues and hit records start every attempt as 0xFF bytes  */
    RTX_DEBUG_GEN_TIMING = 4,        /* per-generation kernel times and counters on stderr            */
    RTX_DEBUG_NO_WIDE = 8,           /* no wave-wide traversal of a generation's last jobs            */
    RTX_DEBUG_NO_SEARCH = 16,        /* instances always replayed in index order (utils.cpp:756-790)  */
    RTX_DEBUG_NO_WIDE_OLD = 32,      /* wave-wide traversal only in generations of <= 512 rays        */
    RTX_DEBUG_ALL_GENERATIONS = 64,  /* every ray generation launched, empty or not                   */
    RTX_DEBUG_HALF_GRID = 128,       /* the trace grid at one wave per SIMD instead of two (a measurement of how
                                        much of the kernel's time is latency that more waves would hide)   */
    RTX_DEBUG_FAIL_WIDE_BLAS_CHECK = 256, /* rtx_scene_set_wide_blas: behave as if the end-of-job check had failed (the
                                        frame is then redone with the four-wide kernels: exercises that path)    */
    RTX_DEBUG_NO_EARLY_SHADOW = 512, /* shadow rays that no object can hold up are queued and traced like the others
                                     * instead of being settled where they are emitted                          */
    RTX_DEBUG_ALL = 1023
};
int rtx_scene_set_debug(rtx_scene* scene, uint32_t flags);

void rtx_scene_destroy(rtx_scene* scene);

const char* rtx_last_error(void);
int rtx_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_AMD_H */
