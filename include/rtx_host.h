/*
 * rtx_host.h -- C-ABI of the host-side scene pipeline that feeds rtx_amd.h.
 *
 * This is the host half of the seam: it produces, from a scene file in the reference's
 * JSON(+PLY) format, the finished data the reference holds after
 * raytracer.cpp:216-246 -- parsed scene (scene.cpp:62-672), per-mesh BVH
 * (bvh.cpp:80-102,187-258), object/instance/camera/light transforms
 * (scene.cpp:827-990, utils.cpp:1184-1361) and per-mesh vertex normals
 * (precompute.cpp:9-44) -- laid out as an `rtx_scene_desc`.  A host that keeps the
 * reference's own loader does not need this library: it fills `rtx_scene_desc` from its
 * `scene::Scene` directly.
 *
 * All functions return 0 on success, negative on failure (rtx_host_last_error()).
 */
#ifndef RTX_HOST_H
#define RTX_HOST_H

#include "rtx_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtx_host_scene rtx_host_scene;

/* Scene::loadSceneFromFile + buildBVH (when use_bvh) + precomputeTransformations +
 * precomputeMeshNormals.  enable_cull is the -c flag (raytracer.cpp:199-207, default 0),
 * use_bvh the -b flag (raytracer.cpp:190-198, default 1). */
int rtx_host_load(const char* json_path, int use_bvh, int enable_cull, rtx_host_scene** out);

/* Next frame of an animation.  The reference starts one process per frame file and redoes
 * parsing, BVH build and precompute each time (runner.py:105-141); frames of an animation
 * usually differ only in cameras and point lights.  When `json_path` describes the geometry
 * `prev` holds -- same document apart from "Cameras" and "Lights"/"PointLight", same
 * directory, PLY files unchanged on disk -- only cameras and lights are re-read into `prev`
 * (transformations applied as in scene.cpp:971-987), `*out = prev`, `*reused = 1`, and the
 * description's geometry pointers stay as they were: a device scene made from it needs
 * rtx_scene_update_lights only.  Otherwise the file is loaded from scratch with prev's flags,
 * `prev` is freed, `*out` is the new scene and `*reused = 0`.  On failure `prev` stays valid. */
int rtx_host_load_next(const char* json_path, rtx_host_scene* prev, rtx_host_scene** out, int* reused);

/* Give mesh `mesh` of a scene loaded with use_bvh = 0 a BVH built elsewhere -- in practice by
 * rtx_build_bvh (rtx_amd.h), the device-side MeshBVH::build -- in the reference's layout
 * (bvh.h:50-66).  The arrays are copied; the description's mesh entry points at the copies.
 * World boxes of transformed objects do not change: the root box the reference takes them from
 * (scene.cpp:867-879) equals the box over the mesh's vertices used without a BVH. */
int rtx_host_attach_bvh(rtx_host_scene* scene, int mesh, const rtx_bvh_node* nodes, int n_nodes,
                        const int32_t* face_indices);

/* The description; valid until rtx_host_free (light array: until the next rtx_host_load_next). */
const rtx_scene_desc* rtx_host_desc(const rtx_host_scene* scene);

int rtx_host_camera_count(const rtx_host_scene* scene);

/* Camera `index` (after its own Transformations were baked in, scene.cpp:971-979) and its
 * ImageName (truncated to name_cap-1 bytes). */
int rtx_host_camera(const rtx_host_scene* scene, int index, rtx_camera* out,
                    char* image_name, int name_cap);

/* Milliseconds spent in BVH build + transform/normal precompute (the reference's
 * preprocessingTimeMs, raytracer.cpp:229-249) and in parsing. */
int rtx_host_timings(const rtx_host_scene* scene, double* parse_ms, double* preprocess_ms);

/* Sectioned binary dump of the finished scene: repeated {char name[24]; u64 nbytes;
 * payload}.  Used by the tests to compare this pipeline bit-for-bit with the reference's
 * own Scene object. */
int rtx_host_dump(const rtx_host_scene* scene, const char* path);

void rtx_host_free(rtx_host_scene* scene);

/* ASCII P3 writer with the reference's exact byte layout (scene.cpp:675-697); the
 * extension of `path` is replaced by .ppm exactly as the reference does. */
int rtx_host_write_ppm(const char* path, const uint8_t* rgb, int width, int height);

/* 8-bit RGB PNG writer (stored deflate blocks, no external library). */
int rtx_host_write_png(const char* path, const uint8_t* rgb, int width, int height);

const char* rtx_host_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTX_HOST_H */
