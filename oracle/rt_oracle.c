/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU, recursive restatement of the reference renderer's per-pixel
 * pipeline, evaluated over the same `rtx_scene_desc` the device library consumes.
 * It exists only to check the HIP path: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; nothing the product ships links it.
 *
 * Parity is PINNED: tests/test_oracle.py and tests/test_fuzz_cpu.py compare this file's images and
 * hit records byte-for-byte with the compiled reference (oracle/_ref, built from the
 * reference's own sources) on every unit scene, and tests/golden/ holds the reference's
 * outputs for the same scenes so that the check also runs where the reference is absent.
 *
 * Each function cites the reference lines it restates.  Arithmetic is FP64 in the
 * reference's operation order; build with -ffp-contract=off (the reference binary has
 * no fused multiply-adds).  Float roundings (sqrtf, float literals widened to double)
 * are reproduced where the reference has them.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#include "rtx_amd.h"

typedef struct { double x, y, z; } v3;

typedef struct {
    uint64_t rays, primary_rays, shadow_rays, secondary_rays;
    uint64_t node_pops;        /* BVH nodes whose box was tested (bvh.cpp:132)            */
    uint64_t tri_tests;        /* rayHitsTriangle calls (utils.cpp:298)                   */
    uint64_t world_box_tests;  /* world AABB tests (utils.cpp:451, 761)                   */
    uint64_t ray_transforms;   /* rays taken to object space (utils.cpp:481-489)          */
    uint64_t transformed_hits; /* accepted back-transformed hits (utils.cpp:572-659)      */
    uint64_t shades;           /* computeShading calls with a material                    */
    /* how far the triangle tests got (utils.cpp:342, 352, 359): beyond the determinant guard and
     * culling, beyond the beta test, beyond the gamma test -- the stages differ in arithmetic */
    uint64_t tri_past_det, tri_past_beta, tri_past_gamma;
} rt_oracle_counters;

/* Intersection (scene.h:248-264) + the producing object for attribution. */
typedef struct {
    int hit;
    double distance;
    v3 point, gn, sn;
    int kind, container, face;
    double beta, gamma;
    int material;
    int src;
} isect_t;

typedef struct { v3 o, d; int depth; } ray_t;

/* consistent / any_hit: NOT reference behaviour.  They restate the device library's optional
 * "consistent" mode (rtx_scene_set_consistent) so that this mode too has a CPU checker:
 *  - a transformed mesh or instance is searched with the ray's bound carried into object
 *    space (bound / |M^-1 d| as the reference normalises it), and a hit whose world distance
 *    is not below the bound leaves bound and record untouched;
 *  - a shadow ray starts with the light's distance as its bound and is occluded iff any hit
 *    is accepted (any_hit: a transformed mesh accepts on the object-space test alone); the hit
 *    flag of a record means exactly that.
 * With both flags 0 every function below is the reference's. */
typedef struct {
    const rtx_scene_desc* s;
    rt_oracle_counters* c;
    int consistent, any_hit;
} ctx_t;

/* ---- overloads.cpp:10-65 ------------------------------------------------------------ */
static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, double b) { return V(a.x * b, a.y * b, a.z * b); }
static inline v3 sscale(double a, v3 b) { return V(a * b.x, a * b.y, a * b.z); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline double dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize(v3 a) {
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return V(a.x / len, a.y / len, a.z / len);
}
static inline double dmax(double a, double b) { return (a < b) ? b : a; }  /* std::max */
static inline double dmin(double a, double b) { return (b < a) ? b : a; }  /* std::min */
static inline v3 ld3(const double* p) { return V(p[0], p[1], p[2]); }

/* ---- utils.cpp:96-123 --------------------------------------------------------------- */
static inline v3 xf_point(const double* m, v3 p) {
    double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w > (double)1.00001f || w < (double)0.99999f) {
        double invW = 1.0 / w;
        return V(x * invW, y * invW, z * invW);
    }
    return V(x, y, z);
}
static inline v3 xf_dir(const double* m, v3 d) {
    return V(m[0] * d.x + m[1] * d.y + m[2] * d.z,
             m[4] * d.x + m[5] * d.y + m[6] * d.z,
             m[8] * d.x + m[9] * d.y + m[10] * d.z);
}

/* ---- utils.cpp:134-158 -------------------------------------------------------------- */
static ray_t cast_ray(const rtx_camera* cam, int x, int y, int width, int height) {
    v3 w = vneg(normalize(ld3(cam->gaze)));
    v3 v = normalize(ld3(cam->up));
    v3 u = cross(v, w);
    double l = cam->near_plane[0], r = cam->near_plane[1];
    double b = cam->near_plane[2], t = cam->near_plane[3];
    double s_u = (x + 0.5) * (r - l) / width;
    double s_v = (y + 0.5) * (t - b) / height;
    v3 e = ld3(cam->position);
    v3 m = vsub(e, vscale(w, cam->near_distance));
    v3 q = vadd(vadd(m, sscale(l, u)), sscale(t, v));
    v3 s = vsub(vadd(q, vscale(u, s_u)), vscale(v, s_v));
    ray_t ray;
    ray.o = e;
    ray.d = normalize(vsub(s, e));
    ray.depth = 0;
    return ray;
}

/* ---- bvh.cpp:23-56 ------------------------------------------------------------------ */
static int aabb_hit(const double* bmin, const double* bmax, const ray_t* ray, double t_min,
                    double t_max) {
    double t0 = t_min, t1 = t_max;
    const double o[3] = {ray->o.x, ray->o.y, ray->o.z};
    const double d[3] = {ray->d.x, ray->d.y, ray->d.z};
    for (int a = 0; a < 3; a++) {
        if (fabs(d[a]) < (double)1e-9f) {
            if (o[a] < bmin[a] || o[a] > bmax[a]) return 0;
            continue;
        }
        double invD = 1.0 / d[a];
        double tNear = (bmin[a] - o[a]) * invD;
        double tFar = (bmax[a] - o[a]) * invD;
        if (tNear > tFar) { double tmp = tNear; tNear = tFar; tFar = tmp; }
        t0 = dmax(t0, tNear);
        t1 = dmin(t1, tFar);
        if (t1 < t0) return 0;
    }
    return t1 >= t0;
}

/* ---- utils.cpp:298-394 -------------------------------------------------------------- */
static int tri_hit(const ctx_t* cx, const ray_t* ray, const int32_t* face, double* t_min,
                   isect_t* is, int material, int container, int face_index) {
    const rtx_scene_desc* s = cx->s;
    cx->c->tri_tests++;
    const v3 a = ld3(s->vertices + 3 * (size_t)face[0]);
    const v3 b = ld3(s->vertices + 3 * (size_t)face[1]);
    const v3 c = ld3(s->vertices + 3 * (size_t)face[2]);
    const v3 e1 = vsub(b, a), e2 = vsub(c, a);
    const v3 gn = normalize(cross(e1, e2));
    if (s->enable_back_face_culling) {
        if (dot(ray->d, gn) > 0.0) return 0;
    }
    const double dx = ray->d.x, dy = ray->d.y, dz = ray->d.z;
    const double e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const double e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
    const double rx = ray->o.x - a.x, ry = ray->o.y - a.y, rz = ray->o.z - a.z;
    double det = -(e1x * (e2y * dz - e2z * dy) - e1y * (e2x * dz - e2z * dx) +
                   e1z * (e2x * dy - e2y * dx));
    if (fabs(det) < 1e-15) return 0;
    const double invDet = 1.0 / det;
    cx->c->tri_past_det++;
    double detB = -(rx * (e2y * dz - e2z * dy) - ry * (e2x * dz - e2z * dx) +
                    rz * (e2x * dy - e2y * dx));
    double beta = detB * invDet;
    if (beta < 0.0 || beta > 1.0) return 0;
    cx->c->tri_past_beta++;
    double detG = -(e1x * (ry * dz - rz * dy) - e1y * (rx * dz - rz * dx) +
                    e1z * (rx * dy - ry * dx));
    double gamma = detG * invDet;
    if (gamma < 0.0 || gamma > 1.0 || beta + gamma > 1.0) return 0;
    cx->c->tri_past_gamma++;
    double detT = e1x * (e2y * rz - e2z * ry) - e1y * (e2x * rz - e2z * rx) +
                  e1z * (e2x * ry - e2y * rx);
    double t = detT * invDet;
    if (t < s->intersection_test_epsilon) return 0;
    if (t < *t_min) {
        *t_min = t;
        is->hit = 1;
        is->distance = t;
        is->point = vadd(vadd(a, sscale(beta, vsub(b, a))), sscale(gamma, vsub(c, a)));
        is->gn = gn;
        is->sn = gn;
        is->beta = beta;
        is->gamma = gamma;
        is->material = material;
        is->container = container;
        is->face = face_index;
        return 1;
    }
    return 0;
}

/* ---- bvh.cpp:104-185 ---------------------------------------------------------------- */
static int bvh_traverse(const ctx_t* cx, const ray_t* ray, const rtx_mesh* mesh, double* t_min,
                        isect_t* is, int mesh_index, int material) {
    if (mesh->n_nodes <= 0 || !mesh->nodes) {
        /* no BVH (-b 0): brute-force loop in face order (utils.cpp:434-441, 519-526) */
        int hit = 0;
        for (int i = 0; i < mesh->n_faces; i++)
            if (tri_hit(cx, ray, mesh->faces + 3 * (size_t)i, t_min, is, material, mesh_index, i)) {
                hit = 1;
                is->kind = RTX_KIND_MESH;
            }
        return hit;
    }
    int hit = 0;
    int stack[64];
    int sp = 0;
    stack[sp++] = 0;
    const double dir[3] = {ray->d.x, ray->d.y, ray->d.z};
    while (sp > 0) {
        const rtx_bvh_node* n = &mesh->nodes[stack[--sp]];
        cx->c->node_pops++;
        if (!aabb_hit(n->bounds_min, n->bounds_max, ray, 0.0, *t_min)) continue;
        if (n->left == -1) {
            for (int i = 0; i < n->face_count; i++) {
                int f = mesh->face_indices[n->first_face + i];
                if (f < 0 || f >= mesh->n_faces) continue;
                if (tri_hit(cx, ray, mesh->faces + 3 * (size_t)f, t_min, is, material, mesh_index,
                            f)) {
                    hit = 1;
                    is->kind = RTX_KIND_MESH;
                }
            }
        } else {
            if (n->left < 0 || n->right < 0) continue;
            if (dir[n->split_axis] >= 0.0) { stack[sp++] = n->right; stack[sp++] = n->left; }
            else                           { stack[sp++] = n->left;  stack[sp++] = n->right; }
        }
    }
    return hit;
}

/* ---- utils.cpp:396-669 --------------------------------------------------------------
 * xform_index < 0: untransformed test straight in world space (utils.cpp:427-442).
 * Otherwise world-AABB early-out, object-space traversal bounded by the *world* t_min,
 * float-rounded world distance, and -- when that distance is not smaller than the bound
 * the call started with -- restore t_min but leave the record as the traversal wrote it
 * (utils.cpp:567-570). */
static int mesh_hit(const ctx_t* cx, const ray_t* ray, int mesh_index, int xform_index,
                    int material, double* t_min, isect_t* is) {
    const rtx_scene_desc* s = cx->s;
    const rtx_mesh* mesh = &s->meshes[mesh_index];
    if (xform_index < 0) return bvh_traverse(cx, ray, mesh, t_min, is, mesh_index, material);
    const rtx_xform* xf = &s->xforms[xform_index];
    if (xf->has_bounds) {
        cx->c->world_box_tests++;
        if (!aabb_hit(xf->bounds_min, xf->bounds_max, ray, 0.0, *t_min)) return 0;
    }
    const double original_t_min = *t_min;
    cx->c->ray_transforms++;
    ray_t oray;
    oray.o = xf_point(xf->inverse, ray->o);
    v3 od = xf_dir(xf->inverse, ray->d);
    double lenSq = od.x * od.x + od.y * od.y + od.z * od.z;
    double invLen = 1.0 / sqrtf(lenSq);
    oray.d = V(od.x * invLen, od.y * invLen, od.z * invLen);
    oray.depth = ray->depth;
    isect_t before;
    if (cx->consistent) {
        before = *is;
        double bound = *t_min / invLen;
        if (!bvh_traverse(cx, &oray, mesh, &bound, is, mesh_index, material)) return 0;
        if (cx->any_hit) return 1;
    } else if (!bvh_traverse(cx, &oray, mesh, t_min, is, mesh_index, material)) return 0;

    v3 wp = xf_point(xf->transform, is->point);
    double dx = wp.x - ray->o.x, dy = wp.y - ray->o.y, dz = wp.z - ray->o.z;
    double worldDistance = sqrtf(dx * dx + dy * dy + dz * dz);
    if (worldDistance >= original_t_min) {
        if (cx->consistent) *is = before;
        *t_min = original_t_min;
        return 0;
    }
    cx->c->transformed_hits++;
    v3 wg = xf_dir(xf->normal, is->gn);
    double nl = wg.x * wg.x + wg.y * wg.y + wg.z * wg.z;
    double inl = 1.0 / sqrtf(nl);
    wg.x *= inl; wg.y *= inl; wg.z *= inl;
    v3 ws;
    int smooth_done = 0;
    if (mesh->smooth && mesh->vertex_normals && is->face >= 0 && is->face < mesh->n_faces) {
        const int32_t* f = mesh->faces + 3 * (size_t)is->face;
        if (f[0] < s->n_vertices && f[1] < s->n_vertices && f[2] < s->n_vertices) {
            double u = is->beta, v = is->gamma, w = 1.0 - u - v;
            const double* n0 = mesh->vertex_normals + 3 * (size_t)f[0];
            const double* n1 = mesh->vertex_normals + 3 * (size_t)f[1];
            const double* n2 = mesh->vertex_normals + 3 * (size_t)f[2];
            double nx = w * n0[0] + u * n1[0] + v * n2[0];
            double ny = w * n0[1] + u * n1[1] + v * n2[1];
            double nz = w * n0[2] + u * n1[2] + v * n2[2];
            double nlen = sqrtf(nx * nx + ny * ny + nz * nz);
            double ninv = 1.0 / nlen;
            nx *= ninv; ny *= ninv; nz *= ninv;
            ws = xf_dir(xf->normal, V(nx, ny, nz));
            smooth_done = 1;
        }
    }
    if (!smooth_done) ws = xf_dir(xf->normal, is->sn);
    {
        double slen = sqrtf(ws.x * ws.x + ws.y * ws.y + ws.z * ws.z);
        double sinv = 1.0 / slen;
        ws.x *= sinv; ws.y *= sinv; ws.z *= sinv;
    }
    if (dot(wg, ray->d) > 0.0) { wg = vneg(wg); ws = vneg(ws); }
    is->point = wp;
    is->gn = wg;
    is->sn = ws;
    is->distance = worldDistance;
    *t_min = worldDistance;
    is->kind = RTX_KIND_MESH;
    is->container = -2;
    return 1;
}

/* ---- utils.cpp:160-219 -------------------------------------------------------------- */
static int plane_hit(const ctx_t* cx, const ray_t* ray, int index, double* t_min, isect_t* is) {
    const rtx_scene_desc* s = cx->s;
    const rtx_plane* pl = &s->planes[index];
    const rtx_xform* xf = pl->xform >= 0 ? &s->xforms[pl->xform] : NULL;
    v3 o = ray->o, d = ray->d;
    v3 n = ld3(pl->normal);
    v3 a = ld3(s->vertices + 3 * (size_t)pl->point);
    if (xf) {
        o = xf_point(xf->inverse, ray->o);
        d = normalize(xf_dir(xf->inverse, ray->d));
    }
    double denom = dot(d, n);
    if (fabs(denom) < 1e-9) return 0;
    double t = dot(vsub(a, o), n) / denom;
    if (t > 0 && t < *t_min) {
        v3 op = vadd(o, vscale(d, t));
        if (xf) {
            v3 wp = xf_point(xf->transform, op);
            v3 wn = normalize(xf_dir(xf->normal, n));
            double wd = sqrt(dot(vsub(wp, ray->o), vsub(wp, ray->o)));
            if (wd < *t_min) {
                *t_min = wd;
                is->hit = 1; is->distance = wd; is->point = wp; is->gn = wn; is->sn = wn;
                is->material = pl->material; is->kind = RTX_KIND_PLANE; is->container = index;
                is->src = index;
                return 1;
            }
        } else {
            *t_min = t;
            is->hit = 1; is->distance = t; is->point = op; is->gn = n; is->sn = n;
            is->material = pl->material; is->kind = RTX_KIND_PLANE; is->container = index;
            is->src = index;
            return 1;
        }
    }
    return 0;
}

/* ---- utils.cpp:221-295 -------------------------------------------------------------- */
static int sphere_hit(const ctx_t* cx, const ray_t* ray, int index, double* t_min, isect_t* is) {
    const rtx_scene_desc* s = cx->s;
    const rtx_sphere* sp = &s->spheres[index];
    const rtx_xform* xf = sp->xform >= 0 ? &s->xforms[sp->xform] : NULL;
    v3 o = ray->o, d = ray->d;
    if (xf) {
        o = xf_point(xf->inverse, ray->o);
        v3 dir = xf_dir(xf->inverse, ray->d);
        double lenSq = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z;
        double invLen = 1.0 / sqrtf(lenSq);
        d = V(dir.x * invLen, dir.y * invLen, dir.z * invLen);
    }
    v3 c = ld3(s->vertices + 3 * (size_t)sp->center);
    double r = sp->radius;
    v3 oc = vsub(o, c);
    double D = dot(d, oc) * dot(d, oc) - (dot(d, d) * (dot(oc, oc) - r * r));
    if (D < 0) return 0;
    double t1 = (-dot(d, oc) + sqrt(D)) / dot(d, d);
    double t2 = (-dot(d, oc) - sqrt(D)) / dot(d, d);
    double t = dmin(t1, t2);
    v3 op = vadd(o, vscale(d, t));
    v3 on = normalize(vsub(op, c));
    if (t > 0 && t < *t_min) {
        if (xf) {
            v3 wp = xf_point(xf->transform, op);
            v3 wn = xf_dir(xf->normal, on);
            double nlen = sqrtf(wn.x * wn.x + wn.y * wn.y + wn.z * wn.z);
            wn.x /= nlen; wn.y /= nlen; wn.z /= nlen;
            double dx = wp.x - ray->o.x, dy = wp.y - ray->o.y, dz = wp.z - ray->o.z;
            double wd = sqrtf(dx * dx + dy * dy + dz * dz);
            if (wd < *t_min) {
                *t_min = wd;
                is->hit = 1; is->distance = wd; is->point = wp; is->gn = wn; is->sn = wn;
                is->material = sp->material; is->kind = RTX_KIND_SPHERE; is->container = index;
                is->src = index;
                return 1;
            }
        } else {
            *t_min = t;
            is->hit = 1; is->distance = t; is->point = op; is->gn = on; is->sn = on;
            is->material = sp->material; is->kind = RTX_KIND_SPHERE; is->container = index;
            is->src = index;
            return 1;
        }
    }
    return 0;
}

/* ---- utils.cpp:671-865 -------------------------------------------------------------- */
static isect_t intersect_within(const ctx_t* cx, const ray_t* ray, double t_min) {
    const rtx_scene_desc* s = cx->s;
    cx->c->rays++;
    isect_t is;
    memset(&is, 0, sizeof is);
    is.kind = RTX_KIND_NONE; is.container = -1; is.face = -1; is.material = -1; is.src = -1;
    int hit = 0;

    for (int i = 0; i < s->n_planes; i++) hit = plane_hit(cx, ray, i, &t_min, &is) || hit;

    for (int i = 0; i < s->n_triangles; i++) {                       /* utils.cpp:692-729 */
        const rtx_triangle* tr = &s->triangles[i];
        if (tr->xform >= 0) {
            const rtx_xform* xf = &s->xforms[tr->xform];
            ray_t oray;
            oray.o = xf_point(xf->inverse, ray->o);
            oray.d = normalize(xf_dir(xf->inverse, ray->d));
            oray.depth = ray->depth;
            double original_t_min = t_min;
            if (tri_hit(cx, &oray, tr->v, &t_min, &is, tr->material, -1, i)) {
                v3 wp = xf_point(xf->transform, is.point);
                v3 wg = normalize(xf_dir(xf->normal, is.gn));
                v3 ws = normalize(xf_dir(xf->normal, is.sn));
                double wd = sqrt(dot(vsub(wp, ray->o), vsub(wp, ray->o)));
                if (wd < original_t_min) {
                    is.point = wp; is.gn = wg; is.sn = ws; is.distance = wd;
                    t_min = wd; hit = 1; is.kind = RTX_KIND_TRIANGLE; is.src = -1;
                } else {
                    t_min = original_t_min;
                }
            }
        } else if (tri_hit(cx, ray, tr->v, &t_min, &is, tr->material, -1, i)) {
            hit = 1; is.kind = RTX_KIND_TRIANGLE; is.src = -1;
        }
    }

    for (int i = 0; i < s->n_meshes; i++) {                          /* utils.cpp:736-739 */
        const rtx_mesh* m = &s->meshes[i];
        int h = mesh_hit(cx, ray, i, m->xform, m->material, &t_min, &is);
        if (h) is.src = i;
        hit = h || hit;
    }

    for (int i = 0; i < s->n_spheres; i++) hit = sphere_hit(cx, ray, i, &t_min, &is) || hit;

    for (int i = 0; i < s->n_instances; i++) {                       /* utils.cpp:756-790 */
        const rtx_instance* in = &s->instances[i];
        if (in->base_mesh < 0) continue;
        /* An instance without a matrix of its own is tested through the base mesh's
         * transform, if any (hasTransform = mesh.hasTransformation || transformMatrix,
         * utils.cpp:418); the instance-loop AABB test (utils.cpp:760-766) repeats the one
         * inside rayHitsMesh with the same box and bound, so one test decides both. */
        int xf = in->xform >= 0 ? in->xform : s->meshes[in->base_mesh].xform;
        if (mesh_hit(cx, ray, in->base_mesh, xf, in->material, &t_min, &is)) {
            hit = 1;
            is.src = s->n_meshes + i;
        }
    }

    /* consistent mode: the flag of the returned record means "a hit was accepted" (in the
     * reference a transformed free triangle that is found but then rejected leaves the flag set,
     * utils.cpp:692-729, which its callers read) */
    if (cx->consistent) is.hit = hit;
    if (!hit) return is;

    const double u = is.beta, v = is.gamma, w = 1.0 - u - v;          /* utils.cpp:807-853 */
    is.gn = normalize(is.gn);
    is.sn = normalize(is.sn);
    if (is.kind == RTX_KIND_MESH && is.container != -2) {
        int mi = is.container;
        if (mi >= 0 && mi < s->n_meshes) {
            const rtx_mesh* m = &s->meshes[mi];
            if (m->smooth && m->vertex_normals && is.face >= 0 && is.face < m->n_faces) {
                const int32_t* f = m->faces + 3 * (size_t)is.face;
                if (f[0] < s->n_vertices && f[1] < s->n_vertices && f[2] < s->n_vertices) {
                    v3 n0 = ld3(m->vertex_normals + 3 * (size_t)f[0]);
                    v3 n1 = ld3(m->vertex_normals + 3 * (size_t)f[1]);
                    v3 n2 = ld3(m->vertex_normals + 3 * (size_t)f[2]);
                    v3 sm = normalize(vadd(vadd(sscale(w, n0), sscale(u, n1)), sscale(v, n2)));
                    if (m->xform >= 0) sm = normalize(xf_dir(s->xforms[m->xform].normal, sm));
                    is.sn = sm;
                    if (dot(is.gn, ray->d) > 0.0) { is.gn = vneg(is.gn); is.sn = vneg(is.sn); }
                    else if (dot(is.sn, is.gn) < 0.0) is.sn = vneg(is.sn);
                }
            }
        }
    }
    return is;
}

static isect_t intersect(const ctx_t* cx, const ray_t* ray) { return intersect_within(cx, ray, DBL_MAX); }

static v3 pixel_color(const ctx_t* cx, const ray_t* ray, const isect_t* is);

/* utils.cpp:1167-1182 */
static ray_t reflect_ray(const ray_t* ray, v3 normal, v3 point, double eps) {
    v3 n = normalize(normal);
    v3 dir = vsub(ray->d, sscale(2.0 * dot(ray->d, n), n));
    ray_t r;
    r.o = vadd(point, sscale(eps, n));
    r.d = normalize(dir);
    r.depth = ray->depth + 1;
    return r;
}

/* utils.cpp:1134-1165 */
static ray_t refract_ray(const ray_t* ray, v3 normal, double n1, double n2, v3 point, double eps,
                         int* tir) {
    double eta = n1 / n2;
    double cosTheta = -dot(ray->d, normal);
    cosTheta = dmax(-1.0, dmin(1.0, cosTheta));
    double sinThetaTSq = eta * eta * (1.0 - cosTheta * cosTheta);
    *tir = (sinThetaTSq > 1.0);
    ray_t r;
    if (*tir) { r.o = point; r.d = ray->d; r.depth = ray->depth; return r; }
    double cosPhi = sqrt(dmax(0.0, 1.0 - sinThetaTSq));
    v3 wt = vsub(vscale(vadd(ray->d, vscale(normal, cosTheta)), eta), vscale(normal, cosPhi));
    r.o = vsub(point, sscale(eps, normalize(normal)));
    r.d = normalize(wt);
    r.depth = ray->depth + 1;
    return r;
}

/* utils.cpp:1089-1111 */
static double fresnel_conductor(double cosTheta, double n, double k) {
    cosTheta = dmax(0.0, dmin(1.0, cosTheta));
    double c2 = cosTheta * cosTheta;
    double n2 = n * n, k2 = k * k, nk = n2 + k2;
    double twoNCos = 2.0 * n * cosTheta;
    double Rs = (nk - twoNCos + c2) / (nk + twoNCos + c2);
    double Rp = (nk * c2 - twoNCos + 1.0) / (nk * c2 + twoNCos + 1.0);
    double R = 0.5 * (Rs + Rp);
    if (R < 0.0) R = 0.0;
    if (R > 1.0) R = 1.0;
    return R;
}

/* utils.cpp:1113-1129 */
static double fresnel_dielectric(double cosTheta, double n1, double n2) {
    cosTheta = dmax(-1.0, dmin(1.0, cosTheta));
    double eta = n1 / n2;
    double s2 = eta * eta * (1.0 - cosTheta * cosTheta);
    if (s2 >= 1.0) return 1.0;
    double cosPhi = sqrt(dmax(0.0, 1.0 - s2));
    double rPerp = (n1 * cosTheta - n2 * cosPhi) / (n1 * cosTheta + n2 * cosPhi);
    double rPara = (n2 * cosTheta - n1 * cosPhi) / (n2 * cosTheta + n1 * cosPhi);
    return 0.5 * (rPerp * rPerp + rPara * rPara);
}

/* utils.cpp:1066-1084 */
static int in_shadow(const ctx_t* cx, const rtx_light* light, const isect_t* is) {
    const rtx_scene_desc* s = cx->s;
    v3 L = ld3(light->position);
    ray_t sr;
    sr.o = vadd(is->point, sscale(s->shadow_ray_epsilon, is->gn));
    sr.d = normalize(vsub(L, is->point));
    sr.depth = 0;
    cx->c->shadow_rays++;
    if (cx->consistent) {
        ctx_t any = *cx;
        any.any_hit = 1;
        return intersect_within(&any, &sr, sqrt(dot(vsub(L, is->point), vsub(L, is->point)))).hit;
    }
    isect_t si = intersect(cx, &sr);
    double distToLight = sqrt(dot(vsub(L, is->point), vsub(L, is->point)));
    return si.hit && si.distance < distToLight;
}

/* A child ray whose depth exceeds the limit is traced by the reference and then discarded
 * (utils.cpp:924-926 vs 1047); it is not traced here and not counted as a ray. */
static v3 trace(const ctx_t* cx, const ray_t* ray, isect_t* out_is) {
    if (ray->depth > cx->s->max_recursion_depth) {
        if (out_is) { memset(out_is, 0, sizeof *out_is); }
        return V(0, 0, 0);
    }
    cx->c->secondary_rays++;
    isect_t is = intersect(cx, ray);
    if (out_is) *out_is = is;
    return pixel_color(cx, ray, &is);
}

/* utils.cpp:868-1041 */
static v3 shade(const ctx_t* cx, const ray_t* ray, const isect_t* is) {
    const rtx_scene_desc* s = cx->s;
    v3 color = V(0, 0, 0);
    if (is->material < 0) return color;
    const rtx_material* m = &s->materials[is->material];
    cx->c->shades++;
    v3 normal = normalize(is->sn);
    color = vadd(color, vmul(ld3(m->ambient), ld3(s->ambient)));

    if (m->type == RTX_MAT_CONDUCTOR) {
        for (int li = 0; li < s->n_lights; li++) {
            const rtx_light* light = &s->lights[li];
            if (in_shadow(cx, light, is)) continue;
            v3 lv = vsub(ld3(light->position), is->point);
            double d2 = dot(lv, lv);
            double dist = sqrt(d2);
            v3 ldir = vscale(lv, 1.0 / dist);
            double atten = 1.0 / d2;
            v3 view = normalize(vneg(ray->d));
            v3 half = normalize(vadd(ldir, view));
            double sf = pow(dmax(0.0, dot(normal, half)), m->phong_exponent);
            v3 spec = vscale(vscale(vmul(ld3(m->mirror), ld3(light->intensity)), sf), atten);
            color = vadd(color, spec);
        }
        double cosTheta = dot(vneg(ray->d), normal);
        cosTheta = dmax(0.0, dmin(1.0, cosTheta));
        double F = fresnel_conductor(cosTheta, m->refraction_index, m->absorption_index);
        ray_t rr = reflect_ray(ray, normal, is->point, s->shadow_ray_epsilon);
        v3 rc = trace(cx, &rr, NULL);
        color = vadd(color, vmul(sscale(F, ld3(m->mirror)), rc));
        return color;
    }

    if (m->type == RTX_MAT_DIELECTRIC) {
        double n1 = 1.0, n2 = m->refraction_index;
        int entering = dot(ray->d, normal) < 0.0;
        if (!entering) { normal = vneg(normal); double t = n1; n1 = n2; n2 = t; }
        double cosTheta = dot(vneg(ray->d), normal);
        cosTheta = dmax(-1.0, dmin(1.0, cosTheta));
        double F = fresnel_dielectric(cosTheta, n1, n2);
        ray_t rr = reflect_ray(ray, normal, is->point, s->shadow_ray_epsilon);
        v3 rc = trace(cx, &rr, NULL);
        int tir = 0;
        ray_t tr = refract_ray(ray, normal, n1, n2, is->point, s->shadow_ray_epsilon, &tir);
        if (tir) return vadd(color, rc);
        isect_t ti;
        v3 tc = trace(cx, &tr, &ti);
        if (entering && ti.hit) {
            v3 ab = vscale(ld3(m->absorption_coefficient), ti.distance);
            tc = vmul(tc, V(exp(-ab.x), exp(-ab.y), exp(-ab.z)));
        }
        return vadd(color, vadd(sscale(F, rc), sscale(1.0 - F, tc)));
    }

    if (m->is_mirror) {
        ray_t rr = reflect_ray(ray, is->sn, is->point, s->shadow_ray_epsilon);
        v3 mc = trace(cx, &rr, NULL);
        color = vadd(color, vmul(ld3(m->mirror), mc));
    }
    for (int li = 0; li < s->n_lights; li++) {
        const rtx_light* light = &s->lights[li];
        if (in_shadow(cx, light, is)) continue;
        v3 lv = vsub(ld3(light->position), is->point);
        double d2 = dot(lv, lv);
        double dist = sqrt(d2);
        v3 ldir = vscale(lv, 1.0 / dist);
        double atten = 1.0 / d2;
        double df = dmax(0.0, dot(normal, ldir));
        v3 diffuse = vscale(vscale(vmul(ld3(m->diffuse), ld3(light->intensity)), df), atten);
        v3 view = normalize(vneg(ray->d));
        v3 half = normalize(vadd(ldir, view));
        double sf = pow(dmax(0.0, dot(normal, half)), m->phong_exponent);
        v3 spec = vscale(vscale(vmul(ld3(m->specular), ld3(light->intensity)), sf), atten);
        color = vadd(color, vadd(diffuse, spec));
    }
    return color;
}

/* utils.cpp:1046-1061 */
static v3 pixel_color(const ctx_t* cx, const ray_t* ray, const isect_t* is) {
    if (ray->depth > cx->s->max_recursion_depth) return V(0, 0, 0);
    if (is->hit) return shade(cx, ray, is);
    if (ray->depth == 0) return ld3(cx->s->background);
    return V(0, 0, 0);
}

/* ---- driver: raytracer.cpp:45-51, 70-91 --------------------------------------------- */
typedef struct {
    const rtx_scene_desc* s;
    const rtx_camera* cam;
    uint8_t* rgb;
    rtx_hit_id* hits;
    int tid, nthreads, consistent;
    rt_oracle_counters c;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    ctx_t cx = {j->s, &j->c, j->consistent, 0};
    int w = j->cam->width, h = j->cam->height;
    for (int y = j->tid; y < h; y += j->nthreads) {
        for (int x = 0; x < w; x++) {
            ray_t ray = cast_ray(j->cam, x, y, w, h);
            cx.c->primary_rays++;
            isect_t is = intersect(&cx, &ray);
            v3 c = pixel_color(&cx, &ray, &is);
            if (c.x < 0) c.x = 0; if (c.x > 255) c.x = 255;           /* utils.cpp:125-132 */
            if (c.y < 0) c.y = 0; if (c.y > 255) c.y = 255;
            if (c.z < 0) c.z = 0; if (c.z > 255) c.z = 255;
            size_t p = (size_t)y * w + x;
            if (j->rgb) {
                j->rgb[p * 3 + 0] = (unsigned char)round(c.x);       /* raytracer.cpp:78-80 */
                j->rgb[p * 3 + 1] = (unsigned char)round(c.y);
                j->rgb[p * 3 + 2] = (unsigned char)round(c.z);
            }
            if (j->hits) {
                rtx_hit_id* hr = &j->hits[p];
                hr->kind = is.hit ? is.kind : 0;
                hr->container = is.hit ? is.container : -1;
                hr->face = (is.hit && (is.kind == RTX_KIND_MESH || is.kind == RTX_KIND_TRIANGLE))
                               ? is.face : -1;
                hr->src = is.hit ? (is.kind == RTX_KIND_MESH ? is.src : is.container) : -1;
                hr->distance = is.hit ? is.distance : 0.0;
            }
        }
    }
    return NULL;
}

/* Render one camera on `nthreads` host threads (rows interleaved).  rgb / hits / counters
 * may be NULL.  Returns elapsed milliseconds, or a negative value on bad arguments. */
static double render(const rtx_scene_desc* s, const rtx_camera* cam, uint8_t* rgb,
                     rtx_hit_id* hits, rt_oracle_counters* counters, int nthreads, int consistent) {
    if (!s || !cam || cam->width <= 0 || cam->height <= 0) return -1.0;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    job_t* jobs = (job_t*)calloc((size_t)nthreads, sizeof(job_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
        jobs[i].s = s; jobs[i].cam = cam; jobs[i].rgb = rgb; jobs[i].hits = hits;
        jobs[i].tid = i; jobs[i].nthreads = nthreads; jobs[i].consistent = consistent;
        if (nthreads > 1) pthread_create(&th[i], NULL, worker, &jobs[i]);
    }
    if (nthreads == 1) worker(&jobs[0]);
    else for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (counters) {
        memset(counters, 0, sizeof *counters);
        for (int i = 0; i < nthreads; i++) {
            const uint64_t* a = (const uint64_t*)&jobs[i].c;
            uint64_t* b = (uint64_t*)counters;
            for (size_t k = 0; k < sizeof(rt_oracle_counters) / sizeof(uint64_t); k++) b[k] += a[k];
        }
    }
    free(jobs); free(th);
    return (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
}

double rt_oracle_render(const rtx_scene_desc* s, const rtx_camera* cam, uint8_t* rgb,
                        rtx_hit_id* hits, rt_oracle_counters* counters, int nthreads) {
    return render(s, cam, rgb, hits, counters, nthreads, 0);
}

/* The same for the device library's consistent mode (see ctx_t); not reference behaviour. */
double rt_oracle_render_consistent(const rtx_scene_desc* s, const rtx_camera* cam, uint8_t* rgb,
                                   rtx_hit_id* hits, rt_oracle_counters* counters, int nthreads) {
    return render(s, cam, rgb, hits, counters, nthreads, 1);
}

int rt_oracle_abi_version(void) { return RTX_ABI_VERSION; }
