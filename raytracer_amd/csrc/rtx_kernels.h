// HIP kernels of the ray-casting core (gfx950, wave64).
//
// A frame is processed in *generations*: generation 0 are the camera rays (generated
// in-kernel, utils.cpp:134-158), generation g+1 holds every ray spawned while shading
// generation g (shadow rays, mirror/conductor reflections, dielectric reflection +
// refraction).  The reference's recursion (utils.cpp:868-1061) is linear in every child
// colour, so each queued ray carries an RGB weight and adds weight x (its own shaded terms)
// to its pixel; Beer-Lambert absorption depends on the child's hit distance and is applied
// when the child resolves.  Per generation two kernels run:
//
//   k_trace  closest-hit query for every queued ray (the dominant kernel).  Shadow rays
//            finish here (their carried contribution is added iff unoccluded); radiance
//            rays leave a hit record in HBM.
//   k_shade  shades the radiance hits and appends the next generation's rays.
//
// Work distribution is persistent-thread style: every wave repeatedly claims a batch of
// rays with one atomic add (lane 0) + readfirstlane broadcast; children are appended with a
// ballot-compacted reservation (one atomic per wave per child slot), so queues stay dense.
//
// Traversal is wave-cooperative.  Each lane walks its own ray through the BVH with its own
// stack, but the 128-byte node (or 80-byte triangle) records the lanes need are fetched by
// the whole wave: eight lanes read one record as eight adjacent 16-byte pieces, so one
// load instruction touches 8 records x 128 contiguous bytes instead of 64 scattered
// 16-byte pieces, and the record is handed to its owner lane through a per-wave LDS row.
// Lanes alternate between an inner-node phase and a leaf phase by majority vote
// (while-while traversal), which keeps each lane's own visit order untouched.
//
// All geometry arithmetic is FP64 in the reference's operation order and this file must be
// compiled with -ffp-contract=off: primary-hit identity has to match the reference bit for
// bit, and every secondary ray's origin/direction is derived from those same values.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <limits.h>
#include <math.h>
#include "rtx_amd.h"
#include "rtx_device_types.h"

namespace rtxdev {

#define RTX_DEV __device__ __forceinline__

struct V3 { double x, y, z; };
RTX_DEV V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RTX_DEV V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }
RTX_DEV V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RTX_DEV V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RTX_DEV V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RTX_DEV V3 vscale(V3 a, double b) { return mk(a.x * b, a.y * b, a.z * b); }
RTX_DEV V3 sscale(double a, V3 b) { return mk(a * b.x, a * b.y, a * b.z); }
RTX_DEV V3 vneg(V3 a) { return mk(-a.x, -a.y, -a.z); }
RTX_DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RTX_DEV V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// overloads.cpp:18-21: three divisions by the length (not a multiply by the reciprocal)
RTX_DEV V3 normalize(V3 a) {
    double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk(a.x / len, a.y / len, a.z / len);
}
RTX_DEV double smax(double a, double b) { return (a < b) ? b : a; }   // std::max
RTX_DEV double smin(double a, double b) { return (b < a) ? b : a; }   // std::min

// utils.cpp:96-106
RTX_DEV V3 xf_point(const double* m, V3 p) {
    double x = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    double y = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    double z = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    double w = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (w > (double)1.00001f || w < (double)0.99999f) {
        double iw = 1.0 / w;
        return mk(x * iw, y * iw, z * iw);
    }
    return mk(x, y, z);
}
// utils.cpp:108-123 (direction and normal use the same 3x3 product)
RTX_DEV V3 xf_dir(const double* m, V3 d) {
    return mk(m[0] * d.x + m[1] * d.y + m[2] * d.z, m[4] * d.x + m[5] * d.y + m[6] * d.z,
              m[8] * d.x + m[9] * d.y + m[10] * d.z);
}
// "fast normalize" of utils.cpp:485-489, 581-585: reciprocal of a float-rounded sqrt
RTX_DEV V3 normalize_f32(V3 a) {
    double lenSq = a.x * a.x + a.y * a.y + a.z * a.z;
    double inv = 1.0 / (double)sqrtf((float)lenSq);
    return mk(a.x * inv, a.y * inv, a.z * inv);
}

// Intersection (scene.h:248-264) plus the producing object (src) and the device triangle id.
struct Rec {
    double dist, beta, gamma;
    V3 point, gn, sn;
    int hit, kind, container, face, tri, mat, src;
};

// Per-ray slab constants: 1/d per axis (bvh.cpp:43 computes it per node; the quotient is the
// same value) and the "parallel" predicate |d| < 1e-9f (bvh.cpp:36).
struct Slab { double ox, oy, oz, ix, iy, iz; int par; };

RTX_DEV Slab make_slab(V3 o, V3 d) {
    Slab s;
    s.ox = o.x; s.oy = o.y; s.oz = o.z;
    const double eps = (double)1e-9f;
    s.par = (fabs(d.x) < eps ? 1 : 0) | (fabs(d.y) < eps ? 2 : 0) | (fabs(d.z) < eps ? 4 : 0);
    s.ix = 1.0 / d.x; s.iy = 1.0 / d.y; s.iz = 1.0 / d.z;
    return s;
}

// bvh.cpp:23-56 split into its bound-independent part.  Returns false when the box is
// missed for every bound; otherwise `enter` = t0 after all axes and the box is hit for a
// bound t_min exactly when t_min >= enter (t1 = min(t_min, exits) >= t0).
RTX_DEV bool box_enter(const double* mn, const double* mx, const Slab& s, double& enter) {
    double t0 = 0.0, t1 = INFINITY;
    if (s.par) {
        if ((s.par & 1) && (s.ox < mn[0] || s.ox > mx[0])) return false;
        if ((s.par & 2) && (s.oy < mn[1] || s.oy > mx[1])) return false;
        if ((s.par & 4) && (s.oz < mn[2] || s.oz > mx[2])) return false;
    }
    if (!(s.par & 1)) {
        double a = (mn[0] - s.ox) * s.ix, b = (mx[0] - s.ox) * s.ix;
        if (a > b) { double t = a; a = b; b = t; }
        t0 = smax(t0, a); t1 = smin(t1, b);
    }
    if (!(s.par & 2)) {
        double a = (mn[1] - s.oy) * s.iy, b = (mx[1] - s.oy) * s.iy;
        if (a > b) { double t = a; a = b; b = t; }
        t0 = smax(t0, a); t1 = smin(t1, b);
    }
    if (!(s.par & 4)) {
        double a = (mn[2] - s.oz) * s.iz, b = (mx[2] - s.oz) * s.iz;
        if (a > b) { double t = a; a = b; b = t; }
        t0 = smax(t0, a); t1 = smin(t1, b);
    }
    enter = t0;
    return t1 >= t0;
}

// utils.cpp:327-370 on a pre-gathered triangle (a, e1 = b-a, e2 = c-a).  Accept iff
// t < t_min (strict).  Back-face culling (utils.cpp:322-326) is applied by the caller.
RTX_DEV bool tri_test(const double* T, V3 o, V3 d, double t_min, double eps, double& t_out,
                      double& beta_out, double& gamma_out) {
    const double ax = T[0], ay = T[1], az = T[2];
    const double e1x = T[3], e1y = T[4], e1z = T[5];
    const double e2x = T[6], e2y = T[7], e2z = T[8];
    const double rx = o.x - ax, ry = o.y - ay, rz = o.z - az;
    const double dx = d.x, dy = d.y, dz = d.z;
    double det = -(e1x * (e2y * dz - e2z * dy) - e1y * (e2x * dz - e2z * dx) +
                   e1z * (e2x * dy - e2y * dx));
    if (fabs(det) < 1e-15) return false;
    const double invDet = 1.0 / det;
    double detB = -(rx * (e2y * dz - e2z * dy) - ry * (e2x * dz - e2z * dx) +
                    rz * (e2x * dy - e2y * dx));
    double beta = detB * invDet;
    if (beta < 0.0 || beta > 1.0) return false;
    double detG = -(e1x * (ry * dz - rz * dy) - e1y * (rx * dz - rz * dx) +
                    e1z * (rx * dy - ry * dx));
    double gamma = detG * invDet;
    if (gamma < 0.0 || gamma > 1.0 || beta + gamma > 1.0) return false;
    double detT = e1x * (e2y * rz - e2z * ry) - e1y * (e2x * rz - e2z * rx) +
                  e1z * (e2x * ry - e2y * rx);
    double t = detT * invDet;
    if (t < eps) return false;
    if (t < t_min) { t_out = t; beta_out = beta; gamma_out = gamma; return true; }
    return false;
}

struct Tally { unsigned node_visits, tri_tests, world_box, obj_trav, tlas_visits;
               unsigned inner_steps, inner_lanes, leaf_steps, leaf_lanes; };   // wave-step occupancy (lane 0 only)

#define RTX_STACK 64
#define RTX_ROW_DOUBLES 18          // 144-byte LDS row per lane: 128-byte record + 16 bytes of
                                    // padding so that 16 lanes' ds_read_b128 hit distinct banks

RTX_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Per-wave context: this wave's 64 LDS rows and the lane number.
struct Wave { double* rows; unsigned lane; };

// Wave-cooperative gather of one fixed-size record per requesting lane into that lane's LDS
// row.  `mask` = ballot of the requesting lanes, `index` = record number (valid in requesting
// lanes), records start at base + index*STRIDE and are NCHUNK*16 bytes long.  Eight lanes
// serve one owner (owner = 8g + lane/8, piece = lane%8 in round g), so each load instruction
// reads up to 8 whole records.  Must be executed by every lane of the wave.
template <int NCHUNK, int STRIDE>
RTX_DEV void coop_fetch(const Wave& wv, unsigned long long mask, int index, const char* __restrict__ base) {
    const unsigned sub = wv.lane >> 3, piece = wv.lane & 7u;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        const unsigned gm = (unsigned)(mask >> (8 * g)) & 0xFFu;        // wave-uniform
        if (gm) {
            const int owner = 8 * g + (int)sub;
            const int oidx = __shfl(index, owner);
            if (((gm >> sub) & 1u) && (int)piece < NCHUNK) {
                const float4 v = *(const float4*)(base + (size_t)oidx * STRIDE + piece * 16u);
                *(float4*)((char*)wv.rows + owner * (RTX_ROW_DOUBLES * 8) + piece * 16u) = v;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// BLAS traversal with the visit order and the visited set of bvh.cpp:104-185: children are
// ordered near/far by the sign of dir[splitAxis] (near popped first), a node is entered iff
// its box test passes with the bound current at the moment the reference would pop it.
// Both children's boxes are evaluated when their parent is visited; the far child is pushed
// together with its entry distance and re-checked against the (possibly smaller) bound when
// popped, which is the same predicate the reference evaluates at that moment.
//
// Called by the whole wave; `go` marks the lanes that actually traverse (their own mesh,
// ray and bound).  Returns per lane whether a closer triangle was found.
template <bool COUNT>
RTX_DEV bool blas_traverse(const Scene& S, const Wave& wv, bool go, int mesh_index, V3 o, V3 d, double& t_min,
                           int& best_tri, double& best_beta, double& best_gamma, Tally& tl) {
    const Slab sl = make_slab(o, d);
    bool alive = false;
    int cur_ref = 0, cur_cnt = 0;
    if (go) {
        const Mesh& m = S.meshes[mesh_index];
        double enter;
        alive = box_enter(m.rmin, m.rmax, sl, enter) && (t_min >= enter);
        cur_ref = m.root_ref; cur_cnt = m.root_count;
    }
    int stk_ref[RTX_STACK], stk_cnt[RTX_STACK];
    double stk_t[RTX_STACK];
    int sp = 0;
    bool found = false;
    const double* row = wv.rows + wv.lane * RTX_ROW_DOUBLES;

    while (true) {
        const bool at_inner = alive && cur_cnt == 0;
        const bool at_leaf = alive && cur_cnt > 0;
        const unsigned long long m_inner = __ballot(at_inner);
        const unsigned long long m_leaf = __ballot(at_leaf);
        if ((m_inner | m_leaf) == 0ull) break;
        bool need_pop = false;
        // Phase choice by majority: a leaf step costs several inner steps, so leaves wait until
        // they clearly outnumber the lanes still descending.
        if (m_inner != 0ull && 2 * __popcll(m_inner) >= __popcll(m_leaf)) {
            if (COUNT && wv.lane == 0) { tl.inner_steps++; tl.inner_lanes += (unsigned)__popcll(m_inner); }
            coop_fetch<8, 128>(wv, m_inner, cur_ref, (const char*)S.nodes);
            if (at_inner) {
                if (COUNT) tl.node_visits++;
                const int* ri = (const int*)(row + 12);
                double el, er;
                const bool okl = box_enter(row + 0, row + 3, sl, el) && (t_min >= el);
                const bool okr = box_enter(row + 6, row + 9, sl, er) && (t_min >= er);
                const int lref = ri[0], lcnt = ri[1], rref = ri[2], rcnt = ri[3], axis = ri[4];
                const double da = axis == 0 ? d.x : (axis == 1 ? d.y : d.z);
                const bool left_near = da >= 0.0;
                const int nref = left_near ? lref : rref, ncnt = left_near ? lcnt : rcnt;
                const int fref = left_near ? rref : lref, fcnt = left_near ? rcnt : lcnt;
                const bool okn = left_near ? okl : okr, okf = left_near ? okr : okl;
                const double ef = left_near ? er : el;
                if (okn) {
                    if (okf && sp < RTX_STACK) { stk_ref[sp] = fref; stk_cnt[sp] = fcnt; stk_t[sp] = ef; sp++; }
                    cur_ref = nref; cur_cnt = ncnt;
                } else if (okf) {
                    cur_ref = fref; cur_cnt = fcnt;
                } else {
                    need_pop = true;
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // leaf phase: triangle k of every leaf lane, in leaf order
            for (int k = 0;; k++) {
                const bool want = at_leaf && k < cur_cnt;
                const unsigned long long mk_ = __ballot(want);
                if (mk_ == 0ull) break;
                if (COUNT && wv.lane == 0) { tl.leaf_steps++; tl.leaf_lanes += (unsigned)__popcll(mk_); }
                coop_fetch<5, 80>(wv, mk_, cur_ref + k, (const char*)S.tris);
                if (want) {
                    const int ti = cur_ref + k;
                    if (COUNT) tl.tri_tests++;
                    bool culled = false;
                    if (S.cull) {
                        const double* g = S.tri_gn + 3 * (size_t)ti;
                        culled = d.x * g[0] + d.y * g[1] + d.z * g[2] > 0.0;
                    }
                    double t, b, g2;
                    if (!culled && tri_test(row, o, d, t_min, S.isect_eps, t, b, g2)) {
                        t_min = t; best_tri = ti; best_beta = b; best_gamma = g2; found = true;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            need_pop = at_leaf;
        }
        if (need_pop) {
            alive = false;
            while (sp > 0) {
                --sp;
                if (t_min >= stk_t[sp]) { cur_ref = stk_ref[sp]; cur_cnt = stk_cnt[sp]; alive = true; break; }
            }
        }
    }
    return found;
}

// Fields a successful triangle test writes (utils.cpp:370-381) + kind (bvh.cpp:162).
RTX_DEV void write_tri_hit(const Scene& S, Rec& r, int tri, double t, double beta, double gamma,
                           int material, int container) {
    const Tri& T = S.tris[tri];
    V3 a = ld3(T.a), e1 = ld3(T.e1), e2 = ld3(T.e2);
    r.hit = 1;
    r.dist = t;
    r.point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
    r.gn = ld3(S.tri_gn + 3 * (size_t)tri);
    r.sn = r.gn;
    r.beta = beta; r.gamma = gamma;
    r.mat = material;
    r.container = container;
    r.face = T.face;
    r.tri = tri;
    r.kind = RTX_KIND_MESH;
}

// rayHitsMesh (utils.cpp:396-669) for mesh `mi` under transform `xfi` (-1: none).  Called by
// the whole wave; lanes with act == false only help with the cooperative fetches.
template <bool COUNT>
RTX_DEV bool mesh_hit(const Scene& S, const Wave& wv, bool act, V3 o, V3 d, int mi, int xfi, int material,
                      double& t_min, Rec& r, Tally& tl) {
    bool go = act;
    if (go && S.meshes[mi].n_faces <= 0) go = false;
    V3 ro = o, rd = d;
    const double original_t_min = t_min;
    if (go && xfi >= 0) {
        const Xform& xf = S.xforms[xfi];
        if (xf.has_bounds) {
            if (COUNT) tl.world_box++;
            const Slab sl = make_slab(o, d);
            double enter;
            if (!box_enter(xf.bmin, xf.bmax, sl, enter) || !(t_min >= enter)) go = false;
        }
        if (go) {
            if (COUNT) tl.obj_trav++;
            ro = xf_point(xf.inv, o);
            rd = normalize_f32(xf_dir(xf.inv, d));
        }
    }
    int tri = -1;
    double beta = 0, gamma = 0;
    const bool found = blas_traverse<COUNT>(S, wv, go, mi, ro, rd, t_min, tri, beta, gamma, tl);
    if (!(go && found)) return false;
    write_tri_hit(S, r, tri, t_min, beta, gamma, material, mi);
    if (xfi < 0) return true;

    const Xform& xf = S.xforms[xfi];
    const Mesh& m = S.meshes[mi];
    V3 wp = xf_point(xf.m, r.point);
    double dx = wp.x - o.x, dy = wp.y - o.y, dz = wp.z - o.z;
    double wd = (double)sqrtf((float)(dx * dx + dy * dy + dz * dz));
    if (wd >= original_t_min) {           // utils.cpp:567-570: bound restored, record is not
        t_min = original_t_min;
        return false;
    }
    V3 wg = normalize_f32(xf_dir(xf.nrm, r.gn));
    V3 ws;
    if (m.smooth && m.has_vn) {           // utils.cpp:589-621
        double u = r.beta, v = r.gamma, w = 1.0 - u - v;
        const double* vn = S.tri_vn + 9 * (size_t)tri;
        double nx = w * vn[0] + u * vn[3] + v * vn[6];
        double ny = w * vn[1] + u * vn[4] + v * vn[7];
        double nz = w * vn[2] + u * vn[5] + v * vn[8];
        double nlen = (double)sqrtf((float)(nx * nx + ny * ny + nz * nz));
        double ninv = 1.0 / nlen;
        nx *= ninv; ny *= ninv; nz *= ninv;
        ws = xf_dir(xf.nrm, mk(nx, ny, nz));
    } else {
        ws = xf_dir(xf.nrm, r.sn);
    }
    {
        double slen = (double)sqrtf((float)(ws.x * ws.x + ws.y * ws.y + ws.z * ws.z));
        double sinv = 1.0 / slen;
        ws.x *= sinv; ws.y *= sinv; ws.z *= sinv;
    }
    if (dot(wg, d) > 0.0) { wg = vneg(wg); ws = vneg(ws); }
    r.point = wp; r.gn = wg; r.sn = ws; r.dist = wd;
    t_min = wd;
    r.kind = RTX_KIND_MESH;
    r.container = -2;
    return true;
}

// utils.cpp:160-219
RTX_DEV bool plane_hit(const Scene& S, V3 ro, V3 rd, int index, double& t_min, Rec& r) {
    const Plane& pl = S.planes[index];
    V3 o = ro, d = rd;
    V3 n = ld3(pl.n), a = ld3(pl.a);
    const bool has_xf = pl.xform >= 0;
    if (has_xf) {
        const Xform& xf = S.xforms[pl.xform];
        o = xf_point(xf.inv, ro);
        d = normalize(xf_dir(xf.inv, rd));
    }
    double denom = dot(d, n);
    if (fabs(denom) < 1e-9) return false;
    double t = dot(vsub(a, o), n) / denom;
    if (t > 0 && t < t_min) {
        V3 op = vadd(o, vscale(d, t));
        if (has_xf) {
            const Xform& xf = S.xforms[pl.xform];
            V3 wp = xf_point(xf.m, op);
            V3 wn = normalize(xf_dir(xf.nrm, n));
            V3 dv = vsub(wp, ro);
            double wd = sqrt(dot(dv, dv));
            if (wd < t_min) {
                t_min = wd;
                r.hit = 1; r.dist = wd; r.point = wp; r.gn = wn; r.sn = wn; r.mat = pl.material;
                r.kind = RTX_KIND_PLANE; r.container = index; r.src = index;
                return true;
            }
        } else {
            t_min = t;
            r.hit = 1; r.dist = t; r.point = op; r.gn = n; r.sn = n; r.mat = pl.material;
            r.kind = RTX_KIND_PLANE; r.container = index; r.src = index;
            return true;
        }
    }
    return false;
}

// utils.cpp:221-295
RTX_DEV bool sphere_hit(const Scene& S, V3 ro, V3 rd, int index, double& t_min, Rec& r) {
    const Sphere& sp = S.spheres[index];
    V3 o = ro, d = rd;
    const bool has_xf = sp.xform >= 0;
    if (has_xf) {
        const Xform& xf = S.xforms[sp.xform];
        o = xf_point(xf.inv, ro);
        d = normalize_f32(xf_dir(xf.inv, rd));
    }
    V3 c = ld3(sp.c);
    double rad = sp.r;
    V3 oc = vsub(o, c);
    double D = dot(d, oc) * dot(d, oc) - (dot(d, d) * (dot(oc, oc) - rad * rad));
    if (D < 0) return false;
    double t1 = (-dot(d, oc) + sqrt(D)) / dot(d, d);
    double t2 = (-dot(d, oc) - sqrt(D)) / dot(d, d);
    double t = smin(t1, t2);
    if (t > 0 && t < t_min) {
        V3 op = vadd(o, vscale(d, t));
        V3 on = normalize(vsub(op, c));
        if (has_xf) {
            const Xform& xf = S.xforms[sp.xform];
            V3 wp = xf_point(xf.m, op);
            V3 wn = xf_dir(xf.nrm, on);
            double nlen = (double)sqrtf((float)(wn.x * wn.x + wn.y * wn.y + wn.z * wn.z));
            wn.x /= nlen; wn.y /= nlen; wn.z /= nlen;
            double dx = wp.x - ro.x, dy = wp.y - ro.y, dz = wp.z - ro.z;
            double wd = (double)sqrtf((float)(dx * dx + dy * dy + dz * dz));
            if (wd < t_min) {
                t_min = wd;
                r.hit = 1; r.dist = wd; r.point = wp; r.gn = wn; r.sn = wn; r.mat = sp.material;
                r.kind = RTX_KIND_SPHERE; r.container = index; r.src = index;
                return true;
            }
        } else {
            t_min = t;
            r.hit = 1; r.dist = t; r.point = op; r.gn = on; r.sn = on; r.mat = sp.material;
            r.kind = RTX_KIND_SPHERE; r.container = index; r.src = index;
            return true;
        }
    }
    return false;
}

// Free triangles, utils.cpp:692-729.
template <bool COUNT>
RTX_DEV void free_tri_hit(const Scene& S, V3 ro, V3 rd, int index, double& t_min, Rec& r, bool& hit, Tally& tl) {
    const FreeTri& ft = S.ftris[index];
    if (COUNT) tl.tri_tests++;
    V3 o = ro, d = rd;
    const bool has_xf = ft.xform >= 0;
    if (has_xf) {
        const Xform& xf = S.xforms[ft.xform];
        o = xf_point(xf.inv, ro);
        d = normalize(xf_dir(xf.inv, rd));
    }
    V3 gn = ld3(ft.gn);
    if (S.cull && dot(d, gn) > 0.0) return;
    double t, beta, gamma;
    const double original_t_min = t_min;
    if (!tri_test(ft.a, o, d, t_min, S.isect_eps, t, beta, gamma)) return;
    V3 a = ld3(ft.a), e1 = ld3(ft.e1), e2 = ld3(ft.e2);
    t_min = t;
    r.hit = 1; r.dist = t;
    r.point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
    r.gn = gn; r.sn = gn; r.beta = beta; r.gamma = gamma; r.mat = ft.material;
    r.container = -1; r.face = index; r.tri = -1;
    if (!has_xf) { hit = true; r.kind = RTX_KIND_TRIANGLE; r.src = -1; return; }
    const Xform& xf = S.xforms[ft.xform];
    V3 wp = xf_point(xf.m, r.point);
    V3 wg = normalize(xf_dir(xf.nrm, r.gn));
    V3 ws = normalize(xf_dir(xf.nrm, r.sn));
    V3 dv = vsub(wp, ro);
    double wd = sqrt(dot(dv, dv));
    if (wd < original_t_min) {
        r.point = wp; r.gn = wg; r.sn = ws; r.dist = wd;
        t_min = wd; hit = true; r.kind = RTX_KIND_TRIANGLE; r.src = -1;
    } else {
        t_min = original_t_min;          // record keeps the object-space values (utils.cpp:719-721)
    }
}

#define RTX_CAND 8

// Collect, for one ray, the RTX_CAND smallest instance indices > last whose world box the ray
// meets within the bound (TLAS walk; per-lane, no cross-lane work).
template <bool COUNT>
RTX_DEV void tlas_collect(const Scene& S, const Slab& sl, double t_min, int last, int (&buf)[RTX_CAND], Tally& tl) {
    double enter;
    if (!(S.tl_max_idx > last && box_enter(S.tl_min, S.tl_max, sl, enter) && (t_min >= enter))) return;
    int stk_ref[RTX_STACK], stk_cnt[RTX_STACK];
    int sp = 1;
    stk_ref[0] = S.tl_ref; stk_cnt[0] = S.tl_count;
    while (sp > 0) {
        --sp;
        const int ref = stk_ref[sp], cnt = stk_cnt[sp];
        if (cnt > 0) {
            for (int k = 0; k < cnt; k++) {
                const int idx = S.tlas_order[ref + k];
                if (idx > last && idx < buf[RTX_CAND - 1]) {
                    int x = idx;                 // sorted insert, largest falls off
#pragma unroll
                    for (int q = 0; q < RTX_CAND; q++) {
                        const int lo = min(buf[q], x), hi = max(buf[q], x);
                        buf[q] = lo; x = hi;
                    }
                }
            }
        } else {
            const TlasNode& n = S.tlas[ref];
            if (COUNT) tl.tlas_visits++;
            double e;
            if (n.lmax_idx > last && n.lmin_idx < buf[RTX_CAND - 1] && box_enter(n.lmin, n.lmax, sl, e) &&
                (t_min >= e) && sp < RTX_STACK) {
                stk_ref[sp] = n.lref; stk_cnt[sp] = n.lcount; sp++;
            }
            if (n.rmax_idx > last && n.rmin_idx < buf[RTX_CAND - 1] && box_enter(n.rmin, n.rmax, sl, e) &&
                (t_min >= e) && sp < RTX_STACK) {
                stk_ref[sp] = n.rref; stk_cnt[sp] = n.rcount; sp++;
            }
        }
    }
}

// Closest-hit query with the reference's object order and state machine (utils.cpp:671-865).
// Executed by the whole wave; `active` marks the lanes that carry a ray.
template <bool COUNT>
RTX_DEV void intersect(const Scene& S, const Wave& wv, bool active, V3 o, V3 d, Rec& r, Tally& tl) {
    double t_min = DBL_MAX;
    r.hit = 0; r.dist = 0.0; r.beta = 0.0; r.gamma = 0.0;
    r.point = mk(0, 0, 0); r.gn = mk(0, 0, 0); r.sn = mk(0, 0, 0);
    r.kind = RTX_KIND_NONE; r.container = -1; r.face = -1; r.tri = -1; r.mat = -1; r.src = -1;
    bool hit = false;

    if (active) {
        for (int i = 0; i < S.n_planes; i++) hit = plane_hit(S, o, d, i, t_min, r) || hit;
        for (int i = 0; i < S.n_ftris; i++) free_tri_hit<COUNT>(S, o, d, i, t_min, r, hit, tl);
    }
    for (int i = 0; i < S.n_meshes; i++) {                      // wave-uniform loop
        const Mesh& m = S.meshes[i];
        if (mesh_hit<COUNT>(S, wv, active, o, d, i, m.xform, m.material, t_min, r, tl)) { hit = true; r.src = i; }
    }
    if (active)
        for (int i = 0; i < S.n_spheres; i++) hit = sphere_hit(S, o, d, i, t_min, r) || hit;

    // Instances (utils.cpp:756-790).  The TLAS only selects candidates -- instances whose
    // world box the ray meets within the current bound; a box missed now stays missed
    // because the bound never grows.  Candidates are then replayed in ascending index,
    // RTX_CAND at a time, through the reference's own per-instance logic.
    if (S.n_insts > 0) {
        const Slab sl = make_slab(o, d);
        int last = -1;
        bool more = active;
        while (__ballot(more) != 0ull) {
            int buf[RTX_CAND];
#pragma unroll
            for (int k = 0; k < RTX_CAND; k++) buf[k] = INT_MAX;
            if (more) tlas_collect<COUNT>(S, sl, t_min, last, buf, tl);
            const bool full = buf[RTX_CAND - 1] != INT_MAX;
            bool have = more && buf[0] != INT_MAX;
            while (__ballot(have) != 0ull) {
                const int idx = buf[0];
#pragma unroll
                for (int q = 0; q + 1 < RTX_CAND; q++) buf[q] = buf[q + 1];
                buf[RTX_CAND - 1] = INT_MAX;
                int base_mesh = 0, xform = -1, material = -1;
                if (have) {
                    last = idx;
                    const Inst& in = S.insts[idx];
                    base_mesh = in.base_mesh; xform = in.xform; material = in.material;
                }
                if (mesh_hit<COUNT>(S, wv, have, o, d, base_mesh, xform, material, t_min, r, tl)) {
                    hit = true; r.src = S.n_meshes + idx;
                }
                have = have && buf[0] != INT_MAX;
            }
            more = more && full;
        }
    }

    if (!(active && hit)) { r.hit = 0; return; }
    // utils.cpp:807-853.  r.hit can be left set by a rejected transformed hit only when an
    // earlier object already hit, so `hit` and r.hit agree here.
    const double u = r.beta, v = r.gamma, w = 1.0 - u - v;
    r.gn = normalize(r.gn);
    r.sn = normalize(r.sn);
    if (r.kind == RTX_KIND_MESH && r.container >= 0 && r.container < S.n_meshes) {
        const Mesh& m = S.meshes[r.container];
        if (m.smooth && m.has_vn && r.tri >= 0) {
            const double* vn = S.tri_vn + 9 * (size_t)r.tri;
            V3 n0 = ld3(vn), n1 = ld3(vn + 3), n2 = ld3(vn + 6);
            V3 sm = normalize(vadd(vadd(sscale(w, n0), sscale(u, n1)), sscale(v, n2)));
            if (m.xform >= 0) sm = normalize(xf_dir(S.xforms[m.xform].nrm, sm));
            r.sn = sm;
            if (dot(r.gn, d) > 0.0) { r.gn = vneg(r.gn); r.sn = vneg(r.sn); }
            else if (dot(r.sn, r.gn) < 0.0) r.sn = vneg(r.sn);
        }
    }
}

// utils.cpp:1089-1111
RTX_DEV double fresnel_conductor(double cosTheta, double n, double k) {
    cosTheta = smax(0.0, smin(1.0, cosTheta));
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
// utils.cpp:1113-1129
RTX_DEV double fresnel_dielectric(double cosTheta, double n1, double n2) {
    cosTheta = smax(-1.0, smin(1.0, cosTheta));
    double eta = n1 / n2;
    double s2 = eta * eta * (1.0 - cosTheta * cosTheta);
    if (s2 >= 1.0) return 1.0;
    double cosPhi = sqrt(smax(0.0, 1.0 - s2));
    double rPerp = (n1 * cosTheta - n2 * cosPhi) / (n1 * cosTheta + n2 * cosPhi);
    double rPara = (n2 * cosTheta - n1 * cosPhi) / (n2 * cosTheta + n1 * cosPhi);
    return 0.5 * (rPerp * rPerp + rPara * rPara);
}

#define RTX_META(depth, beer) ((unsigned)(depth) | ((unsigned)((beer) + 1) << 8))

// Append one ray per requesting lane: ballot, one reservation per wave, dense stores.
// Radiance rays go to slot k, shadow rays to slot capacity-1-k.  Must be reached by every
// lane of the wave.
RTX_DEV void emit(const Queue& q, GenCtrl* next, bool shadow, bool want, V3 o, V3 d, V3 w, double aux,
                  unsigned pixel, unsigned meta) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return;
    const unsigned lane = lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    const unsigned n = (unsigned)__popcll(mask);
    unsigned long long old = 0;
    if ((int)lane == leader) old = atomicAdd(&next->counts, shadow ? ((unsigned long long)n << 32) : (unsigned long long)n);
    old = __shfl(old, leader);
    const unsigned base = shadow ? (unsigned)(old >> 32) : (unsigned)old;
    const unsigned other = shadow ? (unsigned)old : (unsigned)(old >> 32);
    // Reservations at both ends are totally ordered by the single atomic word, so
    // base + n + other <= capacity means these slots are not, and will never be, claimed from
    // the other end.  Otherwise flag the overflow (the frame is redone with larger queues)
    // and drop the rays.
    if ((unsigned long long)base + n + other > q.capacity) { if ((int)lane == leader) next->overflow = 1u; return; }
    if (want) {
        const unsigned k = base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        const unsigned idx = shadow ? q.capacity - 1u - k : k;
        q.ox[idx] = o.x; q.oy[idx] = o.y; q.oz[idx] = o.z;
        q.dx[idx] = d.x; q.dy[idx] = d.y; q.dz[idx] = d.z;
        q.wx[idx] = w.x; q.wy[idx] = w.y; q.wz[idx] = w.z;
        q.aux[idx] = aux; q.pixel[idx] = pixel; q.meta[idx] = meta;
    }
}

RTX_DEV void add_color(double* accum, unsigned pixel, V3 c) {
    if (c.x != 0.0) atomicAdd(accum + 3 * (size_t)pixel, c.x);
    if (c.y != 0.0) atomicAdd(accum + 3 * (size_t)pixel + 1, c.y);
    if (c.z != 0.0) atomicAdd(accum + 3 * (size_t)pixel + 2, c.z);
}

// Generation-0 work item -> pixel.  64 consecutive items form an 8x8 pixel block of one tile,
// so the rays of a wave are coherent.  Returns false for padding items outside the image.
RTX_DEV bool primary_pixel(const Frame& F, unsigned item, int& px, int& py, unsigned& local_pixel) {
    const int bpt_x = (F.tile_w + 7) >> 3, bpt_y = (F.tile_h + 7) >> 3;
    const unsigned blk = item >> 6, l = item & 63u;
    const unsigned bpt = (unsigned)(bpt_x * bpt_y);
    const unsigned k = blk / bpt, b = blk % bpt;
    const int tx = (int)(b % (unsigned)bpt_x) * 8 + (int)(l & 7u);
    const int ty = (int)(b / (unsigned)bpt_x) * 8 + (int)(l >> 3);
    const unsigned tile = (unsigned)F.tile_first + k * (unsigned)F.tile_stride;
    px = (int)(tile % (unsigned)F.tiles_x) * F.tile_w + tx;
    py = (int)(tile / (unsigned)F.tiles_x) * F.tile_h + ty;
    local_pixel = k * (unsigned)(F.tile_w * F.tile_h) + (unsigned)(ty * F.tile_w + tx);
    return tx < F.tile_w && ty < F.tile_h && px < F.width && py < F.height;
}

// utils.cpp:142-156 with the camera basis precomputed per frame.
RTX_DEV V3 primary_direction(const Frame& F, int px, int py) {
    double s_u = (px + 0.5) * F.r_minus_l / F.width;
    double s_v = (py + 0.5) * F.t_minus_b / F.height;
    V3 e = ld3(F.e), u = ld3(F.u), v = ld3(F.v), q = ld3(F.q);
    V3 s = vsub(vadd(q, vscale(u, s_u)), vscale(v, s_v));
    return normalize(vsub(s, e));
}

// Queue slot of trace work item w: radiance rays first, then shadow rays (stored from the back).
RTX_DEV unsigned queue_slot(unsigned w, unsigned n_radiance, unsigned capacity) {
    return w < n_radiance ? w : capacity - 1u - (w - n_radiance);
}

template <bool COUNT>
__global__ void __launch_bounds__(256)
k_trace(const Scene S, const Frame F, const Queue qin, const HitBuf hb, GenCtrl* ctrl, const int gen,
        double* __restrict__ accum, rtx_hit_id* __restrict__ aov, Counters* __restrict__ counters) {
    __shared__ __attribute__((aligned(16))) double lds_rows[4][64 * RTX_ROW_DOUBLES];
    Wave wv;
    wv.lane = lane_id();
    wv.rows = &lds_rows[threadIdx.x >> 6][0];
    const unsigned lane = wv.lane;
    GenCtrl* me = ctrl + gen;
    unsigned n_rad = 0, n_in = 0;
    if (gen == 0) { n_rad = (unsigned)F.n_items0; n_in = n_rad; }
    else {
        // after an overflow the counters may exceed what was stored; clamp (the frame is redone)
        const unsigned long long c = me->counts;
        n_rad = min((unsigned)c, qin.capacity);
        n_in = n_rad + min((unsigned)(c >> 32), qin.capacity - n_rad);
    }
    Tally tl = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned n_rays = 0, n_primary = 0, n_shadow = 0, n_secondary = 0;

    // Rays claimed per wave and fetch: 64 when the generation can fill the whole grid,
    // fewer when it cannot -- a thin generation (deep reflection chains) is latency bound,
    // and spreading its rays over more waves shortens the slowest wave.
    unsigned batch = 64u;
    if (gen != 0) {
        const unsigned waves = gridDim.x * (blockDim.x >> 6);
        batch = (n_in + waves - 1u) / waves;
        batch = batch < 1u ? 1u : (batch > 64u ? 64u : batch);
    }

    while (true) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&me->head_trace, batch);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_in) break;
        const unsigned item = base + lane;
        bool active = lane < batch && item < n_in;

        V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
        unsigned slot = item, pixel = 0;
        int px = 0, py = 0;
        bool is_shadow = false;
        if (gen == 0) {
            active = primary_pixel(F, item, px, py, pixel) && active;
            if (active) { o = ld3(F.e); d = primary_direction(F, px, py); }
        } else if (active) {
            is_shadow = item >= n_rad;
            slot = queue_slot(item, n_rad, qin.capacity);
            o = mk(qin.ox[slot], qin.oy[slot], qin.oz[slot]);
            d = mk(qin.dx[slot], qin.dy[slot], qin.dz[slot]);
        }

        Rec r;
        intersect<COUNT>(S, wv, active, o, d, r, tl);
        if (COUNT && active) { n_rays++; if (gen == 0) n_primary++; else if (is_shadow) n_shadow++; else n_secondary++; }
        if (!active) continue;

        if (gen == 0 && aov != nullptr) {
            rtx_hit_id h;
            h.kind = r.hit ? r.kind : 0;
            h.container = r.hit ? r.container : -1;
            h.face = (r.hit && (r.kind == RTX_KIND_MESH || r.kind == RTX_KIND_TRIANGLE)) ? r.face : -1;
            h.src = r.hit ? (r.kind == RTX_KIND_MESH ? r.src : r.container) : -1;
            h.distance = r.hit ? r.dist : 0.0;
            aov[(size_t)py * F.width + px] = h;
        }
        if (is_shadow) {                                             // utils.cpp:1083
            if (!(r.hit && r.dist < qin.aux[slot]))
                add_color(accum, qin.pixel[slot], mk(qin.wx[slot], qin.wy[slot], qin.wz[slot]));
        } else {
            hb.code[slot] = r.hit ? r.mat + 1 : -1;
            if (r.hit) {
                hb.dist[slot] = r.dist;
                hb.px[slot] = r.point.x; hb.py[slot] = r.point.y; hb.pz[slot] = r.point.z;
                hb.gx[slot] = r.gn.x; hb.gy[slot] = r.gn.y; hb.gz[slot] = r.gn.z;
                hb.sx[slot] = r.sn.x; hb.sy[slot] = r.sn.y; hb.sz[slot] = r.sn.z;
            }
        }
    }

    if (COUNT && counters) {
        unsigned long long v[13] = {n_rays, n_primary, n_shadow, n_secondary, tl.node_visits, tl.tri_tests,
                                    tl.world_box, tl.obj_trav, tl.tlas_visits, tl.inner_steps, tl.inner_lanes,
                                    tl.leaf_steps, tl.leaf_lanes};
#pragma unroll
        for (int k = 0; k < 13; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(((unsigned long long*)counters) + k, x);
        }
    }
}

// Shading of the radiance hits of one generation (utils.cpp:868-1061) and emission of the
// next generation's rays.
__global__ void __launch_bounds__(256)
k_shade(const Scene S, const Frame F, const Queue qin, const Queue qout, const HitBuf hb, GenCtrl* ctrl,
        const int gen, double* __restrict__ accum) {
    const unsigned lane = lane_id();
    GenCtrl* me = ctrl + gen;
    GenCtrl* next = ctrl + gen + 1;
    const unsigned n_in = gen == 0 ? (unsigned)F.n_items0 : min((unsigned)me->counts, qin.capacity);

    while (true) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&me->head_shade, 64u);
        base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= n_in) break;
        const unsigned slot = base + lane;
        bool active = slot < n_in;

        V3 d = mk(0, 0, 1), W = mk(1, 1, 1);
        unsigned pixel = 0, depth = 0;
        int beer = -1;
        if (gen == 0) {
            int px, py;
            active = primary_pixel(F, slot, px, py, pixel) && active;
            if (active) d = primary_direction(F, px, py);
        } else if (active) {
            d = mk(qin.dx[slot], qin.dy[slot], qin.dz[slot]);
            W = mk(qin.wx[slot], qin.wy[slot], qin.wz[slot]);
            pixel = qin.pixel[slot];
            const unsigned meta = qin.meta[slot];
            depth = meta & 0xFFu; beer = (int)(meta >> 8) - 1;
        }

        bool shade = false;
        int mat = -1;
        V3 P = mk(0, 0, 0), gn = mk(0, 0, 1), sn = mk(0, 0, 1);
        if (active) {
            const int code = hb.code[slot];
            if (code < 0) {
                if (depth == 0) add_color(accum, pixel, vmul(W, ld3(S.bg)));      // utils.cpp:1055-1058
            } else {
                if (beer >= 0) {                                                     // utils.cpp:975-984
                    const Material& pm = S.materials[beer];
                    V3 ab = vscale(ld3(pm.sigma), hb.dist[slot]);
                    W = vmul(W, mk(exp(-ab.x), exp(-ab.y), exp(-ab.z)));
                }
                mat = code - 1;
                if (mat >= 0) {
                    shade = true;
                    P = mk(hb.px[slot], hb.py[slot], hb.pz[slot]);
                    gn = mk(hb.gx[slot], hb.gy[slot], hb.gz[slot]);
                    sn = mk(hb.sx[slot], hb.sy[slot], hb.sz[slot]);
                }
            }
        }

        // Children of this hit.  Radiance children: c0 (reflection), c1 (refraction).
        V3 normal = mk(0, 0, 1);
        int mtype = 0;
        bool plain = false;
        V3 lobe_a = mk(0, 0, 0), lobe_s = mk(0, 0, 0);      // diffuse / specular colour of the light loop
        bool c0 = false, c1 = false;
        V3 c0o = P, c0d = d, c0w = W, c1o = P, c1d = d, c1w = W;
        int c1beer = -1;
        double phong = 0.0;
        if (shade) {
            const Material& m = S.materials[mat];
            mtype = m.type;
            phong = m.phong;
            normal = normalize(sn);
            add_color(accum, pixel, vmul(W, vmul(ld3(m.ambient), ld3(S.ambient))));   // utils.cpp:880
            const bool deeper = (int)depth + 1 <= S.max_depth;
            if (mtype == RTX_MAT_CONDUCTOR) {                        // utils.cpp:885-930
                lobe_s = ld3(m.mirror);
                double cosTheta = dot(vneg(d), normal);
                cosTheta = smax(0.0, smin(1.0, cosTheta));
                double Fr = fresnel_conductor(cosTheta, m.n, m.k);
                if (deeper) {
                    V3 n = normalize(normal);                        // reflect(), utils.cpp:1167-1182
                    V3 dir = vsub(d, sscale(2.0 * dot(d, n), n));
                    c0 = true;
                    c0o = vadd(P, sscale(S.shadow_eps, n));
                    c0d = normalize(dir);
                    c0w = vmul(W, sscale(Fr, ld3(m.mirror)));
                }
            } else if (mtype == RTX_MAT_DIELECTRIC) {                // utils.cpp:935-988
                double n1 = 1.0, n2 = m.n;
                const bool entering = dot(d, normal) < 0.0;
                if (!entering) { normal = vneg(normal); double t = n1; n1 = n2; n2 = t; }
                double cosTheta = dot(vneg(d), normal);
                cosTheta = smax(-1.0, smin(1.0, cosTheta));
                double Fr = fresnel_dielectric(cosTheta, n1, n2);
                // refract(), utils.cpp:1134-1165
                double eta = n1 / n2;
                double cosT = -dot(d, normal);
                cosT = smax(-1.0, smin(1.0, cosT));
                double s2 = eta * eta * (1.0 - cosT * cosT);
                const bool tir = s2 > 1.0;
                if (deeper) {
                    V3 n = normalize(normal);
                    V3 dir = vsub(d, sscale(2.0 * dot(d, n), n));
                    c0 = true;
                    c0o = vadd(P, sscale(S.shadow_eps, n));
                    c0d = normalize(dir);
                    c0w = tir ? W : vscale(W, Fr);
                    if (!tir) {
                        double cosPhi = sqrt(smax(0.0, 1.0 - s2));
                        V3 wt = vsub(vscale(vadd(d, vscale(normal, cosT)), eta), vscale(normal, cosPhi));
                        c1 = true;
                        c1o = vsub(P, sscale(S.shadow_eps, normalize(normal)));
                        c1d = normalize(wt);
                        c1w = vscale(W, 1.0 - Fr);
                        c1beer = entering ? mat : -1;
                    }
                }
            } else {                                                 // utils.cpp:995-1038
                plain = true;
                lobe_a = ld3(m.diffuse);
                lobe_s = ld3(m.specular);
                if (m.is_mirror && deeper) {
                    V3 n = normalize(sn);
                    V3 dir = vsub(d, sscale(2.0 * dot(d, n), n));
                    c0 = true;
                    c0o = vadd(P, sscale(S.shadow_eps, n));
                    c0d = normalize(dir);
                    c0w = vmul(W, ld3(m.mirror));
                }
            }
        }

        // Light loop: one shadow ray per light for conductor and plain materials, carrying the
        // unshadowed contribution (utils.cpp:887-913, 1008-1038); isInShadow: utils.cpp:1066-1084.
        const bool lit = shade && mtype != RTX_MAT_DIELECTRIC;
        for (int li = 0; li < S.n_lights; li++) {
            V3 so = P, sd = d, sw = W;
            double dist_to_light = 0.0;
            if (lit) {
                const Light& L = S.lights[li];
                V3 lp = ld3(L.p);
                V3 lv = vsub(lp, P);
                double d2 = dot(lv, lv);
                double dist = sqrt(d2);
                V3 ldir = vscale(lv, 1.0 / dist);
                double atten = 1.0 / d2;
                V3 view = normalize(vneg(d));
                V3 half = normalize(vadd(ldir, view));
                double sf = pow(smax(0.0, dot(normal, half)), phong);
                V3 contrib = vscale(vscale(vmul(lobe_s, ld3(L.i)), sf), atten);
                if (plain) {
                    double df = smax(0.0, dot(normal, ldir));
                    V3 diffuse = vscale(vscale(vmul(lobe_a, ld3(L.i)), df), atten);
                    contrib = vadd(diffuse, contrib);
                }
                so = vadd(P, sscale(S.shadow_eps, gn));
                sd = normalize(vsub(lp, P));
                sw = vmul(W, contrib);
                dist_to_light = sqrt(dot(vsub(lp, P), vsub(lp, P)));
            }
            emit(qout, next, true, lit, so, sd, sw, dist_to_light, pixel, 0u);
        }
        emit(qout, next, false, c0, c0o, c0d, c0w, 0.0, pixel, RTX_META(depth + 1, -1));
        emit(qout, next, false, c1, c1o, c1d, c1w, 0.0, pixel, RTX_META(depth + 1, c1beer));
    }
}

// Final pass: clamp to [0,255] (utils.cpp:125-132), round half away from zero and store as
// u8 (raytracer.cpp:78-80).  For whole frames the local pixel order is the image order.
__global__ void k_resolve(const double* __restrict__ accum, uint8_t* __restrict__ rgb, unsigned n_values) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    double c = accum[i];
    if (c < 0) c = 0;
    if (c > 255) c = 255;
    rgb[i] = (uint8_t)round(c);
}

}  // namespace rtxdev
