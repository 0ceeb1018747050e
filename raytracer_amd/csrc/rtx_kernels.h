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
// stack, but the 256-byte records the lanes need (a four-wide collapsed node, or two 128-byte
// triangle records of a leaf) are fetched by the whole wave: sixteen lanes read one record as
// sixteen adjacent 16-byte pieces, so one load instruction touches 4 records x 256 contiguous
// bytes instead of 64 scattered 16-byte pieces.  The loads are LDS-DMA (global_load_lds_dwordx4):
// the pieces land directly in the owner lane's 256-byte LDS row without passing through
// registers.  Inner-node lanes and leaf lanes advance in the same step (both kinds of record
// come through the same fetch), which keeps each lane's own visit order untouched.  Lanes are
// refilled dynamically: a lane that finishes its ray (or the current object of its ray) picks
// up its next job while the other lanes keep traversing (see "Per-lane ray state" below).
//
// All geometry arithmetic is FP64 in the reference's operation order and this file must be
// compiled with -ffp-contract=off: primary-hit identity has to match the reference bit for
// bit, and every secondary ray's origin/direction is derived from those same values.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <stddef.h>
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
// same value) and, in `par`, bits 0-2 the "parallel" predicate |d| < 1e-9f per axis (bvh.cpp:36) and
// bits 4-6 "the far child comes first at a split along this axis", !(d >= 0) (bvh.cpp:139-147).
struct Slab { double ox, oy, oz, ix, iy, iz; int par; };
#define RTX_PAR_AXES 7

RTX_DEV Slab make_slab(V3 o, V3 d) {
    Slab s;
    s.ox = o.x; s.oy = o.y; s.oz = o.z;
    const double eps = (double)1e-9f;
    s.par = (fabs(d.x) < eps ? 1 : 0) | (fabs(d.y) < eps ? 2 : 0) | (fabs(d.z) < eps ? 4 : 0) |
            (!(d.x >= 0.0) ? 16 : 0) | (!(d.y >= 0.0) ? 32 : 0) | (!(d.z >= 0.0) ? 64 : 0);
    s.ix = 1.0 / d.x; s.iy = 1.0 / d.y; s.iz = 1.0 / d.z;
    return s;
}

// bvh.cpp:23-56 split into its bound-independent part.  Returns false when the box is
// missed for every bound; otherwise `enter` = t0 after all axes and the box is hit for a
// bound t_min exactly when t_min >= enter (t1 = min(t_min, exits) >= t0).
RTX_DEV bool box_enter(const double* mn, const double* mx, const Slab& s, double& enter) {
    double t0 = 0.0, t1 = INFINITY;
    if (s.par & RTX_PAR_AXES) {
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

// The same test for a ray with no "parallel" axis (par == 0), written without branches and
// with hardware min/max.  With par == 0 every 1/d is finite, so no product is a NaN, and for
// non-NaN operands "swap if a > b", std::max and std::min are exactly min/max (a zero's sign
// may differ, which no comparison downstream can see): same values as box_enter.
RTX_DEV bool box_enter_nopar(const double* mn, const double* mx, const Slab& s, double& enter) {
    const double ax = (mn[0] - s.ox) * s.ix, bx = (mx[0] - s.ox) * s.ix;
    const double ay = (mn[1] - s.oy) * s.iy, by = (mx[1] - s.oy) * s.iy;
    const double az = (mn[2] - s.oz) * s.iz, bz = (mx[2] - s.oz) * s.iz;
    const double t0 = __builtin_fmax(__builtin_fmax(__builtin_fmax(0.0, __builtin_fmin(ax, bx)), __builtin_fmin(ay, by)),
                                     __builtin_fmin(az, bz));
    const double t1 = __builtin_fmin(__builtin_fmin(__builtin_fmax(ax, bx), __builtin_fmax(ay, by)), __builtin_fmax(az, bz));
    enter = t0;
    return t1 >= t0;
}

// The box test of the service code (object world boxes, root boxes): nearly every ray has no
// axis-parallel component, and for those the branch-free form gives the same values.
RTX_DEV bool box_enter_svc(const double* mn, const double* mx, const Slab& s, double& enter) {
    if ((s.par & RTX_PAR_AXES) == 0) return box_enter_nopar(mn, mx, s, enter);
    return box_enter(mn, mx, s, enter);
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
               unsigned inner_steps, inner_lanes, leaf_steps, leaf_lanes;       // wave-step occupancy (lane 0 only)
               unsigned rays, primary, shadow, secondary;
               unsigned passes;                                                  // service passes (lane 0)
               unsigned long long clk_trav, clk_serv, clk_total, clk_fetch, clk_pop;
               unsigned long long sec[8];
               unsigned hist[8], hist_shadow[8]; unsigned long long longest_ray; };   // per-lane section profile of service_lane: ticks (0-3), calls (4-7)              // shader-clock ticks (lane 0)

#define RTX_STACK 100                // BLAS traversal stack entries per lane (BVH depth <= 64)
#define RTX_TLAS_STACK 48
// Exact mode's search of the instances (search_service): the last entries of a lane's stack area
// hold what the search has found -- one entry of parked state, then up to RTX_SEARCH_LIST instances
// with a hit -- and the stack proper ends below them.  Lane::age holds the phase (low three bits),
// SP_REPLAY (the ray is being done again the plain way) and, from bit 8, the length of the list.
#ifndef RTX_SEARCH_MIN_INSTS
#define RTX_SEARCH_MIN_INSTS 32      // scenes with fewer instances replay them in index order
#endif
#define RTX_SEARCH_LIST 16
#define RTX_SEARCH_STACK (RTX_STACK - 1 - RTX_SEARCH_LIST)
enum { SP_WALK = 0, SP_VERIFY = 1, SP_PROBE = 2, SP_REPLAY = 8 };
#define RTX_ROW_DOUBLES 32          // one 256-byte LDS row per lane

RTX_DEV unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Address-space-qualified views: LDS rows and the private traversal stack are passed to
// out-of-line code through these so that ds_* and scratch_* instructions are used (a plain
// pointer would be generic and turn every access into a flat_* instruction).
#define RTX_GLOBAL __attribute__((address_space(1)))
typedef __attribute__((address_space(3))) double lds_double;
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2d lds_v2d;
typedef __attribute__((address_space(1))) v4i glob_v4i;

// Per-wave context: this wave's 64 LDS rows and the lane number.
struct Wave { lds_double* rows; unsigned lane; lds_double* ctx; };     // ctx: the lane's five doubles of ObjCache::ctx, as an LDS address

// Wave-cooperative gather of one 256-byte record per requesting lane into an LDS row.
// `key` names the record by its offset in 128-byte units from `blas`: a collapsed node, or a pair
// of triangle records.  The requests are COMPACTED: the r-th requesting lane (r = its rank among
// the requesters) owns row r, and only ceil(n/4) load rounds are issued for n requests.  Returns
// the row this lane reads afterwards (its rank; lanes without a request get some row too, whose
// contents mean nothing).
//
// Sixteen lanes read one record as sixteen adjacent 16-byte pieces, so one load instruction
// touches 4 records x 256 contiguous bytes instead of 64 scattered pieces.  The loads are
// LDS-DMA (global_load_lds_dwordx4, destination = wave-uniform base + 16*lane): in round g
// lane l serves row 4g + l/16, and the round fills rows 4g..4g+3.  The lanes learn the rows'
// keys through the (still unused) row memory: the owner of row w stores its key at dword
// (w & 3)*16 + (w >> 2), so the sixteen keys a lane needs are 64 contiguous bytes.  Before that
// every slot is preset with the first requester's key (two LDS writes of one wave take effect in
// order): the last round's unused rows then fetch a record that is on its way anyway, and no load
// needs a per-lane condition -- a round costs four instructions, and the rounds beyond the last
// request are skipped by scalar branches.
// To keep the owners' later 16-byte row reads free of bank conflicts the pieces are stored
// XOR-swizzled: slot s of row w holds piece s ^ (w & 15) -- applied to the *source* address
// here and to the slot index in row_read.  Must be executed by every lane of the wave.
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) v4i lds_v4i;
RTX_DEV unsigned coop_fetch(const Wave& wv, const bool wants, const int key, const char* __restrict__ blas) {
    const unsigned lane = wv.lane, quarter = lane >> 4, slot = lane & 15u;
    const unsigned long long m_req = __ballot(wants);
    const unsigned lo = (unsigned)m_req, hi = (unsigned)(m_req >> 32);
    const unsigned rank = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    const int n_req = __popcll(m_req);
    if (n_req == 0) return rank;
    const int key0 = __builtin_amdgcn_readlane(key, (int)__builtin_ctzll(m_req));
    ((lds_int*)wv.rows)[(lane & 3u) * 16u + (lane >> 2)] = key0;
    if (wants) ((lds_int*)wv.rows)[(rank & 3u) * 16u + (rank >> 2)] = key;
    __builtin_amdgcn_wave_barrier();
    int okeys[16];
    {
        const lds_v4i* kp = (const lds_v4i*)wv.rows + quarter * 4u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const v4i v = kp[q];
            okeys[4 * q] = v.x; okeys[4 * q + 1] = v.y; okeys[4 * q + 2] = v.z; okeys[4 * q + 3] = v.w;
        }
    }
    // the keys must have left the row memory before the first record lands in it
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0); vmcnt and expcnt left at "no wait"
    __builtin_amdgcn_wave_barrier();
    // Source address of this lane's piece in round g: piece = slot ^ ((4g + quarter) & 15)
    // = (slot ^ quarter) ^ 4(g & 3), so four per-lane bases cover the sixteen rounds; the record
    // offset is key * 128, formed together with the base by one 32x32+64-bit multiply-add (the
    // factor is kept opaque so that it is not split into a 64-bit shift and a 64-bit add).
    unsigned unit = 128u;
    asm volatile("" : "+s"(unit));
    // (likewise the rows' LDS address: sixteen round destinations computed here with one scalar
    // add each, instead of sixteen loop-invariant values that would have to live in registers)
    unsigned rows_at = (unsigned)(size_t)wv.rows;
    asm volatile("" : "+s"(rows_at));
    unsigned long long base[4];
#pragma unroll
    for (int j = 0; j < 4; j++) base[j] = (unsigned long long)blas + (((slot ^ quarter) ^ (4u * j)) << 4);
#pragma unroll
    for (int g = 0; g < 16; g++) {
        if (4 * g >= n_req) break;               // (wave-uniform: a scalar branch to the end)
        const unsigned long long gp = base[g & 3] + (unsigned long long)(unsigned)okeys[g] * unit;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                         (__attribute__((address_space(3))) void*)(size_t)(rows_at + (unsigned)g * (4u * RTX_ROW_DOUBLES * 8u)),
                                         16, 0, 0);
    }
    // vmcnt(0), as a builtin rather than inline assembly so that the compiler's own wait-count
    // bookkeeping knows that nothing is outstanding after it (gfx9 encoding: vmcnt bits 3:0 and
    // 15:14 zero, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_wave_barrier();
    return rank;
}

// Read pieces FIRST .. FIRST+N-1 of row `row` (undoing the swizzle) into out[0 .. 2N).
// The rows are 256-byte aligned, so the address of piece p is (row address | swizzle << 4) ^ (p << 4):
// one XOR per read from a single register.  That register is made opaque on every call: otherwise
// the sixteen addresses are formed one by one and held in (or spilled from) sixteen registers.
template <int FIRST, int N>
RTX_DEV void row_read(const Wave& wv, const unsigned row, double* out) {
    unsigned at = ((unsigned)(size_t)wv.rows + row * (RTX_ROW_DOUBLES * 8u)) | ((row & 15u) << 4);
#ifndef RTX_ROW_HOIST
    asm volatile("" : "+v"(at));
#endif
#pragma unroll
    for (int p = 0; p < N; p++) {
        const v2d v = *(const lds_v2d*)(size_t)(at ^ ((unsigned)(FIRST + p) << 4));
        out[2 * p] = v.x; out[2 * p + 1] = v.y;
    }
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

// ---------------------------------------------------------------------------------------
// Per-lane ray state of the trace kernel.
//
// A lane owns one ray at a time and walks it through the reference's object sequence
// (utils.cpp:671-865): planes -> free triangles -> meshes (in index order) -> spheres ->
// instances (in index order).  Meshes and instances are *traversal jobs*: (mesh, object-space
// ray, bound).  The wave runs all its lanes' jobs together in one cooperative traversal
// loop; a lane whose job ends waits until enough lanes need service, then the wave runs one
// divergent "service pass" that finishes jobs, selects each lane's next job, retires
// completed rays and claims fresh rays for idle lanes.  This keeps the lanes of a wave busy
// although rays differ wildly in length.
//
// Only what a traversal step touches lives in registers.  The Intersection record
// (scene.h:248-264) lives in HBM, in the ray's slot of the hit buffer: the few events that
// write it (an accepted hit, or the half-written record of utils.cpp:567-570 / 719-721)
// store exactly the fields the reference writes at that point, so the slot always holds
// what the reference's `intersection` holds; registers keep only the bound, the record's
// distance and the hit flag.
// ---------------------------------------------------------------------------------------
enum { LS_EMPTY = 0, LS_NEED_JOB = 1, LS_JOB_DONE = 2, LS_IN_JOB = 3 };

#ifndef RTX_SERVICE_THRESHOLD
#define RTX_SERVICE_THRESHOLD 24     // lanes waiting for service that trigger a service pass ...
#endif
#ifndef RTX_SERVICE_THRESHOLD_INSTANCED
#define RTX_SERVICE_THRESHOLD_INSTANCED 36   // ... and in scenes where a pass replays many instance candidates
#endif
#ifndef RTX_SERVICE_REPS
#define RTX_SERVICE_REPS 2           // rounds of a service pass: lanes whose ray ended in one round are refilled in the next
#endif
#ifndef RTX_SPARSE_LANES
#define RTX_SPARSE_LANES 2           // at most this many traversing lanes: load records directly
#endif
#ifndef RTX_LINEAR_INSTS
#define RTX_LINEAR_INSTS 4           // up to this many instances are tried one after the other, without a TLAS walk (< 32)
#endif
#ifndef RTX_SCATTER_THIN_RAYS
#define RTX_SCATTER_THIN_RAYS (1u << 19)   // generations of fewer rays are scattered in finer groups:
#endif
#ifndef RTX_SCATTER_SHIFT_THIN
#define RTX_SCATTER_SHIFT_THIN 0           // log2 of their group size
#endif
#ifndef RTX_REPLAY_CAP
#define RTX_REPLAY_CAP 3             // instance candidates a lane may turn down per service visit before it yields (instance-heavy scenes)
#endif
#ifndef RTX_LDS_OBJS
#define RTX_LDS_OBJS 10              // object records (512 B each) kept in LDS per block: the first so many of the scene
#endif
#ifndef RTX_SCATTER0_SHIFT
#define RTX_SCATTER0_SHIFT 6         // generation 0: whole 8x8 pixel blocks (64 items) are scattered
#endif
#ifndef RTX_SCATTER_SHIFT
#define RTX_SCATTER_SHIFT 12         // rays of later generations are handed out in scattered groups of 4096 (finer costs locality)
#endif
#ifndef RTX_INST_QUORUM
#define RTX_INST_QUORUM 16           // lanes at an instance slot that are waited for before the wave runs the instance set-up (1: none)
#endif
#ifndef RTX_LEAF_QUORUM
#define RTX_LEAF_QUORUM 16           // ... lanes at leaves, before it runs the triangle tests
#endif
#ifndef RTX_NODE_QUORUM
#define RTX_NODE_QUORUM 16           // ... lanes at nodes, before it runs the box tests
#endif
#ifndef RTX_CHUNK
#define RTX_CHUNK 64                 // rays a wave claims per atomic, at most ...
#endif
#ifndef RTX_CHUNK_INSTANCED
#define RTX_CHUNK_INSTANCED 256       // ... and in scenes dominated by instance replay
#endif
#ifndef RTX_TRACE_WAVES
#define RTX_TRACE_WAVES 2            // waves per SIMD the trace kernel is compiled for (2: 256 VGPRs)
#endif

struct Lane {
    V3 o, d;                // world-space ray
    double t_min, dist;     // current bound; Intersection::distance of the record
    unsigned slot;          // queue slot (work item for generation 0) = hit-record slot
    int is_shadow, hit;
    // object iterator: it <= n_meshes: meshes (job = object it-1); it > n_meshes: instances, taken in
    // ascending index from the lane's two-level instance bitmap: bitmap word `word` is being
    // consumed (`bits` = its remaining candidates); which words hold any is recorded in summary
    // words, of which `sword` is being consumed (`sbits`) and `whi` is the last that has any.
    // While the TLAS walk runs, word/whi are the lowest/highest bitmap word it has marked.
    int it, word, whi, sword;
    unsigned bits, sbits;
    // current traversal job
    int obj;
    double orig_t_min;
    V3 ro, rd;
    double ix, iy, iz;
    int par;
    int cur_ref, cur_cnt, sp, btri;
    int state;
    int age;                // traversal steps spent on the current job (with counters on: on the current ray)
    v4i top;                // top stack entry (valid while sp > 0)
    double bound_before;    // eight-wide records: the bound that the job's best triangle was tested against
};

// Where a lane's record goes.  Shadow rays need only (hit, distance), which stay in registers.
struct RecOut {
    RTX_GLOBAL Hit* rec; bool on; bool ids;
    RTX_DEV void geom(double dist, V3 p, V3 g, V3 s) const {
        if (!on) return;
        rec->dist = dist;
        rec->p[0] = p.x; rec->p[1] = p.y; rec->p[2] = p.z;
        rec->g[0] = g.x; rec->g[1] = g.y; rec->g[2] = g.z;
        rec->s[0] = s.x; rec->s[1] = s.y; rec->s[2] = s.z;
    }
    RTX_DEV void bary(double beta, double gamma, int face, int tri) const {
        if (!on) return;
        rec->beta = beta; rec->gamma = gamma; rec->face = face; rec->tri = tri;
    }
    RTX_DEV void owner(int mat, int container) const { if (on) { rec->mat = mat; rec->container = container; } }
    RTX_DEV void kind(int k) const { if (on) rec->kind = k; }
    RTX_DEV void src(int s) const { if (on && ids) rec->src = s; }
};

// A fresh ray: empty record, then the classes that come before the meshes
// (planes, utils.cpp:683-685; free triangles, utils.cpp:692-729).
// The scene's first object records, copied to LDS when the block starts.  A lane that moves on
// to its next object reads the record in a chain of dependent loads (flags and boxes, then the
// matrix, then the root); from LDS each link costs a fraction of a trip to L2.  Records beyond
// the cached ones are read from memory (one pointer type serves both).
// ctx: five doubles of LDS per lane for what a shadow ray needs again after its traversal --
// its light's distance (at every change of object) and, if it ends unoccluded, pixel and
// weight -- taken from the ray record together with origin and direction, in the same trip.
struct ObjCache {
    const Obj* lds;
    const double* tiny;   // the triangle pairs of the scene's tiny meshes (RTX_LDS_TINY slots of 256 bytes)
    int n;
    double* ctx;          // [0] light distance, [1..3] weight, [4] pixel (low word)
    RTX_DEV const Obj* at(const Scene& S, int oi) const { return oi < n ? lds + oi : S.objs + oi; }
};

template <bool COUNT>
RTX_DEV void ray_begin(const Scene& S, Lane& L, const RecOut& out, const double bound0, Tally& tl) {
    L.t_min = bound0;                 // DBL_MAX in the reference (utils.cpp:676)
    L.dist = 0.0;
    L.hit = 0;
    if (S.n_planes > 0 || S.n_ftris > 0) {
        Rec r;
        r.hit = 0; r.dist = 0.0; r.beta = 0.0; r.gamma = 0.0;
        r.point = mk(0, 0, 0); r.gn = mk(0, 0, 0); r.sn = mk(0, 0, 0);
        r.kind = RTX_KIND_NONE; r.container = -1; r.face = -1; r.tri = -1; r.mat = -1; r.src = -1;
        bool hit = false;
        for (int i = 0; i < S.n_planes; i++) hit = plane_hit(S, L.o, L.d, i, L.t_min, r) || hit;
        for (int i = 0; i < S.n_ftris; i++) free_tri_hit<COUNT>(S, L.o, L.d, i, L.t_min, r, hit, tl);
        if (r.hit) {
            out.geom(r.dist, r.point, r.gn, r.sn);
            out.bary(r.beta, r.gamma, r.face, r.tri);
            out.owner(r.mat, r.container);
            out.kind(r.kind);
            out.src(r.src);
            L.dist = r.dist;
        }
        L.hit = hit ? 1 : 0;
    }
    L.it = 0; L.word = 0; L.whi = -1; L.bits = 0u; L.sword = 0; L.sbits = 0u; L.age = 0;
    L.state = LS_NEED_JOB;
}

// Advance the lane's object iterator until an object needs traversal (returns 1, job set
// up: rayHitsMesh's prologue, utils.cpp:418-494 + the root test of bvh.cpp:132), the object
// sequence is exhausted (returns 0), or the lane has turned down CAP instance candidates in
// this call (returns 2: call again in the next service pass).
template <bool COUNT, bool FAST, int CAP, int NEST, bool B8>
RTX_DEV int job_next(const Scene& S, const ObjCache& oc, Lane& L, const RecOut& out, unsigned* inst_bits, glob_v4i* stk, const double light_dist, Tally& tl) {
    const unsigned n_summary = rtx_summary_words(S.n_insts);
    const Slab ws = make_slab(L.o, L.d);         // world-space slab constants, once per call
    int turned_down = 0;
    while (true) {
        // A service pass lasts as long as its slowest lane.  A lane that keeps turning instance
        // candidates down (each a chain of dependent loads) yields after CAP of them and goes on in
        // the next pass, instead of holding up the lanes that already have their jobs (CAP = 0: no
        // limit; the variant for scenes with few instances has none).
        if (CAP > 0 && L.it > S.n_meshes && turned_down++ >= CAP) return 2;
        int oi;
        if (L.it < S.n_meshes) {                                     // utils.cpp:736-739
            oi = L.it; L.it++;
        } else {
            if (L.it == S.n_meshes) {                                // utils.cpp:746-748
                L.it++;
                if (S.n_spheres > 0) {
                    Rec r;
                    r.kind = -1; r.dist = 0.0; r.hit = 0; r.mat = -1; r.container = -1; r.src = -1;
                    r.point = mk(0, 0, 0); r.gn = r.point; r.sn = r.point;
                    bool hit = L.hit != 0;
                    for (int i = 0; i < S.n_spheres; i++) hit = sphere_hit(S, L.o, L.d, i, L.t_min, r) || hit;
                    if (r.kind != -1) {                              // fields a sphere hit writes, utils.cpp:269-291
                        out.geom(r.dist, r.point, r.gn, r.sn);
                        out.owner(r.mat, r.container);
                        out.kind(r.kind);
                        out.src(r.src);
                        L.dist = r.dist;
                    }
                    L.hit = hit ? 1 : 0;
                }
                // Instances (utils.cpp:756-790).  A walk of the TLAS -- a traversal job like a
                // mesh's, run by the cooperative loop with the world-space ray -- marks the
                // candidates in the lane's bitmap: instances whose world box the ray can meet
                // within the current bound (a box missed now stays missed because the bound
                // never grows).  They are then replayed in ascending index through the
                // reference's own per-instance logic (world-box test against the then-current
                // bound included).
                L.word = INT_MAX; L.whi = -1; L.bits = 0u; L.sbits = 0u; L.sword = -1;
                if (NEST == 1 || (NEST == 2 && !(L.age & SP_REPLAY))) {
                    // the walk enters the instances itself, nearest first (see traverse_steps):
                    // word = stack base of the job inside an instance, whi = the instance the
                    // current nearest hit came from (-1: none, or something that is not an instance)
                    L.word = 0; L.whi = -1;
                    double tl_min[3] = {S.tl_min[0], S.tl_min[1], S.tl_min[2]}, tl_max[3] = {S.tl_max[0], S.tl_max[1], S.tl_max[2]};
                    double enter;
                    if (!box_enter_svc(tl_min, tl_max, ws, enter) || !(L.t_min * 1.000001 >= enter)) return 0;
                    L.obj = -1;
                    L.is_shadow &= 1;
                    if (NEST == 1 && L.is_shadow) L.dist = L.t_min;      // a shadow ray's bound in world space, for the way back from an instance
                    if (NEST == 2) {
                        // exact mode's search (search_service): the record's distance is parked in the
                        // lane's stack area, its register carries the bound the instance loop starts with
                        v4i h; h.x = __double2loint(L.dist); h.y = __double2hiint(L.dist); h.z = 0; h.w = 0;
                        stk[RTX_STACK - 1] = h;
                        L.dist = L.t_min;
                        L.sword = -1; L.sbits = 0x7FFFFFFFu;
                        L.age = SP_WALK;
                    } else {
                        L.age = 0;
                    }
                    L.orig_t_min = L.t_min;
                    L.ro = L.o; L.rd = L.d; L.ix = ws.ix; L.iy = ws.iy; L.iz = ws.iz; L.par = ws.par;
                    L.cur_ref = S.tl_key; L.cur_cnt = 0; L.sp = 0; L.btri = -1;
                    return 1;
                }
                if (S.n_insts > 0 && S.n_insts <= RTX_LINEAR_INSTS) {
                    // a handful of instances: every one is a candidate (the reference's own loop,
                    // utils.cpp:756-790); a TLAS walk would cost more than it can save
                    L.word = 0; L.bits = (1u << S.n_insts) - 1u; L.sword = 0; L.whi = 0;
                } else if (S.n_insts > 0) {
                    double tl_min[3] = {S.tl_min[0], S.tl_min[1], S.tl_min[2]}, tl_max[3] = {S.tl_max[0], S.tl_max[1], S.tl_max[2]};
                    double enter;
                    if (box_enter_svc(tl_min, tl_max, ws, enter) && (L.t_min >= enter)) {
                        L.obj = -1;                                  // marks the job as the TLAS walk
                        L.is_shadow &= 1;
                        if (!COUNT && NEST != 2) L.age = 0;
                        L.orig_t_min = L.t_min;
                        L.ro = L.o; L.rd = L.d; L.ix = ws.ix; L.iy = ws.iy; L.iz = ws.iz; L.par = ws.par;
                        L.cur_ref = S.tl_key; L.cur_cnt = 0; L.sp = 0; L.btri = -1;
                        return 1;
                    }
                }
            }
            while (L.bits == 0u) {
                while (L.sbits == 0u) {
                    if (L.sword >= L.whi) return 0;
                    L.sword++;
                    L.sbits = atomicExch(inst_bits + L.sword, 0u);           // summary words come first in the lane's area
                }
                L.word = L.sword * 32 + (__ffs((int)L.sbits) - 1);
                L.sbits &= L.sbits - 1u;
                L.bits = atomicExch(inst_bits + n_summary + L.word, 0u);     // take the word's candidates, leave it clear
            }
            const int idx = L.word * 32 + (__ffs((int)L.bits) - 1);
            L.bits &= L.bits - 1u;
            oi = S.n_meshes + idx;
        }
        // second 128 bytes of the object record in one batch: world box, root box, root, flags
        const Obj* ob = oc.at(S, oi);
        double ob2[16];
        {
            const v2d* p2 = (const v2d*)ob->bmin;
#pragma unroll
            for (int k = 0; k < (B8 ? 8 : 7); k++) { const v2d v = p2[k]; ob2[2 * k] = v.x; ob2[2 * k + 1] = v.y; }
        }
        // While that record is on its way: if this was its bitmap word's last candidate, take the next
        // word (or the next summary word) now, in the same trip, instead of in a trip of its own later.
        if (oi >= S.n_meshes && L.bits == 0u) {
            if (L.sbits != 0u) {
                L.word = L.sword * 32 + (__ffs((int)L.sbits) - 1);
                L.sbits &= L.sbits - 1u;
                L.bits = atomicExch(inst_bits + n_summary + L.word, 0u);
            } else if (L.sword < L.whi) {
                L.sword++;
                L.sbits = atomicExch(inst_bits + L.sword, 0u);
            }
        }
        const int root_ref = __double2loint(ob2[12]), root_count = __double2hiint(ob2[12]);
        const int flags = __double2loint(ob2[13]);
        if (flags & OBJ_SKIP) continue;
        if ((flags >> 8) != 0) {
            // A tiny mesh (untransformed, one leaf of one or two triangles, kept in LDS): tested here.
            // As a traversal job it would be one step between two service passes; the tests, their
            // order and what an accepted hit writes are the job's (root box of bvh.cpp:132, leaf
            // triangles in order, utils.cpp:370-381 and 427-442).
            double enter;
            if (!box_enter_svc(ob2 + 6, ob2 + 9, ws, enter) || !(L.t_min >= enter)) continue;
            const double* T = oc.tiny + ((flags >> 8) - 1) * 32;
            int won = -1;
            double beta = 0.0, gamma = 0.0;
            for (int k = 0; k < 2; k++) {
                if (k >= root_count) break;
                const double* tk = T + 16 * k;
                if (COUNT) tl.tri_tests++;
                if (S.cull && L.d.x * tk[10] + L.d.y * tk[11] + L.d.z * tk[12] > 0.0) continue;      // utils.cpp:322-326
                double t, b, g;
                if (tri_test(tk, L.o, L.d, L.t_min, S.isect_eps, t, b, g)) { L.t_min = t; won = k; beta = b; gamma = g; }
            }
            if (won >= 0) {
                L.dist = L.t_min; L.hit = 1;
                if (out.on) {
                    const double* tk = T + 16 * won;
                    const V3 a = mk(tk[0], tk[1], tk[2]), e1 = mk(tk[3], tk[4], tk[5]), e2 = mk(tk[6], tk[7], tk[8]);
                    const V3 gn = mk(tk[10], tk[11], tk[12]);
                    const V3 point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
                    out.bary(beta, gamma, __double2loint(tk[9]), root_ref - S.tri_key0 + won);
                    out.kind(RTX_KIND_MESH);
                    out.geom(L.t_min, point, gn, gn);
                    out.owner(ob->material, ob->mesh);
                    out.src(oi);
                }
                // a shadow ray with an accepted hit in front of its light is decided (see below)
                if (!COUNT && L.is_shadow && (FAST || L.t_min < light_dist)) return 0;
            }
            continue;
        }
        V3 ro = L.o, rd = L.d;
        V3 od = L.d;                                                 // the direction in object space, before it is normalised
        double bound = L.t_min;
        if (flags & OBJ_XF) {
            if (flags & OBJ_BOUNDS) {                                // utils.cpp:446-466, 760-766
                if (COUNT) tl.world_box++;
                double enter;
                if (!box_enter_svc(ob2 + 0, ob2 + 3, ws, enter) || !(L.t_min >= enter)) continue;
            }
            if (COUNT) tl.obj_trav++;
            double inv[16];
            {
                const v2d* p1 = (const v2d*)ob->inv;
#pragma unroll
                for (int k = 0; k < 8; k++) { const v2d v = p1[k]; inv[2 * k] = v.x; inv[2 * k + 1] = v.y; }
            }
            ro = xf_point(inv, L.o);                                 // utils.cpp:481-489
            od = xf_dir(inv, L.d);
            rd = normalize_f32(od);
            if (FAST) {
                // consistent mode: the bound goes to object space with the ray (the divisor is the
                // factor normalize_f32 applied to the direction)
                bound = L.t_min / (1.0 / (double)sqrtf((float)(od.x * od.x + od.y * od.y + od.z * od.z)));
            }
        }
        // (an untransformed object is met by the world-space ray: same constants, no second set of divisions)
        const Slab sl = (flags & OBJ_XF) ? make_slab(ro, rd) : ws;
        double enter;
        if (!box_enter_svc(ob2 + 6, ob2 + 9, sl, enter) || !(bound >= enter)) continue;
        L.obj = oi;
        L.orig_t_min = L.t_min;
        if (FAST) L.t_min = bound;
        L.is_shadow &= 1;
        if (!FAST && !COUNT && L.is_shadow && (flags & OBJ_XF) && (flags & OBJ_AFFINE)) {
            // A shadow ray inside a transformed object (utils.cpp:476-570 decides acceptance only at
            // the job's end, from the float-rounded world distance of the nearest triangle).  Under
            // an affine transform that distance is t * f up to float rounding, f the factor
            // normalize_f32 applied to the direction, and the nearest triangle is no farther than
            // any accepted one: an accepted t with t * f * (1 + 1e-6) below both the bound the job
            // started with and the light's distance means the job ends with an accepted hit nearer
            // than the light, which decides the ray (see the untransformed case below).  The
            // threshold, rounded down to float, rides in the spare half of the context's pixel word.
            const double f = 1.0 / (double)sqrtf((float)(od.x * od.x + od.y * od.y + od.z * od.z));
            const double thr = smin(L.t_min, light_dist) / (f * 1.000001);
            float thr_f = (float)thr;
            if ((double)thr_f > thr) thr_f = __int_as_float(__float_as_int(thr_f) - 1);       // (thr is positive)
            if (thr_f > 0.0f && thr == thr) {
                oc.ctx[4] = __hiloint2double(__float_as_int(thr_f), __double2loint(oc.ctx[4]));
                L.is_shadow = 5;
            }
        }
        if (!COUNT && NEST != 2) L.age = 0;       // (steps of the job; with counters on: of the ray, for the histogram)
        if (FAST && L.is_shadow) {
            // consistent mode: a shadow ray's bound starts at the light's distance, so the first
            // accepted triangle of any job, transformed or not, decides it
            L.is_shadow = 3;
            L.orig_t_min = bound;
        } else if (!COUNT && L.is_shadow && !(flags & OBJ_XF)) {
            // A shadow ray asks only whether something lies in front of the light
            // (utils.cpp:1066-1084: hit && distance < |L - P|).  Once a hit has been accepted
            // with the bound below that distance the answer is yes whatever follows: the bound
            // never grows, every later write to the record's distance is below the bound of
            // its moment, and the hit flag is never taken back.  In an untransformed mesh an
            // accepted triangle is such a hit, so the job may stop there (bit 1 of is_shadow;
            // orig_t_min, unused by untransformed jobs, carries the light's distance).
            // Transformed jobs decide acceptance only at their end (utils.cpp:559-570).
            // With counters on every test of the reference is carried out.
            L.is_shadow = 3;
            L.orig_t_min = light_dist;
        }
        L.ro = ro; L.rd = rd; L.ix = sl.ix; L.iy = sl.iy; L.iz = sl.iz; L.par = sl.par;
        L.cur_ref = B8 ? __double2hiint(ob2[14]) : root_ref;        // (the same root in the eight-wide BLAS: Obj::root8_ref)
        L.cur_cnt = root_count; L.sp = 0; L.btri = -1;
        return 1;
    }
}

// A shadow ray that nothing in the scene can hold up, decided where the ray would be emitted: every mesh
// and instance turns it down at the FIRST test job_next would make for it -- the world box of a
// transformed object (utils.cpp:446-466, 760-766), the root box of an untransformed one (bvh.cpp:132), the
// TLAS root for instances beyond the linear few -- with the same arithmetic on the same origin and
// direction the queue would carry; the bound plays no part (a box missed for every bound is missed
// for the light's distance, the consistent mode's bound, as for DBL_MAX, the default mode's).  Such a
// ray runs no job, accepts no hit and ends by adding its unshadowed contribution (service_lane, "ray
// complete"), which the caller does at once instead.
// Only for scenes without planes, spheres and free triangles (the host decides: Launch::early_shadow);
// anything not plainly turned down -- an object without a world box, a box test that passes -- leaves the
// ray to the trace kernel.
RTX_DEV bool shadow_ray_is_free(const Scene& S, V3 o, V3 d) {
    const Slab ws = make_slab(o, d);
    const int n_lin = S.n_meshes + (S.n_insts <= RTX_LINEAR_INSTS ? S.n_insts : 0);
    for (int oi = 0; oi < n_lin; oi++) {
        const RTX_GLOBAL v2d* p2 = (const RTX_GLOBAL v2d*)((const RTX_GLOBAL Obj*)S.objs + oi)->bmin;
        double ob2[14];
#pragma unroll
        for (int k = 0; k < 7; k++) { const v2d v = p2[k]; ob2[2 * k] = v.x; ob2[2 * k + 1] = v.y; }
        const int flags = __double2loint(ob2[13]);
        if (flags & OBJ_SKIP) continue;
        double enter;
        if (flags & OBJ_XF) {
            if (!(flags & OBJ_BOUNDS) || box_enter_svc(ob2 + 0, ob2 + 3, ws, enter)) return false;
        } else if (box_enter_svc(ob2 + 6, ob2 + 9, ws, enter)) {
            return false;
        }
    }
    if (S.n_insts > RTX_LINEAR_INSTS) {
        double tl_min[3] = {S.tl_min[0], S.tl_min[1], S.tl_min[2]}, tl_max[3] = {S.tl_max[0], S.tl_max[1], S.tl_max[2]};
        double enter;
        if (box_enter_svc(tl_min, tl_max, ws, enter)) return false;
    }
    return true;
}

// Instances entered from the walk of the TLAS (see traverse_steps, NEST), after a job that found
// something: back to the walk with the world-space ray.  Returns 1 if the walk has entries left
// (the lane asks for a pop), 0 if the ray has seen every instance it can meet.
RTX_DEV int nest_resume(Lane& L) {
    const Slab ws = make_slab(L.o, L.d);
    L.obj = -1; L.word = 0;
    L.ro = L.o; L.rd = L.d; L.ix = ws.ix; L.iy = ws.iy; L.iz = ws.iz; L.par = ws.par;
    L.btri = -1;
    if (L.sp <= 0) return 0;
    L.cur_ref = 0; L.cur_cnt = -1;
    return 1;
}
// End of a traversal job: what rayHitsMesh does after bvh->traverse returns
// (utils.cpp:533-668), including the fields the winning triangle test wrote
// (utils.cpp:370-381, bvh.cpp:162).  beta and gamma of the winning triangle are recomputed
// here with the same arithmetic that accepted it (same inputs, same bits).
// A job of the eight-wide BLAS has ended at triangle L.btri, object-space distance L.t_min.  The
// traversal visited a superset of the reference's leaves in the reference's order and accepted with
// the reference's own test, so this is the reference's triangle unless it lies in a leaf the
// reference's box test (bvh.cpp:23-56 at bvh.cpp:132) keeps the ray out of.  The reference enters a
// leaf when that test passes with the bound of the moment.  This traversal's bound at the same point of
// the visiting order is never above the reference's (it has accepted everything the reference has, and
// possibly more), so the test on the leaf's exact box with the bound this lane held when it accepted
// the triangle (Lane::bound_before) says "entered" only if the reference's says so too.  An accepted
// triangle that was not the reference's to test can only have lowered the bound: whatever that kept
// out lies at or beyond it, hence beyond the final hit, and the final hit is the one checked.  When the
// check fails (a hit within rounding of its leaf's box faces, met through the wider float boxes), the
// frame is flagged and the host renders it again with the four-wide kernels.
RTX_DEV void b8_verify(const Scene& S, const Lane& L, const RTX_GLOBAL Tri* Tg) {
    Slab sl;
    sl.ox = L.ro.x; sl.oy = L.ro.y; sl.oz = L.ro.z; sl.ix = L.ix; sl.iy = L.iy; sl.iz = L.iz; sl.par = L.par;
    double enter;
    {
        // First the leaf's box rounded inward to floats, from the triangle's own record (the line the traversal
        // has just had; the exact boxes are a cold array): a box inside the exact one that lets the ray in under
        // the bound says that the exact one does (the slab arithmetic is monotone in every plane).
        const RTX_GLOBAL v2f* ib = (const RTX_GLOBAL v2f*)Tg->pad2;
        const v2f p = ib[0], q = ib[1], r = ib[2];
        const double mn[3] = {(double)p.x, (double)p.y, (double)q.x}, mx[3] = {(double)q.y, (double)r.x, (double)r.y};
        if (mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2] && box_enter(mn, mx, sl, enter) && L.bound_before >= enter) return;
    }
    const RTX_GLOBAL v2d* ax = (const RTX_GLOBAL v2d*)(S.tri_aux + (L.btri - S.t8_key0));
    const v2d a = ax[0], b = ax[1], c = ax[2];
    const double mn[3] = {a.x, a.y, b.x}, mx[3] = {b.y, c.x, c.y};
    if (!box_enter(mn, mx, sl, enter) || !(L.bound_before >= enter))
        __hip_atomic_store(S.limit_flag + (offsetof(GenCtrl, inexact) - offsetof(GenCtrl, limit)) / 4, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool FAST, int NEST, bool B8>
RTX_DEV void job_finalize(const Scene& S, const ObjCache& oc, Lane& L, const RecOut& out) {
    if (L.btri < 0) {
        if (FAST) L.t_min = L.orig_t_min;         // (a transformed job ran with the object-space bound)
        return;
    }
    const RTX_GLOBAL Tri* Tg = (const RTX_GLOBAL Tri*)(S.blas + (size_t)(unsigned)L.btri * 128u);
    // (a job that the wave-wide traversal finished ends at a triangle of the reference-order array, found
    // through the exact four-wide boxes: nothing to check)
    const bool fast_key = B8 && L.btri >= S.t8_key0;
    if (fast_key) b8_verify(S, L, Tg);
    const Obj* ob = oc.at(S, L.obj);
    if (!out.on && !(ob->flags & OBJ_XF)) {
        // a shadow ray keeps no record: of an untransformed hit only distance and flag remain
        L.dist = L.t_min; L.hit = 1;
        return;
    }
    // triangle index in the reference-order array (the traversal works with record keys; the
    // eight-wide BLAS's triangles are copies in its own order that carry that index)
    const int tri = fast_key ? Tg->pad : L.btri - S.tri_key0;
    double T[9];
#pragma unroll
    for (int k = 0; k < 3; k++) { T[k] = Tg->a[k]; T[3 + k] = Tg->e1[k]; T[6 + k] = Tg->e2[k]; }
    const int face = Tg->face;
    const V3 gn = mk(Tg->gn[0], Tg->gn[1], Tg->gn[2]);
    const int flags = ob->flags, mat = ob->material, mi = ob->mesh;
    double t, beta, gamma;
    tri_test(T, L.ro, L.rd, INFINITY, S.isect_eps, t, beta, gamma);
    const V3 a = mk(T[0], T[1], T[2]), e1 = mk(T[3], T[4], T[5]), e2 = mk(T[6], T[7], T[8]);
    const V3 point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
    if (!FAST) {
        out.bary(beta, gamma, face, tri);
        out.kind(RTX_KIND_MESH);
    }
    if (!(flags & OBJ_XF)) {
        if (NEST && L.it > S.n_meshes) {
            // an untransformed instance visited out of index order (NEST): its job ran with a bound
            // one ulp above the current one, so a hit at exactly the current distance shows up
            // here and the lower instance index takes it
            const int idx = L.obj - S.n_meshes;
            if (L.t_min >= L.orig_t_min && !(idx < L.whi)) { L.t_min = L.orig_t_min; return; }
            L.whi = idx;
        }
        if (FAST) { out.bary(beta, gamma, face, tri); out.kind(RTX_KIND_MESH); }
        out.geom(L.t_min, point, gn, gn);
        out.owner(mat, mi);
        out.src(L.obj);
        L.dist = L.t_min; L.hit = 1;
        return;
    }
    double m[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = ob->m[k];
    V3 wp = xf_point(m, point);
    double dx = wp.x - L.o.x, dy = wp.y - L.o.y, dz = wp.z - L.o.z;
    double wd = (double)sqrtf((float)(dx * dx + dy * dy + dz * dz));
    if (FAST) {
        // consistent mode: a hit that is not nearer than the bound leaves no trace.  Instances
        // visited nearest first (NEST): among hits at the same distance the lowest instance index
        // wins, as it does when the instances are taken in index order.
        const bool nested = NEST && L.it > S.n_meshes;
        const int idx = L.obj - S.n_meshes;
        const bool tie_won = nested && wd == L.orig_t_min && idx < L.whi;
        if (wd >= L.orig_t_min && !tie_won) { L.t_min = L.orig_t_min; return; }
        if (nested) L.whi = idx;
        out.bary(beta, gamma, face, tri);
        out.kind(RTX_KIND_MESH);
    }
    if (!FAST && wd >= L.orig_t_min) {    // utils.cpp:567-570: bound restored, record is not
        out.geom(L.t_min, point, gn, gn);
        out.owner(mat, mi);
        L.dist = L.t_min;
        L.t_min = L.orig_t_min;
        return;
    }
    if (!out.on) { L.dist = wd; L.t_min = wd; L.hit = 1; return; }      // (shadow ray: no record)
    double nrm[12];
#pragma unroll
    for (int k = 0; k < 12; k++) nrm[k] = ob->nrm[k];
    V3 wg = normalize_f32(xf_dir(nrm, gn));
    V3 ws;
    if (flags & OBJ_SMOOTH_VN) {          // utils.cpp:589-621
        double u = beta, v = gamma, w = 1.0 - u - v;
        const RTX_GLOBAL double* vn = (const RTX_GLOBAL double*)S.tri_vn + RTX_VN_STRIDE * (size_t)tri;
        double nx = w * vn[0] + u * vn[3] + v * vn[6];
        double ny = w * vn[1] + u * vn[4] + v * vn[7];
        double nz = w * vn[2] + u * vn[5] + v * vn[8];
        double nlen = (double)sqrtf((float)(nx * nx + ny * ny + nz * nz));
        double ninv = 1.0 / nlen;
        nx *= ninv; ny *= ninv; nz *= ninv;
        ws = xf_dir(nrm, mk(nx, ny, nz));
    } else {
        ws = xf_dir(nrm, gn);
    }
    {
        double slen = (double)sqrtf((float)(ws.x * ws.x + ws.y * ws.y + ws.z * ws.z));
        double sinv = 1.0 / slen;
        ws.x *= sinv; ws.y *= sinv; ws.z *= sinv;
    }
    if (dot(wg, L.d) > 0.0) { wg = vneg(wg); ws = vneg(ws); }
    out.geom(wd, wp, wg, ws);
    out.owner(mat, -2);
    out.src(L.obj);
    L.dist = wd;
    L.t_min = wd;
    L.hit = 1;
}

// ---------------------------------------------------------------------------------------
// Exact mode, scenes with many instances: the instance loop (utils.cpp:756-790) as a search.
//
// The reference takes the instances in index order; instance j is *found* when its nearest
// triangle lies at an object-space distance t_j below the bound t of that moment (the world
// distance of the nearest accepted hit so far -- compared with an object-space distance as it
// is), which overwrites the record's triangle fields; it is *accepted*, the record completed and
// t lowered, when that triangle's float-rounded world distance w_j is below t as well
// (utils.cpp:559-570).  Hence, with t0 the bound the loop starts with:
//   * Let k be the instance of least w among those found with bound t0 (lowest index on a tie).
//     Every instance accepted before k has w > w_k, so the bound at k's turn lies in (w_k, t0];
//     if k's traversal ends at the same triangle under a bound just above w_k and under t0 -- the
//     visited sets are nested and the nearest tested triangle wins, so it does under every bound
//     between -- k is found and accepted at its turn, whatever came before, and rewrites every
//     field an instance writes.  From then on t = w_k: nothing after k is accepted.
//   * Of the instances after k, those found with bound w_k overwrite the triangle fields again;
//     the last of them (highest index) leaves its values.  An instance not found under t0 is not
//     found under w_k either, and one found under t0 at t_j >= w_k is not.
//   * Without such a k the bound stays t0 and only the last found instance matters.
// So the walk of the TLAS enters instances nearest first (traverse_steps, NEST), pruning with the
// least w so far (plus a relative 1e-6: w is rounded to float, and t * |M d| is w only up to that
// rounding), runs each entered instance under bound t0 and lists those with a hit; then k is run
// again under the bound just above w_k (SP_VERIFY) and finalised, and the listed instances after
// k with t_j < w_k are run under w_k, highest index first, until one is found (SP_PROBE).  Any
// outcome the argument does not cover (the verification fails, the list is full, an instance
// after k is accepted after all) starts the ray again with the plain replay in index order
// (SP_REPLAY).  An instance that matters at w_k has its world box within w_k, which no pruning
// of the walk cuts off, so it is entered and listed.
// ---------------------------------------------------------------------------------------

// rayHitsMesh's prologue for instance `idx` with the bound `bound` (utils.cpp:418-494, bvh.cpp:132).
// 1: the traversal job is set up on top of the lane's stack; 0: the instance is turned down.
RTX_DEV int search_enter(const Scene& S, const ObjCache& oc, Lane& L, const int idx, const double bound) {
    const int oi = S.n_meshes + idx;
    const Obj* ob = oc.at(S, oi);
    double ob2[16];
    {
        const v2d* p2 = (const v2d*)ob->bmin;
#pragma unroll
        for (int k = 0; k < 7; k++) { const v2d v = p2[k]; ob2[2 * k] = v.x; ob2[2 * k + 1] = v.y; }
    }
    const int root_ref = __double2loint(ob2[12]), root_count = __double2hiint(ob2[12]);
    const int flags = __double2loint(ob2[13]);
    if (flags & OBJ_SKIP) return 0;
    const Slab ws = make_slab(L.o, L.d);
    V3 ro = L.o, rd = L.d;
    if (flags & OBJ_XF) {
        if (flags & OBJ_BOUNDS) {
            double enter;
            if (!box_enter_svc(ob2 + 0, ob2 + 3, ws, enter) || !(bound >= enter)) return 0;
        }
        double inv[16];
        {
            const v2d* p1 = (const v2d*)ob->inv;
#pragma unroll
            for (int k = 0; k < 8; k++) { const v2d v = p1[k]; inv[2 * k] = v.x; inv[2 * k + 1] = v.y; }
        }
        ro = xf_point(inv, L.o);
        rd = normalize_f32(xf_dir(inv, L.d));
    }
    const Slab sl = (flags & OBJ_XF) ? make_slab(ro, rd) : ws;
    double enter;
    if (!box_enter_svc(ob2 + 6, ob2 + 9, sl, enter) || !(bound >= enter)) return 0;
    L.obj = oi;
    L.word = L.sp;
    L.orig_t_min = L.t_min;
    L.t_min = bound;
    L.is_shadow &= 1;
    L.ro = ro; L.rd = rd; L.ix = sl.ix; L.iy = sl.iy; L.iz = sl.iz; L.par = sl.par;
    L.cur_ref = root_ref; L.cur_cnt = root_count; L.btri = -1;
    return 1;
}

// World distance of the hit a job has found, with job_finalize's arithmetic (utils.cpp:559-566).
RTX_DEV double found_world_distance(const Scene& S, const ObjCache& oc, const Lane& L) {
    const Obj* ob = oc.at(S, L.obj);
    if (!(ob->flags & OBJ_XF)) return L.t_min;
    const RTX_GLOBAL Tri* Tg = (const RTX_GLOBAL Tri*)(S.blas + (size_t)(unsigned)L.btri * 128u);
    double T[9];
#pragma unroll
    for (int k = 0; k < 3; k++) { T[k] = Tg->a[k]; T[3 + k] = Tg->e1[k]; T[6 + k] = Tg->e2[k]; }
    double t, beta, gamma;
    tri_test(T, L.ro, L.rd, INFINITY, S.isect_eps, t, beta, gamma);
    const V3 a = mk(T[0], T[1], T[2]), e1 = mk(T[3], T[4], T[5]), e2 = mk(T[6], T[7], T[8]);
    const V3 point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
    double m[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = ob->m[k];
    const V3 wp = xf_point(m, point);
    const double dx = wp.x - L.o.x, dy = wp.y - L.o.y, dz = wp.z - L.o.z;
    return (double)sqrtf((float)(dx * dx + dy * dy + dz * dz));
}

// One service visit of a lane whose ray is in the search.  1: a job (or the walk's next pop) is
// set up; 0: the ray is complete; 3: the ray has to be done again by the plain replay.
#ifndef RTX_SEARCH_TALLY
#define RTX_SEARCH_TALLY 0          // 1 (diagnostic builds): event counts in GenCtrl::tally of the first control block
#endif
#define SEARCH_TALLY(k) do { if (RTX_SEARCH_TALLY) atomicAdd(S.limit_flag + (offsetof(GenCtrl, tally) - offsetof(GenCtrl, limit)) / 4 + (k), 1u); } while (0)
RTX_DEV int search_service(const Scene& S, const ObjCache& oc, Lane& L, const RecOut& out, glob_v4i* stk, const double light_dist) {
    int n = L.age >> 8;
    if ((L.age & 7) == SP_WALK) {
        bool over = true;
        if (L.obj != -1) {
            // an entered instance's job (bound t0) has ended
            double best = L.orig_t_min;
            SEARCH_TALLY(0);
            if (L.btri >= 0) {
                SEARCH_TALLY(1);
                if (n >= RTX_SEARCH_LIST) return 3 + 4 * 0;
                const int idx = L.obj - S.n_meshes;
                const double tj = L.t_min;
                const double wd = found_world_distance(S, oc, L);
                v4i e; e.x = __double2loint(tj); e.y = __double2hiint(tj); e.z = idx; e.w = L.btri;
                stk[RTX_STACK - 2 - n] = e;
                n++;
                if (wd < best || (wd == best && idx < L.whi)) { best = wd; L.whi = idx; L.sword = L.btri; }
                // A shadow ray asks whether the record ends up with a hit nearer than the light
                // (utils.cpp:1083).  An instance found under t0 with w_j below the light's distance
                // settles that if it is also found under the bound just above w_j: at its turn the
                // bound either is no greater than w_j already -- an accepted hit that near exists --
                // or lies in (w_j, t0], and the instance is found and accepted; from then on every
                // distance written to the record is smaller still.  No need for the nearest of all.
                if (L.is_shadow && wd < light_dist && tj <= wd) {
                    L.whi = idx; L.sword = L.btri; L.t_min = wd;
                    L.age = SP_WALK | (n << 8);
                    L.sp = 0;
                    nest_resume(L);
                    goto walk_over;
                }
            }
            L.t_min = best;
            L.age = SP_WALK | (n << 8);
            over = nest_resume(L) == 0;
        }
        if (!over) return 1;
    walk_over:
        // ---- the walk is over: L.whi is the k of the argument above (if any), L.t_min its w (else t0)
        L.sp = 0;
        SEARCH_TALLY(3);
        if (L.whi >= 0) {
            SEARCH_TALLY(4);
            L.age = SP_VERIFY | (n << 8);
            const double w = L.t_min;
            if (!search_enter(S, oc, L, L.whi, __longlong_as_double(__double_as_longlong(w) + 1ll))) return 3 + 4 * 1;
            return 1;
        }
        L.age = SP_PROBE | (n << 8);
        L.obj = -1;
    } else if ((L.age & 7) == SP_VERIFY) {
        if (L.btri != L.sword) return 3 + 4 * 2;
        const double w = L.orig_t_min;
        L.orig_t_min = L.dist;                    // any bound of (w, t0] decides alike: w < it
        job_finalize<false, 0, false>(S, oc, L, out);    // accepted: the record is k's, the bound its world distance
        if (!(L.t_min == w) || L.hit == 0) return 3 + 4 * 3;
        if (L.is_shadow && L.t_min < light_dist) return 0;       // occluded whatever is overwritten next (distances only shrink)
        L.age = SP_PROBE | (n << 8);
        L.obj = -1;
    } else if (L.obj != -1) {
        // SP_PROBE: the job of an instance after k, bound w, has ended
        SEARCH_TALLY(5);
        if (L.btri >= 0) {
            SEARCH_TALLY(6);
            const double w = L.orig_t_min;
            job_finalize<false, 0, false>(S, oc, L, out);    // found, not nearer: triangle fields overwritten, bound restored
            if (!(L.t_min == w)) return 3 + 4 * 4;
            return 0;
        }
        L.t_min = L.orig_t_min;
        L.obj = -1;
    }
    // ---- SP_PROBE: the listed instance of highest index after k (below the ones already tried) with t_j < w
    while (true) {
        int cand = -1;
        for (int i = 0; i < n; i++) {
            const v4i e = stk[RTX_STACK - 2 - i];
            if (e.z > L.whi && e.z < (int)L.sbits && e.z > cand && __hiloint2double(e.y, e.x) < L.t_min) cand = e.z;
        }
        if (cand < 0) break;
        L.sbits = (unsigned)cand;
        L.sp = 0;
        SEARCH_TALLY(7);
        if (search_enter(S, oc, L, cand, L.t_min)) return 1;
    }
    if (L.whi < 0) {
        // nothing came of the instances: the record's distance is the parked one
        const v4i h = stk[RTX_STACK - 1];
        L.dist = __hiloint2double(h.y, h.x);
    }
    return 0;
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

// Rays are appended to the wave's sub-queue: radiance rays from slot 0 upwards, shadow rays from
// slot sub_capacity-1 downwards, both counts in one 64-bit word.  A batch of hits makes ONE
// reservation for everything it emits (the counts are known before the rays are: a lane emits a
// shadow ray per light or none) -- the counter of a sub-queue is one address, and atomics on one
// address are served one after the other, a few tens of nanoseconds each, whoever asks.
struct Emit {
    unsigned rad, sh;       // first free slot of this batch's radiance / shadow rays (wave-uniform)
    bool ok;
};
// Must be reached by every lane of the wave.
RTX_DEV Emit emit_reserve(const Queue& q, GenCtrl* next, unsigned sub, unsigned n_rad, unsigned n_sh) {
    Emit e; e.rad = 0; e.sh = 0; e.ok = false;
    if (n_rad + n_sh == 0u) return e;
    unsigned long long old = 0;
    if (lane_id() == 0u) old = atomicAdd(&next->sub[sub].counts, ((unsigned long long)n_sh << 32) | (unsigned long long)n_rad);
    old = __shfl(old, 0);
    e.rad = (unsigned)old; e.sh = (unsigned)(old >> 32);
    // Reservations at both ends are totally ordered by the single atomic word, so
    // rad + n_rad + sh + n_sh <= sub_capacity means these slots are not, and will never be, claimed
    // from the other end.  Otherwise flag the overflow (the frame is redone with larger queues)
    // and drop the rays.
    if ((unsigned long long)e.rad + n_rad + e.sh + n_sh > q.sub_capacity) { if (lane_id() == 0u) next->overflow = 1u; return e; }
    e.ok = true;
    return e;
}
// Store one ray per requesting lane into reserved slots from `first` on; returns the number stored.
// Must be reached by every lane of the wave.
RTX_DEV unsigned emit(const Queue& q, unsigned sub, const Emit& e, unsigned first, bool shadow, bool want, V3 o, V3 d, V3 w, double aux,
                      unsigned pixel, unsigned meta) {
    const unsigned long long mask = __ballot(want);
    if (want && e.ok) {
        const unsigned k = first + (unsigned)__popcll(mask & ((1ull << lane_id()) - 1ull));
        const unsigned idx = sub * q.sub_capacity + (shadow ? q.sub_capacity - 1u - k : k);
        RTX_GLOBAL Ray* r = (RTX_GLOBAL Ray*)q.rays + idx;
        r->o[0] = o.x; r->o[1] = o.y; r->o[2] = o.z;
        r->d[0] = d.x; r->d[1] = d.y; r->d[2] = d.z;
        r->w[0] = w.x; r->w[1] = w.y; r->w[2] = w.z;
        r->aux = aux; r->pixel = pixel; r->meta = meta;
    }
    return (unsigned)__popcll(mask);
}

// The rays one generation received, as a list of work items: all radiance rays (sub-queue
// after sub-queue), then all shadow rays.  The prefix sums are wave-uniform.
struct GenItems {
    unsigned rad_end[RTX_NSUB], sh_end[RTX_NSUB];     // cumulative counts through sub-queue q
    unsigned n_rad, n_in, sub_capacity;
    unsigned grp_r, grp_s, mul_r, mul_s, shift;

    RTX_DEV void load(const GenCtrl* me, unsigned sub_cap) {
        sub_capacity = sub_cap;
        unsigned r_acc = 0, s_acc = 0;
#pragma unroll
        for (int q = 0; q < RTX_NSUB; q++) {
            // after an overflow the counters may exceed what was stored; clamp (the frame is redone)
            const unsigned long long c = me->sub[q].counts;
            const unsigned r = min((unsigned)c, sub_cap);
            const unsigned sdw = min((unsigned)(c >> 32), sub_cap - r);
            r_acc += r; s_acc += sdw;
            rad_end[q] = r_acc; sh_end[q] = s_acc;
        }
        n_rad = r_acc; n_in = r_acc + s_acc;
        // scattered hand-out order (see slot): multipliers coprime to the group counts
        // (a thin generation lasts as long as the wave that got most of its long rays, and rays
        // emitted together are long together: there the groups are single rays)
        shift = n_in < RTX_SCATTER_THIN_RAYS ? RTX_SCATTER_SHIFT_THIN : RTX_SCATTER_SHIFT;
        grp_r = n_rad >> shift; grp_s = (n_in - n_rad) >> shift;
        mul_r = coprime_near_golden(grp_r); mul_s = coprime_near_golden(grp_s);
    }
    RTX_DEV static unsigned coprime_near_golden(unsigned n) {
        if (n < 8u) return 1u;
        unsigned m = (unsigned)((unsigned long long)n * 2654435769ull >> 32) | 1u;       // n / golden ratio, odd
        while (true) {
            unsigned a = m, b = n;
            while (b != 0u) { const unsigned t = a % b; a = b; b = t; }
            if (a == 1u) return m % n;
            m += 2u;
        }
    }
    // queue slot of work item w < n_in
    // Work items are handed out in a scattered order: groups of 2^RTX_SCATTER_SHIFT consecutive
    // rays of a class are permuted (group g -> g * mul mod groups).  Rays emitted together are of
    // similar length; scattered, every wave gets its share of the long ones, and few enough of
    // them at the end to finish them wave-wide.
    RTX_DEV unsigned slot(unsigned w, bool& is_shadow) const {
        is_shadow = w >= n_rad;
        unsigned v = is_shadow ? w - n_rad : w;
        {
            const unsigned grp = is_shadow ? grp_s : grp_r, mul = is_shadow ? mul_s : mul_r;
            const unsigned g = v >> shift;
            if (g < grp) v = (unsigned)(((unsigned long long)g * mul) % grp) << shift | (v & ((1u << shift) - 1u));
        }
        unsigned q = 0, prev = 0;
#pragma unroll
        for (int k = 0; k + 1 < RTX_NSUB; k++) {
            const unsigned e = is_shadow ? sh_end[k] : rad_end[k];
            if (v >= e) { q = (unsigned)k + 1u; prev = e; }
        }
        const unsigned k = v - prev;
        return q * sub_capacity + (is_shadow ? sub_capacity - 1u - k : k);
    }
};

// A pixel's colour is the sum of the contributions of the rays of its tree (utils.cpp:868-1061),
// which arrive here in no fixed order.  They are accumulated as 64-bit fixed-point numbers (30
// fraction bits): integer addition does not depend on the order, so a frame is the same bytes
// whenever and however it is rendered -- whole, in tiles, on one GPU or on eight.  A contribution
// is rounded to 2^-30 (nine orders of magnitude below the 8-bit quantisation step) and limited to
// +-2^20 (the picture clamps at 255), which leaves room for 8192 such contributions per pixel.
#define RTX_ACCUM_SCALE 1073741824.0        // 2^30
RTX_DEV long long to_fixed(double c) {
    c = c > 1048576.0 ? 1048576.0 : (c < -1048576.0 ? -1048576.0 : c);      // (a NaN fails both tests and becomes 0 below)
    return c == c ? __double2ll_rn(c * RTX_ACCUM_SCALE) : 0ll;
}
RTX_DEV void add_color(double* accum, unsigned pixel, V3 c) {
    unsigned long long* acc = (unsigned long long*)accum + 3 * (size_t)pixel;
    const long long x = to_fixed(c.x), y = to_fixed(c.y), z = to_fixed(c.z);
    if (x != 0ll) atomicAdd(acc, (unsigned long long)x);
    if (y != 0ll) atomicAdd(acc + 1, (unsigned long long)y);
    if (z != 0ll) atomicAdd(acc + 2, (unsigned long long)z);
}

// Generation-0 work item -> pixel.  64 consecutive items form an 8x8 pixel block of one tile,
// so the rays of a wave are coherent.  Returns false for padding items outside the image.
RTX_DEV bool primary_pixel(const Frame& F, unsigned item, int& px, int& py, unsigned& local_pixel) {
    const int bpt_x = (F.tile_w + 7) >> 3, bpt_y = (F.tile_h + 7) >> 3;
    // (scattered in groups of 2^RTX_SCATTER0_SHIFT items: F.n_blocks counts those groups)
    const unsigned it2 = (unsigned)(((unsigned long long)(item >> RTX_SCATTER0_SHIFT) * F.block_mul) % F.n_blocks) << RTX_SCATTER0_SHIFT |
                         (item & ((1u << RTX_SCATTER0_SHIFT) - 1u));
    const unsigned blk = it2 >> 6, l = it2 & 63u;
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

// Divergent per-lane part of a service pass.  `fresh`: the lane has just been handed a new ray
// (L.slot / L.o / L.d / L.is_shadow set).
template <bool COUNT, bool FAST, int CAP, int NEST, bool B8>
RTX_DEV void service_lane(const Launch* __restrict__ Lp, const ObjCache& oc, const int gen, double* __restrict__ accum,
                          rtx_hit_id* __restrict__ aov, Lane& L, unsigned* inst_bits, glob_v4i* stk, const bool fresh, Tally& tl) {
    const Scene& S = Lp->S;
    const Frame& F = Lp->F;
    const Queue& qin = Lp->q[(gen + 1) & 1];
    const unsigned slot = L.slot;
    RecOut out;
    out.rec = (RTX_GLOBAL Hit*)Lp->hb.recs + slot; out.on = L.is_shadow == 0; out.ids = (gen == 0 && aov != nullptr);
    long long c0 = COUNT ? clock64() : 0;
    if (fresh) {
        if (COUNT) { tl.rays++; if (gen == 0) tl.primary++; else if (L.is_shadow) tl.shadow++; else tl.secondary++; }
        // consistent mode: a shadow ray looks no further than its light
        ray_begin<COUNT>(S, L, out, FAST && L.is_shadow ? oc.ctx[0] : DBL_MAX, tl);
        if (COUNT) { const long long c = clock64(); tl.sec[0] += (unsigned long long)(c - c0); tl.sec[4]++; c0 = c; }
    }
    // instances entered from the walk of the TLAS itself (traverse_steps, NEST): consistent mode, or
    // exact mode's search
    bool nested = NEST && L.it > S.n_meshes && !(NEST == 2 && (L.age & SP_REPLAY));
    int nest = -1;            // 0: the walk is over, 1: a job is set up, 2: a job inside an instance has ended
    int srch = -1;            // exact mode's search: what search_service decided
    if (NEST == 2 && nested && L.state == LS_JOB_DONE) {
        srch = search_service(S, oc, L, out, stk, L.is_shadow ? oc.ctx[0] : 0.0);
        L.state = LS_NEED_JOB;
        if ((srch & 3) == 3) {
            atomicAdd(S.limit_flag + (offsetof(GenCtrl, search_back) - offsetof(GenCtrl, limit)) / 4 + (srch >> 2), 1u);   // tally by cause
            // not covered by the search's argument: the ray again from the start, instances replayed in index order
            ray_begin<COUNT>(S, L, out, DBL_MAX, tl);
            L.age = SP_REPLAY;
            L.sp = 0;
            nested = false;
            srch = -1;
        }
    } else if (L.state == LS_JOB_DONE) {
        if (nested && L.obj == -1) {
            nest = 0;            // (instances are entered and, on a miss, left inside the traversal loop: the walk is over)
        } else if (L.obj == -1) {
            // end of the TLAS walk: L.word .. L.whi is the range of bitmap words it marked;
            // consumption starts in front of the first summary word that covers them
            L.sword = L.word == INT_MAX ? -1 : (L.word >> 5) - 1;
            L.whi = L.whi < 0 ? -1 : (L.whi >> 5);
            L.bits = 0u; L.sbits = 0u;
            L.obj = -2;
        } else if (FAST && L.is_shadow) {
            // consistent mode: any accepted triangle occludes; the bound is the light's distance again
            if (L.btri >= 0) L.hit = 1;
            L.t_min = oc.ctx[0];
        } else {
            // (the search's own rays are finalised by search_service; what arrives here with NEST = 2 is
            // a mesh job or a ray handed back to the index-order replay, whose L.whi is the last summary
            // word of its candidate bitmap: no nearest-first tie-break, which would overwrite it)
            job_finalize<FAST, NEST == 1 ? 1 : 0, B8>(S, oc, L, out);
        }
        if (nested && nest == -1) nest = 2;
        L.state = LS_NEED_JOB;
        if (COUNT) { const long long c = clock64(); tl.sec[1] += (unsigned long long)(c - c0); tl.sec[5]++; c0 = c; }
    }
    // A shadow ray with an accepted hit in front of its light is decided (see job_next) and is
    // retired here.  Instance candidates it has collected but not replayed are dropped: the
    // lane's candidate bitmap must be clear when the next ray starts.
    double light_dist = 0.0;
    if (!COUNT && L.is_shadow) light_dist = oc.ctx[0];
    const bool decided = !COUNT && L.is_shadow && L.hit != 0 && (FAST || L.t_min < light_dist);
    if (decided && L.it > S.n_meshes && !nested) {
        // (the current word was cleared when it was taken; what is still set in memory: the
        // words flagged in the current summary word, and the summary words not yet taken)
        const unsigned n_summary = rtx_summary_words(S.n_insts);
        unsigned sb = L.sbits;
        int sw = L.sword;
        while (true) {
            while (sb != 0u) {
                const int w = sw * 32 + (__ffs((int)sb) - 1);
                sb &= sb - 1u;
                __hip_atomic_store(inst_bits + n_summary + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (sw >= L.whi) break;
            sw++;
            sb = atomicExch(inst_bits + sw, 0u);
        }
        L.bits = 0u; L.sbits = 0u; L.sword = L.whi;
    }
    const int more = srch >= 0 ? srch : decided ? 0 : (nest == 0 || nest == 1) ? nest : nest == 2 ? nest_resume(L)
                             : job_next<COUNT, FAST, CAP, NEST, B8>(S, oc, L, out, inst_bits, stk, light_dist, tl);
    if (COUNT) { const long long c = clock64(); tl.sec[2] += (unsigned long long)(c - c0); tl.sec[6]++; c0 = c; }
    if (more == 1) { L.state = LS_IN_JOB; return; }
    if (more == 2) { L.state = LS_NEED_JOB; return; }          // yielded: more candidates next time
    // ---- ray complete ------------------------------------------------------------------------
    const bool hit = L.hit != 0;
    if (L.is_shadow) {                                            // utils.cpp:1083
        if (!(hit && (FAST || L.dist < oc.ctx[0])))
            add_color(accum, (unsigned)__double2loint(oc.ctx[4]), mk(oc.ctx[1], oc.ctx[2], oc.ctx[3]));
    } else {
        if (gen == 0 && aov != nullptr) {
            int px, py; unsigned pixel;
            primary_pixel(F, slot, px, py, pixel);
            rtx_hit_id h;
            h.kind = 0; h.container = -1; h.face = -1; h.src = -1; h.distance = 0.0;
            if (hit) {
                // read back what this lane stored (bypassing the vector L1)
                Hit* rec = Lp->hb.recs + slot;
                h.kind = __hip_atomic_load(&rec->kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h.container = __hip_atomic_load(&rec->container, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int face = __hip_atomic_load(&rec->face, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int src = __hip_atomic_load(&rec->src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                h.face = (h.kind == RTX_KIND_MESH || h.kind == RTX_KIND_TRIANGLE) ? face : -1;
                h.src = h.kind == RTX_KIND_MESH ? src : h.container;
                h.distance = L.dist;
            }
            aov[(size_t)py * F.width + px] = h;
        }
        out.rec->code = hit ? 1 : -1;
    }
    L.state = LS_EMPTY;
    if (COUNT) {
        tl.sec[3] += (unsigned long long)(clock64() - c0); tl.sec[7]++;
        // ray length histogram: < 16, < 32, ... < 1024, >= 1024 steps
        int b = 0;
        for (int a = L.age >> 4; a > 0 && b < 7; a >>= 1) b++;
        tl.hist[b]++;
        if (L.is_shadow) tl.hist_shadow[b]++;
        unsigned pix = slot;
        if (gen != 0) pix = ((const RTX_GLOBAL Ray*)qin.rays + slot)->pixel;
        const unsigned long long packed = ((unsigned long long)(unsigned)L.age << 32) | (unsigned long long)(pix & 0x7FFFFFFFu) |
                                          (L.is_shadow ? 0x80000000ull : 0ull);
        if (packed > tl.longest_ray) tl.longest_ray = packed;
    }
}

// The traversal loop: runs cooperative steps for the lanes that are inside a job until the
// wave needs a service pass (enough lanes wait for one) or no lane is traversing.
// `refillable`: idle lanes could be handed new rays, so they count as waiting for service.
//
// In one step every traversing lane advances by one 256-byte record: a lane at an inner node
// gets the collapsed node (the boxes of up to four grandchildren) and descends two binary
// levels, a lane in a leaf gets the leaf's next two triangles and tests them in order.  Both kinds of record arrive through the same cooperative fetch, so the step has
// a single exposed memory latency whatever the mix of lanes.
//
// Visit order and visited set of bvh.cpp:104-185: children ordered near/far by the sign of
// dir[splitAxis] (near first); a node is entered iff its box test passes with the bound current
// when the reference would pop it.  A child's box is evaluated at its (grand)parent; children
// that are not entered at once are pushed with their entry distance and re-checked against the
// (possibly smaller) bound when popped -- the predicate the reference evaluates at that
// moment.  Skipping the intermediate binary node's own test changes nothing: the slab
// arithmetic is monotone, so a box that passes lies inside a parent box that passes too, and
// the bound only shrinks between the parent's turn and the child's.
//
// Stack entries are 16 bytes (entry distance, ref, count) in a per-lane region of global
// memory (private "scratch" arrays interleave lanes dword by dword, which makes a 16-byte
// access at lane-varying depth touch four cache lines per lane).  The top entry is kept in
// registers (Lane::top); memory holds the entries below it, so the load a pop issues to refill
// the top is not needed before the next pop and overlaps the next record fetch.
// One candidate child of a collapsed node while the visit order is being established.
struct Cand { double e; int ref, cnt; bool ok; };
RTX_DEV void cand_swap(Cand& a, Cand& b, bool doit) {
    const Cand x = a, y = b;
    a.e = doit ? y.e : x.e; a.ref = doit ? y.ref : x.ref; a.cnt = doit ? y.cnt : x.cnt; a.ok = doit ? y.ok : x.ok;
    b.e = doit ? x.e : y.e; b.ref = doit ? x.ref : y.ref; b.cnt = doit ? x.cnt : y.cnt; b.ok = doit ? x.ok : y.ok;
}

// ---- the last jobs of a wave, traversed by the whole wave ----------------------------------
//
// When a generation has drained, what is left of it is a handful of long rays, each a chain of
// several hundred dependent steps in an otherwise idle wave; the kernel's duration is that
// chain.  Here the wave's 64 lanes work through the *frontier* of such jobs instead: every
// lane takes one open node, tests it, and puts the children that pass back (a LIFO frontier in
// the wave's LDS rows, so the order stays close to depth-first and the bounds tighten early).
//
// The result is the reference's.  bvh.cpp:104-185 returns the triangle with the smallest t
// below the incoming bound, ties going to the one it visits first (strict `t < t_min`).  A node
// is skipped only if its entry distance exceeds the current bound -- then every hit inside is
// farther than a hit already found -- so visiting *more* nodes than the reference cannot change
// the minimum, and ties are decided by the visiting order itself, carried along as a key: the
// rank of the frontier entry the path started from (the current node, then the stack from the
// top), then two bits per record level = the child's position in the reference's near-to-far
// order; within a leaf the triangle's position.  Every triangle is tested against the incoming
// bound and the minimum of (t, key, triangle) is taken at the end.
#ifndef RTX_WIDE_LANES
#define RTX_WIDE_LANES 24            // at most this many unfinished jobs in a drained wave: run them wave-wide
#endif
#ifndef RTX_WIDE_MAX_RAYS
#define RTX_WIDE_MAX_RAYS 512        // generations with at most this many rays use it
#endif
#ifndef RTX_WIDE_LANES_THIN
#define RTX_WIDE_LANES_THIN 16       // the two thresholds in generations below RTX_SCATTER_THIN_RAYS rays
#endif
#ifndef RTX_WIDE_AGE_THIN
#define RTX_WIDE_AGE_THIN 8
#endif
#ifndef RTX_WIDE_AGE
#define RTX_WIDE_AGE 8               // in a generation of size: only jobs that have taken this many steps already
#endif
#ifndef RTX_WIDE_CAP
#define RTX_WIDE_CAP 432             // frontier entries (32 bytes) in the wave's LDS rows, the jobs' slots behind them
#endif
#define RTX_WIDE_JOBS 16             // jobs traversed side by side (128-byte slots)
static_assert(RTX_WIDE_CAP * 32 + RTX_WIDE_JOBS * 128 <= 64 * RTX_ROW_DOUBLES * 8, "frontier and job slots share the wave's LDS rows");
#define RTX_WIDE_KEY_SHIFT 52        // first record level's two key bits; ten bits above them for the starting rank

RTX_DEV double bcast_d(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
RTX_DEV double wave_min_d(double v) {
    for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
RTX_DEV unsigned long long wave_min_u64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
RTX_DEV int wave_min_i(int v) {
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
RTX_DEV int wave_max_i(int v) {
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}

#ifndef RTX_WAVE_TALLY
#define RTX_WAVE_TALLY 0
#endif
#if RTX_WAVE_TALLY
    // (a clock read that the compiler does not move: the wave tally attributes time to sections)
#define RTX_TICK() ([]() __attribute__((always_inline)) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                         const long long t_ = (long long)__builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
                         __builtin_amdgcn_sched_barrier(0); return t_; }())
#endif
// Exclusive prefix sum and total over the wave of a number between 0 and 7 per lane: three ballots
// and their bit counts below the lane, instead of six rounds through the cross-lane network.
RTX_DEV void small_prefix(const int v, int& excl, int& total) {
    const unsigned long long b0 = __ballot((v & 1) != 0), b1 = __ballot((v & 2) != 0), b2 = __ballot((v & 4) != 0);
    const unsigned c0 = __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
    const unsigned c1 = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
    const unsigned c2 = __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u));
    excl = (int)(c0 + 2u * c1 + 4u * c2);
    total = __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
}

// What the wave needs of the owner's lane, and what it hands back.  Passed by value: the function is
// not inlined (inlined, its registers spoil the allocation of the traversal loop around the call).
struct WideIn { double rox, roy, roz, rdx, rdy, rdz, ix, iy, iz, t_min, stop; int par, shadow, sp, cur_ref, cur_cnt; v4i top; };
struct WideOut { double t_min; int btri, wmin, wmax; unsigned long long sel; };

// The jobs of the lanes in `m` (at most RTX_WIDE_JOBS, their stacks together within the frontier's
// room: the caller sees to both), traversed by the whole wave *side by side*: one frontier holds the
// open nodes of all of them, an entry names its job, and a lane takes whatever entry is its turn.  What
// a job needs lies in a 128-byte slot behind the frontier: the ray, the incoming bound, and the best hit
// so far, whose distance is the job's current bound.  Hits that improve a slot are rare and are written
// one lane after the other.  Each owner lane gets its own job's result.
__device__ __attribute__((noinline)) WideOut wide_multi(const Scene& S, const Wave wv, const WideIn L, glob_v4i* stk, unsigned* inst_bits,
                                                        const unsigned long long m_all) {
    const unsigned lane = wv.lane;
    // this call's jobs: as many of m_all as there are slots and as their stacks leave the frontier room
    unsigned long long m = 0ull;
    {
        int seeds = 0, n = 0;
        for (unsigned long long mm = m_all; mm != 0ull && n < RTX_WIDE_JOBS; mm &= mm - 1ull, n++) {
            const int src = __ffsll((long long)mm) - 1;
            const int need = __builtin_amdgcn_readlane(L.sp, src) + 1;
            if (n > 0 && seeds + need > RTX_WIDE_CAP - 104 - 3) break;
            m |= 1ull << src; seeds += need;
        }
    }
    const char* const blas = S.blas;
    const unsigned n_summary = rtx_summary_words(S.n_insts);
    const bool cull = S.cull != 0;
    const double isect_eps = S.isect_eps;
    const unsigned long long stk_a = (unsigned long long)(size_t)stk, bits_a = (unsigned long long)(size_t)inst_bits;

    // frontier: entry i = two 16-byte halves {enter, ref, cnt} {key, shift, job}
    lds_v4i* F = (lds_v4i*)wv.rows;
    // slot of a job, 16 doubles: slab origin and inverse direction (0-5), direction (6-8), incoming bound (9),
    // any-hit threshold (10), best t = current bound (11), best key (12), address of the owner's instance
    // bitmap (13), {flags, best triangle} (14), {lowest, highest bitmap word marked} (15)
    lds_double* T = wv.rows + RTX_WIDE_CAP * 4;
    const bool mine = ((m >> lane) & 1ull) != 0ull;
    const int my_slot = __popcll(m & ((1ull << lane) - 1ull));
    __builtin_amdgcn_wave_barrier();
    if (mine) {
        lds_double* t = T + 16 * my_slot;
        t[0] = L.rox; t[1] = L.roy; t[2] = L.roz; t[3] = L.ix; t[4] = L.iy; t[5] = L.iz;
        t[6] = L.rdx; t[7] = L.rdy; t[8] = L.rdz; t[9] = L.t_min; t[10] = L.stop; t[11] = L.t_min;
        t[12] = __hiloint2double(-1, -1);
        t[13] = __hiloint2double((int)(bits_a >> 32), (int)bits_a);
        t[14] = __hiloint2double(INT_MAX, L.par | ((L.shadow & 2) ? 256 : 0));
        t[15] = __hiloint2double(-1, INT_MAX);
    }
    // every job's stack from the bottom, its register-held top, and on top of those the node it is at
    int count = 0, n_slots = 0;
    for (unsigned long long mm = m; mm != 0ull; mm &= mm - 1ull, n_slots++) {
        const int src = __ffsll((long long)mm) - 1;
        const int o_sp = __builtin_amdgcn_readlane(L.sp, src);
        const glob_v4i* o_stk = (const glob_v4i*)(size_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(stk_a >> 32), src) << 32) |
                                                           (unsigned)__builtin_amdgcn_readlane((int)stk_a, src));
        for (int base = 0; base < o_sp - 1; base += 64) {
            const int i = base + (int)lane;
            if (i < o_sp - 1) {
                const v4i e = o_stk[i];
                const unsigned long long key = (unsigned long long)(unsigned)(o_sp - i) << (RTX_WIDE_KEY_SHIFT + 2);
                v4i k; k.x = (int)(unsigned)key; k.y = (int)(unsigned)(key >> 32); k.z = RTX_WIDE_KEY_SHIFT; k.w = n_slots;
                F[2 * (count + i)] = e; F[2 * (count + i) + 1] = k;
            }
        }
        v4i t;
        t.x = __builtin_amdgcn_readlane(L.top.x, src); t.y = __builtin_amdgcn_readlane(L.top.y, src);
        t.z = __builtin_amdgcn_readlane(L.top.z, src); t.w = __builtin_amdgcn_readlane(L.top.w, src);
        v4i c;
        c.x = __double2loint(-DBL_MAX); c.y = __double2hiint(-DBL_MAX);
        c.z = __builtin_amdgcn_readlane(L.cur_ref, src); c.w = __builtin_amdgcn_readlane(L.cur_cnt, src);
        if (lane == 0) {
            v4i k; k.z = RTX_WIDE_KEY_SHIFT; k.w = n_slots;
            if (o_sp > 0) {
                const unsigned long long key = 1ull << (RTX_WIDE_KEY_SHIFT + 2);
                k.x = (int)(unsigned)key; k.y = (int)(unsigned)(key >> 32);
                F[2 * (count + o_sp - 1)] = t; F[2 * (count + o_sp - 1) + 1] = k;
            }
            k.x = 0; k.y = 0;
            F[2 * (count + o_sp)] = c; F[2 * (count + o_sp) + 1] = k;
        }
        count += o_sp + 1;
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();

#if RTX_WIDE_TALLY
    unsigned w_steps = 0, w_active = 0, w_go = 0, w_seeds = (unsigned)count;
#endif
    while (count > 0) {
        // The top 64 entries, one per lane, as long as the frontier stays below its limit (see the append);
        // above it the wave descends depth-first from the top entry.
        int W = count <= RTX_WIDE_CAP - 104 - 3 ? 64 : 1;
        W = W > count ? count : W;
        const bool active = (int)lane < W;
        const int idx = count - 1 - (int)lane;
        count -= W;
        v4i ea = (v4i)(0), eb = (v4i)(0);
        if (active) { ea = F[2 * idx]; eb = F[2 * idx + 1]; }
        lds_double* const t = T + 16 * eb.w;
        const lds_v2d* const t2 = (const lds_v2d*)t;
        const v2d stop_b = t2[5];                         // {any-hit threshold, current bound}
        const int flags = __double2loint(t[14]);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
        const double B = stop_b.y;
        const bool go = active && B >= __hiloint2double(ea.y, ea.x) && !((flags & 256) && B < stop_b.x);
#if RTX_WIDE_TALLY
        w_steps++; w_active += (unsigned)W; w_go += (unsigned)__popcll(__ballot(go));
#endif
        const int ref = ea.z, cnt = ea.w;
        const unsigned long long key = ((unsigned long long)(unsigned)eb.y << 32) | (unsigned)eb.x;
        const int shift = eb.z;
        double rec[32];
        Slab sl; V3 rd = mk(0, 0, 1);
        sl.ox = sl.oy = sl.oz = sl.ix = sl.iy = sl.iz = 0.0; sl.par = flags & 255;
        if (go) {
            const RTX_GLOBAL v2d* gp = (const RTX_GLOBAL v2d*)(blas + ((size_t)(unsigned)ref << 7));
#pragma unroll
            for (int p = 0; p < 16; p++) { const v2d v = gp[p]; rec[2 * p] = v.x; rec[2 * p + 1] = v.y; }
            const v2d a = t2[0], b = t2[1], c = t2[2], d = t2[3];
            sl.ox = a.x; sl.oy = a.y; sl.oz = b.x; sl.ix = b.y; sl.iy = c.x; sl.iz = c.y;
            rd = mk(d.x, d.y, t[8]);
        }
        // What this lane puts back: up to four entries, farthest first.  (Slot k of c[] goes back if
        // c[k].ok, behind the slots above it: fixed places, no list built at run time, which would
        // live in memory instead of registers.)  leaf_rest: the rest of a leaf, the key stays.
        Cand c[4];
        c[0].ok = c[1].ok = c[2].ok = c[3].ok = false;
        c[0].e = c[1].e = c[2].e = c[3].e = 0.0; c[0].ref = c[1].ref = c[2].ref = c[3].ref = 0; c[0].cnt = c[1].cnt = c[2].cnt = c[3].cnt = 0;
        bool leaf_rest = false;
        bool hit = false; double ht = 0.0; int htri = 0;
        int wlo = INT_MAX, whi = -1;
        // (rays with an axis-parallel component are rare: the general box test only when an entry's job has one)
        const bool any_par = __ballot(go && (sl.par & RTX_PAR_AXES) != 0) != 0ull;
        if (go && cnt == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c[k].ref = k & 1 ? __double2hiint(rec[24 + (k >> 1)]) : __double2loint(rec[24 + (k >> 1)]);
                c[k].cnt = k & 1 ? __double2hiint(rec[26 + (k >> 1)]) : __double2loint(rec[26 + (k >> 1)]);
                double e;
                bool hitb;
                if (any_par) hitb = box_enter(rec + 6 * k, rec + 6 * k + 3, sl, e);
                else hitb = box_enter_nopar(rec + 6 * k, rec + 6 * k + 3, sl, e);
                c[k].e = e;
                const bool pass = hitb & (B >= e);
                c[k].ok = pass & (c[k].cnt >= 0);
                if (pass && c[k].cnt == -2) {            // TLAS walk: candidate instance
                    const double ba = t[13];
                    unsigned* o_bits = (unsigned*)(size_t)(((unsigned long long)(unsigned)__double2hiint(ba) << 32) | (unsigned)__double2loint(ba));
                    const int w = c[k].ref >> 5;
                    atomicOr(o_bits + n_summary + w, 1u << (c[k].ref & 31));
                    atomicOr(o_bits + (w >> 5), 1u << (w & 31));
                    wlo = min(wlo, w); whi = max(whi, w);
                }
            }
            // (the job's direction signs ride in its `par` word, see make_slab)
            const int axis0 = __double2loint(rec[28]), axisL = __double2hiint(rec[28]), axisR = __double2loint(rec[29]);
            const unsigned far_first = (unsigned)sl.par >> 4;
            cand_swap(c[0], c[1], ((far_first >> axisL) & 1u) && c[1].cnt >= 0);
            cand_swap(c[2], c[3], ((far_first >> axisR) & 1u) && c[3].cnt >= 0);
            const bool right_first = ((far_first >> axis0) & 1u) != 0u;
            cand_swap(c[0], c[2], right_first);
            cand_swap(c[1], c[3], right_first);
        } else if (go && cnt > 0) {
            const double* td = rec;
            const double* tq = rec + 16;
            const V3 ro = mk(sl.ox, sl.oy, sl.oz);
            const double b_init = t[9];
            const bool two = cnt >= 2;
            bool culled0 = false, culled1 = false;
            if (cull) {
                culled0 = rd.x * td[10] + rd.y * td[11] + rd.z * td[12] > 0.0;
                culled1 = rd.x * tq[10] + rd.y * tq[11] + rd.z * tq[12] > 0.0;
            }
            // every triangle is tested against the incoming bound, so that a tie with the best so far is seen
            double tt, b, g2;
            if (!culled0 && tri_test(td, ro, rd, b_init, isect_eps, tt, b, g2)) { hit = true; ht = tt; htri = ref; }
            if (two && !culled1 && tri_test(tq, ro, rd, b_init, isect_eps, tt, b, g2)) {
                if (!hit || tt < ht) { ht = tt; htri = ref + 1; }
                hit = true;
            }
            hit = hit && ht <= B;
            if (cnt > 2) {
                // the rest of a long leaf (a mesh without a BVH is one leaf): in two halves, so that
                // the lanes working on it double with every round
                const int rest = cnt - 2;
                const int h = rest > 2 ? ((rest >> 1) + 1) & ~1 : rest;        // an even number of triangles first
                leaf_rest = true;
                c[3].e = c[2].e = __hiloint2double(ea.y, ea.x);
                c[3].ref = ref + 2; c[3].cnt = h; c[3].ok = true;
                if (rest > h) { c[2].ref = ref + 2 + h; c[2].cnt = rest - h; c[2].ok = true; }
            }
        }
        // the slots' best hits and marked ranges: one lane after the other
        for (unsigned long long mc = __ballot(hit || whi >= 0); mc != 0ull; mc &= mc - 1ull) {
            if ((int)lane == __ffsll((long long)mc) - 1) {
                if (hit) {
                    const double bt = t[11], bk_d = t[12];
                    const unsigned long long bk = ((unsigned long long)(unsigned)__double2hiint(bk_d) << 32) | (unsigned)__double2loint(bk_d);
                    const int btri = __double2hiint(t[14]);
                    if (ht < bt || (ht == bt && (key < bk || (key == bk && htri < btri)))) {
                        t[11] = ht; t[12] = __hiloint2double((int)(unsigned)(key >> 32), (int)(unsigned)key);
                        t[14] = __hiloint2double(htri, flags);
                    }
                }
                if (whi >= 0) {
                    const double r = t[15];
                    t[15] = __hiloint2double(max(__double2hiint(r), whi), min(__double2loint(r), wlo));
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
        // append: exclusive prefix of the push counts over the lanes
        int npush = (int)c[0].ok + (int)c[1].ok + (int)c[2].ok + (int)c[3].ok;
        int excl, total;
        small_prefix(npush, excl, total);
        if (W > 1 && count + total > RTX_WIDE_CAP - 104) {
            // More children than the frontier may hold: the leading lanes whose children fit put them
            // back, the others their own entry, to be taken again (what they found meanwhile stands: a
            // best hit and a mark are the same the second time).  Lane 0 always fits, by the choice of W.
            const int room = RTX_WIDE_CAP - 104 - count;
            const unsigned long long fm = __ballot(active && excl + npush + (W - 1 - (int)lane) <= room);
            const int p = __ffsll((long long)~fm) - 1;
            if ((int)lane >= p) {
                c[0].ok = c[1].ok = c[2].ok = false;
                c[3].ok = go; c[3].e = __hiloint2double(ea.y, ea.x); c[3].ref = ref; c[3].cnt = cnt; leaf_rest = true;
                npush = go ? 1 : 0;
            }
            small_prefix(npush, excl, total);
        }
        int at = count + excl;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            if (c[k].ok && at < RTX_WIDE_CAP) {
                v4i a; a.x = __double2loint(c[k].e); a.y = __double2hiint(c[k].e); a.z = c[k].ref; a.w = c[k].cnt;
                unsigned long long k2 = key; int sh = shift;
                if (!leaf_rest) { k2 = key | ((unsigned long long)(unsigned)k << (shift > 0 ? shift : 0)); sh = shift - 2; }
                v4i bb; bb.x = (int)(unsigned)k2; bb.y = (int)(unsigned)(k2 >> 32); bb.z = sh; bb.w = eb.w;
                F[2 * at] = a; F[2 * at + 1] = bb;
                at++;
            }
        }
        count += total;
        // Cannot happen (should it ever, the frame fails rather than show a picture with something
        // missing): with all lanes at work the frontier ends a step
        // at RTX_WIDE_CAP - 104 entries or below; once W is 1 the wave descends depth-first from the top
        // entry, which adds at most three entries per record level (at most 26 levels, Launch::wide_ok)
        // plus one per halving of a long leaf (< 26): fewer than the 104 entries held back.
        if (count > RTX_WIDE_CAP) { if (lane == 0) *S.limit_flag = 1u; count = RTX_WIDE_CAP; }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
    }
#if RTX_WIDE_TALLY
    if (lane == 0) {      // diagnostic builds: calls, jobs, steps, lanes with an entry, lanes that fetched, entries at the start
        unsigned* tw = S.limit_flag + (offsetof(GenCtrl, search_back) - offsetof(GenCtrl, limit)) / 4;
        atomicAdd(tw, 1u); atomicAdd(tw + 1, (unsigned)n_slots); atomicAdd(tw + 2, w_steps); atomicAdd(tw + 3, w_active);
        atomicAdd(tw + 4, w_go); atomicAdd(tw + 5, w_seeds);
    }
#endif
    // every owner's result: smallest t, then first in the reference's visiting order
    WideOut out;
    out.t_min = INFINITY; out.btri = -1; out.wmin = INT_MAX; out.wmax = -1; out.sel = m;
    if (mine) {
        const lds_double* t = T + 16 * my_slot;
        const double bk_d = t[12], tr = t[14], r = t[15];
        if ((__double2hiint(bk_d) & __double2loint(bk_d)) != -1) { out.t_min = t[11]; out.btri = __double2hiint(tr); }
        out.wmin = __double2loint(r); out.wmax = __double2hiint(r);
    }
    __builtin_amdgcn_wave_barrier();
    return out;
}

// Run the jobs of the lanes in `m` wave-wide, as many at a time as wide_multi takes.
RTX_DEV void wide_jobs(const Scene& S, const Wave& wv, Lane& L, glob_v4i* stk, unsigned* inst_bits,
                       unsigned long long m) {
    WideIn in;
    in.rox = L.ro.x; in.roy = L.ro.y; in.roz = L.ro.z; in.rdx = L.rd.x; in.rdy = L.rd.y; in.rdz = L.rd.z;
    in.ix = L.ix; in.iy = L.iy; in.iz = L.iz; in.t_min = L.t_min; in.stop = L.orig_t_min;
    in.par = L.par; in.shadow = L.is_shadow; in.sp = L.sp; in.cur_ref = L.cur_ref; in.cur_cnt = L.cur_cnt; in.top = L.top;
    while (m != 0ull) {
        const WideOut out = wide_multi(S, wv, in, stk, inst_bits, m);
        const unsigned long long sel = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(out.sel >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)out.sel);
        m &= ~sel;
        if ((sel >> wv.lane) & 1ull) {
            if (out.t_min < INFINITY) { L.t_min = out.t_min; L.btri = out.btri; }
            L.word = min(L.word, out.wmin); L.whi = max(L.whi, out.wmax);
            L.sp = 0;
            L.state = LS_JOB_DONE;
        }
    }
}

// NEST (1: consistent mode, 2: exact mode's search): instances are not collected for a replay in
// index order; the TLAS walk enters them itself, nearest first.  A TLAS slot that holds an
// instance is a child like any other; when it becomes the lane's current entry (cur_cnt == -2)
// the lane's record of that step is the set-up half of the instance's object record: the ray goes
// to object space and the traversal of the instance's BLAS starts on top of the walk's stack
// entries (Lane::word = the stack depth the job started at; pops stop there).  A job that ends
// without a hit takes the ray back to world space and asks for the walk's next entry
// (cur_cnt == -1: pop, nothing to fetch); one that found something ends in a service pass.  The
// walk prunes with the bound as it shrinks, so most instances behind the first hit are never touched.
template <bool COUNT, int WIDE, int NEST>
RTX_DEV unsigned long long traverse_steps(const Scene& S, const Wave& wv, Lane& L, glob_v4i* __restrict__ stk, unsigned* inst_bits,
                            const int service_threshold, const bool refillable, const int wide_lanes, const int wide_age, Tally& tl,
                            unsigned& n_records) {
    const unsigned lane = wv.lane;
    const unsigned n_summary = rtx_summary_words(S.n_insts);
    const char* const blas = S.blas;
    const bool cull = S.cull != 0;
    const double isect_eps = S.isect_eps;
    // (the slab constants of the lane's current job are the lane's own fields: no second copy to keep in step)
#define RTX_SL (Slab{L.ro.x, L.ro.y, L.ro.z, L.ix, L.iy, L.iz, L.par})

    while (true) {
        const bool in_job = L.state == LS_IN_JOB;
        const bool at_leaf = in_job && L.cur_cnt > 0;
        const unsigned long long m_job = __ballot(in_job);
        if (m_job == 0ull) break;
        const unsigned long long m_wait = __ballot(L.state == LS_NEED_JOB || L.state == LS_JOB_DONE ||
                                                   (refillable && L.state == LS_EMPTY));
        if (__popcll(m_wait) >= service_threshold) break;
        // the generation has drained and this wave holds its last few jobs: the caller finishes them wave-wide
        if (!COUNT && WIDE != 0 && !refillable && __popcll(m_job) <= wide_lanes) {
            const unsigned long long m_old = WIDE == 2 ? m_job : __ballot(in_job && L.age >= wide_age);
            if (m_old != 0ull) return m_old;
        }
        if (COUNT) {
            const int nl = __popcll(__ballot(at_leaf)), nj = __popcll(m_job);
            if (lane == 0) {
                if (nj > nl) { tl.inner_steps++; tl.inner_lanes += (unsigned)(nj - nl); }
                if (nl) { tl.leaf_steps++; tl.leaf_lanes += (unsigned)nl; }
            }
        }
        // rays with an axis-parallel component are rare: the wave takes the general box test
        // only when one of its traversing lanes has one
        const bool any_par = __ballot(in_job && (L.par & RTX_PAR_AXES) != 0) != 0ull;
        // ... and looks for instance slots only when one of them walks the TLAS (object -1)
        const bool any_walk = __ballot(in_job && L.obj == -1) != 0ull;

        // The lane's 256-byte record: a collapsed node, or two consecutive triangle records.
        // A wave with only a lane or two still traversing is on the frame's critical path (the
        // longest rays of a generation); it loads the records directly, one memory round trip
        // without the detour through LDS.  Otherwise the wave fetches cooperatively.
        const long long f0 = COUNT ? clock64() : 0;
#if RTX_WAVE_TALLY == 3
        const long long k_f0 = RTX_TICK();
#endif
        double rec[32];
        // NEST: setting an instance up (transform, reciprocals, two box tests) is some three hundred vector
        // instructions that the whole wave issues whenever ONE lane stands at an instance slot -- nearly every
        // step in a scene of thousands of instances.  Such lanes wait (they fetch nothing and do nothing this
        // step) until RTX_INST_QUORUM of them have gathered or no other lane has anything to do.
        // The same for the other kinds of step in these kernels (the ones that walk a TLAS of many instances; in
        // the plain mesh kernels holding lanes back costs more than the sections it saves): the wave runs a section
        // (four box tests for the lanes at nodes, two triangle tests for the lanes at leaves) for the most numerous
        // kind in any case, and for another kind only once enough of its lanes have gathered.
        bool held = false;
        if (!COUNT && NEST && (RTX_INST_QUORUM > 1 || RTX_LEAF_QUORUM > 1 || RTX_NODE_QUORUM > 1)) {
            const bool k_inst = in_job && L.cur_cnt == -2, k_node = in_job && L.cur_cnt == 0;
            const int n_inst = __popcll(__ballot(k_inst)), n_node = __popcll(__ballot(k_node)), n_leaf = __popcll(__ballot(at_leaf));
            const int top = max(n_inst, max(n_node, n_leaf));
            const bool run_inst = n_inst >= RTX_INST_QUORUM || n_inst == top;
            const bool run_node = n_node >= RTX_NODE_QUORUM || n_node == top;
            const bool run_leaf = n_leaf >= RTX_LEAF_QUORUM || n_leaf == top;
            held = (k_inst && !run_inst) || (k_node && !run_node) || (at_leaf && !run_leaf);
        }
        const bool act = in_job && !held;
        const bool wants = act && !(NEST && L.cur_cnt == -1);      // (a lane that only asks for a pop fetches nothing)
        // NEST: a lane whose entry is an instance fetches the set-up half of that instance's object record
        const int key = (NEST && L.cur_cnt == -2) ? S.obj_key0 + 4 * (S.n_meshes + L.cur_ref) : L.cur_ref;
#ifndef RTX_NO_RECORD_COUNT
        n_records += (unsigned)__popcll(__ballot(wants));          // (wave-uniform: scalar arithmetic)
#endif
        if (__popcll(m_job) <= RTX_SPARSE_LANES) {
            if (wants) {
                const RTX_GLOBAL v2d* gp = (const RTX_GLOBAL v2d*)(blas + ((size_t)(unsigned)key << 7));
#pragma unroll
                for (int p = 0; p < 16; p++) { const v2d v = gp[p]; rec[2 * p] = v.x; rec[2 * p + 1] = v.y; }
            }
        } else {
            const unsigned row = coop_fetch(wv, wants, key, blas);
            // (every lane reads its row, whether it asked for a record or not: a read under the
            // lanes' own condition makes the compiler keep the other lanes' thirty-two registers
            // alive through it, sixty register moves a step)
            row_read<0, 16>(wv, row, rec);
            __builtin_amdgcn_wave_barrier();
        }
        if (COUNT) tl.clk_fetch += (unsigned long long)(clock64() - f0);
#if RTX_WAVE_TALLY == 3
        const long long k_f1 = RTX_TICK();
        tl.clk_fetch += (unsigned long long)(k_f1 - k_f0);
#endif

        bool need_pop = false;
        // the walk of the TLAS in NEST mode compares with a bound a hair above the current one: an
        // instance whose hit ties with the current one (world distances are rounded to float) must
        // still be visited, because the lower instance index wins a tie whatever the visiting order
        // (a ray that exact mode's search has handed back to the plain replay walks the TLAS the plain way)
        const bool nl = NEST == 1 || (NEST == 2 && !(L.age & SP_REPLAY));
        double t_cmp = 0.0;
        if (NEST) t_cmp = (nl && L.obj == -1) ? L.t_min * 1.000001 : L.t_min;
        if (act) {
            if (NEST != 2) L.age++;       // (the search keeps its phase there)
            if (NEST && L.cur_cnt == -1) {
                need_pop = true;
            } else if (NEST && L.cur_cnt == -2) {
                // ---- an instance met by the walk: rayHitsMesh's prologue (utils.cpp:418-494) and the
                // root test of bvh.cpp:132, from the record just fetched; the traversal of its BLAS
                // starts on top of the walk's stack entries
                const int oi = S.n_meshes + L.cur_ref;
                const int root_ref = __double2loint(rec[28]), root_count = __double2hiint(rec[28]);
                const int flags = __double2loint(rec[29]);
                const bool shadow = L.is_shadow != 0;
                bool enter_it = !(flags & OBJ_SKIP);
                // exact mode's search runs the instance under the bound the instance loop starts with
                // (parked in Lane::dist), world-box test included; consistent mode under the current bound
                double bound = NEST == 2 ? L.dist : L.t_min;
                if (NEST == 2 && enter_it && (flags & OBJ_XF) && (flags & OBJ_BOUNDS)) {
                    double e;
                    enter_it = box_enter_svc(rec + 16, rec + 19, RTX_SL, e) && (bound >= e);
                }
                V3 ro = L.o, rd = L.d;
                if (enter_it && (flags & OBJ_XF)) {
                    ro = xf_point(rec, L.o);
                    const V3 od = xf_dir(rec, L.d);
                    rd = normalize_f32(od);
                    if (NEST == 1) {
                        // the bound goes to object space with the ray; a radiance ray's gets a hair of
                        // slack so that a hit tying with the current one is found (job_finalize)
                        bound = L.t_min / (1.0 / (double)sqrtf((float)(od.x * od.x + od.y * od.y + od.z * od.z)));
                        if (!shadow) bound *= 1.000001;
                    }
                } else if (NEST == 1 && !shadow && bound < DBL_MAX) {
                    bound = __longlong_as_double(__double_as_longlong(bound) + 1ll);
                }
                if (enter_it) {
                    const Slab os = (flags & OBJ_XF) ? make_slab(ro, rd) : RTX_SL;
                    double e;
                    enter_it = box_enter_svc(rec + 22, rec + 25, os, e) && (bound >= e);
                    if (enter_it) {
                        L.obj = oi;
                        L.word = L.sp;
                        L.orig_t_min = L.t_min;
                        L.t_min = bound;
                        L.is_shadow &= 1;
                        if (NEST == 1) { L.age = 0; if (shadow) { L.is_shadow = 3; L.orig_t_min = bound; } }
                        L.ro = ro; L.rd = rd; L.ix = os.ix; L.iy = os.iy; L.iz = os.iz; L.par = os.par;
                        L.cur_ref = root_ref; L.cur_cnt = root_count; L.btri = -1;
                    }
                }
                need_pop = !enter_it;
            } else if (!at_leaf) {
                // ---- collapsed inner node: up to four grandchildren of a binary node ------------
                if (COUNT) { if (L.obj == -1) tl.tlas_visits++; else tl.node_visits++; }
                const double* nd = rec;
                Cand c[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    c[k].ref = k & 1 ? __double2hiint(nd[24 + (k >> 1)]) : __double2loint(nd[24 + (k >> 1)]);
                    c[k].cnt = k & 1 ? __double2hiint(nd[26 + (k >> 1)]) : __double2loint(nd[26 + (k >> 1)]);
                    double e;
                    bool hit;
                    if (any_par) hit = box_enter(nd + 6 * k, nd + 6 * k + 3, RTX_SL, e);
                    else hit = box_enter_nopar(nd + 6 * k, nd + 6 * k + 3, RTX_SL, e);
                    c[k].e = e;
                    const bool pass = hit & ((NEST ? t_cmp : L.t_min) >= e);
                    c[k].ok = pass & (c[k].cnt >= 0 || (NEST && nl && c[k].cnt == -2));
                    if (any_walk && (!NEST || !nl) && pass && c[k].cnt == -2) {
                        // TLAS walk: the slot is a mesh instance whose world box the ray can meet
                        const int w = c[k].ref >> 5;
                        atomicOr(inst_bits + n_summary + w, 1u << (c[k].ref & 31));
                        atomicOr(inst_bits + (w >> 5), 1u << (w & 31));
                        L.word = min(L.word, w); L.whi = max(L.whi, w);
                    }
                }
                // Visit order of the reference: at each binary split the near child first, near =
                // left iff dir[splitAxis] >= 0 (bvh.cpp:139-147).  Slots 0,1 hang under the left
                // child (split axisL), slots 2,3 under the right child (axisR), the two pairs
                // under the node itself (axis0).
                // (the direction's sign per axis is kept in the lane's `par` word: a shift and a mask per split)
                const int axis0 = __double2loint(nd[28]), axisL = __double2hiint(nd[28]), axisR = __double2loint(nd[29]);
                const unsigned far_first = (unsigned)L.par >> 4;
                // a leaf child occupies the first slot of its pair: nothing to order there
                cand_swap(c[0], c[1], ((far_first >> axisL) & 1u) && (NEST ? c[1].cnt != -1 : c[1].cnt >= 0));
                cand_swap(c[2], c[3], ((far_first >> axisR) & 1u) && (NEST ? c[3].cnt != -1 : c[3].cnt >= 0));
                const bool right_first = ((far_first >> axis0) & 1u) != 0u;
                cand_swap(c[0], c[2], right_first);
                cand_swap(c[1], c[3], right_first);
                // c[0..3] is now near -> far.  The nearest passing child is entered; the others
                // are pushed, farthest first, each with its entry distance for the re-check at pop.
                const bool any0 = c[0].ok, any1 = any0 | c[1].ok, any2 = any1 | c[2].ok;
                // Room for the three entries a step can push is checked once.  (The host accepts a tree
                // only if three entries per record level fit, so a lane is never short of room; if it
                // is after all, the frame is reported as failed instead of silently losing a subtree.)
                const bool room = L.sp + 3 <= (NEST == 2 ? RTX_SEARCH_STACK : RTX_STACK);
                bool lost = false;
#pragma unroll
                for (int k = 3; k >= 1; k--) {
                    const bool earlier = k == 3 ? any2 : (k == 2 ? any1 : any0);
                    const bool want = c[k].ok && earlier;
                    lost |= want && !room;
                    if (want && room) {
                        if (L.sp > 0) stk[L.sp - 1] = L.top;
                        L.top.x = __double2loint(c[k].e); L.top.y = __double2hiint(c[k].e); L.top.z = c[k].ref; L.top.w = c[k].cnt;
                        L.sp++;
                    }
                }
                if (lost) *S.limit_flag = 1u;
                if (any2 | c[3].ok) {
                    const int sel = any0 ? 0 : (c[1].ok ? 1 : (c[2].ok ? 2 : 3));
                    L.cur_ref = sel == 0 ? c[0].ref : (sel == 1 ? c[1].ref : (sel == 2 ? c[2].ref : c[3].ref));
                    L.cur_cnt = sel == 0 ? c[0].cnt : (sel == 1 ? c[1].cnt : (sel == 2 ? c[2].cnt : c[3].cnt));
                } else {
                    need_pop = true;
                }
            } else {
                // ---- leaf: its next two triangles, in leaf order (bvh.cpp:150-170) ----------------
                const int ti = L.cur_ref;
                const bool two = L.cur_cnt >= 2;
                if (COUNT) tl.tri_tests += two ? 2u : 1u;
                const double* td = rec;
                const double* tq = rec + 16;
                bool culled0 = false, culled1 = false;
                if (cull) {                                                                   // utils.cpp:322-326
                    culled0 = L.rd.x * td[10] + L.rd.y * td[11] + L.rd.z * td[12] > 0.0;
                    culled1 = L.rd.x * tq[10] + L.rd.y * tq[11] + L.rd.z * tq[12] > 0.0;
                }
                double t, b, g2;
                if (!culled0 && tri_test(td, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.t_min = t; L.btri = ti; }
                if (two && !culled1 && tri_test(tq, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.t_min = t; L.btri = ti + 1; }
                L.cur_ref += 2; L.cur_cnt -= 2;
                need_pop = L.cur_cnt <= 0;
                if (!COUNT && (L.is_shadow & 2) && L.btri >= 0 && L.t_min < L.orig_t_min) {
                    need_pop = true; L.sp = 0;         // occluder found: the job ends here
                }
                if (!COUNT && !NEST && (L.is_shadow & 4) && L.btri >= 0 &&
                    L.t_min < (double)__int_as_float(__double2hiint(wv.ctx[4]))) {
                    need_pop = true; L.sp = 0;         // the same inside a transformed object (job_next)
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const long long p0 = COUNT ? clock64() : 0;
#if RTX_WAVE_TALLY == 3
        const long long k_p0 = RTX_TICK();
#endif
        int base = 0;                  // entries below belong to the suspended walk
        double t_pop = 0.0;            // (the leaf tests may have pulled the bound in)
        if (NEST) { base = (nl && L.obj != -1) ? L.word : 0; t_pop = (nl && L.obj == -1) ? L.t_min * 1.000001 : L.t_min; }
        if (NEST && need_pop && nl && L.obj != -1 && L.sp <= base && L.btri < 0 && L.it > S.n_meshes &&
            (NEST == 1 || (L.age & 7) == SP_WALK)) {
            // the job inside an instance has ended without a hit: the ray goes back to world space
            // and asks for the walk's next entry -- nothing a service pass would have to do
            L.t_min = (NEST == 1 && L.is_shadow) ? L.dist : L.orig_t_min;
            L.obj = -1; L.word = 0;
            L.ro = L.o; L.rd = L.d;
            const Slab ws = make_slab(L.o, L.d);
            L.ix = ws.ix; L.iy = ws.iy; L.iz = ws.iz; L.par = ws.par;
            L.is_shadow &= 1;
            L.cur_ref = 0; L.cur_cnt = -1;
            need_pop = L.sp <= 0;            // (with entries left the pop is next step's; without, the walk is over)
            base = 0;
        }
        if (need_pop) {
            L.state = LS_JOB_DONE;
            if (L.sp > (NEST ? base : 0)) {
                // First pop peeled off the loop: it consumes only registers, and the refill load
                // it issues stays in flight (inside the loop every iteration has to wait for the
                // previous iteration's refill).  Everything that reads the old top comes before
                // the load, so the load can target the registers that hold it.
                int pass = (NEST ? t_pop : L.t_min) >= __hiloint2double(L.top.y, L.top.x) ? 1 : 0;
                if (pass) { L.cur_ref = L.top.z; L.cur_cnt = L.top.w; L.state = LS_IN_JOB; }
                asm volatile("" : "+v"(pass), "+v"(L.cur_ref), "+v"(L.cur_cnt), "+v"(L.state));   // old top is dead from here
                --L.sp;
                L.top = stk[L.sp > 0 ? L.sp - 1 : 0];          // (entry 0 is always readable)
                if (!pass) {
                    // The entry failed its re-check.  After a hit has pulled the bound in, most of
                    // what is left on the stack fails too, one entry after the other: the entries
                    // are examined four at a time (adjacent in memory, one round trip; the first
                    // of them is the refill already in flight), from the top down.
                    while (L.sp > (NEST ? base : 0)) {
                        const int avail = NEST ? L.sp - base : L.sp;
                        v4i e1 = stk[L.sp >= 2 ? L.sp - 2 : 0];
                        v4i e2 = stk[L.sp >= 3 ? L.sp - 3 : 0];
                        v4i e3 = stk[L.sp >= 4 ? L.sp - 4 : 0];
                        // (each entry stays the one 16-byte load it is: left to itself the compiler
                        // fetches the distance halves a second time, one dependent round trip each)
                        asm volatile("" : "+v"(e1), "+v"(e2), "+v"(e3));
                        const bool p0 = (NEST ? t_pop : L.t_min) >= __hiloint2double(L.top.y, L.top.x);
                        const bool p1 = avail >= 2 && (NEST ? t_pop : L.t_min) >= __hiloint2double(e1.y, e1.x);
                        const bool p2 = avail >= 3 && (NEST ? t_pop : L.t_min) >= __hiloint2double(e2.y, e2.x);
                        const bool p3 = avail >= 4 && (NEST ? t_pop : L.t_min) >= __hiloint2double(e3.y, e3.x);
                        const int k = p0 ? 0 : (p1 ? 1 : (p2 ? 2 : (p3 ? 3 : 4)));     // first that passes; 4: none
                        if (k < 4) {
                            L.cur_ref = k == 0 ? L.top.z : (k == 1 ? e1.z : (k == 2 ? e2.z : e3.z));
                            L.cur_cnt = k == 0 ? L.top.w : (k == 1 ? e1.w : (k == 2 ? e2.w : e3.w));
                            L.state = LS_IN_JOB;
                        }
                        const int used = k < 4 ? k + 1 : 4;
                        L.sp = avail > used ? L.sp - used : (NEST ? base : 0);
                        if (L.sp > 0) {                   // the new top: the entry below the last one used
                            if (k == 0) L.top = e1;
                            else if (k == 1) L.top = e2;
                            else if (k == 2) L.top = e3;
                            else L.top = stk[L.sp - 1];
                        }
                        if (k < 4) break;
                    }
                }
            }
        }
        if (COUNT) tl.clk_pop += (unsigned long long)(clock64() - p0);
#if RTX_WAVE_TALLY == 3
        { const long long k_e = RTX_TICK(); tl.clk_pop += (unsigned long long)(k_e - k_p0); tl.clk_trav += (unsigned long long)(k_p0 - k_f1); tl.passes++; }
#endif
    }
    return 0ull;
}


// ---- the eight-wide BLAS (Node8, rtx_device_types.h) ------------------------------------------
//
// The same jobs as traverse_steps -- (object-space ray, bound) -> nearest triangle below the bound,
// the first in the reference's visiting order among equals -- over records that hold three levels of
// the reference's tree with float boxes rounded outward.  What makes the step lean:
//   * a box test is conservative, never exact: float arithmetic with an explicit error margin.  It
//     may let a ray into a box the reference's FP64 test keeps it out of, never the other way round.
//   * children are visited in the reference's order (near side first at each of the three splits,
//     from the split axes and the ray's direction signs), found by permuting the 8-bit mask of the
//     children that passed; ONE stack entry per record holds the rest of that mask, the two base
//     keys, the children's kinds and the least entry distance of the group (rounded down to its
//     upper 16 bits): a pop takes the next child of the top group without touching memory, and
//     drops a group whose least entry distance the bound has since fallen below.
//   * no per-child entry distances, no re-check of a child at its pop: its own children (or
//     triangles) meet the then-current bound when it is visited.
// A superset of the reference's leaves in the reference's order, with the reference's own FP64
// triangle test and strict `t < t_min`, ends at the reference's triangle unless a triangle was
// accepted inside a leaf the reference would not have entered; job_finalize checks the winning
// triangle's leaf against the reference's exact box test (see there).
struct B8Ray { float ohx, ohy, ohz, ivx, ivy, ivz, clx, cly, clz, slack; };

RTX_DEV B8Ray b8_ray(const Lane& L) {
    B8Ray r;
    // origin as a float and the remainder, so that (x - origin) loses nothing to the origin's rounding;
    // an axis the reference treats as parallel (|d| < 1e-9f, bvh.cpp:36) gets no constraint here
    r.ohx = (float)L.ro.x; r.ohy = (float)L.ro.y; r.ohz = (float)L.ro.z;
    r.ivx = (L.par & 1) ? 0.0f : (float)L.ix; r.ivy = (L.par & 2) ? 0.0f : (float)L.iy; r.ivz = (L.par & 4) ? 0.0f : (float)L.iz;
    r.clx = (float)(L.ro.x - (double)r.ohx) * r.ivx; r.cly = (float)(L.ro.y - (double)r.ohy) * r.ivy; r.clz = (float)(L.ro.z - (double)r.ohz) * r.ivz;
    const float cmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(r.ohx * r.ivx), __builtin_fabsf(r.ohy * r.ivy)), __builtin_fabsf(r.ohz * r.ivz));
    r.slack = cmax * 1.1368684e-13f + 1e-30f;          // 2^-43 of the largest |origin x 1/d|, and a floor
    return r;
}

#define RTX_B8_EPS 4.7683716e-7f        // 2^-21: relative margin of a float slab distance (its error is below 2^-22)

// the dword `d` of a record held as 32 doubles, as a float / as an integer
#define B8_F(rec, d) __int_as_float(((d) & 1) ? __double2hiint((rec)[(d) >> 1]) : __double2loint((rec)[(d) >> 1]))
#define B8_U(rec, d) ((unsigned)(((d) & 1) ? __double2hiint((rec)[(d) >> 1]) : __double2loint((rec)[(d) >> 1])))

// child in slot `c` of a record with base keys (child_base, tri_base) and kinds `counts`
RTX_DEV void b8_child(const unsigned child_base, const unsigned tri_base, const unsigned counts, const unsigned c, int& ref, int& cnt) {
    const unsigned nib = (counts >> (4u * c)) & 15u;
    const unsigned m15 = counts & (counts >> 1) & (counts >> 2) & (counts >> 3) & 0x11111111u;      // bit 4s: slot s is inner
    const unsigned below = (1u << (4u * c)) - 1u;
    const unsigned rank = (unsigned)__popc(m15 & below);
    const unsigned leaves = (counts & ~(m15 * 15u)) & below;                                          // triangle counts of the leaf slots before c
    const unsigned sum = ((((leaves & 0x0F0F0F0Fu) + ((leaves >> 4) & 0x0F0F0F0Fu)) * 0x01010101u) >> 24);
    const bool inner = nib == 15u;
    ref = (int)(inner ? child_base + 2u * rank : tri_base + sum);
    cnt = inner ? 0 : (int)nib;
}
// position in the visiting order -> slot: at each of the three levels the far side comes first where the flip bit is set
RTX_DEV unsigned b8_slot(const unsigned p, const unsigned flips) {
    const unsigned b2 = ((p >> 2) ^ flips) & 1u;
    const unsigned b1 = ((p >> 1) ^ (flips >> (1u + b2))) & 1u;
    const unsigned b0 = (p ^ (flips >> (3u + 2u * b2 + b1))) & 1u;
    return b2 << 2 | b1 << 1 | b0;
}

template <bool FAST, int WIDE>
RTX_DEV unsigned long long traverse_steps8(const Scene& S, const Wave& wv, Lane& L, glob_v4i* __restrict__ stk,
                             const int service_threshold, const bool refillable, const int wide_lanes, const int wide_age,
                             unsigned& n_records, Tally& tl) {
    const char* const blas = S.blas;
    const bool cull = S.cull != 0;
    const double isect_eps = S.isect_eps;
    // (the lane's job changes only in service passes, between two calls)
    const B8Ray R = b8_ray(L);

    while (true) {
        const bool in_job = L.state == LS_IN_JOB;
        const bool at_leaf = in_job && L.cur_cnt > 0;
        const unsigned long long m_job = __ballot(in_job);
        if (m_job == 0ull) break;
        const unsigned long long m_wait = __ballot(L.state == LS_NEED_JOB || L.state == LS_JOB_DONE ||
                                                   (refillable && L.state == LS_EMPTY));
        if (__popcll(m_wait) >= service_threshold) break;
        // the generation has drained and this wave holds its last few jobs: the caller finishes them wave-wide
        if (WIDE != 0 && !refillable && __popcll(m_job) <= wide_lanes) {
            const unsigned long long m_old = __ballot(in_job && L.age >= wide_age);
            if (m_old != 0ull) return m_old;
        }
        const bool any_par = __ballot(in_job && (L.par & RTX_PAR_AXES) != 0) != 0ull;

#if RTX_WAVE_TALLY == 3
        const long long k_f0 = RTX_TICK();
#endif
        double rec[32];
        n_records += (unsigned)__popcll(m_job);
        if (__popcll(m_job) <= RTX_SPARSE_LANES) {
            if (in_job) {
                const RTX_GLOBAL v2d* gp = (const RTX_GLOBAL v2d*)(blas + ((size_t)(unsigned)L.cur_ref << 7));
#pragma unroll
                for (int p = 0; p < 16; p++) { const v2d v = gp[p]; rec[2 * p] = v.x; rec[2 * p + 1] = v.y; }
            }
        } else {
            const unsigned row = coop_fetch(wv, in_job, L.cur_ref, blas);
            row_read<0, 16>(wv, row, rec);
            __builtin_amdgcn_wave_barrier();
        }
#if RTX_WAVE_TALLY == 3
        const long long k_f1 = RTX_TICK();
        tl.clk_fetch += (unsigned long long)(k_f1 - k_f0);
#endif

        bool need_pop = false;
        if (in_job) {
            L.age++;
            if (!at_leaf) {
                // ---- eight children: conservative float slab tests ------------------------------
                const unsigned child_base = B8_U(rec, 48), tri_base = B8_U(rec, 49), counts = B8_U(rec, 50), axes = B8_U(rec, 51);
                // the bound as a float that is not below it
                float bound = (float)L.t_min;
                if ((double)bound < L.t_min) bound = __int_as_float(__float_as_int(bound) + 1);      // (positive)
                unsigned H = 0u;
                float emin = INFINITY;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const float tx0 = __builtin_fmaf(B8_F(rec, c) - R.ohx, R.ivx, -R.clx), tx1 = __builtin_fmaf(B8_F(rec, 24 + c) - R.ohx, R.ivx, -R.clx);
                    const float ty0 = __builtin_fmaf(B8_F(rec, 8 + c) - R.ohy, R.ivy, -R.cly), ty1 = __builtin_fmaf(B8_F(rec, 32 + c) - R.ohy, R.ivy, -R.cly);
                    const float tz0 = __builtin_fmaf(B8_F(rec, 16 + c) - R.ohz, R.ivz, -R.clz), tz1 = __builtin_fmaf(B8_F(rec, 40 + c) - R.ohz, R.ivz, -R.clz);
                    float nx = __builtin_fminf(tx0, tx1), fx = __builtin_fmaxf(tx0, tx1);
                    float ny = __builtin_fminf(ty0, ty1), fy = __builtin_fmaxf(ty0, ty1);
                    float nz = __builtin_fminf(tz0, tz1), fz = __builtin_fmaxf(tz0, tz1);
                    if (any_par) {          // (an axis without a constraint)
                        if (L.par & 1) { nx = -INFINITY; fx = INFINITY; }
                        if (L.par & 2) { ny = -INFINITY; fy = INFINITY; }
                        if (L.par & 4) { nz = -INFINITY; fz = INFINITY; }
                    }
                    const float enter = __builtin_fmaxf(__builtin_fmaxf(nx, ny), __builtin_fmaxf(nz, 0.0f));
                    const float exit = __builtin_fminf(__builtin_fminf(fx, fy), fz);
                    // margins: a computed distance is within 2^-22 of itself (+ slack) of the true one
                    const float enter_c = __builtin_fmaf(-__builtin_fabsf(enter), RTX_B8_EPS, enter) - R.slack;
                    const float exit_c = __builtin_fmaf(__builtin_fabsf(exit), RTX_B8_EPS, exit) + R.slack;
                    const bool hit = (exit_c >= enter_c) & (bound >= enter_c) & (((counts >> (4 * c)) & 15u) != 0u);
                    H |= hit ? (1u << c) : 0u;
                    emin = hit ? __builtin_fminf(emin, enter_c) : emin;
                }
                // ---- the reference's visiting order (bvh.cpp:139-147): bit p of Hp = the child visited p-th
                const unsigned ff = (unsigned)L.par >> 4;                 // per axis: the far side of a split along it comes first
                unsigned flips = (ff >> (axes & 3u)) & 1u;
                flips |= ((ff >> ((axes >> 2) & 3u)) & 1u) << 1 | ((ff >> ((axes >> 4) & 3u)) & 1u) << 2;
                flips |= ((ff >> ((axes >> 6) & 3u)) & 1u) << 3 | ((ff >> ((axes >> 8) & 3u)) & 1u) << 4 |
                         ((ff >> ((axes >> 10) & 3u)) & 1u) << 5 | ((ff >> ((axes >> 12) & 3u)) & 1u) << 6;
                unsigned Hp = H;
                {
                    const unsigned f2 = flips >> 3;
                    const unsigned e0 = (f2 & 1u) | (f2 & 2u) << 1 | (f2 & 4u) << 2 | (f2 & 8u) << 3;             // pairs whose two slots swap
                    unsigned t = ((Hp >> 1) ^ Hp) & e0;
                    Hp ^= t | (t << 1);
                    const unsigned e1 = ((flips & 2u) ? 0x03u : 0u) | ((flips & 4u) ? 0x30u : 0u);               // halves whose two pairs swap
                    t = ((Hp >> 2) ^ Hp) & e1;
                    Hp ^= t | (t << 2);
                    Hp = (flips & 1u) ? ((Hp >> 4) | (Hp << 4)) & 0xFFu : Hp;                                       // the halves themselves
                }
                if (Hp == 0u) {
                    need_pop = true;
                } else {
                    const unsigned p = (unsigned)__ffs((int)Hp) - 1u;
                    Hp &= Hp - 1u;
                    b8_child(child_base, tri_base, counts, b8_slot(p, flips), L.cur_ref, L.cur_cnt);
                    if (Hp != 0u) {
                        // the others: one entry for the record
                        if (L.sp >= RTX_STACK) { *S.limit_flag = 1u; }
                        else {
                            if (L.sp > 0) stk[L.sp - 1] = L.top;
                            const unsigned e16 = emin > 0.0f ? ((unsigned)__float_as_int(emin) >> 16) : 0u;     // (truncated: not above emin)
                            L.top.x = (int)child_base; L.top.y = (int)tri_base; L.top.z = (int)counts;
                            L.top.w = (int)(Hp | flips << 8 | e16 << 16);
                            L.sp++;
                        }
                    }
                }
            } else {
                // ---- leaf: its next two triangles, in leaf order (bvh.cpp:150-170) ----------------
                const int ti = L.cur_ref;
                const bool two = L.cur_cnt >= 2;
                const double* td = rec;
                const double* tq = rec + 16;
                bool culled0 = false, culled1 = false;
                if (cull) {                                                                   // utils.cpp:322-326
                    culled0 = L.rd.x * td[10] + L.rd.y * td[11] + L.rd.z * td[12] > 0.0;
                    culled1 = L.rd.x * tq[10] + L.rd.y * tq[11] + L.rd.z * tq[12] > 0.0;
                }
                double t, b, g2;
                if (!culled0 && tri_test(td, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.bound_before = L.t_min; L.t_min = t; L.btri = ti; }
                if (two && !culled1 && tri_test(tq, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.bound_before = L.t_min; L.t_min = t; L.btri = ti + 1; }
                L.cur_ref += 2; L.cur_cnt -= 2;
                need_pop = L.cur_cnt <= 0;
                if ((L.is_shadow & 2) && L.btri >= 0 && L.t_min < L.orig_t_min) {
                    need_pop = true; L.sp = 0;         // occluder found: the job ends here
                }
                if ((L.is_shadow & 4) && L.btri >= 0 && L.t_min < (double)__int_as_float(__double2hiint(wv.ctx[4]))) {
                    need_pop = true; L.sp = 0;         // the same inside a transformed object (job_next)
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#if RTX_WAVE_TALLY == 3
        const long long k_p0 = RTX_TICK();
#endif
        if (need_pop) {
            L.state = LS_JOB_DONE;
            while (L.sp > 0) {
                const unsigned w = (unsigned)L.top.w;
                unsigned Hp = w & 0xFFu;
                // a group whose least entry distance lies beyond the bound by now is dropped whole
                const bool stale = L.t_min < (double)__int_as_float((int)(w & 0xFFFF0000u));
                if (!stale) {
                    const unsigned p = (unsigned)__ffs((int)Hp) - 1u;
                    Hp &= Hp - 1u;
                    b8_child((unsigned)L.top.x, (unsigned)L.top.y, (unsigned)L.top.z, b8_slot(p, (w >> 8) & 0x7Fu), L.cur_ref, L.cur_cnt);
                    L.state = LS_IN_JOB;
                    if (Hp != 0u) { L.top.w = (int)((w & ~0xFFu) | Hp); break; }
                }
                --L.sp;
                if (L.sp > 0) L.top = stk[L.sp - 1];
                if (!stale) break;
            }
        }
#if RTX_WAVE_TALLY == 3
        { const long long k_e = RTX_TICK(); tl.clk_pop += (unsigned long long)(k_e - k_p0); tl.clk_trav += (unsigned long long)(k_p0 - k_f1); tl.passes++; }
#endif
    }
    return 0ull;
}

// The last jobs of a drained wave of the eight-wide BLAS, finished by the whole wave (wide_multi).
// The wave-wide traversal works on the four-wide records, whose stack entries are of another kind, so
// a job starts over at its mesh's root there -- with what it has found so far as its bound: one ulp
// above the distance of its best triangle, so that the triangle itself and any other at exactly that
// distance are met again and the first of them in the reference's order is the result, as always.
RTX_DEV void wide_jobs8(const Scene& S, const Wave& wv, const ObjCache& oc, Lane& L, glob_v4i* stk, unsigned* inst_bits,
                        unsigned long long m) {
    WideIn in;
    in.rox = L.ro.x; in.roy = L.ro.y; in.roz = L.ro.z; in.rdx = L.rd.x; in.rdy = L.rd.y; in.rdz = L.rd.z;
    in.ix = L.ix; in.iy = L.iy; in.iz = L.iz; in.stop = L.orig_t_min;
    in.t_min = (L.btri >= 0 && L.t_min < DBL_MAX) ? __longlong_as_double(__double_as_longlong(L.t_min) + 1ll) : L.t_min;
    in.par = L.par; in.shadow = L.is_shadow; in.sp = 0; in.top = (v4i)(0);
    in.cur_ref = 0; in.cur_cnt = 0;
    if ((m >> wv.lane) & 1ull) {
        const Obj* ob = oc.at(S, L.obj);
        in.cur_ref = ob->root_ref; in.cur_cnt = ob->root_count;
    }
    while (m != 0ull) {
        const WideOut out = wide_multi(S, wv, in, stk, inst_bits, m);
        const unsigned long long sel = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(out.sel >> 32)) << 32) |
                                       (unsigned)__builtin_amdgcn_readfirstlane((int)out.sel);
        m &= ~sel;
        if ((sel >> wv.lane) & 1ull) {
            if (out.t_min < INFINITY) { L.t_min = out.t_min; L.btri = out.btri; }      // (a key of the reference-order triangle array)
            L.sp = 0;
            L.state = LS_JOB_DONE;
        }
    }
}

// WIDE = 1: main variant in which drained waves finish their jobs wave-wide (wide_multi): those that
// have run RTX_WIDE_AGE steps once at most RTX_WIDE_LANES are left, and every job from its first step in
// generations of at most RTX_WIDE_MAX_RAYS rays.
// WIDE = 0: main variant without any of it.  (Variants rather than branches: the mere presence
// of the call changes the register allocation of the code around it -- the service passes pay
// for it, which scenes dominated by instance replay notice -- so the host picks per scene.)
// WIDE = 2: launched beside a WIDE = 0 main variant, for generations of at most RTX_WIDE_MAX_RAYS
// rays, all of whose jobs it traverses wave-wide; exactly one of the two does the work, the other
// returns at once.
template <bool COUNT, int WIDE, bool FAST, int NEST = 0, bool B8 = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RTX_TRACE_WAVES, RTX_TRACE_WAVES)))
k_trace(const Launch* __restrict__ Lp, GenCtrl* ctrl, const int gen, double* __restrict__ accum,
        rtx_hit_id* __restrict__ aov, Counters* __restrict__ counters) {
    const Scene& S = Lp->S;
    const Frame& F = Lp->F;
    const Queue& qin = Lp->q[(gen + 1) & 1];
    __shared__ __attribute__((aligned(256))) double lds_rows[4][64 * RTX_ROW_DOUBLES];     // 64 KB per block
    Wave wv;
    // (wave-uniform by construction; saying so keeps the LDS-DMA destinations in scalar registers)
    wv.rows = (lds_double*)(size_t)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_double*)&lds_rows[threadIdx.x >> 6][0]);
    wv.lane = lane_id();
    const unsigned lane = wv.lane;
    GenCtrl* me = ctrl + gen;
    // A generation whose queue overflowed while it was being filled is incomplete, and the slots
    // its counters claim beyond the queue were never written: it is not traced (k_shade hands the
    // flag on, so every later generation stops too, and the host renders the frame again).
    if (gen != 0 && me->overflow != 0u) return;
    GenItems items;
    unsigned n_in;
    if (gen == 0) n_in = (unsigned)F.n_items0;
    else { items.load(me, qin.sub_capacity); n_in = items.n_in; }
    // (an empty generation -- every one after a path's last bounce -- costs its launches and nothing else)
    if (n_in == 0u) return;
    // (a kernel with the wave-wide traversal serves tiny generations itself: see wide_age below)
    const bool tiny = !COUNT && Lp->wide_ok != 0 && n_in <= RTX_WIDE_MAX_RAYS;
    if (!COUNT && ((WIDE == 0 && tiny) || (WIDE == 2 && !tiny))) return;
    __shared__ __attribute__((aligned(16))) double lds_objs[RTX_LDS_OBJS * (sizeof(Obj) / 8)];
    ObjCache oc;
    oc.lds = (const Obj*)lds_objs;
    oc.n = min(S.n_meshes + S.n_insts, RTX_LDS_OBJS);
    __shared__ double lds_ctx[256 * 5];
    oc.ctx = lds_ctx + threadIdx.x * 5;
    wv.ctx = (lds_double*)(size_t)(unsigned)(size_t)(lds_double*)&lds_ctx[threadIdx.x * 5];
    for (unsigned i = threadIdx.x; i < (unsigned)oc.n * (unsigned)(sizeof(Obj) / 16); i += blockDim.x)
        ((v2d*)lds_objs)[i] = ((const RTX_GLOBAL v2d*)S.objs)[i];
    __shared__ __attribute__((aligned(16))) double lds_tiny[RTX_LDS_TINY * 32];
    oc.tiny = lds_tiny;
    // the tiny meshes among them: their one or two triangle records (sixteen 16-byte pieces a mesh)
    if (threadIdx.x < (unsigned)oc.n * 16u) {
        const RTX_GLOBAL Obj* ob = (const RTX_GLOBAL Obj*)S.objs + (threadIdx.x >> 4);
        const int slot = (ob->flags >> 8) - 1;
        if (slot >= 0 && slot < RTX_LDS_TINY)
            ((v2d*)lds_tiny)[slot * 16 + (threadIdx.x & 15u)] =
                ((const RTX_GLOBAL v2d*)(S.blas + (size_t)(unsigned)ob->root_ref * 128u))[threadIdx.x & 15u];
    }
    __syncthreads();
#if RTX_WAVE_TALLY
    const long long clk_begin_all = RTX_TICK();
    long long clk_drained = clk_begin_all, clk_wide = 0, clk_steps = 0, clk_service = 0, clk_lane = 0, clk_claim = 0;
    unsigned n_wide_jobs = 0, n_passes = 0;
#endif
    Tally tl = {};
    unsigned n_records = 0;                      // records this wave fetches (wave-uniform; a wave stays far below 2^32)
    const long long clk_begin = COUNT ? clock64() : 0;

    // Rays are claimed in chunks (one atomic per chunk per wave) and handed to lanes as they
    // fall idle.  A thin generation is latency bound: smaller chunks spread it over more waves.
    // Chunks shrink as the generation drains (guided scheduling: a share of what the last claim
    // saw remaining), so that no wave is left with several rounds of rays when the others run dry.
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = (n_in + 2u * n_waves - 1u) / (2u * n_waves);
    const unsigned chunk_max = Lp->chunk_max;
    chunk = chunk < 1u ? 1u : (chunk > chunk_max ? chunk_max : chunk);
    // A wave's first chunk is its share by position, without the counter: two thousand waves that
    // start together and all ask one counter wait in line for it (tens of microseconds for the last of
    // them, which is most of a thin generation).  The counter hands out what lies beyond those shares.
    const unsigned static_end = n_waves * chunk;
    unsigned chunk_next = min((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * chunk, n_in);      // wave-uniform
    unsigned chunk_end = min(chunk_next + chunk, n_in);
    bool drained = false;                        // wave-uniform: the generation has no rays left

    glob_v4i* const stk = (glob_v4i*)Lp->stacks +
                          ((size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane) * RTX_STACK;
    unsigned* const inst_bits = Lp->inst_bits + ((size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane) * Lp->inst_words;
    const int service_threshold = (int)Lp->service_threshold;
    const bool thin = gen != 0 && n_in < RTX_SCATTER_THIN_RAYS;
    const int wide_lanes = thin ? RTX_WIDE_LANES_THIN : RTX_WIDE_LANES, wide_age = tiny ? 0 : thin ? RTX_WIDE_AGE_THIN : RTX_WIDE_AGE;
    Lane L;
    L.state = LS_EMPTY; L.cur_cnt = 0; L.cur_ref = 0; L.sp = 0; L.slot = 0; L.is_shadow = 0; L.btri = -1; L.obj = 0;
    L.top = (v4i)(0); L.t_min = 0.0; L.ro = mk(0, 0, 0); L.rd = mk(0, 0, 1); L.ix = L.iy = L.iz = 0.0; L.par = 0;
    L.o = mk(0, 0, 0); L.d = mk(0, 0, 1); L.it = 0; L.word = 0; L.whi = -1; L.bits = 0u; L.sword = 0; L.sbits = 0u; L.age = 0;

    while (true) {
        const int state = L.state;
        const unsigned long long m_empty = __ballot(state == LS_EMPTY);
        const unsigned long long m_need = __ballot(state == LS_NEED_JOB || state == LS_JOB_DONE);
        const unsigned long long m_trav = __ballot(state == LS_IN_JOB);
        const int n_service = __popcll(m_need) + (drained ? 0 : __popcll(m_empty));

        if (n_service > 0 && (m_trav == 0ull || n_service >= service_threshold)) {
            // ---- service pass ---------------------------------------------------------------
            const long long c0 = COUNT ? clock64() : 0;
#if RTX_WAVE_TALLY
            const long long s0 = RTX_TICK();
#endif
            for (int rep = 0; rep < RTX_SERVICE_REPS; rep++) {
                const unsigned long long me_ = __ballot(L.state == LS_EMPTY);
                bool fresh = false;
                if (me_ != 0ull && !drained) {
                    if (chunk_next == chunk_end) {
                        unsigned base = n_in;
                        if (static_end < n_in) {
#if RTX_WAVE_TALLY
                            const long long a0 = RTX_TICK();
#endif
                            if (lane == 0) base = static_end + atomicAdd(&me->head_trace, chunk);
                            base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
#if RTX_WAVE_TALLY
                            clk_claim += RTX_TICK() - a0;
#endif
                        }
                        if (base >= n_in) {
                            drained = true; chunk_next = chunk_end = 0;
#if RTX_WAVE_TALLY
                            clk_drained = RTX_TICK();
#endif
                        }
                        else {
                            chunk_next = base; chunk_end = min(base + chunk, n_in);
                            const unsigned share = (n_in - chunk_end) / (2u * n_waves);
                            chunk = share < 16u ? 16u : (share > chunk_max ? chunk_max : share);
                            if (chunk > n_in) chunk = n_in;
                        }
                    }
                    if (!drained) {
                        const unsigned take = min((unsigned)__popcll(me_), chunk_end - chunk_next);
                        const unsigned rank = (unsigned)__popcll(me_ & ((1ull << lane) - 1ull));
                        if (L.state == LS_EMPTY && rank < take) {
                            const unsigned item = chunk_next + rank;
                            if (gen == 0) {
                                int px, py; unsigned pixel;
                                L.slot = item; L.is_shadow = 0;
                                if (primary_pixel(F, item, px, py, pixel)) {
                                    L.o = ld3(F.e); L.d = primary_direction(F, px, py); fresh = true;
                                } else {
                                    ((RTX_GLOBAL Hit*)Lp->hb.recs + item)->code = -2;      // padding item outside the image
                                }
                            } else {
                                bool sh;
                                L.slot = items.slot(item, sh);
                                L.is_shadow = sh ? 1 : 0;
                                const RTX_GLOBAL Ray* r = (const RTX_GLOBAL Ray*)qin.rays + L.slot;
                                L.o = mk(r->o[0], r->o[1], r->o[2]);
                                L.d = mk(r->d[0], r->d[1], r->d[2]);
                                if (sh) {
                                    oc.ctx[0] = r->aux; oc.ctx[1] = r->w[0]; oc.ctx[2] = r->w[1]; oc.ctx[3] = r->w[2];
                                    oc.ctx[4] = __hiloint2double(0, (int)r->pixel);
                                }
                                fresh = true;
                            }
                        }
                        chunk_next += take;
                    }
                }
#if RTX_WAVE_TALLY
                const long long l0 = RTX_TICK();
#endif
                if (fresh || L.state == LS_JOB_DONE || L.state == LS_NEED_JOB)
                    service_lane<COUNT, FAST, (WIDE == 0 && !COUNT && !B8) ? RTX_REPLAY_CAP : 0, NEST, B8>(Lp, oc, gen, accum, aov, L, inst_bits, stk, fresh, tl);
#if RTX_WAVE_TALLY
                clk_lane += RTX_TICK() - l0;
#endif
                if (drained || __ballot(L.state == LS_EMPTY) == 0ull) break;
            }
            if (COUNT) { tl.clk_serv += (unsigned long long)(clock64() - c0); tl.passes++; }
#if RTX_WAVE_TALLY
            clk_service += RTX_TICK() - s0; n_passes++;
#endif
            continue;
        }
        if (m_trav == 0ull) break;
        const long long c1 = COUNT ? clock64() : 0;
#if RTX_WAVE_TALLY
        const long long t0 = RTX_TICK();
#endif
        unsigned long long m_wide = 0ull;
        if (B8) m_wide = traverse_steps8<FAST, WIDE>(S, wv, L, stk, service_threshold, !drained, wide_lanes, wide_age, n_records, tl);
        else m_wide = traverse_steps<COUNT, WIDE, NEST>(S, wv, L, stk, inst_bits, service_threshold, !drained, wide_lanes, wide_age, tl, n_records);
#if RTX_WAVE_TALLY
        const long long w0 = RTX_TICK();
        clk_steps += w0 - t0;
#endif
        if (!COUNT && WIDE != 0 && m_wide != 0ull) {
            if (B8) wide_jobs8(S, wv, oc, L, stk, inst_bits, m_wide);
            else wide_jobs(S, wv, L, stk, inst_bits, m_wide);
        }
#if RTX_WAVE_TALLY
        if (m_wide != 0ull) { clk_wide += RTX_TICK() - w0; n_wide_jobs += (unsigned)__popcll(m_wide); }
#endif
        if (COUNT) tl.clk_trav += (unsigned long long)(clock64() - c1);
    }

    if (lane == 0 && n_records != 0u) atomicAdd(&me->records, (unsigned long long)n_records);
#if RTX_WAVE_TALLY
    // diagnostic builds: lifetimes of this generation's waves (sum, maximum, count) in its control block's padding
    if (lane == 0) {
        const unsigned long long life = (unsigned long long)(RTX_TICK() - clk_begin_all);
        atomicAdd((unsigned long long*)&me->tally[8], life);
        // the longest wave and what it spent its time on: lifetime in units of 64 ticks, then per cent of it
        // in wave-wide jobs, in traversal steps and in service passes, then its numbers of wave-wide jobs and of
        // service passes (RTX_WAVE_TALLY=2: in claims and in the lanes' service instead of the first two)
        if (RTX_WAVE_TALLY == 3) {      // the traversal steps' sections instead: record fetch, tests, stack; steps / 8
            clk_wide = (long long)tl.clk_fetch; clk_steps = (long long)tl.clk_trav; clk_service = (long long)tl.clk_pop; n_passes = tl.passes >> 3;
        }
        atomicMax((unsigned long long*)&me->tally[10], ((life >> 6) << 38) | ((unsigned long long)((RTX_WAVE_TALLY == 2 ? clk_claim : clk_wide) * 100 / (long long)life) << 31) |
                  ((unsigned long long)((RTX_WAVE_TALLY == 2 ? clk_lane : clk_steps) * 100 / (long long)life) << 24) | ((unsigned long long)(clk_service * 100 / (long long)life) << 17) |
                  ((unsigned long long)(n_wide_jobs > 255u ? 255u : n_wide_jobs) << 9) | (n_passes > 511u ? 511u : n_passes));
        atomicAdd(&me->tally[12], 1u);
        if (RTX_WAVE_TALLY == 4) {      // sums over all waves instead of the bins: ticks in traversal steps, service passes, wave-wide jobs; passes
            atomicAdd((unsigned long long*)&me->tally[0], (unsigned long long)clk_steps);
            atomicAdd((unsigned long long*)&me->tally[2], (unsigned long long)clk_service);
            atomicAdd((unsigned long long*)&me->tally[4], (unsigned long long)clk_wide);
            atomicAdd((unsigned long long*)&me->tally[6], (unsigned long long)n_passes);
        } else {
        const unsigned bin = (unsigned)(life / (150ull * 2400ull));          // 150 us bins at 2.4 GHz
        atomicAdd(&me->tally[bin < 7u ? bin : 7u], 1u);
        }
        // ... and how long the wave went on after it found the queue dry (sum and maximum)
        const unsigned long long after = (unsigned long long)(RTX_TICK() - clk_drained);
        atomicAdd((unsigned long long*)&me->tally[14], after);
        atomicMax(&me->tally[13], (unsigned)(after >> 8));
    }
#endif
    if (COUNT && counters) {
        const bool l0 = lane == 0;
        unsigned long long v[46] = {tl.rays, tl.primary, tl.shadow, tl.secondary, tl.node_visits, tl.tri_tests,
                                    tl.world_box, tl.obj_trav, tl.tlas_visits, tl.inner_steps, tl.inner_lanes,
                                    tl.leaf_steps, tl.leaf_lanes, l0 ? tl.passes : 0u, l0 ? tl.clk_trav : 0ull,
                                    l0 ? tl.clk_serv : 0ull, l0 ? (unsigned long long)(clock64() - clk_begin) : 0ull,
                                    l0 ? 1ull : 0ull, l0 ? tl.clk_fetch : 0ull, l0 ? tl.clk_pop : 0ull, 0ull, 0ull,
                                    tl.sec[0], tl.sec[1], tl.sec[2], tl.sec[3], tl.sec[4], tl.sec[5], tl.sec[6], tl.sec[7],
                                    tl.hist[0], tl.hist[1], tl.hist[2], tl.hist[3], tl.hist[4], tl.hist[5], tl.hist[6], tl.hist[7],
                                    tl.hist_shadow[0], tl.hist_shadow[1], tl.hist_shadow[2], tl.hist_shadow[3], tl.hist_shadow[4],
                                    tl.hist_shadow[5], tl.hist_shadow[6], tl.hist_shadow[7]};
#pragma unroll
        for (int k = 0; k < 46; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(((unsigned long long*)counters) + k, x);
        }
        {
            unsigned long long x = tl.longest_ray;
            for (int off = 32; off > 0; off >>= 1) { const unsigned long long y = __shfl_down(x, off); x = y > x ? y : x; }
            if (lane == 0 && x) atomicMax(&counters->longest_ray, x);
        }
        if (lane == 0) {
            const unsigned long long life = (unsigned long long)(clock64() - clk_begin);
            const unsigned long long prev = atomicMax(&counters->max_wave_ticks, life);
            if (life > prev) counters->max_wave_steps = tl.inner_steps + tl.leaf_steps + ((unsigned long long)tl.passes << 32);
        }
    }
}

// Shading of the radiance hits of one generation (utils.cpp:868-1061) and emission of the
// next generation's rays.
__global__ void __launch_bounds__(256)
k_shade(const Launch* __restrict__ Lp, GenCtrl* ctrl, const int gen, double* __restrict__ accum) {
    const Scene& S = Lp->S;
    const Frame& F = Lp->F;
    const Queue& qin = Lp->q[(gen + 1) & 1];
    const Queue& qout = Lp->q[gen & 1];
    const HitBuf& hb = Lp->hb;
    const unsigned lane = lane_id();
    GenCtrl* me = ctrl + gen;
    GenCtrl* next = ctrl + gen + 1;
    // an overflowed generation was not traced (see k_trace): nothing to shade, and the next one stops too
    if (gen != 0 && me->overflow != 0u) {
        if (blockIdx.x == 0 && threadIdx.x == 0) next->overflow = 1u;
        return;
    }
    GenItems items;
    unsigned n_in;
    if (gen == 0) n_in = (unsigned)F.n_items0;
    else { items.load(me, qin.sub_capacity); n_in = items.n_rad; }
    if (n_in == 0u) return;
    const unsigned my_sub = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (RTX_NSUB - 1u);

    // Batches of 64 hits are dealt to the waves round-robin: shading cost is nearly uniform, so
    // a static deal balances as well as a shared counter without serialising on it.
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
    for (unsigned base = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u; base < n_in; base += n_waves * 64u) {
        const unsigned item = base + lane;
        bool active = item < n_in;
        unsigned slot = item;
        if (gen != 0 && active) { bool sh; slot = items.slot(item, sh); }

        V3 d = mk(0, 0, 1), W = mk(1, 1, 1);
        unsigned pixel = 0, depth = 0;
        int beer = -1;
        if (gen == 0) {
            int px, py;
            active = primary_pixel(F, slot, px, py, pixel) && active;
            if (active) d = primary_direction(F, px, py);
        } else if (active) {
            const RTX_GLOBAL Ray* r = (const RTX_GLOBAL Ray*)qin.rays + slot;
            d = mk(r->d[0], r->d[1], r->d[2]);
            W = mk(r->w[0], r->w[1], r->w[2]);
            pixel = r->pixel;
            const unsigned meta = r->meta;
            depth = meta & 0xFFu; beer = (int)(meta >> 8) - 1;
        }

        bool shade = false;
        int mat = -1;
        V3 P = mk(0, 0, 0), gn = mk(0, 0, 1), sn = mk(0, 0, 1);
        if (active) {
            const RTX_GLOBAL Hit* h = (const RTX_GLOBAL Hit*)hb.recs + slot;
            const int code = h->code;
            if (code < 0) {
                if (depth == 0 && code == -1) add_color(accum, pixel, vmul(W, ld3(S.bg)));      // utils.cpp:1055-1058
            } else {
                mat = h->mat;
                if (beer >= 0) {                                                     // utils.cpp:975-984
                    const Material& pm = S.materials[beer];
                    V3 ab = vscale(ld3(pm.sigma), h->dist);
                    W = vmul(W, mk(exp(-ab.x), exp(-ab.y), exp(-ab.z)));
                }
                if (mat >= 0) {
                    shade = true;
                    P = mk(h->p[0], h->p[1], h->p[2]);
                    // tail of intersect(), utils.cpp:807-853: normalise both normals; for an
                    // untransformed smooth mesh (or a record a rejected transformed hit left
                    // behind) interpolate the vertex normals and face them against the ray
                    gn = normalize(mk(h->g[0], h->g[1], h->g[2]));
                    sn = normalize(mk(h->s[0], h->s[1], h->s[2]));
                    const int cont = h->container;
                    int ptri = -1, pflags = 0;
                    if (h->kind == RTX_KIND_MESH && cont >= 0 && cont < S.n_meshes) {
                        pflags = S.objs[cont].flags;          // object `cont` is mesh `cont` itself
                        if (pflags & OBJ_SMOOTH_VN) ptri = h->tri;
                    }
                    if (ptri >= 0) {
                        const double u = h->beta, v = h->gamma, w = 1.0 - u - v;
                        const double* vn = S.tri_vn + RTX_VN_STRIDE * (size_t)ptri;
                        V3 n0 = ld3(vn), n1 = ld3(vn + 3), n2 = ld3(vn + 6);
                        V3 sm = normalize(vadd(vadd(sscale(w, n0), sscale(u, n1)), sscale(v, n2)));
                        if (pflags & OBJ_XF) sm = normalize(xf_dir(S.objs[cont].nrm, sm));
                        sn = sm;
                        if (dot(gn, d) > 0.0) { gn = vneg(gn); sn = vneg(sn); }
                        else if (dot(sn, gn) < 0.0) sn = vneg(sn);
                    }
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
        // Which of this hit's shadow rays nothing can hold up (bit li: light li): those add their
        // contribution below instead of being queued.  The reservation needs the count first.
        const V3 s_origin = vadd(P, sscale(S.shadow_eps, gn));
        unsigned free_lights = 0u, n_queued = 0u, n_early = 0u;
        if (Lp->early_shadow != 0u) {
            for (int li = 0; li < S.n_lights; li++) {
                bool is_free = false;
                if (lit) is_free = shadow_ray_is_free(S, s_origin, normalize(vsub(ld3(S.lights[li].p), P)));
                free_lights |= is_free ? (1u << li) : 0u;
                const unsigned n_free = (unsigned)__popcll(__ballot(is_free));
                n_early += n_free;
                n_queued += (unsigned)__popcll(__ballot(lit)) - n_free;
            }
        } else {
            n_queued = (unsigned)__popcll(__ballot(lit)) * (unsigned)S.n_lights;
        }
        const Emit em = emit_reserve(qout, next, my_sub, (unsigned)__popcll(__ballot(c0)) + (unsigned)__popcll(__ballot(c1)), n_queued);
        if (n_early != 0u && lane == 0u) atomicAdd(&next->sub[my_sub].early, (unsigned long long)n_early);
        unsigned sh_at = em.sh;                     // first slot of the next light's rays (wave-uniform)
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
                so = s_origin;
                sd = normalize(vsub(lp, P));
                sw = vmul(W, contrib);
                dist_to_light = sqrt(dot(vsub(lp, P), vsub(lp, P)));
            }
            const bool settled = lit && ((free_lights >> li) & 1u) != 0u;
            if (settled) add_color(accum, pixel, sw);                          // utils.cpp:1083: not in shadow
            sh_at += emit(qout, my_sub, em, sh_at, true, lit && !settled, so, sd, sw, dist_to_light, pixel, 0u);
        }
        const unsigned n_c0 = emit(qout, my_sub, em, em.rad, false, c0, c0o, c0d, c0w, 0.0, pixel, RTX_META(depth + 1, -1));
        emit(qout, my_sub, em, em.rad + n_c0, false, c1, c1o, c1d, c1w, 0.0, pixel, RTX_META(depth + 1, c1beer));
    }
}

// Final pass: clamp to [0,255] (utils.cpp:125-132), round half away from zero and store as
// u8 (raytracer.cpp:78-80).  For whole frames the local pixel order is the image order.
__global__ void k_resolve(const double* __restrict__ accum, uint8_t* __restrict__ rgb, unsigned n_values) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_values) return;
    double c = (double)((const long long*)accum)[i] / RTX_ACCUM_SCALE;
    if (c < 0) c = 0;
    if (c > 255) c = 255;
    rgb[i] = (uint8_t)round(c);
}

}  // namespace rtxdev
