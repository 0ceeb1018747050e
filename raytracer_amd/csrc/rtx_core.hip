// C-ABI of the ray-casting core (include/rtx_amd.h): scene flattening + upload, the
// per-frame generation loop, download.  Host code only orchestrates; every pixel is
// computed by the kernels in rtx_kernels.h.  There is no CPU rendering path here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "rtx_amd.h"
#include "rtx_device_types.h"
#include "rtx_kernels.h"

using namespace rtxdev;

namespace {

thread_local std::string g_error;

int fail(int code, const std::string& msg) { g_error = msg; return code; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(RTX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t ensure(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    template <typename T> hipError_t upload(const std::vector<T>& v) {
        hipError_t e = ensure(std::max<size_t>(v.size() * sizeof(T), 16));
        if (e != hipSuccess) return e;
        if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
};

struct HBox {
    double mn[3], mx[3];
    HBox() { for (int i = 0; i < 3; i++) { mn[i] = (double)1e30f; mx[i] = (double)-1e30f; } }   // bvh.h:13
    void expand(const double* p) { for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); } }
    void merge(const HBox& o) { for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], o.mn[i]); mx[i] = std::max(mx[i], o.mx[i]); } }
};

}  // namespace

// One version of a scene's point lights.  rtx_scene_update_lights publishes a new version; a frame
// keeps the version it was submitted with until it has been waited for.
struct LightSet { DevBuf buf; int n = 0; };

struct rtx_frame;

struct rtx_scene {
    int device = 0;
    int cus = 256;                  // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    Scene dev{};                    // kernel argument (device pointers inside); lights: see `lights`
    int n_meshes = 0;
    bool counters_on = false;
    bool consistent = false;
    bool generation_timing = true;  // events round every generation's trace launches (rtx_stats.trace_ms)
    bool wide_blas = false;         // rtx_scene_set_wide_blas: mesh traversal over the eight-wide float-box records
    uint32_t debug = 0;             // RTX_DEBUG_* development switches (rtx_scene_set_debug)
    int64_t queue_limit = (int64_t)32 << 30;
    DevBuf materials, xforms, planes, spheres, ftris, meshes, insts, blas, tri_vn, tri_aux;
    bool b8_built = false;          // the scene was created with RTX_SCENE_WIDE_BLAS
    bool b8_ok = false;             // ... and the eight-wide BLAS could be built for every mesh (rtx_device_types.h, Node8)
    std::atomic<int> b8_fallbacks{0};   // frames that had to be redone with the four-wide kernels
    // Geometry is shared read-only by every frame of the scene; the lights are versioned.
    std::mutex lights_mutex;
    std::shared_ptr<LightSet> lights;                      // current version
    std::vector<std::shared_ptr<LightSet>> light_pool;     // every version ever made (reused once no frame holds it)
    int record_levels = 0;          // deepest chain of records in any BLAS or the TLAS
    // Entering instances from the walk of the TLAS stacks a BLAS traversal on top of the walk's
    // entries: the most entries that can need (three per record level), and whether every instance
    // transform is affine (the pruning of such a walk relates object-space to world distances).
    int nest_stack_need = 0;
    bool affine_insts = true;
    // Exact mode's search hands a ray back to the index-order replay when its argument does not
    // cover the ray (shrunk instances, mostly).  A scene where that happens to more than one ray in
    // ten pays for both; from the next frame on it is rendered by the replay alone.
    std::atomic<bool> search_unprofitable{false};
    // Frames (rtx_frame_create) of this scene; `own` is the one behind rtx_render / rtx_render_noalloc.
    std::mutex frames_mutex;
    std::vector<rtx_frame*> frames;
    rtx_frame* own = nullptr;
};

// ---- scratch of the entry point that allocates nothing (rtx_render_noalloc) -------------------
// One fixed array per GPU, shared by every call on that GPU (calls that overlap on one GPU race).
// In host code the array's name is only a stand-in: its device address is looked up once per
// device (rtx_scene_create, with that device current) and only that address is ever used.
#define RTX_NOALLOC_SCRATCH_BYTES (64u << 20)
#define RTX_NOALLOC_BLOCKS 32u        // blocks of its trace grid (each lane has a traversal stack in the scratch)
#define RTX_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[RTX_NOALLOC_SCRATCH_BYTES];
static unsigned char* g_scratch_at[RTX_MAX_DEVICES];   // device address of `scratch` on each GPU (under g_scratch_mutex)
static std::mutex g_scratch_mutex;

// Device address of this GPU's copy of the scratch array; `device` must be the current device.
// Looked up once per device (rtx_scene_create does it; the no-allocation entry point finds it there).
static hipError_t scratch_address(int device, unsigned char** out) {
    *out = nullptr;
    if (device < 0 || device >= RTX_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_at[device]) {
        unsigned char* at = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&at, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_at[device] = at;
    }
    *out = g_scratch_at[device];
    return hipSuccess;
}

namespace {

// ---- BLAS flattening ---------------------------------------------------------------------
struct Flat {
    std::vector<Node4> nodes;
    std::vector<Tri> tris;
    // the eight-wide BLAS: records, triangles in its own order, per-triangle leaf boxes, per-mesh roots
    std::vector<Node8> nodes8;
    std::vector<Tri> tris8;
    std::vector<TriAux> aux;            // parallel to tris8: the exact box of the reference leaf the triangle lies in
    std::vector<std::pair<int32_t, int32_t>> root8;   // per mesh: (index into nodes8 or tris8, triangle count or 0)
    bool b8_ok = true;                  // every leaf holds at most RTX_B8_MAX_LEAF triangles, all coordinates fit a float
    std::vector<double> tri_vn;         // RTX_VN_STRIDE doubles per triangle
    std::vector<Mesh> meshes;
    bool any_vn = false;
    int record_levels = 0;              // deepest chain of records (collapsed nodes + leaf) in any tree
    int blas_levels = 0;                // ... in any BLAS
};

// floats that bracket a double from below / from above
float float_below(double x) {
    float f = (float)x;
    if ((double)f > x) f = std::nextafter(f, -INFINITY);
    return f;
}
float float_above(double x) {
    float f = (float)x;
    if ((double)f < x) f = std::nextafter(f, INFINITY);
    return f;
}

int push_tri(const rtx_scene_desc& d, const rtx_mesh& m, int face, Flat& F) {
    if (face < 0 || face >= m.n_faces) return fail(RTX_EINVAL, "face index outside the mesh");
    const int32_t* f = m.faces + 3 * (size_t)face;
    for (int k = 0; k < 3; k++)
        if (f[k] < 0 || f[k] >= d.n_vertices) return fail(RTX_EINVAL, "face references a vertex outside the pool");
    const double* a = d.vertices + 3 * (size_t)f[0];
    const double* b = d.vertices + 3 * (size_t)f[1];
    const double* c = d.vertices + 3 * (size_t)f[2];
    Tri t;
    memset(&t, 0, sizeof t);
    for (int k = 0; k < 3; k++) { t.a[k] = a[k]; t.e1[k] = b[k] - a[k]; t.e2[k] = c[k] - a[k]; }
    t.face = face;
    // geometricNormal = normalize(crossProduct(e1, e2)), utils.cpp:319-321
    double nx = t.e1[1] * t.e2[2] - t.e1[2] * t.e2[1];
    double ny = t.e1[2] * t.e2[0] - t.e1[0] * t.e2[2];
    double nz = t.e1[0] * t.e2[1] - t.e1[1] * t.e2[0];
    double len = std::sqrt(nx * nx + ny * ny + nz * nz);
    t.gn[0] = nx / len; t.gn[1] = ny / len; t.gn[2] = nz / len;
    F.tris.push_back(t);
    if (m.smooth && m.vertex_normals) {
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 3; q++) F.tri_vn.push_back(m.vertex_normals[3 * (size_t)f[k] + q]);
        for (int q = 9; q < RTX_VN_STRIDE; q++) F.tri_vn.push_back(0.0);
    } else {
        for (int q = 0; q < RTX_VN_STRIDE; q++) F.tri_vn.push_back(0.0);
    }
    return RTX_OK;
}

int flatten_mesh(const rtx_scene_desc& d, const rtx_mesh& m, Flat& F) {
    Mesh out{};
    out.material = m.material; out.smooth = m.smooth; out.xform = m.xform; out.n_faces = m.n_faces;
    out.has_vn = (m.smooth && m.vertex_normals) ? 1 : 0;
    if (out.has_vn) F.any_vn = true;
    if (m.n_faces < 0 || (m.n_faces > 0 && !m.faces)) return fail(RTX_EINVAL, "mesh without faces pointer");
    if (m.n_faces == 0) { F.meshes.push_back(out); F.root8.emplace_back(0, 0); return RTX_OK; }
    // (F.b8_ok: the eight-wide records are wanted and, so far, possible)
    // a leaf's triangles into the eight-wide BLAS's own triangle array (Tri::pad = index in the
    // reference-order array), and beside each the leaf's exact box
    auto copy_leaf8 = [&F](int32_t first, int count, const double* mn, const double* mx) {
        TriAux a;
        memset(&a, 0, sizeof a);
        memcpy(a.mn, mn, 24); memcpy(a.mx, mx, 24);
        // ... and in the record's spare 24 bytes the same box as floats rounded INWARD (on an axis where the
        // leaf is thinner than a float step the two cross; the kernel looks for that)
        float in[6];
        for (int k = 0; k < 3; k++) { in[k] = float_above(mn[k]); in[3 + k] = float_below(mx[k]); }
        for (int k = 0; k < count; k++) {
            Tri t = F.tris[(size_t)first + (size_t)k];
            t.pad = first + k;
            memcpy(t.pad2, in, sizeof in);
            F.tris8.push_back(t);
            F.aux.push_back(a);
        }
    };
    if (m.n_nodes <= 0 || !m.nodes) {
        // No BVH supplied (the -b 0 mode): one leaf holding every face in face order, which is
        // the order of the reference's brute-force loop (utils.cpp:434-441).
        HBox box;
        out.root_ref = (int32_t)F.tris.size();
        out.root_count = m.n_faces;
        for (int f = 0; f < m.n_faces; f++) {
            int rc = push_tri(d, m, f, F);
            if (rc) return rc;
            for (int k = 0; k < 3; k++) box.expand(d.vertices + 3 * (size_t)m.faces[3 * (size_t)f + k]);
        }
        memcpy(out.rmin, box.mn, sizeof box.mn); memcpy(out.rmax, box.mx, sizeof box.mx);
        if (F.b8_ok && m.n_faces <= RTX_B8_MAX_LEAF) { F.root8.emplace_back((int32_t)F.tris8.size(), m.n_faces); copy_leaf8(out.root_ref, m.n_faces, box.mn, box.mx); }
        else { F.root8.emplace_back(0, 0); F.b8_ok = false; }
        F.meshes.push_back(out);
        return RTX_OK;
    }
    if (!m.face_indices) return fail(RTX_EINVAL, "mesh BVH without face_indices");
    const int n = m.n_nodes;
    // triangle ranges for leaves, in node order (= DFS pre-order = the order of faceIndices)
    std::vector<int32_t> leaf_start((size_t)n, -1);
    for (int i = 0; i < n; i++) {
        const rtx_bvh_node& nd = m.nodes[i];
        if (nd.left == -1) {
            if (nd.face_count <= 0 || nd.first_face < 0 || (int64_t)nd.first_face + nd.face_count > m.n_faces)
                return fail(RTX_EINVAL, "BVH leaf range outside face_indices");
            leaf_start[(size_t)i] = (int32_t)F.tris.size();
            for (int k = 0; k < nd.face_count; k++) {
                int rc = push_tri(d, m, m.face_indices[nd.first_face + k], F);
                if (rc) return rc;
            }
            if (nd.face_count > RTX_B8_MAX_LEAF) F.b8_ok = false;
        } else {
            if (nd.left < 0 || nd.left >= n || nd.right < 0 || nd.right >= n || nd.split_axis < 0 || nd.split_axis > 2)
                return fail(RTX_EINVAL, "BVH inner node with invalid children");
        }
    }
    // tree and depth check: a collapsed node pushes up to three entries per two binary levels
    {
        std::vector<std::pair<int, int>> st;
        st.emplace_back(0, 1);
        int max_depth = 0;
        size_t visited = 0;
        while (!st.empty()) {
            auto [i, dep] = st.back(); st.pop_back();
            if (++visited > (size_t)n) return fail(RTX_EINVAL, "BVH is not a tree");
            max_depth = std::max(max_depth, dep);
            if (m.nodes[i].left != -1) { st.emplace_back(m.nodes[i].left, dep + 1); st.emplace_back(m.nodes[i].right, dep + 1); }
        }
        // (max_depth counts the nodes on the longest root-to-leaf path; a record stands for an inner
        // node at an even level, so a path crosses max_depth / 2 records)
        if (3 * (max_depth / 2) > RTX_STACK) return fail(RTX_ELIMIT, "BVH deeper than the traversal stack");
        F.record_levels = std::max(F.record_levels, max_depth / 2 + 1);
        F.blas_levels = std::max(F.blas_levels, max_depth / 2 + 1);
    }
    // collapse two binary levels per record, records in DFS pre-order
    struct Collapse {
        const rtx_mesh& m; const std::vector<int32_t>& leaf_start; Flat& F;
        void slot(Node4& o, int s, int c, int32_t ref, int32_t cnt) {
            memcpy(o.box[s], m.nodes[c].bounds_min, 24); memcpy(o.box[s] + 3, m.nodes[c].bounds_max, 24);
            o.ref[s] = ref; o.cnt[s] = cnt;
        }
        int32_t make(int i) {
            const int32_t id = (int32_t)F.nodes.size();
            F.nodes.emplace_back();
            Node4 o;
            memset(&o, 0, sizeof o);
            for (int s = 0; s < 4; s++) o.cnt[s] = -1;
            const rtx_bvh_node& nd = m.nodes[i];
            o.axis0 = nd.split_axis;
            const int part[2] = {nd.left, nd.right};
            for (int p = 0; p < 2; p++) {
                const rtx_bvh_node& pn = m.nodes[part[p]];
                if (pn.left == -1) { slot(o, 2 * p, part[p], leaf_start[(size_t)part[p]], pn.face_count); continue; }
                (p == 0 ? o.axisL : o.axisR) = pn.split_axis;
                const int kid[2] = {pn.left, pn.right};
                for (int j = 0; j < 2; j++) {
                    const rtx_bvh_node& x = m.nodes[kid[j]];
                    if (x.left == -1) slot(o, 2 * p + j, kid[j], leaf_start[(size_t)kid[j]], x.face_count);
                    else slot(o, 2 * p + j, kid[j], make(kid[j]), 0);
                }
            }
            F.nodes[(size_t)id] = o;
            return id;
        }
    };
    memcpy(out.rmin, m.nodes[0].bounds_min, 24); memcpy(out.rmax, m.nodes[0].bounds_max, 24);
    if (m.nodes[0].left == -1) { out.root_ref = leaf_start[0]; out.root_count = m.nodes[0].face_count; }
    else { Collapse c{m, leaf_start, F}; out.root_ref = c.make(0); out.root_count = 0; }
    // ---- the same tree three levels per record (Node8), breadth first so that a record's inner
    // children are consecutive records; the triangles of its leaf children consecutive in tris8
    if (!F.b8_ok) {
        F.root8.emplace_back(0, 0);
    } else if (m.nodes[0].left == -1) {
        F.root8.emplace_back((int32_t)F.tris8.size(), m.nodes[0].face_count);
        if (m.nodes[0].face_count <= RTX_B8_MAX_LEAF) copy_leaf8(leaf_start[0], m.nodes[0].face_count, m.nodes[0].bounds_min, m.nodes[0].bounds_max);
    } else {
        F.root8.emplace_back((int32_t)F.nodes8.size(), 0);
        std::vector<std::pair<int, int32_t>> todo;          // (reference node, its record)
        todo.emplace_back(0, (int32_t)F.nodes8.size());
        F.nodes8.emplace_back();
        for (size_t at = 0; at < todo.size(); at++) {
            const int ref = todo[at].first;
            const int32_t id = todo[at].second;
            int slot_node[8];
            for (int& x : slot_node) x = -1;
            Node8 o;
            memset(&o, 0, sizeof o);
            const rtx_bvh_node& n0 = m.nodes[ref];
            o.axes = (uint32_t)n0.split_axis;
            const int kid[2] = {n0.left, n0.right};
            for (int a = 0; a < 2; a++) {
                const rtx_bvh_node& n1 = m.nodes[kid[a]];
                if (n1.left == -1) { slot_node[4 * a] = kid[a]; continue; }
                o.axes |= (uint32_t)n1.split_axis << (2 + 2 * a);
                const int kid1[2] = {n1.left, n1.right};
                for (int b = 0; b < 2; b++) {
                    const rtx_bvh_node& n2 = m.nodes[kid1[b]];
                    if (n2.left == -1) { slot_node[4 * a + 2 * b] = kid1[b]; continue; }
                    o.axes |= (uint32_t)n2.split_axis << (6 + 2 * (2 * a + b));
                    slot_node[4 * a + 2 * b] = n2.left;
                    slot_node[4 * a + 2 * b + 1] = n2.right;
                }
            }
            bool first_inner = true, first_leaf = true;
            for (int sl = 0; sl < 8; sl++) {
                for (int k = 0; k < 3; k++) { o.lo[k][sl] = FLT_MAX; o.hi[k][sl] = -FLT_MAX; }
                const int c = slot_node[sl];
                if (c < 0) continue;
                const rtx_bvh_node& nc = m.nodes[c];
                for (int k = 0; k < 3; k++) {
                    if (!(std::fabs(nc.bounds_min[k]) < 1e30) || !(std::fabs(nc.bounds_max[k]) < 1e30)) F.b8_ok = false;
                    o.lo[k][sl] = float_below(nc.bounds_min[k]); o.hi[k][sl] = float_above(nc.bounds_max[k]);
                }
                if (nc.left == -1) {
                    if (first_leaf) { o.tri_base = (uint32_t)F.tris8.size(); first_leaf = false; }
                    o.counts |= (uint32_t)std::min(nc.face_count, 15) << (4 * sl);
                    copy_leaf8(leaf_start[(size_t)c], nc.face_count, nc.bounds_min, nc.bounds_max);
                } else {
                    const int32_t cid = (int32_t)F.nodes8.size();
                    if (first_inner) { o.child_base = (uint32_t)cid; first_inner = false; }
                    o.counts |= 15u << (4 * sl);
                    F.nodes8.emplace_back();
                    todo.emplace_back(c, cid);
                }
            }
            F.nodes8[(size_t)id] = o;
        }
    }
    F.meshes.push_back(out);
    return RTX_OK;
}

// ---- TLAS ---------------------------------------------------------------------------------
struct TlasItem { HBox box; int idx; double c[3]; };

// TLAS over the instances' world boxes, written as Node4 records of the same array the BLAS
// records live in: median split on the widest centroid axis, two binary levels per record.  A
// slot with cnt == -2 is an instance (ref = instance index).  The lower side of a split goes to
// the first slots and the split axes are recorded like a BLAS record's, so that a walk visits the
// near side first (the consistent mode prunes with the bound as it goes; the default mode's walk
// only marks candidates, in any order).
struct TlasBuild {
    std::vector<TlasItem>& items;
    std::vector<Node4>& nodes;

    HBox range_box(int lo, int hi) const {
        HBox b;
        for (int i = lo; i < hi; i++) b.merge(items[(size_t)i].box);
        return b;
    }
    static double half_area(const HBox& b) {
        const double dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return dx * dy + dy * dz + dz * dx;
    }
    // Partition items [lo, hi) in two: binned surface-area heuristic over the three axes, median
    // of the widest centroid axis when no binned split separates the items or `sah` is off.  The
    // TLAS only selects candidates, so its shape is free (unlike a BLAS, which is the reference's
    // own tree).  The walk pushes up to three entries per record level: the heuristic is used
    // for the first kSahLevels levels and the median below, which bounds the depth at
    // kSahLevels + 16 levels (96 entries of the RTX_STACK-entry stack).
    static constexpr int kSahLevels = 16;
    int split(int lo, int hi, bool sah, int32_t& axis_out) {
        double cmn[3] = {1e300, 1e300, 1e300}, cmx[3] = {-1e300, -1e300, -1e300};
        for (int i = lo; i < hi; i++)
            for (int k = 0; k < 3; k++) { cmn[k] = std::min(cmn[k], items[(size_t)i].c[k]); cmx[k] = std::max(cmx[k], items[(size_t)i].c[k]); }
        int axis = 0;
        for (int k = 1; k < 3; k++) if (cmx[k] - cmn[k] > cmx[axis] - cmn[axis]) axis = k;
        constexpr int kBins = 16;
        int best_axis = -1, best_bin = -1;
        double best_cost = 1e300;
        if (sah && hi - lo > 4) {
            for (int k = 0; k < 3; k++) {
                const double w = cmx[k] - cmn[k];
                if (!(w > 0.0) || !(w < 1e299)) continue;
                HBox bb[kBins]; int bn[kBins] = {};
                const double scale = kBins / w;
                for (int i = lo; i < hi; i++) {
                    const int b = std::min(kBins - 1, std::max(0, (int)((items[(size_t)i].c[k] - cmn[k]) * scale)));
                    bb[b].merge(items[(size_t)i].box); bn[b]++;
                }
                double right_area[kBins]; int right_n[kBins];
                HBox acc; int n = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.merge(bb[b]); n += bn[b]; right_area[b] = n ? half_area(acc) : 0.0; right_n[b] = n; }
                HBox left; int ln = 0;
                for (int b = 0; b + 1 < kBins; b++) {
                    left.merge(bb[b]); ln += bn[b];
                    if (ln == 0 || right_n[b + 1] == 0) continue;
                    const double cost = half_area(left) * ln + right_area[b + 1] * right_n[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = b; }
                }
            }
        }
        if (best_axis >= 0) {
            const int k = best_axis;
            const double scale = kBins / (cmx[k] - cmn[k]), c0 = cmn[k];
            const int bin = best_bin;
            auto mid = std::partition(items.begin() + lo, items.begin() + hi, [=](const TlasItem& a) {
                return std::min(kBins - 1, std::max(0, (int)((a.c[k] - c0) * scale))) <= bin; });
            const int m = (int)(mid - items.begin());
            if (m > lo && m < hi) { axis_out = k; return m; }
        }
        const int mid = lo + (hi - lo) / 2;
        std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                         [axis](const TlasItem& a, const TlasItem& b) { return a.c[axis] < b.c[axis]; });
        axis_out = axis;
        return mid;
    }
    void slot(Node4& o, int s, const HBox& box, int32_t ref, int32_t cnt) {
        memcpy(o.box[s], box.mn, 24); memcpy(o.box[s] + 3, box.mx, 24);
        o.ref[s] = ref; o.cnt[s] = cnt;
    }
    void fill(Node4& o, int s, int lo, int hi, int level) {       // one slot for items [lo, hi)
        if (hi - lo == 1) slot(o, s, items[(size_t)lo].box, items[(size_t)lo].idx, -2);
        else { const int32_t child = make(lo, hi, level + 1); slot(o, s, range_box(lo, hi), child, 0); }
    }
    int max_level = 0;
    int32_t make(int lo, int hi, int level = 0) {       // record for items [lo, hi), at least one
        max_level = std::max(max_level, level);
        const int32_t id = (int32_t)nodes.size();
        nodes.emplace_back();
        Node4 o;
        memset(&o, 0, sizeof o);
        for (int s = 0; s < 4; s++) o.cnt[s] = -1;
        const bool sah = level < kSahLevels;
        if (hi - lo == 1) fill(o, 0, lo, hi, level);
        else {
            const int mid = split(lo, hi, sah, o.axis0);
            const int part[3] = {lo, mid, hi};
            for (int p = 0; p < 2; p++) {
                const int a = part[p], b = part[p + 1];
                if (b - a == 1) { fill(o, 2 * p, a, b, level); continue; }
                const int m2 = split(a, b, sah, p == 0 ? o.axisL : o.axisR);
                fill(o, 2 * p, a, m2, level);
                fill(o, 2 * p + 1, m2, b, level);
            }
        }
        nodes[(size_t)id] = o;
        return id;
    }
};

int check_index(int v, int n, bool allow_neg, const char* what) {
    if ((v < 0 && !(allow_neg && v == -1)) || v >= n) return fail(RTX_EINVAL, std::string("index out of range: ") + what);
    return RTX_OK;
}

int build_scene(rtx_scene& sc, const rtx_scene_desc& d) {
    if (d.abi_version != RTX_ABI_VERSION) return fail(RTX_EINVAL, "rtx_scene_desc.abi_version mismatch");
    if (d.max_recursion_depth < 0 || d.max_recursion_depth > 250) return fail(RTX_ELIMIT, "max_recursion_depth outside 0..250");
    if (d.n_vertices < 0 || (d.n_vertices > 0 && !d.vertices)) return fail(RTX_EINVAL, "vertices");
    auto need = [](const void* p, int n) { return n < 0 || (n > 0 && !p); };
    if (need(d.materials, d.n_materials) || need(d.lights, d.n_lights) || need(d.xforms, d.n_xforms) ||
        need(d.planes, d.n_planes) || need(d.spheres, d.n_spheres) || need(d.triangles, d.n_triangles) ||
        need(d.meshes, d.n_meshes) || need(d.instances, d.n_instances))
        return fail(RTX_EINVAL, "null array with non-zero count");

    Scene& S = sc.dev;
    memset(&S, 0, sizeof S);
    memcpy(S.bg, d.background, 24); memcpy(S.ambient, d.ambient, 24);
    S.shadow_eps = d.shadow_ray_epsilon; S.isect_eps = d.intersection_test_epsilon;
    S.max_depth = d.max_recursion_depth; S.cull = d.enable_back_face_culling ? 1 : 0;
    S.n_lights = d.n_lights; S.n_planes = d.n_planes; S.n_spheres = d.n_spheres; S.n_ftris = d.n_triangles;
    S.n_meshes = d.n_meshes; S.n_insts = 0;
    sc.n_meshes = d.n_meshes;

    std::vector<Material> mats((size_t)d.n_materials);
    for (int i = 0; i < d.n_materials; i++) {
        const rtx_material& m = d.materials[i];
        Material& o = mats[(size_t)i];
        memcpy(o.ambient, m.ambient, 24); memcpy(o.diffuse, m.diffuse, 24); memcpy(o.specular, m.specular, 24);
        memcpy(o.mirror, m.mirror, 24); memcpy(o.sigma, m.absorption_coefficient, 24);
        o.phong = m.phong_exponent; o.n = m.refraction_index; o.k = m.absorption_index;
        o.type = m.type; o.is_mirror = m.is_mirror;
    }
    std::vector<Light> lights((size_t)d.n_lights);
    for (int i = 0; i < d.n_lights; i++) { memcpy(lights[(size_t)i].p, d.lights[i].position, 24); memcpy(lights[(size_t)i].i, d.lights[i].intensity, 24); }
    std::vector<Xform> xfs((size_t)d.n_xforms);
    for (int i = 0; i < d.n_xforms; i++) {
        const rtx_xform& x = d.xforms[i];
        Xform& o = xfs[(size_t)i];
        memcpy(o.m, x.transform, 128); memcpy(o.inv, x.inverse, 128); memcpy(o.nrm, x.normal, 128);
        memcpy(o.bmin, x.bounds_min, 24); memcpy(o.bmax, x.bounds_max, 24);
        o.has_bounds = x.has_bounds; o.pad = 0;
    }
    int rc;
    std::vector<Plane> planes((size_t)d.n_planes);
    for (int i = 0; i < d.n_planes; i++) {
        const rtx_plane& p = d.planes[i];
        if ((rc = check_index(p.point, (int)d.n_vertices, false, "plane point"))) return rc;
        if ((rc = check_index(p.material, d.n_materials, true, "plane material"))) return rc;
        if ((rc = check_index(p.xform, d.n_xforms, true, "plane xform"))) return rc;
        memcpy(planes[(size_t)i].n, p.normal, 24); memcpy(planes[(size_t)i].a, d.vertices + 3 * (size_t)p.point, 24);
        planes[(size_t)i].material = p.material; planes[(size_t)i].xform = p.xform;
    }
    std::vector<Sphere> spheres((size_t)d.n_spheres);
    for (int i = 0; i < d.n_spheres; i++) {
        const rtx_sphere& p = d.spheres[i];
        if ((rc = check_index(p.center, (int)d.n_vertices, false, "sphere center"))) return rc;
        if ((rc = check_index(p.material, d.n_materials, true, "sphere material"))) return rc;
        if ((rc = check_index(p.xform, d.n_xforms, true, "sphere xform"))) return rc;
        memcpy(spheres[(size_t)i].c, d.vertices + 3 * (size_t)p.center, 24);
        spheres[(size_t)i].r = p.radius; spheres[(size_t)i].material = p.material; spheres[(size_t)i].xform = p.xform;
    }
    std::vector<FreeTri> ftris((size_t)d.n_triangles);
    for (int i = 0; i < d.n_triangles; i++) {
        const rtx_triangle& p = d.triangles[i];
        for (int k = 0; k < 3; k++) if ((rc = check_index(p.v[k], (int)d.n_vertices, false, "triangle vertex"))) return rc;
        if ((rc = check_index(p.material, d.n_materials, true, "triangle material"))) return rc;
        if ((rc = check_index(p.xform, d.n_xforms, true, "triangle xform"))) return rc;
        const double* a = d.vertices + 3 * (size_t)p.v[0];
        const double* b = d.vertices + 3 * (size_t)p.v[1];
        const double* c = d.vertices + 3 * (size_t)p.v[2];
        FreeTri& o = ftris[(size_t)i];
        for (int k = 0; k < 3; k++) { o.a[k] = a[k]; o.e1[k] = b[k] - a[k]; o.e2[k] = c[k] - a[k]; }
        double nx = o.e1[1] * o.e2[2] - o.e1[2] * o.e2[1];
        double ny = o.e1[2] * o.e2[0] - o.e1[0] * o.e2[2];
        double nz = o.e1[0] * o.e2[1] - o.e1[1] * o.e2[0];
        double len = std::sqrt(nx * nx + ny * ny + nz * nz);
        o.gn[0] = nx / len; o.gn[1] = ny / len; o.gn[2] = nz / len;
        o.material = p.material; o.xform = p.xform;
    }
    Flat F;
    F.b8_ok = (d.build_flags & RTX_SCENE_WIDE_BLAS) != 0;
    sc.b8_built = F.b8_ok;
    for (int i = 0; i < d.n_meshes; i++) {
        const rtx_mesh& m = d.meshes[i];
        if ((rc = check_index(m.material, d.n_materials, true, "mesh material"))) return rc;
        if ((rc = check_index(m.xform, d.n_xforms, true, "mesh xform"))) return rc;
        if ((rc = flatten_mesh(d, m, F))) return rc;
    }
    if (F.tris.size() > (size_t)INT_MAX / 2) return fail(RTX_ELIMIT, "too many triangles");
    { Tri pad_tri; memset(&pad_tri, 0, sizeof pad_tri); F.tris.push_back(pad_tri); F.tris8.push_back(pad_tri); }   // leaf steps fetch triangles in pairs
    { TriAux pad_aux; memset(&pad_aux, 0, sizeof pad_aux); F.aux.push_back(pad_aux); }
    // instances: effective transform (utils.cpp:418,446,476) and TLAS over their world boxes
    std::vector<Inst> insts((size_t)d.n_instances);
    std::vector<TlasItem> items;
    for (int i = 0; i < d.n_instances; i++) {
        const rtx_instance& in = d.instances[i];
        Inst& o = insts[(size_t)i];
        o.base_mesh = in.base_mesh; o.material = in.material; o.xform = -1; o.pad = 0;
        if (in.base_mesh < 0) continue;
        if ((rc = check_index(in.base_mesh, d.n_meshes, false, "instance base mesh"))) return rc;
        if ((rc = check_index(in.material, d.n_materials, true, "instance material"))) return rc;
        if ((rc = check_index(in.xform, d.n_xforms, true, "instance xform"))) return rc;
        const Mesh& bm = F.meshes[(size_t)in.base_mesh];
        o.xform = in.xform >= 0 ? in.xform : bm.xform;
        if (o.xform >= 0) {
            const Xform& x = xfs[(size_t)o.xform];
            // (to rounding: an inverse computed numerically has a last row of (0, 0, 0, 1) only up to that)
            auto near = [](double v, double want) { return std::fabs(v - want) <= 1e-12; };
            if (!(near(x.inv[12], 0.0) && near(x.inv[13], 0.0) && near(x.inv[14], 0.0) && near(x.inv[15], 1.0) &&
                  near(x.m[12], 0.0) && near(x.m[13], 0.0) && near(x.m[14], 0.0) && near(x.m[15], 1.0))) sc.affine_insts = false;
        }
        if (bm.n_faces <= 0) continue;      // cannot produce a hit or touch the record
        TlasItem it;
        it.idx = i;
        if (o.xform >= 0 && xfs[(size_t)o.xform].has_bounds) {
            memcpy(it.box.mn, xfs[(size_t)o.xform].bmin, 24); memcpy(it.box.mx, xfs[(size_t)o.xform].bmax, 24);
        } else if (o.xform < 0) {
            memcpy(it.box.mn, bm.rmin, 24); memcpy(it.box.mx, bm.rmax, 24);
        } else {
            for (int k = 0; k < 3; k++) { it.box.mn[k] = -1e300; it.box.mx[k] = 1e300; }
        }
        for (int k = 0; k < 3; k++) it.c[k] = 0.5 * it.box.mn[k] + 0.5 * it.box.mx[k];
        items.push_back(it);
    }
    int32_t tlas_root = -1;
    if (!items.empty()) {
        TlasBuild tb{items, F.nodes};
        const HBox all = tb.range_box(0, (int)items.size());
        memcpy(S.tl_min, all.mn, 24); memcpy(S.tl_max, all.mx, 24);
        tlas_root = tb.make(0, (int)items.size());
        // the walk pushes up to three entries per record level, like a BLAS traversal
        if (3 * (tb.max_level + 1) > RTX_STACK) return fail(RTX_ELIMIT, "instance hierarchy deeper than the traversal stack");
        F.record_levels = std::max(F.record_levels, tb.max_level + 1);
        sc.nest_stack_need = 3 * (tb.max_level + 1) + 3 * F.blas_levels;
        S.n_insts = d.n_instances;
    }
    // record indices -> keys (offsets in 128-byte units into the joint node + triangle allocation)
    if (2 * F.nodes.size() + F.tris.size() > (size_t)INT_MAX) return fail(RTX_ELIMIT, "BLAS larger than 256 GB");
    const int32_t tri_key0 = (int32_t)(2 * F.nodes.size());
    auto to_key = [&](int32_t& ref, int32_t cnt) { if (cnt > 0) ref += tri_key0; else if (cnt == 0) ref *= 2; };
    for (Node4& n : F.nodes) for (int k = 0; k < 4; k++) to_key(n.ref[k], n.cnt[k]);
    for (Mesh& m : F.meshes) if (m.n_faces > 0) to_key(m.root_ref, m.root_count);
    S.tri_key0 = tri_key0;

    S.tl_key = tlas_root >= 0 ? 2 * tlas_root : 0;
    sc.record_levels = F.record_levels;

    // per-object records: meshes first, then instances
    std::vector<Obj> objs((size_t)d.n_meshes + (size_t)d.n_instances);
    auto fill_obj = [&](Obj& o, int mesh, int xform, int material) {
        memset(&o, 0, sizeof o);
        o.mesh = mesh; o.material = material; o.flags = OBJ_SKIP;
        if (mesh < 0) return;
        const Mesh& bm = F.meshes[(size_t)mesh];
        memcpy(o.rmin, bm.rmin, 24); memcpy(o.rmax, bm.rmax, 24);
        o.root_ref = bm.root_ref; o.root_count = bm.root_count;
        o.flags = bm.n_faces > 0 ? 0 : OBJ_SKIP;
        if (bm.smooth && bm.has_vn) o.flags |= OBJ_SMOOTH_VN;
        if (xform >= 0) {
            const Xform& x = xfs[(size_t)xform];
            o.flags |= OBJ_XF;
            if (x.has_bounds) o.flags |= OBJ_BOUNDS;
            auto near = [](double v, double want) { return std::fabs(v - want) <= 1e-12; };
            if (near(x.inv[12], 0.0) && near(x.inv[13], 0.0) && near(x.inv[14], 0.0) && near(x.inv[15], 1.0) &&
                near(x.m[12], 0.0) && near(x.m[13], 0.0) && near(x.m[14], 0.0) && near(x.m[15], 1.0)) o.flags |= OBJ_AFFINE;
            memcpy(o.inv, x.inv, 128); memcpy(o.m, x.m, 128); memcpy(o.nrm, x.nrm, 96);
            memcpy(o.bmin, x.bmin, 24); memcpy(o.bmax, x.bmax, 24);
        }
    };
    for (int i = 0; i < d.n_meshes; i++) fill_obj(objs[(size_t)i], i, F.meshes[(size_t)i].xform, F.meshes[(size_t)i].material);
    // tiny meshes (see RTX_LDS_TINY): untransformed, one leaf of at most two triangles, among the records kept in LDS
    for (int i = 0, used = 0; i < d.n_meshes && i < RTX_LDS_OBJS && used < RTX_LDS_TINY; i++) {
        Obj& o = objs[(size_t)i];
        if (!(o.flags & (OBJ_SKIP | OBJ_XF)) && o.root_count >= 1 && o.root_count <= 2) o.flags |= (++used) << 8;
    }
    for (int i = 0; i < d.n_instances; i++)
        fill_obj(objs[(size_t)d.n_meshes + (size_t)i], insts[(size_t)i].base_mesh, insts[(size_t)i].xform, insts[(size_t)i].material);

    HIP_TRY(sc.materials.upload(mats));   S.materials = (const Material*)sc.materials.p;
    {
        auto ls = std::make_shared<LightSet>();
        HIP_TRY(ls->buf.upload(lights));
        ls->n = d.n_lights;
        sc.lights = ls; sc.light_pool.push_back(ls);
        S.lights = (const Light*)ls->buf.p;
    }
    HIP_TRY(sc.xforms.upload(xfs));       S.xforms = (const Xform*)sc.xforms.p;
    HIP_TRY(sc.planes.upload(planes));    S.planes = (const Plane*)sc.planes.p;
    HIP_TRY(sc.spheres.upload(spheres));  S.spheres = (const Sphere*)sc.spheres.p;
    HIP_TRY(sc.ftris.upload(ftris));      S.ftris = (const FreeTri*)sc.ftris.p;
    HIP_TRY(sc.meshes.upload(F.meshes));  S.meshes = (const Mesh*)sc.meshes.p;
    HIP_TRY(sc.insts.upload(insts));      S.insts = (const Inst*)sc.insts.p;
    {
        // (the object records share the allocation so that the traversal's record fetch reaches them;
        // behind them the eight-wide BLAS: its records, then its triangles in its own order)
        const size_t nb = F.nodes.size() * sizeof(Node4), tb = F.tris.size() * sizeof(Tri), ob = objs.size() * sizeof(Obj);
        const bool b8 = F.b8_ok && F.root8.size() == (size_t)d.n_meshes;
        const size_t n8b = b8 ? F.nodes8.size() * sizeof(Node8) : 0, t8b = b8 ? F.tris8.size() * sizeof(Tri) : 0;
        if ((nb + tb + ob + n8b + t8b) / 128 > (size_t)INT_MAX) return fail(RTX_ELIMIT, "BLAS larger than 256 GB");
        if (b8) {
            // record and triangle indices -> keys
            const uint32_t n8_key0 = (uint32_t)((nb + tb + ob) / 128), t8_key0 = (uint32_t)((nb + tb + ob + n8b) / 128);
            S.t8_key0 = (int32_t)t8_key0;
            for (Node8& n : F.nodes8) { n.child_base = n8_key0 + 2u * n.child_base; n.tri_base = t8_key0 + n.tri_base; }
            for (int i = 0; i < d.n_meshes; i++) {
                const auto& r = F.root8[(size_t)i];
                objs[(size_t)i].root8_ref = (int32_t)(r.second > 0 ? t8_key0 + (uint32_t)r.first : n8_key0 + 2u * (uint32_t)r.first);
            }
            for (int i = 0; i < d.n_instances; i++)
                if (insts[(size_t)i].base_mesh >= 0) objs[(size_t)d.n_meshes + (size_t)i].root8_ref = objs[(size_t)insts[(size_t)i].base_mesh].root8_ref;
        }
        HIP_TRY(sc.blas.ensure(std::max<size_t>(nb + tb + ob + n8b + t8b, 16)));
        if (nb) HIP_TRY(hipMemcpy(sc.blas.p, F.nodes.data(), nb, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy((char*)sc.blas.p + nb, F.tris.data(), tb, hipMemcpyHostToDevice));
        if (ob) HIP_TRY(hipMemcpy((char*)sc.blas.p + nb + tb, objs.data(), ob, hipMemcpyHostToDevice));
        if (n8b) HIP_TRY(hipMemcpy((char*)sc.blas.p + nb + tb + ob, F.nodes8.data(), n8b, hipMemcpyHostToDevice));
        if (t8b) HIP_TRY(hipMemcpy((char*)sc.blas.p + nb + tb + ob + n8b, F.tris8.data(), t8b, hipMemcpyHostToDevice));
        S.blas = (const char*)sc.blas.p;
        S.obj_key0 = (int32_t)((nb + tb) / 128);
        S.objs = (const Obj*)(S.blas + nb + tb);
        sc.b8_ok = b8;
        S.tri_aux = nullptr;
        if (b8) { HIP_TRY(sc.tri_aux.upload(F.aux)); S.tri_aux = (const TriAux*)sc.tri_aux.p; }
    }
    if (F.any_vn) { HIP_TRY(sc.tri_vn.upload(F.tri_vn)); S.tri_vn = (const double*)sc.tri_vn.p; }
    else {
        // no smooth mesh: keep has_vn = 0 everywhere, nothing to upload
        S.tri_vn = nullptr;
    }
    return RTX_OK;
}

struct TileGeom { int tiles_x, tiles_y, n_tiles, n_sel; int64_t out_bytes; };

int tile_geometry(const rtx_camera* cam, int tw, int th, int first, int stride, TileGeom& g) {
    if (!cam || cam->width <= 0 || cam->height <= 0) return fail(RTX_EINVAL, "camera resolution");
    if (tw <= 0 || th <= 0 || stride <= 0 || first < 0) return fail(RTX_EINVAL, "tile parameters");
    g.tiles_x = (cam->width + tw - 1) / tw; g.tiles_y = (cam->height + th - 1) / th;
    g.n_tiles = g.tiles_x * g.tiles_y;
    g.n_sel = first >= g.n_tiles ? 0 : (g.n_tiles - first + stride - 1) / stride;
    g.out_bytes = (int64_t)g.n_sel * tw * th * 3;
    return RTX_OK;
}

Queue make_queue(void* base, uint32_t cap) {
    Queue q;
    q.rays = (Ray*)base; q.sub_capacity = cap / RTX_NSUB; q.pad = 0;
    return q;
}
constexpr size_t kRayBytes = sizeof(Ray);
constexpr size_t kHitBytes = sizeof(Hit);

HitBuf make_hitbuf(void* base, size_t) {
    HitBuf h;
    h.recs = (Hit*)base;
    return h;
}

}  // namespace

// rtx_bvh_build.hip
namespace rtxbuild {
void set_error(const std::string& s) { g_error = s; }
int build(int device, const double* vertices, int64_t n_vertices, const int32_t* faces, int32_t n_faces,
          rtx_bvh_node* nodes_out, int32_t* n_nodes_out, int32_t* face_indices_out, double* build_ms);
}  // namespace rtxbuild

extern "C" {

const char* rtx_last_error(void) { return g_error.c_str(); }

int rtx_build_bvh(int32_t device, const double* vertices, int64_t n_vertices, const int32_t* faces, int32_t n_faces,
                  rtx_bvh_node* nodes_out, int32_t* n_nodes_out, int32_t* face_indices_out, double* build_ms) {
    return rtxbuild::build(device, vertices, n_vertices, faces, n_faces, nodes_out, n_nodes_out, face_indices_out, build_ms);
}

int rtx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int64_t rtx_render_output_bytes(const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride) {
    TileGeom g;
    if (tile_geometry(cam, tw, th, first, stride, g)) return -1;
    return g.out_bytes;
}

int rtx_unpack_tiles(const uint8_t* packed, int32_t width, int32_t height, int32_t tw, int32_t th,
                     int32_t first, int32_t stride, uint8_t* frame) {
    if (!packed || !frame) return fail(RTX_EINVAL, "null argument");
    rtx_camera cam{};
    cam.width = width; cam.height = height;
    TileGeom g;
    int rc = tile_geometry(&cam, tw, th, first, stride, g);
    if (rc) return rc;
    for (int k = 0; k < g.n_sel; k++) {
        int tile = first + k * stride;
        int x0 = (tile % g.tiles_x) * tw, y0 = (tile / g.tiles_x) * th;
        const uint8_t* src = packed + (size_t)k * tw * th * 3;
        for (int y = 0; y < th && y0 + y < height; y++) {
            int w = std::min(tw, width - x0);
            memcpy(frame + ((size_t)(y0 + y) * width + x0) * 3, src + (size_t)y * tw * 3, (size_t)w * 3);
        }
    }
    return RTX_OK;
}

namespace {
#define RTX_MAX_PARTS 64
struct TileParts { const uint8_t* p[RTX_MAX_PARTS]; };
// One thread per pixel slot of every tile: tile t comes from part t % n_parts, where it is the
// (t / n_parts)-th packed tile.
__global__ void k_assemble_tiles(const TileParts parts, const int n_parts, const int width, const int height, const int tw,
                                 const int th, const int tiles_x, const long long n_slots, uint8_t* __restrict__ frame) {
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_slots) return;
    const int per_tile = tw * th;
    const int tile = (int)(id / per_tile), within = (int)(id % per_tile);
    const int x = (tile % tiles_x) * tw + within % tw, y = (tile / tiles_x) * th + within / tw;
    if (x >= width || y >= height) return;
    const uint8_t* src = parts.p[tile % n_parts] + ((size_t)(tile / n_parts) * per_tile + within) * 3;
    uint8_t* dst = frame + ((size_t)y * width + x) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}
std::mutex g_assemble_mutex;
DevBuf g_assemble_frame[RTX_MAX_DEVICES];
}  // namespace

int rtx_assemble_tiles(int32_t device, const uint8_t* const* parts, int32_t n_parts, int32_t width, int32_t height,
                       int32_t tw, int32_t th, uint8_t* frame_out, int32_t frame_is_device) {
    if (!parts || !frame_out) return fail(RTX_EINVAL, "null argument");
    if (n_parts < 1 || n_parts > RTX_MAX_PARTS) return fail(RTX_EINVAL, "between 1 and 64 tile buffers");
    if (device < 0 || device >= RTX_MAX_DEVICES) return fail(RTX_EINVAL, "bad device");
    rtx_camera cam{};
    cam.width = width; cam.height = height;
    TileGeom g;
    int rc = tile_geometry(&cam, tw, th, 0, 1, g);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(device));
    TileParts tp{};
    for (int r = 0; r < n_parts; r++) {
        // (a part without tiles -- more ranks than tiles -- may be null)
        if (!parts[r] && r < g.n_tiles) return fail(RTX_EINVAL, "null tile buffer");
        // Every buffer must be device memory of `device` as this process's HIP runtime knows it: a
        // pointer the runtime has never seen (host memory, another GPU, memory of a second copy of
        // the runtime in the process) is refused here instead of being handed to a kernel.
        if (parts[r]) {
            hipPointerAttribute_t attr;
            memset(&attr, 0, sizeof attr);
            const hipError_t e = hipPointerGetAttributes(&attr, parts[r]);
            if (e != hipSuccess || attr.type != hipMemoryTypeDevice || attr.device != device) {
                (void)hipGetLastError();
                char msg[256];
                snprintf(msg, sizeof msg, "rtx_assemble_tiles: parts[%d] is not device memory of GPU %d known to this HIP runtime (%s)",
                         r, (int)device, e != hipSuccess ? hipGetErrorString(e) : attr.type != hipMemoryTypeDevice ? "not device memory" : "another device");
                return fail(RTX_EINVAL, msg);
            }
        }
        tp.p[r] = parts[r];
    }
    const size_t bytes = (size_t)width * height * 3;
    const long long n_slots = (long long)g.n_tiles * tw * th;
    if (n_slots > (long long)0x7FFFFFFF * 256) return fail(RTX_ELIMIT, "frame too large");
    std::lock_guard<std::mutex> lock(g_assemble_mutex);
    uint8_t* frame = frame_out;
    if (!frame_is_device) {
        HIP_TRY(g_assemble_frame[device].ensure(bytes));
        frame = (uint8_t*)g_assemble_frame[device].p;
    }
    hipLaunchKernelGGL(k_assemble_tiles, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, 0, tp, n_parts, width, height,
                       tw, th, g.tiles_x, n_slots, frame);
    HIP_TRY(hipGetLastError());
    if (!frame_is_device) HIP_TRY(hipMemcpy(frame_out, frame, bytes, hipMemcpyDeviceToHost));
    else HIP_TRY(hipStreamSynchronize(0));
    return RTX_OK;
}

}  // extern "C"


// ---- frames ---------------------------------------------------------------------------------
//
// A frame context (rtx_frame) owns everything one render call needs besides the scene itself: a
// stream, event pairs, page-locked staging, ray queues, hit records, traversal stacks, instance
// bitmaps, pixel sums and control blocks.  The scene's geometry, TLAS and object records are
// shared read-only, so several frames of one uploaded scene can be in flight at once: the last
// generations of a frame are a few long rays in an otherwise idle GPU, and a neighbouring
// frame's kernels fill that time.  rtx_render is "submit + wait" on the scene's own frame.
namespace {

// Device memory one frame works in: the frame context's own growing buffers, or pieces of the
// fixed scratch array (rtx_render_noalloc).
struct FrameMem {
    void *accum = nullptr, *rgb = nullptr, *ctrl = nullptr, *counters = nullptr, *aov = nullptr, *stacks = nullptr;
    void *launch = nullptr, *hitbuf = nullptr, *queue[2] = {nullptr, nullptr};
    unsigned* inst_bits = nullptr;
    uint32_t queue_capacity = 0;      // rays per queue
    size_t hit_capacity = 0;
};

using TraceKernel = void (*)(const Launch*, GenCtrl*, int, double*, rtx_hit_id*, Counters*);

// One submitted render: the arguments, what was derived from them, the scene state it was
// submitted under, and how far it has got.
struct Job {
    rtx_camera cam{};
    int32_t tw = 0, th = 0, first = 0, stride = 1;
    uint8_t* rgb_out = nullptr;
    int32_t rgb_is_device = 0;
    rtx_hit_id* hit_ids = nullptr;
    bool noalloc = false;
    bool stage_rgb = false;         // the picture comes down into the frame's page-locked staging and is copied out at the wait
    TileGeom g{};
    Frame F{};
    int G = 0;
    int64_t n_items0 = 0, n_local_px = 0, queue_want = 0;
    unsigned persistent_blocks = 0;
    uint32_t inst_words = 0;
    size_t accum_bytes = 0, ctrl_bytes = 0, stack_bytes = 0, bits_bytes = 0;
    FrameMem M;
    // the scene as it was when the frame was submitted
    Scene dev{};
    std::shared_ptr<LightSet> lights;
    bool counters_on = false, consistent = false, timing = true, wide_blas = false;
    uint32_t debug = 0;
    // progress
    uint32_t overflows = 0, trace_launches = 0, instance_path = 0;
    float kernel_ms = 0.f, trace_ms = 0.f;
    int g_first = 0;
    bool search = false, two_variants = false;
    bool no_b8 = false;             // the eight-wide BLAS is not to be used (this render is a redo after its check failed)
    bool b8 = false;                // this attempt runs the eight-wide BLAS's kernels
    TraceKernel trace_main = nullptr, trace_tiny = nullptr;
    std::chrono::high_resolution_clock::time_point wall0;
};

}  // namespace

struct rtx_frame {
    rtx_scene* sc = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> gen_events;   // 2 per generation + 1: brackets of the trace launches
    // page-locked staging, sized at creation: the frame's Launch block on its way to the device,
    // the control blocks and counters on their way back (no pageable copies, nothing allocated per frame)
    Launch* h_launch = nullptr;
    GenCtrl* h_ctrl = nullptr;
    Counters* h_counters = nullptr;
    // ... and the picture's, when the caller's buffer is pageable host memory: a copy into such memory
    // is staged by the runtime and holds the submitting thread until the frame is done
    uint8_t* h_stage = nullptr;
    size_t h_stage_bytes = 0;
    DevBuf accum, rgb, aov, ctrl, counters, queue[2], hitbuf, launch, stacks, inst_bits;
    size_t inst_bits_cleared = 0;   // bytes of inst_bits known to be all zero
    uint32_t queue_capacity = 0;    // rays per queue, grows on overflow and is then kept
    int gens_hint = 0;              // generations the next render launches before it looks whether more are needed (0: all)
    bool pending = false;           // a render has been submitted and not yet waited for
    Job job;
    ~rtx_frame() {
        for (hipEvent_t e : gen_events) (void)hipEventDestroy(e);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
        if (h_launch) (void)hipHostFree(h_launch);
        if (h_ctrl) (void)hipHostFree(h_ctrl);
        if (h_counters) (void)hipHostFree(h_counters);
        if (h_stage) (void)hipHostFree(h_stage);
    }
};

namespace {

// Development switches (rtx_scene_set_debug).  A scene starts with the ones named in the process
// environment when its first scene is created (RTX_AMD_NO_SCATTER, RTX_AMD_POISON_QUEUES,
// RTX_AMD_GEN_TIMING, RTX_AMD_NO_WIDE, RTX_AMD_NO_SEARCH, RTX_AMD_NO_WIDE_OLD,
// RTX_AMD_ALL_GENERATIONS, RTX_AMD_HALF_GRID: read once per process); nothing reads the environment per frame.
uint32_t debug_from_environment() {
    static const uint32_t flags = [] {
        uint32_t f = 0;
        if (getenv("RTX_AMD_NO_SCATTER")) f |= RTX_DEBUG_NO_SCATTER;
        if (getenv("RTX_AMD_POISON_QUEUES")) f |= RTX_DEBUG_POISON_QUEUES;
        if (getenv("RTX_AMD_GEN_TIMING")) f |= RTX_DEBUG_GEN_TIMING;
        if (getenv("RTX_AMD_NO_WIDE")) f |= RTX_DEBUG_NO_WIDE;
        if (getenv("RTX_AMD_NO_SEARCH")) f |= RTX_DEBUG_NO_SEARCH;
        if (getenv("RTX_AMD_NO_WIDE_OLD")) f |= RTX_DEBUG_NO_WIDE_OLD;
        if (getenv("RTX_AMD_ALL_GENERATIONS")) f |= RTX_DEBUG_ALL_GENERATIONS;
        if (getenv("RTX_AMD_HALF_GRID")) f |= RTX_DEBUG_HALF_GRID;
        if (getenv("RTX_AMD_NO_EARLY_SHADOW")) f |= RTX_DEBUG_NO_EARLY_SHADOW;
        return f;
    }();
    return flags;
}

int frame_create(rtx_scene* sc, rtx_frame** out) {
    HIP_TRY(hipSetDevice(sc->device));
    std::unique_ptr<rtx_frame> fr(new rtx_frame());
    fr->sc = sc;
    // everything a render needs besides device memory is created here, once
    const size_t G = (size_t)sc->dev.max_depth + 2;
    HIP_TRY(hipStreamCreateWithFlags(&fr->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&fr->ev0));
    HIP_TRY(hipEventCreate(&fr->ev1));
    for (size_t k = 0; k < 2 * G + 1; k++) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        fr->gen_events.push_back(ev);
    }
    HIP_TRY(hipHostMalloc((void**)&fr->h_launch, sizeof(Launch), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&fr->h_ctrl, (G + 1) * sizeof(GenCtrl), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&fr->h_counters, G * sizeof(Counters), hipHostMallocDefault));
    {
        std::lock_guard<std::mutex> lock(sc->frames_mutex);
        sc->frames.push_back(fr.get());
    }
    *out = fr.release();
    return RTX_OK;
}

void frame_destroy(rtx_frame* fr) {
    rtx_scene* sc = fr->sc;
    (void)hipSetDevice(sc->device);
    if (fr->stream) (void)hipStreamSynchronize(fr->stream);
    {
        std::lock_guard<std::mutex> lock(sc->frames_mutex);
        sc->frames.erase(std::remove(sc->frames.begin(), sc->frames.end(), fr), sc->frames.end());
        if (sc->own == fr) sc->own = nullptr;
    }
    delete fr;
}

// Synchronise the stream when a call leaves early: copies into the caller's buffers and into the
// frame's staging memory may still be in flight.
struct StreamGuard {
    rtx_frame* fr; bool armed = true;
    ~StreamGuard() { if (armed) { (void)hipStreamSynchronize(fr->stream); fr->pending = false; fr->job.lights.reset(); } }
};

// Arguments -> everything that does not change between the attempts of one render; device memory.
int job_prepare(rtx_frame* fr, const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride,
                uint8_t* rgb_out, int32_t rgb_is_device, rtx_hit_id* hit_ids, const bool noalloc, bool& nothing_to_do) {
    rtx_scene* sc = fr->sc;
    nothing_to_do = false;
    if (!cam || !rgb_out) return fail(RTX_EINVAL, "null argument");
    Job& J = fr->job;
    J = Job();
    int rc = tile_geometry(cam, tw, th, first, stride, J.g);
    if (rc) return rc;
    const TileGeom& g = J.g;
    const bool whole = (g.n_tiles == 1 && first == 0);
    if (hit_ids && !whole) return fail(RTX_EINVAL, "hit_ids are only available for whole-frame renders");
    if (sc->counters_on && sc->consistent) return fail(RTX_EINVAL, "work counters are not available in consistent mode");
    J.cam = *cam; J.tw = tw; J.th = th; J.first = first; J.stride = stride;
    J.rgb_out = rgb_out; J.rgb_is_device = rgb_is_device; J.hit_ids = hit_ids; J.noalloc = noalloc;
    J.wall0 = std::chrono::high_resolution_clock::now();
    if (g.n_sel == 0) { nothing_to_do = true; return RTX_OK; }
    HIP_TRY(hipSetDevice(sc->device));
    // the scene's state as of now: flags, and the current version of the lights
    J.counters_on = sc->counters_on; J.consistent = sc->consistent; J.debug = sc->debug; J.wide_blas = sc->wide_blas;
    J.timing = sc->generation_timing || (sc->debug & RTX_DEBUG_GEN_TIMING);
    {
        // (rtx_scene_update_lights writes the scene's light fields under the same lock)
        std::lock_guard<std::mutex> lock(sc->lights_mutex);
        J.dev = sc->dev;
        J.lights = sc->lights;
    }
    J.dev.lights = (const Light*)J.lights->buf.p;
    J.dev.n_lights = J.lights->n;

    // camera basis: utils.cpp:135-146, evaluated once per frame instead of once per pixel
    Frame& F = J.F;
    {
        auto norm = [](const double* a, double* o) {
            double len = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
            o[0] = a[0] / len; o[1] = a[1] / len; o[2] = a[2] / len;
        };
        double w[3], v[3], u[3], m[3];
        norm(cam->gaze, w);
        for (double& x : w) x = -x;
        norm(cam->up, v);
        u[0] = v[1] * w[2] - v[2] * w[1]; u[1] = v[2] * w[0] - v[0] * w[2]; u[2] = v[0] * w[1] - v[1] * w[0];
        const double l = cam->near_plane[0], r = cam->near_plane[1], b = cam->near_plane[2], t = cam->near_plane[3];
        for (int k = 0; k < 3; k++) {
            m[k] = cam->position[k] - w[k] * cam->near_distance;
            F.q[k] = (m[k] + l * u[k]) + t * v[k];
            F.e[k] = cam->position[k]; F.u[k] = u[k]; F.v[k] = v[k];
        }
        F.r_minus_l = r - l; F.t_minus_b = t - b;
        F.width = cam->width; F.height = cam->height;
        F.tile_w = tw; F.tile_h = th; F.tiles_x = g.tiles_x; F.tile_first = first; F.tile_stride = stride;
    }
    const int64_t blocks_per_tile = (int64_t)((tw + 7) / 8) * ((th + 7) / 8);
    J.n_items0 = (int64_t)g.n_sel * blocks_per_tile * 64;
    J.n_local_px = (int64_t)g.n_sel * tw * th;
    if (J.n_items0 > (int64_t)0x7FFFFFF0 || J.n_local_px * 3 > (int64_t)0x7FFFFFF0) return fail(RTX_ELIMIT, "frame too large for one call");
    F.n_items0 = (int32_t)J.n_items0;
    {
        const uint64_t nb = (uint64_t)(J.n_items0 >> RTX_SCATTER0_SHIFT);
        uint64_t mul = 1;
        if (nb > 8 && !(J.debug & RTX_DEBUG_NO_SCATTER)) {
            auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
            mul = (uint64_t)((double)nb * 0.6180339887498949) | 1u;
            while (gcd(mul, nb) != 1) mul += 2;
            mul %= nb;
        }
        F.block_mul = (uint32_t)mul; F.n_blocks = (uint32_t)nb;
    }

    J.G = J.dev.max_depth + 2;
    const int G = J.G;
    // One grid of resident blocks: RTX_TRACE_WAVES waves per SIMD = that many 256-thread blocks per CU.
    J.persistent_blocks = (unsigned)sc->cus * (unsigned)RTX_TRACE_WAVES;
    if (J.debug & RTX_DEBUG_HALF_GRID) J.persistent_blocks = (unsigned)sc->cus;       // diagnostic: one wave per SIMD
    if (noalloc) J.persistent_blocks = std::min(J.persistent_blocks, RTX_NOALLOC_BLOCKS);
    // one bit per instance per lane; every ray leaves its lane's bitmap clear
    J.inst_words = rtx_bitmap_words(J.dev.n_insts) + rtx_summary_words(J.dev.n_insts);
    J.accum_bytes = (size_t)J.n_local_px * 3 * sizeof(double);
    J.ctrl_bytes = (size_t)(G + 1) * sizeof(GenCtrl);
    J.stack_bytes = (size_t)J.persistent_blocks * 256u * RTX_STACK * 16u;
    J.bits_bytes = (size_t)J.persistent_blocks * 256u * J.inst_words * sizeof(uint32_t);
    J.queue_want = std::max<int64_t>(J.n_local_px * (int64_t)(J.dev.n_lights + 2), 1 << 12);
    if (fr->gen_events.size() < (size_t)2 * G + 1) return fail(RTX_EHIP, "event pool smaller than the generation count");

    FrameMem& M = J.M;
    if (!noalloc) {
        HIP_TRY(fr->accum.ensure(J.accum_bytes));                   M.accum = fr->accum.p;
        HIP_TRY(fr->rgb.ensure((size_t)J.n_local_px * 3));          M.rgb = fr->rgb.p;
        HIP_TRY(fr->ctrl.ensure(J.ctrl_bytes));                     M.ctrl = fr->ctrl.p;
        HIP_TRY(fr->counters.ensure((size_t)G * sizeof(Counters))); M.counters = fr->counters.p;   // one block per generation
        if (hit_ids) { HIP_TRY(fr->aov.ensure((size_t)cam->width * cam->height * sizeof(rtx_hit_id))); M.aov = fr->aov.p; }
        HIP_TRY(fr->stacks.ensure(J.stack_bytes));                  M.stacks = fr->stacks.p;
        HIP_TRY(fr->launch.ensure(sizeof(Launch)));                 M.launch = fr->launch.p;
        if (J.bits_bytes > ((size_t)8 << 30)) return fail(RTX_ELIMIT, "too many instances for the per-lane instance bitmaps");
        if (J.bits_bytes > fr->inst_bits.bytes || fr->inst_bits_cleared < J.bits_bytes) {
            HIP_TRY(fr->inst_bits.ensure(J.bits_bytes));
            HIP_TRY(hipMemsetAsync(fr->inst_bits.p, 0, fr->inst_bits.bytes, fr->stream));    // zeroed once
            fr->inst_bits_cleared = fr->inst_bits.bytes;
        }
        M.inst_bits = (unsigned*)fr->inst_bits.p;
        if (!rgb_is_device) {
            // page-locked (or otherwise known to the runtime): the final copy is a plain DMA into it
            hipPointerAttribute_t attr;
            memset(&attr, 0, sizeof attr);
            const hipError_t e = hipPointerGetAttributes(&attr, rgb_out);
            if (e != hipSuccess) (void)hipGetLastError();
            J.stage_rgb = e != hipSuccess || attr.type == hipMemoryTypeUnregistered;
            const size_t need = (size_t)J.n_local_px * 3;
            if (J.stage_rgb && fr->h_stage_bytes < need) {
                if (fr->h_stage) { (void)hipHostFree(fr->h_stage); fr->h_stage = nullptr; fr->h_stage_bytes = 0; }
                HIP_TRY(hipHostMalloc((void**)&fr->h_stage, need, hipHostMallocDefault));
                fr->h_stage_bytes = need;
            }
        }
        if (fr->queue_capacity == 0) fr->queue_capacity = (uint32_t)std::min<int64_t>(std::max<int64_t>(J.queue_want, 1 << 16), 0x7FFFFFF0);
    } else {
        // carve the frame's memory out of this GPU's scratch array; nothing is allocated
        if (hit_ids) return fail(RTX_EINVAL, "the no-allocation entry point returns no hit_ids");
        unsigned char* base = nullptr;
        if (scratch_address(sc->device, &base) != hipSuccess || !base)
            return fail(RTX_EHIP, "the scratch array's device address is not known for this GPU");
        size_t at = 0;
        auto take = [&](size_t n) { unsigned char* p = base + at; at += (n + 255) & ~(size_t)255; return (void*)p; };
        M.launch = take(sizeof(Launch));
        M.ctrl = take(J.ctrl_bytes);
        M.counters = take((size_t)G * sizeof(Counters));
        M.accum = take(J.accum_bytes);
        M.rgb = take((size_t)J.n_local_px * 3);
        M.stacks = take(J.stack_bytes);
        M.inst_bits = (unsigned*)take(J.bits_bytes);
        // the rest: two ray queues (96 B per ray each) and the hit records (128 B per ray)
        const int64_t need = std::max<int64_t>(J.queue_want, J.n_items0);
        const int64_t room = (int64_t)RTX_NOALLOC_SCRATCH_BYTES - (int64_t)at;
        const int64_t cap = room > 0 ? (room / (int64_t)(2 * kRayBytes + kHitBytes)) / RTX_NSUB * RTX_NSUB : 0;
        if (cap < need || at > RTX_NOALLOC_SCRATCH_BYTES) {
            char msg[256];
            snprintf(msg, sizeof msg, "rtx_render_noalloc: %lld pixels with %d lights and %d instances need %lld bytes of scratch, "
                     "the fixed array has %u; render fewer tiles per call or use rtx_render",
                     (long long)J.n_local_px, J.dev.n_lights, J.dev.n_insts,
                     (long long)at + need * (long long)(2 * kRayBytes + kHitBytes), RTX_NOALLOC_SCRATCH_BYTES);
            return fail(RTX_ELIMIT, msg);
        }
        M.queue_capacity = (uint32_t)std::min<int64_t>(cap, 0x7FFFFFF0);
        M.hit_capacity = (size_t)M.queue_capacity;
        M.queue[0] = take((size_t)M.queue_capacity * kRayBytes);
        M.queue[1] = take((size_t)M.queue_capacity * kRayBytes);
        M.hitbuf = take(M.hit_capacity * kHitBytes);
        // the array keeps the last call's contents (possibly another scene's): clear the candidate bitmaps
        HIP_TRY(hipMemsetAsync(M.inst_bits, 0, J.bits_bytes, fr->stream));
    }
    return RTX_OK;
}

// Launch generations [g0, g1) with their shading, the resolve pass and the copies that bring the
// results down; nothing is waited for.
int job_enqueue(rtx_frame* fr, const int g0, const int g1) {
    Job& J = fr->job;
    const FrameMem& M = J.M;
    const int G = J.G;
    const Launch* lp = (const Launch*)M.launch;
    std::vector<hipEvent_t>& gev = fr->gen_events;
    HIP_TRY(hipEventRecord(fr->ev0, fr->stream));
    for (int gen = g0; gen < g1; gen++) {
        if (J.timing) HIP_TRY(hipEventRecord(gev[2 * gen], fr->stream));
        unsigned blocks = J.persistent_blocks;
        if (gen == 0) blocks = (unsigned)std::min<int64_t>(J.persistent_blocks, (J.n_items0 + 255) / 256);
        rtx_hit_id* aov = (gen == 0 && J.hit_ids) ? (rtx_hit_id*)M.aov : nullptr;
        hipLaunchKernelGGL(J.trace_main, dim3(blocks), dim3(256), 0, fr->stream, lp, (GenCtrl*)M.ctrl, gen,
                           (double*)M.accum, aov, J.counters_on ? (Counters*)M.counters + gen : (Counters*)nullptr);
        // the variant for tiny generations (one of the two returns at once; see k_trace)
        if (J.two_variants)
            hipLaunchKernelGGL(J.trace_tiny, dim3(blocks), dim3(256), 0, fr->stream, lp, (GenCtrl*)M.ctrl, gen,
                               (double*)M.accum, aov, (Counters*)nullptr);
        if (J.timing) HIP_TRY(hipEventRecord(gev[2 * gen + 1], fr->stream));
        // the last generation holds shadow rays only: nothing left to shade
        if (gen + 1 < G)
            hipLaunchKernelGGL(k_shade, dim3(blocks), dim3(256), 0, fr->stream, lp, (GenCtrl*)M.ctrl, gen,
                               (double*)M.accum);
    }
    const unsigned n_values = (unsigned)(J.n_local_px * 3);
    hipLaunchKernelGGL(k_resolve, dim3((n_values + 255) / 256), dim3(256), 0, fr->stream,
                       (const double*)M.accum, (uint8_t*)M.rgb, n_values);
    HIP_TRY(hipEventRecord(gev[2 * G], fr->stream));
    HIP_TRY(hipEventRecord(fr->ev1, fr->stream));
    HIP_TRY(hipGetLastError());
    // The results follow the kernels down the stream before it is known whether a ray queue
    // overflowed (that is rare, and the frame is then rendered and copied again): one wait
    // for kernels, control block and results together.
    HIP_TRY(hipMemcpyAsync(J.stage_rgb ? fr->h_stage : J.rgb_out, M.rgb, (size_t)J.n_local_px * 3,
                           J.rgb_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, fr->stream));
    if (J.hit_ids && g0 == 0)
        HIP_TRY(hipMemcpyAsync(J.hit_ids, M.aov, (size_t)J.cam.width * J.cam.height * sizeof(rtx_hit_id), hipMemcpyDeviceToHost, fr->stream));
    if (J.counters_on) HIP_TRY(hipMemcpyAsync(fr->h_counters, M.counters, (size_t)G * sizeof(Counters), hipMemcpyDeviceToHost, fr->stream));
    HIP_TRY(hipMemcpyAsync(fr->h_ctrl, M.ctrl, J.ctrl_bytes, hipMemcpyDeviceToHost, fr->stream));
    return RTX_OK;
}

// Wait for what job_enqueue(g0, g1) put on the stream and add up its times.
int job_collect(rtx_frame* fr, const int g0, const int g1) {
    Job& J = fr->job;
    HIP_TRY(hipStreamSynchronize(fr->stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, fr->ev0, fr->ev1));
    J.kernel_ms += ms;
    for (int gen = g0; gen < g1 && J.timing; gen++) {
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, fr->gen_events[2 * gen], fr->gen_events[2 * gen + 1]));
        J.trace_ms += t;
    }
    J.trace_launches += (uint32_t)(g1 - g0);
    return RTX_OK;
}

// One attempt at the frame with the queues as large as they are now: queues, control blocks and
// pixel sums set up, the kernel variants chosen, the first generations launched.
int job_start_attempt(rtx_frame* fr) {
    rtx_scene* sc = fr->sc;
    Job& J = fr->job;
    FrameMem& M = J.M;
    const int G = J.G;
    if (!J.noalloc) {
        if ((int64_t)fr->queue_capacity * (int64_t)kRayBytes * 2 > sc->queue_limit) {
            uint32_t cap = (uint32_t)(sc->queue_limit / (2 * (int64_t)kRayBytes));
            if (J.overflows > 0 && cap <= fr->queue_capacity / 2 + 1)
                return fail(RTX_ENOMEM, "ray queues exceed the configured limit; render fewer tiles per call or raise the limit");
            fr->queue_capacity = cap;
        }
        HIP_TRY(fr->queue[0].ensure((size_t)fr->queue_capacity * kRayBytes));
        HIP_TRY(fr->queue[1].ensure((size_t)fr->queue_capacity * kRayBytes));
        M.hit_capacity = std::max<size_t>(fr->queue_capacity, (size_t)J.n_items0);
        HIP_TRY(fr->hitbuf.ensure(M.hit_capacity * kHitBytes));
        M.queue[0] = fr->queue[0].p; M.queue[1] = fr->queue[1].p; M.hitbuf = fr->hitbuf.p;
        M.queue_capacity = fr->queue_capacity;
    }
    if (J.debug & RTX_DEBUG_POISON_QUEUES) {      // test aid: queues and hit records start as 0xFF
        HIP_TRY(hipMemsetAsync(M.queue[0], 0xFF, (size_t)M.queue_capacity * kRayBytes, fr->stream));
        HIP_TRY(hipMemsetAsync(M.queue[1], 0xFF, (size_t)M.queue_capacity * kRayBytes, fr->stream));
        HIP_TRY(hipMemsetAsync(M.hitbuf, 0xFF, M.hit_capacity * kHitBytes, fr->stream));
    }
    const bool gen_timing = (J.debug & RTX_DEBUG_GEN_TIMING) != 0;     // per-generation kernel times on stderr
    // page-locked, so the copy below is a plain DMA; the previous render's (or attempt's) copy of the
    // block finished before that render's final wait returned
    Launch& launch = *fr->h_launch;
    launch.S = J.dev; launch.F = J.F;
    launch.S.limit_flag = &((GenCtrl*)M.ctrl)->limit;
    launch.q[0] = make_queue(M.queue[0], M.queue_capacity);
    launch.q[1] = make_queue(M.queue[1], M.queue_capacity);
    launch.hb = make_hitbuf(M.hitbuf, M.hit_capacity);
    launch.stacks = M.stacks;
    launch.inst_bits = M.inst_bits; launch.inst_words = J.inst_words;
    // wave-wide traversal of a drained generation's last jobs: its visiting-order keys hold 26 record levels
    launch.wide_ok = (sc->record_levels <= 26 && !(J.debug & RTX_DEBUG_NO_WIDE)) ? 1u : 0u;
    // Instances entered from the walk of the TLAS, nearest first: always in consistent mode; in the
    // default mode as a search (rtx_kernels.h, search_service) where instances are many.  The
    // wave-wide traversal knows one job per ray, so these kernels do without it.
    // (with few instances of large meshes -- marching_dragons: eight -- the replay with its wave-wide
    // finishing is the faster of the two in either mode: 4.0 against 4.2 ms in consistent mode)
    const bool nest = J.consistent && !J.counters_on && J.dev.n_insts >= RTX_SEARCH_MIN_INSTS && sc->affine_insts &&
                      sc->nest_stack_need <= RTX_STACK;
    const bool search = !J.consistent && !J.counters_on && J.dev.n_insts >= RTX_SEARCH_MIN_INSTS && sc->affine_insts &&
                        sc->nest_stack_need <= RTX_SEARCH_STACK && !sc->search_unprofitable.load() && !(J.debug & RTX_DEBUG_NO_SEARCH);
    if (nest || search) launch.wide_ok = 0u;
    J.search = search;
    J.instance_path = nest ? 2u : search ? 1u : 0u;
    // A service pass lasts as long as its slowest lane.  Replaying instance candidates makes
    // that long (a chain of dependent loads per candidate), so scenes whose many instances are
    // replayed gather more lanes per pass and claim rays in larger chunks.
    const bool replay_heavy = J.dev.n_insts >= 256 && !nest && !search;
    launch.service_threshold = replay_heavy ? RTX_SERVICE_THRESHOLD_INSTANCED : RTX_SERVICE_THRESHOLD;
    if (gen_timing) fprintf(stderr, "[rtx] instances %d: nearest-first walk %s (affine %d, stack need %d)\n", J.dev.n_insts,
                            nest ? "consistent" : search ? "search" : "off", (int)sc->affine_insts, sc->nest_stack_need);
    // ... and, in scenes that are not dominated by instance replay, of the long-running jobs of
    // every generation's drained waves (a kernel variant of its own; see k_trace)
    launch.chunk_max = replay_heavy ? RTX_CHUNK_INSTANCED : RTX_CHUNK;
    // shadow rays that no object can hold up are settled by the shading kernel (shadow_ray_is_free), in the default
    // and the consistent mode (whose bound, the light's distance, turns down nothing that no bound lets in); with
    // counters on every query is carried out
    {
        const int n_lin = J.dev.n_meshes + (J.dev.n_insts <= RTX_LINEAR_INSTS ? J.dev.n_insts : 0);
        launch.early_shadow = (!J.counters_on && !(J.debug & RTX_DEBUG_NO_EARLY_SHADOW) &&
                               J.dev.n_planes == 0 && J.dev.n_spheres == 0 && J.dev.n_ftris == 0 && n_lin <= 16 &&
                               J.dev.n_lights <= 32) ? 1u : 0u;
    }
    launch.wide_old = (launch.wide_ok && J.dev.n_insts < 256 && !(J.debug & RTX_DEBUG_NO_WIDE_OLD)) ? 1u : 0u;
    HIP_TRY(hipMemcpyAsync(M.launch, &launch, sizeof launch, hipMemcpyHostToDevice, fr->stream));

    HIP_TRY(hipMemsetAsync(M.ctrl, 0, J.ctrl_bytes, fr->stream));
    HIP_TRY(hipMemsetAsync(M.accum, 0, J.accum_bytes, fr->stream));
    if (J.counters_on) HIP_TRY(hipMemsetAsync(M.counters, 0, (size_t)G * sizeof(Counters), fr->stream));
    // every generation's trace launches (main variant and the one for tiny generations) are
    // bracketed by events: rtx_stats.trace_ms is the sum of their durations
    // The eight-wide BLAS (rtx_device_types.h, Node8): scenes whose instances are few enough to be taken
    // one after the other, without a walk of the TLAS (that walk and the candidate marks work on the
    // four-wide records).  Its drained waves hand their last jobs to the wave-wide traversal of the
    // four-wide records, where they start over with what they have found as their bound.
    J.b8 = J.wide_blas && sc->b8_ok && !J.no_b8 && !J.counters_on && !nest && !search && J.dev.n_insts <= RTX_LINEAR_INSTS;
    J.trace_main =
        J.b8 ? (launch.wide_old ? (J.consistent ? k_trace<false, 1, true, 0, true> : k_trace<false, 1, false, 0, true>)
                                : (J.consistent ? k_trace<false, 0, true, 0, true> : k_trace<false, 0, false, 0, true>))
        : J.counters_on ? k_trace<true, 0, false>
        : nest ? k_trace<false, 0, true, 1>
        : search ? k_trace<false, 0, false, 2>
        : J.consistent ? (launch.wide_old ? k_trace<false, 1, true> : k_trace<false, 0, true>)
                       : (launch.wide_old ? k_trace<false, 1, false> : k_trace<false, 0, false>);
    J.trace_tiny = J.consistent ? k_trace<false, 2, true> : k_trace<false, 2, false>;
    J.two_variants = !J.counters_on && launch.wide_ok && !launch.wide_old;
    // Generations after a frame's last bounce are empty, and an empty generation still costs its
    // launches (about 20 microseconds with the gaps round them; four to six of them in a scene of
    // diffuse surfaces).  The frame therefore launches one generation more than the previous render
    // of this frame context had rays for, and looks at the next generation's queue when it has the
    // control blocks back: only if that queue holds rays after all are the remaining generations
    // launched, and the picture resolved and copied, a second time.  (An empty queue means empty
    // queues ever after: nothing is skipped that would have done anything.)
    J.g_first = G;
    if (fr->gens_hint > 0 && !J.counters_on && !gen_timing && !(J.debug & RTX_DEBUG_ALL_GENERATIONS)) J.g_first = std::min(G, fr->gens_hint);
    return job_enqueue(fr, 0, J.g_first);
}

void print_generation_timing(rtx_frame* fr) {
    Job& J = fr->job;
    const GenCtrl* ctrl = fr->h_ctrl;
    const int G = J.G;
    std::vector<hipEvent_t>& gev = fr->gen_events;
    for (int gen = 0; gen < G; gen++) {
        float t_trace = 0.f, t_shade = 0.f;
        (void)hipEventElapsedTime(&t_trace, gev[2 * gen], gev[2 * gen + 1]);
        (void)hipEventElapsedTime(&t_shade, gev[2 * gen + 1], gev[2 * gen + 2]);
        unsigned long long c = (unsigned long long)J.n_items0;
        if (gen > 0) { c = 0; for (const SubCount& sub : ctrl[gen].sub) c += sub.counts; }
        fprintf(stderr, "[rtx] gen %d: radiance %u shadow %u trace %.1f us shade (+resolve) %.1f us\n", gen,
                (unsigned)c, (unsigned)(c >> 32), t_trace * 1e3, t_shade * 1e3);
    }
    fprintf(stderr, "[rtx] rays handed back to the replay by the search: list full %u, k turned down %u, another triangle %u, "
            "not accepted %u, accepted after k %u\n", ctrl[0].search_back[0], ctrl[0].search_back[1], ctrl[0].search_back[2],
            ctrl[0].search_back[3], ctrl[0].search_back[4]);
    if (RTX_WAVE_TALLY)
        for (int gen = 0; gen < G; gen++) {
            const unsigned long long sum = *(const unsigned long long*)&ctrl[gen].tally[8], packed = *(const unsigned long long*)&ctrl[gen].tally[10];
            const unsigned long long mx = (packed >> 38) << 6;
            const unsigned n = ctrl[gen].tally[12];
            if (n && RTX_WAVE_TALLY == 4) {
                const unsigned long long* t = (const unsigned long long*)&ctrl[gen].tally[0];
                fprintf(stderr, "[rtx] gen %d: %u waves, mean lifetime %.1f us, longest %.1f us; of all waves' lifetimes %.1f %% traversal steps, %.1f %% service passes "
                        "(%.1f per wave, %.2f us each), %.1f %% wave-wide jobs\n", gen, n, (double)sum / n / 2400.0, (double)mx / 2400.0, 100.0 * (double)t[0] / (double)sum,
                        100.0 * (double)t[1] / (double)sum, (double)t[3] / n, t[3] ? (double)t[1] / (double)t[3] / 2400.0 : 0.0, 100.0 * (double)t[2] / (double)sum);
            } else
            if (n) fprintf(stderr, "[rtx] gen %d: %u waves, mean lifetime %.1f us, longest %.1f us (at 2.4 GHz; %u %% of it in %u wave-wide jobs, %u %% in traversal steps, %u %% in %u service passes): filled %.0f %%; "
                           "waves by lifetime in 150 us bins: %u %u %u %u %u %u %u %u+; after the queue ran dry: mean %.1f us, longest %.1f us\n", gen, n,
                           (double)sum / n / 2400.0, (double)mx / 2400.0, (unsigned)(packed >> 31) & 127u, (unsigned)(packed >> 9) & 255u,
                           (unsigned)(packed >> 24) & 127u, (unsigned)(packed >> 17) & 127u, (unsigned)packed & 511u, 100.0 * (double)sum / n / (double)mx,
                           ctrl[gen].tally[0], ctrl[gen].tally[1], ctrl[gen].tally[2], ctrl[gen].tally[3], ctrl[gen].tally[4],
                           ctrl[gen].tally[5], ctrl[gen].tally[6], ctrl[gen].tally[7],
                           (double)*(const unsigned long long*)&ctrl[gen].tally[14] / n / 2400.0, (double)ctrl[gen].tally[13] * 256.0 / 2400.0);
        }
    if (RTX_SEARCH_TALLY)
        fprintf(stderr, "[rtx] search: instance jobs ended %u (with a hit %u), instances met by the walk %u, walks ended %u (with a k %u), "
                "probe jobs ended %u (found %u), probes tried %u\n", ctrl[0].tally[0], ctrl[0].tally[1], ctrl[0].tally[2], ctrl[0].tally[3],
                ctrl[0].tally[4], ctrl[0].tally[5], ctrl[0].tally[6], ctrl[0].tally[7]);
}

void print_counter_timing(const Job& J, const Counters* per_gen, const Counters& ctr) {
    for (int gen = 0; gen < J.G; gen++) {
        const Counters& c = per_gen[(size_t)gen];
        fprintf(stderr, "[rtx] gen %d: %llu rays, %llu wave-steps, %llu service passes; longest wave %.1f us "
                "(at 2.4 GHz): %llu steps, %llu service passes\n", gen, c.rays, c.inner_steps + c.leaf_steps,
                c.service_passes, (double)c.max_wave_ticks / 2400.0, c.max_wave_steps & 0xFFFFFFFFull, c.max_wave_steps >> 32);
        fprintf(stderr, "[rtx]        longest ray: %llu steps, %s, local pixel %llu\n", c.longest_ray >> 32,
                (c.longest_ray & 0x80000000ull) ? "shadow" : "radiance", c.longest_ray & 0x7FFFFFFFull);
        const double nst = (double)std::max(1ull, c.inner_steps + c.leaf_steps);
        fprintf(stderr, "[rtx]        ticks per wave-step: %.0f (fetch %.0f, pop %.0f); per service pass: %.0f; "
                "mean wave lifetime %.1f us (traversing %.1f, servicing %.1f)\n",
                (double)c.clk_traverse / nst, (double)c.clk_fetch / nst, (double)c.clk_pop / nst,
                (double)c.clk_service / std::max(1ull, c.service_passes),
                (double)c.clk_total / std::max(1ull, c.waves) / 2400.0, (double)c.clk_traverse / std::max(1ull, c.waves) / 2400.0,
                (double)c.clk_service / std::max(1ull, c.waves) / 2400.0);
        fprintf(stderr, "[rtx]        rays by steps (<16 <32 <64 <128 <256 <512 <1024 more): %llu %llu %llu %llu %llu %llu %llu %llu\n",
                c.ray_steps_hist[0], c.ray_steps_hist[1], c.ray_steps_hist[2], c.ray_steps_hist[3], c.ray_steps_hist[4],
                c.ray_steps_hist[5], c.ray_steps_hist[6], c.ray_steps_hist[7]);
        fprintf(stderr, "[rtx]        of which shadow rays: %llu %llu %llu %llu %llu %llu %llu %llu\n",
                c.shadow_steps_hist[0], c.shadow_steps_hist[1], c.shadow_steps_hist[2], c.shadow_steps_hist[3],
                c.shadow_steps_hist[4], c.shadow_steps_hist[5], c.shadow_steps_hist[6], c.shadow_steps_hist[7]);
    }
    fprintf(stderr, "[rtx] trace waves %llu: ticks/wave total %.0f traverse %.0f service %.0f; service passes %llu "
            "(%.0f ticks each), traversal steps %llu (%.0f ticks each; fetch %.0f, pop %.0f)\n", ctr.waves,
            (double)ctr.clk_total / std::max(1ull, ctr.waves), (double)ctr.clk_traverse / std::max(1ull, ctr.waves),
            (double)ctr.clk_service / std::max(1ull, ctr.waves), ctr.service_passes,
            (double)ctr.clk_service / std::max(1ull, ctr.service_passes), ctr.inner_steps + ctr.leaf_steps,
            (double)ctr.clk_traverse / std::max(1ull, ctr.inner_steps + ctr.leaf_steps),
            (double)ctr.clk_fetch / std::max(1ull, ctr.inner_steps + ctr.leaf_steps),
            (double)ctr.clk_pop / std::max(1ull, ctr.inner_steps + ctr.leaf_steps));
    static const char* names[4] = {"begin", "finalize", "next-job", "complete"};
    for (int k = 0; k < 4; k++)
        fprintf(stderr, "[rtx]   service %-9s %10llu lane-calls, %8.0f ticks each\n", names[k], ctr.section[4 + k],
                (double)ctr.section[k] / std::max(1ull, ctr.section[4 + k]));
}

int frame_submit(rtx_frame* fr, const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride,
                 uint8_t* rgb_out, int32_t rgb_is_device, rtx_hit_id* hit_ids, const bool noalloc) {
    if (!fr) return fail(RTX_EINVAL, "null frame");
    if (fr->pending) return fail(RTX_EINVAL, "the frame has a render in flight: wait for it first");
    bool nothing = false;
    int rc = job_prepare(fr, cam, tw, th, first, stride, rgb_out, rgb_is_device, hit_ids, noalloc, nothing);
    if (rc) { fr->job.lights.reset(); return rc; }
    fr->pending = true;
    if (nothing) { fr->job.G = 0; return RTX_OK; }       // (no tile selected: the wait returns empty statistics)
    StreamGuard guard{fr};
    rc = job_start_attempt(fr);
    if (rc) return rc;
    guard.armed = false;
    return RTX_OK;
}

int frame_wait(rtx_frame* fr, rtx_stats* stats) {
    if (!fr) return fail(RTX_EINVAL, "null frame");
    if (!fr->pending) return fail(RTX_EINVAL, "the frame has no render in flight");
    rtx_scene* sc = fr->sc;
    Job& J = fr->job;
    if (stats) memset(stats, 0, sizeof *stats);
    if (J.G == 0) { fr->pending = false; return RTX_OK; }
    HIP_TRY(hipSetDevice(sc->device));
    StreamGuard guard{fr};
    const int G = J.G;
    const bool gen_timing = (J.debug & RTX_DEBUG_GEN_TIMING) != 0;
    const GenCtrl* ctrl = fr->h_ctrl;
    int rc;
    while (true) {
        if ((rc = job_collect(fr, 0, J.g_first))) return rc;
        if (J.g_first < G) {
            uint64_t pending = 0;
            bool stopped = false;
            for (const SubCount& sub : ctrl[J.g_first].sub) pending += (sub.counts & 0xFFFFFFFFull) + (sub.counts >> 32);
            for (int k = 0; k <= G; k++) stopped |= ctrl[k].overflow != 0 || ctrl[k].limit != 0;
            if (pending != 0 && !stopped) {
                if ((rc = job_enqueue(fr, J.g_first, G))) return rc;
                if ((rc = job_collect(fr, J.g_first, G))) return rc;
            }
        }
        {
            // what the next render of this frame context starts with: one generation more than this one had rays for
            int used = 1;
            for (int k = 1; k < G; k++) {
                uint64_t n = 0;
                for (const SubCount& sub : ctrl[k].sub) n += (sub.counts & 0xFFFFFFFFull) + (sub.counts >> 32);
                if (n != 0) used = k + 1;
            }
            fr->gens_hint = std::min(G, used + 1);
        }
        if (J.search) {
            uint64_t back = 0, rays = (uint64_t)J.n_items0;
            for (unsigned b : ctrl[0].search_back) back += b;
            for (int k = 1; k <= G; k++)
                for (const SubCount& sub : ctrl[k].sub) rays += (sub.counts & 0xFFFFFFFFull) + (sub.counts >> 32);
            if (back * 10 > rays) sc->search_unprofitable.store(true);
        }
        if (gen_timing) print_generation_timing(fr);
        bool overflow = false, limit = false;
        uint32_t peak = 0;
        for (int k = 0; k <= G; k++) {
            const GenCtrl& c = ctrl[k];
            overflow |= c.overflow != 0;
            limit |= c.limit != 0;
            // the fullest sub-queue decides: all sub-queues have the same size
            for (const SubCount& sub : c.sub)
                peak = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(peak, ((sub.counts & 0xFFFFFFFFull) + (sub.counts >> 32)) * RTX_NSUB), 0x7FFFFFF0ull);
        }
        if (limit) return fail(RTX_ELIMIT, "a traversal stack or frontier overflowed on the device: the picture would be incomplete");
        if (J.b8 && (ctrl[0].inexact != 0 || (J.debug & RTX_DEBUG_FAIL_WIDE_BLAS_CHECK))) {
            // a triangle was accepted in a leaf the reference's exact box test may keep the ray out of: the
            // frame again, with the four-wide kernels (whose visited leaves are the reference's own)
            J.no_b8 = true;
            sc->b8_fallbacks.fetch_add(1);
            if ((rc = job_start_attempt(fr))) return rc;
            continue;
        }
        if (!overflow) break;
        J.overflows++;
        if (J.noalloc) {
            char msg[200];
            snprintf(msg, sizeof msg, "rtx_render_noalloc: the ray queues (%u rays each, fixed scratch) overflowed; "
                     "render fewer tiles per call or use rtx_render", J.M.queue_capacity);
            return fail(RTX_ENOMEM, msg);
        }
        // (an overflowed generation stops the frame, so `peak` is a lower bound of what is needed: at least double)
        int64_t want = std::max<int64_t>((int64_t)peak + (peak >> 3), (int64_t)fr->queue_capacity * 2);
        fr->queue_capacity = (uint32_t)std::min<int64_t>(want, 0x7FFFFFF0);
        if (J.overflows > 8) return fail(RTX_ENOMEM, "ray queues kept overflowing");
        if ((rc = job_start_attempt(fr))) return rc;
    }
    guard.armed = false;
    fr->pending = false;
    J.lights.reset();
    if (J.stage_rgb) memcpy(J.rgb_out, fr->h_stage, (size_t)J.n_local_px * 3);

    const Counters* per_gen = fr->h_counters;
    Counters ctr{};
    if (J.counters_on) {
        static_assert(sizeof(Counters) % sizeof(unsigned long long) == 0, "Counters is an array of u64");
        for (int gen = 0; gen < G; gen++)
            for (size_t k = 0; k < sizeof(Counters) / sizeof(unsigned long long); k++)
                ((unsigned long long*)&ctr)[k] += ((const unsigned long long*)&per_gen[gen])[k];
        if (gen_timing) print_counter_timing(J, per_gen, ctr);
    }
    if (stats) {
        stats->generations = (uint32_t)G;
        stats->queue_overflows = J.overflows;
        if (J.counters_on) {
            stats->rays = ctr.rays; stats->primary_rays = ctr.primary; stats->shadow_rays = ctr.shadow;
            stats->secondary_rays = ctr.secondary;
        } else {
            // The ray counts need no device counters: one primary ray per pixel of the selected
            // tiles, and every later ray passed through a queue whose fill levels came back with
            // the control blocks.
            uint64_t primary = 0, radiance = 0, shadow = 0;
            for (int k = 0; k < J.g.n_sel; k++) {
                const int tile = J.first + k * J.stride;
                const int x0 = (tile % J.g.tiles_x) * J.tw, y0 = (tile / J.g.tiles_x) * J.th;
                primary += (uint64_t)std::max(0, std::min(J.tw, J.cam.width - x0)) * (uint64_t)std::max(0, std::min(J.th, J.cam.height - y0));
            }
            for (int k = 1; k <= G; k++)
                for (const SubCount& sub : ctrl[k].sub) { radiance += sub.counts & 0xFFFFFFFFull; shadow += (sub.counts >> 32) + sub.early; }
            stats->primary_rays = primary; stats->secondary_rays = radiance; stats->shadow_rays = shadow;
            stats->rays = primary + radiance + shadow;
        }
        stats->node_visits = ctr.node_visits; stats->tri_tests = ctr.tri_tests;
        stats->world_box_tests = ctr.world_box_tests; stats->object_space_traversals = ctr.object_traversals;
        stats->tlas_node_visits = ctr.tlas_visits;
        stats->wave_inner_steps = ctr.inner_steps; stats->wave_inner_lanes = ctr.inner_lanes;
        stats->wave_leaf_steps = ctr.leaf_steps; stats->wave_leaf_lanes = ctr.leaf_lanes;
        stats->kernel_ms = J.kernel_ms;
        stats->trace_ms = J.trace_ms; stats->trace_launches = J.trace_launches; stats->instance_path = J.instance_path;
        stats->wide_blas = J.b8 ? 1u : 0u; stats->wide_blas_redone = J.no_b8 ? 1u : 0u;
        for (int k = 1; k <= G; k++)
            for (const SubCount& sub : ctrl[k].sub) stats->shadow_rays_settled_early += sub.early;
        for (int k = 0; k <= G; k++) stats->record_fetches += ctrl[k].records;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - J.wall0).count();
    }
    return RTX_OK;
}

}  // namespace

extern "C" {

int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out) {
    if (!desc || !out) return fail(RTX_EINVAL, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RTX_ENODEVICE, "no HIP device available");
    if (device < 0 || device >= n) return fail(RTX_ENODEVICE, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    rtx_scene* sc = new rtx_scene();
    sc->device = device;
    sc->debug = debug_from_environment();
    int rc = build_scene(*sc, *desc);
    if (rc == RTX_OK) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) sc->cus = prop.multiProcessorCount;
        // device address of this GPU's copy of the no-allocation scratch array (first scene on the device)
        unsigned char* at = nullptr;
        if (device < RTX_MAX_DEVICES) {
            const hipError_t e = scratch_address(device, &at);
            if (e != hipSuccess) rc = fail(RTX_EHIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
        }
    }
    // the frame context behind rtx_render and rtx_render_noalloc
    if (rc == RTX_OK) rc = frame_create(sc, &sc->own);
    if (rc != RTX_OK) { if (sc->own) frame_destroy(sc->own); delete sc; return rc; }
    *out = sc;
    return RTX_OK;
}

int rtx_scene_update_lights(rtx_scene* sc, const rtx_light* lights, int32_t n_lights) {
    if (!sc || n_lights < 0 || (n_lights > 0 && !lights)) return fail(RTX_EINVAL, "bad argument");
    HIP_TRY(hipSetDevice(sc->device));
    std::vector<Light> l((size_t)n_lights);
    for (int i = 0; i < n_lights; i++) { memcpy(l[(size_t)i].p, lights[i].position, 24); memcpy(l[(size_t)i].i, lights[i].intensity, 24); }
    // A new version: frames in flight keep the one they were submitted with.  A version that no
    // frame and not the scene holds any more is written over instead of allocating another.
    std::lock_guard<std::mutex> lock(sc->lights_mutex);
    std::shared_ptr<LightSet> ls;
    for (auto& p : sc->light_pool)
        if (p.use_count() == 1) { ls = p; break; }
    if (!ls) { ls = std::make_shared<LightSet>(); sc->light_pool.push_back(ls); }
    HIP_TRY(ls->buf.upload(l));          // (blocking copy: complete before any frame can be submitted with it)
    ls->n = n_lights;
    sc->lights = ls;
    sc->dev.lights = (const Light*)ls->buf.p;
    sc->dev.n_lights = n_lights;
    return RTX_OK;
}

int rtx_scene_set_counters(rtx_scene* sc, int32_t enable) {
    if (!sc) return fail(RTX_EINVAL, "null scene");
    sc->counters_on = enable != 0;
    return RTX_OK;
}

int rtx_scene_set_consistent(rtx_scene* sc, int32_t enable) {
    if (!sc) return fail(RTX_EINVAL, "null scene");
    sc->consistent = enable != 0;
    return RTX_OK;
}

int rtx_scene_set_generation_timing(rtx_scene* sc, int32_t enable) {
    if (!sc) return fail(RTX_EINVAL, "null scene");
    sc->generation_timing = enable != 0;
    return RTX_OK;
}

int rtx_scene_set_queue_limit(rtx_scene* sc, int64_t bytes) {
    if (!sc || bytes < (int64_t)(1 << 20)) return fail(RTX_EINVAL, "bad queue limit");
    sc->queue_limit = bytes;
    return RTX_OK;
}

int rtx_scene_set_wide_blas(rtx_scene* sc, int32_t enable) {
    if (!sc) return fail(RTX_EINVAL, "null scene");
    if (enable && !sc->b8_built)
        return fail(RTX_EINVAL, "rtx_scene_set_wide_blas: the scene was created without RTX_SCENE_WIDE_BLAS in rtx_scene_desc::build_flags");
    sc->wide_blas = enable != 0;
    return RTX_OK;
}

int64_t rtx_scene_record_bytes(const rtx_scene* sc) {
    return sc ? (int64_t)sc->blas.bytes : -1;
}

int rtx_scene_set_debug(rtx_scene* sc, uint32_t flags) {
    if (!sc) return fail(RTX_EINVAL, "null scene");
    if (flags & ~(uint32_t)RTX_DEBUG_ALL) return fail(RTX_EINVAL, "unknown debug flag");
    sc->debug = flags;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
    // frames the caller has not destroyed go with the scene (their handles are dead from here on)
    while (true) {
        rtx_frame* fr = nullptr;
        {
            std::lock_guard<std::mutex> lock(sc->frames_mutex);
            if (!sc->frames.empty()) fr = sc->frames.back();
        }
        if (!fr) break;
        frame_destroy(fr);
    }
    delete sc;
}

int rtx_frame_create(rtx_scene* sc, rtx_frame** out) {
    if (!sc || !out) return fail(RTX_EINVAL, "null argument");
    return frame_create(sc, out);
}

void rtx_frame_destroy(rtx_frame* fr) {
    if (fr) frame_destroy(fr);
}

int rtx_render_async(rtx_frame* fr, const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride,
                     uint8_t* rgb_out, int32_t rgb_is_device, rtx_hit_id* hit_ids) {
    return frame_submit(fr, cam, tw, th, first, stride, rgb_out, rgb_is_device, hit_ids, false);
}

int rtx_frame_wait(rtx_frame* fr, rtx_stats* stats) {
    return frame_wait(fr, stats);
}

int rtx_render(rtx_scene* sc, const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride,
               uint8_t* rgb_out, int32_t rgb_is_device, rtx_hit_id* hit_ids, rtx_stats* stats) {
    if (!sc) return fail(RTX_EINVAL, "null argument");
    if (stats) memset(stats, 0, sizeof *stats);
    const int rc = frame_submit(sc->own, cam, tw, th, first, stride, rgb_out, rgb_is_device, hit_ids, false);
    if (rc) return rc;
    return frame_wait(sc->own, stats);
}

int rtx_render_noalloc(rtx_scene* sc, const rtx_camera* cam, int32_t tw, int32_t th, int32_t first, int32_t stride,
                       uint8_t* rgb_out, int32_t rgb_is_device, rtx_stats* stats) {
    if (!sc) return fail(RTX_EINVAL, "null argument");
    if (stats) memset(stats, 0, sizeof *stats);
    const int rc = frame_submit(sc->own, cam, tw, th, first, stride, rgb_out, rgb_is_device, nullptr, true);
    if (rc) return rc;
    return frame_wait(sc->own, stats);
}

int64_t rtx_noalloc_scratch_bytes(void) { return (int64_t)RTX_NOALLOC_SCRATCH_BYTES; }

}  // extern "C"
