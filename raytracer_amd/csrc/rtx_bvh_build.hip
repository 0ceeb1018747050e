// rtx_build_bvh -- the reference's per-mesh BVH (bvh.cpp:80-102, 187-258), built on the GPU.
//
// The reference builds top-down and recursively: node box over the faces of the range, leaf
// when <= 4 faces / depth 64 / centroid extent below 1e-5f, otherwise the range is split at
// its median position after std::nth_element on the centroid coordinate of the widest
// centroid axis; nodes are numbered in depth-first pre-order.  Which faces land on which side
// of a median is a property of the ranks alone, so the same tree comes out of a
// level-synchronous build: per level, one segmented reduction (node boxes and centroid
// boxes of all open ranges), one decision per range, and one segmented sort of the ranges
// that split (a sorted range has the median property).  The pre-order numbers follow from
// subtree sizes afterwards.  Boxes, topology, split axes and the face set of every leaf equal
// the reference's; the order of faces *inside* a leaf is the sorted one where the reference
// has whatever nth_element left (and faces whose centroid coordinates tie exactly at a median
// may fall on the other side).
//
// Arithmetic is the reference's (centroid = (a + b + c) * (1.0 / 3.0) per coordinate,
// min/max boxes from 1e30f, extent compare against 1e-5f), compiled without contraction.
#include <cstring>              // (before rocprim.hpp: one of its headers calls memset unqualified)

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "rtx_amd.h"

namespace rtxbuild {

void set_error(const std::string& s);      // rtx_core.hip: the library's last-error string

namespace {

struct Red { double v[12]; };             // face box min, max; centroid min, max
struct RedOp {
    __host__ __device__ Red operator()(const Red& a, const Red& b) const {
        Red r;
        for (int k = 0; k < 3; k++) {
            r.v[k] = a.v[k] < b.v[k] ? a.v[k] : b.v[k];
            r.v[3 + k] = a.v[3 + k] > b.v[3 + k] ? a.v[3 + k] : b.v[3 + k];
            r.v[6 + k] = a.v[6 + k] < b.v[6 + k] ? a.v[6 + k] : b.v[6 + k];
            r.v[9 + k] = a.v[9 + k] > b.v[9 + k] ? a.v[9 + k] : b.v[9 + k];
        }
        return r;
    }
};
__host__ __device__ inline Red red_identity() {
    Red r;
    for (int k = 0; k < 3; k++) { r.v[k] = (double)1e30f; r.v[3 + k] = (double)-1e30f; r.v[6 + k] = (double)1e30f; r.v[9 + k] = (double)-1e30f; }
    return r;
}

// per face: its box and its centroid (as a degenerate box), in face order
__global__ void k_faces(const double* __restrict__ verts, const int32_t* __restrict__ faces, int n, Red* __restrict__ fred,
                        double* __restrict__ cen, int* __restrict__ idx, int* __restrict__ pos_seg) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* a = verts + 3 * (size_t)faces[3 * (size_t)i];
    const double* b = verts + 3 * (size_t)faces[3 * (size_t)i + 1];
    const double* c = verts + 3 * (size_t)faces[3 * (size_t)i + 2];
    const double inv3 = 1.0 / 3.0;
    Red r;
    for (int k = 0; k < 3; k++) {
        // Box::expand from (1e30f, -1e30f), vertex by vertex (bvh.h:13-21)
        double mn = (double)1e30f, mx = (double)-1e30f;
        mn = a[k] < mn ? a[k] : mn; mx = a[k] > mx ? a[k] : mx;
        mn = b[k] < mn ? b[k] : mn; mx = b[k] > mx ? b[k] : mx;
        mn = c[k] < mn ? c[k] : mn; mx = c[k] > mx ? c[k] : mx;
        const double ce = (a[k] + b[k] + c[k]) * inv3;
        r.v[k] = mn; r.v[3 + k] = mx;
        r.v[6 + k] = ce < (double)1e30f ? ce : (double)1e30f;
        r.v[9 + k] = ce > (double)-1e30f ? ce : (double)-1e30f;
        cen[(size_t)k * n + i] = ce;
    }
    fred[i] = r;
    idx[i] = i;
    pos_seg[i] = 0;
}

struct Gather {
    const Red* fred; const int* idx;
    __host__ __device__ Red operator()(int pos) const { return fred[idx[pos]]; }
};

// Node arrays in creation (level) order.
struct Nodes {
    double* box;          // 6 per node
    int *start, *end, *left, *right, *axis, *size, *pre;
};

// One open range per thread: store the node, decide leaf / split, open the children.
__global__ void k_decide(int n_seg, int depth, const int* __restrict__ seg_node, const Red* __restrict__ red, Nodes N,
                         int next_base, int* __restrict__ counters /* [0] splits */, int* __restrict__ split_begin,
                         int* __restrict__ split_end, int* __restrict__ seg_axis, int* __restrict__ seg_child,
                         int* __restrict__ next_seg_node) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_seg) return;
    const int node = seg_node[s];
    const Red r = red[s];
    for (int k = 0; k < 6; k++) N.box[6 * (size_t)node + k] = r.v[k];
    const int start = N.start[node], end = N.end[node], count = end - start;
    N.left[node] = -1; N.right[node] = -1; N.axis[node] = 0;
    seg_child[s] = -1;
    seg_axis[s] = 0;
    if (count <= 4 || depth >= 64) return;                                    // bvh.cpp:205
    const double ext[3] = {r.v[9] - r.v[6], r.v[10] - r.v[7], r.v[11] - r.v[8]};
    int axis = 0;
    double best = ext[0];
    if (ext[1] > best) { axis = 1; best = ext[1]; }
    if (ext[2] > best) { axis = 2; best = ext[2]; }
    if (best < (double)1e-5f) return;                                         // bvh.cpp:229
    const int k = atomicAdd(&counters[0], 1);
    const int mid = start + count / 2;
    const int l = next_base + 2 * k, rr = l + 1;
    N.left[node] = l; N.right[node] = rr; N.axis[node] = axis;
    N.start[l] = start; N.end[l] = mid;
    N.start[rr] = mid; N.end[rr] = end;
    split_begin[k] = start; split_end[k] = end;
    seg_axis[s] = axis;
    seg_child[s] = 2 * k;                    // index of the left child among the next level's ranges
    next_seg_node[2 * k] = l; next_seg_node[2 * k + 1] = rr;
}

// sort key of every position inside a splitting range
__global__ void k_keys(int n, const int* __restrict__ pos_seg, const int* __restrict__ seg_child, const int* __restrict__ seg_axis,
                       const int* __restrict__ idx, const double* __restrict__ cen, double* __restrict__ keys) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = pos_seg[p];
    if (s < 0 || seg_child[s] < 0) return;
    keys[p] = cen[(size_t)seg_axis[s] * n + idx[p]];
}

// take the sorted permutation inside splitting ranges and hand every position to its child range
__global__ void k_advance(int n, int* __restrict__ pos_seg, const int* __restrict__ seg_child, const int* __restrict__ seg_node,
                          Nodes N, const int* __restrict__ sorted, int* __restrict__ idx) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int s = pos_seg[p];
    if (s < 0) return;
    const int c = seg_child[s];
    if (c < 0) { pos_seg[p] = -1; return; }
    idx[p] = sorted[p];
    const int node = seg_node[s];
    const int mid = N.end[N.left[node]];
    pos_seg[p] = p < mid ? c : c + 1;
}

__global__ void k_sizes(int first, int count, Nodes N) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count) return;
    N.size[i] = N.left[i] < 0 ? 1 : 1 + N.size[N.left[i]] + N.size[N.right[i]];
}
__global__ void k_preorder(int first, int count, Nodes N) {
    const int i = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= first + count || N.left[i] < 0) return;
    N.pre[N.left[i]] = N.pre[i] + 1;
    N.pre[N.right[i]] = N.pre[i] + 1 + N.size[N.left[i]];
}
__global__ void k_emit(int n_nodes, Nodes N, rtx_bvh_node* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    rtx_bvh_node o;
    for (int k = 0; k < 3; k++) { o.bounds_min[k] = N.box[6 * (size_t)i + k]; o.bounds_max[k] = N.box[6 * (size_t)i + 3 + k]; }
    const bool leaf = N.left[i] < 0;
    o.left = leaf ? -1 : N.pre[N.left[i]];
    o.right = leaf ? -1 : N.pre[N.right[i]];
    o.first_face = leaf ? N.start[i] : -1;
    o.face_count = leaf ? N.end[i] - N.start[i] : 0;
    o.split_axis = N.axis[i];
    o.pad = 0;
    out[N.pre[i]] = o;
}

struct Buf {
    void* p = nullptr;
    hipError_t get(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    ~Buf() { if (p) (void)hipFree(p); }
    template <typename T> T* as() const { return (T*)p; }
};

#define BUILD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error(std::string("rtx_build_bvh: ") + hipGetErrorString(e_)); return RTX_EHIP; } } while (0)

}  // namespace

int build(int device, const double* vertices, int64_t n_vertices, const int32_t* faces, int32_t n_faces,
          rtx_bvh_node* nodes_out, int32_t* n_nodes_out, int32_t* face_indices_out, double* build_ms) {
    if (!n_nodes_out || n_faces < 0 || (n_faces > 0 && (!vertices || !faces || !nodes_out || !face_indices_out))) {
        set_error("rtx_build_bvh: bad argument");
        return RTX_EINVAL;
    }
    *n_nodes_out = 0;
    if (build_ms) *build_ms = 0.0;
    if (n_faces == 0) return RTX_OK;
    for (int64_t i = 0; i < 3 * (int64_t)n_faces; i++)
        if (faces[i] < 0 || faces[i] >= n_vertices) { set_error("rtx_build_bvh: face references a vertex outside the pool"); return RTX_EINVAL; }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) { set_error("no HIP device available"); return RTX_ENODEVICE; }
    if (device < 0 || device >= n_dev) { set_error("rtx_build_bvh: no such device"); return RTX_EINVAL; }
    BUILD_TRY(hipSetDevice(device));
    const auto t0 = std::chrono::high_resolution_clock::now();
    const int n = n_faces;
    const size_t max_nodes = 2 * (size_t)n;          // a split always leaves >= 2 faces per child
    const int T = 256;
    auto blocks = [&](int m) { return (unsigned)((m + T - 1) / T); };

    Buf d_verts, d_faces, d_fred, d_cen, d_idx, d_sorted, d_keys, d_keys_out, d_pos_seg;
    Buf d_box, d_start, d_end, d_left, d_right, d_axis, d_size, d_pre, d_out;
    Buf d_seg_node[2], d_red, d_counters, d_split_begin, d_split_end, d_seg_axis, d_seg_child, d_tmp;
    BUILD_TRY(d_verts.get((size_t)n_vertices * 24));
    BUILD_TRY(d_faces.get((size_t)n * 12));
    BUILD_TRY(d_fred.get((size_t)n * sizeof(Red)));
    BUILD_TRY(d_cen.get((size_t)n * 24));
    BUILD_TRY(d_idx.get((size_t)n * 4));
    BUILD_TRY(d_sorted.get((size_t)n * 4));
    BUILD_TRY(d_keys.get((size_t)n * 8));
    BUILD_TRY(d_keys_out.get((size_t)n * 8));
    BUILD_TRY(d_pos_seg.get((size_t)n * 4));
    BUILD_TRY(d_box.get(max_nodes * 48));
    for (Buf* b : {&d_start, &d_end, &d_left, &d_right, &d_axis, &d_size, &d_pre}) BUILD_TRY(b->get(max_nodes * 4));
    BUILD_TRY(d_out.get(max_nodes * sizeof(rtx_bvh_node)));
    // a level never has more open ranges than faces / 2 + 1
    const size_t max_seg = (size_t)n / 2 + 2;
    for (Buf& b : d_seg_node) BUILD_TRY(b.get(max_seg * 4));
    BUILD_TRY(d_red.get(max_seg * sizeof(Red)));
    BUILD_TRY(d_counters.get(16));
    for (Buf* b : {&d_split_begin, &d_split_end, &d_seg_axis, &d_seg_child}) BUILD_TRY(b->get(max_seg * 4));

    BUILD_TRY(hipMemcpy(d_verts.p, vertices, (size_t)n_vertices * 24, hipMemcpyHostToDevice));
    BUILD_TRY(hipMemcpy(d_faces.p, faces, (size_t)n * 12, hipMemcpyHostToDevice));
    Nodes N{d_box.as<double>(), d_start.as<int>(), d_end.as<int>(), d_left.as<int>(), d_right.as<int>(),
            d_axis.as<int>(), d_size.as<int>(), d_pre.as<int>()};
    hipLaunchKernelGGL(k_faces, dim3(blocks(n)), dim3(T), 0, 0, d_verts.as<double>(), d_faces.as<int32_t>(), n,
                       d_fred.as<Red>(), d_cen.as<double>(), d_idx.as<int>(), d_pos_seg.as<int>());
    {
        const int root_range[2] = {0, n}, zero = 0;
        BUILD_TRY(hipMemcpy(d_start.p, &root_range[0], 4, hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_end.p, &root_range[1], 4, hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_seg_node[0].p, &zero, 4, hipMemcpyHostToDevice));
    }

    // temporary storage for the two segmented primitives, sized for the worst level
    size_t tmp_reduce = 0, tmp_sort = 0;
    auto in_it = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), Gather{d_fred.as<Red>(), d_idx.as<int>()});
    BUILD_TRY(rocprim::segmented_reduce(nullptr, tmp_reduce, in_it, d_red.as<Red>(), (unsigned)max_seg, d_start.as<int>(),
                                        d_end.as<int>(), RedOp(), red_identity(), 0));
    BUILD_TRY(rocprim::segmented_radix_sort_pairs(nullptr, tmp_sort, d_keys.as<double>(), d_keys_out.as<double>(), d_idx.as<int>(),
                                                  d_sorted.as<int>(), (unsigned)n, (unsigned)max_seg, d_split_begin.as<int>(),
                                                  d_split_end.as<int>(), 0, 64, 0));
    // The segmented primitives give one workgroup to a segment; near the root the segments are
    // few and huge, so levels whose ranges exceed kWide elements use the device-wide primitives,
    // one call per range.
    constexpr int kWide = 1 << 15;
    size_t tmp_wide_reduce = 0, tmp_wide_sort = 0;
    BUILD_TRY(rocprim::reduce(nullptr, tmp_wide_reduce, in_it, d_red.as<Red>(), red_identity(), (size_t)n, RedOp(), 0));
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tmp_wide_sort, d_keys.as<double>(), d_keys_out.as<double>(), d_idx.as<int>(),
                                        d_sorted.as<int>(), (size_t)n, 0, 64, 0));
    BUILD_TRY(d_tmp.get(std::max(std::max(tmp_reduce, tmp_sort), std::max(tmp_wide_reduce, tmp_wide_sort))));
    // ranges of the open nodes of a level, gathered for the reduction
    Buf d_rb, d_re;
    BUILD_TRY(d_rb.get(max_seg * 4));
    BUILD_TRY(d_re.get(max_seg * 4));

    std::vector<int> level_first, level_count;
    int n_seg = 1, n_nodes = 1, cur = 0;
    for (int depth = 0; n_seg > 0; depth++) {
        level_first.push_back(n_nodes - n_seg);
        level_count.push_back(n_seg);
        // ranges of this level's nodes (they were allocated contiguously: [n_nodes - n_seg, n_nodes))
        BUILD_TRY(hipMemcpy(d_rb.p, d_start.as<int>() + (n_nodes - n_seg), (size_t)n_seg * 4, hipMemcpyDeviceToDevice));
        BUILD_TRY(hipMemcpy(d_re.p, d_end.as<int>() + (n_nodes - n_seg), (size_t)n_seg * 4, hipMemcpyDeviceToDevice));
        const bool wide = n_seg <= 64 && n / n_seg > kWide;
        std::vector<int> h_b, h_e;
        if (wide) {
            h_b.resize((size_t)n_seg); h_e.resize((size_t)n_seg);
            BUILD_TRY(hipMemcpy(h_b.data(), d_rb.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost));
            BUILD_TRY(hipMemcpy(h_e.data(), d_re.p, (size_t)n_seg * 4, hipMemcpyDeviceToHost));
            for (int s = 0; s < n_seg; s++) {
                size_t tb = tmp_wide_reduce;
                BUILD_TRY(rocprim::reduce(d_tmp.p, tb, in_it + h_b[(size_t)s], d_red.as<Red>() + s, red_identity(),
                                          (size_t)(h_e[(size_t)s] - h_b[(size_t)s]), RedOp(), 0));
            }
        } else {
            size_t tb = tmp_reduce;
            BUILD_TRY(rocprim::segmented_reduce(d_tmp.p, tb, in_it, d_red.as<Red>(), (unsigned)n_seg, d_rb.as<int>(), d_re.as<int>(),
                                                RedOp(), red_identity(), 0));
        }
        BUILD_TRY(hipMemset(d_counters.p, 0, 16));
        hipLaunchKernelGGL(k_decide, dim3(blocks(n_seg)), dim3(T), 0, 0, n_seg, depth, d_seg_node[cur].as<int>(), d_red.as<Red>(), N,
                           n_nodes, d_counters.as<int>(), d_split_begin.as<int>(), d_split_end.as<int>(), d_seg_axis.as<int>(),
                           d_seg_child.as<int>(), d_seg_node[cur ^ 1].as<int>());
        int n_split = 0;
        BUILD_TRY(hipMemcpy(&n_split, d_counters.p, 4, hipMemcpyDeviceToHost));
        if (n_split > 0) {
            hipLaunchKernelGGL(k_keys, dim3(blocks(n)), dim3(T), 0, 0, n, d_pos_seg.as<int>(), d_seg_child.as<int>(),
                               d_seg_axis.as<int>(), d_idx.as<int>(), d_cen.as<double>(), d_keys.as<double>());
            if (wide) {
                h_b.resize((size_t)n_split); h_e.resize((size_t)n_split);
                BUILD_TRY(hipMemcpy(h_b.data(), d_split_begin.p, (size_t)n_split * 4, hipMemcpyDeviceToHost));
                BUILD_TRY(hipMemcpy(h_e.data(), d_split_end.p, (size_t)n_split * 4, hipMemcpyDeviceToHost));
                for (int k = 0; k < n_split; k++) {
                    const int b = h_b[(size_t)k];
                    size_t ts = tmp_wide_sort;
                    BUILD_TRY(rocprim::radix_sort_pairs(d_tmp.p, ts, d_keys.as<double>() + b, d_keys_out.as<double>() + b,
                                                        d_idx.as<int>() + b, d_sorted.as<int>() + b, (size_t)(h_e[(size_t)k] - b),
                                                        0, 64, 0));
                }
            } else {
                size_t ts = tmp_sort;
                BUILD_TRY(rocprim::segmented_radix_sort_pairs(d_tmp.p, ts, d_keys.as<double>(), d_keys_out.as<double>(),
                                                              d_idx.as<int>(), d_sorted.as<int>(), (unsigned)n, (unsigned)n_split,
                                                              d_split_begin.as<int>(), d_split_end.as<int>(), 0, 64, 0));
            }
        }
        hipLaunchKernelGGL(k_advance, dim3(blocks(n)), dim3(T), 0, 0, n, d_pos_seg.as<int>(), d_seg_child.as<int>(),
                           d_seg_node[cur].as<int>(), N, d_sorted.as<int>(), d_idx.as<int>());
        n_nodes += 2 * n_split;
        n_seg = 2 * n_split;
        cur ^= 1;
        if ((size_t)n_nodes > max_nodes) { set_error("rtx_build_bvh: node count exceeds its bound"); return RTX_ELIMIT; }
    }
    for (int l = (int)level_first.size() - 1; l >= 0; l--)
        hipLaunchKernelGGL(k_sizes, dim3(blocks(level_count[(size_t)l])), dim3(T), 0, 0, level_first[(size_t)l], level_count[(size_t)l], N);
    BUILD_TRY(hipMemset(d_pre.p, 0, 4));
    for (size_t l = 0; l < level_first.size(); l++)
        hipLaunchKernelGGL(k_preorder, dim3(blocks(level_count[l])), dim3(T), 0, 0, level_first[l], level_count[l], N);
    hipLaunchKernelGGL(k_emit, dim3(blocks(n_nodes)), dim3(T), 0, 0, n_nodes, N, d_out.as<rtx_bvh_node>());
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpy(nodes_out, d_out.p, (size_t)n_nodes * sizeof(rtx_bvh_node), hipMemcpyDeviceToHost));
    BUILD_TRY(hipMemcpy(face_indices_out, d_idx.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    *n_nodes_out = n_nodes;
    if (build_ms) *build_ms = std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t0).count();
    return RTX_OK;
}

}  // namespace rtxbuild
