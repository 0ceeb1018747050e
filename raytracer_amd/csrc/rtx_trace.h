// The closest-hit kernel (k_trace): every lane owns one ray at a time and takes it through the
// reference's object sequence (utils.cpp:671-865) as a state machine whose every transition that
// needs memory is one *step* of a wave-wide loop.
//
// A step: each lane names the one 256-byte record its state needs -- a four-wide BVH node or two
// triangles of a leaf (traversing), the first half of an object record (evaluating a mesh or an
// instance candidate: inverse matrix, world box, root box, root), a triangle record and the second
// half of an object record (finishing a hit: rayHitsMesh's epilogue, utils.cpp:533-668) -- the wave
// fetches all of them with one cooperative LDS-DMA gather (coop_fetch), and every lane then does
// its state's arithmetic.  Lanes that walk the instance bitmap issue their one atomic exchange
// before the gather and find the word after it.  So whatever a lane is doing, its memory latency
// is the step's, shared with the other 63 lanes; nothing waits for a quorum of lanes in the same
// situation.  Only handing out fresh rays and retiring finished ones (shadow contribution, hit
// record status) happens in a divergent "service pass", when enough lanes are idle.
#pragma once
#include "rtx_kernels.h"

namespace rtxdev {

// Lane states.  Everything from ST_ADV up is a ray at work.  A state either needs a record (the
// next step fetches it) or has it in the lane's LDS row and waits for its kind of work to be run.
enum { ST_EMPTY = 0,   // no ray
       ST_DONE = 1,    // ray finished, to be retired by the next service pass
       ST_ADV = 2,     // an object has been dealt with: pick the next one (registers only, at the top of the loop)
       ST_EVAL = 3,    // object L.obj's record is at hand (block's LDS copy, or the lane's row): to be evaluated at the top of the loop
       ST_TRAV = 4,    // inside a traversal job, needs record cur_ref (node if cur_cnt == 0, else leaf triangles)
       ST_TRAV_R = 5,  //   ... and has it
       ST_OBJ = 6,     // object L.obj is next: needs the first half of its object record
       ST_SUM = 7,     // instance replay: taking summary word L.sword of the candidate bitmap
       ST_WORD = 8,    //                  taking bitmap word L.word
       ST_CLEAN = 9,   // decided shadow ray: clearing the candidates it will not replay
       ST_FIN = 10,    // job ended with a hit: needs the winning triangle's record
       ST_FIN_R = 11,  //   ... and has it
       ST_FIN2 = 12,   // ... of a transformed object: needs the second half of the object record
       ST_FIN2_R = 13 };

#ifndef RTX_SERVICE_IDLE
#define RTX_SERVICE_IDLE 16          // idle lanes (empty or finished) that trigger a service pass
#endif
// A kind of work (box tests of a node, triangle tests of a leaf, evaluating an object, finishing a
// hit) costs the wave the same instructions whether one lane takes part or sixty, and the kernel is
// bound by instruction issue: a kind is run when this many lanes are waiting for it (fewer when
// the wave has few rays at work: a third of them; always once the generation has no fresh rays).
#ifndef RTX_Q_NODE
#define RTX_Q_NODE 16
#define RTX_Q_LEAF 12
#define RTX_Q_EVAL 12
#define RTX_Q_FIN 8
#endif
#ifndef RTX_EVAL_REPS
#define RTX_EVAL_REPS 3              // objects a lane may evaluate from the block's LDS copy before the step goes on
#endif

// The record a lane's state asked for, where the step left it: the lane's own LDS row (pieces
// XOR-swizzled, see coop_fetch) or, for the scene's first objects, the block's LDS copy (linear).
// Pieces are read where they are used, so that a 256-byte record never sits in registers whole.
struct RecView {
    const lds_v2d* base; unsigned sw;
    RTX_DEV v2d piece(unsigned p) const { return base[p ^ sw]; }
    // doubles [2 * first, 2 * first + 2 * N) of the record
    template <int N> RTX_DEV void read(unsigned first, double* out) const {
#pragma unroll
        for (int p = 0; p < N; p++) { const v2d v = piece(first + (unsigned)p); out[2 * p] = v.x; out[2 * p + 1] = v.y; }
    }
    RTX_DEV double dbl(unsigned i) const { const v2d v = piece(i >> 1); return (i & 1u) ? v.y : v.x; }
};

struct Lane {
    V3 o, d;                // world-space ray
    double wix, wiy, wiz;   // 1 / d (bvh.cpp:43), once per ray
    double t_min, dist;     // current bound; Intersection::distance of the record
    unsigned slot;          // queue slot (work item for generation 0) = hit-record slot
    int is_shadow, hit;
    // object iterator: it <= n_meshes: meshes; it > n_meshes: instances, taken in ascending index
    // from the lane's two-level instance bitmap: bitmap word `word` is being consumed (`bits` = its
    // remaining candidates); which words hold any is recorded in summary words, of which `sword`
    // is being consumed (`sbits`) and `whi` is the last that has any.  While the TLAS walk runs,
    // word/whi are the lowest/highest bitmap word it has marked.
    int it, word, whi, sword;
    unsigned bits, sbits;
    // current object / traversal job
    int obj;                // -1: the TLAS walk
    int oflags, omat, omesh;
    double orig_t_min;
    V3 ro, rd;              // the job's ray; between ST_FIN and ST_FIN2: hit point and geometric normal
    double ix, iy, iz;      // its 1 / d;     between ST_FIN and ST_FIN2: beta, gamma
    int par;                // "parallel" axes (bvh.cpp:36): bits 0-2 the job's ray, bits 4-6 the world ray
    int cur_ref, cur_cnt, sp, btri;
    int state;
    int age;                // traversal steps spent on the current job (with counters on: on the current ray)
    v4i top;                // top stack entry (valid while sp > 0)
};

// spheres (utils.cpp:746-748): rare, kept out of line so that its registers are not the loop's
struct SphereOut { double t_min, dist; int hit; };
__device__ __attribute__((noinline)) SphereOut sphere_pass(const Scene& S, const double ox, const double oy, const double oz,
                                                           const double dx, const double dy, const double dz, double t_min,
                                                           double dist, int hit_in, Hit* rec_at, const int on, const int ids) {
    RecOut out;
    out.rec = (RTX_GLOBAL Hit*)rec_at; out.on = on != 0; out.ids = ids != 0;
    Rec r;
    r.kind = -1; r.dist = 0.0; r.hit = 0; r.mat = -1; r.container = -1; r.src = -1;
    r.point = mk(0, 0, 0); r.gn = r.point; r.sn = r.point;
    bool hit = hit_in != 0;
    for (int i = 0; i < S.n_spheres; i++) hit = sphere_hit(S, mk(ox, oy, oz), mk(dx, dy, dz), i, t_min, r) || hit;
    if (r.kind != -1) {                              // fields a sphere hit writes, utils.cpp:269-291
        out.geom(r.dist, r.point, r.gn, r.sn);
        out.owner(r.mat, r.container);
        out.kind(r.kind);
        out.src(r.src);
        dist = r.dist;
    }
    SphereOut o;
    o.t_min = t_min; o.dist = dist; o.hit = hit ? 1 : 0;
    return o;
}

// A fresh ray: empty record, then the classes that come before the meshes
// (planes, utils.cpp:683-685; free triangles, utils.cpp:692-729).
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
    const Slab ws = make_slab(L.o, L.d);
    L.wix = ws.ix; L.wiy = ws.iy; L.wiz = ws.iz; L.par = ws.par << 4;
    L.it = 0; L.word = 0; L.whi = -1; L.bits = 0u; L.sword = 0; L.sbits = 0u; L.age = 0;
    L.state = ST_ADV;
}

RTX_DEV Slab world_slab(const Lane& L) {
    Slab s;
    s.ox = L.o.x; s.oy = L.o.y; s.oz = L.o.z; s.ix = L.wix; s.iy = L.wiy; s.iz = L.wiz; s.par = L.par >> 4;
    return s;
}

// The lane has dealt with an object (or has a fresh ray): what comes next in the reference's
// sequence -- meshes by index (utils.cpp:736-739), spheres (746-748), instances (756-790).
// Registers only, apart from stores nobody waits for.
//
// Instances: a walk of the TLAS -- a traversal job like a mesh's, with the world-space ray --
// marks the candidates in the lane's bitmap: instances whose world box the ray can meet within
// the current bound (a box missed now stays missed because the bound never grows).  They are then
// replayed in ascending index through the reference's own per-instance logic (world-box test
// against the then-current bound included), one object evaluation each.
//
// A shadow ray asks only whether something lies in front of its light (utils.cpp:1066-1084:
// hit && distance < |L - P|).  Once a hit has been accepted with the bound below that distance
// the answer is yes whatever follows: the bound never grows, every later write to the record's
// distance is below the bound of its moment, and the hit flag is never taken back.  Such a ray
// is retired here (not with counters on: there every test of the reference is carried out).
// Instance candidates it has collected but not replayed are dropped: the lane's candidate bitmap
// must be clear when the next ray starts.
template <bool COUNT, bool FAST>
RTX_DEV void advance(const Scene& S, const ObjCache& oc, Lane& L, const RecOut& out, unsigned* inst_bits) {
    const unsigned n_summary = rtx_summary_words(S.n_insts);
    if (!COUNT && L.is_shadow != 0 && L.hit != 0 && (FAST || L.t_min < oc.ctx[0])) {
        if (L.it > S.n_meshes) {
            // (the current word was cleared when it was taken; what is still set in memory: the
            // words flagged in the current summary word, and the summary words not yet taken)
            unsigned sb = L.sbits;
            while (sb != 0u) {
                const int w = L.sword * 32 + (__ffs((int)sb) - 1);
                sb &= sb - 1u;
                __hip_atomic_store(inst_bits + n_summary + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            L.bits = 0u; L.sbits = 0u;
            if (L.sword < L.whi) { L.state = ST_CLEAN; return; }
        }
        L.state = ST_DONE;
        return;
    }
    if (L.it < S.n_meshes) { L.obj = L.it; L.it++; L.state = L.obj < oc.n ? ST_EVAL : ST_OBJ; return; }
    if (L.it == S.n_meshes) {
        L.it++;
#ifndef EXP_NO_SPHERES
        if (S.n_spheres > 0) {
            const SphereOut so = sphere_pass(S, L.o.x, L.o.y, L.o.z, L.d.x, L.d.y, L.d.z, L.t_min, L.dist, L.hit,
                                             (Hit*)out.rec, out.on ? 1 : 0, out.ids ? 1 : 0);
            L.t_min = so.t_min; L.dist = so.dist; L.hit = so.hit;
        }
#endif
        L.word = INT_MAX; L.whi = -1; L.bits = 0u; L.sbits = 0u; L.sword = -1;
        if (S.n_insts > 0 && S.n_insts <= RTX_LINEAR_INSTS) {
            // a handful of instances: every one is a candidate (the reference's own loop,
            // utils.cpp:756-790); a TLAS walk would cost more than it can save
            L.word = 0; L.bits = (1u << S.n_insts) - 1u; L.sword = 0; L.whi = 0;
        } else if (S.n_insts > 0) {
            double tl_min[3] = {S.tl_min[0], S.tl_min[1], S.tl_min[2]}, tl_max[3] = {S.tl_max[0], S.tl_max[1], S.tl_max[2]};
            double enter;
            if (box_enter(tl_min, tl_max, world_slab(L), enter) && (L.t_min >= enter)) {
                L.obj = -1;                                  // marks the job as the TLAS walk
                L.is_shadow &= 1;
                if (!COUNT) L.age = 0;
                L.orig_t_min = L.t_min;
                L.ro = L.o; L.rd = L.d; L.ix = L.wix; L.iy = L.wiy; L.iz = L.wiz; L.par = (L.par & 0x70) | (L.par >> 4);
                L.cur_ref = S.tl_key; L.cur_cnt = 0; L.sp = 0; L.btri = -1; L.oflags = 0;
                L.state = ST_TRAV;
                return;
            }
        }
    }
    if (L.bits != 0u) {
        const int idx = L.word * 32 + (__ffs((int)L.bits) - 1);
        L.bits &= L.bits - 1u;
        L.obj = S.n_meshes + idx;
        L.state = L.obj < oc.n ? ST_EVAL : ST_OBJ;
        return;
    }
    if (L.sbits != 0u) {
        L.word = L.sword * 32 + (__ffs((int)L.sbits) - 1);
        L.sbits &= L.sbits - 1u;
        L.state = ST_WORD;
        return;
    }
    if (L.sword >= L.whi) { L.state = ST_DONE; return; }
    L.sword++;
    L.state = ST_SUM;
}

// A traversal job has ended (stack empty): what rayHitsMesh does next depends on what was found.
template <bool COUNT, bool FAST>
RTX_DEV void job_end(const ObjCache& oc, Lane& L) {
    if (L.obj == -1) {
        // end of the TLAS walk: L.word .. L.whi is the range of bitmap words it marked;
        // consumption starts in front of the first summary word that covers them
        L.sword = L.word == INT_MAX ? -1 : (L.word >> 5) - 1;
        L.whi = L.whi < 0 ? -1 : (L.whi >> 5);
        L.bits = 0u; L.sbits = 0u;
        L.obj = -2;
        L.state = ST_ADV;
    } else if (FAST && L.is_shadow) {
        // consistent mode: any accepted triangle occludes; the bound is the light's distance again
        if (L.btri >= 0) L.hit = 1;
        L.t_min = oc.ctx[0];
        L.state = ST_ADV;
    } else if (L.btri < 0) {
        if (FAST) L.t_min = L.orig_t_min;         // (a transformed job ran with the object-space bound)
        L.state = ST_ADV;
    } else if (L.is_shadow != 0 && !(L.oflags & OBJ_XF)) {
        // a shadow ray keeps no record: of an untransformed hit only distance and flag remain
        L.dist = L.t_min; L.hit = 1;
        L.state = ST_ADV;
    } else {
        L.state = ST_FIN;
    }
}

// Run the traversal jobs of the lanes in `m` wave-wide, one after the other (wide_job).
template <bool COUNT, bool FAST>
RTX_DEV void wide_jobs(const Scene& S, const ObjCache& oc, const Wave& wv, Lane& L, glob_v4i* stk, unsigned* inst_bits,
                       unsigned long long m) {
    WideIn in;
    in.rox = L.ro.x; in.roy = L.ro.y; in.roz = L.ro.z; in.rdx = L.rd.x; in.rdy = L.rd.y; in.rdz = L.rd.z;
    in.ix = L.ix; in.iy = L.iy; in.iz = L.iz; in.t_min = L.t_min; in.stop = L.orig_t_min;
    in.par = L.par & 7; in.shadow = L.is_shadow; in.sp = L.sp; in.cur_ref = L.cur_ref; in.cur_cnt = L.cur_cnt; in.top = L.top;
    while (m != 0ull) {
        const int src = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        const WideOut out = wide_job(S, wv, in, stk, inst_bits, src);
        if ((int)wv.lane == src) {
            if (out.t_min < INFINITY) { L.t_min = out.t_min; L.btri = out.btri; }
            L.word = min(L.word, out.wmin); L.whi = max(L.whi, out.wmax);
            L.sp = 0;
            job_end<COUNT, FAST>(oc, L);
        }
    }
}

// WIDE = 2: the variant for generations of at most RTX_WIDE_MAX_RAYS rays, all of whose last
// jobs are traversed wave-wide (wide_job).  It and a main variant are launched for every
// generation and exactly one of them does the work; the other returns at once.
// WIDE = 1: main variant in which drained waves finish their long-running jobs wave-wide too.
// WIDE = 0: main variant without any of it.  (Variants rather than branches: the mere presence
// of the call changes the register allocation of the code around it, so the host picks per scene.)
template <bool COUNT, int WIDE, bool FAST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RTX_TRACE_WAVES, RTX_TRACE_WAVES)))
k_trace(const Launch* __restrict__ Lp, GenCtrl* ctrl, const int gen, double* __restrict__ accum,
        rtx_hit_id* __restrict__ aov, Counters* __restrict__ counters) {
    const Scene& S = Lp->S;
    const Frame& F = Lp->F;
    const Queue& qin = Lp->q[(gen + 1) & 1];
    __shared__ __attribute__((aligned(16))) double lds_rows[4][64 * RTX_ROW_DOUBLES];      // 64 KB per block
    Wave wv;
    // (wave-uniform by construction; saying so keeps the LDS-DMA destinations in scalar registers)
    wv.rows = (lds_double*)(size_t)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_double*)&lds_rows[threadIdx.x >> 6][0]);
    __shared__ int lds_keys[4][64];
    wv.keys = (lds_int*)(size_t)__builtin_amdgcn_readfirstlane((int)(size_t)(lds_int*)&lds_keys[threadIdx.x >> 6][0]);
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
    if (!COUNT) {
        const bool tiny = Lp->wide_ok != 0 && n_in <= RTX_WIDE_MAX_RAYS;
        if (tiny != (WIDE == 2)) return;
    }
    // The scene's first object records, copied to LDS when the block starts: evaluating one of them
    // needs no fetch.  ctx: five doubles of LDS per lane for what a shadow ray needs again after its
    // traversal -- its light's distance (at every change of object) and, if it ends unoccluded,
    // pixel and weight -- taken from the ray record together with origin and direction.
    __shared__ __attribute__((aligned(16))) double lds_objs[RTX_LDS_OBJS * (sizeof(Obj) / 8)];
    ObjCache oc;
    oc.lds = (const Obj*)lds_objs;
    oc.n = min(S.n_meshes + S.n_insts, RTX_LDS_OBJS);
    __shared__ double lds_ctx[256 * 5];
    oc.ctx = lds_ctx + threadIdx.x * 5;
    for (unsigned i = threadIdx.x; i < (unsigned)oc.n * (unsigned)(sizeof(Obj) / 16); i += blockDim.x)
        ((v2d*)lds_objs)[i] = ((const RTX_GLOBAL v2d*)S.objs)[i];
    __syncthreads();
    Tally tl = {};
    const long long clk_begin = COUNT ? clock64() : 0;

    // Rays are claimed in chunks (one atomic per chunk per wave) and handed to lanes as they
    // fall idle.  A thin generation is latency bound: smaller chunks spread it over more waves.
    // Chunks shrink as the generation drains (guided scheduling: a share of what the last claim
    // saw remaining), so that no wave is left with several rounds of rays when the others run dry.
    const unsigned n_waves = gridDim.x * (blockDim.x >> 6);
    unsigned chunk = (n_in + 2u * n_waves - 1u) / (2u * n_waves);
    const unsigned chunk_max = Lp->chunk_max;
    chunk = chunk < 1u ? 1u : (chunk > chunk_max ? chunk_max : chunk);
    unsigned chunk_next = 0, chunk_end = 0;      // wave-uniform
    bool drained = false;                        // wave-uniform: the generation has no rays left

    const size_t lane_global = (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64u + lane;
    glob_v4i* const stk = (glob_v4i*)Lp->stacks + lane_global * RTX_STACK;
    unsigned* const inst_bits = Lp->inst_bits + lane_global * Lp->inst_words;
    const unsigned n_summary = rtx_summary_words(S.n_insts);
    const int service_threshold = (int)Lp->service_threshold;
    const bool thin = gen != 0 && n_in < RTX_SCATTER_THIN_RAYS;
    const int wide_lanes = thin ? RTX_WIDE_LANES_THIN : RTX_WIDE_LANES, wide_age = thin ? RTX_WIDE_AGE_THIN : RTX_WIDE_AGE;
    const char* const blas = S.blas;
    const bool cull = S.cull != 0;
    const double isect_eps = S.isect_eps;
    Lane L;
    L.state = ST_EMPTY; L.cur_cnt = 0; L.cur_ref = 0; L.sp = 0; L.slot = 0; L.is_shadow = 0; L.btri = -1; L.obj = 0;
    L.top = (v4i)(0); L.t_min = 0.0; L.dist = 0.0; L.hit = 0; L.ro = mk(0, 0, 0); L.rd = mk(0, 0, 1); L.ix = L.iy = L.iz = 0.0; L.par = 0;
    L.o = mk(0, 0, 0); L.d = mk(0, 0, 1); L.wix = L.wiy = L.wiz = 0.0; L.orig_t_min = 0.0;
    L.it = 0; L.word = 0; L.whi = -1; L.bits = 0u; L.sword = 0; L.sbits = 0u; L.age = 0; L.oflags = 0; L.omat = -1; L.omesh = 0;

    const int q_node = (int)Lp->quorum[0], q_leaf = (int)Lp->quorum[1], q_eval = (int)Lp->quorum[2], q_fin = (int)Lp->quorum[3];
    bool run_all = false;           // wave-uniform: the last step made no progress, or the generation has drained

    while (true) {
        RecOut out;
        out.rec = (RTX_GLOBAL Hit*)Lp->hb.recs + L.slot; out.on = L.is_shadow == 0; out.ids = (gen == 0 && aov != nullptr);
        // Top of the loop: pick next objects and evaluate those whose record is at hand -- fetched by
        // the last step into the lane's row (still intact: nothing has been fetched since), or one of
        // the scene's first objects in the block's LDS copy, which costs no step at all; a lane may
        // get through several of those here.
        const long long t0 = COUNT ? clock64() : 0;
        bool progress_top = __ballot(L.state == ST_ADV) != 0ull;
        for (int rep = 0; rep < RTX_EVAL_REPS; rep++) {
            if (L.state == ST_ADV) advance<COUNT, FAST>(S, oc, L, out, inst_bits);
            const int n_eval = __popcll(__ballot(L.state == ST_EVAL));
            if (n_eval == 0) break;
            const int n_at_work = __popcll(__ballot(L.state >= ST_ADV));
            if (!(run_all || drained || n_eval >= min(q_eval, (n_at_work + 2) / 3))) break;
            progress_top = true;
            if (L.state == ST_EVAL) {
                RecView E;
                if (L.obj < oc.n) { E.base = (const lds_v2d*)(lds_double*)lds_objs + (unsigned)L.obj * (unsigned)(sizeof(Obj) / 16); E.sw = 0u; }
                else { E.base = (const lds_v2d*)(wv.rows + lane * RTX_ROW_DOUBLES); E.sw = lane & 15u; }
                // ---- rayHitsMesh's prologue for object L.obj (utils.cpp:418-494) + the root test of
                // bvh.cpp:132.  rec: inverse matrix, world box, root box, root, flags, mesh, material.
                double hdr[4];                       // root, root count | flags, mesh | material
                E.read<2>(14u, hdr);
                const int flags = __double2loint(hdr[1]);
                bool go = !(flags & OBJ_SKIP);
                const Slab ws = world_slab(L);
                V3 ro = L.o, rd = L.d;
                double bound = L.t_min;
                if (go && (flags & OBJ_XF)) {
                    if (flags & OBJ_BOUNDS) {                                // utils.cpp:446-466, 760-766
                        if (COUNT) tl.world_box++;
                        double wb[6], enter;
                        E.read<3>(8u, wb);
                        if (!box_enter(wb, wb + 3, ws, enter) || !(L.t_min >= enter)) go = false;
                    }
                    if (go) {
                        if (COUNT) tl.obj_trav++;
                        double inv[16];
                        E.read<8>(0u, inv);
                        ro = xf_point(inv, L.o);                             // utils.cpp:481-489
                        const V3 od = xf_dir(inv, L.d);
                        rd = normalize_f32(od);
                        if (FAST) {
                            // consistent mode: the bound goes to object space with the ray (the divisor is the
                            // factor normalize_f32 applied to the direction)
                            bound = L.t_min / (1.0 / (double)sqrtf((float)(od.x * od.x + od.y * od.y + od.z * od.z)));
                        }
                    }
                }
                if (go) {
                    // (an untransformed object is met by the world-space ray: same constants, no second set of divisions)
                    const Slab sl = (flags & OBJ_XF) ? make_slab(ro, rd) : ws;
                    double rb[6], enter;
                    E.read<3>(11u, rb);
                    if (!box_enter(rb, rb + 3, sl, enter) || !(bound >= enter)) go = false;
                    if (go) {
                        L.orig_t_min = L.t_min;
                        if (FAST) L.t_min = bound;
                        L.is_shadow &= 1;
                        if (!COUNT) L.age = 0;       // (steps of the job; with counters on: of the ray)
                        if (FAST && L.is_shadow) {
                            // consistent mode: a shadow ray's bound starts at the light's distance, so the first
                            // accepted triangle of any job, transformed or not, decides it
                            L.is_shadow = 3;
                            L.orig_t_min = bound;
                        } else if (!COUNT && L.is_shadow && !(flags & OBJ_XF)) {
                            // In an untransformed mesh an accepted triangle in front of the light decides a
                            // shadow ray (see advance), so the job may stop there (bit 1 of is_shadow;
                            // orig_t_min, unused by untransformed jobs, carries the light's distance).
                            // Transformed jobs decide acceptance only at their end (utils.cpp:559-570).
                            L.is_shadow = 3;
                            L.orig_t_min = oc.ctx[0];
                        }
                        L.ro = ro; L.rd = rd; L.ix = sl.ix; L.iy = sl.iy; L.iz = sl.iz; L.par = (L.par & 0x70) | sl.par;
                        L.cur_ref = __double2loint(hdr[0]); L.cur_cnt = __double2hiint(hdr[0]); L.sp = 0; L.btri = -1;
                        L.oflags = flags; L.omesh = __double2hiint(hdr[1]); L.omat = __double2loint(hdr[2]);
                        L.state = ST_TRAV;
                    }
                }
                if (!go) L.state = ST_ADV;
            }
        }
        if (COUNT) tl.clk_pop += (unsigned long long)(clock64() - t0);
        const int st = L.state;
        const unsigned long long m_work = __ballot(st >= ST_ADV);
        const unsigned long long m_done = __ballot(st == ST_DONE);
        const unsigned long long m_empty = __ballot(st == ST_EMPTY);
        const int n_service = __popcll(m_done) + (drained ? 0 : __popcll(m_empty));

        if (n_service > 0 && (m_work == 0ull || n_service >= service_threshold)) {
            // ---- service pass: retire finished rays, hand fresh ones to idle lanes ---------------
            const long long c0 = COUNT ? clock64() : 0;
            for (int rep = 0; rep < RTX_SERVICE_REPS; rep++) {
                if (L.state == ST_DONE) {
                    const bool hit = L.hit != 0;
                    if (L.is_shadow) {                                            // utils.cpp:1083
                        if (!(hit && (FAST || L.dist < oc.ctx[0])))
                            add_color(accum, (unsigned)__double2loint(oc.ctx[4]), mk(oc.ctx[1], oc.ctx[2], oc.ctx[3]));
                    } else {
                        RTX_GLOBAL Hit* rec = (RTX_GLOBAL Hit*)Lp->hb.recs + L.slot;
                        if (gen == 0 && aov != nullptr) {
                            int px, py; unsigned pixel;
                            primary_pixel(F, L.slot, px, py, pixel);
                            rtx_hit_id h;
                            h.kind = 0; h.container = -1; h.face = -1; h.src = -1; h.distance = 0.0;
                            if (hit) {
                                // read back what this lane stored (bypassing the vector L1)
                                Hit* r = Lp->hb.recs + L.slot;
                                h.kind = __hip_atomic_load(&r->kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                h.container = __hip_atomic_load(&r->container, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int face = __hip_atomic_load(&r->face, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int src = __hip_atomic_load(&r->src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                h.face = (h.kind == RTX_KIND_MESH || h.kind == RTX_KIND_TRIANGLE) ? face : -1;
                                h.src = h.kind == RTX_KIND_MESH ? src : h.container;
                                h.distance = L.dist;
                            }
                            aov[(size_t)py * F.width + px] = h;
                        }
                        rec->code = hit ? 1 : -1;
                    }
                    L.state = ST_EMPTY;
                }
                const unsigned long long me_ = __ballot(L.state == ST_EMPTY);
                if (me_ == 0ull || drained) break;
                if (chunk_next == chunk_end) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&me->head_trace, chunk);
                    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
                    if (base >= n_in) { drained = true; chunk_next = chunk_end = 0; }
                    else {
                        chunk_next = base; chunk_end = min(base + chunk, n_in);
                        const unsigned share = (n_in - chunk_end) / (2u * n_waves);
                        chunk = share < 16u ? 16u : (share > chunk_max ? chunk_max : share);
                        if (chunk > n_in) chunk = n_in;
                    }
                }
                if (drained) break;
                const unsigned take = min((unsigned)__popcll(me_), chunk_end - chunk_next);
                const unsigned rank = (unsigned)__popcll(me_ & ((1ull << lane) - 1ull));
                bool fresh = false;
                if (L.state == ST_EMPTY && rank < take) {
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
                if (fresh) {
                    if (COUNT) { tl.rays++; if (gen == 0) tl.primary++; else if (L.is_shadow) tl.shadow++; else tl.secondary++; }
                    RecOut o2;
                    o2.rec = (RTX_GLOBAL Hit*)Lp->hb.recs + L.slot; o2.on = L.is_shadow == 0; o2.ids = (gen == 0 && aov != nullptr);
                    // consistent mode: a shadow ray looks no further than its light
                    ray_begin<COUNT>(S, L, o2, FAST && L.is_shadow ? oc.ctx[0] : DBL_MAX, tl);
                }
            }
            if (COUNT) { tl.clk_serv += (unsigned long long)(clock64() - c0); tl.passes++; }
            continue;
        }
        if (m_work == 0ull) break;

        // the generation has drained and this wave holds its last few jobs: finish them wave-wide
        if (!COUNT && WIDE != 0 && drained && __popcll(m_work) <= wide_lanes) {
            const unsigned long long m_old = __ballot(st == ST_TRAV && (WIDE == 2 || L.age >= wide_age));
            // (the wave-wide traversal uses every lane's row: no lane may be holding a record in its own)
            const bool rows_free = __ballot(st == ST_TRAV_R || st == ST_EVAL || st == ST_FIN_R || st == ST_FIN2_R) == 0ull;
            if (m_old != 0ull && rows_free) {
                wide_jobs<COUNT, FAST>(S, oc, wv, L, stk, inst_bits, m_old);
                continue;
            }
        }

        // ---- one step -----------------------------------------------------------------------------
        const long long s0 = COUNT ? clock64() : 0;
        // the record this lane's state needs (-1: none, it has one, or it is in the block's LDS copy)
        int key = -1;
        if (st == ST_TRAV) key = L.cur_ref;
        else if (st == ST_FIN) key = L.btri;
        else if (st == ST_OBJ) key = S.obj_key0 + 4 * L.obj;
        else if (st == ST_FIN2 && L.obj >= oc.n) key = S.obj_key0 + 4 * L.obj + 2;
        // bitmap walkers: one atomic exchange, taking the word and leaving it clear; the value is
        // there when the step's records are.  A lane that fetches the record of a bitmap word's last
        // candidate takes the next word (or summary word) in the same step.
        unsigned pending = 0u;
        int pending_is = st == ST_WORD ? 1 : (st == ST_SUM || st == ST_CLEAN ? 2 : 0);      // 1: bitmap word, 2: summary word
        if (st == ST_CLEAN) L.sword++;
        if (st == ST_OBJ && L.it > S.n_meshes && L.bits == 0u) {
            if (L.sbits != 0u) {
                L.word = L.sword * 32 + (__ffs((int)L.sbits) - 1);
                L.sbits &= L.sbits - 1u;
                pending_is = 1;
            } else if (L.sword < L.whi) {
                L.sword++;
                pending_is = 2;
            }
        }
        if (pending_is == 2) pending = atomicExch(inst_bits + L.sword, 0u);          // summary words come first in the lane's area
        else if (pending_is == 1) pending = atomicExch(inst_bits + n_summary + L.word, 0u);
        const unsigned long long m_fetch = __ballot(key != -1);
        bool progress = progress_top || m_fetch != 0ull || __ballot(pending_is != 0) != 0ull;

        // A wave with only a lane or two still at work is on the frame's critical path (the
        // longest rays of a generation); it loads the records directly, one memory round trip
        // without the detour through the other lanes.  Otherwise the wave fetches cooperatively.
        RecView R;
        R.base = (const lds_v2d*)(wv.rows + lane * RTX_ROW_DOUBLES); R.sw = lane & 15u;
        if (m_fetch == 0ull) {
            // (no record to fetch: every lane at work has its record and waits for its kind's turn)
        } else if (__popcll(m_work) <= RTX_SPARSE_LANES) {
            if (key != -1) {
                const RTX_GLOBAL v2d* gp = (const RTX_GLOBAL v2d*)(blas + ((size_t)(unsigned)key << 7));
                lds_v2d* row = (lds_v2d*)(wv.rows + lane * RTX_ROW_DOUBLES);
#pragma unroll
                for (int p = 0; p < 16; p++) row[(unsigned)p ^ R.sw] = gp[p];
            }
        } else {
            coop_fetch(wv, key, blas);
        }
        if (COUNT) tl.clk_fetch += (unsigned long long)(clock64() - s0);
        // what arrived
        if (st == ST_TRAV) L.state = ST_TRAV_R;
        else if (st == ST_FIN) L.state = ST_FIN_R;
        else if (st == ST_FIN2) L.state = ST_FIN2_R;
        else if (st == ST_OBJ) {
            L.state = ST_EVAL;                   // evaluated at the top of the loop
            if (pending_is == 1) L.bits = pending;
            else if (pending_is == 2) L.sbits = pending;
        } else if (st == ST_SUM) { L.sbits = pending; L.state = ST_ADV; }
        else if (st == ST_WORD) { L.bits = pending; L.state = ST_ADV; }
        else if (st == ST_CLEAN) {
            unsigned sb = pending;
            while (sb != 0u) {
                const int w = L.sword * 32 + (__ffs((int)sb) - 1);
                sb &= sb - 1u;
                __hip_atomic_store(inst_bits + n_summary + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (L.sword >= L.whi) L.state = ST_DONE;
        }

        // which kinds of work are run in this step
        const int s2 = L.state;
        const bool is_node = s2 == ST_TRAV_R && L.cur_cnt == 0, at_leaf = s2 == ST_TRAV_R && L.cur_cnt > 0;
        const int n_node = __popcll(__ballot(is_node)), n_leaf = __popcll(__ballot(at_leaf));
        const int n_fin1 = __popcll(__ballot(s2 == ST_FIN_R)), n_fin2 = __popcll(__ballot(s2 == ST_FIN2_R));
        const int q_cap = (__popcll(m_work) + 2) / 3;
        const bool everything = run_all || drained;
        const bool run_node = n_node > 0 && (everything || n_node >= min(q_node, q_cap));
        const bool run_leaf = n_leaf > 0 && (everything || n_leaf >= min(q_leaf, q_cap));
        const bool run_fin1 = n_fin1 > 0 && (everything || n_fin1 >= min(q_fin, q_cap));
        const bool run_fin2 = n_fin2 > 0 && (everything || n_fin2 >= min(q_fin, q_cap));
        progress = progress || run_node || run_leaf || run_fin1 || run_fin2;
        if (COUNT && lane == 0) {
            if (run_node) { tl.inner_steps++; tl.inner_lanes += (unsigned)n_node; }
            if (run_leaf) { tl.leaf_steps++; tl.leaf_lanes += (unsigned)n_leaf; }
            tl.sec[0]++; tl.sec[1] += (unsigned)(n_node + n_leaf); tl.sec[2] += (unsigned)__popcll(__ballot(st == ST_OBJ || st == ST_EVAL));
            tl.sec[3] += (unsigned)__popcll(__ballot(pending_is != 0 && st != ST_OBJ)); tl.sec[4] += (unsigned)(n_fin1 + n_fin2);
            tl.sec[5] += (unsigned)__popcll(m_done); tl.sec[6] += (unsigned)__popcll(m_empty);
            tl.sec[7] += (unsigned)__popcll(m_fetch);
        }
        const bool trav = (is_node && run_node) || (at_leaf && run_leaf);
        const bool fin = run_fin1 && s2 == ST_FIN_R, fin2 = run_fin2 && s2 == ST_FIN2_R;
        if (fin2 && L.obj < oc.n) {
            R.base = (const lds_v2d*)(lds_double*)lds_objs + (unsigned)L.obj * (unsigned)(sizeof(Obj) / 16) + 16u;
            R.sw = 0u;
        }
        // rays with an axis-parallel component are rare: the wave takes the general box test
        // only when one of its traversing lanes has one
        const bool any_par = __ballot(trav && (L.par & 7) != 0) != 0ull;

        if (trav) {
            bool need_pop = false;
            L.age++;
            L.state = ST_TRAV;                   // (needs its next record, unless the job ends below)
            Slab sl;
            sl.ox = L.ro.x; sl.oy = L.ro.y; sl.oz = L.ro.z; sl.ix = L.ix; sl.iy = L.iy; sl.iz = L.iz; sl.par = L.par & 7;
            if (!at_leaf) {
                // ---- collapsed inner node: up to four grandchildren of a binary node ------------
                // Visit order and visited set of bvh.cpp:104-185: children ordered near/far by the
                // sign of dir[splitAxis] (near first); a node is entered iff its box test passes with
                // the bound current when the reference would pop it.  A child's box is evaluated at its
                // (grand)parent; children that are not entered at once are pushed with their entry
                // distance and re-checked against the (possibly smaller) bound when popped -- the
                // predicate the reference evaluates at that moment.  Skipping the intermediate binary
                // node's own test changes nothing: the slab arithmetic is monotone, so a box that
                // passes lies inside a parent box that passes too, and the bound only shrinks between
                // the parent's turn and the child's.
                if (COUNT) { if (L.obj == -1) tl.tlas_visits++; else tl.node_visits++; }
                double tail[6];                  // refs, counts, split axes
                R.read<3>(12u, tail);
                Cand c[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    c[k].ref = k & 1 ? __double2hiint(tail[k >> 1]) : __double2loint(tail[k >> 1]);
                    c[k].cnt = k & 1 ? __double2hiint(tail[2 + (k >> 1)]) : __double2loint(tail[2 + (k >> 1)]);
                    double bx[6];
                    R.read<3>(3u * (unsigned)k, bx);
                    double e;
                    bool hit;
                    if (any_par) hit = box_enter(bx, bx + 3, sl, e);
                    else hit = box_enter_nopar(bx, bx + 3, sl, e);
                    c[k].e = e;
                    const bool pass = hit & (L.t_min >= e);
                    c[k].ok = pass & (c[k].cnt >= 0);
                    if (pass && c[k].cnt == -2) {
                        // TLAS walk: the slot is a mesh instance whose world box the ray can meet
                        // (two fire-and-forget atomics: own plain stores may sit behind a stale L1 line)
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
                const int axis0 = __double2loint(tail[4]), axisL = __double2hiint(tail[4]), axisR = __double2loint(tail[5]);
                const double d0 = axis0 == 0 ? L.rd.x : (axis0 == 1 ? L.rd.y : L.rd.z);
                const double dl = axisL == 0 ? L.rd.x : (axisL == 1 ? L.rd.y : L.rd.z);
                const double dr = axisR == 0 ? L.rd.x : (axisR == 1 ? L.rd.y : L.rd.z);
                // a leaf child occupies the first slot of its pair: nothing to order there
                cand_swap(c[0], c[1], !(dl >= 0.0) && c[1].cnt >= 0);
                cand_swap(c[2], c[3], !(dr >= 0.0) && c[3].cnt >= 0);
                const bool right_first = !(d0 >= 0.0);
                cand_swap(c[0], c[2], right_first);
                cand_swap(c[1], c[3], right_first);
                // c[0..3] is now near -> far.  The nearest passing child is entered; the others
                // are pushed, farthest first, each with its entry distance for the re-check at pop.
                // Stack entries are 16 bytes (entry distance, ref, count) in a per-lane region of global
                // memory; the top entry is kept in registers (Lane::top), memory holds the entries below.
                const bool any0 = c[0].ok, any1 = any0 | c[1].ok, any2 = any1 | c[2].ok;
#pragma unroll
                for (int k = 3; k >= 1; k--) {
                    const bool earlier = k == 3 ? any2 : (k == 2 ? any1 : any0);
                    // (a full stack cannot happen with trees the host has accepted; if it does, the
                    // frame is reported as failed instead of silently losing a subtree)
                    if (c[k].ok && earlier && L.sp >= RTX_STACK) *S.limit_flag = 1u;
                    if (c[k].ok && earlier && L.sp < RTX_STACK) {
                        if (L.sp > 0) stk[L.sp - 1] = L.top;
                        L.top.x = __double2loint(c[k].e); L.top.y = __double2hiint(c[k].e); L.top.z = c[k].ref; L.top.w = c[k].cnt;
                        L.sp++;
                    }
                }
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
                bool culled0 = false, culled1 = false;
                if (cull) {                                                                   // utils.cpp:322-326
                    double g0[4], g1[4];
                    R.read<2>(5u, g0); R.read<2>(13u, g1);
                    culled0 = L.rd.x * g0[0] + L.rd.y * g0[1] + L.rd.z * g0[2] > 0.0;
                    culled1 = L.rd.x * g1[0] + L.rd.y * g1[1] + L.rd.z * g1[2] > 0.0;
                }
                double t, b, g2;
                {
                    double td[10];
                    R.read<5>(0u, td);
                    if (!culled0 && tri_test(td, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.t_min = t; L.btri = ti; }
                }
                if (two && !culled1) {
                    double tq[10];
                    R.read<5>(8u, tq);
                    if (tri_test(tq, L.ro, L.rd, L.t_min, isect_eps, t, b, g2)) { L.t_min = t; L.btri = ti + 1; }
                }
                L.cur_ref += 2; L.cur_cnt -= 2;
                need_pop = L.cur_cnt <= 0;
                if (!COUNT && (L.is_shadow & 2) && L.btri >= 0 && L.t_min < L.orig_t_min) {
                    need_pop = true; L.sp = 0;         // occluder found: the job ends here
                }
            }
            if (need_pop) {
                int live = 0;
                if (L.sp > 0) {
                    // First pop peeled off the loop: it consumes only registers, and the refill load
                    // it issues stays in flight (inside the loop every iteration has to wait for the
                    // previous iteration's refill).  Everything that reads the old top comes before
                    // the load, so the load can target the registers that hold it.
                    int pass = L.t_min >= __hiloint2double(L.top.y, L.top.x) ? 1 : 0;
                    if (pass) { L.cur_ref = L.top.z; L.cur_cnt = L.top.w; live = 1; }
                    asm volatile("" : "+v"(pass), "+v"(L.cur_ref), "+v"(L.cur_cnt), "+v"(live));   // old top is dead from here
                    --L.sp;
                    L.top = stk[L.sp > 0 ? L.sp - 1 : 0];          // (entry 0 is always readable)
                    if (!pass) {
                        // The entry failed its re-check.  After a hit has pulled the bound in, most of
                        // what is left on the stack fails too, one entry after the other: the entries
                        // are examined four at a time (adjacent in memory, one round trip; the first
                        // of them is the refill already in flight), from the top down.
                        while (L.sp > 0) {
                            const v4i e1 = stk[L.sp >= 2 ? L.sp - 2 : 0];
                            const v4i e2 = stk[L.sp >= 3 ? L.sp - 3 : 0];
                            const v4i e3 = stk[L.sp >= 4 ? L.sp - 4 : 0];
                            const bool p0 = L.t_min >= __hiloint2double(L.top.y, L.top.x);
                            const bool p1 = L.sp >= 2 && L.t_min >= __hiloint2double(e1.y, e1.x);
                            const bool p2 = L.sp >= 3 && L.t_min >= __hiloint2double(e2.y, e2.x);
                            const bool p3 = L.sp >= 4 && L.t_min >= __hiloint2double(e3.y, e3.x);
                            const int k = p0 ? 0 : (p1 ? 1 : (p2 ? 2 : (p3 ? 3 : 4)));     // first that passes; 4: none
                            if (k < 4) {
                                L.cur_ref = k == 0 ? L.top.z : (k == 1 ? e1.z : (k == 2 ? e2.z : e3.z));
                                L.cur_cnt = k == 0 ? L.top.w : (k == 1 ? e1.w : (k == 2 ? e2.w : e3.w));
                                live = 1;
                            }
                            const int used = k < 4 ? k + 1 : 4;
                            L.sp = L.sp > used ? L.sp - used : 0;
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
                if (!live) job_end<COUNT, FAST>(oc, L);
            }
        } else if (fin) {
            // ---- what rayHitsMesh does after bvh->traverse returned a hit (utils.cpp:533-668), first
            // part: the fields the winning triangle test wrote (utils.cpp:370-381, bvh.cpp:162).
            // beta and gamma are recomputed with the arithmetic that accepted the triangle (same
            // inputs, same bits).  rec: the triangle record.
            const int tri = L.btri - S.tri_key0;
            double T[14];
            R.read<7>(0u, T);
            const int face = __double2loint(T[9]);
            const V3 gn = mk(T[10], T[11], T[12]);
            double t, beta, gamma;
            tri_test(T, L.ro, L.rd, INFINITY, isect_eps, t, beta, gamma);
            const V3 a = mk(T[0], T[1], T[2]), e1 = mk(T[3], T[4], T[5]), e2 = mk(T[6], T[7], T[8]);
            const V3 point = vadd(vadd(a, sscale(beta, e1)), sscale(gamma, e2));
            if (!FAST) {
                out.bary(beta, gamma, face, tri);
                out.kind(RTX_KIND_MESH);
            }
            if (!(L.oflags & OBJ_XF)) {
                if (FAST) { out.bary(beta, gamma, face, tri); out.kind(RTX_KIND_MESH); }
                out.geom(L.t_min, point, gn, gn);
                out.owner(L.omat, L.omesh);
                out.src(L.obj);
                L.dist = L.t_min; L.hit = 1;
                L.state = ST_ADV;
            } else {
                // transformed: the matrices come with the next step; the job's ray is no longer needed
                L.ro = point; L.rd = gn; L.ix = beta; L.iy = gamma; L.cur_ref = face; L.cur_cnt = tri;
                L.state = ST_FIN2;
            }
        } else if (fin2) {
            // ---- second part, transformed objects (utils.cpp:559-659).  rec: object -> world
            // matrix (doubles 0-15), rows 0-2 of the normal matrix (16-27).
            const V3 point = L.ro, gn = L.rd;
            const double beta = L.ix, gamma = L.iy;
            const int face = L.cur_ref, tri = L.cur_cnt;
            double m[16];
            R.read<8>(0u, m);
            const V3 wp = xf_point(m, point);
            const double dx = wp.x - L.o.x, dy = wp.y - L.o.y, dz = wp.z - L.o.z;
            const double wd = (double)sqrtf((float)(dx * dx + dy * dy + dz * dz));
            if (wd >= L.orig_t_min) {
                if (FAST) {
                    // consistent mode: a hit that is not nearer than the bound leaves no trace
                    L.t_min = L.orig_t_min;
                } else {                          // utils.cpp:567-570: bound restored, record is not
                    out.geom(L.t_min, point, gn, gn);
                    out.owner(L.omat, L.omesh);
                    L.dist = L.t_min;
                    L.t_min = L.orig_t_min;
                }
            } else if (!out.on) {                  // (shadow ray: no record)
                L.dist = wd; L.t_min = wd; L.hit = 1;
            } else {
                if (FAST) { out.bary(beta, gamma, face, tri); out.kind(RTX_KIND_MESH); }
                double nrm[12];
                R.read<6>(8u, nrm);
                V3 wg = normalize_f32(xf_dir(nrm, gn));
                V3 ws;
                if (L.oflags & OBJ_SMOOTH_VN) {          // utils.cpp:589-621
                    const double u = beta, v = gamma, w = 1.0 - u - v;
                    const RTX_GLOBAL double* vn = (const RTX_GLOBAL double*)S.tri_vn + RTX_VN_STRIDE * (size_t)tri;
                    double nx = w * vn[0] + u * vn[3] + v * vn[6];
                    double ny = w * vn[1] + u * vn[4] + v * vn[7];
                    double nz = w * vn[2] + u * vn[5] + v * vn[8];
                    const double nlen = (double)sqrtf((float)(nx * nx + ny * ny + nz * nz));
                    const double ninv = 1.0 / nlen;
                    nx *= ninv; ny *= ninv; nz *= ninv;
                    ws = xf_dir(nrm, mk(nx, ny, nz));
                } else {
                    ws = xf_dir(nrm, gn);
                }
                {
                    const double slen = (double)sqrtf((float)(ws.x * ws.x + ws.y * ws.y + ws.z * ws.z));
                    const double sinv = 1.0 / slen;
                    ws.x *= sinv; ws.y *= sinv; ws.z *= sinv;
                }
                if (dot(wg, L.d) > 0.0) { wg = vneg(wg); ws = vneg(ws); }
                out.geom(wd, wp, wg, ws);
                out.owner(L.omat, -2);
                out.src(L.obj);
                L.dist = wd;
                L.t_min = wd;
                L.hit = 1;
            }
            L.state = ST_ADV;
        }
        __builtin_amdgcn_wave_barrier();
        // a step in which nothing moved (every lane waiting for a quorum that cannot form): run every kind next time
        run_all = !progress;
        if (COUNT) tl.clk_trav += (unsigned long long)(clock64() - s0);
    }

    if (COUNT && counters) {
        const bool l0 = lane == 0;
        unsigned long long v[30] = {tl.rays, tl.primary, tl.shadow, tl.secondary, tl.node_visits, tl.tri_tests,
                                    tl.world_box, tl.obj_trav, tl.tlas_visits, tl.inner_steps, tl.inner_lanes,
                                    tl.leaf_steps, tl.leaf_lanes, l0 ? tl.passes : 0u, l0 ? tl.clk_trav : 0ull,
                                    l0 ? tl.clk_serv : 0ull, l0 ? (unsigned long long)(clock64() - clk_begin) : 0ull,
                                    l0 ? 1ull : 0ull, l0 ? tl.clk_fetch : 0ull, l0 ? tl.clk_pop : 0ull, 0ull, 0ull,
                                    tl.sec[0], tl.sec[1], tl.sec[2], tl.sec[3], tl.sec[4], tl.sec[5], tl.sec[6], tl.sec[7]};
#pragma unroll
        for (int k = 0; k < 30; k++) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            if (lane == 0 && x) atomicAdd(((unsigned long long*)counters) + k, x);
        }
    }
}

}  // namespace rtxdev
