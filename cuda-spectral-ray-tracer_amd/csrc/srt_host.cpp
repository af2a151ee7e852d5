// srt_host.cpp -- host half of the C-ABI: scene flattening, tri::init precompute, the two BVH builders,
// spectra baking, camera maths.  Index based (std::vector), no device heap, no HIP.
//
// fp32 expressions that feed the renderer keep the reference's operation order (file:line cited) and
// this file is compiled with -ffp-contract=off.
#include "srt_host.h"

#include <math.h>
#include <float.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "srt_cie_data.h"
#include "srt_color_consts.h"
#include "srt_powf.h"

namespace srt {

// ------------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------------
static std::mutex g_err_mu;
static std::string g_err;
void set_global_error(const std::string &msg) { std::lock_guard<std::mutex> lk(g_err_mu); g_err = msg; }
const char *global_error() { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------------
// small float3 helpers; association order as math/vec3.cuh
// ------------------------------------------------------------------------------------------------------
static inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
static inline F3 f3(const float *p) { return F3{p[0], p[1], p[2]}; }
static inline F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline F3 neg(F3 a) { return f3(-a.x, -a.y, -a.z); }
static inline F3 mul(float t, F3 v) { return f3(t * v.x, t * v.y, t * v.z); }
static inline F3 div(F3 v, float t) { return mul(1 / t, v); }                         // vec3.cuh:144-147
static inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    // vec3.cuh:149
static inline F3 cross3(F3 u, F3 v) {                                                 // vec3.cuh:155
    return f3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static inline float len3(F3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }
static inline F3 unit3(F3 v) { return div(v, len3(v)); }
static inline float comp(F3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }
static inline void put(float *dst, F3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// ------------------------------------------------------------------------------------------------------
// host XORWOW
// ------------------------------------------------------------------------------------------------------
HostRng::HostRng(uint64_t seed) {
    uint32_t s0 = ((uint32_t)seed) ^ 0xaad26b49u, s1 = ((uint32_t)(seed >> 32)) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    d = 6615241u + t1 + t0;
    v[0] = 123456789u + t0; v[1] = 362436069u ^ t0; v[2] = 521288629u + t1; v[3] = 88675123u ^ t1; v[4] = 5783321u + t0;
}
uint32_t HostRng::next() {
    uint32_t t = v[0] ^ (v[0] >> 2);
    v[0] = v[1]; v[1] = v[2]; v[2] = v[3]; v[3] = v[4];
    v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
    d += 362437u;
    return v[4] + d;
}
float HostRng::uniform() { return (float)next() * 2.3283064e-10f + (2.3283064e-10f / 2.0f); }
float HostRng::range(float mn, float mx) { float w = mx - mn; float u = uniform(); return u * w + mn; }
int HostRng::rand_int(int mn, int mx) { return (int)ceilf(range((float)(mn - 1), (float)(mx - 1))); }

// ------------------------------------------------------------------------------------------------------
// tri::init (primitives/tri.cu:47-84)
// ------------------------------------------------------------------------------------------------------
static inline void plane_axes(uint32_t aa_plane, int &w, int &h) {   // tri.cu:157-176
    switch (aa_plane) {
    case SRT_AAP_YZ: w = 1; h = 2; break;
    case SRT_AAP_XZ: w = 0; h = 2; break;
    default:         w = 0; h = 1;
    }
}

void tri_precompute(const srt_tri_in &t, TriRecord &out) {
    const F3 v0 = f3(t.v0), v1 = f3(t.v1), v2 = f3(t.v2);
    const F3 n = unit3(cross3(sub(v1, v0), sub(v2, v0)));
    // perp tests are |dot(normal, axis)| < 1e-8 with the full three-term dot of the reference (tri.cu:59-61)
    const bool perp_x = fabsf(dot3(n, f3(1.f, 0.f, 0.f))) < 1e-8f;
    const bool perp_y = fabsf(dot3(n, f3(0.f, 1.f, 0.f))) < 1e-8f;
    const bool perp_z = fabsf(dot3(n, f3(0.f, 0.f, 1.f))) < 1e-8f;
    uint32_t plane = t.aa_plane;                       // sticky unless axis aligned (Q12)
    if (perp_y && perp_z) plane = SRT_AAP_YZ;
    else if (perp_x && perp_z) plane = SRT_AAP_XZ;
    else if (perp_x && perp_y) plane = SRT_AAP_XY;
    out.n[0] = n.x; out.n[1] = n.y; out.n[2] = n.z;
    out.D = dot3(n, v0);                               // tri.cu:79
    out.aa_plane = plane;
    int w, h;
    plane_axes(plane, w, h);
    // clockwise = double_signed_area_2D(v0, v1, v2) >= 0 (tri.cuh:107-110, tri.cu:181)
    const float area = (comp(v0, w) - comp(v2, w)) * (comp(v1, h) - comp(v2, h)) -
                       (comp(v1, w) - comp(v2, w)) * (comp(v0, h) - comp(v2, h));
    out.clockwise = area >= 0;
    // bbox = aabb(v0,v1,v2).pad() (tri.cuh:54-57, aabb.cuh:48-57,92-102)
    for (int a = 0; a < 3; a++) {
        float lo = fminf(comp(v0, a), fminf(comp(v1, a), comp(v2, a)));
        float hi = fmaxf(comp(v0, a), fmaxf(comp(v1, a), comp(v2, a)));
        const float delta = 0.0001f;
        if (!((hi - lo) >= delta)) { float padding = delta / 2; lo = lo - padding; hi = hi + padding; }
        out.box[2 * a] = lo; out.box[2 * a + 1] = hi;
    }
}

// ------------------------------------------------------------------------------------------------------
// BVH builders.  Node semantics are the reference's (bvh/bvh.cuh:24-109): binary, one triangle per leaf,
// leaf box = padded triangle box, internal box = union of the children's boxes (Q22).
// ------------------------------------------------------------------------------------------------------
static void finish_boxes_and_depth(srt_scene &s) {
    // children are always created after their parent, so a reverse sweep is a post-order pass
    // (bvh.cu:311-346 does the same with an explicit stack).
    std::vector<int> internal_depth(s.nodes.size(), 0);
    for (size_t k = s.nodes.size(); k-- > 0;) {
        BvhNode &nd = s.nodes[k];
        if (nd.prim >= 0) {
            memcpy(nd.box, s.rec[nd.prim].box, sizeof(nd.box));
            internal_depth[k] = 0;
        } else {
            const BvhNode &l = s.nodes[nd.left], &r = s.nodes[nd.right];
            for (int a = 0; a < 3; a++) {
                nd.box[2 * a] = fminf(l.box[2 * a], r.box[2 * a]);              // interval.cuh:19-20
                nd.box[2 * a + 1] = fmaxf(l.box[2 * a + 1], r.box[2 * a + 1]);
            }
            internal_depth[k] = 1 + std::max(internal_depth[nd.left], internal_depth[nd.right]);
        }
    }
    s.depth = s.root >= 0 ? internal_depth[s.root] : 0;
}

// bvh::build_bvh (bvh/bvh.cu:206-309) on a permutation of triangle indices instead of tri* pointers.
int build_bvh_reference(srt_scene &s, uint64_t seed) {
    s.nodes.clear(); s.root = -1; s.bvh_valid = false; s.depth = 0;
    const size_t n = s.raw.size();
    if (n == 0) { set_global_error("BVH: empty scene"); return SRT_ERR_BVH; }
    std::vector<int32_t> order(n);
    for (size_t k = 0; k < n; k++) order[k] = (int32_t)k;
    HostRng rng(seed);                                   // scene.cu:12-14
    auto box_min = [&](int32_t tri, int axis) { return s.rec[tri].box[2 * axis]; };
    auto less = [&](int32_t a, int32_t b, int axis) { return box_min(a, axis) < box_min(b, axis); };   // bvh.cuh:180-184

    struct Span { size_t start, end; int32_t node; };
    std::vector<Span> stack;
    s.nodes.emplace_back();
    s.root = 0;
    stack.push_back({0, n, 0});
    while (!stack.empty()) {
        const Span cur = stack.back();
        stack.pop_back();
        const size_t span = cur.end - cur.start;
        if (span == 0) continue;
        if (span == 1) { s.nodes[cur.node].prim = order[cur.start]; continue; }
        const int axis = rng.rand_int(0, 2);             // never 2 (Q14)
        if (span == 2) {
            const int32_t l = (int32_t)s.nodes.size();
            s.nodes.emplace_back(); s.nodes.emplace_back();
            const bool in_order = less(order[cur.start], order[cur.start + 1], axis);
            s.nodes[l].prim = in_order ? order[cur.start] : order[cur.start + 1];
            s.nodes[l + 1].prim = in_order ? order[cur.start + 1] : order[cur.start];
            s.nodes[cur.node].left = l; s.nodes[cur.node].right = l + 1;
            continue;
        }
        // quicksort_primitives (bvh.cu:14-71): Lomuto partition, last element as pivot, explicit stack,
        // left range pushed first (so the right range is partitioned first).  Not stable: the exact
        // permutation decides the topology, hence restated step by step.
        {
            std::vector<int> qs;
            qs.push_back((int)cur.start); qs.push_back((int)cur.end - 1);
            while (!qs.empty()) {
                const int hi = qs.back(); qs.pop_back();
                const int lo = qs.back(); qs.pop_back();
                int p = lo;
                if (lo != hi) {
                    const int32_t pivot = order[hi];
                    int i = lo - 1;
                    for (int j = lo; j < hi; j++)
                        if (less(order[j], pivot, axis)) { i++; std::swap(order[i], order[j]); }
                    std::swap(order[i + 1], order[hi]);
                    p = i + 1;
                }
                if (p - 1 > lo) { qs.push_back(lo); qs.push_back(p - 1); }
                if (p + 1 < hi) { qs.push_back(p + 1); qs.push_back(hi); }
            }
        }
        const size_t mid = cur.start + span / 2;
        const int32_t l = (int32_t)s.nodes.size();
        s.nodes.emplace_back(); s.nodes.emplace_back();
        s.nodes[cur.node].left = l; s.nodes[cur.node].right = l + 1;
        if (stack.size() + 2 > 64) { set_global_error("BVH: build stack exceeds MAX_DEPTH 64 (bvh.cuh:12)"); return SRT_ERR_BVH; }
        stack.push_back({cur.start, mid, l});            // left pushed first, right popped first (bvh.cu:276-296)
        stack.push_back({mid, cur.end, l + 1});
    }
    finish_boxes_and_depth(s);
    s.bvh_valid = true;
    return SRT_OK;
}

// Insertion-based optimisation of a finished tree (after Bittner, Hapala, Havran: "Fast insertion-based optimization of bounding
// volume hierarchies", 2013): take a subtree out, let its sibling take the parent's place, and put it back where it adds the least
// surface area to the tree -- found by a branch-and-bound search from the root over (area added to the ancestors) + (area of the new
// parent).  The search space contains the position the subtree came from, so with plain areas a step never makes the sum of the
// internal nodes' areas (the SAH cost of a one-triangle-per-leaf tree) worse; the objective used here weighs nodes with a leaf child
// by the cost of a FRINGE visit (below), for which the bookkeeping of a step is local (the type changes at the place a subtree
// leaves are not priced) -- a heuristic, judged by the measured records / triangle tests per ray.  Nodes are processed in order of
// decreasing area, `passes` times.
// Leaves stay one triangle each; only the topology above them changes.  The tree is an input of the traversal: results do not
// depend on it (except where two triangles tie exactly in t, Q11 -- and the CPU checker walks the same tree).
// every internal node has two leaf children or none (what build_bvh_sah's even splits produce for an even triangle count)
bool tree_is_paired(const srt_scene &s) {
    if (s.nodes.size() < 3) return false;
    for (const BvhNode &nd : s.nodes)
        if (nd.prim < 0 && (s.nodes[nd.left].prim >= 0) != (s.nodes[nd.right].prim >= 0)) return false;
    return true;
}

static void optimise_bvh_by_reinsertion(srt_scene &s, int passes) {
    const int32_t n = (int32_t)s.nodes.size();
    if (passes <= 0 || n < 7) return;
    // renumber: parents before children, the two children of a node next to each other (what the builders produce and
    // finish_boxes_and_depth relies on)
    auto renumber = [&](int32_t from_root) {
        std::vector<BvhNode> out;
        out.reserve(n);
        std::vector<int32_t> queue;
        out.push_back(s.nodes[from_root]);
        queue.push_back(0);
        for (size_t h = 0; h < queue.size(); h++) {
            const int32_t k = queue[h];
            if (out[k].prim >= 0) continue;
            const int32_t ol = out[k].left, orr = out[k].right, l = (int32_t)out.size();
            out.push_back(s.nodes[ol]); out.push_back(s.nodes[orr]);
            out[k].left = l; out[k].right = l + 1;
            queue.push_back(l); queue.push_back(l + 1);
        }
        s.nodes.swap(out);
        s.root = 0;
    };
    {
        // canonical form first -- the child that holds the smaller triangle index on the left, breadth-first numbering -- so that the
        // steps below (processing order, ties) do not depend on the child order the top-down build chose for its viewpoint: the same
        // triangles give the same topology for every eye
        std::vector<int32_t> min_prim(n);
        for (int32_t k = n; k-- > 0;) {      // (children have larger indices than their parent)
            BvhNode &nd = s.nodes[k];
            if (nd.prim >= 0) { min_prim[k] = nd.prim; continue; }
            if (min_prim[nd.right] < min_prim[nd.left]) std::swap(nd.left, nd.right);
            min_prim[k] = min_prim[nd.left];
        }
        renumber(s.root);
    }
    std::vector<int32_t> parent(n, -1);
    for (int32_t k = 0; k < n; k++)
        if (s.nodes[k].prim < 0) { parent[s.nodes[k].left] = k; parent[s.nodes[k].right] = k; }
    auto area_of = [](const float *b) { const double dx = (double)b[1] - b[0], dy = (double)b[3] - b[2], dz = (double)b[5] - b[4]; return 2.0 * (dx * dy + dy * dz + dz * dx); };
    std::vector<double> area(n);
    for (int32_t k = 0; k < n; k++) area[k] = area_of(s.nodes[k].box);
    int32_t root = s.root;
    auto refit_up = [&](int32_t k) {      // recompute the boxes from node k to the root (stops when a box does not change)
        while (k >= 0) {
            BvhNode &nd = s.nodes[k];
            const BvhNode &l = s.nodes[nd.left], &r = s.nodes[nd.right];
            float nb[6];
            for (int a = 0; a < 3; a++) { nb[2 * a] = fminf(l.box[2 * a], r.box[2 * a]); nb[2 * a + 1] = fmaxf(l.box[2 * a + 1], r.box[2 * a + 1]); }
            if (memcmp(nb, nd.box, sizeof(nb)) == 0) break;
            memcpy(nd.box, nb, sizeof(nb));
            area[k] = area_of(nb);
            k = parent[k];
        }
    };
    auto replace_child = [&](int32_t p, int32_t old_c, int32_t new_c) {
        if (p < 0) root = new_c;
        else if (s.nodes[p].left == old_c) s.nodes[p].left = new_c;
        else s.nodes[p].right = new_c;
        parent[new_c] = p;
    };
    struct Cand { double induced; int32_t node; bool operator<(const Cand &o) const { return induced > o.induced; } };
    std::vector<Cand> heap;
    // Not every node costs the same to visit: a node with a leaf child is a FRINGE record (box + triangle tests) and the kernel's
    // instruction budget prices a FRINGE visit at 2.5 INNER visits (27 % of the vector instructions for 3.9 visits per ray against
    // 36 % for 13.0).  So the objective is sum(area x visit cost), with cf for nodes that have a leaf child -- it
    // pairs leaves up and turns leaf + subtree nodes into INNER ones where that is cheap: T 3.99 -> 3.95 triangle tests per ray at
    // the same V on cfg 3's scene, another -1.3 % (SRT_BVH_FRINGE_WEIGHT overrides; 1 = plain surface area).
    const double cf = getenv("SRT_BVH_FRINGE_WEIGHT") ? std::max(1.0, atof(getenv("SRT_BVH_FRINGE_WEIGHT"))) : 2.5;
    auto is_leaf = [&](int32_t k) { return s.nodes[k].prim >= 0; };
    // A PAIRED tree (every internal node has two leaf children or none: build_bvh_sah's even splits) stays paired: only internal
    // subtrees move (their triangle count is even) and only next to internal nodes -- a leaf is never taken out of, or put into, a pair.
    const bool paired = tree_is_paired(s);
    auto wgt = [&](int32_t k) { return (!is_leaf(k) && (is_leaf(s.nodes[k].left) || is_leaf(s.nodes[k].right))) ? cf : 1.0; };
    std::vector<int32_t> todo(n);
    for (int pass = 0; pass < passes; pass++) {
        for (int32_t k = 0; k < n; k++) todo[k] = k;
        std::stable_sort(todo.begin(), todo.end(), [&](int32_t a, int32_t b) { return area[a] > area[b]; });
        for (int32_t N : todo) {
            const int32_t P = parent[N];
            if (P < 0 || parent[P] < 0) continue;      // the root and its children stay (the parent node is re-used as the new parent)
            if (paired && is_leaf(N)) continue;
            const int32_t G = parent[P];
            const int32_t S = s.nodes[P].left == N ? s.nodes[P].right : s.nodes[P].left;
            replace_child(G, P, S);                    // take N (and P) out
            refit_up(G);
            const float *nb = s.nodes[N].box;
            const double a_n = area[N];
            double best_cost = DBL_MAX; int32_t best = S;
            heap.clear();
            heap.push_back({0.0, root});
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const Cand c = heap.back(); heap.pop_back();
                const int32_t gx = parent[c.node];
                const double credit_max = cf > 1.0 && gx >= 0 ? (cf - 1.0) * area[gx] : 0.0;
                if (c.induced + a_n - credit_max >= best_cost) break;      // every remaining position costs at least that
                const BvhNode &x = s.nodes[c.node];
                if (paired && x.prim >= 0) continue;      // (not a position: a leaf stays with its partner)
                float u[6];
                for (int a = 0; a < 3; a++) { u[2 * a] = fminf(x.box[2 * a], nb[2 * a]); u[2 * a + 1] = fmaxf(x.box[2 * a + 1], nb[2 * a + 1]); }
                const double ua = area_of(u);
                double direct = ua;
                if (cf > 1.0) {
                    direct = ua * ((is_leaf(c.node) || is_leaf(N)) ? cf : 1.0);
                    // x's parent loses a leaf child when x is a leaf: it turns into an INNER record if its other child is internal
                    if (gx >= 0 && is_leaf(c.node)) {
                        const int32_t other = s.nodes[gx].left == c.node ? s.nodes[gx].right : s.nodes[gx].left;
                        if (!is_leaf(other)) direct -= (cf - 1.0) * area[gx];
                    }
                }
                if (c.induced + direct < best_cost) { best_cost = c.induced + direct; best = c.node; }
                const double below = c.induced + (ua - area[c.node]) * (cf > 1.0 ? wgt(c.node) : 1.0);      // what the ancestors of a position below x pay
                if (x.prim < 0 && below + a_n - (cf > 1.0 ? (cf - 1.0) * area[c.node] : 0.0) < best_cost) {
                    heap.push_back({below, x.left}); std::push_heap(heap.begin(), heap.end());
                    heap.push_back({below, x.right}); std::push_heap(heap.begin(), heap.end());
                }
            }
            const int32_t GX = parent[best];           // put P back as the parent of (best, N)
            replace_child(GX, best, P);
            s.nodes[P].left = best; s.nodes[P].right = N;
            parent[best] = P; parent[N] = P;
            s.nodes[P].box[0] = FLT_MAX;               // (force the refit of P itself)
            refit_up(P);
        }
    }
    renumber(root);
}

// srt_scene_optimise_bvh: the reinsertion passes, boxes and depth again, then the builder's child order (the nearer child of every
// node to the scene's ordering viewpoint on the left).
static void optimise_built_tree(srt_scene &s, int passes) {
    optimise_bvh_by_reinsertion(s, passes);
    finish_boxes_and_depth(s);
    auto dist2 = [&](const float *bx) {
        double d2 = 0;
        for (int ax = 0; ax < 3; ax++) {
            const double p = s.has_order_eye ? s.order_eye[ax] : s.cam.lookfrom[ax], lo = bx[2 * ax], hi = bx[2 * ax + 1];
            const double d = p < lo ? lo - p : (p > hi ? p - hi : 0.0);
            d2 += d * d;
        }
        return d2;
    };
    for (BvhNode &nd : s.nodes)
        if (nd.prim < 0 && dist2(s.nodes[nd.right].box) < dist2(s.nodes[nd.left].box)) std::swap(nd.left, nd.right);
}

// This build's own builder for the large synthetic scenes: binned SAH over all three axes (the reference
// builder never splits on z and sorts by box minimum).  Same node semantics, better tree.
int build_bvh_sah(srt_scene &s) {
    s.nodes.clear(); s.root = -1; s.bvh_valid = false; s.depth = 0;
    const size_t n = s.raw.size();
    if (n == 0) { set_global_error("BVH: empty scene"); return SRT_ERR_BVH; }
    std::vector<int32_t> order(n);
    std::vector<float> cx(n), cy(n), cz(n);
    for (size_t k = 0; k < n; k++) {
        order[k] = (int32_t)k;
        const float *b = s.rec[k].box;
        cx[k] = 0.5f * (b[0] + b[1]); cy[k] = 0.5f * (b[2] + b[3]); cz[k] = 0.5f * (b[4] + b[5]);
    }
    auto centroid = [&](int32_t t, int a) { return a == 0 ? cx[t] : (a == 1 ? cy[t] : cz[t]); };
    struct Box { float lo[3], hi[3];
        void reset() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
        void grow(const float *b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b[2 * a]); hi[a] = std::max(hi[a], b[2 * a + 1]); } }
        void grow(const Box &o) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); } }
        double area() const { double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
                              return (dx < 0 || dy < 0 || dz < 0) ? 0.0 : 2.0 * (dx * dy + dy * dz + dz * dx); } };
    struct Span { size_t start, end; int32_t node; };
    std::vector<Span> stack;
    s.nodes.emplace_back();
    s.root = 0;
    stack.push_back({0, n, 0});
    constexpr int kMaxBins = 256;
    // 128 bins: V (node records per ray, cfg 3) 17.6 with 32 bins, 17.0 with 128, 16.4 with 256; build time is linear in it
    const int kBins = getenv("SRT_SAH_BINS") ? std::min(kMaxBins, std::max(2, atoi(getenv("SRT_SAH_BINS")))) : 256;
    // experiment knob: weight of a child holding N triangles in the split cost (1 = classic SAH area * N; every leaf holds one
    // triangle here, so a subtree's real cost grows faster than N)
    const double kAlpha = getenv("SRT_SAH_ALPHA") ? atof(getenv("SRT_SAH_ALPHA")) : 1.0;
    auto weight = [&](double n_tris) { return kAlpha == 1.0 ? n_tris : (kAlpha < 0 ? n_tris * (1.0 + std::log2(n_tris)) : std::pow(n_tris, kAlpha)); };
    const size_t kSweepMax = getenv("SRT_SAH_SWEEP") ? (size_t)std::max(0, atoi(getenv("SRT_SAH_SWEEP"))) : 8192;   // exact sweep below this span
    // PAIRED trees (round 5).  A node with ONE leaf child costs every FRINGE visit of the render kernel a box test it needs for
    // those nodes alone (a fifth of the visit's instructions; 5-9 % of the FRINGE visits on the benchmark scene go there).  With an
    // even triangle count every span is cut into two EVEN halves, so every internal node has two leaf children or none, and the
    // launcher picks the kernel variant without that box test (render_kernel<.., PAIRED>).  An odd span (odd triangle count) splits
    // freely: such a tree keeps one leaf + subtree node per odd level and the general variant.  SRT_SAH_EVEN=0: round 4's builder.
    const bool kEven = !(getenv("SRT_SAH_EVEN") && atoi(getenv("SRT_SAH_EVEN")) == 0);
    while (!stack.empty()) {
        const Span cur = stack.back();
        stack.pop_back();
        const size_t span = cur.end - cur.start;
        if (span == 1) { s.nodes[cur.node].prim = order[cur.start]; continue; }
        size_t mid = cur.start + span / 2;
        int best_axis = -1;
        if (span > 2) {
            float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (size_t k = cur.start; k < cur.end; k++)
                for (int a = 0; a < 3; a++) { float c = centroid(order[k], a); clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
            double best_cost = DBL_MAX; int best_bin = -1;
            for (int a = 0; a < 3; a++) {
                const float ext = chi[a] - clo[a];
                if (!(ext > 0)) continue;
                Box bb[kMaxBins]; size_t cnt[kMaxBins];
                for (int b = 0; b < kBins; b++) { bb[b].reset(); cnt[b] = 0; }
                const float scale = (float)kBins / ext;
                for (size_t k = cur.start; k < cur.end; k++) {
                    int b = (int)((centroid(order[k], a) - clo[a]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    bb[b].grow(s.rec[order[k]].box); cnt[b]++;
                }
                double right_area[kMaxBins]; size_t right_cnt[kMaxBins];
                Box acc; acc.reset(); size_t c = 0;
                for (int b = kBins - 1; b > 0; b--) { acc.grow(bb[b]); c += cnt[b]; right_area[b] = acc.area(); right_cnt[b] = c; }
                acc.reset(); c = 0;
                for (int b = 0; b < kBins - 1; b++) {
                    acc.grow(bb[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const double cost = acc.area() * (double)c + right_area[b + 1] * (double)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
                }
            }
            // Spans up to kSweepMax triangles: exact sweep instead of bins -- every split position between two consecutive
            // centroids of every axis is priced (sort + prefix / suffix box areas).  cfg 3: V 17.76 (128 bins) / 17.24 (256 bins)
            // -> sweep; the tree is an input of the traversal, so this is exact by construction.
            bool swept = false;
            if (span <= kSweepMax) {
                std::vector<int32_t> tmp(order.begin() + cur.start, order.begin() + cur.end), best_order;
                std::vector<double> right_area(span);
                double sweep_cost = DBL_MAX; size_t sweep_mid = 0;
                for (int a = 0; a < 3; a++) {
                    if (!(chi[a] - clo[a] > 0)) continue;
                    std::stable_sort(tmp.begin(), tmp.end(), [&](int32_t x, int32_t y) { return centroid(x, a) < centroid(y, a); });
                    Box acc; acc.reset();
                    for (size_t k = span - 1; k > 0; k--) { acc.grow(s.rec[tmp[k]].box); right_area[k] = acc.area(); }
                    acc.reset();
                    for (size_t k = 0; k + 1 < span; k++) {
                        acc.grow(s.rec[tmp[k]].box);
                        if (kEven && span % 2 == 0 && (k + 1) % 2 != 0) continue;      // (both halves even)
                        const double cost = acc.area() * weight((double)(k + 1)) + right_area[k + 1] * weight((double)(span - k - 1));
                        if (cost < sweep_cost) { sweep_cost = cost; sweep_mid = k + 1; best_order = tmp; }
                    }
                }
                if (sweep_mid > 0 && sweep_mid < span) {
                    std::copy(best_order.begin(), best_order.end(), order.begin() + cur.start);
                    mid = cur.start + sweep_mid;
                    swept = true;
                    best_axis = 0;
                }
            }
            if (!swept && best_axis >= 0) {
                const float ext = chi[best_axis] - clo[best_axis];
                const float scale = (float)kBins / ext;
                auto it = std::partition(order.begin() + cur.start, order.begin() + cur.end, [&](int32_t t) {
                    int b = (int)((centroid(t, best_axis) - clo[best_axis]) * scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= best_bin;
                });
                mid = (size_t)(it - order.begin());
                if (kEven && span % 2 == 0 && (mid - cur.start) % 2 != 0 && mid > cur.start && mid < cur.end) {
                    // binned split of an even span into two odd halves: move the right half's triangle nearest to the plane across
                    auto nearest = std::min_element(order.begin() + mid, order.begin() + cur.end,
                                                    [&](int32_t a, int32_t b) { return centroid(a, best_axis) < centroid(b, best_axis); });
                    std::iter_swap(order.begin() + mid, nearest);
                    mid++;
                    if (mid == cur.end) mid -= 2;      // (the right half held one triangle: take one from the left instead)
                }
            }
            if (best_axis < 0 || mid == cur.start || mid == cur.end) {
                // coincident centroids: median split on the widest box axis keeps the tree balanced
                mid = cur.start + span / 2;
                if (kEven && span % 2 == 0 && (span / 2) % 2 != 0) mid++;      // (span 4k + 2: halves 2k + 2 and 2k)
                std::nth_element(order.begin() + cur.start, order.begin() + mid, order.begin() + cur.end,
                                 [&](int32_t a, int32_t b) { return cx[a] < cx[b]; });
            }
        }
        // Child order.  The traversal is the reference's: always left first (bvh.cu:154-160), so the order is a property of
        // the tree.  Put the half that is nearer to the scene's camera on the left: camera rays then meet their closest hit
        // early and closest_so_far prunes the far half -- the effect of a front-to-back traversal for 40 % of the rays,
        // without touching the traversal order.
        bool second_first = false;
        {
            Box a, b; a.reset(); b.reset();
            for (size_t k = cur.start; k < mid; k++) a.grow(s.rec[order[k]].box);
            for (size_t k = mid; k < cur.end; k++) b.grow(s.rec[order[k]].box);
            auto dist2 = [&](const Box &bx) {
                double d2 = 0;
                for (int ax = 0; ax < 3; ax++) {
                    const double p = s.has_order_eye ? s.order_eye[ax] : s.cam.lookfrom[ax];
                    const double d = p < bx.lo[ax] ? bx.lo[ax] - p : (p > bx.hi[ax] ? p - bx.hi[ax] : 0.0);
                    d2 += d * d;
                }
                return d2;
            };
            second_first = dist2(b) < dist2(a);
        }
        const int32_t l = (int32_t)s.nodes.size();
        s.nodes.emplace_back(); s.nodes.emplace_back();
        s.nodes[cur.node].left = l; s.nodes[cur.node].right = l + 1;
        if (second_first) {
            stack.push_back({cur.start, mid, l + 1});
            stack.push_back({mid, cur.end, l});
        } else {
            stack.push_back({mid, cur.end, l + 1});
            stack.push_back({cur.start, mid, l});
        }
    }
    finish_boxes_and_depth(s);
    if (const char *ev = getenv("SRT_BVH_OPT_PASSES")) {      // experiment knob: srt_scene_optimise_bvh as part of the build
        s.bvh_valid = true;
        if (atoi(ev) > 0) optimise_built_tree(s, atoi(ev));
    }
    s.bvh_valid = true;
    return SRT_OK;
}

// ------------------------------------------------------------------------------------------------------
// GPU images
// ------------------------------------------------------------------------------------------------------
static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int flatten_scene(const srt_scene &s, FlatScene &out) {
    if (!s.bvh_valid) { set_global_error("upload: BVH not built"); return SRT_ERR_INVALID; }
    const size_t n_tris = s.raw.size(), n_mats = s.mats.size();
    for (size_t k = 0; k < n_tris; k++)
        if (s.raw[k].mat_index >= n_mats) { set_global_error("upload: triangle references a missing material"); return SRT_ERR_INVALID; }
    // the kernel addresses the spectrum tables with 32-bit byte offsets ((n_mats + 1) tables of 768 B, the background last) and
    // forms them with a 24-bit multiply: both bounds, like the 48-byte shade and 96-byte fringe products below
    if ((uint64_t)(n_mats + 1) * 768ull >= (1ull << 32) || n_mats >= (1u << 23)) { set_global_error("upload: too many materials (limit 5 592 403)"); return SRT_ERR_INVALID; }
    if (n_tris >= (1u << 24)) { set_global_error("upload: too many triangles (record offsets are formed with 24-bit multiplies)"); return SRT_ERR_INVALID; }

    // triangles: a = {n, D}, b = {v0[w], v0[h], v1[w], v1[h]}, c = {v2[w], v2[h], flags, 0}
    out.tris.assign(12 * n_tris, 0.f);
    for (size_t k = 0; k < n_tris; k++) {
        const srt_tri_in &t = s.raw[k];
        const TriRecord &r = s.rec[k];
        int w, h;
        plane_axes(r.aa_plane, w, h);
        float *o = &out.tris[12 * k];
        o[0] = r.n[0]; o[1] = r.n[1]; o[2] = r.n[2]; o[3] = r.D;
        o[4] = t.v0[w]; o[5] = t.v0[h]; o[6] = t.v1[w]; o[7] = t.v1[h];
        o[8] = t.v2[w]; o[9] = t.v2[h];
        uint32_t flags = (w == 1 ? 1u : 0u) | (h == 2 ? 2u : 0u) | (r.clockwise ? 4u : 0u) | (t.mat_index << 8);
        o[10] = bits_to_float(flags);
    }

    // Record order: first the INNER records (both children internal) in breadth-first order from the root -- the top of
    // the tree is a prefix that the kernel keeps in LDS -- then the FRINGE records (at least one leaf child) in depth-first
    // order.  `record index >= n_inner` tells the kernel that a visit includes triangle tests.
    std::vector<int32_t> rec_index(s.nodes.size(), -1);
    std::vector<int32_t> pre;
    auto is_internal = [&](int32_t k) { return s.nodes[k].prim < 0; };
    auto is_inner = [&](int32_t k) { return is_internal(k) && is_internal(s.nodes[k].left) && is_internal(s.nodes[k].right); };
    {
        std::vector<int32_t> queue;
        if (is_internal(s.root)) queue.push_back(s.root);
        for (size_t h = 0; h < queue.size(); h++) {
            const int32_t k = queue[h];
            if (is_inner(k)) { rec_index[k] = (int32_t)pre.size(); pre.push_back(k); }
            if (is_internal(s.nodes[k].left)) queue.push_back(s.nodes[k].left);
            if (is_internal(s.nodes[k].right)) queue.push_back(s.nodes[k].right);
        }
    }
    out.n_inner = (int)pre.size();
    {
        std::vector<int32_t> st;
        st.push_back(s.root);
        while (!st.empty()) {
            int32_t k = st.back(); st.pop_back();
            if (!is_internal(k)) continue;
            if (!is_inner(k)) { rec_index[k] = (int32_t)pre.size(); pre.push_back(k); }
            st.push_back(s.nodes[k].right);
            st.push_back(s.nodes[k].left);
        }
    }
    auto child_ref = [&](int32_t k) -> int32_t { return s.nodes[k].prim >= 0 ? ~s.nodes[k].prim : rec_index[k]; };
    // INNER records, 64 B: three axis planes (lo_L, lo_R, hi_L, hi_R), then lref, rref (see NodeSrc in srt_device.h).
    // FRINGE records, 96 B = 12 (left, right) pairs: pair k holds word k of the left child's block and word k of the right
    // child's block side by side -- a 16-byte load delivers two register pairs that feed the packed arithmetic directly.
    // A child's 12-word block is its box (xmin xmax ymin ymax zmin zmax 0 ...) when it is internal, a copy of its triangle
    // record (n.x n.y n.z D | v0w v0h v1w v1h | v2w v2h flags 0) when it is a leaf, so that a fringe visit is ONE round of
    // independent loads instead of record -> triangle.  Word 10 = flags (bit 31 added here: set for a counter-clockwise
    // triangle = the sign flip that turns its `area <= 0` tests into `>= 0`), word 11 = the child reference.
    const size_t n_in = (size_t)out.n_inner, n_fr = pre.size() - n_in;
    out.nodes.assign(16 * std::max<size_t>(n_in, 1), 0.f);
    out.fringe.assign(24 * std::max<size_t>(n_fr, 1), 0.f);
    for (size_t r = 0; r < pre.size(); r++) {
        const BvhNode &nd = s.nodes[pre[r]];
        const BvhNode &l = s.nodes[nd.left], &rr = s.nodes[nd.right];
        if (r < n_in) {
            float *o = &out.nodes[16 * r];
            for (int a = 0; a < 3; a++) {
                o[4 * a + 0] = l.box[2 * a]; o[4 * a + 1] = rr.box[2 * a];
                o[4 * a + 2] = l.box[2 * a + 1]; o[4 * a + 3] = rr.box[2 * a + 1];
            }
            o[12] = bits_to_float((uint32_t)child_ref(nd.left));
            o[13] = bits_to_float((uint32_t)child_ref(nd.right));
        } else {
            float *o = &out.fringe[24 * (r - n_in)];
            const BvhNode *ch[2] = {&l, &rr};
            const int32_t refs[2] = {child_ref(nd.left), child_ref(nd.right)};
            for (int k = 0; k < 2; k++) {
                float b[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (ch[k]->prim >= 0) {
                    memcpy(b, &out.tris[12 * (size_t)ch[k]->prim], 11 * sizeof(float));
                    uint32_t flags; memcpy(&flags, &b[10], 4);
                    if (!(flags & 4u)) flags |= 0x80000000u;
                    b[10] = bits_to_float(flags);
                } else memcpy(b, ch[k]->box, 6 * sizeof(float));
                b[11] = bits_to_float((uint32_t)refs[k]);
                for (int w = 0; w < 12; w++) o[2 * w + k] = b[w];
            }
        }
    }
    // The same INNER records once more, 80 B each, for the visit that reads them from MEMORY (trees that do not fit LDS, production
    // build: inner_burst4_mixed_asm): per axis (lo_L, lo_R, hi_L, hi_R, lo_L, lo_R) -- a 16-byte load at +0 delivers (near pair, far
    // pair) for a ray that travels in the positive direction, the same load at +8 for a negative one, exactly the register image the
    // LDS-served lanes of the same visit get from their two 8-byte reads: the selection of aabb.cu:21-25 by ADDRESS here too, no
    // v_cndmask.  Then lref, rref.
    out.nodes_sw.assign(20 * std::max<size_t>(n_in, 1), 0.f);
    for (size_t r = 0; r < n_in; r++) {
        const float *o = &out.nodes[16 * r];
        float *w = &out.nodes_sw[20 * r];
        for (int a = 0; a < 3; a++) { for (int k = 0; k < 4; k++) w[6 * a + k] = o[4 * a + k]; w[6 * a + 4] = o[4 * a + 0]; w[6 * a + 5] = o[4 * a + 1]; }
        w[18] = o[12]; w[19] = o[13];
    }
    out.root_ref = child_ref(s.root);
    out.n_records = (int)pre.size();
    // Entries the traversal stack can hold at once: a push happens only at an INNER record (both children internal, bvh.cu:154-160:
    // "push right iff both"), one per INNER node on the way down, so the bound is the largest number of INNER nodes on a root-to-leaf
    // path -- at least one less than the tree's depth (the last internal level always has a leaf child).  Round 4 sized the LDS stacks
    // by the depth; one slot per wave is 2 KB of LDS = 39 more cached records, which is what decides whether a 4 802-triangle paired
    // tree (2 400 INNER records) of depth 17 is LDS resident.
    {
        std::vector<int> inner_path(s.nodes.size(), 0);
        for (size_t k = s.nodes.size(); k-- > 0;) {      // (children have larger indices than their parent)
            const BvhNode &nd = s.nodes[k];
            if (nd.prim >= 0) continue;
            inner_path[k] = (is_inner((int32_t)k) ? 1 : 0) + std::max(inner_path[nd.left], inner_path[nd.right]);
        }
        out.stack_depth = std::max(1, s.root >= 0 ? inner_path[s.root] : 1);
    }

    // spectra as (s[k], s[k+1]) pairs; material scalars
    auto pairs = [](const float *sd, float *dst) {
        for (int k = 0; k < 96; k++) { dst[2 * k] = 0.f; dst[2 * k + 1] = 0.f; }
        for (int k = 0; k < SRT_N_CIE_SAMPLES - 1; k++) { dst[2 * k] = sd[k]; dst[2 * k + 1] = sd[k + 1]; }
    };
    // (table n_mats = the background spectrum: a miss multiplies it into the path exactly like a hit multiplies its material's)
    out.mat_sd.assign(192 * (n_mats + 1), 0.f);
    out.mat_par.assign(8 * std::max<size_t>(n_mats, 1), 0.f);
    for (size_t m = 0; m < n_mats; m++) {
        const srt_material &mt = s.mats[m];
        pairs(mt.spectral_distribution, &out.mat_sd[192 * m]);
        float *o = &out.mat_par[8 * m];
        o[0] = bits_to_float(mt.material_type); o[1] = mt.reflection_fuzz;
        o[2] = mt.sellmeier_B[0]; o[3] = mt.sellmeier_B[1]; o[4] = mt.sellmeier_B[2];
        o[5] = mt.sellmeier_C[0]; o[6] = mt.sellmeier_C[1]; o[7] = mt.sellmeier_C[2];
    }
    pairs(s.background, &out.mat_sd[192 * n_mats]);
    // shading records, 48 B per triangle: { n.x n.y n.z bits(mat) } { bits(type) fuzz B0 B1 } { B2 C0 C1 C2 } -- everything
    // material::scatter needs about the hit in ONE round of loads (instead of triangle -> material index -> material)
    out.shade.assign(12 * std::max<size_t>(n_tris, 1), 0.f);
    for (size_t k = 0; k < n_tris; k++) {
        const srt_material &mt = s.mats[s.raw[k].mat_index];
        float *o = &out.shade[12 * k];
        o[0] = s.rec[k].n[0]; o[1] = s.rec[k].n[1]; o[2] = s.rec[k].n[2]; o[3] = bits_to_float(s.raw[k].mat_index);
        o[4] = bits_to_float(mt.material_type); o[5] = mt.reflection_fuzz;
        o[6] = mt.sellmeier_B[0]; o[7] = mt.sellmeier_B[1]; o[8] = mt.sellmeier_B[2];
        o[9] = mt.sellmeier_C[0]; o[10] = mt.sellmeier_C[1]; o[11] = mt.sellmeier_C[2];
    }
    return SRT_OK;
}

static float g_cie[4][SRT_N_CIE_SAMPLES];
static std::once_flag g_cie_once;
static void cie_init() {
    std::call_once(g_cie_once, [] {
        int k = 0;
#define ROW(X, Y, Z, D) g_cie[0][k] = (float)(X); g_cie[1][k] = (float)(Y); g_cie[2][k] = (float)(Z); \
                        g_cie[3][k] = (float)((D) / SRT_D65_NORM_DIVISOR); k++;
        SRT_CIE_ROW_LIST(ROW)
#undef ROW
    });
}
}  // namespace srt
// The constant tables this build computes with, for the pinning test (tests/test_ref_tables.py): rows {x_bar, y_bar, z_bar,
// normalised D65} of utils/cie_const.cu:12-122 and the d65_XYZ_to_sRGB matrix of utils/color_const.cu:17-19.
extern "C" int srt_color_tables(float cmf[SRT_N_CIE_SAMPLES * 4], float xyz_to_srgb[9]) {
    if (!cmf || !xyz_to_srgb) { srt::set_global_error("srt_color_tables: null argument"); return SRT_ERR_INVALID; }
    float rows[96 * 4];
    srt::cmf_rows(rows);
    memcpy(cmf, rows, sizeof(float) * SRT_N_CIE_SAMPLES * 4);
    const float m[9] = {SRT_XYZ2RGB_00, SRT_XYZ2RGB_01, SRT_XYZ2RGB_02, SRT_XYZ2RGB_10, SRT_XYZ2RGB_11, SRT_XYZ2RGB_12,
                        SRT_XYZ2RGB_20, SRT_XYZ2RGB_21, SRT_XYZ2RGB_22};
    memcpy(xyz_to_srgb, m, sizeof(m));
    return SRT_OK;
}
namespace srt {
void cmf_rows(float *rows) {
    cie_init();
    memset(rows, 0, 96 * 4 * sizeof(float));
    for (int k = 0; k < SRT_N_CIE_SAMPLES; k++)
        for (int c = 0; c < 4; c++) rows[4 * k + c] = g_cie[c][k];
}

// ------------------------------------------------------------------------------------------------------
// spectra baking (color/color_to_spectrum.cuh)
// ------------------------------------------------------------------------------------------------------
static float host_interp(const float *sp, float lambda) {   // spectrum/spectrum.cu:11-22
    lambda -= 360.0f;
    lambda *= ((float)SRT_N_CIE_SAMPLES - 1) / (830.0f - 360.0f);
    int offset = (int)lambda;
    if (offset < 0) offset = 0;
    if (offset > SRT_N_CIE_SAMPLES - 2) offset = SRT_N_CIE_SAMPLES - 2;
    float weight = lambda - (float)offset;
    return (1.0f - weight) * sp[offset] + weight * sp[offset + 1];
}
static float sigmoid(float x) {   // color_to_spectrum.cuh:36-40
    if (isinf(x)) return x > 0 ? 1.f : 0.f;
    return 0.5f * x / sqrtf(1.0f + x * x) + 0.5f;
}
static void bake_sigmoid(const float c[3], float scale, bool times_d65, float *out) {   // :173-186, :204-219
    cie_init();
    const float step = (830.0f - 360.0f) / SRT_N_CIE_SAMPLES;   // 470/95, read back at 470/94 (Q4)
    float lambda = 360.0f;
    for (int i = 0; i < SRT_N_CIE_SAMPLES; i++) {
        const float x = lambda * lambda * c[2] + lambda * c[1] + c[0];   // polynomial(), :153-156 (z is the quadratic term, Q2)
        const float sg = sigmoid(x);
        out[i] = times_d65 ? scale * sg * host_interp(g_cie[3], lambda) : sg;
        lambda += step;
    }
}
// Reference quirks that live in SCENE CONSTRUCTION (inputs of the path, not the path): Q1 the Sellmeier constructor copies B into C
// (materials/material.cuh:66-67: NaN / sub-unity indices over most of the spectrum), Q2 grey colours return their sigmoid
// coefficient in the slot the evaluator reads as the QUADRATIC term (color_to_spectrum.cuh:118-120 vs :153-156: albedo 0.73 bakes
// to 1.0, < 0.5 to 0).  Default on -- every parity statement is about the reference as written; srt_set_reference_quirks(0)
// (srt_render --physically-correct) builds and bakes what the author evidently meant.  Not parity-checked.
static int g_reference_quirks = 1;
static bool grey_coeffs(const float rgb[3], float c[3]) {   // :118-120
    if (!(rgb[0] == rgb[1] && rgb[1] == rgb[2])) return false;
    const float r = rgb[0];
    const float k = (r - .5f) / sqrtf(r * (1 - r));
    c[0] = 0.f; c[1] = 0.f; c[2] = 0.f;
    c[g_reference_quirks ? 2 : 0] = k;      // quirk Q2: the constant lands in the lambda^2 slot
    return true;
}

// ------------------------------------------------------------------------------------------------------
// scene construction helpers (primitives/tri_quad.cuh, tri_box.cuh, prism.cuh, pyramid.cuh, transform.cu)
// A "shape" is a list of indices into the scene's triangle list; every triangle carries its sticky
// aa_plane state exactly as tri::init leaves it.
// ------------------------------------------------------------------------------------------------------
// transform::assign_rot_matrix (primitives/transform.cu:4-34; axis = transform::AXIS: 1 X, 2 Y, 3 Z).  Exported as
// srt_rotation_matrix and held bit for bit against the reference's own function (tests/test_ref_host.py).
// `cos(theta)` with a float argument: the reference's translation unit, compiled here from its source, binds the call to the
// double function and narrows the result (cvtss2sd / call cos / cvtsd2ss in its host object), which differs from cosf by an ulp for
// about one angle in a hundred -- none of the three angles the scenes use (+25, -18, +10 degrees give the same bits either way).
// The double form is what the compiled reference computes, so it is what this build computes; what nvcc's device overload
// resolution makes of the same line is unknowable here.
static void assign_rot_matrix(float theta, int axis, float m[9]) {
    const float c = (float)cos((double)theta), sn = (float)sin((double)theta);
    if (axis == 1) { m[4] = c; m[5] = -sn; m[7] = sn; m[8] = c; }
    else if (axis == 2) { m[0] = c; m[2] = sn; m[6] = -sn; m[8] = c; }
    else if (axis == 3) { m[0] = c; m[1] = -sn; m[3] = sn; m[4] = c; }
}

struct Builder {
    srt_scene &s;
    explicit Builder(srt_scene &sc) : s(sc) {}

    void init_tri(int k) {                     // tri::init: only aa_plane is state we need to carry
        TriRecord r;
        tri_precompute(s.raw[k], r);
        s.raw[k].aa_plane = r.aa_plane;
    }
    // tri(v1, v2, v3, mat, defer=false, VECTORS) (tri.cuh:28-47): v[1] = v1+v2, v[2] = v1+v3
    int tri_vectors(F3 q, F3 u, F3 v, uint32_t mat) {
        srt_tri_in t;
        put(t.v0, q); put(t.v1, add(q, u)); put(t.v2, add(q, v));
        t.mat_index = mat; t.aa_plane = SRT_AAP_NONE;
        s.raw.push_back(t);
        init_tri((int)s.raw.size() - 1);
        return (int)s.raw.size() - 1;
    }
    int tri_vertices(F3 a, F3 b, F3 c, uint32_t mat) {
        srt_tri_in t;
        put(t.v0, a); put(t.v1, b); put(t.v2, c);
        t.mat_index = mat; t.aa_plane = SRT_AAP_NONE;
        s.raw.push_back(t);
        init_tri((int)s.raw.size() - 1);
        return (int)s.raw.size() - 1;
    }
    // Mesh triangle of the SYNTHETIC scenes.  The reference's inside test projects a triangle on the plane named by its
    // aa_plane member, which tri::init only sets for axis-aligned triangles and otherwise leaves as it was (heap garbage
    // for a fresh triangle, tri.cuh:90; this build: an explicit input, D2).  With the XY default a facet whose normal lies
    // near the XY plane projects to a sliver and is almost un-hittable: rays leak into a tessellated sphere and bounce
    // inside until the bounce limit.  A mesh therefore passes, per triangle, the plane of its dominant normal axis --
    // a value the member can legally hold -- which keeps tessellated objects watertight.
    int tri_mesh(F3 a, F3 b, F3 c, uint32_t mat) {
        const F3 n = cross3(sub(b, a), sub(c, a));
        const float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
        srt_tri_in t;
        put(t.v0, a); put(t.v1, b); put(t.v2, c);
        t.mat_index = mat;
        t.aa_plane = (ax >= ay && ax >= az) ? SRT_AAP_YZ : ((ay >= az) ? SRT_AAP_XZ : SRT_AAP_XY);
        s.raw.push_back(t);
        init_tri((int)s.raw.size() - 1);
        return (int)s.raw.size() - 1;
    }
    // tri_quad(Q, u, v) (tri_quad.cuh:14-20): halves = tri(Q,u,v) and tri(Q+u+v, -u, -v)
    int quad(F3 q, F3 u, F3 v, uint32_t mat) {
        int first = tri_vectors(q, u, v, mat);
        tri_vectors(add(add(q, u), v), neg(u), neg(v), mat);
        return first;   // the two halves are consecutive
    }
    F3 V(int tri, int k) const { const srt_tri_in &t = s.raw[tri]; return k == 0 ? f3(t.v0) : (k == 1 ? f3(t.v1) : f3(t.v2)); }
    void setV(int tri, int k, F3 v) { srt_tri_in &t = s.raw[tri]; put(k == 0 ? t.v0 : (k == 1 ? t.v1 : t.v2), v); }
    F3 quad_u(int q) const { return sub(V(q, 1), V(q, 0)); }       // tri_quad.cuh:29-31
    F3 quad_v(int q) const { return sub(V(q, 2), V(q, 0)); }       // :34-36
    F3 quad_Q(int q) const { return V(q, 0); }                     // :39-41
    F3 quad_center(int q) const { return add(div(add(quad_u(q), quad_v(q)), 2.0f), quad_Q(q)); }   // :44-46

    void translate(const std::vector<int> &tris, F3 d) {           // tri::translate(dir, false), tri.cu:86-93
        for (int k : tris) for (int c = 0; c < 3; c++) setV(k, c, add(V(k, c), d));
    }
    // tri::rotate(theta, ax, false, false) (tri.cu:96-118) with transform::assign_rot_matrix (transform.cu:4-34)
    void rotate_about_origin(const std::vector<int> &tris, float theta, int axis /*1=X 2=Y 3=Z*/) {
        float m[9] = {1.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 1.f};
        assign_rot_matrix(theta, axis, m);
        for (int k : tris)
            for (int cidx = 0; cidx < 3; cidx++) {
                F3 v = V(k, cidx);   // vec3::matrix_mul, vec3.cuh:80-91
                setV(k, cidx, f3((m[0] * v.x) + (m[1] * v.y) + (m[2] * v.z), (m[3] * v.x) + (m[4] * v.y) + (m[5] * v.z),
                                 (m[6] * v.x) + (m[7] * v.y) + (m[8] * v.z)));
            }
    }
    void reinit(const std::vector<int> &order) { for (int k : order) init_tri(k); }
};

static inline float deg2rad(float d) { return d * 3.1415926535897932385f / 180.0f; }   // cuda_utility.cuh:40-43, utility.h:10

static std::vector<int> range_ids(int first, int count) { std::vector<int> v(count); for (int k = 0; k < count; k++) v[k] = first + k; return v; }

// tri_box(a, b, mats[6]) (tri_box.cuh:11-45): front, right, back, left, top, bottom
static int add_box(Builder &B, F3 a, F3 b, const uint32_t mats[6]) {
    F3 mn = f3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z));
    F3 mx = f3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z));
    F3 dx = f3(mx.x - mn.x, 0.f, 0.f), dy = f3(0, mx.y - mn.y, 0.f), dz = f3(0, 0, mx.z - mn.z);
    int first = B.quad(f3(mn.x, mn.y, mx.z), dx, dy, mats[0]);
    B.quad(f3(mx.x, mn.y, mx.z), neg(dz), dy, mats[1]);
    B.quad(f3(mx.x, mn.y, mn.z), neg(dx), dy, mats[2]);
    B.quad(f3(mn.x, mn.y, mn.z), dz, dy, mats[3]);
    B.quad(f3(mn.x, mx.y, mx.z), dx, neg(dz), mats[4]);
    B.quad(f3(mn.x, mn.y, mn.z), dx, dz, mats[5]);
    return first;   // 12 consecutive triangles
}
// tri_box::rotate(theta, Y, reinit=false, local=true) then translate(dir, reinit=true)
// (tri_box.cu:4-34, tri_box.cuh:119-141: center from sides[5] (bottom) and sides[3] (left))
static void box_rotate_translate(Builder &B, int first, float theta, int axis, F3 shift) {
    const int bottom = first + 10, left = first + 6;
    F3 mn = B.quad_Q(bottom);
    F3 mx = add(add(add(mn, B.quad_u(bottom)), B.quad_v(left)), B.quad_v(bottom));   // min + width + height + depth
    F3 center = add(div(sub(mx, mn), 2.0f), mn);
    std::vector<int> ids = range_ids(first, 12);
    B.translate(ids, neg(center));
    B.rotate_about_origin(ids, theta, axis);
    B.translate(ids, center);
    B.translate(ids, shift);
    B.reinit(ids);      // translate(dir, reinit=true): sides[0..5], halves[0] then halves[1]
}

// ------------------------------------------------------------------------------------------------------
// built-in scenes
// ------------------------------------------------------------------------------------------------------
static srt_material make_material(uint32_t type, float r, float g, float b, float fuzz, float power, const float *B3 = nullptr) {
    srt_material m;
    memset(&m, 0, sizeof(m));
    m.col[0] = r; m.col[1] = g; m.col[2] = b;
    m.reflection_fuzz = fuzz; m.material_type = type; m.emission_power = power;
    if (B3) {
        static const float kFlintC[3] = {0.00997743871f, 0.0470450767f, 111.886764f};      // refraction/sellmeier.cuh:15
        static const float kBK7C[3] = {6.00069867e-3f, 2.00179144e-2f, 1.03560653e2f};     // refraction/sellmeier.cuh:7
        const float *C3 = B3[0] > 1.2f ? kFlintC : kBK7C;
        for (int i = 0; i < 3; i++) { m.sellmeier_B[i] = B3[i]; m.sellmeier_C[i] = g_reference_quirks ? B3[i] : C3[i]; }   // C := B, material.cuh:66-67 (Q1)
    } else {
        m.sellmeier_B[0] = 1.0f;   // material(col, fuzz, ir = 1, power, type), material.cuh:49-61
    }
    return m;
}
static const float kFlintB[3] = {1.34533359f, 0.209073176f, 0.937357162f};   // refraction/sellmeier.cuh:14
static const float kBK7B[3] = {1.03961212f, 0.231792344f, 1.01046945f};      // refraction/sellmeier.cuh:6

// sRGB -> sigmoid-polynomial spectrum for NON-grey colours.  The reference reads these coefficients from the pbrt-v4
// rgb2spec table (utils/srgb_to_spectrum.cu: Jakob & Hanika 2019, "A Low-Dimensional Function Space for Efficient
// Spectral Upsampling"), which is absent from the reference mount.  This is the build's own fit of the same model
// (SURVEY 8(f) row f4): find (c0, c1, c2) such that s(l) = sigmoid(c2 x^2 + c1 x + c0), x = (l-360)/470, seen under D65
// through the CIE 1931 observer and the XYZ->sRGB matrix reproduces the colour's linear sRGB; Gauss-Newton with a
// numerical Jacobian on the 95-sample tables.  "Parity unpinned" against the author's table (which also differs from
// a fit through quirk Q3), the baked 95-sample spectrum is a scene input, so checker and GPU consume identical data.
static void model_rgb(const double c[3], double out[3]) {
    cie_init();
    double X = 0, Y = 0, Z = 0, Yw = 0;
    for (int k = 0; k < SRT_N_CIE_SAMPLES; k++) {
        const double x = (double)k / (SRT_N_CIE_SAMPLES - 1);
        const double p = (c[2] * x + c[1]) * x + c[0];
        const double sg = 0.5 * p / sqrt(1.0 + p * p) + 0.5;
        const double w = g_cie[3][k];
        X += g_cie[0][k] * w * sg; Y += g_cie[1][k] * w * sg; Z += g_cie[2][k] * w * sg; Yw += g_cie[1][k] * w;
    }
    X /= Yw; Y /= Yw; Z /= Yw;
    out[0] = 3.2404542 * X - 1.5371385 * Y - 0.4985314 * Z;
    out[1] = -0.9692660 * X + 1.8760108 * Y + 0.0415560 * Z;
    out[2] = 0.0556434 * X - 0.2040259 * Y + 1.0572252 * Z;
}
static double srgb_to_linear(double v) { return v < 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4); }   // color.cu:8-13

// returns coefficients for the reference's evaluator: polynomial(lambda, c[2], c[1], c[0]) in RAW wavelength (nm)
static void fit_sigmoid_coeffs(const float rgb[3], float out[3]) {
    const double target[3] = {srgb_to_linear(rgb[0]), srgb_to_linear(rgb[1]), srgb_to_linear(rgb[2])};
    double c[3] = {0, 0, 0};
    for (int it = 0; it < 60; it++) {
        double r0[3];
        model_rgb(c, r0);
        double res[3] = {r0[0] - target[0], r0[1] - target[1], r0[2] - target[2]};
        if (fabs(res[0]) + fabs(res[1]) + fabs(res[2]) < 1e-9) break;
        double J[3][3];
        for (int j = 0; j < 3; j++) {
            double cp[3] = {c[0], c[1], c[2]}, rp[3];
            cp[j] += 1e-5;
            model_rgb(cp, rp);
            for (int i = 0; i < 3; i++) J[i][j] = (rp[i] - r0[i]) / 1e-5;
        }
        // solve J * d = -res (3x3, Cramer) with a little damping for the near-saturated colours
        for (int i = 0; i < 3; i++) J[i][i] += 1e-9;
        const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) +
                           J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
        if (fabs(det) < 1e-30) break;
        double d[3];
        for (int j = 0; j < 3; j++) {
            double M[3][3];
            for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) M[a][b] = (b == j) ? -res[a] : J[a][b];
            d[j] = (M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                    M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0])) / det;
        }
        double step = 1.0;
        const double norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        if (norm > 20.0) step = 20.0 / norm;          // trust region
        for (int j = 0; j < 3; j++) c[j] += step * d[j];
    }
    // x = (l - 360)/470  ->  raw-wavelength polynomial
    const double a = 1.0 / 470.0, b = -360.0 / 470.0;
    out[2] = (float)(c[2] * a * a);
    out[1] = (float)(2.0 * c[2] * a * b + c[1] * a);
    out[0] = (float)(c[2] * b * b + c[1] * b + c[0]);
}

static void bake_or_fit(srt_material &m) {
    if (srt_material_bake(&m) == SRT_OK) return;      // grey / white / light / glass: the reference's closed forms
    float coeffs[3];
    fit_sigmoid_coeffs(m.col, coeffs);
    if (m.material_type == SRT_MAT_EMISSIVE) bake_sigmoid(coeffs, srt_powf(m.emission_power, 2.0f), true, m.spectral_distribution);
    else bake_sigmoid(coeffs, 1.0f, false, m.spectral_distribution);
}

static void cornell_walls_and_light(Builder &B, const uint32_t wall_mats[5], uint32_t light_mat) {
    // scene/scene.cu:91-104 (identical in all three scenes up to material ids)
    B.quad(f3(0, 0, 0), f3(0, 0, 555), f3(555, 0, 0), wall_mats[0]);            // bottom
    B.quad(f3(555, 555, 555), f3(-555, 0, 0), f3(0, 0, -555), wall_mats[1]);    // top    (d_list[2])
    B.quad(f3(0, 0, 555.f), f3(0, 555, 0), f3(555, 0, 0), wall_mats[2]);        // back   (d_list[4])
    B.quad(f3(555, 0, 0), f3(0, 0, 555), f3(0, 555, 0), wall_mats[3]);          // left   (d_list[6])
    B.quad(f3(0, 0, 0), f3(0, 555, 0), f3(0, 0, 555), wall_mats[4]);            // right  (d_list[8])
    const F3 center = f3(555.f / 2.f, 554.f, 555.f / 2.f);
    const float width = 100.f, depth = 100.f;
    const F3 Q = f3((center.x + width / 2.f), center.y, (center.z + depth / 2.f));
    B.quad(Q, f3(-width, 0, 0), f3(0, 0, -depth), light_mat);                    // light  (d_list[10])
}

// pyramid(Q, u, v, w, mat) (pyramid.cuh:28-46) + rotate(theta, Y, false) + translate(dir) (pyramid.cu:3-35)
static void add_pyramid(Builder &B, F3 Q, F3 u, F3 v, F3 w, uint32_t mat, float theta, F3 shift) {
    const int base = B.quad(Q, u, v, mat);
    const F3 top = add(B.quad_center(base), w);
    const F3 v1 = add(Q, u), v2 = add(Q, v), v3 = add(v2, u);
    const int s0 = B.tri_vertices(Q, top, v2, mat);
    B.tri_vertices(v1, top, Q, mat);
    B.tri_vertices(v2, top, v3, mat);
    B.tri_vertices(v3, top, v1, mat);
    std::vector<int> ids = range_ids(base, 6);
    const F3 center = B.quad_center(base);
    B.translate(ids, neg(center));
    B.rotate_about_origin(ids, theta, 2);
    B.translate(ids, center);
    B.translate(ids, shift);
    // pyramid::translate(dir, true): sides[0..3] first, then the base quad (pyramid.cu:4-12)
    B.reinit({s0, s0 + 1, s0 + 2, s0 + 3, base, base + 1});
}

static void scene_cornell(srt_scene &s, bool different_mats) {
    // scene/scene.cu:74-130 (CORNELL) and :176-226 (TRIS)
    s.mats.clear();
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, .65f, .05f, .05f, 1.f, 0.f));   // red
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, .12f, .45f, .15f, 1.f, 0.f));   // green
    s.mats.push_back(make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kFlintB));
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, .73f, .73f, .73f, 1.f, 0.f));   // white
    s.mats.push_back(make_material(SRT_MAT_EMISSIVE, 1.f, 1.f, 1.f, 1.f, 5.f));        // light
    s.mats.push_back(make_material(SRT_MAT_METALLIC, .5f, .5f, .5f, 0.3f, 0.f));       // metal
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, .12f, .15f, .45f, 1.f, 0.f));   // blue
    if (different_mats) {
        s.mats.push_back(make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kBK7B));
        s.mats.push_back(make_material(SRT_MAT_METALLIC, .7f, .7f, .7f, 0.8f, 0.f));
    }
    for (auto &m : s.mats) bake_or_fit(m);
    Builder B(s);
    // wall order in d_list: bottom, top, back, left, right (scene.cu:83-95)
    const uint32_t walls_c[5] = {3, 3, 3, 1, 6}, walls_t[5] = {6, 2, 1, 8, 5};
    cornell_walls_and_light(B, different_mats ? walls_t : walls_c, 4);
    const uint32_t m1c[6] = {5, 5, 5, 5, 5, 5}, m1t[6] = {3, 8, 0, 1, 2, 3};
    const uint32_t m2c[6] = {0, 0, 0, 0, 0, 0}, m2t[6] = {7, 6, 8, 7, 1, 2};
    int b1 = add_box(B, f3(0.f, 0.f, 0.f), f3(165.f, 330.f, 165.f), different_mats ? m1t : m1c);
    box_rotate_translate(B, b1, deg2rad(25.f), 2, f3(265.f, 0.f, 295.f));
    int b2 = add_box(B, f3(0.f, 0.f, 0.f), f3(165.f, 165.f, 165.f), different_mats ? m2t : m2c);
    box_rotate_translate(B, b2, deg2rad(-18.f), 2, f3(130.f, 0.f, 65.f));
    add_pyramid(B, f3(165.f, 166.f, 0.f), f3(-165.f, 0.f, 0.f), f3(0.f, 0.f, 165.f), f3(0.f, 165.f, 0.f), 2, deg2rad(-18.f),
                f3(130.f, 0.f, 65.f));
}

static void scene_prism(srt_scene &s) {
    // scene/scene.cu:133-173
    s.mats.clear();
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, (float).73, (float).73, (float).73, 1.f, 0.f));
    s.mats.push_back(make_material(SRT_MAT_EMISSIVE, 1.f, 1.f, 1.f, 1.f, 5.f));
    s.mats.push_back(make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kFlintB));
    for (auto &m : s.mats) bake_or_fit(m);
    Builder B(s);
    const uint32_t walls[5] = {0, 0, 0, 0, 0};
    cornell_walls_and_light(B, walls, 1);
    const F3 center = f3(555.f / 2.f, 554.f, 555.f / 2.f);
    const float width = 100.f, prism_width = 165.f, prism_height = 200.f;
    // prism(Q, u, v, w, mat) (prism.cuh:30-40)
    const F3 Q = f3(center.x - width / 2.f, center.y - 1.f, center.z - prism_height / 2.f);
    const F3 u = f3(0.f, -prism_width, 0.f);
    const F3 v = f3((prism_width * sqrtf(3.f)) / 2.f, -prism_width / 2.f, 0.f);
    const F3 w = f3(0.f, 0.f, 200.f);
    const int t0 = B.tri_vectors(Q, v, u, 2);            // bottom: u and v swapped for an outward normal
    B.tri_vectors(add(Q, w), u, v, 2);                   // top
    B.quad(Q, u, w, 2);
    B.quad(Q, w, v, 2);
    B.quad(add(Q, u), sub(v, u), w, 2);
    // prism::rotate(theta, Y, reinit=true, local=true) (prism.cu:13-35): centroid of the six base vertices
    std::vector<int> ids = range_ids(t0, 8);
    F3 sum = add(add(add(add(add(B.V(t0, 0), B.V(t0, 1)), B.V(t0, 2)), B.V(t0 + 1, 0)), B.V(t0 + 1, 1)), B.V(t0 + 1, 2));
    const F3 c = div(sum, 6.f);
    B.translate(ids, neg(c));
    B.rotate_about_origin(ids, deg2rad(10.f), 2);
    B.translate(ids, c);
    B.reinit(ids);       // prism::init: base[0], base[1], sides[0..2] (prism.cuh:42-49) = creation order
}

// --- synthetic benchmark scenes (this build's own) ----------------------------------------
struct SplitMix {   // scene-layout PRNG (not on the render path)
    uint64_t x;
    explicit SplitMix(uint64_t s) : x(s) {}
    uint64_t next() { uint64_t z = (x += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }   // [0,1)
};

static void add_octahedron(Builder &B, F3 c, float r, uint32_t mat) {
    const F3 px = add(c, f3(r, 0, 0)), nx = add(c, f3(-r, 0, 0)), py = add(c, f3(0, r, 0)), ny = add(c, f3(0, -r, 0)),
             pz = add(c, f3(0, 0, r)), nz = add(c, f3(0, 0, -r));
    B.tri_mesh(px, py, pz, mat); B.tri_mesh(py, nx, pz, mat); B.tri_mesh(nx, ny, pz, mat); B.tri_mesh(ny, px, pz, mat);
    B.tri_mesh(py, px, nz, mat); B.tri_mesh(nx, py, nz, mat); B.tri_mesh(ny, nx, nz, mat); B.tri_mesh(px, ny, nz, mat);
}

static void add_icosphere(Builder &B, F3 c, float r, int subdiv, uint32_t mat) {
    const float t = (1.0f + sqrtf(5.0f)) / 2.0f;
    std::vector<F3> v = {f3(-1, t, 0), f3(1, t, 0), f3(-1, -t, 0), f3(1, -t, 0), f3(0, -1, t), f3(0, 1, t),
                         f3(0, -1, -t), f3(0, 1, -t), f3(t, 0, -1), f3(t, 0, 1), f3(-t, 0, -1), f3(-t, 0, 1)};
    for (auto &p : v) p = unit3(p);
    struct Face { int a, b, c; };
    std::vector<Face> f = {{0, 11, 5}, {0, 5, 1}, {0, 1, 7}, {0, 7, 10}, {0, 10, 11}, {1, 5, 9}, {5, 11, 4}, {11, 10, 2}, {10, 7, 6}, {7, 1, 8},
                           {3, 9, 4}, {3, 4, 2}, {3, 2, 6}, {3, 6, 8}, {3, 8, 9}, {4, 9, 5}, {2, 4, 11}, {6, 2, 10}, {8, 6, 7}, {9, 8, 1}};
    for (int it = 0; it < subdiv; it++) {
        std::vector<Face> nf;
        nf.reserve(f.size() * 4);
        std::vector<std::pair<uint64_t, int>> cache;
        auto midpoint = [&](int a, int b) {
            uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint64_t)std::max(a, b);
            for (auto &kv : cache) if (kv.first == key) return kv.second;
            v.push_back(unit3(mul(0.5f, add(v[a], v[b]))));
            cache.push_back({key, (int)v.size() - 1});
            return (int)v.size() - 1;
        };
        // a linear-probe cache is O(n^2); fine up to subdivision 3, larger meshes use the hash below
        if (f.size() > 2000) {
            std::vector<std::pair<uint64_t, int>> sorted;
            for (auto &fc : f) {
                int e[3][2] = {{fc.a, fc.b}, {fc.b, fc.c}, {fc.c, fc.a}};
                for (auto &ed : e) sorted.push_back({((uint64_t)std::min(ed[0], ed[1]) << 32) | (uint64_t)std::max(ed[0], ed[1]), -1});
            }
            std::sort(sorted.begin(), sorted.end());
            sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
            for (auto &kv : sorted) {
                int a = (int)(kv.first >> 32), b = (int)(kv.first & 0xffffffffu);
                v.push_back(unit3(mul(0.5f, add(v[a], v[b]))));
                kv.second = (int)v.size() - 1;
            }
            auto mid2 = [&](int a, int b) {
                uint64_t key = ((uint64_t)std::min(a, b) << 32) | (uint64_t)std::max(a, b);
                auto it2 = std::lower_bound(sorted.begin(), sorted.end(), std::make_pair(key, -1));
                return it2->second;
            };
            for (auto &fc : f) {
                int ab = mid2(fc.a, fc.b), bc = mid2(fc.b, fc.c), ca = mid2(fc.c, fc.a);
                nf.push_back({fc.a, ab, ca}); nf.push_back({fc.b, bc, ab}); nf.push_back({fc.c, ca, bc}); nf.push_back({ab, bc, ca});
            }
        } else {
            for (auto &fc : f) {
                int ab = midpoint(fc.a, fc.b), bc = midpoint(fc.b, fc.c), ca = midpoint(fc.c, fc.a);
                nf.push_back({fc.a, ab, ca}); nf.push_back({fc.b, bc, ab}); nf.push_back({fc.c, ca, bc}); nf.push_back({ab, bc, ca});
            }
        }
        f.swap(nf);
    }
    for (auto &fc : f) B.tri_mesh(add(c, mul(r, v[fc.a])), add(c, mul(r, v[fc.b])), add(c, mul(r, v[fc.c])), mat);
}

// smooth synthetic reflectance drawn from the layout PRNG: sigmoid(c2 (l-peak)^2 + c0)
static srt_material synthetic_material(SplitMix &rng, uint32_t type, float fuzz) {
    srt_material m = make_material(type, 0.5f, 0.5f, 0.5f, fuzz, 0.f);
    const float peak = 400.f + 300.f * rng.uni();
    const float a0 = -0.3f + 2.3f * rng.uni();              // sigmoid(a0) in ~[0.36, 0.95]
    const float a1 = -2.5f + 2.0f * rng.uni();              // far from the peak: ~[0.04, 0.33]
    const float c2 = (a1 - a0) / (150.f * 150.f);
    const float coeffs[3] = {c2 * peak * peak + a0, -2.f * c2 * peak, c2};
    srt_bake_sigmoid_spectrum(coeffs, 1.0f, 0, m.spectral_distribution);
    return m;
}

static void scene_random_spheres(srt_scene &s, uint64_t seed) {
    // RTIOW cover-scene layout with every "sphere" a triangle mesh (the reference has no sphere primitive):
    // 22x22 small objects = octahedra (8 tris), 3 big = icospheres subdivision 2 (320 tris), ground quad.
    SplitMix rng(seed ? seed : 0x5eed5eedull);
    s.mats.clear();
    Builder B(s);
    {   // ground: grey 0.5 -> spectrum 0.5 everywhere (Q2 keeps 0.5 at 0.5)
        srt_material g = make_material(SRT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 1.f, 0.f);
        bake_or_fit(g);
        s.mats.push_back(g);
        B.quad(f3(-1000.f, 0.f, -1000.f), f3(0.f, 0.f, 2000.f), f3(2000.f, 0.f, 0.f), 0);
    }
    for (int a = -11; a < 11; a++)
        for (int b = -11; b < 11; b++) {
            const float choose = rng.uni();
            const F3 c = f3((float)a + 0.9f * rng.uni(), 0.2f, (float)b + 0.9f * rng.uni());
            if (!(len3(sub(c, f3(4.f, 0.2f, 0.f))) > 0.9f)) continue;
            srt_material m;
            if (choose < 0.8f) m = synthetic_material(rng, SRT_MAT_LAMBERTIAN, 1.f);
            else if (choose < 0.95f) m = synthetic_material(rng, SRT_MAT_METALLIC, 0.5f * rng.uni());
            else { m = make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kFlintB); srt_material_bake(&m); }
            s.mats.push_back(m);
            add_octahedron(B, c, 0.2f, (uint32_t)s.mats.size() - 1);
        }
    {
        srt_material m = make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kFlintB); srt_material_bake(&m);
        s.mats.push_back(m); add_icosphere(B, f3(0.f, 1.f, 0.f), 1.0f, 2, (uint32_t)s.mats.size() - 1);
        s.mats.push_back(synthetic_material(rng, SRT_MAT_LAMBERTIAN, 1.f)); add_icosphere(B, f3(-4.f, 1.f, 0.f), 1.0f, 2, (uint32_t)s.mats.size() - 1);
        s.mats.push_back(synthetic_material(rng, SRT_MAT_METALLIC, 0.0f)); add_icosphere(B, f3(4.f, 1.f, 0.f), 1.0f, 2, (uint32_t)s.mats.size() - 1);
    }
    // sky: background colour (0.70, 0.80, 1.00) -- RTIOW's, and the one commented out in scene.cu:268 -- as an illuminance
    // spectrum: fitted sigmoid times normalised D65, power 1 (srgb_to_illuminance_spectrum, color_to_spectrum.cuh:158-171)
    {
        const float sky_rgb[3] = {0.70f, 0.80f, 1.00f};
        float sky[3];
        fit_sigmoid_coeffs(sky_rgb, sky);
        bake_sigmoid(sky, srt_powf(1.0f, 2.0f), true, s.background);
    }
    s.cam.vfov = 20.f;
    s.cam.lookfrom[0] = 13.f; s.cam.lookfrom[1] = 2.f; s.cam.lookfrom[2] = 3.f;
    s.cam.lookat[0] = s.cam.lookat[1] = s.cam.lookat[2] = 0.f;
    s.cam.defocus_angle = 0.6f; s.cam.focus_dist = 10.f;
}

static void scene_mesh100k(srt_scene &s, uint64_t seed) {
    // Cornell shell + a 81 920-triangle icosphere (subdivision 6, dielectric) + a metal octahedron + a displaced 96x96
    // floor grid (18 432 triangles, lambertian): 100 372 triangles in total.
    SplitMix rng(seed ? seed : 0x100c0ffeeull);
    s.mats.clear();
    s.mats.push_back(make_material(SRT_MAT_LAMBERTIAN, .73f, .73f, .73f, 1.f, 0.f));
    s.mats.push_back(make_material(SRT_MAT_EMISSIVE, 1.f, 1.f, 1.f, 1.f, 5.f));
    s.mats.push_back(make_material(SRT_MAT_DIELECTRIC, 1.f, 1.f, 1.f, 1.f, 0.f, kFlintB));
    for (auto &m : s.mats) bake_or_fit(m);
    s.mats.push_back(synthetic_material(rng, SRT_MAT_LAMBERTIAN, 1.f));
    s.mats.push_back(synthetic_material(rng, SRT_MAT_METALLIC, 0.2f));
    Builder B(s);
    const uint32_t walls[5] = {0, 0, 0, 0, 0};
    cornell_walls_and_light(B, walls, 1);
    add_icosphere(B, f3(340.f, 200.f, 330.f), 120.f, 6, 2);      // the dielectric object (flint glass, Q1 applies)
    add_octahedron(B, f3(140.f, 120.f, 160.f), 70.f, 4);          // a small metal object
    const int G = 96;
    std::vector<float> hgt((G + 1) * (G + 1));
    for (auto &h : hgt) h = 4.f + 22.f * rng.uni();
    const float x0 = 30.f, z0 = 30.f, cell = 495.f / G;
    for (int iz = 0; iz < G; iz++)
        for (int ix = 0; ix < G; ix++) {
            auto P = [&](int a, int b) { return f3(x0 + cell * a, hgt[b * (G + 1) + a], z0 + cell * b); };
            B.tri_mesh(P(ix, iz), P(ix, iz + 1), P(ix + 1, iz), 3);
            B.tri_mesh(P(ix + 1, iz), P(ix, iz + 1), P(ix + 1, iz + 1), 3);
        }
}

bool scene_builtin_known(int id) {
    return id == SRT_SCENE_CORNELL || id == SRT_SCENE_PRISM || id == SRT_SCENE_TRIS || id == SRT_SCENE_RANDOM_SPHERES ||
           id == SRT_SCENE_MESH100K;
}

void scene_builtin(srt_scene &s, int scene_id, uint64_t seed) {
    s.raw.clear(); s.rec.clear(); s.nodes.clear(); s.bvh_valid = false;
    s.cam = CameraSetup();
    for (int k = 0; k < SRT_N_CIE_SAMPLES; k++) s.background[k] = 0.f;   // background (0,0,0) -> all zeros (Q2)
    s.scene_id = scene_id;
    switch (scene_id) {
    case SRT_SCENE_PRISM: s.name = "prism"; scene_prism(s); break;
    case SRT_SCENE_TRIS: s.name = "tris"; scene_cornell(s, true); break;
    case SRT_SCENE_RANDOM_SPHERES: s.name = "random_spheres"; scene_random_spheres(s, seed); break;
    case SRT_SCENE_MESH100K: s.name = "mesh100k"; scene_mesh100k(s, seed); break;
    default: s.name = "cornell"; scene_cornell(s, false); break;
    }
    s.rec.resize(s.raw.size());
    for (size_t k = 0; k < s.raw.size(); k++) tri_precompute(s.raw[k], s.rec[k]);
}

}  // namespace srt

// ======================================================================================================
// C-ABI, host half
// ======================================================================================================
using namespace srt;

extern "C" {

const char *srt_version(void) { return "srt-amd 0.1 (gfx950)"; }

int srt_camera_init(int image_width, int image_height, float vfov, const float lookfrom[3], const float lookat[3],
                    const float vup[3], float defocus_angle, float focus_dist, srt_camera_data *out) {
    if (!out || !lookfrom || !lookat || !vup || image_width <= 0 || image_height <= 0) { set_global_error("camera: bad argument"); return SRT_ERR_INVALID; }
    // camera::initialize, rendering/camera.cu:7-58
    const F3 center = f3(lookfrom);
    const float theta = deg2rad(vfov);
    const float h = tanf(theta / 2.0f) * focus_dist;
    const float viewport_height = 2.0f * h;
    const float viewport_width = viewport_height * ((float)image_width / (float)image_height);
    const F3 w = unit3(sub(f3(lookfrom), f3(lookat)));
    const F3 u = unit3(cross3(f3(vup), w));
    const F3 v = cross3(w, u);
    const F3 viewport_u = mul(viewport_width, u);
    const F3 viewport_v = mul(viewport_height, neg(v));
    const F3 du = div(viewport_u, (float)image_width);
    const F3 dv = div(viewport_v, (float)image_height);
    const F3 upper_left = sub(sub(sub(center, mul(focus_dist, w)), div(viewport_u, 2.f)), div(viewport_v, 2.f));
    const F3 p00 = add(upper_left, mul(0.5f, add(du, dv)));
    const float defocus_radius = focus_dist * tanf(deg2rad(defocus_angle / 2));
    out->width = (uint32_t)image_width; out->height = (uint32_t)image_height;
    put(out->pixel_delta_u, du); put(out->pixel_delta_v, dv); put(out->pixel00_loc, p00);
    out->defocus_angle = defocus_angle;
    put(out->camera_center, center);
    put(out->defocus_disk_u, mul(defocus_radius, u));
    put(out->defocus_disk_v, mul(defocus_radius, v));
    return SRT_OK;
}

srt_scene *srt_scene_create(void) {
    srt_scene *s = new srt_scene();
    for (int k = 0; k < SRT_N_CIE_SAMPLES; k++) s->background[k] = 0.f;
    return s;
}
srt_scene *srt_scene_builtin(int scene_id, uint64_t seed) {
    if (!scene_builtin_known(scene_id)) { set_global_error("scene: unknown built-in id"); return nullptr; }
    srt_scene *s = new srt_scene();
    scene_builtin(*s, scene_id, seed);
    return s;
}
void srt_scene_destroy(srt_scene *s) { delete s; }

int srt_scene_default_camera(const srt_scene *s, int image_width, int image_height, srt_camera_data *out) {
    if (!s) { set_global_error("scene: null"); return SRT_ERR_INVALID; }
    return srt_camera_init(image_width, image_height, s->cam.vfov, s->cam.lookfrom, s->cam.lookat, s->cam.vup, s->cam.defocus_angle,
                           s->cam.focus_dist, out);
}

int srt_scene_set_triangles(srt_scene *s, const srt_tri_in *tris, size_t n) {
    if (!s || (!tris && n)) { set_global_error("scene: bad argument"); return SRT_ERR_INVALID; }
    s->raw.assign(tris, tris + n);
    s->rec.resize(n);
    for (size_t k = 0; k < n; k++) tri_precompute(s->raw[k], s->rec[k]);
    s->nodes.clear(); s->bvh_valid = false;
    return SRT_OK;
}
int srt_scene_set_materials(srt_scene *s, const srt_material *mats, size_t m) {
    if (!s || (!mats && m)) { set_global_error("scene: bad argument"); return SRT_ERR_INVALID; }
    s->mats.assign(mats, mats + m);
    return SRT_OK;
}
int srt_scene_set_background(srt_scene *s, const float *sp) {
    if (!s || !sp) { set_global_error("scene: bad argument"); return SRT_ERR_INVALID; }
    memcpy(s->background, sp, sizeof(s->background));
    return SRT_OK;
}
size_t srt_scene_tri_count(const srt_scene *s) { return s ? s->raw.size() : 0; }
size_t srt_scene_material_count(const srt_scene *s) { return s ? s->mats.size() : 0; }
int srt_scene_get_triangles(const srt_scene *s, srt_tri_in *out) {
    if (!s || !out) return SRT_ERR_INVALID;
    // hand back the state BEFORE the final init so that a consumer's own tri::init reproduces ours:
    // init is idempotent on unchanged geometry, so the post-init state is an equally valid input.
    memcpy(out, s->raw.data(), s->raw.size() * sizeof(srt_tri_in));
    return SRT_OK;
}
int srt_scene_get_materials(const srt_scene *s, srt_material *out) {
    if (!s || !out) return SRT_ERR_INVALID;
    memcpy(out, s->mats.data(), s->mats.size() * sizeof(srt_material));
    return SRT_OK;
}
int srt_scene_get_background(const srt_scene *s, float *out) {
    if (!s || !out) return SRT_ERR_INVALID;
    memcpy(out, s->background, sizeof(s->background));
    return SRT_OK;
}
int srt_scene_get_tri_records(const srt_scene *s, float *out) {
    if (!s || !out) return SRT_ERR_INVALID;
    for (size_t k = 0; k < s->rec.size(); k++) {
        const TriRecord &r = s->rec[k];
        float *o = out + 12 * k;
        o[0] = r.n[0]; o[1] = r.n[1]; o[2] = r.n[2]; o[3] = r.D; o[4] = r.clockwise ? 1.f : 0.f; o[5] = (float)r.aa_plane;
        memcpy(o + 6, r.box, 6 * sizeof(float));
    }
    return SRT_OK;
}

int srt_bake_sigmoid_spectrum(const float coeffs[3], float scale, int times_d65, float *out) {
    if (!coeffs || !out) { set_global_error("bake: bad argument"); return SRT_ERR_INVALID; }
    bake_sigmoid(coeffs, scale, times_d65 != 0, out);
    return SRT_OK;
}
int srt_set_reference_quirks(int on) { const int was = g_reference_quirks; g_reference_quirks = on ? 1 : 0; return was; }
int srt_material_bake(srt_material *m) {
    if (!m) { set_global_error("bake: null material"); return SRT_ERR_INVALID; }
    float c[3];
    switch (m->material_type) {      // material::compute_spectral_distr, material.cuh:71-84
    case SRT_MAT_EMISSIVE:
        if (!grey_coeffs(m->col, c)) { set_global_error("bake: non-grey colour needs the rgb2spec table (absent upstream)"); return SRT_ERR_UNSUPPORTED; }
        bake_sigmoid(c, srt_powf(m->emission_power, 2.0f), true, m->spectral_distribution);   // pow(power, 2.0f), color_to_spectrum.cuh:181 (Q5)
        return SRT_OK;
    case SRT_MAT_DIELECTRIC:
        for (int i = 0; i < SRT_N_CIE_SAMPLES; i++) m->spectral_distribution[i] = 1.0f;
        return SRT_OK;
    default:
        if (!grey_coeffs(m->col, c)) { set_global_error("bake: non-grey colour needs the rgb2spec table (absent upstream)"); return SRT_ERR_UNSUPPORTED; }
        bake_sigmoid(c, 1.0f, false, m->spectral_distribution);
        return SRT_OK;
    }
}
int srt_fit_sigmoid_coeffs(const float rgb[3], float coeffs[3]) {
    if (!rgb || !coeffs) { set_global_error("fit: bad argument"); return SRT_ERR_INVALID; }
    fit_sigmoid_coeffs(rgb, coeffs);
    return SRT_OK;
}
int srt_background_spectrum(const float rgb[3], float *out) {
    float c[3];
    if (!rgb || !out) { set_global_error("bake: bad argument"); return SRT_ERR_INVALID; }
    if (!grey_coeffs(rgb, c)) { set_global_error("bake: non-grey background needs the rgb2spec table (absent upstream)"); return SRT_ERR_UNSUPPORTED; }
    bake_sigmoid(c, srt_powf(1.0f, 2.0f), true, out);
    return SRT_OK;
}

int srt_scene_build_bvh(srt_scene *s, int mode, uint64_t seed) {
    if (!s) { set_global_error("bvh: null scene"); return SRT_ERR_INVALID; }
    if (mode == SRT_BVH_REFERENCE) return build_bvh_reference(*s, seed);
    if (mode == SRT_BVH_SAH) return build_bvh_sah(*s);
    set_global_error("bvh: unknown mode");
    return SRT_ERR_INVALID;
}
// Child order for a viewpoint.  The traversal is the reference's fixed left-first walk (bvh.cu:154-160), so which child is
// "left" is a property of the tree: build_bvh_sah puts the half nearer to the scene's DEFAULT camera on the left.  A caller
// that renders from another viewpoint (srt_set_camera) re-orders the finished tree with this call -- at every internal node the
// child whose box is nearer to `eye` becomes the left one -- and uploads the scene again.  Topology, boxes and depth are
// unchanged; like any change of the tree it can only alter a result where two triangles tie exactly in t (Q11).
int srt_scene_order_children(srt_scene *s, const float eye[3]) {
    if (!s || !s->bvh_valid || !eye) { set_global_error("srt_scene_order_children: BVH not built / null argument"); return SRT_ERR_INVALID; }
    auto dist2 = [&](const float *bx) {
        double d2 = 0;
        for (int ax = 0; ax < 3; ax++) {
            const double p = eye[ax], lo = bx[2 * ax], hi = bx[2 * ax + 1];
            const double d = p < lo ? lo - p : (p > hi ? p - hi : 0.0);
            d2 += d * d;
        }
        return d2;
    };
    for (BvhNode &nd : s->nodes)
        if (nd.left >= 0 && nd.right >= 0 && dist2(s->nodes[nd.right].box) < dist2(s->nodes[nd.left].box)) std::swap(nd.left, nd.right);
    // a later SAH rebuild orders for the same viewpoint; the scene's default camera is NOT touched (srt_scene_default_camera)
    s->order_eye[0] = eye[0]; s->order_eye[1] = eye[1]; s->order_eye[2] = eye[2]; s->has_order_eye = true;
    return SRT_OK;
}
// Topology optimisation of a built tree for THROUGHPUT-bound launches (no reference counterpart; the tree is an input of the
// reference's traversal).  `passes` rounds of insertion-based optimisation (optimise_bvh_by_reinsertion above; 3 converge), boxes
// and depth recomputed, children ordered by distance to the scene's ordering viewpoint as the SAH builder does.  Measured (cfg 3's
// scene): 17.40 -> 17.25 node records per ray, the 1080p x 1024 spp frame 1.1 % faster -- and the 720p x 256 spp frame, which is
// bound by its longest pixel, 5 % SLOWER: a tree with less total work is not a tree with a cheaper worst pixel.  Hence a call of
// its own, for the caller that knows its launch has many pixels per lane, and not part of srt_scene_build_bvh.
int srt_scene_optimise_bvh(srt_scene *s, int passes) {
    if (!s || !s->bvh_valid) { set_global_error("srt_scene_optimise_bvh: BVH not built"); return SRT_ERR_INVALID; }
    if (passes > 0) optimise_built_tree(*s, passes);
    return SRT_OK;
}
int srt_rotation_matrix(float theta, int axis, float m[9]) {
    if (!m) { set_global_error("srt_rotation_matrix: null matrix"); return SRT_ERR_INVALID; }
    assign_rot_matrix(theta, axis, m);
    return SRT_OK;
}
int srt_scene_is_paired(const srt_scene *s) { return (s && s->bvh_valid && tree_is_paired(*s)) ? 1 : 0; }
size_t srt_scene_node_count(const srt_scene *s) { return (s && s->bvh_valid) ? s->nodes.size() : 0; }
int srt_scene_bvh_depth(const srt_scene *s) { return (s && s->bvh_valid) ? s->depth : 0; }

int srt_scene_get_bvh(const srt_scene *s, int32_t *left, int32_t *right, int32_t *prim, float *boxes) {
    if (!s || !s->bvh_valid || !left || !right || !prim || !boxes) { set_global_error("bvh: not built / bad argument"); return SRT_ERR_INVALID; }
    struct Item { int32_t node, parent; bool is_right; };
    std::vector<Item> st;
    st.push_back({s->root, -1, false});
    int32_t count = 0;
    while (!st.empty()) {
        Item it = st.back(); st.pop_back();
        const BvhNode &nd = s->nodes[it.node];
        const int32_t me = count++;
        if (it.parent >= 0) (it.is_right ? right : left)[it.parent] = me;
        left[me] = right[me] = -1;
        prim[me] = nd.prim;
        memcpy(boxes + 6 * (size_t)me, nd.box, 6 * sizeof(float));
        if (nd.prim < 0) { st.push_back({nd.right, me, true}); st.push_back({nd.left, me, false}); }
    }
    return SRT_OK;
}

}  // extern "C"
