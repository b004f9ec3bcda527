/*
 * occl_build.h -- host-side construction of the "occlusion tree": a second hierarchy over the
 * LEAVES of the reference's tree, used only by the any-hit walks (light visibility).
 *
 * Why a second hierarchy gives the same answers.  The reference's walk enters a node only through
 * the test of that node's own box (kdtree.cpp:273), and a far child is skipped only when the near
 * side has already hit something (kdtree.cpp:367).  So "does the ray hit anything" is: is there a
 * leaf whose own box and all of whose ancestors' boxes the ray hits, with a triangle that the
 * triangle test accepts.  Node boxes are unions of their triangles' boxes (kdtree.cpp:60-61), so an
 * ancestor's bounds enclose a leaf's exactly, in binary32, and the slab test's terms
 * fl(fl(b - o) * r) are monotone in the bound b: for a ray whose box tests produce no NaN (a "tame"
 * ray, traverse_core.h) a hit of a leaf's own box implies a hit of every box that encloses it.  The
 * answer is therefore: is there a leaf whose OWN box is hit and one of whose triangles is hit -- a
 * statement about the set of leaves, not about the tree above them.  Any binary tree over the same
 * leaves whose interior boxes enclose their leaves, and which tests every leaf against its own exact
 * box, finds the same answer (the same argument as for quantised nodes in traverse_core.h).
 *
 * The reference splits at the median of a sparse candidate list (kdtree.cpp:92-143); its interior
 * boxes overlap a good deal.  This tree is built top-down with the binned surface-area heuristic
 * over the leaves' boxes, each leaf weighted by its triangle count, and a ray that hits nothing
 * (which has to visit every node whose box it hits) meets fewer of its nodes.
 *
 * Records are pt_flat_node, numbered depth-first in pre-order like the reference's (a subtree over
 * n leaves has n-1 nodes, so a node's right child sits at node + n_left); leaf references are the
 * reference tree's own, so they name the same triangle ranges.  Rays that are not tame, and walks
 * that need the closest hit, use the reference's tree.
 *
 * STUDY TOOL (host code, plain C++), not part of the library: scripts/occl_sim.cpp walks both trees with
 * the kernels' lane logic and counts what each fetches.  Measured on the 1.34 M-triangle stand-in scene
 * (shadow rays of a 1920x1080 frame sample): 25.9 node records per ray against 27.2 on the reference's
 * tree (-5 %), the same 13.5 triangle records, the same answers -- rays that start on a surface have to
 * come down the whole tree to their own neighbourhood whatever the hierarchy, and the leaves (and so the
 * leaf boxes a ray hits) are the reference's.  Too little to pay for a second node array; not built
 * into the kernels.
 */
#ifndef PT_OCCL_BUILD_H
#define PT_OCCL_BUILD_H

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <thread>
#include <vector>

#include "pt_hip.h"
#include "traverse_core.h"

namespace pto {

#define PT_OCC_MAX_DEPTH 56        /* below the any-hit walk's stack capacity (one deferred child per level) */
#define PT_OCC_SAH_DEPTH 36        /* from this depth on ranges are halved, so the rest is at most 20 levels */
#define PT_OCC_BINS 16

struct OccTree {
    std::vector<pt_flat_node> nodes;   /* empty: the scene gets no occlusion tree */
    uint32_t root = PT_REF_NONE;
    int depth = 0;
};

struct LeafPrim {
    float mn[3], mx[3];
    float c[3];        /* box centre (only steers the partition) */
    uint32_t ref;
    uint32_t weight;   /* triangles in the leaf */
};

struct Box3 {
    float mn[3], mx[3];
    void clear() { for (int a = 0; a < 3; a++) { mn[a] = INFINITY; mx[a] = -INFINITY; } }
    void add(const float* lo, const float* hi) { for (int a = 0; a < 3; a++) { if (lo[a] < mn[a]) mn[a] = lo[a]; if (hi[a] > mx[a]) mx[a] = hi[a]; } }
    void add(const Box3& b) { add(b.mn, b.mx); }
    double half_area() const {
        const double x = (double)mx[0] - mn[0], y = (double)mx[1] - mn[1], z = (double)mx[2] - mn[2];
        return (x < 0 || y < 0 || z < 0) ? 0.0 : x * y + y * z + z * x;
    }
};

struct Builder {
    std::vector<LeafPrim> prims;
    std::vector<pt_flat_node> nodes;
    int max_depth = 0;
    std::atomic<bool> failed{false};

    static Box3 box_of(const LeafPrim* p, size_t n) {
        Box3 b; b.clear();
        for (size_t i = 0; i < n; i++) b.add(p[i].mn, p[i].mx);
        return b;
    }

    /* Splits prims[lo, hi) in place; returns the size of the left part (1 .. n-1) and the axis. */
    size_t split(size_t lo, size_t hi, int depth, uint32_t& axis_out) {
        LeafPrim* p = prims.data() + lo;
        const size_t n = hi - lo;
        float cmn[3] = {INFINITY, INFINITY, INFINITY}, cmx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = 0; i < n; i++)
            for (int a = 0; a < 3; a++) { cmn[a] = std::min(cmn[a], p[i].c[a]); cmx[a] = std::max(cmx[a], p[i].c[a]); }
        int widest = 0;
        for (int a = 1; a < 3; a++) if (cmx[a] - cmn[a] > cmx[widest] - cmn[widest]) widest = a;
        axis_out = (uint32_t)widest;
        double best = INFINITY; int best_axis = -1, best_bin = -1;
        if (depth < PT_OCC_SAH_DEPTH && n > 2) {
            for (int a = 0; a < 3; a++) {
                const float ext = cmx[a] - cmn[a];
                if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
                const float scale = (float)PT_OCC_BINS / ext;
                Box3 bb[PT_OCC_BINS]; uint64_t bw[PT_OCC_BINS];
                for (int k = 0; k < PT_OCC_BINS; k++) { bb[k].clear(); bw[k] = 0; }
                for (size_t i = 0; i < n; i++) {
                    int k = (int)((p[i].c[a] - cmn[a]) * scale);
                    k = k < 0 ? 0 : k >= PT_OCC_BINS ? PT_OCC_BINS - 1 : k;
                    bb[k].add(p[i].mn, p[i].mx); bw[k] += p[i].weight;
                }
                double right_cost[PT_OCC_BINS];
                Box3 acc; acc.clear(); uint64_t w = 0;
                for (int k = PT_OCC_BINS - 1; k > 0; k--) { acc.add(bb[k]); w += bw[k]; right_cost[k] = w ? acc.half_area() * (double)w : INFINITY; }
                acc.clear(); w = 0;
                for (int k = 0; k + 1 < PT_OCC_BINS; k++) {
                    acc.add(bb[k]); w += bw[k];
                    if (w == 0 || right_cost[k + 1] == INFINITY) continue;
                    const double cost = acc.half_area() * (double)w + right_cost[k + 1];
                    if (cost < best) { best = cost; best_axis = a; best_bin = k; }
                }
            }
        }
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = (float)PT_OCC_BINS / (cmx[a] - cmn[a]);
            LeafPrim* mid = std::partition(p, p + n, [&](const LeafPrim& q) {
                int k = (int)((q.c[a] - cmn[a]) * scale);
                k = k < 0 ? 0 : k >= PT_OCC_BINS ? PT_OCC_BINS - 1 : k;
                return k <= best_bin;
            });
            const size_t nl = (size_t)(mid - p);
            if (nl > 0 && nl < n) { axis_out = (uint32_t)a; return nl; }
        }
        /* halves by position along the widest axis (also the depth guard) */
        const size_t nl = n / 2;
        std::nth_element(p, p + nl, p + n, [&](const LeafPrim& x, const LeafPrim& y) { return x.c[widest] < y.c[widest]; });
        return nl;
    }

    /* Builds the subtree over prims[lo, hi) (at least two) into nodes[at ...]; returns nothing: the
     * subtree's root is nodes[at]. */
    void build(size_t lo, size_t hi, size_t at, int depth, int* deepest) {
        if (depth > *deepest) *deepest = depth;
        if (depth >= PT_OCC_MAX_DEPTH) { failed = true; return; }
        uint32_t axis = 0;
        const size_t nl = split(lo, hi, depth, axis);
        const size_t mid = lo + nl;
        pt_flat_node nd{};
        const Box3 bl = box_of(prims.data() + lo, nl), br = box_of(prims.data() + mid, hi - mid);
        for (int a = 0; a < 3; a++) { nd.lo[a][0] = bl.mn[a]; nd.lo[a][1] = bl.mx[a]; nd.hi[a][0] = br.mn[a]; nd.hi[a][1] = br.mx[a]; }
        nd.axis = axis;
        nd.lo_ref = nl == 1 ? prims[lo].ref : (uint32_t)(at + 1);
        nd.hi_ref = hi - mid == 1 ? prims[mid].ref : (uint32_t)(at + nl);
        nodes[at] = nd;
        if (nl > 1) build(lo, mid, at + 1, depth + 1, deepest);
        if (hi - mid > 1) build(mid, hi, at + nl, depth + 1, deepest);
    }
};

/* Builds the occlusion tree of a validated flat tree.  Leaves `out.nodes` empty when the scheme does
 * not apply: fewer than two leaves, or a root box that is not of moderate size (every walk then takes
 * the NaN-exact path on the reference's tree). */
inline void build_occlusion_tree(const pt_flat_scene* f, OccTree& out, int threads = 1) {
    out.nodes.clear(); out.root = PT_REF_NONE; out.depth = 0;
    if (f->n_nodes == 0 || f->n_tris == 0 || (f->root_ref & PT_REF_LEAF)) return;
    if (f->n_nodes >= PT_Q_EXACT_BASE) return;
    ptc::RootBox rb;
    rb.mnx = f->root_min[0]; rb.mny = f->root_min[1]; rb.mnz = f->root_min[2];
    rb.mxx = f->root_max[0]; rb.mxy = f->root_max[1]; rb.mxz = f->root_max[2];
    if (!ptc::root_is_tame(rb)) return;
    Builder b;
    b.prims.reserve((size_t)f->n_nodes + 1);
    for (uint64_t i = 0; i < f->n_nodes; i++) {
        const pt_flat_node& nd = f->nodes[i];
        for (int c = 0; c < 2; c++) {
            const uint32_t ref = c == 0 ? nd.lo_ref : nd.hi_ref;
            if (!(ref & PT_REF_LEAF)) continue;
            const float (*box)[2] = c == 0 ? nd.lo : nd.hi;
            LeafPrim p;
            for (int a = 0; a < 3; a++) { p.mn[a] = box[a][0]; p.mx[a] = box[a][1]; p.c[a] = 0.5f * box[a][0] + 0.5f * box[a][1]; }
            p.ref = ref;
            uint32_t count = (ref >> PT_REF_COUNT_SHIFT) & PT_REF_COUNT_MASK;
            if (count == 0) count = f->tris[ref & PT_REF_START_MASK].leaf_count;
            p.weight = count ? count : 1u;
            b.prims.push_back(p);
        }
    }
    const size_t n = b.prims.size();
    if (n < 2) return;
    b.nodes.assign(n - 1, pt_flat_node{});
    /* the top of the tree on this thread down to ranges small enough to hand out, the ranges side by side */
    struct Job { size_t lo, hi, at; int depth; };
    std::vector<Job> jobs, open{{0, n, 0, 0}};
    int deepest = 0;
    const size_t grain = threads > 1 ? std::max<size_t>(n / (size_t)(threads * 8), 1024) : n;
    while (!open.empty()) {
        const Job j = open.back(); open.pop_back();
        if (j.hi - j.lo <= grain || threads <= 1) { jobs.push_back(j); continue; }
        if (j.depth > deepest) deepest = j.depth;
        uint32_t axis = 0;
        const size_t nl = b.split(j.lo, j.hi, j.depth, axis), mid = j.lo + nl;
        pt_flat_node nd{};
        const Box3 bl = Builder::box_of(b.prims.data() + j.lo, nl), br = Builder::box_of(b.prims.data() + mid, j.hi - mid);
        for (int a = 0; a < 3; a++) { nd.lo[a][0] = bl.mn[a]; nd.lo[a][1] = bl.mx[a]; nd.hi[a][0] = br.mn[a]; nd.hi[a][1] = br.mx[a]; }
        nd.axis = axis;
        nd.lo_ref = nl == 1 ? b.prims[j.lo].ref : (uint32_t)(j.at + 1);
        nd.hi_ref = j.hi - mid == 1 ? b.prims[mid].ref : (uint32_t)(j.at + nl);
        b.nodes[j.at] = nd;
        if (nl > 1) open.push_back({j.lo, mid, j.at + 1, j.depth + 1});
        if (j.hi - mid > 1) open.push_back({mid, j.hi, j.at + nl, j.depth + 1});
    }
    std::vector<int> deep(jobs.size(), 0);
    if (threads <= 1 || jobs.size() <= 1) {
        for (size_t k = 0; k < jobs.size(); k++) b.build(jobs[k].lo, jobs[k].hi, jobs[k].at, jobs[k].depth, &deep[k]);
    } else {
        std::vector<std::thread> pool;
        const int T = (int)std::min<size_t>((size_t)threads, jobs.size());
        for (int t = 0; t < T; t++)
            pool.emplace_back([&, t]() { for (size_t k = (size_t)t; k < jobs.size(); k += (size_t)T) b.build(jobs[k].lo, jobs[k].hi, jobs[k].at, jobs[k].depth, &deep[k]); });
        for (auto& th : pool) th.join();
    }
    for (int d : deep) deepest = std::max(deepest, d);
    if (b.failed) return;
    out.nodes.swap(b.nodes);
    out.root = 0u;
    out.depth = deepest + 1;
}

} // namespace pto

#endif
