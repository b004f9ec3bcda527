/*
 * occl_sim.cpp -- STUDY TOOL (CPU): the any-hit walk of path_tracing_amd/csrc/traverse_core.h over the
 * reference's tree and over the occlusion tree of occl_build.h, ray by ray, with a counter on every
 * record fetched.  Built and driven by scripts/occl_eval.py.
 */
#include <cstdint>

#include "occl_build.h"

namespace {
struct Mem {
    const pt_flat_node* nodes;
    const pt_tri_isect* tris;
    mutable uint64_t n_nodes = 0, n_tris = 0;
    void load_node(uint32_t i, ptc::NodeData& n) const {
        n_nodes++;
        const pt_flat_node& s = nodes[i];
        for (int k = 0; k < 3; k++) { n.lo[k] = ptc::pt_f2{s.lo[k][0], s.lo[k][1]}; n.hi[k] = ptc::pt_f2{s.hi[k][0], s.hi[k][1]}; }
        n.lo_ref = s.lo_ref; n.hi_ref = s.hi_ref; n.axis = s.axis;
    }
    void load_tri(uint32_t k, ptc::TriData& t) const {
        n_tris++;
        const pt_tri_isect& s = tris[k];
        t.p0x = s.p0[0]; t.p0y = s.p0[1]; t.p0z = s.p0[2];
        t.e1x = s.e1[0]; t.e1y = s.e1[1]; t.e1z = s.e1[2];
        t.e2x = s.e2[0]; t.e2y = s.e2[1]; t.e2z = s.e2[2];
        t.mesh_index = s.mesh_index; t.leaf_remaining = s.leaf_remaining;
    }
};
struct Stack {
    uint32_t data[64];
    bool overflow = false;
    void push(int& sp, uint32_t v) { if (sp >= 64) { overflow = true; return; } data[sp++] = v; }
    uint32_t peek1(int sp) const { return sp >= 1 ? data[sp - 1] : 0u; }
    void peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const { e = peek1(sp); w1 = peek1(sp - 1); w2 = peek1(sp - 2); }
    void push_frame(int& sp, bool keep, uint32_t a, uint32_t b, uint32_t c) { if (keep) push(sp, a); push(sp, b); push(sp, c); }
};
}

/* Tame rays with dist >= 1 (the ones the any-hit kernel answers): visible[i] from the occlusion tree.
 * counts[0] += such rays, [1], [2] += node and triangle records on the occlusion tree, [3], [4] += on the
 * reference's tree, [5] += answers that differ; [6] = depth of the occlusion tree, [7] = its node count.
 * Other rays get visible[i] = 2.  Returns -1 on stack overflow, -2 if there is no occlusion tree. */
extern "C" int occl_sim(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const float* dist, uint8_t* visible,
                        uint64_t* counts, int threads) {
    pto::OccTree occ;
    pto::build_occlusion_tree(sc, occ, threads);
    if (occ.nodes.empty()) return -2;
    Mem ref; ref.nodes = sc->nodes; ref.tris = sc->tris;
    Mem om; om.nodes = occ.nodes.data(); om.tris = sc->tris;
    ptc::RootBox root;
    root.mnx = sc->root_min[0]; root.mny = sc->root_min[1]; root.mnz = sc->root_min[2];
    root.mxx = sc->root_max[0]; root.mxy = sc->root_max[1]; root.mxz = sc->root_max[2];
    bool overflow = false;
    for (uint64_t i = 0; i < n; i++) {
        ptc::Ray9 r;
        r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
        r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
        ptc::ray_set_recip(r);
        if (!(dist[i] >= 1.0f) || !ptc::ray_is_tame(r)) { visible[i] = 2; continue; }
        Stack s1, s2; ptc::Hit h;
        const bool hit_occ = ptc::trace_lane<true>(om, occ.root, root, r, s1, h);
        const bool hit_ref = ptc::trace_lane<true>(ref, sc->root_ref, root, r, s2, h);
        overflow = overflow || s1.overflow || s2.overflow;
        visible[i] = hit_occ ? 0 : 1;
        counts[0]++; counts[5] += hit_occ != hit_ref;
    }
    counts[1] += om.n_nodes; counts[2] += om.n_tris; counts[3] += ref.n_nodes; counts[4] += ref.n_tris;
    counts[6] = (uint64_t)occ.depth; counts[7] = occ.nodes.size();
    return overflow ? -1 : 0;
}
