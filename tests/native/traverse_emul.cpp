/*
 * traverse_emul.cpp -- lane-level emulation of the device tree walk on the CPU.
 *
 * TEST HARNESS ONLY.  Compiles path_tracing_amd/csrc/traverse_core.h (the per-lane logic the
 * HIP kernels run) with g++ and drives it one ray at a time over a pt_flat_scene, so that the
 * CPU test-suite can compare the kernel's logic with the oracle without a GPU.  Nothing in
 * the product loads this library.
 */
#include <cstring>
#include <vector>

#include "pt_hip.h"
#include "traverse_core.h"

namespace {

struct CpuMem {
    const pt_flat_node* nodes;
    const pt_tri_isect* tris;
    void load_node(uint32_t i, ptc::NodeData& n) const {
        const pt_flat_node& s = nodes[i];
        for (int k = 0; k < 3; k++) { n.lo[k] = ptc::pt_f2{s.lo[k][0], s.lo[k][1]}; n.hi[k] = ptc::pt_f2{s.hi[k][0], s.hi[k][1]}; }
        n.lo_ref = s.lo_ref; n.hi_ref = s.hi_ref; n.axis = s.axis;
    }
    void load_tri(uint32_t k, ptc::TriData& t) const {
        const pt_tri_isect& s = tris[k];
        t.p0x = s.p0[0]; t.p0y = s.p0[1]; t.p0z = s.p0[2];
        t.e1x = s.e1[0]; t.e1y = s.e1[1]; t.e1z = s.e1[2];
        t.e2x = s.e2[0]; t.e2y = s.e2[1]; t.e2z = s.e2[2];
        t.mesh_index = s.mesh_index; t.leaf_remaining = s.leaf_remaining;
    }
};

struct CpuStack {
    uint32_t data[3 * PT_MAX_TREE_DEPTH + 4];
    int max_sp = 0;
    bool overflow = false;
    void push(int& sp, uint32_t v) {
        if (sp >= (int)(sizeof(data) / sizeof(data[0]))) { overflow = true; return; }
        data[sp++] = v;
        if (sp > max_sp) max_sp = sp;
    }
    uint32_t peek(int i) const { return i >= 0 ? data[i] : 0u; }
    uint32_t peek1(int sp) const { return peek(sp - 1); }
    void peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const { e = peek(sp - 1); w1 = peek(sp - 2); w2 = peek(sp - 3); }
    void push_frame(int& sp, bool keep, uint32_t a, uint32_t b, uint32_t c) { if (keep) push(sp, a); push(sp, b); push(sp, c); }
};

ptc::RootBox root_of(const pt_flat_scene* sc) {
    ptc::RootBox b;
    b.mnx = sc->root_min[0]; b.mny = sc->root_min[1]; b.mnz = sc->root_min[2];
    b.mxx = sc->root_max[0]; b.mxy = sc->root_max[1]; b.mxz = sc->root_max[2];
    return b;
}

} // namespace

extern "C" {

/* Returns the largest stack use (dwords) seen, or -1 on overflow. */
int emul_raycast_closest(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, int32_t* idx, float* t,
                         float* u, float* v, int32_t* flat_pos) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    int worst = 0;
    for (uint64_t i = 0; i < n; i++) {
        ptc::Ray9 r;
        r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
        r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
        ptc::ray_set_recip(r);
        CpuStack st;
        ptc::Hit h;
        bool found = sc->n_tris != 0 && ptc::trace_lane<false>(mem, sc->root_ref, root_of(sc), r, st, h);
        if (st.overflow) return -1;
        if (st.max_sp > worst) worst = st.max_sp;
        if (found) { idx[i] = (int32_t)h.mesh; t[i] = h.t; u[i] = h.u; v[i] = h.v; if (flat_pos) flat_pos[i] = (int32_t)h.tri; }
        else { idx[i] = -1; t[i] = u[i] = v[i] = 0.0f; if (flat_pos) flat_pos[i] = -1; }
    }
    return worst;
}

int emul_raycast_shadow(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const float* dist,
                        uint8_t* visible) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    int worst = 0;
    for (uint64_t i = 0; i < n; i++) {
        ptc::Ray9 r;
        r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
        r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
        ptc::ray_set_recip(r);
        CpuStack st;
        ptc::Hit h;
        bool vis;
        if (sc->n_tris == 0) vis = true;
        else if (dist[i] >= 1.0f) {
            /* v <= 1 <= dist for every hit, so any hit occludes */
            vis = !ptc::trace_lane<true>(mem, sc->root_ref, root_of(sc), r, st, h);
        } else {
            bool found = ptc::trace_lane<false>(mem, sc->root_ref, root_of(sc), r, st, h);
            vis = !found || h.v > dist[i];
        }
        if (st.overflow) return -1;
        if (st.max_sp > worst) worst = st.max_sp;
        visible[i] = vis ? 1 : 0;
    }
    return worst;
}

} // extern "C"

/* Step sequence of one ray through the lane state machine, one byte per step:
 * 'T' triangle, 'N' interior node, 'P' pop.  Returns the length (or -1 if cap is too small).
 * Used to study wave scheduling policies offline. */
extern "C" int emul_step_trace(const pt_flat_scene* sc, const float* o, const float* d, char* out, int cap) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    ptc::Lane L;
    L.r.ox = o[0]; L.r.oy = o[1]; L.r.oz = o[2];
    L.r.dx = d[0]; L.r.dy = d[1]; L.r.dz = d[2];
    ptc::ray_set_recip(L.r);
    CpuStack st;
    int n = 0;
    if (sc->n_tris == 0) return 0;
    ptc::lane_start(L, sc->root_ref, root_of(sc), true);
    while (!ptc::lane_finished(L)) {
        if (n >= cap) return -1;
        if (ptc::lane_has_leaf(L)) { ptc::TriData tr; mem.load_tri(ptc::lane_leaf_pos(L), tr); ptc::lane_tri<false>(L, tr); out[n++] = 'T'; }
        else if (ptc::lane_has_node(L)) {
            ptc::NodeData nd; mem.load_node(L.cur, nd);
            if (L.tame) ptc::lane_node<false, true>(L, nd, st); else ptc::lane_node<false, false>(L, nd, st);
            out[n++] = 'N';
        } else { ptc::lane_pop<false>(L, st); out[n++] = 'P'; }
    }
    return n;
}

/* What the device walks FETCH, counted with the same lane logic: one node record per interior-node
 * step and one triangle record per triangle step (the kernels load exactly one record per step
 * when they test one triangle per round).  rays8 holds n rays of 8 floats: origin, direction, light
 * distance, tag (0: closest-hit cast; 1: shadow cast that is traced; anything else: skipped).
 * counts[0..3] += closest-hit casts: rays, node records, triangle records, stack pops
 * counts[4..7] += shadow casts answered by the any-hit walk (light at distance >= 1): the same four
 * counts[8..11] += shadow casts that need the closest hit (nearer lights): the same four */
namespace {
struct CountingMem {
    CpuMem mem;
    mutable uint64_t nodes = 0, tris = 0;
    void load_node(uint32_t i, ptc::NodeData& n) const { nodes++; mem.load_node(i, n); }
    void load_tri(uint32_t k, ptc::TriData& t) const { tris++; mem.load_tri(k, t); }
};

template <bool ANYHIT>
uint64_t count_lane(const CountingMem& mem, const pt_flat_scene* sc, const ptc::Ray9& r) {
    ptc::Lane L;
    L.r = r;
    CpuStack st;
    uint64_t pops = 0;
    const ptc::RootBox root = root_of(sc);
    ptc::lane_start(L, sc->root_ref, root, ptc::root_is_tame(root));
    while (!ptc::lane_finished(L)) {
        if (ptc::lane_has_leaf(L)) { ptc::TriData tr; mem.load_tri(ptc::lane_leaf_pos(L), tr); ptc::lane_tri<ANYHIT>(L, tr); }
        else if (ptc::lane_has_node(L)) {
            ptc::NodeData nd; mem.load_node(L.cur, nd);
            if (L.tame) ptc::lane_node<ANYHIT, true>(L, nd, st); else ptc::lane_node<ANYHIT, false>(L, nd, st);
        } else { ptc::lane_pop<ANYHIT>(L, st); pops++; }
    }
    return pops;
}
} // namespace

extern "C" int emul_work_counts(const pt_flat_scene* sc, uint64_t n, const float* rays8, uint64_t* counts) {
    if (sc->n_tris == 0) return 0;
    CountingMem mem; mem.mem.nodes = sc->nodes; mem.mem.tris = sc->tris;
    for (uint64_t i = 0; i < n; i++) {
        const float* q = rays8 + 8 * i;
        ptc::Ray9 r;
        r.ox = q[0]; r.oy = q[1]; r.oz = q[2]; r.dx = q[3]; r.dy = q[4]; r.dz = q[5];
        ptc::ray_set_recip(r);
        int cls;
        if (q[7] == 0.0f) cls = 0;
        else if (q[7] == 1.0f) cls = q[6] >= 1.0f ? 1 : 2;
        else continue;
        mem.nodes = 0; mem.tris = 0;
        const uint64_t pops = cls == 1 ? count_lane<true>(mem, sc, r) : count_lane<false>(mem, sc, r);
        uint64_t* c = counts + 4 * cls;
        c[0] += 1; c[1] += mem.nodes; c[2] += mem.tris; c[3] += pops;
    }
    return 0;
}

/* emul_work_counts for the shadow casts only, splitting the node records fetched by a per-node flag
 * (flag[i] != 0 or 0): counts[0] += traced shadow rays, [1] += node visits with the flag set,
 * [2] += node visits without, [3] += triangle records.  For layout studies. */
namespace {
struct FlagMem {
    CpuMem mem; const uint8_t* flag;
    mutable uint64_t set = 0, clear = 0, tris = 0;
    void load_node(uint32_t i, ptc::NodeData& n) const { if (flag[i]) set++; else clear++; mem.load_node(i, n); }
    void load_tri(uint32_t k, ptc::TriData& t) const { tris++; mem.load_tri(k, t); }
};
} // namespace
extern "C" int emul_shadow_counts_by_flag(const pt_flat_scene* sc, uint64_t n, const float* rays8, const uint8_t* flag, uint64_t* counts) {
    if (sc->n_tris == 0) return 0;
    FlagMem mem; mem.mem.nodes = sc->nodes; mem.mem.tris = sc->tris; mem.flag = flag;
    const ptc::RootBox root = root_of(sc);
    for (uint64_t i = 0; i < n; i++) {
        const float* q = rays8 + 8 * i;
        if (q[7] != 1.0f || !(q[6] >= 1.0f)) continue;
        ptc::Ray9 r;
        r.ox = q[0]; r.oy = q[1]; r.oz = q[2]; r.dx = q[3]; r.dy = q[4]; r.dz = q[5];
        ptc::ray_set_recip(r);
        CpuStack st; ptc::Hit h;
        ptc::trace_lane<true>(mem, sc->root_ref, root, r, st, h);
        counts[0]++;
    }
    counts[1] += mem.set; counts[2] += mem.clear; counts[3] += mem.tris;
    return 0;
}

/* ---- the any-hit walk over quantised interior nodes (traverse_core.h "q-nodes") ---- */
#include "qnode_build.h"

namespace {
struct QMem {
    CpuMem mem;
    const ptq::QNodes* qn;
    mutable uint64_t q_nodes = 0, exact_nodes = 0, tris = 0;
    void load_node(uint32_t pos, ptc::NodeData& n) const {
        if (pos < PT_Q_EXACT_BASE) { q_nodes++; ptc::qnode_decode(qn->rec[pos].w, qn->grid, n); }
        else { exact_nodes++; mem.load_node(pos - PT_Q_EXACT_BASE, n); ptc::exact_refs_to_q_space(n); }
    }
    void load_tri(uint32_t k, ptc::TriData& t) const { tris++; mem.load_tri(k, t); }
};
} // namespace

extern "C" {

/* emul_raycast_shadow with the any-hit walks on q-nodes wherever the kernels use them: rays whose
 * light is at distance >= 1 and whose box tests cannot produce a NaN (ray and root box tame).
 * counts (optional, 4 words) += rays walked on q-nodes, q-node records, exact node records and
 * triangle records they fetched.  Returns -1 on stack overflow, -2 if the scene gets no q-nodes. */
int emul_raycast_shadow_q(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const float* dist,
                          uint8_t* visible, uint64_t* counts) {
    ptq::QNodes qn;
    ptq::build_qnodes(sc, qn);
    if (sc->n_tris != 0 && qn.rec.empty()) return -2;
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    QMem qmem; qmem.mem = mem; qmem.qn = &qn;
    const ptc::RootBox root = root_of(sc);
    const bool scene_tame = ptc::root_is_tame(root);
    uint64_t q_rays = 0;
    for (uint64_t i = 0; i < n; i++) {
        ptc::Ray9 r;
        r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
        r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
        ptc::ray_set_recip(r);
        CpuStack st;
        ptc::Hit h;
        bool vis;
        if (sc->n_tris == 0) vis = true;
        else if (dist[i] >= 1.0f) {
            if (scene_tame && ptc::ray_is_tame(r)) { vis = !ptc::trace_lane<true>(qmem, qn.root, root, r, st, h); q_rays++; }
            else vis = !ptc::trace_lane<true>(mem, sc->root_ref, root, r, st, h);
        } else {
            bool found = ptc::trace_lane<false>(mem, sc->root_ref, root, r, st, h);
            vis = !found || h.v > dist[i];
        }
        if (st.overflow) return -1;
        visible[i] = vis ? 1 : 0;
    }
    if (counts) { counts[0] += q_rays; counts[1] += qmem.q_nodes; counts[2] += qmem.exact_nodes; counts[3] += qmem.tris; }
    return 0;
}

/* "No triangle nearer than max_t" (the bounded shadow test) by the any-hit walk on q-nodes. */
int emul_raycast_occluded_q(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const float* max_t, uint8_t* visible) {
    ptq::QNodes qn;
    ptq::build_qnodes(sc, qn);
    if (sc->n_tris != 0 && qn.rec.empty()) return -2;
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    QMem qmem; qmem.mem = mem; qmem.qn = &qn;
    const ptc::RootBox root = root_of(sc);
    const bool scene_tame = ptc::root_is_tame(root);
    for (uint64_t i = 0; i < n; i++) {
        ptc::Lane L;
        L.r.ox = o[3 * i]; L.r.oy = o[3 * i + 1]; L.r.oz = o[3 * i + 2];
        L.r.dx = d[3 * i]; L.r.dy = d[3 * i + 1]; L.r.dz = d[3 * i + 2];
        ptc::ray_set_recip(L.r);
        CpuStack st;
        if (sc->n_tris == 0) { visible[i] = 1; continue; }
        const bool use_q = scene_tame && ptc::ray_is_tame(L.r);
        ptc::lane_start(L, use_q ? qn.root : sc->root_ref, root, scene_tame);
        L.best.t = max_t[i];   /* a triangle counts only if it is strictly nearer than the bound */
        while (!ptc::lane_finished(L)) {
            if (ptc::lane_has_leaf(L)) { ptc::TriData tr; mem.load_tri(ptc::lane_leaf_pos(L), tr); ptc::lane_tri<true>(L, tr); }
            else if (ptc::lane_has_node(L)) {
                ptc::NodeData nd;
                if (use_q) qmem.load_node(L.cur, nd); else mem.load_node(L.cur, nd);
                if (L.tame) ptc::lane_node<true, true>(L, nd, st); else ptc::lane_node<true, false>(L, nd, st);
            } else ptc::lane_pop<true>(L, st);
        }
        if (st.overflow) return -1;
        visible[i] = L.found ? 0 : 1;
    }
    return 0;
}

} // extern "C"
