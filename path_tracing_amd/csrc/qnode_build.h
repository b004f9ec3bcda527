/*
 * qnode_build.h -- host-side construction of the quantised interior nodes that the any-hit walks
 * use (see "q-nodes" in traverse_core.h for what they are and why the walks' answers do not change).
 *
 * Host code only (plain C++): pt_scene_create builds the array from the flat tree when it uploads a
 * scene; the CPU-side lane emulation of the tests builds the same array to check the walk.
 */
#ifndef PT_QNODE_BUILD_H
#define PT_QNODE_BUILD_H

#include <cmath>
#include <cstdint>
#include <vector>

#include "pt_hip.h"
#include "traverse_core.h"

namespace ptq {

struct QNodeRec { uint32_t w[8]; };   /* 32 bytes, layout as qnode_decode reads it */

struct QNodes {
    std::vector<QNodeRec> rec;        /* empty: the scene gets no q-nodes (walks use exact records only) */
    ptc::QGrid grid{};
    uint32_t root = PT_REF_NONE;      /* position of the root for walks that use q-nodes */
};

/* Largest grid step whose plane is not above v / smallest whose plane is not below v, judged on the
 * decoded binary32 value exactly as the kernel decodes it. */
inline uint32_t q_floor(float v, float s, float g0) {
    if (!(s > 0.0f)) return 0u;
    double e = std::floor(((double)v - (double)g0) / (double)s);
    int64_t q = e < 0.0 ? 0 : e > 65535.0 ? 65535 : (int64_t)e;
    while (q < 65535 && ptc::q_plane((uint32_t)(q + 1), s, g0) <= v) q++;
    while (q > 0 && ptc::q_plane((uint32_t)q, s, g0) > v) q--;
    return (uint32_t)q;
}
inline uint32_t q_ceil(float v, float s, float g0) {
    if (!(s > 0.0f)) return 0u;
    double e = std::ceil(((double)v - (double)g0) / (double)s);
    int64_t q = e < 0.0 ? 0 : e > 65535.0 ? 65535 : (int64_t)e;
    while (q > 0 && ptc::q_plane((uint32_t)(q - 1), s, g0) >= v) q--;
    while (q < 65535 && ptc::q_plane((uint32_t)q, s, g0) < v) q++;
    return (uint32_t)q;
}

/* Builds the q-nodes of a validated flat tree.  Leaves `out.rec` empty when the scheme does not
 * apply: no interior node, a root box that is not of moderate size (the walks then take their
 * NaN-exact path throughout), too many nodes for the position encoding, or a box that the grid
 * cannot enclose (never, unless a node's box sticks out of the root's). */
inline void build_qnodes(const pt_flat_scene* f, QNodes& out) {
    out.rec.clear();
    out.root = PT_REF_NONE;
    if (f->n_nodes == 0 || f->n_tris == 0 || (f->root_ref & PT_REF_LEAF)) return;
    if (f->n_nodes >= PT_Q_EXACT_BASE) return;
    ptc::RootBox rb;
    rb.mnx = f->root_min[0]; rb.mny = f->root_min[1]; rb.mnz = f->root_min[2];
    rb.mxx = f->root_max[0]; rb.mxy = f->root_max[1]; rb.mxz = f->root_max[2];
    if (!ptc::root_is_tame(rb)) return;
    for (int a = 0; a < 3; a++) {
        const float lo = f->root_min[a], hi = f->root_max[a];
        if (!(lo <= hi)) return;
        float s = (float)(((double)hi - (double)lo) / 65535.0);
        /* the last grid step must reach the root's maximum */
        for (int k = 0; k < 64 && s > 0.0f && ptc::q_plane(65535u, s, lo) < hi; k++) s = std::nextafter(s, INFINITY);
        if (s > 0.0f && ptc::q_plane(65535u, s, lo) < hi) return;
        out.grid.g0[a] = lo;
        out.grid.s[a] = s;
    }
    const uint64_t n = f->n_nodes;
    /* a node gets a q-node when both of its children are interior */
    std::vector<uint32_t> qindex((size_t)n, PT_REF_NONE);
    uint32_t n_q = 0;
    for (uint64_t i = 0; i < n; i++) {
        const pt_flat_node& nd = f->nodes[i];
        if (!(nd.lo_ref & PT_REF_LEAF) && !(nd.hi_ref & PT_REF_LEAF)) qindex[(size_t)i] = n_q++;
    }
    auto position = [&](uint32_t node) { return qindex[node] != PT_REF_NONE ? qindex[node] : PT_Q_EXACT_BASE + node; };
    out.rec.resize(n_q);
    bool ok = true;
    for (uint64_t i = 0; i < n && ok; i++) {
        if (qindex[(size_t)i] == PT_REF_NONE) continue;
        const pt_flat_node& nd = f->nodes[i];
        QNodeRec& r = out.rec[qindex[(size_t)i]];
        for (int c = 0; c < 2; c++) {
            const float (*box)[2] = c == 0 ? nd.lo : nd.hi;
            for (int a = 0; a < 3; a++) {
                const float s = out.grid.s[a], g0 = out.grid.g0[a];
                const uint32_t qmin = q_floor(box[a][0], s, g0), qmax = q_ceil(box[a][1], s, g0);
                /* the guarantee the walk rests on, checked on the decoded values */
                if (!(ptc::q_plane(qmin, s, g0) <= box[a][0]) || !(ptc::q_plane(qmax, s, g0) >= box[a][1])) ok = false;
                r.w[3 * c + a] = qmin | (qmax << 16);
            }
        }
        r.w[6] = position(nd.lo_ref) | ((nd.axis & 1u) ? PT_Q_AXIS_BIT : 0u);
        r.w[7] = position(nd.hi_ref) | ((nd.axis & 2u) ? PT_Q_AXIS_BIT : 0u);
    }
    if (!ok) { out.rec.clear(); return; }
    out.root = position(f->root_ref);
}

} // namespace ptq

#endif
