/*
 * traverse_core.h -- per-lane ray/box, ray/triangle and tree-walk logic of the GPU ray caster.
 *
 * Everything here is written from one lane's point of view and uses no wave intrinsics, so
 * the same source is compiled by hipcc for the kernels and by g++ for a lane-level emulation
 * that the CPU tests run against the oracle (tests/native/traverse_emul.cpp).
 *
 * Arithmetic contract: strict IEEE binary32, no contraction (-ffp-contract=off), correctly
 * rounded division and square root.  Operation order follows the reference line by line:
 *   box test       AABB::does_ray_intersect      reference utils.cpp:59-83
 *   triangle test  Triangle::ray_test            reference utils.cpp:138-163
 *   tree walk      kdTreeNode::ray_test          reference kdtree.cpp:271-396
 * min/max are the reference's select forms (utils.h:29-35), not fminf/fmaxf, because their
 * NaN behaviour (axis-parallel rays grazing a box face) depends on operand order.
 */
#ifndef PT_TRAVERSE_CORE_H
#define PT_TRAVERSE_CORE_H

#include <stdint.h>
#include "pt_hip.h"

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace ptc {

#define PT_REF_NONE   0xffffffffu
#define PT_REF_FRAME  0x40000000u   /* stack-only tag: entry is the head of a 3-dword far-side frame */

/* +inf: what the reference's FLOAT_INF (1e100) becomes when narrowed to float (utils.h:10). */
PT_HD float pt_inf() { return __builtin_huge_valf(); }

PT_HD float sel_min(float x, float y) { return x < y ? x : y; }
PT_HD float sel_max(float x, float y) { return x < y ? y : x; }

PT_HD uint32_t f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
PT_HD float u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

struct Ray9 {
    float ox, oy, oz;
    float dx, dy, dz;
    float rx, ry, rz; /* CastingRay::recip_deltas, reference utils.cpp:29-33 */
};

/* 1.0/d formed in double and narrowed equals the correctly rounded binary32 quotient
 * (double carries more than 2*24+2 bits), so a float division reproduces it. */
PT_HD void ray_set_recip(Ray9& r) {
    r.rx = 1.0f / r.dx;
    r.ry = 1.0f / r.dy;
    r.rz = 1.0f / r.dz;
}

/* Component k of (a, b, c), on values: a conditional expression over struct members would be
 * an lvalue select, i.e. a dynamically addressed load that forces the struct into scratch. */
PT_HD float sel3(uint32_t k, float a, float b, float c) {
    float r = c;
    r = (k == 1u) ? b : r;
    r = (k == 0u) ? a : r;
    return r;
}
PT_HD float ray_origin_axis(const Ray9& r, uint32_t a) { return sel3(a, r.ox, r.oy, r.oz); }
PT_HD float ray_dir_axis(const Ray9& r, uint32_t a) { return sel3(a, r.dx, r.dy, r.dz); }

/* A pair of floats: {minimum, maximum} of a box along one axis. */
typedef float pt_f2 __attribute__((vector_size(8)));

/* A box as three {minimum, maximum} pairs. */
struct Box2 { pt_f2 x, y, z; };

/* (bounds - origin) * reciprocal for both bounds of each axis (utils.cpp:64-69). */
PT_HD void slab_params(const Box2& b, const Ray9& r, pt_f2& tx, pt_f2& ty, pt_f2& tz) {
    /* written per element: packed-f32 arithmetic on these pairs was measured no faster on gfx950 */
    tx = pt_f2{(b.x[0] - r.ox) * r.rx, (b.x[1] - r.ox) * r.rx};
    ty = pt_f2{(b.y[0] - r.oy) * r.ry, (b.y[1] - r.oy) * r.ry};
    tz = pt_f2{(b.z[0] - r.oz) * r.rz, (b.z[1] - r.oz) * r.rz};
}

PT_HD bool box_hit(const Box2& b, const Ray9& r) {
    pt_f2 tx, ty, tz;
    slab_params(b, r, tx, ty, tz);
    float enx = sel_min(tx[0], tx[1]), eny = sel_min(ty[0], ty[1]), enz = sel_min(tz[0], tz[1]);
    float exx = sel_max(tx[0], tx[1]), exy = sel_max(ty[0], ty[1]), exz = sel_max(tz[0], tz[1]);
    float t_start = sel_max(enx, sel_max(eny, enz));
    float t_end = sel_min(exx, sel_min(exy, exz));
    if (t_end < 0.0f) return false;
    if (t_start > t_end) return false;
    return true;
}

/* Same test with hardware min/max (v_min_f32, v_max3_f32 ...).  IEEE minNum/maxNum differ from
 * the select forms above only when an operand is NaN, or in the sign of a zero result.  A NaN
 * needs 0*inf or inf-inf, which cannot occur when the ray's reciprocals are finite and
 * non-zero and origin and box coordinates are finite and of moderate size (lane_is_tame);
 * the sign of zero never changes the two comparisons below.  So for tame rays both forms give
 * the same answer, and this one is about a third cheaper. */
PT_HD bool box_hit_tame(const Box2& b, const Ray9& r) {
    pt_f2 tx, ty, tz;
    slab_params(b, r, tx, ty, tz);
    float enx = __builtin_fminf(tx[0], tx[1]), eny = __builtin_fminf(ty[0], ty[1]), enz = __builtin_fminf(tz[0], tz[1]);
    float exx = __builtin_fmaxf(tx[0], tx[1]), exy = __builtin_fmaxf(ty[0], ty[1]), exz = __builtin_fmaxf(tz[0], tz[1]);
    /* !(t_end < 0) && !(t_start > t_end), folded: without NaNs that is max(t_start, 0) <= t_end */
    float t_start0 = __builtin_fmaxf(__builtin_fmaxf(enx, eny), __builtin_fmaxf(enz, 0.0f));
    float t_end = __builtin_fminf(exx, __builtin_fminf(exy, exz));
    return t_start0 <= t_end;
}

/* A ray is tame when no box test on it can produce a NaN: finite, non-zero reciprocals
 * (no axis-parallel or infinite direction component) and a finite origin of moderate size.
 * Box coordinates are bounded by the root box, which the scene checks once. */
PT_HD bool ray_is_tame(const Ray9& r) {
    const float big = 1.0e30f;
    bool ok = __builtin_fabsf(r.rx) < big && __builtin_fabsf(r.ry) < big && __builtin_fabsf(r.rz) < big;
    ok = ok && r.rx != 0.0f && r.ry != 0.0f && r.rz != 0.0f;
    ok = ok && __builtin_fabsf(r.ox) < big && __builtin_fabsf(r.oy) < big && __builtin_fabsf(r.oz) < big;
    return ok;
}

/* The reference compares det and t with the DOUBLE literal 1e-8.  (float)1e-8 is the largest
 * binary32 value below 1e-8, hence for a float x:  x < 1e-8  <=>  x <= EPS_F,
 * x > -1e-8 <=> x >= -EPS_F,  x <= 1e-8 <=> x <= EPS_F. */
#define PT_EPS_F 9.99999993922529029e-9f

struct TriData {
    float p0x, p0y, p0z;
    float e1x, e1y, e1z;
    float e2x, e2y, e2z;
    uint32_t mesh_index;
    uint32_t leaf_remaining; /* triangles that follow this one in its leaf (0 = last) */
};

PT_HD bool tri_hit(const TriData& tr, const Ray9& r, float& t_out, float& u_out, float& v_out) {
    /* P = direction x edge02 */
    float px = r.dy * tr.e2z - r.dz * tr.e2y;
    float py = r.dz * tr.e2x - r.dx * tr.e2z;
    float pz = r.dx * tr.e2y - r.dy * tr.e2x;
    float det = tr.e1x * px + (tr.e1y * py + tr.e1z * pz);
    if (det >= -PT_EPS_F && det <= PT_EPS_F) return false;
    float inv_det = 1.0f / det;
    float tx = r.ox - tr.p0x, ty = r.oy - tr.p0y, tz = r.oz - tr.p0z;
    float u = (tx * px + (ty * py + tz * pz)) * inv_det;
    if (u < 0.0f || u > 1.0f) return false;
    /* Q = T x edge01 */
    float qx = ty * tr.e1z - tz * tr.e1y;
    float qy = tz * tr.e1x - tx * tr.e1z;
    float qz = tx * tr.e1y - ty * tr.e1x;
    float v = (r.dx * qx + (r.dy * qy + r.dz * qz)) * inv_det;
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = (tr.e2x * qx + (tr.e2y * qy + tr.e2z * qz)) * inv_det;
    if (t <= PT_EPS_F) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

struct NodeData {
    pt_f2 lo[3], hi[3];          /* children's boxes, {minimum, maximum} per axis */
    uint32_t lo_ref, hi_ref, axis;
};

struct Hit {
    float t, u, v;
    uint32_t tri;  /* position in the leaf-ordered triangle arrays */
    uint32_t mesh; /* mesh index of that triangle */
};

/* The root's own box (every other node's box is stored in its parent). */
struct RootBox { float mnx, mny, mnz, mxx, mxy, mxz; };

/* Every box of the tree lies inside the root box, so this bounds all box coordinates. */
PT_HD bool root_is_tame(const RootBox& b) {
    const float big = 1.0e30f;
    return __builtin_fabsf(b.mnx) < big && __builtin_fabsf(b.mny) < big && __builtin_fabsf(b.mnz) < big &&
           __builtin_fabsf(b.mxx) < big && __builtin_fabsf(b.mxy) < big && __builtin_fabsf(b.mxz) < big;
}

PT_HD uint32_t ref_is_leaf(uint32_t ref) { return ref & PT_REF_LEAF; }
PT_HD uint32_t ref_leaf_start(uint32_t ref) { return ref & PT_REF_START_MASK; }
PT_HD uint32_t ref_leaf_count(uint32_t ref) { return (ref >> PT_REF_COUNT_SHIFT) & PT_REF_COUNT_MASK; }

/*
 * Closest hit, one lane, as a state machine.  Reproduces the recursion of
 * kdTreeNode::ray_test with an explicit stack.  The recursion combines sub-results with strict
 * '<' and "first visited wins ties"; that is the minimum over (t, visit order) of all triangle
 * hits in visited leaves, so a single running `best` updated with strict '<' in visit order
 * gives the same answer.  The one place where a SUB-result matters is the far-side test
 * `hit_parameter > t_far` (:367), which uses the near subtree's own closest t.  `local` tracks
 * the closest t found since the innermost pending far-side frame was pushed; a frame saves the
 * enclosing value and restores min(saved, local) when popped.
 *
 * Stack entries, pushed in this order:
 *   plain child reference                         1 dword  (the "overlaps both" case)
 *   frame {[saved local], t_far, far_ref|FRAME}   2 or 3 dwords
 * The saved value is +inf until the enclosing scope has found a hit, which is the common case;
 * it is then not stored, and one bit per pending frame (`frame_bits`, a bit stack: frames nest
 * at most PT_MAX_TREE_DEPTH deep) records whether it was.  At most one entry group per interior
 * ancestor is pending, so 3 * PT_MAX_TREE_DEPTH dwords always suffice.
 *
 * ANYHIT: stop at the first triangle hit; the result is only hit/no-hit.  The far side is then
 * visited iff the near side found nothing, which needs no frame (a plain deferred reference
 * does it), so 1 * PT_MAX_TREE_DEPTH dwords suffice.
 *
 * A lane's position is one word, `cur`: PT_REF_NONE, an interior node index, or a leaf
 * reference whose start field walks through the leaf's triangles (each triangle record says
 * how many follow it, so the count field of the reference is not needed).  A lane is always in
 * exactly one of these situations:
 *   leaf pending   (cur is a leaf reference)  -> lane_tri() with the triangle lane_leaf_pos()
 *   interior node  (cur is a node index)      -> lane_node() with that node
 *   neither, stack not empty                  -> lane_pop()
 *   neither, stack empty                      -> finished
 * Stack is private to the lane and supplies
 *   push(int& sp, v)                         store v at entry sp, advance sp
 *   push_frame(int& sp, keep, a, b, c)       push a (only if keep), then b, then c
 *   peek1(sp)                                entry sp-1
 *   peek3(sp, e, w1, w2)                     entries sp-1, sp-2, sp-3 (any value below entry 0)
 */
struct Lane {
    Ray9 r;
    Hit best;
    float local;
    uint32_t cur;
    uint32_t frame_bits;
    int sp;
    bool found;
    bool tame;                   /* ray_is_tame(r) and the scene's boxes are of moderate size */
};

/* PT_REF_NONE is -1 as a signed word, leaf references are the other negative words. */
PT_HD bool lane_has_leaf(const Lane& L) { return (int32_t)L.cur < -1; }
PT_HD bool lane_has_node(const Lane& L) { return (int32_t)L.cur >= 0; }
PT_HD uint32_t lane_leaf_pos(const Lane& L) { return L.cur & PT_REF_START_MASK; }
PT_HD bool lane_finished(const Lane& L) { return L.cur == PT_REF_NONE && L.sp == 0; }

PT_HD void lane_enter(Lane& L, uint32_t ref) { L.cur = ref; }

PT_HD void lane_start(Lane& L, uint32_t root_ref, const RootBox& root, bool scene_tame) {
    const float inf = pt_inf();
    L.tame = scene_tame && ray_is_tame(L.r);
    L.best.t = inf; L.best.u = 0.0f; L.best.v = 0.0f; L.best.tri = 0xffffffffu; L.best.mesh = 0xffffffffu;
    L.local = inf;
    L.frame_bits = 0; L.sp = 0; L.found = false;
    Box2 rb;
    rb.x = pt_f2{root.mnx, root.mxx}; rb.y = pt_f2{root.mny, root.mxy}; rb.z = pt_f2{root.mnz, root.mxz};
    lane_enter(L, box_hit(rb, L.r) ? root_ref : PT_REF_NONE);
}

/* One pop, given the top entries of the lane's stack: e = entry sp-1, w1 = entry sp-2, w2 = entry
 * sp-3 (w1 and w2 are used only by the closest-hit walk, and only where a frame has them). */
template <bool ANYHIT>
PT_HD void lane_pop_entries(Lane& L, uint32_t e, uint32_t w1, uint32_t w2) {
    if (ANYHIT) {
        L.sp -= 1;
        lane_enter(L, e);
        return;
    }
    const float inf = pt_inf();
    const bool frame = (e & PT_REF_FRAME) != 0;
    const bool kept = frame && (L.frame_bits & 1u) != 0;
    const float t_far = u2f(w1);
    const float saved = kept ? u2f(w2) : inf;
    const float near_local = L.local;
    /* near side missed (:393-394) or its closest hit lies beyond the far slab (:367) */
    const bool visit = !(near_local < inf) || (near_local > t_far);
    const float merged = saved < near_local ? saved : near_local;
    L.sp -= frame ? (kept ? 3 : 2) : 1;
    L.frame_bits = frame ? (L.frame_bits >> 1) : L.frame_bits;
    L.local = frame ? merged : near_local;
    lane_enter(L, (frame && !visit) ? PT_REF_NONE : (e & ~PT_REF_FRAME));
}

template <bool ANYHIT, class Stack>
PT_HD void lane_pop(Lane& L, Stack& st) {
    /* The top three entries are read together (a frame is at most three dwords) so that the
     * reads do not wait on one another; how many are consumed is decided afterwards. */
    uint32_t e, w1 = 0u, w2 = 0u;
    if (ANYHIT) e = st.peek1(L.sp);
    else st.peek3(L.sp, e, w1, w2);
    lane_pop_entries<ANYHIT>(L, e, w1, w2);
}

/* TAME selects the cheaper box test; it may be used only for lanes with L.tame set. */
template <bool ANYHIT, bool TAME, class Stack>
PT_HD void lane_node(Lane& L, const NodeData& n, Stack& st) {
    /* What visiting one interior node decides (kdtree.cpp:307-316, 324-394), with both
     * children's box tests (the first statement of each child's own ray_test, :273) folded in;
     * a child whose box is missed contributes nothing, exactly as its ray_test returning false.
     * Arranged so that the only data-dependent branch is the one that pushes: `first` is the
     * child entered first (the low child when the origin
     * overlaps both, :324-342, otherwise the near child, :315-316), `second` the other one. */
    Box2 lo, hi;
    lo.x = n.lo[0]; lo.y = n.lo[1]; lo.z = n.lo[2];
    hi.x = n.hi[0]; hi.y = n.hi[1]; hi.z = n.hi[2];
    const bool hit_lo = TAME ? box_hit_tame(lo, L.r) : box_hit(lo, L.r);
    const bool hit_hi = TAME ? box_hit_tame(hi, L.r) : box_hit(hi, L.r);
    const float origin = ray_origin_axis(L.r, n.axis);
    const float hi_min_a = sel3(n.axis, n.hi[0][0], n.hi[1][0], n.hi[2][0]);
    const float lo_max_a = sel3(n.axis, n.lo[0][1], n.lo[1][1], n.lo[2][1]);
    const bool over_hi = hi_min_a < origin;  /* :310 */
    const bool over_lo = origin <= lo_max_a; /* :311 */
    const bool first_is_lo = !over_hi || over_lo;
    /* Where the lane goes next: the child whose box is hit if there is only one, the first of the
     * two if both are.  Written as selects under plain conditions so that these stay wave masks
     * (a select between two outcomes would turn them into 0/1 registers). */
    uint32_t next = hit_lo ? n.lo_ref : PT_REF_NONE;
    next = hit_hi ? n.hi_ref : next;
    next = (hit_lo && hit_hi && first_is_lo) ? n.lo_ref : next;
    if (hit_lo && hit_hi) {
        const uint32_t second = first_is_lo ? n.hi_ref : n.lo_ref;
        if (ANYHIT || (over_hi && over_lo)) {
            /* entered afterwards whatever the first child finds; any-hit stops at the first hit,
             * so there "afterwards" already means "the first child found nothing" */
            st.push(L.sp, second);
        } else {
            const float inf = pt_inf();
            const float e = first_is_lo ? hi_min_a : lo_max_a;              /* :362 */
            const float t_far = (e - origin) / ray_dir_axis(L.r, n.axis);   /* :364, true division */
            const bool keep = L.local < inf;
            st.push_frame(L.sp, keep, f2u(L.local), f2u(t_far), second | PT_REF_FRAME);
            L.frame_bits = (L.frame_bits << 1) | (keep ? 1u : 0u);
            L.local = inf;
        }
    }
    lane_enter(L, next);
}

/*
 * Quantised interior nodes for the any-hit walks ("q-nodes").
 *
 * The answer of an any-hit walk is order-free: it is "does the ray hit a triangle of a leaf whose
 * box, and whose ancestors' boxes, the ray hits".  Node boxes are unions of their triangles' boxes
 * (kdtree.cpp:60-61), so along every axis an ancestor's bounds enclose its descendants' exactly, in
 * binary32.  The slab test's per-axis terms fl(fl(b - o) * r) are monotone in the bound b, hence for
 * a ray whose box tests produce no NaN (a "tame" ray) a hit of a box implies a hit of every box
 * that encloses it -- in the test's own binary32 arithmetic, not only in exact arithmetic.  So
 *   - the leaves whose triangles the reference tests are exactly the leaves whose OWN box is hit, and
 *   - an interior node may be tested against ANY box that encloses its true one: that can only add
 *     visits to subtrees in which no leaf box is hit, never remove a leaf.
 * A q-node stores the two child boxes on a 16-bit grid over the root box, rounded outwards (the
 * builder checks the decoded binary32 values, computed as the kernel computes them, against the true
 * bounds), in 32 bytes instead of 64.  Only nodes whose children are both interior have a q-node;
 * a node with a leaf child keeps its exact record, so every leaf is entered through the exact test
 * of its own box.  Below such a node the walk stays on exact records.
 *
 * A lane's position `cur` in this scheme: PT_REF_NONE, a leaf reference, a q-node index (below
 * PT_Q_EXACT_BASE) or PT_Q_EXACT_BASE + an exact node's index.
 */
#define PT_Q_EXACT_BASE 0x02000000u     /* node positions from here on mean "exact record number (position - base)" */
#define PT_Q_AXIS_BIT   0x40000000u     /* in a q-node's two references: bit 0 and bit 1 of the split axis */

struct QGrid { float g0[3], s[3]; };    /* plane of grid step q on axis a: (float)q * s[a] + g0[a] */

/* Product and sum rounded separately (no contraction), on the host as on the device: each then takes
 * one scalar operand, which a fused multiply-add with two scalars could not. */
PT_HD float q_plane(uint32_t q, float s, float g0) { const float p = (float)q * s; return p + g0; }

/* The eight words of a q-node: six {minimum | maximum << 16} pairs (low child x, y, z, high child
 * x, y, z), then the two child references. */
PT_HD void qnode_decode(const uint32_t w[8], const QGrid& g, NodeData& n) {
    n.lo[0] = pt_f2{q_plane(w[0] & 0xffffu, g.s[0], g.g0[0]), q_plane(w[0] >> 16, g.s[0], g.g0[0])};
    n.lo[1] = pt_f2{q_plane(w[1] & 0xffffu, g.s[1], g.g0[1]), q_plane(w[1] >> 16, g.s[1], g.g0[1])};
    n.lo[2] = pt_f2{q_plane(w[2] & 0xffffu, g.s[2], g.g0[2]), q_plane(w[2] >> 16, g.s[2], g.g0[2])};
    n.hi[0] = pt_f2{q_plane(w[3] & 0xffffu, g.s[0], g.g0[0]), q_plane(w[3] >> 16, g.s[0], g.g0[0])};
    n.hi[1] = pt_f2{q_plane(w[4] & 0xffffu, g.s[1], g.g0[1]), q_plane(w[4] >> 16, g.s[1], g.g0[1])};
    n.hi[2] = pt_f2{q_plane(w[5] & 0xffffu, g.s[2], g.g0[2]), q_plane(w[5] >> 16, g.s[2], g.g0[2])};
    n.lo_ref = w[6] & ~PT_Q_AXIS_BIT;
    n.hi_ref = w[7] & ~PT_Q_AXIS_BIT;
    n.axis = ((w[6] >> 30) & 1u) | ((w[7] >> 29) & 2u);
}

/* An exact record met by a walk that uses q-nodes: its interior children are exact records too. */
PT_HD void exact_refs_to_q_space(NodeData& n) {
    n.lo_ref = ref_is_leaf(n.lo_ref) ? n.lo_ref : n.lo_ref + PT_Q_EXACT_BASE;
    n.hi_ref = ref_is_leaf(n.hi_ref) ? n.hi_ref : n.hi_ref + PT_Q_EXACT_BASE;
}

/* One triangle of the pending leaf, in stored order, strict '<' (:281-293). */
template <bool ANYHIT>
PT_HD void lane_tri(Lane& L, const TriData& tr) {
    float t, u, v;
    if (tri_hit(tr, L.r, t, u, v)) {
        if (t < L.best.t) { L.best.t = t; L.best.u = u; L.best.v = v; L.best.tri = lane_leaf_pos(L); L.best.mesh = tr.mesh_index; L.found = true; }
        if (t < L.local) L.local = t;
    }
    L.cur = tr.leaf_remaining ? L.cur + 1u : PT_REF_NONE;
    if (ANYHIT && L.found) { L.cur = PT_REF_NONE; L.sp = 0; } /* nothing left to do */
}

/* Runs one lane to completion (the CPU emulation's driver; the kernels interleave the same
 * steps across the lanes of a wave).  Mem supplies load_node(index, NodeData&) and
 * load_tri(position, TriData&). */
template <bool ANYHIT, class Mem, class Stack>
PT_HD bool trace_lane(const Mem& mem, uint32_t root_ref, const RootBox& root, const Ray9& r, Stack& st, Hit& best) {
    Lane L;
    L.r = r;
    lane_start(L, root_ref, root, root_is_tame(root));
    while (!lane_finished(L)) {
        if (lane_has_leaf(L)) { TriData tr; mem.load_tri(lane_leaf_pos(L), tr); lane_tri<ANYHIT>(L, tr); }
        else if (lane_has_node(L)) {
            NodeData n;
            mem.load_node(L.cur, n);
            if (L.tame) lane_node<ANYHIT, true>(L, n, st); else lane_node<ANYHIT, false>(L, n, st);
        }
        else lane_pop<ANYHIT>(L, st);
    }
    best = L.best;
    return L.found;
}

} // namespace ptc

#endif
