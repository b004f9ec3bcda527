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

PT_HD bool box_hit(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, const Ray9& r) {
    float t0x = (mnx - r.ox) * r.rx, t0y = (mny - r.oy) * r.ry, t0z = (mnz - r.oz) * r.rz;
    float t1x = (mxx - r.ox) * r.rx, t1y = (mxy - r.oy) * r.ry, t1z = (mxz - r.oz) * r.rz;
    float enx = sel_min(t0x, t1x), eny = sel_min(t0y, t1y), enz = sel_min(t0z, t1z);
    float exx = sel_max(t0x, t1x), exy = sel_max(t0y, t1y), exz = sel_max(t0z, t1z);
    float t_start = sel_max(enx, sel_max(eny, enz));
    float t_end = sel_min(exx, sel_min(exy, exz));
    if (t_end < 0.0f) return false;
    if (t_start > t_end) return false;
    return true;
}

/* The reference compares det and t with the DOUBLE literal 1e-8.  (float)1e-8 is the largest
 * binary32 value below 1e-8, hence for a float x:  x < 1e-8  <=>  x <= EPS_F,
 * x > -1e-8 <=> x >= -EPS_F,  x <= 1e-8 <=> x <= EPS_F. */
#define PT_EPS_F 9.99999993922529029e-9f

struct TriData {
    float p0x, p0y, p0z;
    float e1x, e1y, e1z;
    float e2x, e2y, e2z;
    uint32_t mesh_index, leaf_count;
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
    float lo_min[3], lo_max[3], hi_min[3], hi_max[3];
    uint32_t lo_ref, hi_ref, axis;
};

PT_HD float pick3(const float* a, uint32_t k) { return sel3(k, a[0], a[1], a[2]); }

/* What visiting one interior node decides (kdtree.cpp:307-316, 324-394), with both children's
 * box tests (the first statement of each child's own ray_test, :273) already folded in:
 *   next        child to enter now, or PT_REF_NONE
 *   deferred    child to enter afterwards unconditionally (the "overlaps both" case), or NONE
 *   far         child whose visit depends on the near side's result, or NONE; t_far is the ray
 *               parameter of its slab along the split axis (:362-364)
 * A child whose box is missed contributes nothing, exactly as its ray_test returning false. */
struct StepOut {
    uint32_t next, deferred, far;
    float t_far;
};

PT_HD StepOut interior_step(const NodeData& n, const Ray9& r) {
    StepOut o;
    o.next = PT_REF_NONE; o.deferred = PT_REF_NONE; o.far = PT_REF_NONE; o.t_far = 0.0f;
    bool hit_lo = box_hit(n.lo_min[0], n.lo_min[1], n.lo_min[2], n.lo_max[0], n.lo_max[1], n.lo_max[2], r);
    bool hit_hi = box_hit(n.hi_min[0], n.hi_min[1], n.hi_min[2], n.hi_max[0], n.hi_max[1], n.hi_max[2], r);
    float origin = ray_origin_axis(r, n.axis);
    float hi_min_a = pick3(n.hi_min, n.axis), lo_max_a = pick3(n.lo_max, n.axis);
    bool over_hi = hi_min_a < origin;  /* :310 */
    bool over_lo = origin <= lo_max_a; /* :311 */
    if (over_hi && over_lo) {          /* :324-342: low first, then high, both unconditionally */
        if (hit_lo) { o.next = n.lo_ref; if (hit_hi) o.deferred = n.hi_ref; }
        else if (hit_hi) o.next = n.hi_ref;
        return o;
    }
    bool near_is_lo = !over_hi;        /* :315-316 */
    uint32_t near_ref = near_is_lo ? n.lo_ref : n.hi_ref;
    uint32_t far_ref = near_is_lo ? n.hi_ref : n.lo_ref;
    bool hit_near = near_is_lo ? hit_lo : hit_hi;
    bool hit_far = near_is_lo ? hit_hi : hit_lo;
    if (hit_near) {
        o.next = near_ref;
        if (hit_far) {
            float e = near_is_lo ? hi_min_a : lo_max_a;      /* :362 */
            o.far = far_ref;
            o.t_far = (e - origin) / ray_dir_axis(r, n.axis); /* :364, true division */
        }
    } else if (hit_far) {
        o.next = far_ref;              /* :393-394 */
    }
    return o;
}

struct Hit {
    float t, u, v;
    uint32_t tri;  /* position in the leaf-ordered triangle arrays */
    uint32_t mesh; /* mesh index of that triangle */
};

/* The root's own box (every other node's box is stored in its parent). */
struct RootBox { float mnx, mny, mnz, mxx, mxy, mxz; };

PT_HD uint32_t ref_is_leaf(uint32_t ref) { return ref & PT_REF_LEAF; }
PT_HD uint32_t ref_leaf_start(uint32_t ref) { return ref & PT_REF_START_MASK; }
PT_HD uint32_t ref_leaf_count(uint32_t ref) { return (ref >> PT_REF_COUNT_SHIFT) & PT_REF_COUNT_MASK; }

/*
 * Closest hit, one lane.  Reproduces the recursion of kdTreeNode::ray_test with an explicit
 * stack.  The recursion combines sub-results with strict '<' and "first visited wins ties";
 * that is the minimum over (t, visit order) of all triangle hits in visited leaves, so a
 * single running `best` updated with strict '<' in visit order gives the same answer.  The
 * one place where a SUB-result matters is the far-side test `hit_parameter > t_far`
 * (:367), which uses the near subtree's own closest t.  `local` tracks the closest t found
 * since the innermost pending far-side frame was pushed; a frame saves the enclosing value
 * and restores min(saved, local) when popped.
 *
 * Stack entries: a plain child reference (1 dword), or a frame of 3 dwords pushed as
 * {saved local, t_far, far_ref | PT_REF_FRAME}.  At most one entry group per interior
 * ancestor is pending, so 3 * PT_MAX_TREE_DEPTH dwords suffice.
 *
 * ANYHIT: stop at the first triangle hit; the result is only hit/no-hit.  The far side is then
 * visited iff the near side found nothing, which needs no frame (a plain deferred reference
 * does it), so 1 * PT_MAX_TREE_DEPTH dwords suffice.
 *
 * Mem supplies load_node(index, NodeData&) and load_tri(position, TriData&).
 * Stack supplies push(uint32_t) and pop() and is private to the lane.
 */
template <bool ANYHIT, class Mem, class Stack>
PT_HD bool trace_lane(const Mem& mem, uint32_t root_ref, const RootBox& root, const Ray9& r, Stack& st, Hit& best) {
    const float inf = pt_inf();
    best.t = inf; best.u = 0.0f; best.v = 0.0f; best.tri = 0xffffffffu; best.mesh = 0xffffffffu;
    bool found = false;
    float local = inf;
    int sp = 0;
    uint32_t cur = box_hit(root.mnx, root.mny, root.mnz, root.mxx, root.mxy, root.mxz, r) ? root_ref : PT_REF_NONE;
    bool done = false;
    while (!done) {
        /* descend until this lane holds a leaf (or has nothing left) */
        while (!done && !(cur != PT_REF_NONE && ref_is_leaf(cur))) {
            if (cur == PT_REF_NONE) {
                if (sp == 0) { done = true; break; }
                uint32_t e = st.pop(); sp--;
                if (!ANYHIT && (e & PT_REF_FRAME)) {
                    float t_far = u2f(st.pop());
                    float saved = u2f(st.pop());
                    sp -= 2;
                    float near_local = local;
                    /* near side missed (:393-394) or its closest hit lies beyond the far slab (:367) */
                    bool visit = !(near_local < inf) || (near_local > t_far);
                    local = saved < near_local ? saved : near_local;
                    cur = visit ? (e & ~PT_REF_FRAME) : PT_REF_NONE;
                } else {
                    cur = e;
                }
                continue;
            }
            NodeData n;
            mem.load_node(cur, n);
            StepOut o = interior_step(n, r);
            if (o.deferred != PT_REF_NONE) { st.push(o.deferred); sp++; }
            if (o.far != PT_REF_NONE) {
                if (ANYHIT) { st.push(o.far); sp++; }
                else {
                    st.push(f2u(local)); st.push(f2u(o.t_far)); st.push(o.far | PT_REF_FRAME); sp += 3;
                    local = inf;
                }
            }
            cur = o.next;
        }
        if (done) break;
        /* leaf: every triangle, in stored order, strict '<' (:281-293) */
        uint32_t start = ref_leaf_start(cur);
        uint32_t count = ref_leaf_count(cur);
        TriData tr;
        mem.load_tri(start, tr);
        if (count == 0) count = tr.leaf_count;
        for (uint32_t k = 0;;) {
            float t, u, v;
            if (tri_hit(tr, r, t, u, v)) {
                if (t < best.t) { best.t = t; best.u = u; best.v = v; best.tri = start + k; best.mesh = tr.mesh_index; found = true; }
                if (t < local) local = t;
            }
            if (++k >= count) break;
            mem.load_tri(start + k, tr);
        }
        cur = PT_REF_NONE;
        if (ANYHIT && found) done = true;
    }
    return found;
}

} // namespace ptc

#endif
