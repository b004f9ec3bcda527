/*
 * wave_sim.cpp -- offline model of how one 64-lane wave schedules the lane state machine.
 *
 * STUDY TOOL (CPU).  Drives path_tracing_amd/csrc/traverse_core.h for 64 lanes with the same
 * refill rule as the trace kernel and charges every round an instruction cost per executed
 * section, so that scheduling policies can be compared before touching the kernel:
 *   policy 0  every round runs the triangle section, up to max_pops pop sections and the node
 *             section (each only if some lane needs it)
 *   policy 1  a round runs EITHER the triangle section (when at least tri_threshold lanes wait
 *             on a leaf, or no lane wants a node) OR the pop+node sections
 *   policy 5  every lane holds TWO rays; each section of a round serves, per lane, one of its
 *             rays that needs that section (a ray takes at most one step per round), so a
 *             section finds work in more lanes (wave_sim_two)
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -Iinclude -Ipath_tracing_amd/csrc -shared -fPIC
 */
#include <cstdint>
#include <cstring>

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
    void push(int& sp, uint32_t v) { data[sp++] = v; }
    uint32_t peek(int i) const { return i >= 0 ? data[i] : 0u; }
    uint32_t peek1(int sp) const { return peek(sp - 1); }
    void peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const { e = peek(sp - 1); w1 = peek(sp - 2); w2 = peek(sp - 3); }
    void push_frame(int& sp, bool keep, uint32_t a, uint32_t b, uint32_t c) { if (keep) push(sp, a); push(sp, b); push(sp, c); }
};
}

extern "C" {

struct SimParams;
struct SimStats;
void wave_sim_two(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const SimParams* p, SimStats* out);

struct SimParams {
    int policy, refill_min, tri_threshold, max_pops, tris_per_round, any_hit;
    double c_round, c_tri, c_pop, c_node, c_start;
};
struct SimStats {
    double cost;
    uint64_t rounds, tri_sections, pop_sections, node_sections, start_sections;
    uint64_t lane_tri, lane_pop, lane_node, lane_pop_culled;
};

void wave_sim(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const SimParams* p, SimStats* out) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    ptc::RootBox root;
    root.mnx = sc->root_min[0]; root.mny = sc->root_min[1]; root.mnz = sc->root_min[2];
    root.mxx = sc->root_max[0]; root.mxy = sc->root_max[1]; root.mxz = sc->root_max[2];
    const bool scene_tame = ptc::root_is_tame(root);
    static ptc::Lane L[64];
    static CpuStack st[64];
    bool active[64];
    for (int i = 0; i < 64; i++) active[i] = false;
    uint64_t next = 0;
    SimStats s;
    memset(&s, 0, sizeof s);
    const bool any = p->any_hit != 0;
    while (true) {
        int idle = 0;
        for (int i = 0; i < 64; i++) idle += !active[i];
        bool starting = false;
        int start_list[64], n_start = 0;
        if (idle && next < n && (idle >= p->refill_min || idle == 64)) {
            for (int i = 0; i < 64 && next < n; i++) if (!active[i]) start_list[n_start++] = i, next++;
            starting = n_start > 0;
        }
        if (next >= n && idle == 64 && !starting) break;
        s.rounds++;
        s.cost += p->c_round;
        int nt = 0, nn = 0;
        for (int i = 0; i < 64; i++) if (active[i]) { if (ptc::lane_has_leaf(L[i])) nt++; else nn++; }
        bool run_tri = nt > 0, run_node = nn > 0;
        if (p->policy == 1 && nt > 0 && nn > 0) {
            if (nt >= p->tri_threshold) run_node = false; else run_tri = false;
        }
        auto tri_section = [&](bool* did) {
            bool some = false;
            for (int i = 0; i < 64; i++) if (active[i] && ptc::lane_has_leaf(L[i])) {
                ptc::TriData tr; mem.load_tri(ptc::lane_leaf_pos(L[i]), tr);
                if (any) ptc::lane_tri<true>(L[i], tr); else ptc::lane_tri<false>(L[i], tr);
                s.lane_tri++; some = true; if (did) did[i] = true;
            }
            if (some) { s.tri_sections++; s.cost += p->c_tri; }
        };
        auto pop_sections = [&](const bool* skip) {
            for (int k = 0; k < p->max_pops; k++) {
                bool some = false;
                for (int i = 0; i < 64; i++) {
                    ptc::Lane& l = L[i];
                    if (!active[i] || (skip && skip[i]) || l.cur != PT_REF_NONE || l.sp <= 0) continue;
                    if (any) ptc::lane_pop<true>(l, st[i]); else ptc::lane_pop<false>(l, st[i]);
                    s.lane_pop++; some = true;
                    if (l.cur == PT_REF_NONE) s.lane_pop_culled++;
                }
                if (some) { s.pop_sections++; s.cost += p->c_pop; }
            }
        };
        auto node_section = [&](const bool* skip) {
            bool some = false;
            for (int i = 0; i < 64; i++) {
                ptc::Lane& l = L[i];
                if (!active[i] || (skip && skip[i]) || !ptc::lane_has_node(l)) continue;
                ptc::NodeData nd; mem.load_node(l.cur, nd);
                if (any) { if (l.tame) ptc::lane_node<true, true>(l, nd, st[i]); else ptc::lane_node<true, false>(l, nd, st[i]); }
                else { if (l.tame) ptc::lane_node<false, true>(l, nd, st[i]); else ptc::lane_node<false, false>(l, nd, st[i]); }
                s.lane_node++; some = true;
            }
            if (some) { s.node_sections++; s.cost += p->c_node; }
        };
        bool did_tri[64];
        for (int i = 0; i < 64; i++) did_tri[i] = false;
        if (p->policy == 2) {        /* triangles, pops, then nodes (second load group), pops */
            tri_section(nullptr); pop_sections(nullptr); node_section(nullptr); pop_sections(nullptr);
        } else if (p->policy == 3) { /* nodes, pops, then triangles (second load group), pops */
            node_section(nullptr); pop_sections(nullptr); tri_section(nullptr); pop_sections(nullptr);
        } else if (p->policy == 4) { /* one load group; kernel order: triangle or node, then pops */
            tri_section(did_tri); node_section(did_tri); pop_sections(nullptr);
        } else {
            if (run_tri) tri_section(did_tri);
            if (run_node) { pop_sections(did_tri); node_section(did_tri); }
        }
        for (int i = 0; i < 64; i++) if (active[i] && ptc::lane_finished(L[i])) active[i] = false;
        if (starting) {
            s.start_sections++; s.cost += p->c_start;
            for (int k = 0; k < n_start; k++) {
                int i = start_list[k];
                uint64_t r = next - n_start + k;
                L[i].r.ox = o[3 * r]; L[i].r.oy = o[3 * r + 1]; L[i].r.oz = o[3 * r + 2];
                L[i].r.dx = d[3 * r]; L[i].r.dy = d[3 * r + 1]; L[i].r.dz = d[3 * r + 2];
                ptc::ray_set_recip(L[i].r);
                ptc::lane_start(L[i], sc->root_ref, root, scene_tame);
                active[i] = !ptc::lane_finished(L[i]);
            }
        }
    }
    *out = s;
}


/* Two rays per lane.  Slot k of lane i is ray state L2[k][i].  tri_threshold is reused as the
 * extra cost per section for choosing between a lane's two rays. */
void wave_sim_two(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const SimParams* p, SimStats* out) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    ptc::RootBox root;
    root.mnx = sc->root_min[0]; root.mny = sc->root_min[1]; root.mnz = sc->root_min[2];
    root.mxx = sc->root_max[0]; root.mxy = sc->root_max[1]; root.mxz = sc->root_max[2];
    const bool scene_tame = ptc::root_is_tame(root);
    static ptc::Lane L[2][64];
    static CpuStack st[2][64];
    bool active[2][64];
    for (int k = 0; k < 2; k++) for (int i = 0; i < 64; i++) active[k][i] = false;
    uint64_t next = 0;
    SimStats s;
    memset(&s, 0, sizeof s);
    const bool any = p->any_hit != 0;
    const double c_sel = p->tri_threshold;
    while (true) {
        int idle = 0;
        for (int k = 0; k < 2; k++) for (int i = 0; i < 64; i++) idle += !active[k][i];
        int start_slot[128], start_lane[128], n_start = 0;
        if (idle && next < n && (idle >= p->refill_min || idle == 128)) {
            for (int k = 0; k < 2; k++) for (int i = 0; i < 64 && next < n; i++) if (!active[k][i]) { start_slot[n_start] = k; start_lane[n_start++] = i; next++; }
        }
        if (next >= n && idle == 128 && n_start == 0) break;
        s.rounds++;
        s.cost += p->c_round;
        bool did[2][64];
        for (int k = 0; k < 2; k++) for (int i = 0; i < 64; i++) did[k][i] = false;
        {   /* triangle section */
            bool some = false;
            for (int i = 0; i < 64; i++) for (int k = 0; k < 2; k++) if (active[k][i] && ptc::lane_has_leaf(L[k][i])) {
                ptc::TriData tr; mem.load_tri(ptc::lane_leaf_pos(L[k][i]), tr);
                if (any) ptc::lane_tri<true>(L[k][i], tr); else ptc::lane_tri<false>(L[k][i], tr);
                s.lane_tri++; some = true; did[k][i] = true;
                break;
            }
            if (some) { s.tri_sections++; s.cost += p->c_tri + c_sel; }
        }
        {   /* node section */
            bool some = false;
            for (int i = 0; i < 64; i++) for (int k = 0; k < 2; k++) {
                ptc::Lane& l = L[k][i];
                if (!active[k][i] || did[k][i] || !ptc::lane_has_node(l)) continue;
                ptc::NodeData nd; mem.load_node(l.cur, nd);
                if (any) { if (l.tame) ptc::lane_node<true, true>(l, nd, st[k][i]); else ptc::lane_node<true, false>(l, nd, st[k][i]); }
                else { if (l.tame) ptc::lane_node<false, true>(l, nd, st[k][i]); else ptc::lane_node<false, false>(l, nd, st[k][i]); }
                s.lane_node++; some = true; did[k][i] = true;
                break;
            }
            if (some) { s.node_sections++; s.cost += p->c_node + c_sel; }
        }
        for (int r = 0; r < p->max_pops; r++) {   /* pop sections: one ray per lane each */
            bool some = false;
            for (int i = 0; i < 64; i++) for (int k = 0; k < 2; k++) {
                ptc::Lane& l = L[k][i];
                if (!active[k][i] || l.cur != PT_REF_NONE || l.sp <= 0) continue;
                if (any) ptc::lane_pop<true>(l, st[k][i]); else ptc::lane_pop<false>(l, st[k][i]);
                s.lane_pop++; some = true;
                if (l.cur == PT_REF_NONE) s.lane_pop_culled++;
                break;
            }
            if (some) { s.pop_sections++; s.cost += p->c_pop + c_sel; }
        }
        for (int k = 0; k < 2; k++) for (int i = 0; i < 64; i++) if (active[k][i] && ptc::lane_finished(L[k][i])) active[k][i] = false;
        if (n_start) {
            s.start_sections++; s.cost += p->c_start;
            for (int q = 0; q < n_start; q++) {
                int i = start_lane[q], k = start_slot[q];
                uint64_t r = next - n_start + q;
                L[k][i].r.ox = o[3 * r]; L[k][i].r.oy = o[3 * r + 1]; L[k][i].r.oz = o[3 * r + 2];
                L[k][i].r.dx = d[3 * r]; L[k][i].r.dy = d[3 * r + 1]; L[k][i].r.dz = d[3 * r + 2];
                ptc::ray_set_recip(L[k][i].r);
                ptc::lane_start(L[k][i], sc->root_ref, root, scene_tame);
                active[k][i] = !ptc::lane_finished(L[k][i]);
            }
        }
    }
    *out = s;
}

}

/* Packet model: rays [64k, 64k+64) form a packet that one wave would walk TOGETHER, visiting a
 * node or testing a triangle once if any of its rays does.  Every ray is walked alone with the lane
 * state machine; the packet's node visits and triangle tests are the unions over its rays.
 * out[0] += sum over rays of node visits, out[1] += sum over rays of triangle tests,
 * out[2] += sum over packets of distinct nodes, out[3] += sum over packets of distinct triangles,
 * out[4] += packets. */
#include <vector>
#include <algorithm>
extern "C" void packet_model(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, uint64_t* out) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    ptc::RootBox root;
    root.mnx = sc->root_min[0]; root.mny = sc->root_min[1]; root.mnz = sc->root_min[2];
    root.mxx = sc->root_max[0]; root.mxy = sc->root_max[1]; root.mxz = sc->root_max[2];
    const bool scene_tame = ptc::root_is_tame(root);
    std::vector<uint32_t> nodes, tris;
    for (uint64_t base = 0; base + 64 <= n; base += 64) {
        nodes.clear(); tris.clear();
        for (uint64_t r = base; r < base + 64; r++) {
            ptc::Lane l; CpuStack st;
            l.r.ox = o[3 * r]; l.r.oy = o[3 * r + 1]; l.r.oz = o[3 * r + 2];
            l.r.dx = d[3 * r]; l.r.dy = d[3 * r + 1]; l.r.dz = d[3 * r + 2];
            ptc::ray_set_recip(l.r);
            ptc::lane_start(l, sc->root_ref, root, scene_tame);
            while (!ptc::lane_finished(l)) {
                if (ptc::lane_has_leaf(l)) {
                    uint32_t pos = ptc::lane_leaf_pos(l);
                    ptc::TriData tr; mem.load_tri(pos, tr);
                    ptc::lane_tri<false>(l, tr);
                    tris.push_back(pos); out[1]++;
                } else if (ptc::lane_has_node(l)) {
                    nodes.push_back(l.cur); out[0]++;
                    ptc::NodeData nd; mem.load_node(l.cur, nd);
                    if (l.tame) ptc::lane_node<false, true>(l, nd, st); else ptc::lane_node<false, false>(l, nd, st);
                } else if (l.sp > 0) {
                    ptc::lane_pop<false>(l, st);
                }
            }
        }
        std::sort(nodes.begin(), nodes.end()); std::sort(tris.begin(), tris.end());
        out[2] += std::unique(nodes.begin(), nodes.end()) - nodes.begin();
        out[3] += std::unique(tris.begin(), tris.end()) - tris.begin();
        out[4]++;
    }
}

/* Policy 6 (any-hit walks only): DEFERRED, DENSE triangle tests.  A lane that reaches a leaf does not
 * test its triangles: it appends {lane, first triangle, count} to a FIFO of the wave and goes on
 * walking at once (the any-hit answer does not depend on the order of the tests).  Whenever the
 * FIFO holds at least 64 triangle tests -- or the wave has nothing else to do -- one DENSE pass
 * tests the next 64 (or fewer) of them, one per lane whatever lane the ray belongs to; a hit marks
 * the owner's ray, which stops at its next round.  `lag` rounds pass between taking tests off the
 * FIFO and their results (the records are fetched a round ahead).  A ray is finished when its walk
 * is over and none of its tests is outstanding.
 * Costs: c_round per round (bookkeeping, pops, loads, enqueue), c_node per node section, c_tri per
 * dense pass, c_start per refill.  out->lane_tri counts triangle tests, out->tri_sections dense passes. */
#include <deque>
#include <vector>
static uint8_t* g_deferred_vis = nullptr;   /* optional: per ray, 1 = no triangle hit (set with wave_sim_deferred_vis) */
extern "C" void wave_sim_deferred_vis(uint8_t* vis) { g_deferred_vis = vis; }
extern "C" void wave_sim_deferred(const pt_flat_scene* sc, uint64_t n, const float* o, const float* d, const SimParams* p, SimStats* out) {
    CpuMem mem; mem.nodes = sc->nodes; mem.tris = sc->tris;
    ptc::RootBox root;
    root.mnx = sc->root_min[0]; root.mny = sc->root_min[1]; root.mnz = sc->root_min[2];
    root.mxx = sc->root_max[0]; root.mxy = sc->root_max[1]; root.mxz = sc->root_max[2];
    const bool scene_tame = ptc::root_is_tame(root);
    static ptc::Lane L[64];
    static CpuStack st[64];
    bool active[64], found[64];
    int outstanding[64];
    uint64_t ray_of[64];
    for (int i = 0; i < 64; i++) { active[i] = false; found[i] = false; outstanding[i] = 0; }
    struct Item { int lane; uint32_t tri; uint32_t count; };
    std::deque<Item> fifo;
    uint64_t queued_tests = 0;
    struct Batch { std::vector<std::pair<int, uint32_t>> tests; int due; };
    std::deque<Batch> in_flight;
    const int lag = p->tri_threshold;   /* reused: rounds between taking tests and having their results */
    uint64_t next = 0;
    SimStats s;
    memset(&s, 0, sizeof s);
    int round = 0;
    while (true) {
        int idle = 0;
        for (int i = 0; i < 64; i++) idle += !active[i];
        int start_list[64], n_start = 0;
        if (idle && next < n && (idle >= p->refill_min || idle == 64)) {
            for (int i = 0; i < 64 && next < n; i++) if (!active[i]) start_list[n_start++] = i, next++;
        }
        if (next >= n && idle == 64 && n_start == 0) break;
        round++;
        s.rounds++;
        s.cost += p->c_round;
        /* results of batches that are due */
        while (!in_flight.empty() && in_flight.front().due <= round) {
            for (auto& t : in_flight.front().tests) {
                ptc::TriData tr; mem.load_tri(t.second, tr);
                float tt, uu, vv;
                /* a hit counts if it is nearer than the best so far, which is +inf until the walk stops (kdtree.cpp:286) */
                if (ptc::tri_hit(tr, L[t.first].r, tt, uu, vv) && tt < ptc::pt_inf()) found[t.first] = true;
                outstanding[t.first]--;
                s.lane_tri++;
            }
            in_flight.pop_front();
        }
        /* rays marked as hit stop walking */
        for (int i = 0; i < 64; i++) if (active[i] && found[i]) { L[i].cur = PT_REF_NONE; L[i].sp = 0; }
        /* node section */
        bool some = false;
        for (int i = 0; i < 64; i++) {
            ptc::Lane& l = L[i];
            if (!active[i] || !ptc::lane_has_node(l)) continue;
            ptc::NodeData nd; mem.load_node(l.cur, nd);
            if (l.tame) ptc::lane_node<true, true>(l, nd, st[i]); else ptc::lane_node<true, false>(l, nd, st[i]);
            s.lane_node++; some = true;
        }
        if (some) { s.node_sections++; s.cost += p->c_node; }
        /* leaves reached (by the step or by last round's pop) are queued, then one pop */
        auto enqueue = [&](int i) {
            ptc::Lane& l = L[i];
            if (!ptc::lane_has_leaf(l)) return;
            uint32_t pos = ptc::lane_leaf_pos(l);
            uint32_t count = sc->tris[pos].leaf_remaining + 1u;
            fifo.push_back({i, pos, count});
            queued_tests += count; outstanding[i] += (int)count;
            l.cur = PT_REF_NONE;
        };
        for (int i = 0; i < 64; i++) if (active[i]) enqueue(i);
        for (int i = 0; i < 64; i++) {
            ptc::Lane& l = L[i];
            if (!active[i] || l.cur != PT_REF_NONE || l.sp <= 0) continue;
            ptc::lane_pop<true>(l, st[i]);
            s.lane_pop++;
            enqueue(i);   /* a popped leaf is queued in the same round; the lane pops again next round */
        }
        /* dense pass: when 64 tests wait, or when no lane has a node to visit and tests remain */
        bool walkers = false;
        for (int i = 0; i < 64; i++) if (active[i] && (ptc::lane_has_node(L[i]) || L[i].sp > 0)) walkers = true;
        if (queued_tests >= 64 || (queued_tests > 0 && (!walkers || (int)in_flight.size() == 0 && !walkers))) {
            Batch b; b.due = round + lag;
            int room = 64;
            while (room > 0 && !fifo.empty()) {
                Item& it = fifo.front();
                uint32_t take = it.count < (uint32_t)room ? it.count : (uint32_t)room;
                for (uint32_t k = 0; k < take; k++) b.tests.push_back({it.lane, it.tri + k});
                it.tri += take; it.count -= take; room -= (int)take; queued_tests -= take;
                if (it.count == 0) fifo.pop_front();
            }
            in_flight.push_back(b);
            s.tri_sections++; s.cost += p->c_tri;
        }
        /* retire: walk over and nothing outstanding */
        for (int i = 0; i < 64; i++)
            if (active[i] && ptc::lane_finished(L[i]) && outstanding[i] == 0) {
                active[i] = false;
                if (g_deferred_vis) g_deferred_vis[ray_of[i]] = found[i] ? 0 : 1;
            }
        if (n_start) {
            s.start_sections++; s.cost += p->c_start;
            for (int k = 0; k < n_start; k++) {
                int i = start_list[k];
                uint64_t r = next - n_start + k;
                L[i].r.ox = o[3 * r]; L[i].r.oy = o[3 * r + 1]; L[i].r.oz = o[3 * r + 2];
                L[i].r.dx = d[3 * r]; L[i].r.dy = d[3 * r + 1]; L[i].r.dz = d[3 * r + 2];
                ptc::ray_set_recip(L[i].r);
                ptc::lane_start(L[i], sc->root_ref, root, scene_tame);
                found[i] = false; outstanding[i] = 0; ray_of[i] = r;
                active[i] = !ptc::lane_finished(L[i]);
                if (!active[i] && g_deferred_vis) g_deferred_vis[r] = 1;   /* misses the root box */
            }
        }
    }
    *out = s;
}
