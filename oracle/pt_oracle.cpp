/*
 * pt_oracle.cpp -- CPU oracle for the ray-casting path of the reference path tracer.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (see pt_oracle.h).  PARITY UNPINNED: nothing in
 * the reference (no tests, no fixtures) and no compiled reference binary pins these
 * results; they follow from reading the cited reference lines.
 *
 * Arithmetic contract (what "the reference's bits" means here).  The reference does all
 * vector math through Eigen::Vector3f and is normally built with -ffast-math, which leaves
 * its own bits unspecified.  The oracle fixes them as strict IEEE-754 binary32 with no
 * contraction, evaluated the way Eigen 3.3/3.4 evaluates fixed-size 3-vectors:
 *   dot(a,b)      = a0*b0 + (a1*b1 + a2*b2)        (Eigen's unrolled non-vectorised redux
 *                                                    splits 3 terms as 1 + 2)
 *   cross(a,b)    = (a1*b2 - a2*b1, a2*b0 - a0*b2, a0*b1 - a1*b0)
 *   norm(a)       = sqrtf(dot(a,a))
 *   normalize(a)  : z = dot(a,a); if (z > 0) a_i = a_i / sqrtf(z)   (true division)
 *   a / s, a * s  : per component; a double scalar is narrowed to float first
 *                   (Eigen's scalar promotion for float expressions)
 * Scalar expressions the reference writes in double (1.0/x, comparisons against the double
 * literal 1e-8, the camera's pixel offsets) are evaluated in double here too.
 *
 * Random numbers.  The reference draws from a per-thread std::mt19937 seeded from
 * std::random_device (integrator.h:50-51), which no parallel renderer can replay.  The
 * oracle and the device code instead share a counter-based stream, defined in this file
 * (philox4x32_10, uniform01, normal_from_word, and the counter layout in rng_words).
 *
 * Build: g++ -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include "pt_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <map>
#include <thread>
#include <vector>

namespace {

/* ------------------------------------------------------------------ vectors */

struct V3 {
    float x, y, z;
    float operator()(int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

inline V3 mk(float x, float y, float z) { V3 v; v.x = x; v.y = y; v.z = z; return v; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float norm(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalized(V3 a) {
    float z = dot(a, a);
    if (z > 0.0f) { float n = sqrtf(z); return mk(a.x / n, a.y / n, a.z / n); }
    return a;
}

/* utils.h:29-35 -- note the operand order: NaN behaviour depends on it. */
inline float real_min(float x, float y) { return x < y ? x : y; }
inline float real_max(float x, float y) { return x < y ? y : x; }

const float F_INF = (float)1e100; /* utils.h:10 FLOAT_INF narrowed to float = +inf */
const double EPSILON = 1e-8;      /* utils.cpp:7, compared in double */

/* ---------------------------------------------------------------- geometry */

struct AABB {
    V3 lo, hi;
    AABB() { lo = mk(F_INF, F_INF, F_INF); hi = mk(-F_INF, -F_INF, -F_INF); } /* utils.cpp:35-40 */
    void update(const AABB& o) {                                               /* utils.cpp:51-54 */
        lo = mk(real_min(lo.x, o.lo.x), real_min(lo.y, o.lo.y), real_min(lo.z, o.lo.z));
        hi = mk(real_max(hi.x, o.hi.x), real_max(hi.y, o.hi.y), real_max(hi.z, o.hi.z));
    }
    void update(V3 p) {                                                        /* utils.cpp:46-49 */
        lo = mk(real_min(lo.x, p.x), real_min(lo.y, p.y), real_min(lo.z, p.z));
        hi = mk(real_max(hi.x, p.x), real_max(hi.y, p.y), real_max(hi.z, p.z));
    }
};

struct Ray { V3 o, d; };
struct CastingRay { V3 recip; Ray ray; };

/* utils.cpp:20-23: the Ray constructor normalises the direction. */
inline Ray make_ray(V3 o, V3 d) { Ray r; r.o = o; r.d = normalized(d); return r; }

/* utils.cpp:29-33: reciprocal formed in double, narrowed to float. */
inline CastingRay make_casting(const Ray& r) {
    CastingRay c; c.ray = r;
    c.recip = mk((float)(1.0 / (double)r.d.x), (float)(1.0 / (double)r.d.y), (float)(1.0 / (double)r.d.z));
    return c;
}

/* utils.cpp:59-83 */
inline bool box_hit(const AABB& b, const CastingRay& c) {
    V3 t0 = mk((b.lo.x - c.ray.o.x) * c.recip.x, (b.lo.y - c.ray.o.y) * c.recip.y, (b.lo.z - c.ray.o.z) * c.recip.z);
    V3 t1 = mk((b.hi.x - c.ray.o.x) * c.recip.x, (b.hi.y - c.ray.o.y) * c.recip.y, (b.hi.z - c.ray.o.z) * c.recip.z);
    V3 en = mk(real_min(t0.x, t1.x), real_min(t0.y, t1.y), real_min(t0.z, t1.z));
    V3 ex = mk(real_max(t0.x, t1.x), real_max(t0.y, t1.y), real_max(t0.z, t1.z));
    float t_start = real_max(en.x, real_max(en.y, en.z));
    float t_end = real_min(ex.x, real_min(ex.y, ex.z));
    if (t_end < 0) return false;
    if (t_start > t_end) return false;
    return true;
}

struct Tri {
    V3 p[3];
    V3 e1, e2, normal;
    AABB box;
    V3 base_n, u_n, v_n;
};

/* utils.cpp:116-129 */
inline Tri make_tri(V3 p0, V3 p1, V3 p2) {
    Tri t;
    t.p[0] = p0; t.p[1] = p1; t.p[2] = p2;
    t.e1 = p1 - p0;
    t.e2 = p2 - p0;
    t.normal = normalized(cross(t.e1, t.e2));
    t.box.lo = t.box.hi = p0; /* set_to_point */
    t.box.update(p1);
    t.box.update(p2);
    t.base_n = t.u_n = t.v_n = mk(0, 0, 0);
    return t;
}

/* utils.cpp:138-163 (Moller-Trumbore, two-sided, thresholds compared in double). */
inline bool tri_hit(const Tri& tr, const Ray& ray, float& t_out, float& u_out, float& v_out) {
    V3 P = cross(ray.d, tr.e2);
    float det = dot(tr.e1, P);
    if ((double)det > -EPSILON && (double)det < EPSILON) return false;
    float inv_det = (float)(1.0 / (double)det);
    V3 T = ray.o - tr.p[0];
    float u = dot(T, P) * inv_det;
    if (u < 0 || u > 1) return false;
    V3 Q = cross(T, tr.e1);
    float v = dot(ray.d, Q) * inv_det;
    if (v < 0 || u + v > 1) return false;
    float t = dot(tr.e2, Q) * inv_det;
    if ((double)t <= EPSILON) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

/* utils.cpp:165-170 */
inline float plane_parameter(const Tri& tr) {
    float s = (dot(tr.normal, tr.p[0]) + dot(tr.normal, tr.p[1])) + dot(tr.normal, tr.p[2]);
    return (float)((double)s / 3.0);
}
inline V3 project_to_altitude(const Tri& tr, V3 point, float altitude) {
    float parameter = dot(tr.normal, point);
    float change = (plane_parameter(tr) - parameter) + altitude;
    return point + change * tr.normal;
}

/* Runs f(a, b) over [0, n) cut into at most `threads` consecutive ranges, one thread each. */
template <typename F>
void parallel_for(int64_t n, int threads, F f) {
    if (threads <= 1 || n < 2) { f(0, n); return; }
    std::vector<std::thread> pool;
    int64_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; t++) {
        int64_t a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        pool.emplace_back([=]() { f(a, b); });
    }
    for (auto& th : pool) th.join();
}

/* --------------------------------------------------- STL + vertex normals */

struct V3Less { /* stlreader.cpp:46-60 */
    bool operator()(const V3& a, const V3& b) const {
        for (int i = 0; i < 3; i++) {
            if (a(i) < b(i)) return true;
            if (a(i) > b(i)) return false;
        }
        return false;
    }
};

/* stlreader.cpp:62-81: vertex normal = normalised sum (in mesh order) of the face normals
 * of every triangle corner that is bit-identical to the vertex.
 *
 * The reference keys a std::map<Vec, vector<int>> by the corner points and appends triangle
 * indices in mesh order.  Here the corners are SORTED by (point under the same ordering, corner
 * number) instead: a run of corners that the ordering cannot tell apart is one map entry, and
 * inside a run the corner numbers ascend, which is the map's append order -- so every sum adds
 * the same normals in the same order.  The order is total, hence the sorted sequence does not
 * depend on how it is produced, and it is produced by a threaded merge sort (64 M corners of the
 * largest test scene take minutes through a map). */
void compute_barycentric_normals(std::vector<Tri>& tris, int threads) {
    const int64_t nc = (int64_t)tris.size() * 3;
    if (threads < 1) threads = 1;
    V3Less less;
    auto before = [&](int64_t a, int64_t b) {
        const V3& pa = tris[(size_t)(a / 3)].p[a % 3];
        const V3& pb = tris[(size_t)(b / 3)].p[b % 3];
        if (less(pa, pb)) return true;
        if (less(pb, pa)) return false;
        return a < b;
    };
    std::vector<int64_t> order((size_t)nc), tmp;
    for (int64_t c = 0; c < nc; c++) order[(size_t)c] = c;
    int parts = 1;
    while (parts < threads && nc / (parts * 2) >= 65536) parts *= 2;
    const int64_t chunk = (nc + parts - 1) / parts;
    parallel_for(parts, parts, [&](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++)
            std::sort(order.begin() + std::min(nc, k * chunk), order.begin() + std::min(nc, (k + 1) * chunk), before);
    });
    for (int64_t width = chunk; width < nc; width *= 2) {     /* merge neighbouring sorted runs */
        tmp.resize((size_t)nc);
        const int64_t pairs = (nc + 2 * width - 1) / (2 * width);
        parallel_for(pairs, threads, [&](int64_t a, int64_t b) {
            for (int64_t k = a; k < b; k++) {
                const int64_t lo = k * 2 * width, mid = std::min(nc, lo + width), hi = std::min(nc, lo + 2 * width);
                std::merge(order.begin() + lo, order.begin() + mid, order.begin() + mid, order.begin() + hi, tmp.begin() + lo, before);
            }
        });
        order.swap(tmp);
    }
    /* runs of equal points; every corner of a run gets the run's sum */
    std::vector<V3> corner_normal((size_t)nc);
    std::vector<int64_t> run_start;
    for (int64_t i = 0; i < nc; i++) {
        if (i == 0) { run_start.push_back(0); continue; }
        const int64_t a = order[(size_t)i - 1], b = order[(size_t)i];
        const V3& pa = tris[(size_t)(a / 3)].p[a % 3];
        const V3& pb = tris[(size_t)(b / 3)].p[b % 3];
        if (less(pa, pb) || less(pb, pa)) run_start.push_back(i);
    }
    run_start.push_back(nc);
    parallel_for((int64_t)run_start.size() - 1, threads, [&](int64_t a, int64_t b) {
        for (int64_t r = a; r < b; r++) {
            V3 n = mk(0, 0, 0);
            for (int64_t i = run_start[(size_t)r]; i < run_start[(size_t)r + 1]; i++) n = n + tris[(size_t)(order[(size_t)i] / 3)].normal;
            const V3 vn = normalized(n);
            for (int64_t i = run_start[(size_t)r]; i < run_start[(size_t)r + 1]; i++) corner_normal[(size_t)order[(size_t)i]] = vn;
        }
    });
    for (size_t i = 0; i < tris.size(); i++) {
        Tri& tr = tris[i];
        const V3* vn = &corner_normal[i * 3];
        tr.base_n = vn[0];          /* utils.cpp:131-135 */
        tr.u_n = vn[1] - vn[0];
        tr.v_n = vn[2] - vn[0];
    }
}

/* The reference's formulation word for word (a std::map keyed by the points); kept so that the
 * tests can hold the sorted formulation above against it. */
void compute_barycentric_normals_map(std::vector<Tri>& tris) {
    std::map<V3, std::vector<int>, V3Less> point_to_tris;
    for (size_t i = 0; i < tris.size(); i++)
        for (int k = 0; k < 3; k++) point_to_tris[tris[i].p[k]].push_back((int)i);
    for (auto& tr : tris) {
        V3 vn[3];
        for (int k = 0; k < 3; k++) {
            V3 n = mk(0, 0, 0);
            for (int other : point_to_tris[tr.p[k]]) n = n + tris[other].normal;
            vn[k] = normalized(n);
        }
        tr.base_n = vn[0];
        tr.u_n = vn[1] - vn[0];
        tr.v_n = vn[2] - vn[0];
    }
}

/* stlreader.cpp:10-43: binary STL, file normals ignored, attribute count must be 0. */
bool read_stl(const char* path, std::vector<Tri>& out) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    char header[80];
    uint32_t count = 0;
    if (fread(header, 80, 1, fp) != 1 || fread(&count, 4, 1, fp) != 1) { fclose(fp); return false; }
    out.clear();
    out.reserve(count);
    for (uint32_t i = 0; i < count; i++) {
        float rec[12];
        uint16_t attr;
        if (fread(rec, 48, 1, fp) != 1 || fread(&attr, 2, 1, fp) != 1 || attr != 0) { fclose(fp); return false; }
        out.push_back(make_tri(mk(rec[3], rec[4], rec[5]), mk(rec[6], rec[7], rec[8]), mk(rec[9], rec[10], rec[11])));
    }
    fclose(fp);
    return true;
}

/* ---------------------------------------------------------------- the tree */

const int LEAF_THRESHOLD = 8;  /* kdtree.cpp:12 */
const int MAXIMUM_DEPTH = 19;  /* kdtree.cpp:13 */

struct Node {
    int depth = 0, axis = 0;
    float split = 0;
    Node* lo = nullptr;
    Node* hi = nullptr;
    AABB box;
    bool leaf = false;
    std::vector<int> tris; /* mesh indices, in the order the reference copies them (kdtree.cpp:17-30) */
    ~Node() { delete lo; delete hi; }
};

struct BuildCtx {
    const std::vector<Tri>* mesh;
    int spawn_depth; /* children above this depth are built on their own thread */
    int threads;     /* threads of the whole build; a node at depth d may use threads >> d of them */
};

typedef std::vector<int> IdxList;

/* kdtree.cpp:32-262.  lists[0][axis] is sorted by AABB minimum, lists[1][axis] by maximum. */
Node* build_node(const BuildCtx& ctx, int depth, IdxList lists[2][3]) {
    const std::vector<Tri>& mesh = *ctx.mesh;
    Node* n = new Node();
    n->depth = depth;
    const IdxList& ours = lists[0][0];
    unsigned int count = (unsigned int)ours.size();
    for (int index : ours) n->box.update(mesh[index].box);
    if (count <= (unsigned)LEAF_THRESHOLD || depth >= MAXIMUM_DEPTH) {
        n->leaf = true;
        n->tris = ours;
        return n;
    }
    float best_split = 0.0f;
    int best_axis = -1;
    float best_score = F_INF;
    int inc = 8;                     /* kdtree.cpp:92-98 */
    if (count < 10000) inc = 4;
    if (count < 1000) inc = 2;
    if (count < 100) inc = 1;
    /* The candidates in the reference's order (axis 0 first, then along the by-minimum-x list),
     * the winner being the FIRST one with the smallest score (strict '<', kdtree.cpp:129-143).
     * Wide nodes cut that sequence into consecutive ranges scored on several threads; taking the
     * ranges' winners in order with the same strict '<' picks the same candidate. */
    struct Best { float split; int axis; float score; };
    const unsigned int per_axis = (count + inc - 1) / inc;
    const int node_threads = count >= 200000 ? std::max(1, ctx.threads >> std::min(depth, 30)) : 1;
    std::vector<Best> part((size_t)node_threads, Best{0.0f, -1, F_INF});
    std::vector<int> part_used((size_t)node_threads, 0);
    const int64_t total_cand = (int64_t)per_axis * 3;
    const int64_t cand_chunk = (total_cand + node_threads - 1) / node_threads;
    parallel_for(total_cand, node_threads, [&](int64_t a, int64_t b) {
        Best best{0.0f, -1, F_INF};
        for (int64_t c = a; c < b; c++) {
            const int axis = (int)(c / per_axis);
            const unsigned int tt = (unsigned int)(c % per_axis) * (unsigned int)inc;
            float sample = mesh[ours[tt]].box.hi(axis);
            int side[2] = {0, 0};
            for (int mm = 0; mm < 2; mm++) {
                unsigned int low = 0, high = count;
                while (low + 1 < high) {
                    unsigned int mid = (low + high) / 2;
                    const Tri& tr = mesh[lists[mm][axis][mid]];
                    float coord = mm == 0 ? tr.box.lo(axis) : tr.box.hi(axis);
                    if (coord <= sample) low = mid; else high = mid;
                }
                if (mm == 0) side[mm] = (int)(low + 1);
                else side[mm] = (int)(count - high);
            }
            float score = real_max((float)side[0], (float)side[1]);
            if (score < best.score) { best.split = sample; best.axis = axis; best.score = score; }
        }
        part[(size_t)(a / cand_chunk)] = best;
        part_used[(size_t)(a / cand_chunk)] = 1;
    });
    for (int k = 0; k < node_threads; k++)
        if (part_used[(size_t)k] && part[(size_t)k].score < best_score) {
            best_split = part[(size_t)k].split; best_axis = part[(size_t)k].axis; best_score = part[(size_t)k].score;
        }
    n->axis = best_axis;
    n->split = best_split;
    IdxList lo_l[2][3], hi_l[2][3];
    /* the six lists are filtered independently of one another (one thread each on wide nodes) */
    parallel_for(6, node_threads > 1 ? 6 : 1, [&](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++) {
            const int axis = (int)(k / 2), mm = (int)(k % 2);
            for (unsigned int i = 0; i < count; i++) {
                int ti = lists[mm][axis][i];
                bool below = mesh[ti].box.lo(best_axis) <= best_split; /* kdtree.cpp:172-183 */
                if (below) lo_l[mm][axis].push_back(ti);
                else hi_l[mm][axis].push_back(ti);
            }
        }
    });
    unsigned int lo_n = (unsigned int)lo_l[0][0].size(), hi_n = (unsigned int)hi_l[0][0].size();
    if (hi_n == count || lo_n == count) { /* kdtree.cpp:206-208: no improvement -> leaf */
        n->leaf = true;
        n->axis = 0; n->split = 0;
        n->tris = ours;
        return n;
    }
    /* free the parent's lists before recursing (they are not needed any more) */
    for (int axis = 0; axis < 3; axis++) for (int mm = 0; mm < 2; mm++) { IdxList().swap(lists[mm][axis]); }
    if (depth < ctx.spawn_depth) {
        std::future<Node*> f = std::async(std::launch::async, [&]() { return build_node(ctx, depth + 1, lo_l); });
        n->hi = build_node(ctx, depth + 1, hi_l);
        n->lo = f.get();
    } else {
        n->lo = build_node(ctx, depth + 1, lo_l);
        n->hi = build_node(ctx, depth + 1, hi_l);
    }
    return n;
}

/* kdtree.cpp:466-494: six index lists sorted once with std::sort and a strict '<' on one
 * AABB coordinate.  The order among equal keys is whatever libstdc++'s std::sort leaves for
 * this input sequence; using std::sort on the same sequence reproduces it. */
Node* build_tree(const std::vector<Tri>& mesh, int threads) {
    IdxList lists[2][3];
    /* six independent sorts of the same initial sequence: one thread each gives the same lists */
    parallel_for(6, threads > 1 ? 6 : 1, [&](int64_t a, int64_t b) {
        for (int64_t k = a; k < b; k++) {
            const int axis = (int)(k / 2), mm = (int)(k % 2);
            IdxList& l = lists[mm][axis];
            l.resize(mesh.size());
            for (size_t i = 0; i < mesh.size(); i++) l[i] = (int)i;
            if (mm == 0)
                std::sort(l.begin(), l.end(), [&](const int& x, const int& y) { return mesh[x].box.lo(axis) < mesh[y].box.lo(axis); });
            else
                std::sort(l.begin(), l.end(), [&](const int& x, const int& y) { return mesh[x].box.hi(axis) < mesh[y].box.hi(axis); });
        }
    });
    BuildCtx ctx;
    ctx.mesh = &mesh;
    ctx.threads = threads;
    ctx.spawn_depth = 0;
    while ((1 << ctx.spawn_depth) < threads) ctx.spawn_depth++;
    return build_node(ctx, 0, lists);
}

struct Hit { float t, u, v; int idx; };

struct Counters { uint64_t box = 0, tri = 0; };

/* kdtree.cpp:271-396.  Outputs are written only on a hit. */
template <bool COUNT>
bool node_ray_test(const Node* n, const std::vector<Tri>& mesh, const CastingRay& c, Hit& out, Counters* cn) {
    if (COUNT) cn->box++;
    if (!box_hit(n->box, c)) return false;
    if (n->leaf) {
        bool overall = false;
        Hit best; best.t = F_INF; best.u = best.v = 0; best.idx = -1;
        for (size_t i = 0; i < n->tris.size(); i++) {
            float t, u, v;
            if (COUNT) cn->tri++;
            bool r = tri_hit(mesh[n->tris[i]], c.ray, t, u, v);
            if (r && t < best.t) { best.t = t; best.u = u; best.v = v; best.idx = n->tris[i]; overall = true; }
        }
        if (overall) out = best;
        return overall;
    }
    float origin = c.ray.o(n->axis);
    bool over_hi = n->hi != nullptr && n->hi->box.lo(n->axis) < origin;
    bool over_lo = n->lo != nullptr && origin <= n->lo->box.hi(n->axis);
    const Node* near_side = over_hi ? n->hi : n->lo;
    const Node* far_side = over_hi ? n->lo : n->hi;
    if (over_hi && over_lo) {
        bool lo_r = node_ray_test<COUNT>(n->lo, mesh, c, out, cn);
        Hit tmp;
        bool hi_r = node_ray_test<COUNT>(n->hi, mesh, c, tmp, cn);
        if (!hi_r) return lo_r;
        if (!lo_r || tmp.t < out.t) out = tmp;
        return true;
    }
    if (near_side != nullptr) {
        if (node_ray_test<COUNT>(near_side, mesh, c, out, cn)) {
            if (far_side == nullptr) return true;
            float e = near_side == n->lo ? far_side->box.lo(n->axis) : far_side->box.hi(n->axis);
            float t_far = (e - origin) / c.ray.d(n->axis);
            if (out.t > t_far) {
                Hit tmp;
                bool r = node_ray_test<COUNT>(far_side, mesh, c, tmp, cn);
                if (r && tmp.t < out.t) out = tmp;
            }
            return true;
        }
    }
    if (far_side != nullptr) return node_ray_test<COUNT>(far_side, mesh, c, out, cn);
    return false;
}

/* Extension with no counterpart in the reference (the bound its note at kdtree.cpp:356-357
 * asks for): is any triangle hit at a ray parameter t < bound?  Every node whose box the ray
 * hits (AABB::does_ray_intersect, as in ray_test's first statement) is searched, so the answer
 * does not depend on a visit order. */
bool node_any_within(const Node* n, const std::vector<Tri>& mesh, const CastingRay& c, float bound) {
    if (n == nullptr || !box_hit(n->box, c)) return false;
    if (n->leaf) {
        for (size_t i = 0; i < n->tris.size(); i++) {
            float t, u, v;
            if (tri_hit(mesh[n->tris[i]], c.ray, t, u, v) && t < bound) return true;
        }
        return false;
    }
    return node_any_within(n->lo, mesh, c, bound) || node_any_within(n->hi, mesh, c, bound);
}

/* ------------------------------------------------------------ random stream */

/* Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11). */
inline void philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Top 24 bits -> the midpoint of one of 2^24 equal cells of (0,1); exact in double and float. */
inline double uniform01(uint32_t w) { return ((double)(w >> 8) + 0.5) * (1.0 / 16777216.0); }

/* ln(y) for a normal binary32 y in (0, 2), from +,-,*,/ only so that every IEEE machine agrees:
 * y = m * 2^e with m in (sqrt(1/2), sqrt(2)], ln m = 2 atanh((m-1)/(m+1)), odd series to s^9. */
inline float det_logf(float y) {
    uint32_t bits;
    memcpy(&bits, &y, 4);
    int e = (int)((bits >> 23) & 0xffu) - 127;
    bits = (bits & 0x007fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &bits, 4);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 1.0f / 9.0f;
    p = p * z + 1.0f / 7.0f;
    p = p * z + 1.0f / 5.0f;
    p = p * z + 1.0f / 3.0f;
    p = p * z + 1.0f;
    return (float)e * 0.693147182f + (2.0f * s) * p;
}

/* Inverse error function in binary32 (M. Giles, "Approximating the erfinv function", 2011). */
inline float det_erfinvf(float x) {
    float w = -det_logf((1.0f - x) * (1.0f + x));
    float p;
    if (w < 5.0f) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3.0f;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}

/* Standard normal from one random word by inverse CDF: bit 31 is the sign, bits 30..8 pick one
 * of 2^23 cell midpoints x = (j + 0.5) / 2^23 in (0, 1); the variate is sqrt(2) * erfinv(+-x). */
inline float normal_from_word(uint32_t w) {
    float x = ((float)((w >> 8) & 0x7fffffu) + 0.5f) * (1.0f / 8388608.0f);
    if (w & 0x80000000u) x = -x;
    return 1.41421354f * det_erfinvf(x);
}

/* Counter layout: one Philox block per (pixel, sample, stream).
 *   stream 0                 : camera -- words 0,1 = anti-aliasing x,y; words 2,3 = aperture x,y
 *   stream 16 + 16*depth + l : path vertex `depth` (0 = first hit), light l -- words 0..2 = the
 *                              three light-delocalisation normals
 * key = the two halves of the 64-bit seed. */
const uint32_t STREAM_CAMERA = 0;
const int MAX_LIGHTS = 16;
inline void rng_words(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t out[4]) {
    uint32_t ctr[4] = {pixel, sample, stream, 0u};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    philox(ctr, key, out);
}

/* -------------------------------------------------------------- integrator */

struct Scene {
    std::vector<Tri> mesh;
    Node* root = nullptr;
    ~Scene() { delete root; }
};

struct CameraBasis { V3 right, up; float aspect; };

/* integrator.cpp:215-220 (identical in get_ray_for_pixel, :198-203) */
inline CameraBasis camera_basis(const orc_camera& cam, int W, int H) {
    CameraBasis b;
    V3 dir = mk(cam.direction[0], cam.direction[1], cam.direction[2]);
    V3 up = mk(cam.up[0], cam.up[1], cam.up[2]);
    b.right = normalized(cross(dir, up));
    b.up = normalized(cross(b.right, dir));
    b.aspect = (float)H / (float)W;
    return b;
}

/* integrator.cpp:197-209: note the integer W/2 and H/2. */
inline Ray pixel_ray_plain(const orc_camera& cam, const CameraBasis& b, int W, int H, int x, int y) {
    float ipw = cam.image_plane_width;
    float dx = ipw * (float)(x - W / 2) / (float)W;
    float dy = -ipw * (float)(y - H / 2) * b.aspect / (float)H;
    V3 offset = b.right * dx + b.up * dy;
    V3 dir = mk(cam.direction[0], cam.direction[1], cam.direction[2]);
    return make_ray(mk(cam.origin[0], cam.origin[1], cam.origin[2]), dir + offset);
}

/* integrator.cpp:248-259: jittered pixel offset in double, narrowed on assignment; aperture
 * offsets move the origin and tilt the direction, which is NOT renormalised. */
inline Ray pixel_ray_jittered(const orc_camera& cam, const CameraBasis& b, int W, int H, int x, int y,
                              uint32_t sample, uint64_t seed) {
    uint32_t w[4];
    rng_words(seed, (uint32_t)(y * W + x), sample, STREAM_CAMERA, w);
    double jx = uniform01(w[0]) - 0.5, jy = uniform01(w[1]) - 0.5;
    float ipw = cam.image_plane_width;
    float dx = (float)((double)ipw * ((double)x + jx - (double)W / 2.0) / (double)(float)W);
    float dy = (float)((double)(-ipw) * ((double)y + jy - (double)H / 2.0) * (double)b.aspect / (double)(float)H);
    V3 offset = b.right * dx + b.up * dy;
    V3 dir = mk(cam.direction[0], cam.direction[1], cam.direction[2]);
    Ray ray = make_ray(mk(cam.origin[0], cam.origin[1], cam.origin[2]), dir + offset);
    float dof_x = normal_from_word(w[2]) * cam.dof_dispersion;
    float dof_y = normal_from_word(w[3]) * cam.dof_dispersion;
    ray.o = ray.o + dof_x * b.right;
    ray.o = ray.o + dof_y * b.up;
    ray.d = ray.d - (dof_x / cam.plane_of_focus_distance) * b.right;
    ray.d = ray.d - (dof_y / cam.plane_of_focus_distance) * b.up;
    return ray;
}

inline float square(float x) { return x * x; }

struct PathCtx {
    const Scene* scene;
    const orc_light* lights;
    int n_lights;
    V3 sky;
    uint64_t seed;
    uint32_t pixel, sample;
    uint32_t bounce_mode, shadow_mode;
    uint64_t closest_casts, shadow_casts;
    Counters* work_closest; /* optional: box/triangle tests spent on closest-hit casts */
    Counters* work_shadow;  /* ... and on shadow casts */
    /* Bookkeeping for the GPU renderer's own ray accounting (not part of the reference): a light
     * sample whose weight is exactly zero for all three colours adds nothing whether its light is
     * visible or not; the GPU renderer counts such a shadow ray but does not trace it.
     * shadow_weighted counts the others; capture (optional) receives every ray cast, 8 floats each:
     * origin, direction, light distance (0 for closest-hit casts), and a tag (0 closest-hit cast,
     * 1 shadow cast with a non-zero weight, 2 shadow cast with zero weight). */
    uint64_t shadow_weighted = 0;
    std::vector<float>* capture = nullptr;
    /* Added to the stream numbers of the bounce draws and of the light draws below the first hit:
     * 256 * k for sub-path k of a cast_ray call with several branches, 0 otherwise. */
    uint32_t stream_off = 0;
    /* Product of the bounce weights down to the current path vertex, one binary32 product per bounce
     * (1 at the first hit): what the GPU renderer multiplies a light sample's weight by.  Used only
     * for the bookkeeping of shadow_weighted; the radiance is formed as the reference forms it. */
    float throughput = 1.0f;
};

inline void capture_ray(PathCtx& ctx, const Ray& r, float dist, float tag) {
    if (!ctx.capture) return;
    const float rec[8] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, dist, tag};
    ctx.capture->insert(ctx.capture->end(), rec, rec + 8);
}

inline bool scene_ray_test(const Scene* s, const Ray& ray, Hit& h, Counters* cn = nullptr) {
    CastingRay c = make_casting(ray); /* kdtree.cpp:577-580: implicit Ray -> CastingRay */
    if (cn) return node_ray_test<true>(s->root, s->mesh, c, h, cn);
    return node_ray_test<false>(s->root, s->mesh, c, h, nullptr);
}

/* integrator.cpp:51-156.  depth counts path vertices from the first cast (0).  With the counter-based
 * random stream the `branches` sub-paths of the first hit (:77) need streams of their own to be
 * independent estimates: sub-path k adds 256*k to its stream numbers (k = 0: the plain streams). */
V3 cast_ray(PathCtx& ctx, const Ray& ray, int recursions, int depth, int branches = 1) {
    Hit h;
    ctx.closest_casts++;
    capture_ray(ctx, ray, 0.0f, 0.0f);
    bool result = scene_ray_test(ctx.scene, ray, h, ctx.work_closest);
    V3 energy = mk(0, 0, 0);
    if (!result) return ctx.sky; /* :153 */
    const Tri& tr = ctx.scene->mesh[h.idx];
    V3 n = normalized((tr.base_n + h.u * tr.u_n) + h.v * tr.v_n);   /* :66-67 */
    V3 hit = ray.o + h.t * ray.d;                                   /* :69 */
    hit = project_to_altitude(tr, hit, (float)1e-3);                /* :72 */
    V3 reflection = normalized(ray.d - (2.0f * dot(n, ray.d)) * n); /* :73-74 */
    for (int branch = 0; recursions > 0 && branch < branches; branch++) { /* :76-77 */
        /* :79-84 draw a scatter direction that :104 then replaces by the mirror direction; the
         * draws only advance the reference's engine and have no counterpart in a counter-based
         * stream.  Ray's constructor normalises the (already unit) direction again. */
        const uint32_t outer_off = ctx.stream_off;
        if (branches > 1) ctx.stream_off = 256u * (uint32_t)branch;
        V3 scatter_direction = reflection;
        if (ctx.bounce_mode == 1) {
            /* :79-84 as written, kept instead of being overwritten at :104.  sample_unit_sphere
             * (utils.cpp:177-185) is three standard normals, normalised; stream 1 + depth. */
            uint32_t sw[4];
            rng_words(ctx.seed, ctx.pixel, ctx.sample, 1u + (uint32_t)depth + ctx.stream_off, sw);
            V3 local = normalized(mk(normal_from_word(sw[0]), normal_from_word(sw[1]), normal_from_word(sw[2])));
            local.x = fabsf(local.x);                                   /* :80 */
            V3 d1 = normalized(tr.e1);                                  /* :82 */
            V3 d2 = cross(tr.normal, d1);                               /* :83 */
            scatter_direction = (local.x * tr.normal + local.y * d1) + local.z * d2; /* :84 */
        }
        Ray scattered = make_ray(hit, scatter_direction);
        float w = (float)((1.0 / branches) * 0.8);
        const float outer_throughput = ctx.throughput;
        ctx.throughput = w * outer_throughput;
        energy = energy + w * cast_ray(ctx, scattered, recursions - 1, depth + 1, 1); /* :107 */
        ctx.throughput = outer_throughput;
        ctx.stream_off = outer_off;
    }
    for (int l = 0; l < ctx.n_lights; l++) { /* :112-148 */
        uint32_t w[4];
        rng_words(ctx.seed, ctx.pixel, ctx.sample, 16u + 16u * (uint32_t)depth + (uint32_t)l + ctx.stream_off, w);
        V3 jitter = mk(0.2f * normal_from_word(w[0]), 0.2f * normal_from_word(w[1]),
                       0.2f * normal_from_word(w[2]));
        V3 lp = mk(ctx.lights[l].position[0], ctx.lights[l].position[1], ctx.lights[l].position[2]);
        V3 lc = mk(ctx.lights[l].color[0], ctx.lights[l].color[1], ctx.lights[l].color[2]);
        V3 to_light = (jitter + lp) - hit;
        Ray shadow = make_ray(hit, to_light);
        Hit sh;
        ctx.shadow_casts++;
        float dist = norm(to_light);
        {   /* the sample's weight as the GPU renderer forms it: the path's throughput times
             * colour / dist^2 * (lambert + phong^16) */
            const float thr = ctx.throughput;
            V3 wgt = (lc / (dist * dist)) * (real_max(0.0f, dot(n, to_light) / dist) +
                                             square(square(square(square(real_max(0.0f, dot(reflection, to_light) / dist))))));
            const bool zero = thr * wgt.x == 0.0f && thr * wgt.y == 0.0f && thr * wgt.z == 0.0f;
            if (!zero) ctx.shadow_weighted++;
            capture_ray(ctx, shadow, dist, zero ? 2.0f : 1.0f);
        }
        bool visible;
        if (ctx.shadow_mode == 1) {
            /* flagged extension: a light is hidden only by a triangle nearer than the light */
            visible = !node_any_within(ctx.scene->root, ctx.scene->mesh, make_casting(shadow), dist);
        } else {
        bool shadow_result = scene_ray_test(ctx.scene, shadow, sh, ctx.work_shadow);
        /* :127 passes (ignore_u, ignore_v, shadow_param) where (t, u, v) are expected, so the
         * value compared with the light distance is the barycentric v of the closest hit. */
        visible = !shadow_result || sh.v > dist;
        }
        if (visible) {
            V3 contribution = lc / (dist * dist);
            float lambert = dot(n, to_light) / dist;
            lambert = real_max(0.0f, lambert);
            float phong = real_max(0.0f, dot(reflection, to_light) / dist);
            phong = square(square(square(square(phong))));
            energy = energy + contribution * (lambert + phong);
        }
    }
    return energy;
}

inline Ray input_ray(const float* o, const float* d, int64_t i, int normalize) {
    V3 oo = mk(o[3 * i], o[3 * i + 1], o[3 * i + 2]);
    V3 dd = mk(d[3 * i], d[3 * i + 1], d[3 * i + 2]);
    if (normalize) return make_ray(oo, dd);
    Ray r; r.o = oo; r.d = dd; return r;
}

void finish_scene(Scene* s, int threads) {
    const char* how = getenv("ORC_VERTEX_NORMALS");      /* "map": the reference's std::map formulation */
    if (how && strcmp(how, "map") == 0) compute_barycentric_normals_map(s->mesh);
    else compute_barycentric_normals(s->mesh, threads);
    s->root = build_tree(s->mesh, threads < 1 ? 1 : threads);
}

void dump_rec(const Node* n, std::vector<float>* nodes, std::vector<int32_t>* leaves, int64_t* stats) {
    if (stats) {
        stats[0]++;
        if (n->leaf) { stats[1]++; stats[4] += (int64_t)n->tris.size(); }
        if (n->depth > stats[2]) stats[2] = n->depth;
        if ((int64_t)n->tris.size() > stats[3]) stats[3] = (int64_t)n->tris.size();
    }
    if (nodes) {
        float rec[10] = {n->leaf ? 1.0f : 0.0f, (float)n->axis, n->split, n->box.lo.x, n->box.lo.y, n->box.lo.z,
                         n->box.hi.x, n->box.hi.y, n->box.hi.z, (float)n->tris.size()};
        nodes->insert(nodes->end(), rec, rec + 10);
    }
    if (leaves && n->leaf) leaves->insert(leaves->end(), n->tris.begin(), n->tris.end());
    if (!n->leaf) { dump_rec(n->lo, nodes, leaves, stats); dump_rec(n->hi, nodes, leaves, stats); }
}

} // namespace

struct orc_scene { Scene s; };

extern "C" {

orc_scene* orc_scene_load_stl(const char* path, int build_threads) {
    orc_scene* h = new orc_scene();
    if (!read_stl(path, h->s.mesh)) { delete h; return nullptr; }
    finish_scene(&h->s, build_threads);
    return h;
}

orc_scene* orc_scene_from_triangles(const float* verts, int64_t n, int build_threads) {
    orc_scene* h = new orc_scene();
    h->s.mesh.reserve((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        const float* p = verts + 9 * i;
        h->s.mesh.push_back(make_tri(mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), mk(p[6], p[7], p[8])));
    }
    finish_scene(&h->s, build_threads);
    return h;
}

void orc_scene_destroy(orc_scene* s) { delete s; }

int64_t orc_scene_num_triangles(const orc_scene* s) { return (int64_t)s->s.mesh.size(); }

void orc_scene_triangle_data(const orc_scene* s, float* out) {
    for (size_t i = 0; i < s->s.mesh.size(); i++) {
        const Tri& t = s->s.mesh[i];
        const V3 v[9] = {t.p[0], t.p[1], t.p[2], t.e1, t.e2, t.normal, t.base_n, t.u_n, t.v_n};
        float* o = out + 30 * i;
        for (int k = 0; k < 9; k++) { o[3 * k] = v[k].x; o[3 * k + 1] = v[k].y; o[3 * k + 2] = v[k].z; }
        o[27] = plane_parameter(t);
        o[28] = o[29] = 0.0f;
    }
}

void orc_tree_stats(const orc_scene* s, int64_t stats[5]) {
    for (int i = 0; i < 5; i++) stats[i] = 0;
    dump_rec(s->s.root, nullptr, nullptr, stats);
}

void orc_tree_dump(const orc_scene* s, float* nodes10, int32_t* leaf_indices) {
    std::vector<float> nodes;
    std::vector<int32_t> leaves;
    dump_rec(s->s.root, &nodes, &leaves, nullptr);
    memcpy(nodes10, nodes.data(), nodes.size() * sizeof(float));
    if (!leaves.empty()) memcpy(leaf_indices, leaves.data(), leaves.size() * sizeof(int32_t));
}

uint64_t orc_tree_hash(const orc_scene* s) {
    std::vector<float> nodes;
    std::vector<int32_t> leaves;
    dump_rec(s->s.root, &nodes, &leaves, nullptr);
    uint64_t h = 1469598103934665603ull;
    auto feed = [&](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    };
    feed(nodes.data(), nodes.size() * sizeof(float));
    feed(leaves.data(), leaves.size() * sizeof(int32_t));
    return h;
}

void orc_raycast(const orc_scene* s, int64_t n, const float* origins, const float* dirs, int normalize,
                 int32_t* idx, float* t, float* u, float* v, int threads) {
    parallel_for(n, threads, [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            Ray r = input_ray(origins, dirs, i, normalize);
            Hit h;
            if (scene_ray_test(&s->s, r, h)) { idx[i] = h.idx; t[i] = h.t; u[i] = h.u; v[i] = h.v; }
            else { idx[i] = -1; t[i] = u[i] = v[i] = 0.0f; }
        }
    });
}

void orc_raycast_counts(const orc_scene* s, int64_t n, const float* origins, const float* dirs, int normalize,
                        uint64_t* box_tests, uint64_t* tri_tests) {
    Counters cn;
    for (int64_t i = 0; i < n; i++) {
        Ray r = input_ray(origins, dirs, i, normalize);
        CastingRay c = make_casting(r);
        Hit h;
        node_ray_test<true>(s->s.root, s->s.mesh, c, h, &cn);
    }
    *box_tests = cn.box;
    *tri_tests = cn.tri;
}

void orc_raycast_brute(const orc_scene* s, int64_t n, const float* origins, const float* dirs, int normalize,
                       int32_t* idx, float* t, float* u, float* v, int threads) {
    parallel_for(n, threads, [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            Ray r = input_ray(origins, dirs, i, normalize);
            Hit best; best.t = F_INF; best.u = best.v = 0; best.idx = -1;
            for (size_t k = 0; k < s->s.mesh.size(); k++) {
                float tt, uu, vv;
                if (tri_hit(s->s.mesh[k], r, tt, uu, vv) && tt < best.t) { best.t = tt; best.u = uu; best.v = vv; best.idx = (int)k; }
            }
            idx[i] = best.idx;
            if (best.idx >= 0) { t[i] = best.t; u[i] = best.u; v[i] = best.v; }
            else { t[i] = u[i] = v[i] = 0.0f; }
        }
    });
}

void orc_shadow_visible(const orc_scene* s, int64_t n, const float* origins, const float* dirs, const float* dist,
                        uint8_t* visible, int threads) {
    parallel_for(n, threads, [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            Ray r = input_ray(origins, dirs, i, 0);
            Hit h;
            bool res = scene_ray_test(&s->s, r, h);
            visible[i] = (!res || h.v > dist[i]) ? 1 : 0;
        }
    });
}

void orc_occlusion_visible(const orc_scene* s, int64_t n, const float* origins, const float* dirs, const float* max_t,
                           uint8_t* visible, int threads) {
    parallel_for(n, threads, [=](int64_t a, int64_t b) {
        for (int64_t i = a; i < b; i++) {
            Ray r = input_ray(origins, dirs, i, 0);
            visible[i] = node_any_within(s->s.root, s->s.mesh, make_casting(r), max_t[i]) ? 0 : 1;
        }
    });
}

void orc_camera_rays(const orc_camera* cam, int width, int height, int x0, int y0, int w, int h, int jittered,
                     uint32_t sample, uint64_t seed, float* origins, float* dirs) {
    CameraBasis b = camera_basis(*cam, width, height);
    int64_t k = 0;
    for (int y = y0; y < y0 + h; y++)
        for (int x = x0; x < x0 + w; x++, k++) {
            Ray r = jittered ? pixel_ray_jittered(*cam, b, width, height, x, y, sample, seed)
                             : pixel_ray_plain(*cam, b, width, height, x, y);
            origins[3 * k] = r.o.x; origins[3 * k + 1] = r.o.y; origins[3 * k + 2] = r.o.z;
            dirs[3 * k] = r.d.x; dirs[3 * k + 1] = r.d.y; dirs[3 * k + 2] = r.d.z;
        }
}

static void render_impl(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int width,
                        int height, int x0, int y0, int w, int h, uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                        float* rgb_sum, int32_t* passes, uint64_t* ray_counts, int threads, bool count_work,
                        std::vector<float>* capture = nullptr) {
    if (n_lights > MAX_LIGHTS) n_lights = MAX_LIGHTS;
    /* PassDescriptor::clamp_bounds, integrator.cpp:164-183 */
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x0 > width - 1) x0 = width - 1;
    if (y0 > height - 1) y0 = height - 1;
    if (w == -1) w = width;
    if (h == -1) h = height;
    if (w > width - x0) w = width - x0;
    if (h > height - y0) h = height - y0;
    CameraBasis basis = camera_basis(*cam, width, height);
    int nthreads = threads < 1 ? 1 : threads;
    if (capture) nthreads = 1;   /* rays are captured in the order one thread casts them */
    std::vector<uint64_t> counts((size_t)nthreads * 7, 0);
    /* Rows are dealt round-robin in blocks so threads finish together. */
    std::vector<std::thread> pool;
    auto work = [&](int tid) {
        PathCtx ctx;
        ctx.scene = &s->s; ctx.lights = lights; ctx.n_lights = n_lights;
        ctx.sky = mk(cam->sky_color[0], cam->sky_color[1], cam->sky_color[2]);
        ctx.bounce_mode = cam->bounce_mode;
        ctx.shadow_mode = cam->shadow_mode;
        ctx.seed = seed; ctx.closest_casts = 0; ctx.shadow_casts = 0;
        Counters wc, ws;
        ctx.work_closest = count_work ? &wc : nullptr;
        ctx.work_shadow = count_work ? &ws : nullptr;
        ctx.capture = capture;
        for (int y = y0 + tid; y < y0 + h; y += nthreads)
            for (int x = x0; x < x0 + w; x++) {
                int pix = y * width + x;
                for (uint32_t sm = first_sample; sm < first_sample + n_samples; sm++) {
                    ctx.pixel = (uint32_t)pix; ctx.sample = sm;
                    Ray ray = pixel_ray_jittered(*cam, basis, width, height, x, y, sm, seed);
                    V3 c = cast_ray(ctx, ray, 10, 0);       /* integrator.cpp:261 */
                    rgb_sum[3 * pix] += c.x;                  /* :263 */
                    rgb_sum[3 * pix + 1] += c.y;
                    rgb_sum[3 * pix + 2] += c.z;
                    passes[pix] += 1;                         /* :265 */
                }
            }
        uint64_t* c = &counts[7 * (size_t)tid];
        c[0] = ctx.closest_casts; c[1] = ctx.shadow_casts;
        c[2] = wc.box; c[3] = wc.tri; c[4] = ws.box; c[5] = ws.tri;
        c[6] = ctx.shadow_weighted;
    };
    if (nthreads == 1) work(0);
    else {
        for (int t = 0; t < nthreads; t++) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    if (ray_counts)
        for (int t = 0; t < nthreads; t++) {
            const uint64_t* c = &counts[7 * (size_t)t];
            if (count_work) { for (int k = 0; k < 7; k++) ray_counts[k] += c[k]; }
            else { ray_counts[0] += c[0]; ray_counts[1] += c[1]; ray_counts[2] += c[6]; }
        }
}

void orc_render(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int width,
                int height, int x0, int y0, int w, int h, uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                float* rgb_sum, int32_t* passes, uint64_t* ray_counts, int threads) {
    render_impl(s, cam, lights, n_lights, width, height, x0, y0, w, h, first_sample, n_samples, seed, rgb_sum, passes,
                ray_counts, threads, false);
}

void orc_render_counting(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int width,
                         int height, int x0, int y0, int w, int h, uint32_t first_sample, uint32_t n_samples,
                         uint64_t seed, float* rgb_sum, int32_t* passes, uint64_t* work_counts, int threads) {
    render_impl(s, cam, lights, n_lights, width, height, x0, y0, w, h, first_sample, n_samples, seed, rgb_sum, passes,
                work_counts, threads, true);
}

/* The same render on one thread, handing back every ray it casts, in casting order: 8 floats per
 * ray (origin, direction, light distance, tag; see PathCtx::capture).  Returns the number of rays;
 * rays beyond `cap` are counted but not stored. */
int64_t orc_render_capture(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int width,
                           int height, int x0, int y0, int w, int h, uint32_t first_sample, uint32_t n_samples,
                           uint64_t seed, float* rays8, int64_t cap) {
    std::vector<float> rgb((size_t)width * height * 3, 0.0f), got;
    std::vector<int32_t> passes((size_t)width * height, 0);
    render_impl(s, cam, lights, n_lights, width, height, x0, y0, w, h, first_sample, n_samples, seed, rgb.data(),
                passes.data(), nullptr, 1, false, &got);
    const int64_t n = (int64_t)(got.size() / 8);
    if (rays8 && cap > 0) memcpy(rays8, got.data(), (size_t)(n < cap ? n : cap) * 8 * sizeof(float));
    return n;
}

/* Integrator::cast_ray(ray, recursions, branches) (integrator.cpp:51-156) for rays of the caller's
 * own.  Directions are used as given.  Ray i draws from the stream keyed by ids[i] (NULL: i) in place
 * of a pixel number, and by `sample`.  counts = {closest casts, shadow casts, shadow casts of light
 * samples with non-zero weight}. */
void orc_cast_rays(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int64_t n,
                   const float* origins, const float* dirs, const uint32_t* ids, uint32_t sample, int recursions,
                   int branches, uint64_t seed, float* rgb, uint64_t* counts) {
    if (n_lights > MAX_LIGHTS) n_lights = MAX_LIGHTS;
    PathCtx ctx;
    ctx.scene = &s->s; ctx.lights = lights; ctx.n_lights = n_lights;
    ctx.sky = mk(cam->sky_color[0], cam->sky_color[1], cam->sky_color[2]);
    ctx.bounce_mode = cam->bounce_mode;
    ctx.shadow_mode = cam->shadow_mode;
    ctx.seed = seed; ctx.closest_casts = 0; ctx.shadow_casts = 0;
    ctx.work_closest = nullptr; ctx.work_shadow = nullptr;
    for (int64_t i = 0; i < n; i++) {
        ctx.pixel = ids ? ids[i] : (uint32_t)i;
        ctx.sample = sample;
        Ray r = input_ray(origins, dirs, i, 0);
        V3 c = cast_ray(ctx, r, recursions, 0, branches);
        rgb[3 * i] = c.x; rgb[3 * i + 1] = c.y; rgb[3 * i + 2] = c.z;
    }
    if (counts) { counts[0] = ctx.closest_casts; counts[1] = ctx.shadow_casts; counts[2] = ctx.shadow_weighted; }
}

/* canvas.cpp:39-45 */
void orc_tonemap(const float* rgb_sum, const int32_t* passes, int64_t n_pixels, float gain, uint8_t* rgb8) {
    for (int64_t i = 0; i < n_pixels; i++) {
        float denom = real_max((float)passes[i], 1.0f);
        for (int k = 0; k < 3; k++) {
            float c = rgb_sum[3 * i + k] / denom;
            rgb8[3 * i + k] = (uint8_t)real_max(0.0f, real_min(255.0f, (float)(c * gain)));
        }
    }
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox(ctr, key, out); }
double orc_uniform01(uint32_t word) { return uniform01(word); }
float orc_normal_from_word(uint32_t word) { return normal_from_word(word); }

} // extern "C"
