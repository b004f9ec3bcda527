/*
 * scene_build.cpp -- host-side scene preparation for the GPU ray caster.
 *
 * Produces, on the CPU, exactly the tree the reference builds (kdTreeNode::kdTreeNode,
 * reference kdtree.cpp:32-262) and flattens it into the arrays of pt_flat_scene.  The
 * construction differs from the reference's: instead of six std::vector<int> per node it
 * keeps six global index arrays and gives every node a contiguous segment of each, split
 * in place by a stable partition.  Because the partition is stable and the low part comes
 * first, the final by-minimum-x array is the concatenation of all leaves in left-to-right
 * order, i.e. it IS the leaf-order triangle permutation, and a leaf is just (begin, count).
 *
 * All float results that the GPU later consumes (edges, normals, plane parameters, boxes)
 * are computed here in strict binary32 in the operation order documented below; build with
 * -ffp-contract=off and without fast-math.
 *
 * 3-vector conventions (the evaluation order Eigen 3.3/3.4 uses for fixed-size 3-vectors):
 * dot = a0*b0 + (a1*b1 + a2*b2); normalize divides each component by sqrtf(dot(a,a)) and
 * leaves a zero vector unchanged.
 */
#include "pt_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#include <unistd.h>

namespace {

struct F3 { float x, y, z; };
inline F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
inline F3 sub(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 add(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline float dot3(F3 a, F3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline F3 unit(F3 a) {
    float z = dot3(a, a);
    if (z > 0.0f) { float n = sqrtf(z); return f3(a.x / n, a.y / n, a.z / n); }
    return a;
}
inline float sel_min(float x, float y) { return x < y ? x : y; } /* reference utils.h:29-31 */
inline float sel_max(float x, float y) { return x < y ? y : x; } /* reference utils.h:33-35 */

double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

/* Mesh in SoA form: what the builder and the flattener read. */
struct Mesh {
    int64_t n = 0;
    std::vector<F3> p0, p1, p2, normal, vn0, vn1, vn2; /* corners, face normal, vertex normals */
    std::vector<float> bmin[3], bmax[3];               /* triangle AABBs per axis */
};

/* Runs fn(first, last) over [0, n) cut into one contiguous range per thread. */
template <class F>
void parallel_ranges(int64_t n, int threads, F fn) {
    int64_t t = std::max<int64_t>(1, std::min<int64_t>(threads, n / 16384));
    if (t == 1) { fn((int64_t)0, n); return; }
    std::vector<std::thread> pool;
    for (int64_t k = 0; k < t; k++) pool.emplace_back([=]() { fn(n * k / t, n * (k + 1) / t); });
    for (auto& th : pool) th.join();
}

int default_threads() {
    int t = (int)std::thread::hardware_concurrency();
    return t > 0 ? t : 8;
}

/* Triangle::Triangle (reference utils.cpp:116-129). */
void mesh_from_soup(Mesh& m, const float* v, int64_t n, int threads) {
    m.n = n;
    m.p0.resize(n); m.p1.resize(n); m.p2.resize(n); m.normal.resize(n);
    for (int a = 0; a < 3; a++) { m.bmin[a].resize(n); m.bmax[a].resize(n); }
    parallel_ranges(n, threads, [&m, v](int64_t first, int64_t last) {
    for (int64_t i = first; i < last; i++) {
        const float* q = v + 9 * i;
        F3 a = f3(q[0], q[1], q[2]), b = f3(q[3], q[4], q[5]), c = f3(q[6], q[7], q[8]);
        m.p0[i] = a; m.p1[i] = b; m.p2[i] = c;
        m.normal[i] = unit(cross3(sub(b, a), sub(c, a)));
        const float pa[3] = {a.x, a.y, a.z}, pb[3] = {b.x, b.y, b.z}, pc[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; k++) {
            m.bmin[k][i] = sel_min(sel_min(pa[k], pb[k]), pc[k]);
            m.bmax[k][i] = sel_max(sel_max(pa[k], pb[k]), pc[k]);
        }
    }
    });
}

struct VKey {
    uint32_t a, b, c;
    bool operator==(const VKey& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct VKeyHash {
    size_t operator()(const VKey& k) const {
        uint64_t h = k.a * 0x9E3779B97F4A7C15ull;
        h ^= (h >> 32); h += k.b; h *= 0xC2B2AE3D27D4EB4Full; h ^= (h >> 29); h += k.c; h *= 0x165667B19E3779F9ull;
        return (size_t)(h ^ (h >> 32));
    }
};
inline uint32_t fbits(float f) {
    if (f == 0.0f) f = 0.0f; /* -0 and +0 are the same map key for the reference's '<' ordering */
    uint32_t u; memcpy(&u, &f, 4); return u;
}
inline VKey vkey(F3 p) { VKey k; k.a = fbits(p.x); k.b = fbits(p.y); k.c = fbits(p.z); return k; }

/* compute_barycentric_normals (reference stlreader.cpp:62-81): a corner's normal is the
 * normalised sum, taken in mesh order, of the face normals of all corners at the same
 * position.  Hash tables of vertex positions replace the reference's ordered map.  Each thread
 * owns the positions whose hash falls in its share (hashes and owners are computed once, in
 * parallel), scans the owners of all corners in mesh order and sums only its own, so every
 * running sum visits its contributions in the same order as the reference's per-vertex list
 * whatever the thread count. */
void vertex_normals(Mesh& m, int threads) {
    const int64_t n = m.n;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(threads, 64), n / 16384));
    std::vector<int32_t> corner((size_t)n * 3); /* index into the owner's sums */
    std::vector<uint8_t> owner((size_t)n * 3);
    std::vector<uint32_t> hash_low((size_t)n * 3);
    std::vector<std::vector<F3>> sums((size_t)T);
    parallel_ranges(n, threads, [&](int64_t first, int64_t last) {
        VKeyHash hasher;
        for (int64_t i = first; i < last; i++) {
            const F3 c[3] = {m.p0[i], m.p1[i], m.p2[i]};
            for (int k = 0; k < 3; k++) {
                const uint64_t h = (uint64_t)hasher(vkey(c[k]));
                owner[(size_t)(3 * i + k)] = (uint8_t)(((h >> 32) * (uint64_t)T) >> 32);
                hash_low[(size_t)(3 * i + k)] = (uint32_t)h;
            }
        }
    });
    const double tn0 = now_s();
    auto job = [&](int me) {
        std::vector<F3>& sum = sums[(size_t)me];
        std::vector<VKey> keys;
        size_t cap = 1024; /* open addressing, linear probing, at most half full */
        while (cap < (size_t)(n / T) * 2) cap <<= 1;
        std::vector<int32_t> slot(cap, -1);
        VKeyHash hasher;
        for (int64_t i = 0; i < n; i++) {
            const uint8_t* own = &owner[(size_t)(3 * i)];
            if (own[0] != me && own[1] != me && own[2] != me) continue;
            const F3 c[3] = {m.p0[i], m.p1[i], m.p2[i]};
            for (int k = 0; k < 3; k++) {
                if (own[k] != me) continue;
                const VKey key = vkey(c[k]);
                size_t at = (size_t)hash_low[(size_t)(3 * i + k)] & (cap - 1);
                int32_t id;
                while ((id = slot[at]) >= 0 && !(keys[(size_t)id] == key)) at = (at + 1) & (cap - 1);
                if (id < 0) {
                    id = (int32_t)keys.size();
                    keys.push_back(key);
                    sum.push_back(f3(0, 0, 0));
                    slot[at] = id;
                    if (keys.size() * 2 > cap) { /* rebuild at twice the size */
                        cap <<= 1;
                        slot.assign(cap, -1);
                        for (size_t q = 0; q < keys.size(); q++) {
                            size_t z = (size_t)hasher(keys[q]) & (cap - 1);
                            while (slot[z] >= 0) z = (z + 1) & (cap - 1);
                            slot[z] = (int32_t)q;
                        }
                    }
                }
                corner[(size_t)(3 * i + k)] = id;
                sum[(size_t)id] = add(sum[(size_t)id], m.normal[i]);
            }
        }
        for (auto& v : sum) v = unit(v);
    };
    if (T == 1) job(0);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; t++) pool.emplace_back(job, t);
        for (auto& th : pool) th.join();
    }
    const double tn1 = now_s();
    if (getenv("PT_BUILD_TIMING")) fprintf(stderr, "[normals] tables %.2f s (T=%d)\n", tn1 - tn0, T);
    m.vn0.resize(n); m.vn1.resize(n); m.vn2.resize(n);
    parallel_ranges(n, threads, [&](int64_t first, int64_t last) {
        for (int64_t i = first; i < last; i++) {
            const size_t c = (size_t)(3 * i);
            m.vn0[i] = sums[owner[c]][(size_t)corner[c]];
            m.vn1[i] = sums[owner[c + 1]][(size_t)corner[c + 1]];
            m.vn2[i] = sums[owner[c + 2]][(size_t)corner[c + 2]];
        }
    });
}

/* read_stl (reference stlreader.cpp:10-43): binary only, 80-byte header ignored, file
 * normals ignored, attribute byte count must be zero. */
bool load_stl(const char* path, std::vector<float>& soup, int threads) {
    FILE* fp = fopen(path, "rb");
    if (!fp) return false;
    char header[80];
    uint32_t count = 0;
    bool ok = fread(header, 80, 1, fp) == 1 && fread(&count, 4, 1, fp) == 1;
    if (ok && count > 0) {
        /* the records are read in pieces, every thread its own range of the file */
        const int fd = fileno(fp);
        soup.resize((size_t)count * 9);
        std::atomic<bool> good(true);
        parallel_ranges((int64_t)count, threads, [&](int64_t first, int64_t last) {
            const int64_t piece = 1 << 15; /* records per read */
            std::vector<unsigned char> buf((size_t)piece * 50);
            for (int64_t at = first; at < last && good; at += piece) {
                const int64_t m = std::min(piece, last - at);
                size_t got = 0;
                while (got < (size_t)m * 50) {
                    const ssize_t r = pread(fd, buf.data() + got, (size_t)m * 50 - got, (off_t)(84 + at * 50 + (int64_t)got));
                    if (r <= 0) { good = false; return; }
                    got += (size_t)r;
                }
                for (int64_t i = 0; i < m; i++) {
                    memcpy(&soup[(size_t)(at + i) * 9], &buf[(size_t)i * 50 + 12], 36);
                    uint16_t attr; memcpy(&attr, &buf[(size_t)i * 50 + 48], 2);
                    if (attr != 0) good = false;
                }
            }
        });
        ok = good;
    }
    fclose(fp);
    return ok;
}

/* ------------------------------------------------------- std::sort, in parallel */

/*
 * The leaf order of the reference's tree depends on where std::sort leaves equal keys
 * (kdtree.cpp:479-494), so the lists must come out exactly as libstdc++'s std::sort leaves them.
 * That algorithm is: quicksort steps (pivot = median of the second, middle and last element, swapped
 * to the front; unguarded Hoare partition of the rest) on every range longer than 16, recursing on
 * the right part and looping on the left, with a heap sort of a range whose depth budget
 * (2 * floor(log2 n)) is used up; then one insertion sort over everything.  Two facts make it
 * parallel without changing a single move:
 *   - after a partition step the two parts never exchange elements again (nothing in the right part
 *     compares less than anything in the left part, and insertion moves an element only past larger
 *     ones), so they can be finished by different threads;
 *   - the closing insertion sort is stable and therefore has one possible result per range, whichever
 *     form of insertion sort produces it and however the array is cut at partition boundaries.
 * keyed_sort() below restates the quicksort steps comparison for comparison and hands right parts of
 * at least `spawn_min` elements to new threads.  Its moves depend on the library's algorithm, so it is
 * checked once per process against std::sort itself on a tie-heavy array (keyed_sort_matches_std);
 * if the two ever differ, std::sort is used.
 */
struct Keyed { float key; int32_t index; };
inline bool key_less(const Keyed& a, const Keyed& b) { return a.key < b.key; }

inline Keyed* partition_step(Keyed* first, Keyed* last) {
    Keyed* mid = first + (last - first) / 2;
    Keyed *a = first + 1, *b = mid, *c = last - 1;
    if (key_less(*a, *b)) {
        if (key_less(*b, *c)) std::swap(*first, *b);
        else if (key_less(*a, *c)) std::swap(*first, *c);
        else std::swap(*first, *a);
    } else if (key_less(*a, *c)) std::swap(*first, *a);
    else if (key_less(*b, *c)) std::swap(*first, *c);
    else std::swap(*first, *b);
    Keyed* lo = first + 1;
    Keyed* hi = last;
    const Keyed* pivot = first;
    while (true) {
        while (key_less(*lo, *pivot)) ++lo;
        --hi;
        while (key_less(*pivot, *hi)) --hi;
        if (!(lo < hi)) return lo;
        std::swap(*lo, *hi);
        ++lo;
    }
}

void quick_steps(Keyed* first, Keyed* last, int depth) {
    while (last - first > 16) {
        if (depth == 0) { std::partial_sort(first, last, last, key_less); return; }
        --depth;
        Keyed* cut = partition_step(first, last);
        quick_steps(cut, last, depth);
        last = cut;
    }
}

inline void insertion_pass(Keyed* first, Keyed* last) {
    for (Keyed* i = first + 1; i < last; ++i) {
        const Keyed v = *i;
        Keyed* j = i;
        while (j > first && key_less(v, *(j - 1))) { *j = *(j - 1); --j; }
        *j = v;
    }
}

/* Threads that sorts have started and not yet joined, process-wide, and the most there may be. */
std::atomic<int> g_sort_threads(0);
const int MAX_SORT_THREADS = 96;

void keyed_sort_range(Keyed* first, Keyed* last, int depth, int64_t spawn_min) {
    std::vector<std::thread> kids;
    std::vector<std::pair<Keyed*, Keyed*>> mine;   /* ranges this thread finishes itself */
    while (last - first > 16) {
        if (depth == 0) { std::partial_sort(first, last, last, key_less); break; }
        --depth;
        Keyed* cut = partition_step(first, last);
        bool handed_over = false;
        if (last - cut >= spawn_min) {
            if (g_sort_threads.fetch_add(1) < MAX_SORT_THREADS) { kids.emplace_back(keyed_sort_range, cut, last, depth, spawn_min); handed_over = true; }
            else g_sort_threads.fetch_sub(1);
        }
        if (!handed_over) { quick_steps(cut, last, depth); mine.emplace_back(cut, last); }
        last = cut;
    }
    mine.emplace_back(first, last);
    for (auto& r : mine) insertion_pass(r.first, r.second);
    for (auto& t : kids) { t.join(); g_sort_threads.fetch_sub(1); }
}

int floor_log2(int64_t n) { int k = 0; while (n > 1) { n >>= 1; k++; } return k; }

void keyed_sort(Keyed* first, Keyed* last, int64_t spawn_min) {
    if (last - first > 1) keyed_sort_range(first, last, 2 * floor_log2(last - first), std::max<int64_t>(spawn_min, 64));
}

bool keyed_sort_matches_std() {
    static const bool same = []() {
        const int n = 300000;
        std::vector<Keyed> a((size_t)n), b;
        uint32_t x = 12345u;
        for (int i = 0; i < n; i++) {
            x = x * 1664525u + 1013904223u;
            const uint32_t r = x >> 8;
            /* a third distinct keys, a third from 64 values, a third from 3 values, some signed zeros */
            float k = (i % 3 == 0) ? (float)r : (i % 3 == 1) ? (float)(r & 63u) : (float)(r % 3u);
            if ((r & 1023u) == 7u) k = -0.0f;
            a[(size_t)i].key = k; a[(size_t)i].index = i;
        }
        b = a;
        std::sort(a.begin(), a.end(), key_less);
        keyed_sort(b.data(), b.data() + n, 4096);
        for (int i = 0; i < n; i++) if (a[(size_t)i].index != b[(size_t)i].index) return false;
        return true;
    }();
    return same;
}

/* ------------------------------------------------------------------ builder */

const int LEAF_THRESHOLD = 8; /* reference kdtree.cpp:12 */
const int MAXIMUM_DEPTH = 19; /* reference kdtree.cpp:13 */

struct BNode {
    int depth = 0, axis = 0;
    float split = 0;
    float bmin[3], bmax[3];
    bool leaf = false;
    int64_t begin = 0, count = 0; /* segment of the six index arrays */
    BNode* lo = nullptr;
    BNode* hi = nullptr;
    ~BNode() { delete lo; delete hi; }
};

struct Builder {
    const Mesh* m;
    std::vector<int32_t> lists[2][3]; /* [0]=by minimum, [1]=by maximum; per axis */
    std::vector<float> vals[2][3];    /* the coordinate each list is sorted by, in list order */
    std::vector<uint8_t> flags;       /* per triangle, scratch of the node that owns it: SIDE_LOW, CANDIDATE */
    std::vector<int32_t> rank_x;      /* per split candidate: its place in the owning node's by-min-x segment */
    /* Inside the builder a triangle goes by its place in an ordering of the mesh by grid cell (64^3
     * cells over the mesh's box, cells in Morton order): the lists hold those numbers and the
     * per-triangle scratch is indexed by them, so that the triangles of a node, which are neighbours
     * in space, are neighbours in memory as well (a node's passes read per-triangle bytes in list
     * order, i.e. at random).  The numbering changes no decision: the lists are sorted with the mesh's
     * own indices and translated afterwards. */
    std::vector<int32_t> orig;        /* mesh index of a builder number */
    bool plain = true;                /* no NaN among the box coordinates: every segment of a list is sorted by its
                                         value array, which the shortcuts below rely on */
    enum : uint8_t { SIDE_LOW = 1, CANDIDATE = 2 };
    int spawn_depth = 0;
    int wide_threads = 1;

    const float* coord(int mm, int axis) const { return mm == 0 ? m->bmin[axis].data() : m->bmax[axis].data(); }

    /* Best split on one axis (reference kdtree.cpp:104-145).  The reference walks the candidates
     * -- the maximum on this axis of every inc-th triangle of the by-min-x list -- in list order,
     * finds count_low = #(min <= s) and count_high = #(max > s) with two binary searches each,
     * and keeps the first candidate with the strictly smallest max(count_low, count_high).
     * Here the candidates are met in ascending order of s instead, along the by-max list, so the
     * two counts are running positions in the sorted value arrays (one pass over the node, no
     * searches); "first in list order" becomes "smallest place in the by-min-x list among equal
     * scores".  The binary searches never inspect element 0, which is harmless: s is the
     * maximum of a triangle of the node, so the smallest minimum and the smallest maximum are
     * both <= s.  build() has flagged the candidates and stored their places, so the pass reads one
     * byte per triangle and a place only for every inc-th. */
    void best_on_axis(int axis, int64_t begin, int64_t count, int inc, float& score_out, float& split_out, int& axis_out) const {
        const int32_t* imax = lists[1][axis].data() + begin;
        const float* vmax = vals[1][axis].data() + begin;
        const float* vmin = vals[0][axis].data() + begin;
        const int32_t* rk = rank_x.data();
        const uint8_t* fl = flags.data();
        (void)inc;
        int64_t n_min = 0, n_max = 0; /* #(min <= s), #(max <= s) */
        float best = score_out, best_split = split_out;
        int32_t best_place = INT32_MAX;
        for (int64_t j = 0; j < count; j++) {
            if (!(fl[imax[j]] & CANDIDATE)) continue;
            const int32_t place = rk[imax[j]];
            const float s = vmax[j];
            while (n_min < count && vmin[n_min] <= s) n_min++;
            if (n_max <= j) n_max = j + 1;
            while (n_max < count && vmax[n_max] <= s) n_max++;
            const float score = sel_max((float)(int)n_min, (float)(int)(count - n_max));
            if (score < best || (score == best && place < best_place)) { best = score; best_split = s; best_place = place; }
        }
        if (best_place != INT32_MAX) { score_out = best; split_out = best_split; axis_out = axis; }
    }

    /* Stable partition of one list segment (and of its value array): low-side triangles first,
     * order kept on both sides. */
    int64_t partition(int mm, int axis, int64_t begin, int64_t count) {
        int32_t* a = lists[mm][axis].data() + begin;
        float* v = vals[mm][axis].data() + begin;
        std::vector<int32_t> high;
        std::vector<float> high_v;
        high.reserve((size_t)count / 2 + 8);
        high_v.reserve((size_t)count / 2 + 8);
        int64_t nl = 0;
        for (int64_t i = 0; i < count; i++) {
            int32_t t = a[i];
            float x = v[i];
            if (flags[t] & SIDE_LOW) { a[nl] = t; v[nl] = x; nl++; } else { high.push_back(t); high_v.push_back(x); }
        }
        if (!high.empty()) {
            memcpy(a + nl, high.data(), high.size() * sizeof(int32_t));
            memcpy(v + nl, high_v.data(), high_v.size() * sizeof(float));
        }
        return nl;
    }

    BNode* build(int depth, int64_t begin, int64_t count) {
        BNode* n = new BNode();
        n->depth = depth; n->begin = begin; n->count = count;
        const int32_t* ours = lists[0][0].data() + begin;
        const float inf = (float)1e100; /* FLOAT_INF narrowed, reference utils.h:10 */
        /* The node's box (reference kdtree.cpp:51-62: running sel_min / sel_max over its triangles in
         * by-min-x order).  The by-min and by-max segments are sorted, so the bounds are their first and
         * last values -- as numbers.  The running form keeps the LAST of equal minima and the FIRST of
         * equal maxima, which shows only in the sign of a zero: a zero bound is recomputed the long way. */
        for (int a = 0; a < 3; a++) {
            float lo = inf, hi = -inf;
            bool have_lo = false, have_hi = false;
            if (plain && count > 0) {
                lo = vals[0][a][(size_t)begin]; hi = vals[1][a][(size_t)(begin + count - 1)];
                have_lo = lo != 0.0f; have_hi = hi != 0.0f;
            }
            if (!have_lo) { lo = inf; const float* mn = m->bmin[a].data(); for (int64_t i = 0; i < count; i++) lo = sel_min(lo, mn[orig[ours[i]]]); }
            if (!have_hi) { hi = -inf; const float* mx = m->bmax[a].data(); for (int64_t i = 0; i < count; i++) hi = sel_max(hi, mx[orig[ours[i]]]); }
            n->bmin[a] = lo; n->bmax[a] = hi;
        }
        if (count <= LEAF_THRESHOLD || depth >= MAXIMUM_DEPTH) { n->leaf = true; return n; }
        int inc = 8;                 /* reference kdtree.cpp:92-98 */
        if (count < 10000) inc = 4;
        if (count < 1000) inc = 2;
        if (count < 100) inc = 1;
        /* split candidates: every inc-th triangle of the by-min-x list (the flags of this node's
         * triangles hold only their parent's side bit at this point) */
        for (int64_t i = 0; i < count; i += inc) { rank_x[ours[i]] = (int32_t)i; flags[ours[i]] |= CANDIDATE; }
        float best_score = inf, best_split = 0.0f;
        int best_axis = -1;
        bool wide = wide_threads > 1 && count >= 200000;
        if (wide) {
            float sc[3] = {inf, inf, inf}, sp[3] = {0, 0, 0};
            int ax[3] = {-1, -1, -1};
            std::thread th[3];
            for (int a = 0; a < 3; a++) th[a] = std::thread([&, a]() { best_on_axis(a, begin, count, inc, sc[a], sp[a], ax[a]); });
            for (int a = 0; a < 3; a++) th[a].join();
            for (int a = 0; a < 3; a++) if (sc[a] < best_score) { best_score = sc[a]; best_split = sp[a]; best_axis = ax[a]; }
        } else {
            for (int a = 0; a < 3; a++) {
                float sc = inf, sp = 0.0f;
                int ax = -1;
                best_on_axis(a, begin, count, inc, sc, sp, ax);
                if (sc < best_score) { best_score = sc; best_split = sp; best_axis = ax; }
            }
        }
        n->axis = best_axis; n->split = best_split;
        /* low iff the triangle's minimum on the split axis is <= the split height; every other
         * triangle goes high (reference kdtree.cpp:172-183). */
        int64_t lo_n = 0;
        if (plain) {
            /* in the list sorted by this very minimum the low side is a prefix (so that list is
             * partitioned as it stands) */
            const int32_t* seg = lists[0][best_axis].data() + begin;
            const float* v = vals[0][best_axis].data() + begin;
            lo_n = std::upper_bound(v, v + count, best_split) - v;
            if (lo_n != count && lo_n != 0)
                for (int64_t i = 0; i < count; i++) flags[seg[i]] = i < lo_n ? SIDE_LOW : 0;
        } else {
            const float* mn = m->bmin[best_axis].data();
            for (int64_t i = 0; i < count; i++) { bool b = mn[orig[ours[i]]] <= best_split; flags[ours[i]] = b ? SIDE_LOW : 0; lo_n += b; }
        }
        if (lo_n == count || lo_n == 0) { /* reference kdtree.cpp:206-208 */
            n->leaf = true; n->axis = 0; n->split = 0;
            return n;
        }
        const int skip = plain ? best_axis : -1;   /* the by-min list of the split axis */
        if (wide) {
            std::thread th[6];
            for (int k = 0; k < 6; k++) if (k != skip) th[k] = std::thread([&, k]() { partition(k / 3, k % 3, begin, count); });
            for (int k = 0; k < 6; k++) if (k != skip) th[k].join();
        } else {
            for (int k = 0; k < 6; k++) if (k != skip) partition(k / 3, k % 3, begin, count);
        }
        if (depth < spawn_depth) {
            std::thread th([&]() { n->lo = build(depth + 1, begin, lo_n); });
            n->hi = build(depth + 1, begin + lo_n, count - lo_n);
            th.join();
        } else {
            n->lo = build(depth + 1, begin, lo_n);
            n->hi = build(depth + 1, begin + lo_n, count - lo_n);
        }
        return n;
    }

    /* Builder numbers by grid cell: remap[mesh index] = number; fills orig. */
    void number_by_cell(std::vector<int32_t>& remap, int threads) {
        const int64_t n = m->n;
        float lo[3], scale[3];
        for (int a = 0; a < 3; a++) {
            float mn = (float)1e100, mx = -(float)1e100;
            const float* c = m->bmin[a].data();
            for (int64_t i = 0; i < n; i++) { mn = c[i] < mn ? c[i] : mn; mx = c[i] > mx ? c[i] : mx; }
            lo[a] = mn;
            const float w = mx - mn;
            scale[a] = (w > 0.0f && w < (float)1e30) ? 64.0f / w : 0.0f;
        }
        std::vector<uint32_t> cell((size_t)n);
        parallel_ranges(n, threads, [&](int64_t first, int64_t last) {
            for (int64_t i = first; i < last; i++) {
                uint32_t key = 0;
                for (int a = 0; a < 3; a++) {
                    const float f = (m->bmin[a][(size_t)i] - lo[a]) * scale[a];
                    const uint32_t q = f >= 63.0f ? 63u : f >= 0.0f ? (uint32_t)f : 0u;   /* NaN -> 0 */
                    for (int b = 0; b < 6; b++) key |= ((q >> b) & 1u) << (3 * b + a);
                }
                cell[(size_t)i] = key;
            }
        });
        std::vector<int64_t> start((size_t)(1 << 18) + 1, 0);
        for (int64_t i = 0; i < n; i++) start[(size_t)cell[(size_t)i] + 1]++;
        for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
        remap.resize((size_t)n);
        orig.resize((size_t)n);
        for (int64_t i = 0; i < n; i++) {
            const int64_t at = start[(size_t)cell[(size_t)i]]++;
            remap[(size_t)i] = (int32_t)at;
            orig[(size_t)at] = (int32_t)i;
        }
    }

    /* Six index lists sorted once with std::sort and a strict '<' on one box coordinate
     * (reference kdtree.cpp:479-494, comparator :404-409).  Ties end up in whatever order
     * libstdc++'s introsort leaves for the identity input sequence, so the same call on the
     * same sequence is the only way to reproduce the reference's leaf order.  std::sort's moves
     * depend on nothing but the outcomes of its comparisons, so sorting {coordinate, index} pairs by
     * the coordinate performs the same moves as sorting the indices with the reference's indirect
     * comparator -- the same permutation, ties included -- without a random memory access per
     * comparison. */
    BNode* run(int threads) {
        int64_t n = m->n;
        const double t_begin = now_s();
        std::vector<std::thread> pool;
        std::atomic<bool> any_nan(false);
        /* each of the six sorts may use about a third of the threads */
        const bool parallel_sort = threads > 6 && n >= (1 << 20) && keyed_sort_matches_std();
        const int64_t spawn_min = std::max<int64_t>(1 << 16, n / (8 * std::max(1, threads / 3)));
        if (getenv("PT_BUILD_TIMING") && threads > 6 && !keyed_sort_matches_std()) fprintf(stderr, "[build] parallel sort differs from std::sort here: not used\n");
        for (int k = 0; k < 6; k++) {
            auto job = [this, k, n, &any_nan, parallel_sort, spawn_min]() {
                int mm = k / 3, axis = k % 3;
                const float* c = coord(mm, axis);
                std::vector<Keyed> keyed((size_t)n);
                bool nan = false;
                for (int64_t i = 0; i < n; i++) { keyed[(size_t)i].key = c[i]; keyed[(size_t)i].index = (int32_t)i; nan |= c[i] != c[i]; }
                if (nan) any_nan = true;
                if (parallel_sort) keyed_sort(keyed.data(), keyed.data() + n, spawn_min);
                else std::sort(keyed.begin(), keyed.end(), key_less);
                std::vector<int32_t>& l = lists[mm][axis];
                std::vector<float>& v = vals[mm][axis];
                l.resize((size_t)n);
                v.resize((size_t)n);
                for (int64_t i = 0; i < n; i++) { l[(size_t)i] = keyed[(size_t)i].index; v[(size_t)i] = keyed[(size_t)i].key; }
            };
            if (threads > 1) pool.emplace_back(job); else job();
        }
        std::vector<int32_t> remap;
        number_by_cell(remap, threads > 1 ? std::max(1, threads / 4) : 1);   /* beside the sorts */
        for (auto& t : pool) t.join();
        plain = !any_nan;
        pool.clear();
        for (int k = 0; k < 6; k++) {
            auto job = [this, k, n, &remap]() {
                int32_t* l = lists[k / 3][k % 3].data();
                for (int64_t i = 0; i < n; i++) l[i] = remap[(size_t)l[i]];
            };
            if (threads > 1) pool.emplace_back(job); else job();
        }
        for (auto& t : pool) t.join();
        std::vector<int32_t>().swap(remap);
        if (getenv("PT_BUILD_TIMING")) fprintf(stderr, "[build] sorts done at %.2f s\n", now_s() - t_begin);
        flags.assign((size_t)n, 0);
        rank_x.assign((size_t)n, 0);
        spawn_depth = 0;
        while ((1 << spawn_depth) < threads) spawn_depth++;
        wide_threads = threads;
        BNode* root = build(0, 0, n);
        /* the leaf-order permutation, in the mesh's indices again */
        std::vector<int32_t>& perm = lists[0][0];
        parallel_ranges(n, threads, [&](int64_t first, int64_t last) { for (int64_t i = first; i < last; i++) perm[(size_t)i] = orig[(size_t)perm[(size_t)i]]; });
        if (getenv("PT_BUILD_TIMING")) fprintf(stderr, "[build] tree done at %.2f s\n", now_s() - t_begin);
        return root;
    }
};

/* An array whose elements start out uninitialised (a std::vector would clear gigabytes on one
 * thread before the parallel pass overwrites every byte). */
template <class T>
struct RawArray {
    T* p = nullptr;
    ~RawArray() { free(p); }
    void resize(size_t n) { free(p); p = (T*)malloc(std::max<size_t>(n, 1) * sizeof(T)); if (!p) throw std::bad_alloc(); }
    T* data() { return p; }
    T& operator[](size_t i) { return p[i]; }
    RawArray() = default;
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
};

struct Flat {
    std::vector<pt_flat_node> nodes;
    RawArray<pt_tri_isect> tris;     /* every field of every record is written by flatten() */
    RawArray<pt_tri_shade> shade;
    pt_flat_scene desc;
};

uint32_t leaf_ref(const BNode* n) {
    uint32_t cnt = n->count <= (int64_t)PT_REF_COUNT_MASK ? (uint32_t)n->count : 0u;
    return PT_REF_LEAF | (cnt << PT_REF_COUNT_SHIFT) | (uint32_t)n->begin;
}

/* Interior nodes in depth-first pre-order (low subtree directly after its parent). */
uint32_t flatten_rec(const BNode* n, Flat& f, uint32_t& max_depth) {
    if (n->leaf) return leaf_ref(n);
    uint32_t me = (uint32_t)f.nodes.size();
    f.nodes.push_back(pt_flat_node());
    if ((uint32_t)n->depth + 1 > max_depth) max_depth = (uint32_t)n->depth + 1;
    uint32_t lo = flatten_rec(n->lo, f, max_depth);
    uint32_t hi = flatten_rec(n->hi, f, max_depth);
    pt_flat_node& fn = f.nodes[me];
    for (int a = 0; a < 3; a++) {
        fn.lo[a][0] = n->lo->bmin[a]; fn.lo[a][1] = n->lo->bmax[a];
        fn.hi[a][0] = n->hi->bmin[a]; fn.hi[a][1] = n->hi->bmax[a];
    }
    fn.lo_ref = lo; fn.hi_ref = hi; fn.axis = (uint32_t)n->axis; fn.pad = 0;
    return me;
}

/* `spawn` levels of the tree hand their low subtree to a thread of its own (subtrees own disjoint
 * ranges of the triangle array). */
void mark_leaf_counts(const BNode* n, Flat& f, int spawn) {
    if (n->leaf) {
        if (n->count > 0) f.tris[(size_t)n->begin].leaf_count = (uint32_t)n->count;
        for (int64_t k = 0; k < n->count; k++) f.tris[(size_t)(n->begin + k)].leaf_remaining = (uint32_t)(n->count - 1 - k);
        return;
    }
    if (spawn > 0) {
        std::thread low([n, &f, spawn]() { mark_leaf_counts(n->lo, f, spawn - 1); });
        mark_leaf_counts(n->hi, f, spawn - 1);
        low.join();
        return;
    }
    mark_leaf_counts(n->lo, f, 0);
    mark_leaf_counts(n->hi, f, 0);
}

/* Triangle::project_point_to_given_altitude's plane parameter (reference utils.cpp:167). */
inline float plane_of(F3 nrm, F3 a, F3 b, F3 c) {
    float s = (dot3(nrm, a) + dot3(nrm, b)) + dot3(nrm, c);
    return (float)((double)s / 3.0);
}

void flatten(const Mesh& m, const BNode* root, const std::vector<int32_t>& perm, Flat& f, int threads) {
    uint32_t max_depth = 0;
    f.nodes.clear();
    uint32_t root_ref = flatten_rec(root, f, max_depth);
    f.tris.resize((size_t)m.n);
    f.shade.resize((size_t)m.n);
    parallel_ranges(m.n, threads, [&m, &perm, &f](int64_t first, int64_t last) {
    for (int64_t k = first; k < last; k++) {
        int32_t i = perm[(size_t)k];
        pt_tri_isect& ti = f.tris[(size_t)k];
        F3 e1 = sub(m.p1[i], m.p0[i]), e2 = sub(m.p2[i], m.p0[i]);
        ti.p0[0] = m.p0[i].x; ti.p0[1] = m.p0[i].y; ti.p0[2] = m.p0[i].z;
        ti.e1[0] = e1.x; ti.e1[1] = e1.y; ti.e1[2] = e1.z;
        ti.e2[0] = e2.x; ti.e2[1] = e2.y; ti.e2[2] = e2.z;
        ti.mesh_index = (uint32_t)i; ti.leaf_count = 0; ti.leaf_remaining = 0;
        pt_tri_shade& ts = f.shade[(size_t)k];
        memset(&ts, 0, sizeof(ts));
        ts.normal[0] = m.normal[i].x; ts.normal[1] = m.normal[i].y; ts.normal[2] = m.normal[i].z;
        ts.plane = plane_of(m.normal[i], m.p0[i], m.p1[i], m.p2[i]);
        F3 un = sub(m.vn1[i], m.vn0[i]), vn = sub(m.vn2[i], m.vn0[i]); /* Triangle::set_normals, utils.cpp:131-135 */
        ts.base_n[0] = m.vn0[i].x; ts.base_n[1] = m.vn0[i].y; ts.base_n[2] = m.vn0[i].z;
        ts.u_n[0] = un.x; ts.u_n[1] = un.y; ts.u_n[2] = un.z;
        ts.v_n[0] = vn.x; ts.v_n[1] = vn.y; ts.v_n[2] = vn.z;
        F3 d1 = unit(e1); /* hit_triangle->edge01.normalized(), integrator.cpp:82 */
        ts.d1_x = d1.x; ts.d1_y = d1.y; ts.d1_z = d1.z;
    }
    });
    { int spawn = 0; while ((1 << spawn) < threads && spawn < 6) spawn++; mark_leaf_counts(root, f, m.n >= (1 << 18) ? spawn : 0); }
    memset(&f.desc, 0, sizeof(f.desc));
    f.desc.n_nodes = f.nodes.size();
    f.desc.n_tris = (uint64_t)m.n;
    f.desc.nodes = f.nodes.data();
    f.desc.tris = f.tris.data();
    f.desc.shade = f.shade.data();
    f.desc.root_ref = root_ref;
    f.desc.max_depth = max_depth;
    for (int a = 0; a < 3; a++) { f.desc.root_min[a] = root->bmin[a]; f.desc.root_max[a] = root->bmax[a]; }
}

void dump_rec(const BNode* n, const std::vector<int32_t>& perm, std::vector<float>* nodes, std::vector<int32_t>* leaves,
              int64_t* stats) {
    int64_t cnt = n->leaf ? n->count : 0;
    if (stats) {
        stats[0]++;
        if (n->leaf) { stats[1]++; stats[4] += cnt; }
        if (n->depth > stats[2]) stats[2] = n->depth;
        if (cnt > stats[3]) stats[3] = cnt;
    }
    if (nodes) {
        float rec[10] = {n->leaf ? 1.0f : 0.0f, (float)n->axis, n->split, n->bmin[0], n->bmin[1], n->bmin[2],
                         n->bmax[0], n->bmax[1], n->bmax[2], (float)cnt};
        nodes->insert(nodes->end(), rec, rec + 10);
    }
    if (leaves && n->leaf) leaves->insert(leaves->end(), perm.begin() + n->begin, perm.begin() + n->begin + n->count);
    if (!n->leaf) { dump_rec(n->lo, perm, nodes, leaves, stats); dump_rec(n->hi, perm, nodes, leaves, stats); }
}

} // namespace

struct pth_scene {
    Mesh mesh;
    BNode* root = nullptr;
    std::vector<int32_t> perm;
    Flat flat;
    double t_normals = 0, t_build = 0, t_flatten = 0;
    ~pth_scene() { delete root; }
};

static pth_scene* finish(pth_scene* s, int threads, pth_mesh_ready_fn mesh_ready = nullptr, void* ctx = nullptr) {
    if (threads <= 0) threads = default_threads();
    if (s->mesh.n >= (int64_t)PT_MAX_TRIANGLES) { delete s; return nullptr; }
    /* The tree needs only the triangles' boxes and the vertex normals only their positions, so
     * with more than one thread the two run side by side (half of the threads each; the
     * normals finish first and the build's own threads then have the machine). */
    double t0 = now_s(), t_normals_end = t0;
    std::thread normals;
    if (threads > 1) normals = std::thread([s, threads, &t_normals_end, mesh_ready, ctx]() {
        vertex_normals(s->mesh, std::max(1, threads / 2));
        t_normals_end = now_s();
        if (mesh_ready) mesh_ready(s, ctx);
    });
    else { vertex_normals(s->mesh, 1); t_normals_end = now_s(); if (mesh_ready) mesh_ready(s, ctx); }
    double t1 = now_s();
    Builder b;
    b.m = &s->mesh;
    s->root = b.run(threads);
    s->perm.swap(b.lists[0][0]);
    double t2 = now_s();
    if (normals.joinable()) normals.join();
    double t2b = now_s();
    flatten(s->mesh, s->root, s->perm, s->flat, threads);
    double t3 = now_s();
    s->t_normals = t_normals_end - t0; s->t_build = t2 - t1; s->t_flatten = t3 - t2b;
    return s;
}

extern "C" {

pth_scene* pth_scene_from_stl(const char* path, int threads) { return pth_scene_from_stl_with(path, threads, nullptr, nullptr); }

pth_scene* pth_scene_from_stl_with(const char* path, int threads, pth_mesh_ready_fn mesh_ready, void* ctx) {
    std::vector<float> soup;
    const int use = threads > 0 ? threads : default_threads();
    const double t0 = now_s();
    if (!path || !load_stl(path, soup, use)) return nullptr;
    const double t1 = now_s();
    pth_scene* s = new pth_scene();
    mesh_from_soup(s->mesh, soup.data(), (int64_t)(soup.size() / 9), use);
    std::vector<float>().swap(soup);
    if (getenv("PT_BUILD_TIMING")) fprintf(stderr, "[scene] file read %.2f s, triangles %.2f s\n", t1 - t0, now_s() - t1);
    s = finish(s, threads, mesh_ready, ctx);
    if (s && getenv("PT_BUILD_TIMING")) fprintf(stderr, "[scene] normals %.2f s beside tree %.2f s, flat arrays %.2f s\n", s->t_normals, s->t_build, s->t_flatten);
    return s;
}

pth_scene* pth_scene_from_triangles(const float* verts, int64_t n, int threads) {
    if (n < 0 || (n > 0 && !verts)) return nullptr;
    pth_scene* s = new pth_scene();
    mesh_from_soup(s->mesh, verts, n, threads > 0 ? threads : default_threads());
    return finish(s, threads);
}

void pth_scene_destroy(pth_scene* s) { delete s; }

int pth_parallel_sort_matches_std(void) { return keyed_sort_matches_std() ? 1 : 0; }

int64_t pth_num_triangles(const pth_scene* s) { return s->mesh.n; }

const pt_flat_scene* pth_flat_scene(const pth_scene* s) { return &s->flat.desc; }

void pth_timings(const pth_scene* s, double out[3]) { out[0] = s->t_normals; out[1] = s->t_build; out[2] = s->t_flatten; }

static void triangle_data_range(const Mesh& m, int64_t first, int64_t last, float* out) {
    for (int64_t i = first; i < last; i++) {
        F3 e1 = sub(m.p1[i], m.p0[i]), e2 = sub(m.p2[i], m.p0[i]);
        const F3 v[9] = {m.p0[i], m.p1[i], m.p2[i], e1, e2, m.normal[i], m.vn0[i], sub(m.vn1[i], m.vn0[i]), sub(m.vn2[i], m.vn0[i])};
        float* o = out + 30 * (i - first);
        for (int k = 0; k < 9; k++) { o[3 * k] = v[k].x; o[3 * k + 1] = v[k].y; o[3 * k + 2] = v[k].z; }
        o[27] = plane_of(m.normal[i], m.p0[i], m.p1[i], m.p2[i]);
        o[28] = o[29] = 0.0f;
    }
}

void pth_triangle_data(const pth_scene* s, float* out) {
    const Mesh& m = s->mesh;
    parallel_ranges(m.n, default_threads(), [&m, out](int64_t first, int64_t last) { triangle_data_range(m, first, last, out + 30 * first); });
}

void pth_triangle_data_range(const pth_scene* s, int64_t first, int64_t last, float* out) {
    if (first < 0) first = 0;
    if (last > s->mesh.n) last = s->mesh.n;
    if (first < last) triangle_data_range(s->mesh, first, last, out);
}

void pth_tree_stats(const pth_scene* s, int64_t stats[5]) {
    for (int i = 0; i < 5; i++) stats[i] = 0;
    dump_rec(s->root, s->perm, nullptr, nullptr, stats);
}

void pth_tree_dump(const pth_scene* s, float* nodes10, int32_t* leaf_indices) {
    std::vector<float> nodes;
    std::vector<int32_t> leaves;
    dump_rec(s->root, s->perm, &nodes, &leaves, nullptr);
    memcpy(nodes10, nodes.data(), nodes.size() * sizeof(float));
    if (!leaves.empty()) memcpy(leaf_indices, leaves.data(), leaves.size() * sizeof(int32_t));
}

} // extern "C"
