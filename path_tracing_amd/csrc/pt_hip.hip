/*
 * pt_hip.hip -- MI355X (gfx950) ray-casting library: kernels and the C ABI of include/pt_hip.h.
 *
 * Data in HBM
 *   nodes  : pt_flat_node[n_nodes], 64 B each, one per interior node, both child boxes inside
 *   tris   : pt_tri_isect[n_tris], 48 B each, leaf order (what the triangle test reads)
 *   shade  : pt_tri_shade[n_tris], 64 B each, same order (what shading reads after a hit)
 *   rays / hits / queues : structure-of-arrays, one float or int array per component, so a
 *            wave's 64 lanes read and write 256 contiguous bytes per component
 *
 * Kernels (all HBM/latency bound; no matrix work anywhere on this path)
 *   k_trace<CLOSEST>  closest hit, the reference's visit order reproduced exactly
 *   k_trace<ANYHIT>   light visibility for rays whose light is at least one unit away
 *   k_generate        camera rays of a batch of tile passes
 *   k_shade           surface frame, light samples, bounce ray; compacts surviving paths with
 *                     a wave ballot and one atomic per wave
 *   k_accumulate      adds the visible light samples to each path in light order
 *   k_film            adds finished paths to the film in sample order
 * Trace kernels are persistent: a fixed grid of waves pulls 64-ray chunks from an atomic
 * counter until the queue is empty; queue lengths live in device memory, so a whole batch of
 * bounces is enqueued without a host round trip.
 *
 * Build: hipcc -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950
 */
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "pt_hip.h"
#include "shade_core.h"
#include "traverse_core.h"
#include "qnode_build.h"
#include "wave_scan.h"

using namespace ptc;

/* ------------------------------------------------------------------ errors */

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

/* ------------------------------------------------------------ device views */

struct DScene {
    const char* geom;     /* nodes, then (at byte tris_off) the triangle records: ONE allocation below 4 GiB, so that
                             a lane's record address is this base plus a 32-bit offset */
    uint32_t tris_off;
    const float4* shade;
    uint32_t root_ref;
    uint32_t n_tris;
    RootBox root;
    /* quantised interior nodes of the any-hit walks (traverse_core.h "q-nodes"), in the same allocation */
    uint32_t q_off;       /* byte offset of the 32-byte q-node records */
    uint32_t leafwise_q_off; /* ... of the copy that sits behind the leafwise walks' node records (their base is leafwise_nodes) */
    uint32_t q_root;      /* the root's position for walks on q-nodes; PT_REF_NONE: the scene has none */
    uint32_t q_root_exact; /* ... for rays that must stay on exact records (not tame) */
    QGrid grid;
    /* wave-wide leaf tests (k_shadow_leafwise): a copy of the node records whose leaf references name
     * the leaf's place in `soa`, the triangles stored component by component */
    const char* leafwise_nodes;
    const float* soa;
    uint32_t leafwise_root;
};

struct RaySoA {
    const float *ox, *oy, *oz, *dx, *dy, *dz;
};
struct RaySoAW {
    float *ox, *oy, *oz, *dx, *dy, *dz;
};

#ifndef TRACE_BLOCK
#define TRACE_BLOCK 256    /* threads per workgroup of the trace kernels (128 / 256 / 512 measured: see the tuning notes) */
#endif
#define TRACE_MAX_BLOCKS_PER_CU (2048 / TRACE_BLOCK)   /* 32 waves per CU */
#define STACK_MAX_CLOSEST (3 * PT_MAX_TREE_DEPTH) /* worst case of the closest-hit walk */
#define STACK_MAX_ANYHIT (PT_MAX_TREE_DEPTH)
#ifndef STACK_LDS
#define STACK_LDS 16       /* dwords per lane kept in LDS; deeper entries spill to HBM (rare: <1 % of rays) */
#endif
#ifndef TRACE_CHUNK
#define TRACE_CHUNK 64     /* fewest rays a wave takes from the queue per atomic */
#endif
#ifndef TRACE_MAX_CHUNK_MUL
#define TRACE_MAX_CHUNK_MUL 2u /* largest chunk, in units of TRACE_CHUNK: larger chunks lengthen the tail of a launch */
#endif
#ifndef TRACE_XCD_SUPER
#define TRACE_XCD_SUPER 64u /* chunks per super-chunk: consecutive super-chunks go to the 8 XCDs in turn (0: one shared queue) */
#endif
#ifndef TRACE_SETPRIO
#define TRACE_SETPRIO 0    /* 1: a wave runs its pops and issues its record loads at raised priority; 2: everything from the round's wait to there */
#endif
#ifndef TRACE_EARLY_PEEK
#define TRACE_EARLY_PEEK 0 /* 1: the stack entries a round's pops consume are read at the top of the round */
#endif
#ifndef TRACE_MAX_POPS
#define TRACE_MAX_POPS 1   /* stack pops a lane may take in one round before it has to wait for the next
                              (1 / 2 / 3 on the full render: 6 360 / 6 280 / 6 170 Mrays/s) */
#endif
#if TRACE_EARLY_PEEK && TRACE_MAX_POPS > 1
#error "the early stack reads serve one pop per round"
#endif
#ifndef SHADE_GRID_MUL
#define SHADE_GRID_MUL 16  /* workgroups per CU of the shade kernel's grid (2 / 4 / 8 / 16: 33.7 / 33.2 / 30.6 / 29.6 ms per step on an
                              earlier kernel; 8 / 16 / 32 on the final one: 32.9 / 31.2 / 30.8 ms) */
#endif
#ifndef TRACE_MIN_CHUNKS_PER_WAVE
#define TRACE_MIN_CHUNKS_PER_WAVE 16u /* chunks grow only while every wave still gets this many on average */
#endif
/* Idle lanes that trigger a refill while the wave still has work.  A refill costs seven ray loads
 * and the start of the new rays however few lanes take part, so it pays to wait for more of them
 * (measured on the full render, 8 / 16 / 24 / 32 / 48: closest hit 160 / 155 / 157 / 164 / 161 ms,
 * any hit 176 / 161 / 154 / 153 / 169 ms). */
#ifndef REFILL_MIN_CLOSEST
#define REFILL_MIN_CLOSEST 16
#endif
#ifndef REFILL_MIN_CLOSEST_LARGE_LEAVES
#define REFILL_MIN_CLOSEST_LARGE_LEAVES 32 /* the several-triangles-per-round kernels: 8 / 16 / 32 give 376 / 352 / 343 ms on the 21.5 M-triangle scene */
#endif
#ifndef REFILL_MIN_ANY
#define REFILL_MIN_ANY 32
#endif
#ifndef TRACE_WAVES_PER_EU_3
#define TRACE_WAVES_PER_EU_3 6 /* three triangles per round: 80 registers */
#endif
#ifndef TRACE_WAVES_PER_EU_2
#define TRACE_WAVES_PER_EU_2 7 /* the two-triangles-per-round kernels hold a second record: 72 registers instead of 64 */
#endif
#ifndef TRACE_WAVES_PER_EU
#define TRACE_WAVES_PER_EU 8 /* register budget of the trace kernels: waves per SIMD the compiler must allow
                                (6 -> 8: the same 56-57 registers, but a schedule that is 2 % faster) */
#endif

/* Per-lane stack.  The first STACK_LDS entries live in LDS, dword-interleaved across the block
 * (entry i of thread t at word (i + STACK_PAD) * TRACE_BLOCK + t), so whatever the lanes' depths
 * are, a wave's 64 accesses fall in 64 different banks.  STACK_PAD unused rows below entry 0 let
 * a pop read "the top three entries" of a shallower stack without a bounds test.  Deeper entries
 * go to a per-thread column of a spill buffer in HBM, interleaved the same way across the whole
 * grid.  The common path touches LDS only: the spill path is kept in separate blocks that wait
 * for their own loads, so that a pop never has to wait for the round's outstanding global loads. */
#define STACK_PAD 2
struct LaneStack {
    uint32_t* lds;    /* &stk[STACK_PAD * TRACE_BLOCK + threadIdx.x]: entry 0 of this lane */
    uint32_t* spill;  /* &spill[global thread id] */
    uint32_t stride;  /* threads in the grid */
    __device__ __forceinline__ uint32_t slow_read(int i) const {
        if (i < 0) return 0u;
        if (i < STACK_LDS) return lds[i * TRACE_BLOCK];
        return spill[(size_t)(i - STACK_LDS) * stride];
    }
    __device__ __forceinline__ void push(int& sp, uint32_t v) {
        if (__builtin_expect(sp < STACK_LDS, 1)) lds[sp * TRACE_BLOCK] = v;
        else spill[(size_t)(sp - STACK_LDS) * stride] = v;
        sp++;
    }
    __device__ __forceinline__ void push_frame(int& sp, bool keep, uint32_t a, uint32_t b, uint32_t c) {
        if (__builtin_expect(sp + 3 <= STACK_LDS, 1)) {
            /* a is stored unconditionally and overwritten by b when it is not kept */
            lds[sp * TRACE_BLOCK] = a;
            sp += keep ? 1 : 0;
            lds[sp * TRACE_BLOCK] = b;
            lds[(sp + 1) * TRACE_BLOCK] = c;
            sp += 2;
        } else {
            if (keep) push(sp, a);
            push(sp, b);
            push(sp, c);
        }
    }
    /* the LDS part alone: what peek1 / peek3 return while sp <= STACK_LDS */
    __device__ __forceinline__ uint32_t lds_peek1(int sp) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;
        return lds[(s - 1) * TRACE_BLOCK];
    }
    __device__ __forceinline__ void lds_peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;
        e = lds[(s - 1) * TRACE_BLOCK];
        w1 = lds[(s - 2) * TRACE_BLOCK];
        w2 = lds[(s - 3) * TRACE_BLOCK];
    }
    __device__ __forceinline__ uint32_t peek1(int sp) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;
        uint32_t e = lds[(s - 1) * TRACE_BLOCK];
        if (__builtin_expect(sp > STACK_LDS, 0)) {
            e = slow_read(sp - 1);
            __builtin_amdgcn_s_waitcnt(0);   /* vmcnt(0) expcnt(0) lgkmcnt(0): the spill load is complete here */
        }
        return e;
    }
    __device__ __forceinline__ void peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;   /* keeps the reads inside the LDS array */
        e = lds[(s - 1) * TRACE_BLOCK];
        w1 = lds[(s - 2) * TRACE_BLOCK];
        w2 = lds[(s - 3) * TRACE_BLOCK];
        if (__builtin_expect(sp > STACK_LDS, 0)) {
            e = slow_read(sp - 1);
            w1 = slow_read(sp - 2);
            w2 = slow_read(sp - 3);
            __builtin_amdgcn_s_waitcnt(0);   /* the spill loads are complete here */
        }
    }
};

/* ----------------------------------------------------------------- tracing */

enum TraceMode {
    TRACE_CLOSEST_MESH = 0, /* outputs mesh index, t, u, v              (pt_raycast_closest) */
    TRACE_CLOSEST_FLAT = 1, /* outputs leaf-order position, t, u, v     (renderer)           */
    TRACE_SHADOW_ANY = 2,   /* visibility for dist >= 1; queues the others for the exact pass */
    TRACE_SHADOW_EXACT = 3, /* visibility for the queued rays: closest hit, v > dist          */
    TRACE_SHADOW_BOUNDED = 4 /* visible iff no hit with t < dist; stops at the first such hit  */
};

struct TraceArgs {
    RaySoA rays;
    const uint32_t* d_count; /* number of work items = *d_count * count_mul, in device memory */
    uint32_t count_mul;
    uint32_t* d_next_chunk;  /* atomic head of the ray queue: chunks handed out beyond one per wave; zero at launch */
    int32_t* out_idx;        /* CLOSEST_MESH: four separate arrays (the C ABI's layout) */
    float *out_t, *out_u, *out_v;
    float4* out_hit;         /* CLOSEST_FLAT: one 16-byte record {position bits, t, u, v} per ray, so a
                                retiring lane issues one store instead of four partial-line writes */
    const float* dist;       /* shadow modes */
    uint8_t* vis;            /* shadow modes */
    const uint32_t* target;  /* shadow modes, optional: ray i answers into vis[target[i]] instead of vis[i] */
    uint32_t* defer_list;    /* SHADOW_ANY appends ray indices here, SHADOW_EXACT reads them */
    uint32_t* defer_count;   /* SHADOW_ANY: atomic length of defer_list (zero at launch) */
    uint32_t* spill;         /* stack spill area: (STACK_MAX - STACK_LDS) * threads-in-grid dwords */
};

/*
 * WAVE-WIDE leaf tests (k_trace<MODE, 0>), for scenes whose leaves are large (the reference caps its
 * tree at 19 levels, kdtree.cpp:13, so tens of millions of triangles end up 40 to a leaf).
 *
 * In k_trace a lane walks its leaf alone, a few 48-byte records per round, each through per-lane
 * addresses: on such scenes the kernels are bound by the rate at which the vector-memory path takes
 * distinct lane addresses (asking a round early for the next round's lines, two more addresses per
 * lane, made them 20 % slower).  Here a lane that reaches a leaf does not test it: in the next round
 * the WAVE tests that leaf for it, one triangle per lane, from a copy of the triangles stored
 * component by component in pieces of 64 (nine rows of 64 floats per piece, a leaf of more than 64
 * triangles in consecutive pieces), so that a piece is fetched by nine loads of consecutive words at
 * one scalar base -- a few dozen cache-line lookups instead of three per triangle -- and 36 instead
 * of 48 bytes per triangle.  The pending leaves of a round are worked off one piece after the other,
 * the loads of the next two pieces in flight; the ray is read from its owner's registers.
 *
 * Same results bit for bit: the same leaves are entered through the same box tests in the same order,
 * and every triangle of an entered leaf meets the same triangle test.  An any-hit walk only asks
 * whether any of them hits.  A closest-hit walk hands the hits of a piece to the owner in triangle
 * order, each through the reference's two strict comparisons (kdtree.cpp:281-293), so the best hit and
 * the leaf's nearest hit come out as from the lane's own loop; hits are named by piece and slot and
 * translated afterwards (k_leafwise_positions).  Leaf references on the device are
 * PT_REF_LEAF | first piece << 8 | count (count <= 255) in a copy of the node array made for these kernels.
 */
#ifndef LEAFWISE_WAVES_PER_EU
#define LEAFWISE_WAVES_PER_EU 6
#endif
#define LEAFWISE_PIECE 64u                         /* triangles per piece = lanes */
#define LEAFWISE_PIECE_WORDS (9u * LEAFWISE_PIECE)
#define LEAFWISE_MAX_COUNT 255u
#define LEAFWISE_MAX_PIECES (1u << 22)
#ifndef LEAFWISE_QNODES_BY_DEFAULT
#define LEAFWISE_QNODES_BY_DEFAULT 0               /* 1: the leafwise any-hit walks of scenes beyond the Infinity Cache read quantised interior nodes */
#endif
#ifndef LEAFWISE_MIN_AVERAGE_LEAF
#define LEAFWISE_MIN_AVERAGE_LEAF 32u              /* triangles per leaf, on average, from which a scene gets these kernels
                                                      (measured on MI355X: 41 per leaf +23 %, 10 per leaf -34 %, 5.6 per leaf -54 %) */
#endif

struct LeafPiece {            /* wave-uniform: one piece (up to 64 triangles) of one pending leaf */
    uint64_t rest;            /* lanes whose leaves come after this one */
    uint32_t lane;            /* the owner */
    uint32_t piece;           /* its number in the pool */
    uint32_t left;            /* triangles of the leaf from this piece on */
    uint32_t valid;
    uint32_t dead;            /* the owner has its hit: the leaf's remaining pieces are not tested */
};

/* The piece after `p` in the round's sequence: the next 64 triangles of the same leaf, else the
 * leaf of the next lane in `rest`. */
__device__ __forceinline__ void leaf_piece_next(LeafPiece& p, uint32_t cur) {
    if (p.valid && p.left > LEAFWISE_PIECE) { p.left -= LEAFWISE_PIECE; p.piece += 1u; return; }
    if (p.rest == 0ull) { p.valid = 0u; return; }
    const uint32_t b = (uint32_t)__builtin_ctzll(p.rest);
    p.rest &= p.rest - 1ull;
    const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)cur, (int)b);
    p.lane = b;
    p.left = ref & 255u;
    p.piece = (ref >> 8) & (LEAFWISE_MAX_PIECES - 1u);
    p.valid = 1u;
    p.dead = 0u;
}

struct Tri9 { float p0x, p0y, p0z, e1x, e1y, e1z, e2x, e2y, e2z; };

/* Lane j of the wave asks for triangle j of the piece (lanes past the leaf's end repeat its last
 * triangle): one scalar base, one 32-bit lane offset, nine constant row offsets. */
__device__ __forceinline__ void leaf_piece_load(const LeafPiece& p, const float* __restrict__ soa, uint32_t lane, Tri9& t) {
    if (!p.valid) return;
    const uint32_t j = lane < p.left ? lane : p.left - 1u;
    const char* rows = (const char*)soa + (size_t)p.piece * (LEAFWISE_PIECE_WORDS * 4u) + j * 4u;
    t.p0x = *(const float*)(rows);
    t.p0y = *(const float*)(rows + 1u * LEAFWISE_PIECE * 4u);
    t.p0z = *(const float*)(rows + 2u * LEAFWISE_PIECE * 4u);
    t.e1x = *(const float*)(rows + 3u * LEAFWISE_PIECE * 4u);
    t.e1y = *(const float*)(rows + 4u * LEAFWISE_PIECE * 4u);
    t.e1z = *(const float*)(rows + 5u * LEAFWISE_PIECE * 4u);
    t.e2x = *(const float*)(rows + 6u * LEAFWISE_PIECE * 4u);
    t.e2y = *(const float*)(rows + 7u * LEAFWISE_PIECE * 4u);
    t.e2z = *(const float*)(rows + 8u * LEAFWISE_PIECE * 4u);
}

/* One piece against its owner's ray; the owner takes the outcome. */
template <int MODE>
__device__ __forceinline__ void leaf_piece_test(LeafPiece& now, const Tri9& t, Lane& L, uint32_t lane) {
    /* the whole set is waited for here, along every path (the test below reads some of its words only
     * behind its early exits): afterwards the set can be loaded anew without waiting for the other one */
    asm volatile("" :: "v"(t.p0x), "v"(t.p0y), "v"(t.p0z), "v"(t.e1x), "v"(t.e1y), "v"(t.e1z), "v"(t.e2x), "v"(t.e2y), "v"(t.e2z));
    if (now.dead) return;
    const int b = (int)now.lane;
    Ray9 r;
    r.ox = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.ox), b));
    r.oy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.oy), b));
    r.oz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.oz), b));
    r.dx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.dx), b));
    r.dy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.dy), b));
    r.dz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.r.dz), b));
    r.rx = 0.0f; r.ry = 0.0f; r.rz = 0.0f;   /* the triangle test does not use the reciprocals */
    /* any-hit: a hit counts if it is nearer than the best so far (kdtree.cpp:286): +inf, or the light's distance */
    const float bound = MODE == TRACE_SHADOW_BOUNDED ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(L.best.t), b)) : pt_inf();
    TriData tr;
    tr.p0x = t.p0x; tr.p0y = t.p0y; tr.p0z = t.p0z;
    tr.e1x = t.e1x; tr.e1y = t.e1y; tr.e1z = t.e1z;
    tr.e2x = t.e2x; tr.e2y = t.e2y; tr.e2z = t.e2z;
    tr.mesh_index = 0u; tr.leaf_remaining = 0u;
    float tt = 0.0f, uu = 0.0f, vv = 0.0f;
    constexpr bool ANY = MODE == TRACE_SHADOW_ANY || MODE == TRACE_SHADOW_BOUNDED;
    const bool mine = lane == now.lane;
    const bool last = now.left <= LEAFWISE_PIECE;
    if (ANY) {
        const bool hit = lane < now.left && tri_hit(tr, r, tt, uu, vv) && tt < bound;
        const bool any = __ballot(hit) != 0ull;
        /* written as selects: the owner's state stays in registers */
        L.found = L.found || (mine && any);
        L.cur = (mine && (any || last)) ? PT_REF_NONE : L.cur;
        L.sp = (mine && any) ? 0 : L.sp;
        if (any) now.dead = 1u;
    } else {
        /* the hits of the piece reach the owner in triangle order, each with the two strict
         * comparisons of the reference's loop (kdtree.cpp:281-293): a nearer hit replaces the best, an
         * equally near one does not */
        const bool hit = lane < now.left && tri_hit(tr, r, tt, uu, vv);
        uint64_t hm = __ballot(hit);
        while (hm) {
            const int h = __builtin_ctzll(hm);
            hm &= hm - 1ull;
            const float th = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tt), h));
            const float uh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(uu), h));
            const float vh = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), h));
            const bool better = mine && th < L.best.t;
            L.best.t = better ? th : L.best.t; L.best.u = better ? uh : L.best.u; L.best.v = better ? vh : L.best.v;
            L.best.tri = better ? ((now.piece << 6) | (uint32_t)h) : L.best.tri;   /* piece and slot: see k_leafwise_positions */
            L.found = L.found || better;
            L.local = (mine && th < L.local) ? th : L.local;
        }
        L.cur = (mine && last) ? PT_REF_NONE : L.cur;
    }
}

/*
 * Persistent trace kernel.  Every wave loops over "rounds"; in a round each lane that holds a
 * ray advances it by one step of the lane state machine (traverse_core.h): one triangle of its
 * pending leaf, or one interior node, followed by stack pops while it has neither a leaf nor a
 * node (at most TRACE_MAX_POPS per round).
 *
 * A round consumes ONE group of global loads, all in flight before any is used, so it costs one
 * memory round trip:
 *   - the rays of the lanes being refilled, and
 *   - one 64-byte record per lane: the lane's next triangle (48 bytes used) or its interior
 *     node, fetched through one per-lane address so that the same four load instructions
 *     serve both kinds of lane.
 * Stack traffic is LDS only on the common path and needs no wait on global memory.
 * TPR = 2, 3 (scenes with large leaves) adds the TPR-1 records after a lane's triangle to the
 * group and tests them in the same round while the leaf goes on; order and comparisons are unchanged.
 *
 * Lanes whose ray is finished write their result and go idle; when enough lanes are idle the
 * wave hands them the next rays of its chunk (TRACE_CHUNK..TRACE_MAX_CHUNK_MUL*TRACE_CHUNK consecutive rays,
 * one atomic on the queue head per chunk).  The loop ends when the queue is empty and every
 * lane is idle, so the grid always drains.
 */
template <int MODE, int TPR = 1, bool QN = false>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(TPR == 0 ? LEAFWISE_WAVES_PER_EU : TPR == 1 ? TRACE_WAVES_PER_EU : TPR == 2 ? TRACE_WAVES_PER_EU_2 : TRACE_WAVES_PER_EU_3))) void k_trace(DScene sc, TraceArgs a) {
    constexpr bool ANY = (MODE == TRACE_SHADOW_ANY || MODE == TRACE_SHADOW_BOUNDED);
    /* TPR = 0: a lane does not walk its leaf; the wave tests it in the next round ("WAVE-WIDE leaf
     * tests" above).  Hits are then named by piece and slot, k_leafwise_positions translates. */
    constexpr bool COOP = TPR == 0;
    /* QN: interior nodes whose children are both interior are read as 32-byte quantised records (two
     * load pieces instead of four); any-hit walks only, whose answer does not depend on which
     * enclosing boxes the interior levels are tested against (traverse_core.h "q-nodes") */
    static_assert(!QN || ANY, "q-nodes are for the any-hit walks");
    constexpr bool CLOSEST_OUT = (MODE == TRACE_CLOSEST_MESH || MODE == TRACE_CLOSEST_FLAT);
    __shared__ uint32_t stk[(STACK_LDS + STACK_PAD) * TRACE_BLOCK];
    const uint32_t n = *a.d_count * a.count_mul;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    LaneStack st;
    st.lds = stk + STACK_PAD * TRACE_BLOCK + threadIdx.x;
    st.stride = gridDim.x * TRACE_BLOCK;
    st.spill = a.spill + (blockIdx.x * TRACE_BLOCK + threadIdx.x);

    Lane L;
    L.cur = PT_REF_NONE; L.sp = 0; L.found = false; L.frame_bits = 0; L.tame = true;
    int32_t ray = -1;         /* index of the ray this lane is tracing, or -1 (idle) */
    uint32_t tgt = 0;         /* shadow modes: where this ray's answer goes (loaded with the ray) */
    float dist = 0.0f;
    uint32_t pool_next = 0, pool_end = 0; /* wave-uniform: unclaimed rays of the current chunk */
    bool exhausted = false;               /* wave-uniform: the queue has no more chunks */
    /* The ray queue is handed out in chunks of consecutive rays: TRACE_CHUNK rays for short
     * queues (finer balance in the tail, and the rays in flight stay close together), growing
     * to TRACE_MAX_CHUNK_MUL times that while every wave still gets TRACE_MIN_CHUNKS_PER_WAVE chunks on average, so
     * that a long queue does not saturate the atomic (one address sustains about 90 M/s).
     * A wave's first chunk is static (its own number in its queue, see below); later ones come
     * from a shared head, which therefore starts after the last wave's.  So a launch does not
     * open with every wave queueing on one atomic, and a launch that the static chunks cover
     * (most of the deep bounces) touches no atomic at all.
     * (Measured and rejected: reading the head before bumping it, to spare the last failing
     * atomic per wave -- a device-scope load per chunk made the kernel four times slower; and
     * chunk sizes that shrink with the work left -- faster on coherent camera rays, slower on
     * everything else.) */
    const uint32_t n_waves = gridDim.x * (TRACE_BLOCK / 64);
#if !TRACE_XCD_SUPER
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * (TRACE_BLOCK / 64) + (threadIdx.x >> 6));
#endif
    uint32_t chunk = TRACE_CHUNK;
    while (chunk < TRACE_MAX_CHUNK_MUL * TRACE_CHUNK && (uint64_t)chunk * 2u * TRACE_MIN_CHUNKS_PER_WAVE * n_waves <= n) chunk *= 2u;
    const uint32_t total_chunks = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
    bool first_chunk = true;              /* wave-uniform */
#if TRACE_XCD_SUPER
    /* Workgroups are dealt to the 8 XCDs in turn, and every XCD has its own L2.  The queue is
     * therefore cut into super-chunks of TRACE_XCD_SUPER chunks that go to the XCDs in turn:
     * neighbouring rays are traced on ONE XCD and share its L2 instead of pulling the same
     * records into all eight.  Each XCD has its own queue head; an XCD that runs dry takes from
     * the next one's.  The k-th chunk of XCD x is chunk ((k / SUPER) * 8 + x) * SUPER + k % SUPER. */
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t xcd_waves = n_waves >> 3;
    const uint32_t xcd_wave = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * (TRACE_BLOCK / 64) + (threadIdx.x >> 6));
    uint32_t from_xcd = xcd, tried = 0;   /* wave-uniform: whose head this wave currently takes from */
    /* the static chunks cover every full row of 8 super-chunks that the XCD's waves span */
    const bool static_only = total_chunks <= (xcd_waves / TRACE_XCD_SUPER) * TRACE_XCD_SUPER * 8u;
#endif
    const bool scene_tame = root_is_tame(sc.root);

    /* The loop is software-pipelined: the loads a round consumes are issued during the round
     * before it -- a lane's next record as soon as its pops are done, the rays of refilled
     * lanes once the retiring lanes are known -- and the result stores, the queue bookkeeping
     * and the start of new rays run while they are in flight. */
    float4 q0, q1, q2, q3;                    /* the lane's record for this round; meaningful where rec_* says so */
    bool rec_tri = false, rec_node = false;   /* what was loaded for this lane: a triangle / an interior node */
    /* TPR > 1 (scenes with large leaves): a lane with a pending leaf also takes the TPR-1 records
     * that follow its triangle in the array and tests them in the same round while the leaf goes on */
    float4 qx[TPR > 1 ? TPR - 1 : 1][3];
#pragma unroll
    for (int k = 0; k < TPR - 1; k++) { qx[k][0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); qx[k][1] = qx[k][0]; qx[k][2] = qx[k][0]; }
    int rec_more = 0;                         /* records loaded after the lane's triangle */
    bool starting = false;                    /* this lane's registers are receiving a new ray */
    uint32_t new_ray = 0;
    /* COOP: the lanes that held a leaf when the previous round ended, the piece the next load is for,
     * and the two register sets the pieces alternate between */
    uint64_t pending = 0ull;
    LeafPiece ahead;
    ahead.rest = 0ull; ahead.lane = 0u; ahead.piece = 0u; ahead.left = 1u; ahead.valid = 0u; ahead.dead = 0u;
    Tri9 ta, tb;
    ta.p0x = ta.p0y = ta.p0z = ta.e1x = ta.e1y = ta.e1z = ta.e2x = ta.e2y = ta.e2z = 0.0f;
    tb = ta;

    while (true) {
        /* The round's single wait: every register a load of the previous round may have targeted
         * is named as an input of an empty statement, so those loads are complete from here on
         * along every control path and nothing below waits on global memory again. */
        asm volatile("" :: "v"(q0.x), "v"(q0.y), "v"(q0.z), "v"(q0.w), "v"(q1.x), "v"(q1.y), "v"(q1.z), "v"(q1.w),
                           "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(q2.w), "v"(q3.x), "v"(q3.y), "v"(q3.z));
        asm volatile("" :: "v"(L.r.ox), "v"(L.r.oy), "v"(L.r.oz), "v"(L.r.dx), "v"(L.r.dy), "v"(L.r.dz), "v"(dist), "v"(tgt));
#pragma unroll
        for (int k = 0; k < TPR - 1; k++)
            asm volatile("" :: "v"(qx[k][0].x), "v"(qx[k][0].y), "v"(qx[k][0].z), "v"(qx[k][0].w), "v"(qx[k][1].x), "v"(qx[k][1].y), "v"(qx[k][1].z),
                               "v"(qx[k][1].w), "v"(qx[k][2].x), "v"(qx[k][2].y), "v"(qx[k][2].w));

        if (COOP) {
            asm volatile("" :: "v"(ta.p0x), "v"(ta.p0y), "v"(ta.p0z), "v"(ta.e1x), "v"(ta.e1y), "v"(ta.e1z), "v"(ta.e2x), "v"(ta.e2y), "v"(ta.e2z));
            asm volatile("" :: "v"(tb.p0x), "v"(tb.p0y), "v"(tb.p0z), "v"(tb.e1x), "v"(tb.e1y), "v"(tb.e1z), "v"(tb.e2x), "v"(tb.e2y), "v"(tb.e2z));
            /* ---- the leaves that were pending when the previous round ended, piece by piece: pieces
             * alternate between the two register sets, the first of a round in `ta`; when a set has
             * been tested, the piece after the one in the other set is loaded into it ---- */
            if (pending) {
                LeafPiece now;
                now.rest = pending; now.lane = 0u; now.piece = 0u; now.left = 1u; now.valid = 0u; now.dead = 0u;
                leaf_piece_next(now, L.cur);
                while (true) {
                    leaf_piece_test<MODE>(now, ta, L, lane);
                    leaf_piece_next(ahead, L.cur);
                    leaf_piece_load(ahead, sc.soa, lane, ta);
                    leaf_piece_next(now, L.cur);
                    if (!now.valid) break;
                    leaf_piece_test<MODE>(now, tb, L, lane);
                    leaf_piece_next(ahead, L.cur);
                    leaf_piece_load(ahead, sc.soa, lane, tb);
                    leaf_piece_next(now, L.cur);
                    if (!now.valid) break;
                }
            }
        }

#if TRACE_SETPRIO == 2
        __builtin_amdgcn_s_setprio(2);
#endif
#if TRACE_EARLY_PEEK
        /* The stack entries this round's pops will consume are asked for NOW, while the sections below
         * run: a lane that pops at the end of this round has not pushed in it (a push leaves the lane
         * with a child to enter; a lane that starts a ray begins with an empty stack), so the top of its
         * stack is already what the pop will find, and the LDS round trip is off the round's critical
         * path (the next records' addresses wait on the pops). */
        uint32_t pk_e = 0u, pk_w1 = 0u, pk_w2 = 0u;
        if (ray >= 0 && L.sp > 0) {
            if (ANY) pk_e = st.lds_peek1(L.sp);
            else st.lds_peek3(L.sp, pk_e, pk_w1, pk_w2);
        }
#endif

        /* ---- start the refilled lanes (they take their first record with everyone else below) ---- */
        bool defer = false;
        if (starting) {
            /* Every hit has v <= 1, so with dist >= 1 the reference's test `v > dist` fails for any
             * hit and the answer is hit/no-hit.  Nearer lights need the closest hit's v: those rays
             * are queued for the exact pass. */
            if (MODE == TRACE_SHADOW_ANY && !(dist >= 1.0f)) defer = true;
            else {
                ray_set_recip(L.r);
                /* with q-nodes every node position is in their numbering; rays that are not tame start
                 * on the exact root and stay on exact records */
                /* (a scene has q-nodes only if its root is interior, and interior positions are the same
                 * in the leafwise copy of the node array) */
                const uint32_t root_pos = QN ? ((scene_tame && ray_is_tame(L.r)) ? sc.q_root : sc.q_root_exact) : COOP ? sc.leafwise_root : sc.root_ref;
                lane_start(L, root_pos, sc.root, scene_tame);
                /* a triangle counts only if it is strictly nearer than the best so far: starting
                 * from the bound instead of infinity makes that "nearer than the light" */
                if (MODE == TRACE_SHADOW_BOUNDED) L.best.t = dist;
                ray = (int32_t)new_ray;
            }
        }
        if (MODE == TRACE_SHADOW_ANY) {
            const uint64_t dm = __ballot(defer);
            if (dm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.defer_count, (uint32_t)__popcll(dm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (defer) a.defer_list[base + (uint32_t)__popcll(dm & lt_mask)] = new_ray;
            }
        }
        const bool active = ray >= 0;

        /* ---- one step per lane that has its record ---- */
        if (!COOP && rec_tri) {
            TriData tr;
            tr.p0x = q0.x; tr.p0y = q0.y; tr.p0z = q0.z;
            tr.e1x = q0.w; tr.e1y = q1.x; tr.e1z = q1.y;
            tr.e2x = q1.z; tr.e2y = q1.w; tr.e2z = q2.x;
            tr.mesh_index = __float_as_uint(q2.y); tr.leaf_remaining = __float_as_uint(q2.w);
            lane_tri<ANY>(L, tr);
#pragma unroll
            for (int k = 0; k < TPR - 1; k++) {
                if (k < rec_more && lane_has_leaf(L)) {   /* the leaf goes on: its next triangle, same rules */
                    tr.p0x = qx[k][0].x; tr.p0y = qx[k][0].y; tr.p0z = qx[k][0].z;
                    tr.e1x = qx[k][0].w; tr.e1y = qx[k][1].x; tr.e1z = qx[k][1].y;
                    tr.e2x = qx[k][1].z; tr.e2y = qx[k][1].w; tr.e2z = qx[k][2].x;
                    tr.mesh_index = __float_as_uint(qx[k][2].y); tr.leaf_remaining = __float_as_uint(qx[k][2].w);
                    lane_tri<ANY>(L, tr);
                }
            }
        }
        {
            if (QN && rec_node) {
                /* q-node lanes expand their eight words into the exact record's layout, in place and
                 * back to front (no word is overwritten before it has been read), so that the step
                 * below reads one set of registers whatever the record was */
                if (L.cur < PT_Q_EXACT_BASE) {
                    NodeData t;
                    const uint32_t w[8] = {__float_as_uint(q0.x), __float_as_uint(q0.y), __float_as_uint(q0.z), __float_as_uint(q0.w),
                                           __float_as_uint(q1.x), __float_as_uint(q1.y), __float_as_uint(q1.z), __float_as_uint(q1.w)};
                    qnode_decode(w, sc.grid, t);
                    q3.x = __uint_as_float(t.lo_ref); q3.y = __uint_as_float(t.hi_ref); q3.z = __uint_as_float(t.axis);
                    q2.z = t.hi[2][0]; q2.w = t.hi[2][1]; q2.x = t.hi[1][0]; q2.y = t.hi[1][1];
                    q1.z = t.hi[0][0]; q1.w = t.hi[0][1]; q1.x = t.lo[2][0]; q1.y = t.lo[2][1];
                    q0.z = t.lo[1][0]; q0.w = t.lo[1][1]; q0.x = t.lo[0][0]; q0.y = t.lo[0][1];
                } else {
                    /* an exact record's interior children are exact records too */
                    const uint32_t lo_ref = __float_as_uint(q3.x), hi_ref = __float_as_uint(q3.y);
                    q3.x = __uint_as_float(ref_is_leaf(lo_ref) ? lo_ref : lo_ref + PT_Q_EXACT_BASE);
                    q3.y = __uint_as_float(ref_is_leaf(hi_ref) ? hi_ref : hi_ref + PT_Q_EXACT_BASE);
                }
            }
            NodeData nd;
            nd.lo[0] = pt_f2{q0.x, q0.y}; nd.lo[1] = pt_f2{q0.z, q0.w}; nd.lo[2] = pt_f2{q1.x, q1.y};
            nd.hi[0] = pt_f2{q1.z, q1.w}; nd.hi[1] = pt_f2{q2.x, q2.y}; nd.hi[2] = pt_f2{q2.z, q2.w};
            nd.lo_ref = __float_as_uint(q3.x); nd.hi_ref = __float_as_uint(q3.y); nd.axis = __float_as_uint(q3.z);
            if (__any(rec_node && !L.tame)) {   /* wave-uniform: some lane needs the NaN-exact box test */
                if (rec_node) lane_node<ANY, false>(L, nd, st);
            } else {
                if (rec_node) lane_node<ANY, true>(L, nd, st);
            }
        }

#if TRACE_SETPRIO == 1
        __builtin_amdgcn_s_setprio(2);
#endif
        /* ---- lanes left with neither a leaf nor a node take their next one from the stack ---- */
        if (active && L.cur == PT_REF_NONE && L.sp > 0) {
#if TRACE_EARLY_PEEK
            if (__builtin_expect(L.sp > STACK_LDS, 0)) {   /* the top is in the spill area */
                if (ANY) pk_e = st.peek1(L.sp); else st.peek3(L.sp, pk_e, pk_w1, pk_w2);
            }
            lane_pop_entries<ANY>(L, pk_e, pk_w1, pk_w2);
#else
            lane_pop<ANY>(L, st);
#endif
#if TRACE_MAX_POPS >= 2
            if (L.cur == PT_REF_NONE && L.sp > 0) lane_pop<ANY>(L, st);
#endif
#if TRACE_MAX_POPS >= 3
            if (L.cur == PT_REF_NONE && L.sp > 0) lane_pop<ANY>(L, st);
#endif
        }
        /* ---- next round's records (one load site: a second one for the lanes that know their
         * record before the pops would make the compiler wait for the first, since both target
         * the same registers) ---- */
        rec_tri = !COOP && active && lane_has_leaf(L);
        rec_node = active && lane_has_node(L);
        if (rec_tri || rec_node) {
            const bool rec_exact = !QN || L.cur >= PT_Q_EXACT_BASE;   /* (for a node) an exact 64-byte record */
            const uint32_t off = rec_tri ? sc.tris_off + lane_leaf_pos(L) * (uint32_t)sizeof(pt_tri_isect)
                               : !QN ? L.cur * (uint32_t)sizeof(pt_flat_node)
                               : rec_exact ? (L.cur - PT_Q_EXACT_BASE) * (uint32_t)sizeof(pt_flat_node) : (COOP ? sc.leafwise_q_off : sc.q_off) + L.cur * 32u;
            const float4* rec = (const float4*)((COOP ? sc.leafwise_nodes : sc.geom) + off);
            q0 = rec[0]; q1 = rec[1];
            if (!QN || rec_tri || rec_exact) q2 = rec[2];
            if (rec_node && rec_exact) q3 = rec[3];
            if (TPR > 1) {
                const uint32_t after = sc.n_tris - 1u - lane_leaf_pos(L);   /* records that follow in the array */
                rec_more = rec_tri ? (int)min(after, (uint32_t)(TPR - 1)) : 0;
#pragma unroll
                for (int k = 0; k < TPR - 1; k++)
                    if (k < rec_more) { qx[k][0] = rec[3 + 3 * k]; qx[k][1] = rec[4 + 3 * k]; qx[k][2] = rec[5 + 3 * k]; }
            }
        }

#if TRACE_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (COOP) {   /* the first two pieces of the leaves now pending */
            pending = __ballot(active && lane_has_leaf(L));
            ahead.rest = pending; ahead.valid = 0u;
            leaf_piece_next(ahead, L.cur);
            leaf_piece_load(ahead, sc.soa, lane, ta);
            leaf_piece_next(ahead, L.cur);
            leaf_piece_load(ahead, sc.soa, lane, tb);
        }

        /* ---- retire finished rays ---- */
        if (active && lane_finished(L)) {
            const uint32_t i = (uint32_t)ray;
            if (CLOSEST_OUT) {
                if (MODE == TRACE_CLOSEST_FLAT) {
                    float4 h;
                    h.x = __uint_as_float(L.found ? L.best.tri : 0xffffffffu);
                    h.y = L.found ? L.best.t : 0.0f; h.z = L.found ? L.best.u : 0.0f; h.w = L.found ? L.best.v : 0.0f;
                    a.out_hit[i] = h;
                } else {
                    a.out_idx[i] = L.found ? (int32_t)(COOP ? L.best.tri : L.best.mesh) : -1;
                    a.out_t[i] = L.found ? L.best.t : 0.0f;
                    a.out_u[i] = L.found ? L.best.u : 0.0f;
                    a.out_v[i] = L.found ? L.best.v : 0.0f;
                }
            } else if (ANY) {
                a.vis[a.target ? tgt : i] = L.found ? 0 : 1;
            } else {
                a.vis[a.target ? tgt : i] = (!L.found || L.best.v > dist) ? 1 : 0;
            }
            ray = -1;
        }

        /* ---- refill idle lanes: their rays are loaded now and started at the top of the next round ---- */
        const uint64_t idle = __ballot(ray < 0);
        starting = false;
        if (idle) {
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            if (!exhausted && (n_idle >= (ANY ? REFILL_MIN_ANY : TPR > 1 ? REFILL_MIN_CLOSEST_LARGE_LEAVES : REFILL_MIN_CLOSEST) || idle == ~0ull)) {
                if (pool_next >= pool_end) {
#if TRACE_XCD_SUPER
                    uint32_t c = total_chunks;
                    if (first_chunk) {
                        c = ((xcd_wave / TRACE_XCD_SUPER) * 8u + xcd) * TRACE_XCD_SUPER + xcd_wave % TRACE_XCD_SUPER;
                    } else if (!static_only) {
                        while (tried < 8u) {
                            uint32_t k = 0;
                            if (lane == 0) k = atomicAdd(a.d_next_chunk + from_xcd * 8u, 1u);
                            k = xcd_waves + __builtin_amdgcn_readfirstlane(k);
                            c = ((k / TRACE_XCD_SUPER) * 8u + from_xcd) * TRACE_XCD_SUPER + k % TRACE_XCD_SUPER;
                            if (c < total_chunks) break;
                            from_xcd = (from_xcd + 1u) & 7u; tried++;   /* that XCD's share is used up */
                            c = total_chunks;
                        }
                    }
#else
                    uint32_t c = wave_id;
                    if (!first_chunk) {
                        if (lane == 0) c = atomicAdd(a.d_next_chunk, 1u);
                        c = n_waves + __builtin_amdgcn_readfirstlane(c);
                    }
#endif
                    first_chunk = false;
                    if (c < total_chunks) {
                        const uint64_t first = (uint64_t)c * chunk;
                        pool_next = (uint32_t)first;
                        pool_end = (uint32_t)min((uint64_t)n, first + chunk);
                    } else {
                        exhausted = true;
                    }
                }
                if (!exhausted) {
                    uint32_t k = pool_next + (uint32_t)__popcll(idle & lt_mask);
                    starting = (ray < 0) && (k < pool_end);
                    new_ray = k;
                    pool_next = min(pool_end, pool_next + n_idle);
                }
            }
            if (exhausted && idle == ~0ull) break;
        }
        /* A starting lane is idle, so its ray registers are dead: the new ray is loaded straight
         * into them. */
        if (starting) {
            if (MODE == TRACE_SHADOW_EXACT) new_ray = a.defer_list[new_ray];
            if (!CLOSEST_OUT) { dist = a.dist[new_ray]; if (a.target) tgt = a.target[new_ray]; }
            L.r.ox = a.rays.ox[new_ray]; L.r.oy = a.rays.oy[new_ray]; L.r.oz = a.rays.oz[new_ray];
            L.r.dx = a.rays.dx[new_ray]; L.r.dy = a.rays.dy[new_ray]; L.r.dz = a.rays.dz[new_ray];
        }
    }
}

/*
 * Any-hit walk with DEFERRED, DENSE triangle tests (light visibility for rays whose light is at
 * least one unit away, the same answers as k_trace<TRACE_SHADOW_ANY>).
 *
 * k_trace runs a triangle section and a node section in every round, each with only the lanes that
 * are in that phase (about 30 % and 60 % of them).  The any-hit answer is order-free -- "does the
 * ray hit any triangle of any leaf whose box it hits" -- so here a lane that reaches a leaf does not
 * test its triangles: it appends items {lane, first triangle, count <= 4} to a FIFO of its wave
 * (in LDS; up to two items, eight triangles, per round) and walks on at once.  When 16 items wait,
 * ONE dense pass hands them to the wave's 64 lanes, four lanes per item, whatever lane each item's
 * ray lives in (the ray is read from its owner's registers with ds_bpermute); a hit is delivered
 * to the owner lane (a loop over the few hits of a pass, through scalar registers), which stops
 * walking.  The triangle section thus runs with most lanes busy and only every few rounds, and the
 * node section sees no lanes parked at leaves.  A ray retires when its walk is over and every item
 * it queued has been tested (items are tested in FIFO order, so "my last item's number <= items
 * done" says so), which also guarantees that no test of a previous ray can reach a lane's new ray.
 *
 * Same triangle test, same box tests, same leaves as the reference's walk; only the order of the
 * tests differs, which cannot change "hit or not".  Scenes with leaves of more than 15 triangles
 * (count not in the child reference) use k_trace.
 */
#ifndef DENSE_STACK_LDS
#define DENSE_STACK_LDS 8      /* any-hit stacks hold one reference per pending level; deeper entries spill */
#endif
#ifndef DENSE_FIFO
#define DENSE_FIFO 256u        /* items per wave (a power of two; a round adds at most 128) */
#endif
#ifndef DENSE_REFILL_MIN
#define DENSE_REFILL_MIN 24
#endif
#ifndef DENSE_WAVES_PER_EU
#define DENSE_WAVES_PER_EU 7   /* register budget: 72 */
#endif
#ifndef DENSE_WAVES_PER_EU_2
#define DENSE_WAVES_PER_EU_2 5 /* two rays per lane: register budget 102 */
#endif
#ifndef DENSE_RAYS_PER_LANE
#define DENSE_RAYS_PER_LANE 1   /* 2 was measured: 485-588 ms against 325 ms for the shadow launches of two bench steps */
#endif
#define DENSE_SPILL_ENTRIES (STACK_MAX_ANYHIT - DENSE_STACK_LDS)   /* per ray, beyond the LDS part */

/* popcount of the bits of `mask` below this lane */
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

struct AnyStack {
    uint32_t* lds;          /* entry 0 of this lane; entry i at lds[i * TRACE_BLOCK] */
    uint32_t* spill_base;   /* the grid's spill area; this thread's column is found when needed (rare) */
    __device__ __forceinline__ uint32_t* column() const { return spill_base + (blockIdx.x * TRACE_BLOCK + threadIdx.x); }
    __device__ __forceinline__ size_t stride() const { return (size_t)gridDim.x * TRACE_BLOCK; }
    __device__ __forceinline__ void push(int& sp, uint32_t v) {
        if (__builtin_expect(sp < DENSE_STACK_LDS, 1)) lds[sp * TRACE_BLOCK] = v;
        else column()[(size_t)(sp - DENSE_STACK_LDS) * stride()] = v;
        sp++;
    }
    __device__ __forceinline__ void push_frame(int&, bool, uint32_t, uint32_t, uint32_t) {}   /* closest-hit only */
    /* the LDS part alone: what peek1 / peek3 return while sp <= STACK_LDS */
    __device__ __forceinline__ uint32_t lds_peek1(int sp) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;
        return lds[(s - 1) * TRACE_BLOCK];
    }
    __device__ __forceinline__ void lds_peek3(int sp, uint32_t& e, uint32_t& w1, uint32_t& w2) const {
        const int s = sp < STACK_LDS ? sp : STACK_LDS;
        e = lds[(s - 1) * TRACE_BLOCK];
        w1 = lds[(s - 2) * TRACE_BLOCK];
        w2 = lds[(s - 3) * TRACE_BLOCK];
    }
    __device__ __forceinline__ uint32_t peek1(int sp) const {
        const int s = sp < DENSE_STACK_LDS ? sp : DENSE_STACK_LDS;
        uint32_t e = lds[(s - 1) * TRACE_BLOCK];
        if (__builtin_expect(sp > DENSE_STACK_LDS, 0)) {
            e = column()[(size_t)(sp - 1 - DENSE_STACK_LDS) * stride()];
            __builtin_amdgcn_s_waitcnt(0);
        }
        return e;
    }
    __device__ __forceinline__ void peek3(int, uint32_t&, uint32_t&, uint32_t&) const {}      /* closest-hit only */
};

/* NW = rays per lane (1 or 2).  With leaves deferred, every ray a lane holds is in the same phase
 * (a node step, then a pop), so a lane can walk two rays without choosing between them: the node
 * section simply runs once per ray.  That doubles the work a wave does per memory round trip,
 * which the one-ray form lacks (it executes fewer instructions than k_trace but waits for its loads
 * half of the time). */
template <int NW>
__global__ __launch_bounds__(TRACE_BLOCK) __attribute__((amdgpu_waves_per_eu(NW == 1 ? DENSE_WAVES_PER_EU : DENSE_WAVES_PER_EU_2))) void k_shadow_dense(DScene sc, TraceArgs a) {
    constexpr uint32_t FIFO = NW == 1 ? DENSE_FIFO : 2u * DENSE_FIFO;
    __shared__ uint32_t stk[NW * DENSE_STACK_LDS * TRACE_BLOCK];
    /* The FIFO carries values from one lane of a wave to another.  The lanes of a wave run in
     * lockstep and LDS serves a wave's accesses in order, so no barrier is needed -- but the compiler
     * reasons per thread, so every access is volatile. */
    __shared__ uint64_t s_fifo[TRACE_BLOCK / 64][FIFO];   /* {lane | ray-of-lane << 6 | count << 8, first triangle} */
    const uint32_t n = *a.d_count * a.count_mul;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = threadIdx.x >> 6;
    AnyStack st[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) { st[w].lds = stk + w * DENSE_STACK_LDS * TRACE_BLOCK + threadIdx.x; st[w].spill_base = a.spill + (size_t)w * DENSE_SPILL_ENTRIES * gridDim.x * TRACE_BLOCK; }
    volatile uint64_t* const fifo = s_fifo[wid];

    Lane L[NW];
    bool active[NW];                      /* the lane's w-th ray exists */
    uint32_t out_index[NW];               /* where the ray's answer goes */
    float dist[NW];
    uint32_t my_last[NW];                 /* number of the item after this ray's last one (0: none queued) */
    float4 q0[NW], q1[NW], q2[NW], q3[NW];   /* walker: the node record of this round */
    bool rec_node[NW], starting[NW];
    uint32_t new_ray[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) {
        L[w].cur = PT_REF_NONE; L[w].sp = 0; L[w].found = false; L[w].frame_bits = 0; L[w].tame = true;
        L[w].best.t = 0.0f; L[w].best.u = 0.0f; L[w].best.v = 0.0f; L[w].best.tri = 0u; L[w].best.mesh = 0u; L[w].local = 0.0f;
        active[w] = false; out_index[w] = 0; dist[w] = 0.0f; my_last[w] = 0; rec_node[w] = false; starting[w] = false; new_ray[w] = 0;
    }
    /* wave-uniform queue state: items are numbered from 0 as they are queued */
    uint32_t q_head = 0, q_tail = 0;      /* first item not yet handed out / next item number */
    uint32_t items_done = 0;              /* items whose tests have been carried out */
    uint32_t done_after_batch = 0;        /* ... once the batch in flight has been */
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;

    /* the ray queue is dealt to the XCDs exactly as in k_trace */
    const uint32_t n_waves = gridDim.x * (TRACE_BLOCK / 64);
    uint32_t chunk = TRACE_CHUNK;
    while (chunk < TRACE_MAX_CHUNK_MUL * TRACE_CHUNK && (uint64_t)chunk * 2u * TRACE_MIN_CHUNKS_PER_WAVE * n_waves <= n) chunk *= 2u;
    const uint32_t total_chunks = (uint32_t)(((uint64_t)n + chunk - 1) / chunk);
    bool first_chunk = true;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t xcd_waves = n_waves >> 3;
    const uint32_t xcd_wave = __builtin_amdgcn_readfirstlane((blockIdx.x >> 3) * (TRACE_BLOCK / 64) + (threadIdx.x >> 6));
    uint32_t from_xcd = xcd, tried = 0;
    const bool static_only = total_chunks <= (xcd_waves / TRACE_XCD_SUPER) * TRACE_XCD_SUPER * 8u;
    const bool scene_tame = root_is_tame(sc.root);

    float4 t0, t1; float t2x = 0.0f;      /* tester: the triangle record of this round's test (nine floats) */
    bool test_valid = false;              /* this lane carries one test of the batch in flight */
    uint32_t test_owner = lane;           /* lane (| ray-of-lane << 6) whose ray the test belongs to */
    uint32_t rounds = 0;

    while (true) {
        /* the round's single wait on global memory (see k_trace) */
#pragma unroll
        for (int w = 0; w < NW; w++) {
            asm volatile("" :: "v"(q0[w].x), "v"(q0[w].y), "v"(q0[w].z), "v"(q0[w].w), "v"(q1[w].x), "v"(q1[w].y), "v"(q1[w].z), "v"(q1[w].w),
                               "v"(q2[w].x), "v"(q2[w].y), "v"(q2[w].z), "v"(q2[w].w), "v"(q3[w].x), "v"(q3[w].y), "v"(q3[w].z));
            asm volatile("" :: "v"(L[w].r.ox), "v"(L[w].r.oy), "v"(L[w].r.oz), "v"(L[w].r.dx), "v"(L[w].r.dy), "v"(L[w].r.dz), "v"(dist[w]), "v"(out_index[w]));
        }
        asm volatile("" :: "v"(t0.x), "v"(t0.y), "v"(t0.z), "v"(t0.w), "v"(t1.x), "v"(t1.y), "v"(t1.z), "v"(t1.w), "v"(t2x));
        /* a mis-scheduled queue must not hang the GPU: no launch needs anywhere near this many rounds */
        if (++rounds > (1u << 22)) break;

#pragma unroll
        for (int w = 0; w < NW; w++) {
            /* ---- start the refilled rays ---- */
            bool defer = false;
            if (starting[w]) {
                if (!(dist[w] >= 1.0f)) defer = true;     /* nearer lights need the closest hit's v: the exact pass */
                else {
                    ray_set_recip(L[w].r);
                    lane_start(L[w], sc.root_ref, sc.root, scene_tame);
                    active[w] = true;
                    my_last[w] = 0;
                }
            }
            const uint64_t dm = __ballot(defer);
            if (dm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(a.defer_count, (uint32_t)__popcll(dm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (defer) a.defer_list[base + lanes_below(dm)] = new_ray[w];
            }

            /* ---- one node step per ray that has its record ---- */
            NodeData nd;
            nd.lo[0] = pt_f2{q0[w].x, q0[w].y}; nd.lo[1] = pt_f2{q0[w].z, q0[w].w}; nd.lo[2] = pt_f2{q1[w].x, q1[w].y};
            nd.hi[0] = pt_f2{q1[w].z, q1[w].w}; nd.hi[1] = pt_f2{q2[w].x, q2[w].y}; nd.hi[2] = pt_f2{q2[w].z, q2[w].w};
            nd.lo_ref = __float_as_uint(q3[w].x); nd.hi_ref = __float_as_uint(q3[w].y); nd.axis = __float_as_uint(q3[w].z);
            if (__any(rec_node[w] && !L[w].tame)) {
                if (rec_node[w]) lane_node<true, false>(L[w], nd, st[w]);
            } else {
                if (rec_node[w]) lane_node<true, true>(L[w], nd, st[w]);
            }
        }

        /* ---- leaves are queued, not tested: up to two items of up to four triangles per ray and round ---- */
        {
            const bool room = FIFO - (q_tail - q_head) >= 128u * NW;     /* wave-uniform; otherwise the rays wait a round */
#pragma unroll
            for (int w = 0; w < NW; w++) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const bool put = active[w] && lane_has_leaf(L[w]) && room;
                    const uint64_t pm = __ballot(put);
                    if (pm) {
                        if (put) {
                            const uint32_t left = ref_leaf_count(L[w].cur), take = left < 4u ? left : 4u;
                            const uint32_t item = q_tail + lanes_below(pm);
                            fifo[item & (FIFO - 1u)] = (uint64_t)(lane | ((uint32_t)w << 6) | (take << 8)) | ((uint64_t)ref_leaf_start(L[w].cur) << 32);
                            my_last[w] = item + 1u;
                            /* the rest of the leaf: same reference, four triangles further, four fewer */
                            L[w].cur = left > 4u ? L[w].cur + 4u - (4u << PT_REF_COUNT_SHIFT) : PT_REF_NONE;
                        }
                        q_tail += (uint32_t)__popcll(pm);
                    }
                }
            }
        }

        /* ---- rays with nowhere to go take their next position from the stack ---- */
#pragma unroll
        for (int w = 0; w < NW; w++)
            if (active[w] && L[w].cur == PT_REF_NONE && L[w].sp > 0) lane_pop<true>(L[w], st[w]);

        /* ---- choose the next dense batch and start reading its items ---- */
        bool go = false;
        uint64_t item_word = 0;
        {
            const uint32_t n_items = q_tail - q_head;
            uint32_t n_walking = 0, n_resting = 0;
#pragma unroll
            for (int w = 0; w < NW; w++) {
                n_walking += (uint32_t)__popcll(__ballot(active[w] && (L[w].cur != PT_REF_NONE || L[w].sp > 0)));
                n_resting += (uint32_t)__popcll(__ballot(!active[w] || (L[w].cur == PT_REF_NONE && L[w].sp == 0)));
            }
            /* a full pass, or rays are waiting for results that only a pass brings: nobody walks any
             * more, or so many rays wait (or are idle) that the wave should refill */
            go = n_items > 0u && (n_items >= 16u || n_walking == 0u || n_resting >= (uint32_t)(DENSE_REFILL_MIN * NW));
            if (go && (lane >> 2) < n_items) item_word = fifo[(q_head + (lane >> 2)) & (FIFO - 1u)];
        }

        /* ---- the dense triangle pass: one test per lane, for whichever ray it belongs to ---- */
        if (__any(test_valid)) {
            /* every lane takes part in the exchange (a lane without a test reads its own ray) */
            const int src = (int)((test_owner & 63u) << 2);
            Ray9 r;
            r.ox = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.ox)));
            r.oy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.oy)));
            r.oz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.oz)));
            r.dx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.dx)));
            r.dy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.dy)));
            r.dz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[0].r.dz)));
            if (NW > 1) {
                const bool second = (test_owner & 64u) != 0u;
                const float bx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.ox)));
                const float by = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.oy)));
                const float bz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.oz)));
                const float cx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.dx)));
                const float cy = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.dy)));
                const float cz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(L[NW - 1].r.dz)));
                r.ox = second ? bx : r.ox; r.oy = second ? by : r.oy; r.oz = second ? bz : r.oz;
                r.dx = second ? cx : r.dx; r.dy = second ? cy : r.dy; r.dz = second ? cz : r.dz;
            }
            r.rx = 0.0f; r.ry = 0.0f; r.rz = 0.0f;   /* the triangle test does not use the reciprocals */
            bool hit = false;
            if (test_valid) {
                TriData tr;
                tr.p0x = t0.x; tr.p0y = t0.y; tr.p0z = t0.z;
                tr.e1x = t0.w; tr.e1y = t1.x; tr.e1z = t1.y;
                tr.e2x = t1.z; tr.e2y = t1.w; tr.e2z = t2x;
                tr.mesh_index = 0u; tr.leaf_remaining = 0u;
                float tt, uu, vv;
                /* a hit counts if it is nearer than the best so far (kdtree.cpp:286): +inf as long as the walk goes on */
                hit = tri_hit(tr, r, tt, uu, vv) && tt < pt_inf();
            }
            /* each hit goes to its owner, which stops walking (a pass has few hits) */
            uint64_t hm = __ballot(hit);
            while (hm) {
                const int b = __builtin_ctzll(hm);
                hm &= hm - 1ull;
                const uint32_t owner = (uint32_t)__builtin_amdgcn_readlane((int)test_owner, b);
#pragma unroll
                for (int w = 0; w < NW; w++)
                    if (lane == (owner & 63u) && (owner >> 6) == (uint32_t)w) { L[w].found = true; L[w].cur = PT_REF_NONE; L[w].sp = 0; }
            }
            test_valid = false;
            items_done = done_after_batch;
        }

        /* ---- next round's node records ---- */
#pragma unroll
        for (int w = 0; w < NW; w++) {
            rec_node[w] = active[w] && lane_has_node(L[w]);
            if (rec_node[w]) {
                const float4* rec = (const float4*)(sc.geom + L[w].cur * (uint32_t)sizeof(pt_flat_node));
                q0[w] = rec[0]; q1[w] = rec[1]; q2[w] = rec[2]; q3[w] = rec[3];
            }
        }

        /* ---- the next batch's triangle records: four lanes per item ---- */
        if (go) {
            const uint32_t n_items = q_tail - q_head;
            const uint32_t taken = n_items < 16u ? n_items : 16u;
            const uint32_t w_lo = (uint32_t)item_word, w_hi = (uint32_t)(item_word >> 32);
            test_valid = (lane >> 2) < taken && (lane & 3u) < ((w_lo >> 8) & 7u);
            test_owner = test_valid ? (w_lo & 127u) : lane;
            if (test_valid) {
                const uint32_t tri = w_hi + (lane & 3u);
                const float4* rec = (const float4*)(sc.geom + sc.tris_off + tri * (uint32_t)sizeof(pt_tri_isect));
                t0 = rec[0]; t1 = rec[1]; t2x = ((const float*)rec)[8];
            }
            q_head += taken;
            done_after_batch = q_head;
        }

        /* ---- retire: the walk is over and every queued item of the ray has been tested ---- */
        uint64_t idle_mask[NW];
        uint32_t n_idle = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (active[w] && L[w].cur == PT_REF_NONE && L[w].sp == 0 && my_last[w] <= items_done) {
                a.vis[out_index[w]] = L[w].found ? 0 : 1;
                active[w] = false;
            }
            idle_mask[w] = __ballot(!active[w]);
            n_idle += (uint32_t)__popcll(idle_mask[w]);
            starting[w] = false;
        }

        /* ---- refill idle ray slots (as k_trace) ---- */
        if (n_idle) {
            const bool all_idle = n_idle == 64u * NW;
            if (!exhausted && (n_idle >= (uint32_t)(DENSE_REFILL_MIN * NW) || all_idle)) {
                if (pool_next >= pool_end) {
                    uint32_t c = total_chunks;
                    if (first_chunk) {
                        c = ((xcd_wave / TRACE_XCD_SUPER) * 8u + xcd) * TRACE_XCD_SUPER + xcd_wave % TRACE_XCD_SUPER;
                    } else if (!static_only) {
                        while (tried < 8u) {
                            uint32_t k = 0;
                            if (lane == 0) k = atomicAdd(a.d_next_chunk + from_xcd * 8u, 1u);
                            k = xcd_waves + __builtin_amdgcn_readfirstlane(k);
                            c = ((k / TRACE_XCD_SUPER) * 8u + from_xcd) * TRACE_XCD_SUPER + k % TRACE_XCD_SUPER;
                            if (c < total_chunks) break;
                            from_xcd = (from_xcd + 1u) & 7u; tried++;
                            c = total_chunks;
                        }
                    }
                    first_chunk = false;
                    if (c < total_chunks) {
                        const uint64_t first = (uint64_t)c * chunk;
                        pool_next = (uint32_t)first;
                        pool_end = (uint32_t)min((uint64_t)n, first + chunk);
                    } else {
                        exhausted = true;
                    }
                }
                if (!exhausted) {
                    uint32_t before = 0;     /* idle slots of the rays-of-lane already served */
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const uint32_t k = pool_next + before + lanes_below(idle_mask[w]);
                        starting[w] = !active[w] && (k < pool_end);
                        new_ray[w] = k;
                        before += (uint32_t)__popcll(idle_mask[w]);
                    }
                    pool_next = min(pool_end, pool_next + n_idle);
                }
            }
            if (exhausted && all_idle) break;
        }
#pragma unroll
        for (int w = 0; w < NW; w++) {
            if (starting[w]) {
                const uint32_t i = new_ray[w];
                dist[w] = a.dist[i];
                out_index[w] = a.target ? a.target[i] : i;
                L[w].r.ox = a.rays.ox[i]; L[w].r.oy = a.rays.oy[i]; L[w].r.oz = a.rays.oz[i];
                L[w].r.dx = a.rays.dx[i]; L[w].r.dy = a.rays.dy[i]; L[w].r.dz = a.rays.dz[i];
            }
        }
    }
}

/* ------------------------------------------------------- wavefront renderer */

struct DLight { V3f pos, color; };

/* One entry of a batch: n_samples consecutive samples of one rectangle.  The rectangle is cut
 * into 8x8-pixel blocks (`padded` = 64 * blocks path slots per sample) and every block into
 * square units of side 2^side_log2 pixels (8x8, 4x4, 2x2 or single pixels).  Paths are numbered
 * unit by unit, inside a unit sample by sample, the unit's pixels innermost:
 *     path0 + (unit * n_samples + s) * pixels_per_unit + pixel_in_unit
 * so the 64 lanes of a wave start out as a few samples of one small screen patch, and
 * consecutive waves -- which run at the same time, on the same XCD -- work on the SAME patch: they
 * walk the same part of the tree and find its records in L1/L2.  (Numbering sample by sample
 * instead, every pixel of sample s before any of sample s+1, sends each wave's neighbours to
 * other parts of the screen.)  The film sums a pixel's samples in sample order whatever the
 * numbering, so the picture does not depend on it. */
struct SubPass {
    int32_t x, y, w, h;
    uint32_t first_sample, n_samples;
    uint32_t blocks_x, padded; /* ceil(w/8); 64*ceil(w/8)*ceil(h/8) */
    uint32_t path0;
    uint32_t group_first;      /* first entry in the batch with the same rectangle */
    uint32_t side_log2;        /* 0..3: units of 1, 2x2, 4x4, 8x8 pixels */
    uint32_t pad;
};

/* Where path `local` (0 .. padded*n_samples-1) of an entry sits: its sample number within the
 * entry and its slot q = block*64 + ly*8 + lx among the entry's padded pixels. */
__device__ __forceinline__ void path_to_pixel(const SubPass& s, uint32_t local, uint32_t& si, uint32_t& q) {
    const uint32_t sl = s.side_log2, P = 1u << (2u * sl);
    const uint32_t per_unit = s.n_samples << (2u * sl);
    const uint32_t unit = local / per_unit, rem = local - unit * per_unit;
    si = rem >> (2u * sl);
    const uint32_t pix = rem & (P - 1u);
    const uint32_t units_log2 = 6u - 2u * sl;                    /* units per 8x8 block, log2 */
    const uint32_t blk = unit >> units_log2, sub = unit & ((1u << units_log2) - 1u);
    const uint32_t row_log2 = 3u - sl;                           /* units per block row, log2 */
    const uint32_t lx = ((sub & ((1u << row_log2) - 1u)) << sl) + (pix & ((1u << sl) - 1u));
    const uint32_t ly = ((sub >> row_log2) << sl) + (pix >> sl);
    q = (blk << 6) + (ly << 3) + lx;
}

/* The inverse for one pixel slot: path of sample si is base + si * stride. */
__device__ __forceinline__ void pixel_to_path(const SubPass& s, uint32_t q, uint32_t& base, uint32_t& stride) {
    const uint32_t sl = s.side_log2;
    const uint32_t blk = q >> 6, lx = q & 7u, ly = (q >> 3) & 7u;
    const uint32_t row_log2 = 3u - sl, units_log2 = 6u - 2u * sl;
    const uint32_t sub = ((ly >> sl) << row_log2) + (lx >> sl);
    const uint32_t pix = ((ly & ((1u << sl) - 1u)) << sl) + (lx & ((1u << sl) - 1u));
    const uint32_t unit = (blk << units_log2) + sub;
    stride = 1u << (2u * sl);
    base = s.path0 + ((unit * s.n_samples) << (2u * sl)) + pix;
}

struct RenderBuffers {
    /* two ray queues (current / next) */
    RaySoAW q[2];
    uint32_t* qpath[2];
    /* hits of the current queue */
    float4* hit; /* {leaf-order position bits or ~0, t, u, v} */
    /* per path */
    uint32_t *pixel, *sample;
    float *rad_r, *rad_g, *rad_b, *thr;
    /* per light sample of the current depth: slot*n_lights + l */
    RaySoAW sh;
    float* sh_dist;
    /* per light sample (entry slot*n_lights + l): what it adds if its light is visible, and whether
     * it is.  Two sets, used by even and odd depths in turn: the samples of depth d are added to
     * their paths by k_shade of depth d+1, which is writing the samples of depth d+1 meanwhile. */
    float *sh_wr[2], *sh_wg[2], *sh_wb[2];
    uint8_t* sh_vis[2];
    uint32_t* sh_target;  /* per traced shadow ray: the light-sample entry it answers for */
    uint32_t* sh_defer;
    /* counters: [0..11] queue length per depth, [24..35] shadow rays per depth that need the
     * exact pass, [36] camera samples, [64..] the queue heads of every trace launch of the batch,
     * then the number of traced shadow rays per depth, each in a cache line of its own (k_shade
     * bumps it and the next queue's length from every workgroup: on one line the two would queue
     * behind each other) */
    uint32_t* cnt;
};
#define CNT_QUEUE 0
#define CNT_DEFER 24
#define CNT_SAMPLES 36
#define CNT_CHUNKS 64            /* eight queue heads (one per XCD, 32 bytes apart) per (kind, depth): 64 + (kind*12 + depth)*64 */
#define CNT_CHUNK(kind, depth) (CNT_CHUNKS + ((kind) * 12 + (depth)) * 64)
#define CNT_SHADOW_BASE (CNT_CHUNKS + 3 * 12 * 64)
#define CNT_SHADOW(depth) (CNT_SHADOW_BASE + (depth) * 32)
#define CNT_WORDS (CNT_SHADOW_BASE + 12 * 32)

__device__ __forceinline__ uint32_t find_subpass(const SubPass* sp, uint32_t n_sp, uint32_t p) {
    uint32_t lo = 0, hi = n_sp;
    while (lo + 1 < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (sp[mid].path0 <= p) lo = mid; else hi = mid;
    }
    return lo;
}

#ifndef STREAM_BLOCK
#define STREAM_BLOCK 512
#endif
#ifndef PATH_UNIT_SIDE_LOG2
#define PATH_UNIT_SIDE_LOG2 2u  /* default side of a path-numbering unit: 2^2 = 4 pixels (8 / 4 / 2 / 1 on the full render:
                                   5 590 / 5 627 / 5 621 / 5 230 Mrays/s; sample-major numbering before: 5 558) */
#endif

/* Workgroup-wide stream compaction: every thread calls it; threads with `flag` get distinct
 * consecutive slots starting at a base taken from *counter with ONE atomic per workgroup (a
 * single counter sustains only ~90 M atomics per second, which one atomic per wave would
 * exceed at these queue sizes).  s_wave needs STREAM_BLOCK/64 + 1 words of LDS. */
__device__ __forceinline__ uint32_t block_compact(bool flag, uint32_t* counter, uint32_t* s_wave) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    constexpr uint32_t NW = STREAM_BLOCK / 64;
    const uint64_t mask = __ballot(flag);
    __syncthreads(); /* previous use of s_wave is over */
    if (lane == 0) s_wave[wid] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (uint32_t w = 0; w < NW; w++) { uint32_t t = s_wave[w]; s_wave[w] = total; total += t; }
        s_wave[NW] = total ? atomicAdd(counter, total) : 0u;
        s_wave[NW + 1] = total;
    }
    __syncthreads();
    return s_wave[NW] + s_wave[wid] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
/* After block_compact: the first slot the workgroup got and how many it got. */
#define BLOCK_COMPACT_WORDS (STREAM_BLOCK / 64 + 2)
__device__ __forceinline__ uint32_t block_compact_base(const uint32_t* s_wave) { return s_wave[STREAM_BLOCK / 64]; }
__device__ __forceinline__ uint32_t block_compact_total(const uint32_t* s_wave) { return s_wave[STREAM_BLOCK / 64 + 1]; }

/* Camera rays of a batch into queue 0.  A workgroup takes GEN_ROWS rows of STREAM_BLOCK
 * consecutive paths and reserves their queue slots with ONE atomic: with one atomic per row the
 * kernel ran at the rate of that counter (about 90 M atomics per second on one address), twice
 * its arithmetic time. */
#define GEN_ROWS 8
__global__ __launch_bounds__(STREAM_BLOCK) void k_generate(CamParams cam, const SubPass* sp, uint32_t n_sp, uint32_t n_paths,
                                                            uint64_t seed, RenderBuffers b) {
    __shared__ uint32_t s_cnt[GEN_ROWS][STREAM_BLOCK / 64];
    __shared__ uint32_t s_base;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    bool valid[GEN_ROWS];
    int px[GEN_ROWS], py[GEN_ROWS];
    uint32_t sample[GEN_ROWS], off[GEN_ROWS];
#pragma unroll
    for (int r = 0; r < GEN_ROWS; r++) {
        const uint64_t p64 = ((uint64_t)blockIdx.x * GEN_ROWS + (uint64_t)r) * STREAM_BLOCK + threadIdx.x;
        valid[r] = false; px[r] = 0; py[r] = 0; sample[r] = 0;
        if (p64 < n_paths) {
            const uint32_t p = (uint32_t)p64;
            const SubPass s = sp[find_subpass(sp, n_sp, p)];
            uint32_t si, q;
            path_to_pixel(s, p - s.path0, si, q);
            uint32_t blk = q >> 6, lx = q & 7u, ly = (q >> 3) & 7u;
            int bx = (int)(blk % s.blocks_x), by = (int)(blk / s.blocks_x);
            int rx = bx * 8 + (int)lx, ry = by * 8 + (int)ly;
            valid[r] = rx < s.w && ry < s.h;
            px[r] = s.x + rx; py[r] = s.y + ry;
            sample[r] = s.first_sample + si;
            b.pixel[p] = valid[r] ? (uint32_t)(py[r] * cam.width + px[r]) : 0xffffffffu;
            b.sample[p] = sample[r];
            b.rad_r[p] = 0.0f; b.rad_g[p] = 0.0f; b.rad_b[p] = 0.0f; b.thr[p] = 1.0f;
        }
        /* compact valid paths (rectangles that are not multiples of 8 pixels have padding) into queue 0 */
        const uint64_t m = __ballot(valid[r]);
        off[r] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[r][wid] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (int r = 0; r < GEN_ROWS; r++)
            for (uint32_t w = 0; w < STREAM_BLOCK / 64; w++) { uint32_t t = s_cnt[r][w]; s_cnt[r][w] = total; total += t; }
        s_base = total ? atomicAdd(&b.cnt[CNT_QUEUE + 0], total) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < GEN_ROWS; r++) {
        if (valid[r]) {
            const uint32_t p = (blockIdx.x * GEN_ROWS + (uint32_t)r) * STREAM_BLOCK + threadIdx.x;
            const uint32_t slot = s_base + s_cnt[r][wid] + off[r];
            V3f o, d;
            camera_ray_jittered(cam, px[r], py[r], sample[r], seed, o, d);
            b.q[0].ox[slot] = o.x; b.q[0].oy[slot] = o.y; b.q[0].oz[slot] = o.z;
            b.q[0].dx[slot] = d.x; b.q[0].dy[slot] = d.y; b.q[0].dz[slot] = d.z;
            b.qpath[0][slot] = p;
        }
    }
}

/* Path state for rays supplied by the caller (pt_cast_rays): path p is ray p, its random stream
 * is keyed by ids[p] (or p) in place of a pixel number. */
__global__ __launch_bounds__(256) void k_seed_paths(uint32_t n, const uint32_t* ids, uint32_t sample, RenderBuffers b) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    b.pixel[p] = ids ? ids[p] : p;
    b.sample[p] = sample;
    b.rad_r[p] = 0.0f; b.rad_g[p] = 0.0f; b.rad_b[p] = 0.0f; b.thr[p] = 1.0f;
    b.qpath[0][p] = p;
}

/* Camera rays only, row-major over a rectangle (parity checks of the generate step). */
__global__ __launch_bounds__(256) void k_camera_rays(CamParams cam, int x0, int y0, int w, int h, int jittered,
                                                      uint32_t sample, uint64_t seed, RaySoAW out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= (uint32_t)(w * h)) return;
    int x = x0 + (int)(k % (uint32_t)w), y = y0 + (int)(k / (uint32_t)w);
    V3f o, d;
    if (jittered) camera_ray_jittered(cam, x, y, sample, seed, o, d);
    else camera_ray_plain(cam, x, y, o, d);
    out.ox[k] = o.x; out.oy[k] = o.y; out.oz[k] = o.z;
    out.dx[k] = d.x; out.dy[k] = d.y; out.dz[k] = d.z;
}

/* Hit branch of cast_ray for every ray of the current queue (cur = depth & 1).
 *
 * Two phases per workgroup pass.  First one thread per ray: the surface frame (normal, lifted hit
 * point, mirror direction), the bounce ray, and the frame parked in LDS under the path's place
 * among the workgroup's hits.  Then one thread per LIGHT SAMPLE, in the order of the shadow queue
 * (entry slot*n_lights + l): the jittered light sample and its ten words.  So the lights of a
 * surface point are worked on by neighbouring lanes instead of a loop, and every store of the
 * shadow queue is 64 consecutive words instead of a stride-n_lights scatter. */
#define SHADE_FRAME_WORDS 12
#define SHADE_GROUP 3   /* light samples per thread between two compactions of the shadow-ray queue */
/* One of the `branches` estimates of Integrator::cast_ray(ray, recursions, branches) (reference
 * integrator.cpp:76-108): the bounce term of the FIRST hit is the mean of `branches` sub-paths, each
 * weighted (1.0/branches)*0.8.  Estimate k is one run of the bounce loop with that weight on the
 * first bounce and its own random streams (stream_off = 256*k on the bounce draws and on the light
 * draws below the first hit); estimates k > 0 leave out what belongs to the first hit itself (its
 * light samples, the sky of a ray that hits nothing), which estimate 0 has already added. */
struct BranchOpts {
    float first_weight;    /* (float)((1.0/branches)*0.8); 0.8f for the renderer */
    uint32_t stream_off;
    int skip_first;
};

__global__ __launch_bounds__(STREAM_BLOCK) void k_shade(DScene sc, CamParams cam, const DLight* lights, uint32_t n_lights,
                                                         uint32_t depth, int last_depth, int trace_all, uint64_t seed, BranchOpts br,
                                                         RenderBuffers b) {
    __shared__ uint32_t s_wave[BLOCK_COMPACT_WORDS];
    __shared__ float s_frame[SHADE_FRAME_WORDS][STREAM_BLOCK];
    __shared__ uint32_t s_group[SHADE_GROUP][STREAM_BLOCK / 64];
    __shared__ uint32_t s_group_base;
    const uint32_t n = b.cnt[CNT_QUEUE + depth];
    const int cur = depth & 1, nxt = cur ^ 1;
    for (uint32_t base_i = blockIdx.x * STREAM_BLOCK; base_i < n; base_i += gridDim.x * STREAM_BLOCK) {
        uint32_t i = base_i + threadIdx.x;
        bool hit = false;
        uint32_t path = 0;
        int32_t tri = -1;
        float4 hrec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (i < n) {
            path = b.qpath[cur][i];
            hrec = b.hit[i];
            tri = (int32_t)__float_as_uint(hrec.x);
            hit = tri >= 0;
            /* One read-modify-write of the path's radiance: first the visible light samples of
             * the previous path vertex (entry i of this queue is slot i of that depth's samples),
             * lights in order, then the sky if this ray left the scene (reference integrator.cpp:153). */
            const bool sky_adds = !hit && (cam.sky.x != 0.0f || cam.sky.y != 0.0f || cam.sky.z != 0.0f) && !(br.skip_first && depth == 0);
            const bool lights_add = depth > 0 && n_lights > 0;
            if (sky_adds || lights_add) {
                float r = b.rad_r[path], g = b.rad_g[path], bl = b.rad_b[path];
                if (lights_add) {
                    for (uint32_t l = 0; l < n_lights; l++) {
                        const size_t j = (size_t)i * n_lights + l;
                        if (b.sh_vis[nxt][j] == 1) { r += b.sh_wr[nxt][j]; g += b.sh_wg[nxt][j]; bl += b.sh_wb[nxt][j]; }
                    }
                }
                if (sky_adds) {
                    const float thr = b.thr[path];
                    r += thr * cam.sky.x; g += thr * cam.sky.y; bl += thr * cam.sky.z;
                }
                b.rad_r[path] = r; b.rad_g[path] = g; b.rad_b[path] = bl;
            }
        }
        /* surviving paths keep one slot in the next queue and n_lights slots in the shadow queue */
        const uint32_t slot = block_compact(hit, &b.cnt[CNT_QUEUE + depth + 1], s_wave);
        const uint32_t slot0 = block_compact_base(s_wave), n_hit = block_compact_total(s_wave);
        if (hit) {
            const float4* sp = sc.shade + (size_t)tri * 4;
            float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
            ShadeTri st;
            st.normal = v3(s0.x, s0.y, s0.z); st.plane = s0.w;
            st.base_n = v3(s1.x, s1.y, s1.z); st.u_n = v3(s2.x, s2.y, s2.z); st.v_n = v3(s3.x, s3.y, s3.z);
            st.d1 = v3(s1.w, s2.w, s3.w);
            V3f o = v3(b.q[cur].ox[i], b.q[cur].oy[i], b.q[cur].oz[i]);
            V3f d = v3(b.q[cur].dx[i], b.q[cur].dy[i], b.q[cur].dz[i]);
            SurfaceFrame f = surface_frame(o, d, hrec.y, hrec.z, hrec.w, st);
            float thr = b.thr[path];
            uint32_t pixel = b.pixel[path], sample = b.sample[path];
            const uint32_t k = slot - slot0;
            s_frame[0][k] = f.n.x; s_frame[1][k] = f.n.y; s_frame[2][k] = f.n.z;
            s_frame[3][k] = f.hit.x; s_frame[4][k] = f.hit.y; s_frame[5][k] = f.hit.z;
            s_frame[6][k] = f.refl.x; s_frame[7][k] = f.refl.y; s_frame[8][k] = f.refl.z;
            s_frame[9][k] = thr; s_frame[10][k] = __uint_as_float(pixel); s_frame[11][k] = __uint_as_float(sample);
            b.qpath[nxt][slot] = path;
            if (!last_depth) {
                V3f bo, bd;
                if (cam.bounce_mode == PT_BOUNCE_LAMBERT) bounce_ray_lambert(f, st, seed, pixel, sample, depth, br.stream_off, bo, bd);
                else bounce_ray(f, bo, bd);
                b.q[nxt].ox[slot] = bo.x; b.q[nxt].oy[slot] = bo.y; b.q[nxt].oz[slot] = bo.z;
                b.q[nxt].dx[slot] = bd.x; b.q[nxt].dy[slot] = bd.y; b.q[nxt].dz[slot] = bd.z;
                b.thr[path] = (depth == 0 ? br.first_weight : 0.8f) * thr; /* reference integrator.cpp:107 */
            }
        }
        __syncthreads();
        /* One thread per light sample of the workgroup's hits, in the order of the samples'
         * entries (slot*n_lights + l).  The weight goes to the sample's entry.  The shadow ray goes
         * to the ray queue only if the weight is not exactly zero: a zero weight (light behind the
         * surface for both the Lambert and the mirror term, or a black light) adds nothing whether
         * the light is visible or not, so that ray's answer cannot change the picture. */
        const uint32_t n_samples = n_hit * n_lights;
        /* SHADE_GROUP samples per thread and compaction: one atomic on the ray count per
         * workgroup and group (one per sample row would make the counter the bottleneck: a
         * single address takes about 90 M atomics per second) */
        for (uint32_t j0 = 0; j0 < n_samples; j0 += SHADE_GROUP * STREAM_BLOCK) { /* same trip count for every thread */
            bool trace[SHADE_GROUP];
            V3f origin[SHADE_GROUP], dir[SHADE_GROUP];
            float dist[SHADE_GROUP];
            uint32_t off[SHADE_GROUP];
#pragma unroll
            for (int g = 0; g < SHADE_GROUP; g++) {
                const uint32_t j = j0 + (uint32_t)g * STREAM_BLOCK + threadIdx.x;
                trace[g] = false;
                origin[g] = v3(0.0f, 0.0f, 0.0f); dir[g] = origin[g]; dist[g] = 0.0f;
                if (j < n_samples) {
                    const uint32_t k = j / n_lights, l = j - k * n_lights;
                    SurfaceFrame f;
                    f.n = v3(s_frame[0][k], s_frame[1][k], s_frame[2][k]);
                    f.hit = v3(s_frame[3][k], s_frame[4][k], s_frame[5][k]);
                    f.refl = v3(s_frame[6][k], s_frame[7][k], s_frame[8][k]);
                    const float thr = s_frame[9][k];
                    const uint32_t pixel = __float_as_uint(s_frame[10][k]), sample = __float_as_uint(s_frame[11][k]);
                    LightSample ls = light_sample(f, lights[l].pos, lights[l].color, seed, pixel, sample, depth, l, depth > 0 ? br.stream_off : 0u);
                    const bool left_out = br.skip_first && depth == 0;
                    const float wr = left_out ? 0.0f : thr * ls.weight.x, wg = left_out ? 0.0f : thr * ls.weight.y, wb = left_out ? 0.0f : thr * ls.weight.z;
                    const size_t q = (size_t)slot0 * n_lights + j;
                    b.sh_wr[cur][q] = wr; b.sh_wg[cur][q] = wg; b.sh_wb[cur][q] = wb;
                    trace[g] = !left_out && (trace_all || !(wr == 0.0f && wg == 0.0f && wb == 0.0f)); /* NaN weights are traced */
                    if (!trace[g]) b.sh_vis[cur][q] = 0;
                    origin[g] = f.hit; dir[g] = ls.dir; dist[g] = ls.dist;
                }
                const uint64_t m = __ballot(trace[g]);
                off[g] = (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
                if ((threadIdx.x & 63u) == 0) s_group[g][threadIdx.x >> 6] = (uint32_t)__popcll(m);
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t total = 0;
                for (int g = 0; g < SHADE_GROUP; g++)
                    for (uint32_t w = 0; w < STREAM_BLOCK / 64; w++) { uint32_t t = s_group[g][w]; s_group[g][w] = total; total += t; }
                s_group_base = total ? atomicAdd(&b.cnt[CNT_SHADOW(depth)], total) : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < SHADE_GROUP; g++) {
                if (trace[g]) {
                    const uint32_t r = s_group_base + s_group[g][threadIdx.x >> 6] + off[g];
                    const uint32_t j = j0 + (uint32_t)g * STREAM_BLOCK + threadIdx.x;
                    b.sh.ox[r] = origin[g].x; b.sh.oy[r] = origin[g].y; b.sh.oz[r] = origin[g].z;
                    b.sh.dx[r] = dir[g].x; b.sh.dy[r] = dir[g].y; b.sh.dz[r] = dir[g].z;
                    b.sh_dist[r] = dist[g];
                    b.sh_target[r] = (uint32_t)((size_t)slot0 * n_lights + j);
                }
            }
            __syncthreads(); /* s_group is rewritten by the next group */
        }
        /* the next pass's block_compact begins with a barrier, which also protects s_frame */
    }
}

/* Adds the visible light samples of depth `depth` to their paths, lights in order.  Used for
 * the last depth only; the others are added by k_shade of the following depth. */
__global__ __launch_bounds__(256) void k_accumulate(uint32_t n_lights, uint32_t depth, RenderBuffers b) {
    const uint32_t n = b.cnt[CNT_QUEUE + depth + 1];
    const int cur = depth & 1, nxt = cur ^ 1;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint32_t path = b.qpath[nxt][i];
        float r = b.rad_r[path], g = b.rad_g[path], bl = b.rad_b[path];
        for (uint32_t l = 0; l < n_lights; l++) {
            uint32_t j = i * n_lights + l;
            if (b.sh_vis[cur][j] == 1) { r += b.sh_wr[cur][j]; g += b.sh_wg[cur][j]; bl += b.sh_wb[cur][j]; }
        }
        b.rad_r[path] = r; b.rad_g[path] = g; b.rad_b[path] = bl;
    }
}

/* Adds finished paths to the film: one thread per pixel of each rectangle group, walking the
 * group's entries (entry_list[entry_off[g] .. entry_off[g+1])) and their samples in order, so sums
 * do not depend on scheduling or on how paths are numbered. */
__global__ __launch_bounds__(256) void k_film(const SubPass* sp, const uint32_t* group_off, const uint32_t* entry_off,
                                               const uint32_t* entry_list, uint32_t n_groups, uint32_t total,
                                               int width, float4* film, RenderBuffers b) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    /* group g owns threads [group_off[g], group_off[g+1]) */
    uint32_t lo = 0, hi = n_groups;
    while (lo + 1 < hi) { uint32_t mid = (lo + hi) >> 1; if (group_off[mid] <= k) lo = mid; else hi = mid; }
    const uint32_t e0 = entry_off[lo], e1 = entry_off[lo + 1];
    const SubPass s0 = sp[entry_list[e0]];
    uint32_t q = k - group_off[lo];
    uint32_t blk = q >> 6, lx = q & 7u, ly = (q >> 3) & 7u;
    int rx = (int)(blk % s0.blocks_x) * 8 + (int)lx, ry = (int)(blk / s0.blocks_x) * 8 + (int)ly;
    if (rx >= s0.w || ry >= s0.h) return;
    size_t pix = (size_t)(s0.y + ry) * (size_t)width + (size_t)(s0.x + rx);
    float4 acc = film[pix];
    for (uint32_t e = e0; e < e1; e++) {
        const SubPass s = sp[entry_list[e]];
        uint32_t p, stride;
        pixel_to_path(s, q, p, stride);
        for (uint32_t si = 0; si < s.n_samples; si++, p += stride) {
            acc.x += b.rad_r[p]; acc.y += b.rad_g[p]; acc.z += b.rad_b[p];
            acc.w += 1.0f;
        }
    }
    film[pix] = acc;
}

__global__ void k_film_split(const float4* film, size_t n, float* rgb, int32_t* passes) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = film[i];
    rgb[3 * i] = v.x; rgb[3 * i + 1] = v.y; rgb[3 * i + 2] = v.z;
    passes[i] = (int32_t)v.w;
}

__global__ void k_film_add(float4* film, size_t n, const float* rgb, const int32_t* passes) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 v = film[i];
    v.x += rgb[3 * i]; v.y += rgb[3 * i + 1]; v.z += rgb[3 * i + 2];
    v.w += (float)passes[i];
    film[i] = v;
}

/* ------------------------------------------------- scratch that is never allocated */

/* Scratch of pt_raycast_closest_dev_noalloc: what the other entry points keep in hipMalloc'ed
 * buffers of the scene (ray count and queue heads; the stack spill area of the trace grid), in a
 * file-scope device array instead, so that the call allocates nothing.  Every GPU the library
 * runs on holds its own copy of the array (at its own address) until the process ends:
 * NOALLOC_WORDS * 4 bytes = 44 MiB + 1 KiB.  The array's name is only a host-side handle; the
 * address kernels and memsets use is looked up once per device with hipGetSymbolAddress
 * (pt_scene_create) and kept in the scene. */
#define NOALLOC_GRID 1024u                                    /* workgroups of the no-alloc trace launch (4 per CU) */
#define NOALLOC_HEAD_WORDS 256u                               /* [0] ray count, queue heads from SMALL_HEADS0 on */
#define NOALLOC_SPILL_WORDS (NOALLOC_GRID * TRACE_BLOCK * (STACK_MAX_CLOSEST - STACK_LDS))
#define NOALLOC_WORDS (NOALLOC_HEAD_WORDS + NOALLOC_SPILL_WORDS)
static __device__ __align__(256) uint32_t scratch[NOALLOC_WORDS];

/* Ray count into word 0, zero into the queue heads: the spill area needs no clearing (a lane
 * only reads stack entries it has written during the same ray). */
__global__ __launch_bounds__(NOALLOC_HEAD_WORDS) void k_noalloc_prepare(uint32_t* head, uint32_t n) {
    head[threadIdx.x] = threadIdx.x == 0 ? n : 0u;
}

/* ------------------------------------------------- leaves stored by component */

/* One thread per leaf: its triangles' nine floats from the 48-byte records into the nine rows of its
 * pieces (slots past the leaf's end repeat its last triangle). */
__global__ __launch_bounds__(256) void k_leafwise_fill(const uint32_t* __restrict__ tris, const uint32_t* __restrict__ start,
                                                        const uint32_t* __restrict__ piece, const uint32_t* __restrict__ count,
                                                        uint32_t n_leaves, uint32_t* __restrict__ soa) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    if (l >= n_leaves) return;
    const uint32_t s = start[l], cnt = count[l];
    const uint32_t slots = ((cnt + LEAFWISE_PIECE - 1u) / LEAFWISE_PIECE) * LEAFWISE_PIECE;
    uint32_t* dst = soa + (size_t)piece[l] * LEAFWISE_PIECE_WORDS;
    for (uint32_t j = 0; j < slots; j++) {
        const uint32_t* src = tris + (size_t)(s + (j < cnt ? j : cnt - 1u)) * (sizeof(pt_tri_isect) / 4);
        uint32_t* at = dst + (size_t)(j / LEAFWISE_PIECE) * LEAFWISE_PIECE_WORDS + (j % LEAFWISE_PIECE);
#pragma unroll
        for (uint32_t c = 0; c < 9u; c++) at[c * LEAFWISE_PIECE] = src[c];
    }
}

/* Closest hits of the leafwise walk name their triangle by piece and slot (piece << 6 | slot); the
 * rest of the library wants its position in the triangle arrays (FLAT: the word in out_hit) or its
 * index in the caller's mesh (MESH: out_idx). */
__global__ __launch_bounds__(256) void k_leafwise_positions(const uint32_t* __restrict__ d_count, uint32_t count_mul, float4* hit, int32_t* idx,
                                                             const uint32_t* __restrict__ piece_start, const uint32_t* __restrict__ tris) {
    const uint32_t n = *d_count * count_mul;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        if (hit) {
            const uint32_t slot = __float_as_uint(hit[i].x);
            if (slot != 0xffffffffu) hit[i].x = __uint_as_float(piece_start[slot >> 6] + (slot & 63u));
        } else {
            const int32_t slot = idx[i];
            if (slot >= 0) {
                const uint32_t pos = piece_start[(uint32_t)slot >> 6] + ((uint32_t)slot & 63u);
                idx[i] = (int32_t)tris[(size_t)pos * (sizeof(pt_tri_isect) / 4) + offsetof(pt_tri_isect, mesh_index) / 4];
            }
        }
    }
}

/* Host part: the node records with their leaf references rewritten, and where each leaf goes. */
struct LeafwiseBuild {
    std::vector<pt_flat_node> nodes;
    std::vector<uint32_t> start, piece, count;
    std::vector<uint32_t> piece_start;   /* per piece: position of its first triangle in the triangle arrays */
    uint32_t root = PT_REF_NONE;
    uint64_t pieces = 0;
    bool ok = true;
};

static void leafwise_build(const pt_flat_scene* f, LeafwiseBuild& out) {
    out.nodes.assign(f->nodes, f->nodes + f->n_nodes);
    auto convert = [&](uint32_t ref) -> uint32_t {
        if (!(ref & PT_REF_LEAF)) return ref;
        const uint32_t start = ref & PT_REF_START_MASK;
        uint32_t count = (ref >> PT_REF_COUNT_SHIFT) & PT_REF_COUNT_MASK;
        if (count == 0) count = f->tris[start].leaf_count;
        if (count > LEAFWISE_MAX_COUNT) { out.ok = false; return ref; }
        const uint64_t g = out.pieces;
        out.pieces += (count + LEAFWISE_PIECE - 1) / LEAFWISE_PIECE;
        if (out.pieces > LEAFWISE_MAX_PIECES) { out.ok = false; return ref; }
        out.start.push_back(start); out.piece.push_back((uint32_t)g); out.count.push_back(count);
        for (uint32_t k = 0; k < count; k += LEAFWISE_PIECE) out.piece_start.push_back(start + k);
        return PT_REF_LEAF | ((uint32_t)g << 8) | count;
    };
    out.root = convert(f->root_ref);
    for (pt_flat_node& n : out.nodes) { n.lo_ref = convert(n.lo_ref); n.hi_ref = convert(n.hi_ref); }
}

/* --------------------------------------------------------------- host side */

struct TimedLaunch { int cls; hipEvent_t a, b; };

/* layout of pt_scene::d_small: a launch's queue heads are 8 counters, 8 words apart */
#define SMALL_HEADS0 64
#define SMALL_HEADS1 128
#define SMALL_WORDS 192

/* Where a kernel goes: the stream, and the stack spill area that the trace launches of that
 * stream share (launches on one stream run one after another). */
struct LaunchCtx { hipStream_t stream; uint32_t* spill; };

struct pt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    DScene d{};
    void* d_geom = nullptr;  /* nodes and triangle records */
    void* d_shade = nullptr;
    void* d_leafwise = nullptr;   /* k_shadow_leafwise: rewritten node records, then the triangles by component */
    bool use_leafwise = false;
    uint32_t* d_piece_start = nullptr;   /* inside d_leafwise */
    int grid_leafwise = 0, grid_leafwise_closest = 0;
    uint64_t n_tris = 0;
    uint32_t* d_small = nullptr; /* raycast entry points: [0] ray count, [1] deferred count, two sets of queue heads at SMALL_HEADS0/1 */
    uint32_t* d_spill = nullptr; /* stack spill area shared by all trace launches on the stream */
    uint32_t* d_static_scratch = nullptr; /* this device's copy of `scratch` (never allocated, never freed) */
    /* Small host-pointer casts (pt_raycast_closest with n <= PICK_MAX_RAYS: one ray under the mouse,
     * kdTree::ray_test): their own stream, counters, spill area and a pinned host block that the
     * kernel reads the rays from and writes the hits to, made on first use and kept, behind a
     * mutex -- so such a call needs no allocation, no copy, and may run while another thread
     * renders with this scene. */
    std::mutex pick_mu;
    hipStream_t pick_stream = nullptr;
    uint32_t* d_pick = nullptr;   /* PICK_HEAD_WORDS of counters, then the spill area of PICK_GRID workgroups */
    float* h_pick = nullptr;      /* pinned, device-visible: 10 arrays of PICK_MAX_RAYS words */
    float* h_pick_dev = nullptr;  /* the same block as the GPU addresses it */
    /* second render lane (see pt_render_passes): its own stream and spill area, made on first use */
    hipStream_t stream2 = nullptr;
    uint32_t* d_spill2 = nullptr;
    size_t spill_bytes = 0;
    hipEvent_t ev_order = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;   /* PT_OVERLAP_SHADOW: a depth's shadow traces on the second stream */
    uint32_t* d_defer = nullptr; /* deferred shadow rays of pt_raycast_shadow_dev */
    uint64_t defer_cap = 0;
    int grid_closest = 0, grid_any = 0, grid_dense = 0; /* resident blocks of the persistent trace kernels */
    int tris_per_round = 1;             /* 2, 3: the trace kernels that test that many triangles of a leaf per round (large leaves) */
    bool use_qnodes = false;            /* the any-hit launches use the kernels that read quantised interior nodes */
    bool use_qnodes_leafwise = false;   /* ... in their leafwise form (use_leafwise) */
    bool use_dense = false;             /* the reference's light test is answered by k_shadow_dense (deferred, dense triangle tests) */
    std::vector<TimedLaunch> pending;
    std::vector<hipEvent_t> free_events;
    double ms[PT_KERNEL_CLASSES] = {0, 0, 0, 0, 0, 0};
    uint64_t launches[PT_KERNEL_CLASSES] = {0, 0, 0, 0, 0, 0};
    bool timing = true;
};

static int init_trace_launch(pt_scene* s);

static int set_device(const pt_scene* s) {
    HIP_TRY(hipSetDevice(s->device));
    return PT_OK;
}

static hipEvent_t get_event(pt_scene* s) {
    if (!s->free_events.empty()) { hipEvent_t e = s->free_events.back(); s->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static int resolve_timing(pt_scene* s) {
    if (s->pending.empty()) return PT_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
    for (auto& t : s->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { s->ms[t.cls] += ms; s->launches[t.cls]++; }
        s->free_events.push_back(t.a);
        s->free_events.push_back(t.b);
    }
    s->pending.clear();
    return PT_OK;
}

struct ScopedTimer {
    pt_scene* s; int cls; hipStream_t stream; hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(pt_scene* s_, int cls_, hipStream_t stream_ = nullptr) : s(s_), cls(cls_), stream(stream_ ? stream_ : s_->stream) {
        if (!s->timing) return;
        a = get_event(s); b = get_event(s);
        if (a && b) (void)hipEventRecord(a, stream);
    }
    ~ScopedTimer() {
        if (!s->timing || !a || !b) return;
        (void)hipEventRecord(b, stream);
        s->pending.push_back({cls, a, b});
    }
};

/* Walks the flattened tree once on the host so that no kernel ever follows a bad reference. */
static int validate_flat(const pt_flat_scene* f, uint64_t* n_leaves, uint32_t* largest_leaf) {
    *n_leaves = 0;
    *largest_leaf = 0;
    if (f->n_tris >= PT_MAX_TRIANGLES) return fail(PT_ERR_LIMIT, "too many triangles (limit 2^26 - 64)");
    if (f->n_tris == 0) return PT_OK;
    if (!f->tris || !f->shade) return fail(PT_ERR_INVALID, "null triangle arrays");
    if (f->n_nodes > 0 && !f->nodes) return fail(PT_ERR_INVALID, "null node array");
    if (f->n_nodes >= (1u << 26)) return fail(PT_ERR_LIMIT, "too many nodes");
    struct Item { uint32_t ref; uint32_t depth; };
    std::vector<Item> stack;
    std::vector<uint8_t> seen((size_t)f->n_nodes, 0);
    stack.push_back({f->root_ref, 0});
    uint64_t tri_total = 0;
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        if (it.ref == 0xffffffffu || (it.ref & 0x40000000u)) return fail(PT_ERR_INVALID, "malformed child reference");
        if (it.ref & PT_REF_LEAF) {
            uint32_t start = it.ref & PT_REF_START_MASK;
            uint32_t count = (it.ref >> PT_REF_COUNT_SHIFT) & PT_REF_COUNT_MASK;
            if (start >= f->n_tris) return fail(PT_ERR_INVALID, "leaf start outside the triangle array");
            if (count == 0) count = f->tris[start].leaf_count;
            if (count == 0 || (uint64_t)start + count > f->n_tris) return fail(PT_ERR_INVALID, "leaf range outside the triangle array");
            /* the kernels walk a leaf by its records' leaf_remaining fields */
            for (uint32_t k = 0; k < count; k++)
                if (f->tris[start + k].leaf_remaining != count - 1 - k) return fail(PT_ERR_INVALID, "leaf_remaining does not count down to the leaf's end");
            tri_total += count;
            *n_leaves += 1;
            if (count > *largest_leaf) *largest_leaf = count;
            continue;
        }
        if (it.ref >= f->n_nodes) return fail(PT_ERR_INVALID, "node reference out of range");
        if (seen[it.ref]) return fail(PT_ERR_INVALID, "node referenced twice (not a tree)");
        seen[it.ref] = 1;
        if (it.depth + 1 > PT_MAX_TREE_DEPTH) return fail(PT_ERR_LIMIT, "tree deeper than PT_MAX_TREE_DEPTH");
        const pt_flat_node& n = f->nodes[it.ref];
        if (n.axis > 2) return fail(PT_ERR_INVALID, "split axis out of range");
        stack.push_back({n.lo_ref, it.depth + 1});
        stack.push_back({n.hi_ref, it.depth + 1});
    }
    if (tri_total > f->n_tris) return fail(PT_ERR_INVALID, "leaves cover more triangles than the array holds");
    return PT_OK;
}

extern "C" {

const char* pt_last_error(void) { return g_err.c_str(); }

const char* pt_build_info(void) { return "pt_hip 0.1 (gfx950, strict fp32, persistent-wave trace kernels)"; }

int pt_device_count(int* n) {
    if (!n) return fail(PT_ERR_INVALID, "null argument");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(PT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return PT_OK;
}

int pt_device_prepare(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(PT_ERR_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));   /* forces the context into existence */
    return PT_OK;
}

int pt_scene_create(int device, const pt_flat_scene* flat, pt_scene** out) {
    if (!flat || !out) return fail(PT_ERR_INVALID, "null argument");
    *out = nullptr;
    uint64_t n_leaves = 0;
    uint32_t largest_leaf = 0;
    int rc = validate_flat(flat, &n_leaves, &largest_leaf);
    if (rc != PT_OK) return rc;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(PT_ERR_INVALID, "device index out of range");
    pt_scene* s = new pt_scene();
    s->device = device;
    auto cleanup = [&](int code) { pt_scene_destroy(s); return code; };
    if (hipSetDevice(device) != hipSuccess) return cleanup(fail(PT_ERR_DEVICE, "hipSetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) return cleanup(fail(PT_ERR_DEVICE, "hipStreamCreate failed"));
    s->n_tris = flat->n_tris;
    size_t nb = std::max<size_t>(flat->n_nodes, 1) * sizeof(pt_flat_node);
    size_t tb = std::max<size_t>(flat->n_tris, 1) * sizeof(pt_tri_isect);
    size_t sb = std::max<size_t>(flat->n_tris, 1) * sizeof(pt_tri_shade);
    const size_t tris_off = (nb + 255) & ~(size_t)255;
    if (tris_off + tb >= ((size_t)1 << 32)) return cleanup(fail(PT_ERR_LIMIT, "nodes and triangle records exceed 4 GiB"));
    /* Quantised interior nodes for the any-hit walks, behind the triangle records if they still fit
     * below 4 GiB.  They halve the bytes of most node fetches and cost about 90 vector instructions
     * per round to decode.  Where the geometry is served by the caches the trace kernels are bound by
     * vector instructions, not by their loads, and lose (measured on MI355X, 1.34 M triangles: shadow
     * launches 26 % slower); where it streams from HBM they win (21.5 M triangles: shadow launches
     * 4 % faster).  So: only for geometry larger than the 256 MiB Infinity Cache.  PT_QNODES=0 / 1
     * overrides. */
    bool want_leafwise = n_leaves > 0 && flat->n_tris >= (uint64_t)LEAFWISE_MIN_AVERAGE_LEAF * n_leaves && largest_leaf <= LEAFWISE_MAX_COUNT;
    if (const char* e = getenv("PT_LEAFWISE")) want_leafwise = atoi(e) >= 1;
    ptq::QNodes qn;
    {
        bool want = tris_off + tb > ((size_t)256 << 20) && (!want_leafwise || LEAFWISE_QNODES_BY_DEFAULT);
        if (const char* e = getenv("PT_QNODES")) want = atoi(e) == 1;
        if (want) ptq::build_qnodes(flat, qn);
    }
    const size_t q_off = (tris_off + tb + 255) & ~(size_t)255;
    const size_t qb = qn.rec.size() * sizeof(ptq::QNodeRec);
    if (q_off + qb >= ((size_t)1 << 32)) qn.rec.clear();
    /* the leafwise any-hit walks read their q-nodes from a copy behind their own node records */
    const bool with_q = !qn.rec.empty() && !want_leafwise;
    if (hipMalloc(&s->d_geom, with_q ? q_off + qb : tris_off + tb) != hipSuccess ||
        hipMalloc(&s->d_shade, sb) != hipSuccess || hipMalloc((void**)&s->d_small, SMALL_WORDS * sizeof(uint32_t)) != hipSuccess)
        return cleanup(fail(PT_ERR_DEVICE, "hipMalloc failed for the scene arrays"));
    bool ok = true;
    if (flat->n_nodes) ok = ok && hipMemcpy(s->d_geom, flat->nodes, flat->n_nodes * sizeof(pt_flat_node), hipMemcpyHostToDevice) == hipSuccess;
    if (flat->n_tris) {
        ok = ok && hipMemcpy((char*)s->d_geom + tris_off, flat->tris, flat->n_tris * sizeof(pt_tri_isect), hipMemcpyHostToDevice) == hipSuccess;
        ok = ok && hipMemcpy(s->d_shade, flat->shade, flat->n_tris * sizeof(pt_tri_shade), hipMemcpyHostToDevice) == hipSuccess;
    }
    if (with_q) ok = ok && hipMemcpy((char*)s->d_geom + q_off, qn.rec.data(), qb, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) return cleanup(fail(PT_ERR_DEVICE, "hipMemcpy failed for the scene arrays"));
    s->use_qnodes = with_q;
    /* Deferred, dense triangle tests for the light-visibility walk (k_shadow_dense): every leaf's
     * count must fit its child reference (at most 15 triangles), and the kernel exists in the
     * one-triangle-per-round, exact-node form only.  Off unless PT_DENSE_SHADOW=1: it executes 21 %
     * fewer vector instructions at 0.64 instead of 0.41 lane utilisation, but with less work per
     * round and 7 waves per SIMD it no longer hides its loads (measured on MI355X: shadow launches
     * 16 % slower than k_trace<SHADOW_ANY>). */
    s->use_dense = false;
    if (const char* e = getenv("PT_DENSE_SHADOW")) s->use_dense = atoi(e) == 1 && !with_q && largest_leaf > 0 && largest_leaf <= PT_REF_COUNT_MASK;
    /* Wave-wide leaf tests (k_trace<MODE, 0>): for scenes whose leaves are large; every leaf at most
     * 255 triangles.  PT_LEAFWISE = 0 / 1 overrides. */
    s->use_leafwise = false;
    {
        if (want_leafwise && flat->n_tris > 0) {
            LeafwiseBuild lw;
            leafwise_build(flat, lw);
            if (lw.ok) {
                const size_t nodes_b = (std::max<size_t>(flat->n_nodes, 1) * sizeof(pt_flat_node) + 255) & ~(size_t)255;
                const size_t soa_b = std::max<uint64_t>(lw.pieces, 1) * LEAFWISE_PIECE_WORDS * sizeof(float);
                const size_t nl = lw.start.size();
                const size_t ps_b = (std::max<size_t>(lw.piece_start.size(), 1) * sizeof(uint32_t) + 255) & ~(size_t)255;
                const size_t lq_off = nodes_b + soa_b + ps_b;
                const bool lw_q = !qn.rec.empty() && lq_off + qb < ((size_t)1 << 32);
                uint32_t* d_tab = nullptr;
                bool good = hipMalloc(&s->d_leafwise, lq_off + (lw_q ? qb : 0)) == hipSuccess && hipMalloc((void**)&d_tab, std::max<size_t>(nl, 1) * 12) == hipSuccess;
                good = good && hipMemset((char*)s->d_leafwise + nodes_b, 0, soa_b) == hipSuccess;
                if (good && lw_q) good = hipMemcpy((char*)s->d_leafwise + lq_off, qn.rec.data(), qb, hipMemcpyHostToDevice) == hipSuccess;
                if (good && !lw.piece_start.empty())
                    good = hipMemcpy((char*)s->d_leafwise + nodes_b + soa_b, lw.piece_start.data(), lw.piece_start.size() * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess;
                if (good && flat->n_nodes) good = hipMemcpy(s->d_leafwise, lw.nodes.data(), flat->n_nodes * sizeof(pt_flat_node), hipMemcpyHostToDevice) == hipSuccess;
                if (good && nl) {
                    good = hipMemcpy(d_tab, lw.start.data(), nl * 4, hipMemcpyHostToDevice) == hipSuccess &&
                           hipMemcpy(d_tab + nl, lw.piece.data(), nl * 4, hipMemcpyHostToDevice) == hipSuccess &&
                           hipMemcpy(d_tab + 2 * nl, lw.count.data(), nl * 4, hipMemcpyHostToDevice) == hipSuccess;
                    if (good) {
                        hipLaunchKernelGGL(k_leafwise_fill, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, s->stream,
                                           (const uint32_t*)((const char*)s->d_geom + tris_off), d_tab, d_tab + nl, d_tab + 2 * nl, (uint32_t)nl,
                                           (uint32_t*)((char*)s->d_leafwise + nodes_b));
                        good = hipGetLastError() == hipSuccess && hipStreamSynchronize(s->stream) == hipSuccess;
                    }
                }
                if (d_tab) (void)hipFree(d_tab);
                if (!good) return cleanup(fail(PT_ERR_DEVICE, "setting up the component-wise leaves failed"));
                s->use_leafwise = true;
                s->use_qnodes_leafwise = lw_q;
                s->d.leafwise_q_off = (uint32_t)lq_off;
                s->d_piece_start = (uint32_t*)((char*)s->d_leafwise + nodes_b + soa_b);
                s->d.leafwise_nodes = (const char*)s->d_leafwise;
                s->d.soa = (const float*)((const char*)s->d_leafwise + nodes_b);
                s->d.leafwise_root = lw.root;
            }
        }
    }
    s->d.q_off = (uint32_t)q_off;
    s->d.q_root = (with_q || s->use_qnodes_leafwise) ? qn.root : PT_REF_NONE;
    s->d.q_root_exact = PT_Q_EXACT_BASE + flat->root_ref;
    s->d.grid = qn.grid;
    s->d.geom = (const char*)s->d_geom;
    s->d.tris_off = (uint32_t)tris_off;
    s->d.shade = (const float4*)s->d_shade;
    s->d.root_ref = flat->root_ref;
    s->d.n_tris = (uint32_t)flat->n_tris;
    /* Leaves of sixteen and more triangles on average (the reference's depth cap, kdtree.cpp:67,
     * at tens of millions of triangles): the kernels that test two or three triangles per round
     * pay there and cost on ordinary trees (measured: 41 per leaf +15 % with three, 10 per leaf
     * -4 % with two, 5.6 per leaf -14 %).  PT_TRIS_PER_ROUND = 1 / 2 / 3 overrides. */
    s->tris_per_round = n_leaves == 0 ? 1 : flat->n_tris >= 24 * n_leaves ? 3 : flat->n_tris >= 16 * n_leaves ? 2 : 1;
    if (const char* e = getenv("PT_TRIS_PER_ROUND")) { int v = atoi(e); if (v >= 1 && v <= 3) s->tris_per_round = v; }
    if (s->tris_per_round != 1) s->use_dense = false;
    if (init_trace_launch(s) != PT_OK) return cleanup(PT_ERR_DEVICE);
    {   /* the address of this device's copy of the static scratch array (the device is current) */
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return cleanup(fail(PT_ERR_DEVICE, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e)));
        s->d_static_scratch = (uint32_t*)p;
    }
    s->d.root.mnx = flat->root_min[0]; s->d.root.mny = flat->root_min[1]; s->d.root.mnz = flat->root_min[2];
    s->d.root.mxx = flat->root_max[0]; s->d.root.mxy = flat->root_max[1]; s->d.root.mxz = flat->root_max[2];
    *out = s;
    return PT_OK;
}

void pt_scene_destroy(pt_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream2) (void)hipStreamSynchronize(s->stream2);
    for (auto& t : s->pending) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : s->free_events) (void)hipEventDestroy(e);
    if (s->d_geom) (void)hipFree(s->d_geom);
    if (s->d_shade) (void)hipFree(s->d_shade);
    if (s->d_leafwise) (void)hipFree(s->d_leafwise);
    if (s->d_small) (void)hipFree(s->d_small);
    if (s->d_spill) (void)hipFree(s->d_spill);
    if (s->d_spill2) (void)hipFree(s->d_spill2);
    if (s->d_defer) (void)hipFree(s->d_defer);
    if (s->pick_stream) { (void)hipStreamSynchronize(s->pick_stream); (void)hipStreamDestroy(s->pick_stream); }
    if (s->d_pick) (void)hipFree(s->d_pick);
    if (s->h_pick) (void)hipHostFree(s->h_pick);
    if (s->ev_order) (void)hipEventDestroy(s->ev_order);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
    if (s->stream2) (void)hipStreamDestroy(s->stream2);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int pt_scene_sync(pt_scene* s) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    int rc = set_device(s);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream2) HIP_TRY(hipStreamSynchronize(s->stream2));
    return PT_OK;
}

int pt_scene_kernel_time(pt_scene* s, int cls, int reset, double* ms, uint64_t* launches) {
    if (!s || cls < 0 || cls >= PT_KERNEL_CLASSES) return fail(PT_ERR_INVALID, "bad argument");
    int rc = set_device(s);
    if (rc) return rc;
    rc = resolve_timing(s);
    if (rc) return rc;
    if (ms) *ms = s->ms[cls];
    if (launches) *launches = s->launches[cls];
    if (reset) { s->ms[cls] = 0; s->launches[cls] = 0; }
    return PT_OK;
}

} // extern "C"

/* Grid of a persistent trace launch: as many blocks as stay resident, so every wave starts at
 * once and pulls work until the queue is empty. */
static int trace_grid(const pt_scene* s, bool anyhit) { return anyhit ? s->grid_any : s->grid_closest; }

template <int MODE>
static int launch_trace(pt_scene* s, TraceArgs a, const LaunchCtx* ctx = nullptr) {
    if (!ctx && s->pending.size() > 8192) { int rc = resolve_timing(s); if (rc) return rc; }
    const LaunchCtx here = ctx ? *ctx : LaunchCtx{s->stream, s->d_spill};
    a.spill = here.spill;
    if (a.count_mul == 0) a.count_mul = 1;
    ScopedTimer tm(s, (MODE == TRACE_CLOSEST_MESH || MODE == TRACE_CLOSEST_FLAT) ? PT_KERNEL_CLOSEST : MODE == TRACE_SHADOW_EXACT ? PT_KERNEL_SHADOW_EXACT : PT_KERNEL_SHADOW, here.stream);
    constexpr bool ANY = MODE == TRACE_SHADOW_ANY || MODE == TRACE_SHADOW_BOUNDED;
    const dim3 grid(trace_grid(s, ANY));
    if (s->use_leafwise) {
        constexpr int QMODE = ANY ? MODE : TRACE_SHADOW_ANY;   /* (keeps the closest-hit instantiations of this function valid) */
        if (ANY && s->use_qnodes_leafwise) hipLaunchKernelGGL((k_trace<QMODE, 0, true>), dim3(s->grid_leafwise), dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
        else hipLaunchKernelGGL((k_trace<MODE, 0>), dim3(ANY ? s->grid_leafwise : s->grid_leafwise_closest), dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
        if (MODE == TRACE_CLOSEST_FLAT || MODE == TRACE_CLOSEST_MESH) {
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_leafwise_positions, dim3((unsigned)std::max(8, s->num_cus * 8)), dim3(256), 0, here.stream, a.d_count, a.count_mul,
                               MODE == TRACE_CLOSEST_FLAT ? a.out_hit : (float4*)nullptr, MODE == TRACE_CLOSEST_MESH ? a.out_idx : (int32_t*)nullptr,
                               (const uint32_t*)s->d_piece_start, (const uint32_t*)(s->d.geom + s->d.tris_off));
        }
    }
    else if (MODE == TRACE_SHADOW_ANY && s->use_dense) {
        hipLaunchKernelGGL(k_shadow_dense<DENSE_RAYS_PER_LANE>, dim3(s->grid_dense), dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
    }
    else if (ANY && s->use_qnodes) {
        constexpr int QMODE = ANY ? MODE : TRACE_SHADOW_ANY;   /* (keeps the closest-hit instantiations of this function valid) */
        if (s->tris_per_round == 2) hipLaunchKernelGGL((k_trace<QMODE, 2, true>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
        else if (s->tris_per_round == 3) hipLaunchKernelGGL((k_trace<QMODE, 3, true>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
        else hipLaunchKernelGGL((k_trace<QMODE, 1, true>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
    }
    else if (s->tris_per_round == 2) hipLaunchKernelGGL((k_trace<MODE, 2>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
    else if (s->tris_per_round == 3) hipLaunchKernelGGL((k_trace<MODE, 3>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
    else hipLaunchKernelGGL((k_trace<MODE, 1>), grid, dim3(TRACE_BLOCK), 0, here.stream, s->d, a);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

/* Sizes the persistent grids from the occupancy the compiled kernels actually get and
 * allocates the stack spill area for the larger of them. */
static int init_trace_launch(pt_scene* s) {
    int per_cu_c = 0, per_cu_a = 0;
    if (s->tris_per_round == 2) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_c, k_trace<TRACE_CLOSEST_FLAT, 2>, TRACE_BLOCK, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 2>, TRACE_BLOCK, 0));
    } else if (s->tris_per_round == 3) {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_c, k_trace<TRACE_CLOSEST_FLAT, 3>, TRACE_BLOCK, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 3>, TRACE_BLOCK, 0));
    } else {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_c, k_trace<TRACE_CLOSEST_FLAT, 1>, TRACE_BLOCK, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 1>, TRACE_BLOCK, 0));
    }
    if (s->use_qnodes) {   /* the any-hit kernels that will actually run */
        if (s->tris_per_round == 2) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 2, true>, TRACE_BLOCK, 0));
        else if (s->tris_per_round == 3) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 3, true>, TRACE_BLOCK, 0));
        else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_a, k_trace<TRACE_SHADOW_ANY, 1, true>, TRACE_BLOCK, 0));
    }
    if (const char* e = getenv("PT_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) { per_cu_c = v; per_cu_a = v; } }
    int per_cu_d = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_d, k_shadow_dense<DENSE_RAYS_PER_LANE>, TRACE_BLOCK, 0));
    if (const char* e = getenv("PT_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) per_cu_d = v; }
    per_cu_d = std::max(1, std::min(per_cu_d, TRACE_MAX_BLOCKS_PER_CU));
    s->grid_dense = std::max(8, (s->num_cus * per_cu_d) & ~7);
    int per_cu_l = 0;
    if (s->use_qnodes_leafwise) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_l, k_trace<TRACE_SHADOW_ANY, 0, true>, TRACE_BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_l, k_trace<TRACE_SHADOW_ANY, 0>, TRACE_BLOCK, 0));
    if (const char* e = getenv("PT_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) per_cu_l = v; }
    per_cu_l = std::max(1, std::min(per_cu_l, TRACE_MAX_BLOCKS_PER_CU));
    s->grid_leafwise = std::max(8, (s->num_cus * per_cu_l) & ~7);
    int per_cu_lc = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_lc, k_trace<TRACE_CLOSEST_FLAT, 0>, TRACE_BLOCK, 0));
    if (const char* e = getenv("PT_TRACE_BLOCKS_PER_CU")) { int v = atoi(e); if (v > 0) per_cu_lc = v; }
    per_cu_lc = std::max(1, std::min(per_cu_lc, TRACE_MAX_BLOCKS_PER_CU));
    s->grid_leafwise_closest = std::max(8, (s->num_cus * per_cu_lc) & ~7);
    per_cu_c = std::max(1, std::min(per_cu_c, TRACE_MAX_BLOCKS_PER_CU));
    per_cu_a = std::max(1, std::min(per_cu_a, TRACE_MAX_BLOCKS_PER_CU));
    /* a multiple of 8: the kernels deal their queue to the 8 XCDs by workgroup number */
    s->grid_closest = std::max(8, (s->num_cus * per_cu_c) & ~7);
    s->grid_any = std::max(8, (s->num_cus * per_cu_a) & ~7);
    size_t threads = (size_t)std::max(std::max(std::max(s->grid_closest, s->grid_any), s->grid_dense), std::max(s->grid_leafwise, s->grid_leafwise_closest)) * TRACE_BLOCK;
    size_t words = threads * (size_t)(STACK_MAX_CLOSEST - STACK_LDS);
    s->spill_bytes = words * sizeof(uint32_t);
    HIP_TRY(hipMalloc((void**)&s->d_spill, s->spill_bytes));
    return PT_OK;
}

/* The second render lane's stream, spill area and ordering event, made on first use. */
static int ensure_second_lane(pt_scene* s) {
    if (!s->stream2) HIP_TRY(hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking));
    if (!s->d_spill2) HIP_TRY(hipMalloc((void**)&s->d_spill2, s->spill_bytes));
    if (!s->ev_order) HIP_TRY(hipEventCreateWithFlags(&s->ev_order, hipEventDisableTiming));
    return PT_OK;
}

static int fill_miss_closest(pt_scene* s, uint64_t n, int32_t* d_idx, float* d_t, float* d_u, float* d_v) {
    HIP_TRY(hipMemsetAsync(d_idx, 0xff, n * 4, s->stream));
    HIP_TRY(hipMemsetAsync(d_t, 0, n * 4, s->stream));
    HIP_TRY(hipMemsetAsync(d_u, 0, n * 4, s->stream));
    HIP_TRY(hipMemsetAsync(d_v, 0, n * 4, s->stream));
    return PT_OK;
}

extern "C" {

int pt_scene_trace_variant(const pt_scene* s, int32_t out[4]) {
    if (!s || !out) return fail(PT_ERR_INVALID, "null argument");
    out[0] = s->use_leafwise ? 0 : s->tris_per_round;
    out[1] = (s->use_leafwise ? s->use_qnodes_leafwise : s->use_qnodes) ? 1 : 0;
    out[2] = s->use_dense && !s->use_leafwise ? 1 : 0;
    out[3] = 0;
    return PT_OK;
}

int pt_raycast_closest_dev(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                           const float* dy, const float* dz, int32_t* idx, float* t, float* u, float* v) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !idx || !t || !u || !v) return fail(PT_ERR_INVALID, "null array");
    if (n > PT_MAX_RAYS_PER_CALL) return fail(PT_ERR_LIMIT, "more than PT_MAX_RAYS_PER_CALL rays in one call");
    int rc = set_device(s);
    if (rc) return rc;
    if (s->n_tris == 0) return fill_miss_closest(s, n, idx, t, u, v);
    uint32_t init[SMALL_WORDS] = {(uint32_t)n};
    HIP_TRY(hipMemcpyAsync(s->d_small, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
    TraceArgs a{};
    a.rays = {ox, oy, oz, dx, dy, dz};
    a.d_count = s->d_small; a.d_next_chunk = s->d_small + SMALL_HEADS0;
    a.out_idx = idx; a.out_t = t; a.out_u = u; a.out_v = v;
    return launch_trace<TRACE_CLOSEST_MESH>(s, a);
}

int pt_raycast_shadow_dev(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                          const float* dy, const float* dz, const float* dist, uint8_t* vis) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !dist || !vis) return fail(PT_ERR_INVALID, "null array");
    if (n > PT_MAX_RAYS_PER_CALL) return fail(PT_ERR_LIMIT, "more than PT_MAX_RAYS_PER_CALL rays in one call");
    int rc = set_device(s);
    if (rc) return rc;
    if (s->n_tris == 0) { HIP_TRY(hipMemsetAsync(vis, 1, n, s->stream)); return PT_OK; }
    if (n > s->defer_cap) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        if (s->d_defer) (void)hipFree(s->d_defer);
        s->d_defer = nullptr; s->defer_cap = 0;
        HIP_TRY(hipMalloc((void**)&s->d_defer, n * sizeof(uint32_t)));
        s->defer_cap = n;
    }
    uint32_t init[SMALL_WORDS] = {(uint32_t)n};
    HIP_TRY(hipMemcpyAsync(s->d_small, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
    TraceArgs a{};
    a.rays = {ox, oy, oz, dx, dy, dz};
    a.d_count = s->d_small; a.d_next_chunk = s->d_small + SMALL_HEADS0;
    a.dist = dist; a.vis = vis;
    a.defer_list = s->d_defer; a.defer_count = s->d_small + 1;
    rc = launch_trace<TRACE_SHADOW_ANY>(s, a);
    if (rc) return rc;
    a.d_count = s->d_small + 1; a.d_next_chunk = s->d_small + SMALL_HEADS1;
    return launch_trace<TRACE_SHADOW_EXACT>(s, a);
}

int pt_raycast_occluded_dev(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                            const float* dy, const float* dz, const float* max_t, uint8_t* vis) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !max_t || !vis) return fail(PT_ERR_INVALID, "null array");
    if (n > PT_MAX_RAYS_PER_CALL) return fail(PT_ERR_LIMIT, "more than PT_MAX_RAYS_PER_CALL rays in one call");
    int rc = set_device(s);
    if (rc) return rc;
    if (s->n_tris == 0) { HIP_TRY(hipMemsetAsync(vis, 1, n, s->stream)); return PT_OK; }
    uint32_t init[SMALL_WORDS] = {(uint32_t)n};
    HIP_TRY(hipMemcpyAsync(s->d_small, init, sizeof(init), hipMemcpyHostToDevice, s->stream));
    TraceArgs a{};
    a.rays = {ox, oy, oz, dx, dy, dz};
    a.d_count = s->d_small; a.d_next_chunk = s->d_small + SMALL_HEADS0;
    a.dist = max_t; a.vis = vis;
    return launch_trace<TRACE_SHADOW_BOUNDED>(s, a);
}

int pt_raycast_closest_dev_noalloc(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                                   const float* dy, const float* dz, int32_t* idx, float* t, float* u, float* v) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !idx || !t || !u || !v) return fail(PT_ERR_INVALID, "null array");
    if (n > PT_NOALLOC_MAX_RAYS) return fail(PT_ERR_LIMIT, "more than PT_NOALLOC_MAX_RAYS rays in one no-alloc call");
    int rc = set_device(s);
    if (rc) return rc;
    if (s->n_tris == 0) return fill_miss_closest(s, n, idx, t, u, v);
    uint32_t* head = s->d_static_scratch;   /* the looked-up device address, never the array's name */
    hipLaunchKernelGGL(k_noalloc_prepare, dim3(1), dim3(NOALLOC_HEAD_WORDS), 0, s->stream, head, (uint32_t)n);
    HIP_TRY(hipGetLastError());
    TraceArgs a{};
    a.rays = {ox, oy, oz, dx, dy, dz};
    a.d_count = head; a.count_mul = 1; a.d_next_chunk = head + SMALL_HEADS0;
    a.out_idx = idx; a.out_t = t; a.out_u = u; a.out_v = v;
    a.spill = head + NOALLOC_HEAD_WORDS;
    /* no launch timing here: the timers take events from a pool that may have to grow */
    const unsigned grid = std::min<unsigned>((unsigned)s->grid_closest, NOALLOC_GRID);
    hipLaunchKernelGGL(k_trace<TRACE_CLOSEST_MESH>, dim3(grid), dim3(TRACE_BLOCK), 0, s->stream, s->d, a);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

} // extern "C"

/* Host-pointer forms: stage through device buffers that live for the call. */
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { HIP_TRY(hipMalloc(&p, std::max<size_t>(bytes, 4))); return PT_OK; }
};

extern "C" {

/* Up to PICK_MAX_RAYS rays from host arrays: see pt_scene::pick_mu. */
#define PICK_MAX_RAYS 64u
#define PICK_GRID 8u
#define PICK_HEAD_WORDS 256u
static int raycast_closest_small(pt_scene* s, uint32_t n, const float* const* in6, int32_t* idx, float* t, float* u, float* v) {
    std::lock_guard<std::mutex> guard(s->pick_mu);
    /* each resource is made if it is missing, so a call that failed half-way is completed by the next */
    if (!s->pick_stream) HIP_TRY(hipStreamCreateWithFlags(&s->pick_stream, hipStreamNonBlocking));
    if (!s->d_pick) HIP_TRY(hipMalloc((void**)&s->d_pick, (PICK_HEAD_WORDS + (size_t)PICK_GRID * TRACE_BLOCK * (STACK_MAX_CLOSEST - STACK_LDS)) * sizeof(uint32_t)));
    if (!s->h_pick) HIP_TRY(hipHostMalloc((void**)&s->h_pick, 10 * PICK_MAX_RAYS * sizeof(float), hipHostMallocMapped));
    if (!s->h_pick_dev) HIP_TRY(hipHostGetDevicePointer((void**)&s->h_pick_dev, s->h_pick, 0));
    float* h = s->h_pick;
    for (int k = 0; k < 6; k++) memcpy(h + k * PICK_MAX_RAYS, in6[k], n * sizeof(float));
    float* d = s->h_pick_dev;
    hipLaunchKernelGGL(k_noalloc_prepare, dim3(1), dim3(PICK_HEAD_WORDS), 0, s->pick_stream, s->d_pick, n);
    HIP_TRY(hipGetLastError());
    TraceArgs a{};
    a.rays = {d, d + PICK_MAX_RAYS, d + 2 * PICK_MAX_RAYS, d + 3 * PICK_MAX_RAYS, d + 4 * PICK_MAX_RAYS, d + 5 * PICK_MAX_RAYS};
    a.d_count = s->d_pick; a.count_mul = 1; a.d_next_chunk = s->d_pick + SMALL_HEADS0;
    a.out_idx = (int32_t*)(d + 6 * PICK_MAX_RAYS); a.out_t = d + 7 * PICK_MAX_RAYS; a.out_u = d + 8 * PICK_MAX_RAYS; a.out_v = d + 9 * PICK_MAX_RAYS;
    a.spill = s->d_pick + PICK_HEAD_WORDS;
    hipLaunchKernelGGL(k_trace<TRACE_CLOSEST_MESH>, dim3(PICK_GRID), dim3(TRACE_BLOCK), 0, s->pick_stream, s->d, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->pick_stream));
    memcpy(idx, h + 6 * PICK_MAX_RAYS, n * 4); memcpy(t, h + 7 * PICK_MAX_RAYS, n * 4);
    memcpy(u, h + 8 * PICK_MAX_RAYS, n * 4); memcpy(v, h + 9 * PICK_MAX_RAYS, n * 4);
    return PT_OK;
}

int pt_raycast_closest(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                       const float* dy, const float* dz, int32_t* idx, float* t, float* u, float* v) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !idx || !t || !u || !v) return fail(PT_ERR_INVALID, "null array");
    int rc = set_device(s);
    if (rc) return rc;
    if (n <= PICK_MAX_RAYS && s->n_tris > 0) {
        const float* in6[6] = {ox, oy, oz, dx, dy, dz};
        return raycast_closest_small(s, (uint32_t)n, in6, idx, t, u, v);
    }
    DevBuf in[6], out[4];
    const float* src[6] = {ox, oy, oz, dx, dy, dz};
    for (int k = 0; k < 6; k++) {
        rc = in[k].alloc(n * 4); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(in[k].p, src[k], n * 4, hipMemcpyHostToDevice, s->stream));
    }
    for (int k = 0; k < 4; k++) { rc = out[k].alloc(n * 4); if (rc) return rc; }
    rc = pt_raycast_closest_dev(s, n, (float*)in[0].p, (float*)in[1].p, (float*)in[2].p, (float*)in[3].p, (float*)in[4].p,
                                (float*)in[5].p, (int32_t*)out[0].p, (float*)out[1].p, (float*)out[2].p, (float*)out[3].p);
    if (rc) return rc;
    void* dst[4] = {idx, t, u, v};
    for (int k = 0; k < 4; k++) HIP_TRY(hipMemcpyAsync(dst[k], out[k].p, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PT_OK;
}

typedef int (*visibility_dev_fn)(pt_scene*, uint64_t, const float*, const float*, const float*, const float*, const float*,
                                 const float*, const float*, uint8_t*);

static int visibility_from_host(visibility_dev_fn fn, pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz,
                                const float* dx, const float* dy, const float* dz, const float* dist, uint8_t* vis) {
    if (!s) return fail(PT_ERR_INVALID, "null scene");
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !dist || !vis) return fail(PT_ERR_INVALID, "null array");
    int rc = set_device(s);
    if (rc) return rc;
    DevBuf in[7], out;
    const float* src[7] = {ox, oy, oz, dx, dy, dz, dist};
    for (int k = 0; k < 7; k++) {
        rc = in[k].alloc(n * 4); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(in[k].p, src[k], n * 4, hipMemcpyHostToDevice, s->stream));
    }
    rc = out.alloc(n); if (rc) return rc;
    rc = fn(s, n, (float*)in[0].p, (float*)in[1].p, (float*)in[2].p, (float*)in[3].p, (float*)in[4].p, (float*)in[5].p,
            (float*)in[6].p, (uint8_t*)out.p);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(vis, out.p, n, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PT_OK;
}

int pt_raycast_shadow(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                      const float* dy, const float* dz, const float* dist, uint8_t* vis) {
    return visibility_from_host(pt_raycast_shadow_dev, s, n, ox, oy, oz, dx, dy, dz, dist, vis);
}

int pt_raycast_occluded(pt_scene* s, uint64_t n, const float* ox, const float* oy, const float* oz, const float* dx,
                        const float* dy, const float* dz, const float* max_t, uint8_t* vis) {
    return visibility_from_host(pt_raycast_occluded_dev, s, n, ox, oy, oz, dx, dy, dz, max_t, vis);
}

} // extern "C"

/* -------------------------------------------------------------------- film */

/* One render lane: the scratch of one batch in flight, and the host-side staging that the
 * batch's asynchronous copies read from or write to until the lane's stream has been waited for. */
struct RenderLane {
    size_t cap_paths = 0;
    uint32_t cap_lights = 0;
    std::vector<void*> allocs;
    size_t alloc_bytes = 0;
    RenderBuffers rb{};
    DLight* d_lights = nullptr;
    SubPass* d_sub = nullptr;
    uint32_t* d_groups = nullptr; /* group_off[cap+1], entry_off[cap+1], entry_list[cap] */
    size_t cap_sub = 0;
    uint32_t* h_cnt = nullptr;    /* pinned: the batch's counters */
    hipEvent_t film_done = nullptr;
    std::vector<SubPass> h_sub;
    std::vector<uint32_t> h_entry, h_off;
    bool in_flight = false;
    uint32_t flight_lights = 0;
};

#define RENDER_LANES 2

struct pt_film {
    pt_scene* scene = nullptr;
    int width = 0, height = 0;
    float4* d_film = nullptr;
    bool owns = false;
    RenderLane lane[RENDER_LANES]; /* render scratch, allocated on first use */
    /* staging of pt_film_read / pt_films_reduce, made on first use and kept: the split {rgb, passes}
     * copy of the film that goes to the host, and (root film of a reduce) the summed film */
    float* d_read_rgb = nullptr;
    int32_t* d_read_passes = nullptr;
    float4* d_reduce_sum = nullptr;
};

static int film_ensure_read_staging(pt_film* f) {
    const size_t n = (size_t)f->width * f->height;
    if (!f->d_read_rgb) HIP_TRY(hipMalloc((void**)&f->d_read_rgb, n * 12));
    if (!f->d_read_passes) HIP_TRY(hipMalloc((void**)&f->d_read_passes, n * 4));
    return PT_OK;
}

static size_t batch_paths_limit() {
    const char* e = getenv("PT_BATCH_PATHS");
    size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 29;
    if (v < 4096) v = 4096;
    if (v > ((size_t)1 << 30)) v = (size_t)1 << 30;
    return v;
}

/* Number of batches in flight at once: 1 unless PT_RENDER_LANES=2.  Two lanes render 2-3 % more
 * rays per second on a full 1080p step (the lanes fill each other's launch tails), but every
 * kernel then shares the GPU with the other lane's, so per-kernel event times
 * (pt_scene_kernel_time, a profiler's kernel trace) stop meaning busy time; one lane is the
 * default for that reason. */
static int render_lanes() {
    const char* e = getenv("PT_RENDER_LANES");
    return (e && atoi(e) >= 2) ? RENDER_LANES : 1;
}

static void lane_free_scratch(RenderLane* f) {
    for (void* p : f->allocs) (void)hipFree(p);
    f->allocs.clear();
    f->alloc_bytes = 0;
    f->cap_paths = 0; f->cap_lights = 0; f->cap_sub = 0;
    f->d_lights = nullptr; f->d_sub = nullptr; f->d_groups = nullptr;
}

static void film_free_scratch(pt_film* f) {
    for (RenderLane& ln : f->lane) {
        lane_free_scratch(&ln);
        if (ln.h_cnt) { (void)hipHostFree(ln.h_cnt); ln.h_cnt = nullptr; }
        if (ln.film_done) { (void)hipEventDestroy(ln.film_done); ln.film_done = nullptr; }
    }
}

template <class T>
static int film_alloc(RenderLane* f, T** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    f->alloc_bytes += std::max<size_t>(count, 1) * sizeof(T);
    f->allocs.push_back(p);
    *out = (T*)p;
    return PT_OK;
}

/* The lane must be idle (its stream waited for) when this is called. */
static int lane_ensure_scratch(RenderLane* f, size_t paths, uint32_t n_lights, size_t n_sub) {
    if (!f->h_cnt) HIP_TRY(hipHostMalloc((void**)&f->h_cnt, CNT_WORDS * sizeof(uint32_t), hipHostMallocDefault));
    if (!f->film_done) HIP_TRY(hipEventCreateWithFlags(&f->film_done, hipEventDisableTiming));
    uint32_t L = std::max<uint32_t>(n_lights, 1);
    if (paths <= f->cap_paths && L <= f->cap_lights && n_sub <= f->cap_sub) return PT_OK;
    lane_free_scratch(f);
    size_t P = paths, S = paths * L, G = std::max<size_t>(n_sub, 16);
    int rc = 0;
    RenderBuffers& b = f->rb;
    for (int k = 0; k < 2 && !rc; k++) {
        float** q[6] = {&b.q[k].ox, &b.q[k].oy, &b.q[k].oz, &b.q[k].dx, &b.q[k].dy, &b.q[k].dz};
        for (int c = 0; c < 6 && !rc; c++) rc = film_alloc(f, q[c], P);
        if (!rc) rc = film_alloc(f, &b.qpath[k], P);
    }
    if (!rc) rc = film_alloc(f, &b.hit, P);
    float** per_path[8] = {&b.rad_r, &b.rad_g, &b.rad_b, &b.thr, nullptr, nullptr, nullptr, nullptr};
    for (int c = 0; per_path[c] && !rc; c++) rc = film_alloc(f, per_path[c], P);
    if (!rc) rc = film_alloc(f, &b.pixel, P);
    if (!rc) rc = film_alloc(f, &b.sample, P);
    float** per_sh[14] = {&b.sh.ox, &b.sh.oy, &b.sh.oz, &b.sh.dx, &b.sh.dy, &b.sh.dz, &b.sh_dist, &b.sh_wr[0], &b.sh_wg[0], &b.sh_wb[0],
                          &b.sh_wr[1], &b.sh_wg[1], &b.sh_wb[1], nullptr};
    for (int c = 0; per_sh[c] && !rc; c++) rc = film_alloc(f, per_sh[c], S);
    if (!rc) rc = film_alloc(f, &b.sh_vis[0], S);
    if (!rc) rc = film_alloc(f, &b.sh_vis[1], S);
    if (!rc) rc = film_alloc(f, &b.sh_target, S);
    if (!rc) rc = film_alloc(f, &b.sh_defer, S);
    if (!rc) rc = film_alloc(f, &b.cnt, (size_t)CNT_WORDS);
    if (!rc) rc = film_alloc(f, &f->d_lights, (size_t)PT_MAX_LIGHTS);
    if (!rc) rc = film_alloc(f, &f->d_sub, G);
    if (!rc) rc = film_alloc(f, &f->d_groups, 3 * G + 2);
    if (rc) { lane_free_scratch(f); return rc; }
    f->cap_paths = P; f->cap_lights = L; f->cap_sub = G;
    return PT_OK;
}

static void fill_cam(CamParams& c, const pt_camera* cam, int width, int height) {
    c.origin = v3(cam->origin[0], cam->origin[1], cam->origin[2]);
    c.direction = v3(cam->direction[0], cam->direction[1], cam->direction[2]);
    c.width = width; c.height = height;
    cam_basis(c, v3(cam->up[0], cam->up[1], cam->up[2]));
    c.image_plane_width = cam->image_plane_width;
    c.plane_of_focus_distance = cam->plane_of_focus_distance;
    c.dof_dispersion = cam->dof_dispersion;
    c.sky = v3(cam->sky_color[0], cam->sky_color[1], cam->sky_color[2]);
    c.bounce_mode = cam->bounce_mode;
    c.shadow_mode = cam->shadow_mode;
}

/* The bounce loop of cast_ray (reference integrator.cpp:51-156) for the paths whose first rays sit
 * in queue 0 (b.cnt[CNT_QUEUE] of them, counters otherwise zero): per depth the closest-hit trace,
 * the shade kernel, the shadow traces, and after the last depth the light samples' sum.  Depth d
 * is the reference's recursions = max_depth - d.  Everything is queued on ctx.stream with no host
 * round trip. */
static int enqueue_bounces(pt_scene* s, RenderBuffers& b, const LaunchCtx& ctx, const CamParams& cam, const DLight* d_lights,
                           uint32_t n_lights, uint32_t max_depth, uint64_t seed, BranchOpts br = BranchOpts{0.8f, 0u, 0}) {
    hipStream_t st = ctx.stream;
    const int light_grid = s->num_cus * 8;
    const int shade_grid = s->num_cus * SHADE_GRID_MUL;
    /* PT_TRACE_ALL_SHADOW=1: trace the shadow rays of zero-weight light samples too (same picture; for checks) */
    const char* all_env = getenv("PT_TRACE_ALL_SHADOW");
    const int trace_all = (all_env && atoi(all_env) != 0) ? 1 : 0;
    /* PT_OVERLAP_SHADOW=1 (an experiment, off by default): the shadow traces of depth d run on the scene's
     * second stream beside the closest-hit trace of depth d+1 -- both need only k_shade(d) -- and
     * k_shade(d+1), which reads their answers and refills their queue, waits for them.  Measured on
     * MI355X: see the overlap entry of the tuning notes (the two persistent grids share the compute
     * units; nothing is gained). */
    const char* ov_env = getenv("PT_OVERLAP_SHADOW");
    bool overlap = ov_env && atoi(ov_env) == 1 && st == s->stream && n_lights > 0;
    LaunchCtx side = ctx;
    if (overlap) {
        int rc = ensure_second_lane(s);
        if (rc) return rc;
        if (!s->ev_fork) HIP_TRY(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
        if (!s->ev_join) HIP_TRY(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
        side = LaunchCtx{s->stream2, s->d_spill2};
    }
    bool join_pending = false;
    for (uint32_t depth = 0; depth <= max_depth; depth++) {
        const int cur = depth & 1;
        TraceArgs a{};
        a.rays = {b.q[cur].ox, b.q[cur].oy, b.q[cur].oz, b.q[cur].dx, b.q[cur].dy, b.q[cur].dz};
        a.d_count = b.cnt + CNT_QUEUE + depth;
        a.d_next_chunk = b.cnt + CNT_CHUNK(0, depth);
        a.out_hit = b.hit;
        int rc = launch_trace<TRACE_CLOSEST_FLAT>(s, a, &ctx);
        if (rc) return rc;
        if (join_pending) { HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0)); join_pending = false; }   /* the previous depth's shadow answers */
        {
            ScopedTimer tm(s, PT_KERNEL_SHADE, st);
            hipLaunchKernelGGL(k_shade, dim3(shade_grid), dim3(STREAM_BLOCK), 0, st, s->d, cam, d_lights, n_lights, depth,
                               depth == max_depth ? 1 : 0, trace_all, seed, br, b);
            HIP_TRY(hipGetLastError());
        }
        if (n_lights > 0) {
            TraceArgs sa{};
            sa.rays = {b.sh.ox, b.sh.oy, b.sh.oz, b.sh.dx, b.sh.dy, b.sh.dz};
            sa.d_count = b.cnt + CNT_SHADOW(depth); /* the light samples of this depth whose weight is not zero */
            sa.count_mul = 1;
            sa.target = b.sh_target;
            sa.d_next_chunk = b.cnt + CNT_CHUNK(1, depth);
            sa.dist = b.sh_dist; sa.vis = b.sh_vis[cur];
            if (overlap) { HIP_TRY(hipEventRecord(s->ev_fork, st)); HIP_TRY(hipStreamWaitEvent(side.stream, s->ev_fork, 0)); }
            if (cam.shadow_mode == PT_SHADOW_BOUNDED) {
                rc = launch_trace<TRACE_SHADOW_BOUNDED>(s, sa, &side);
                if (rc) return rc;
            } else {
                sa.defer_list = b.sh_defer; sa.defer_count = b.cnt + CNT_DEFER + depth;
                rc = launch_trace<TRACE_SHADOW_ANY>(s, sa, &side);
                if (rc) return rc;
                sa.d_count = b.cnt + CNT_DEFER + depth;
                sa.count_mul = 1;
                sa.d_next_chunk = b.cnt + CNT_CHUNK(2, depth);
                rc = launch_trace<TRACE_SHADOW_EXACT>(s, sa, &side);
                if (rc) return rc;
            }
            if (overlap) {
                HIP_TRY(hipEventRecord(s->ev_join, side.stream));
                join_pending = true;
                if (depth == max_depth) { HIP_TRY(hipStreamWaitEvent(st, s->ev_join, 0)); join_pending = false; }
            }
            if (depth == max_depth) { /* the other depths' samples are added by the next depth's k_shade */
                ScopedTimer tm(s, PT_KERNEL_ACCUMULATE, st);
                hipLaunchKernelGGL(k_accumulate, dim3(light_grid), dim3(256), 0, st, n_lights, depth, b);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    return PT_OK;
}

/* Queues one batch (already clamped, grouped and numbered, in ln.h_sub) on the lane's stream
 * with no host round trip.  The film kernel waits for `after` (the previous batch's film
 * kernel, which may be on the other lane) so that batches reach the film in order; the batch's
 * counters land in ln.h_cnt once the stream has been waited for. */
static int render_batch(pt_film* f, RenderLane& ln, const LaunchCtx& ctx, const CamParams& cam, uint32_t n_lights,
                        uint32_t n_paths, uint64_t seed, hipEvent_t after) {
    pt_scene* s = f->scene;
    RenderBuffers& b = ln.rb;
    hipStream_t st = ctx.stream;
    const std::vector<SubPass>& sub = ln.h_sub;
    /* rectangle groups for the film kernel: per group its thread range and the list of its entries
     * in batch order.  d_groups holds group_off[cap+1], entry_off[cap+1], entry_list[cap]. */
    std::vector<uint32_t>&g_off = ln.h_off, &g_entry = ln.h_entry;
    const uint32_t n_sub = (uint32_t)sub.size();
    g_off.clear();
    std::vector<uint32_t> group_of(n_sub), count;
    uint32_t total = 0;
    for (uint32_t e = 0; e < n_sub; e++) {
        if (sub[e].group_first == e) { group_of[e] = (uint32_t)count.size(); count.push_back(0); g_off.push_back(total); total += sub[e].padded; }
        else group_of[e] = group_of[sub[e].group_first];
        count[group_of[e]]++;
    }
    g_off.push_back(total);
    const uint32_t n_groups = (uint32_t)count.size();
    g_entry.assign((size_t)n_groups + 1 + n_sub, 0);          /* entry_off then entry_list */
    for (uint32_t g = 0; g < n_groups; g++) g_entry[g + 1] = g_entry[g] + count[g];
    {
        std::vector<uint32_t> fill(g_entry.begin(), g_entry.begin() + n_groups);
        for (uint32_t e = 0; e < n_sub; e++) g_entry[(size_t)n_groups + 1 + fill[group_of[e]]++] = e;
    }
    uint32_t* d_group_off = ln.d_groups;
    uint32_t* d_entry_off = ln.d_groups + ln.cap_sub + 1;
    uint32_t* d_entry_list = ln.d_groups + 2 * (ln.cap_sub + 1);
    HIP_TRY(hipMemcpyAsync(ln.d_sub, sub.data(), sub.size() * sizeof(SubPass), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_group_off, g_off.data(), (n_groups + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_entry_off, g_entry.data(), (n_groups + 1) * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_entry_list, g_entry.data() + n_groups + 1, n_sub * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b.cnt, 0, CNT_WORDS * 4, st));
    {
        ScopedTimer tm(s, PT_KERNEL_GENERATE, st);
        hipLaunchKernelGGL(k_generate, dim3((n_paths + STREAM_BLOCK * GEN_ROWS - 1) / (STREAM_BLOCK * GEN_ROWS)), dim3(STREAM_BLOCK), 0, st, cam, ln.d_sub, (uint32_t)sub.size(),
                           n_paths, seed, b);
        HIP_TRY(hipGetLastError());
    }
    {
        int rc = enqueue_bounces(s, b, ctx, cam, ln.d_lights, n_lights, 10, seed); /* cast_ray(ray, 10, 1): recursions 10..0 */
        if (rc) return rc;
    }
    if (after) HIP_TRY(hipStreamWaitEvent(st, after, 0));
    {
        ScopedTimer tm(s, PT_KERNEL_ACCUMULATE, st);
        hipLaunchKernelGGL(k_film, dim3((total + 255) / 256), dim3(256), 0, st, ln.d_sub, d_group_off, d_entry_off, d_entry_list,
                           n_groups, total, f->width, f->d_film, b);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ln.film_done, st));
    HIP_TRY(hipMemcpyAsync(ln.h_cnt, b.cnt, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    return PT_OK;
}

extern "C" {

int pt_film_create(pt_scene* s, int width, int height, float* d_external, pt_film** out) {
    if (!s || !out) return fail(PT_ERR_INVALID, "null argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (int64_t)width * height > ((int64_t)1 << 28)) return fail(PT_ERR_INVALID, "bad film size");
    int rc = set_device(s);
    if (rc) return rc;
    pt_film* f = new pt_film();
    f->scene = s; f->width = width; f->height = height;
    if (d_external) { f->d_film = (float4*)d_external; f->owns = false; }
    else {
        if (hipMalloc((void**)&f->d_film, (size_t)width * height * sizeof(float4)) != hipSuccess) { delete f; return fail(PT_ERR_DEVICE, "hipMalloc failed for the film"); }
        f->owns = true;
        if (hipMemsetAsync(f->d_film, 0, (size_t)width * height * sizeof(float4), s->stream) != hipSuccess) { (void)hipFree(f->d_film); delete f; return fail(PT_ERR_DEVICE, "hipMemset failed"); }
    }
    *out = f;
    return PT_OK;
}

void pt_film_destroy(pt_film* f) {
    if (!f) return;
    (void)hipSetDevice(f->scene->device);
    (void)hipStreamSynchronize(f->scene->stream);
    if (f->scene->stream2) (void)hipStreamSynchronize(f->scene->stream2);
    film_free_scratch(f);
    if (f->d_read_rgb) (void)hipFree(f->d_read_rgb);
    if (f->d_read_passes) (void)hipFree(f->d_read_passes);
    if (f->d_reduce_sum) (void)hipFree(f->d_reduce_sum);
    if (f->owns && f->d_film) (void)hipFree(f->d_film);
    delete f;
}

int pt_film_zero(pt_film* f) {
    if (!f) return fail(PT_ERR_INVALID, "null film");
    int rc = set_device(f->scene);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(f->d_film, 0, (size_t)f->width * f->height * sizeof(float4), f->scene->stream));
    return PT_OK;
}

float* pt_film_device_ptr(pt_film* f) { return f ? (float*)f->d_film : nullptr; }

int pt_film_read(pt_film* f, float* rgb_sum, int32_t* passes) {
    if (!f || !rgb_sum || !passes) return fail(PT_ERR_INVALID, "null argument");
    pt_scene* s = f->scene;
    int rc = set_device(s);
    if (rc) return rc;
    size_t n = (size_t)f->width * f->height;
    rc = film_ensure_read_staging(f);
    if (rc) return rc;
    hipLaunchKernelGGL(k_film_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, f->d_film, n, f->d_read_rgb, f->d_read_passes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_sum, f->d_read_rgb, n * 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(passes, f->d_read_passes, n * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PT_OK;
}

/* RCCL entry points, resolved on first use (see pt_films_reduce in pt_hip.h). */
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::vector<int> devices;        /* device set of the cached communicators */
    std::vector<ncclComm_t> comms;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;   /* pt_films_reduce holds it from loading the library to the end of the collective */

static int rccl_load() {
    if (g_rccl.lib) return PT_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* nm : names) { lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return fail(PT_ERR_DEVICE, std::string("cannot load RCCL: ") + dlerror());
    RcclApi r;
    r.lib = lib;
    r.CommInitAll = (decltype(r.CommInitAll))dlsym(lib, "ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(lib, "ncclCommDestroy");
    r.Reduce = (decltype(r.Reduce))dlsym(lib, "ncclReduce");
    r.GroupStart = (decltype(r.GroupStart))dlsym(lib, "ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))dlsym(lib, "ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.Reduce || !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
        dlclose(lib);
        return fail(PT_ERR_DEVICE, "RCCL library lacks an expected entry point");
    }
    g_rccl = r;
    return PT_OK;
}

#define RCCL_TRY(expr)                                                                               \
    do {                                                                                             \
        ncclResult_t r_ = (expr);                                                                    \
        if (r_ != ncclSuccess) return fail(PT_ERR_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

int pt_films_reduce(pt_film* const* films, int n, float* rgb_sum, int32_t* passes) {
    if (!films || n < 1 || !rgb_sum || !passes) return fail(PT_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; i++) {
        if (!films[i]) return fail(PT_ERR_INVALID, "null film");
        if (films[i]->width != films[0]->width || films[i]->height != films[0]->height) return fail(PT_ERR_INVALID, "films differ in size");
        for (int j = 0; j < i; j++)
            if (films[j]->scene->device == films[i]->scene->device) return fail(PT_ERR_INVALID, "two films on one GPU");
    }
    const char* force = getenv("PT_FILMS_REDUCE");   /* "rccl": use the collective even for one film */
    if (n == 1 && !(force && strcmp(force, "rccl") == 0)) return pt_film_read(films[0], rgb_sum, passes);
    /* the loaded library and the cached communicators are process-wide: one reduce at a time */
    std::lock_guard<std::mutex> guard(g_rccl_mu);
    int rc = rccl_load();
    if (rc) return rc;
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; i++) devs[(size_t)i] = films[i]->scene->device;
    if (g_rccl.devices != devs) {
        for (ncclComm_t c : g_rccl.comms) (void)g_rccl.CommDestroy(c);
        g_rccl.comms.assign((size_t)n, nullptr);
        g_rccl.devices.clear();
        RCCL_TRY(g_rccl.CommInitAll(g_rccl.comms.data(), n, devs.data()));
        g_rccl.devices = devs;
    }
    pt_film* root = films[0];
    const size_t npx = (size_t)root->width * root->height;
    rc = set_device(root->scene);
    if (rc) return rc;
    /* the root film keeps the summed film and the split copy from call to call */
    rc = film_ensure_read_staging(root);
    if (rc) return rc;
    if (!root->d_reduce_sum) HIP_TRY(hipMalloc((void**)&root->d_reduce_sum, npx * 16));
    /* Every film's stream is ordered after its rendering, so the reduce reads finished films.
     * Whatever fails between GroupStart and GroupEnd, the group is closed before returning: an
     * open group would swallow the next collective of this thread. */
    RCCL_TRY(g_rccl.GroupStart());
    std::string group_error;
    for (int i = 0; i < n && group_error.empty(); i++) {
        hipError_t he = hipSetDevice(films[i]->scene->device);
        if (he != hipSuccess) { group_error = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
        ncclResult_t nr = g_rccl.Reduce(films[i]->d_film, i == 0 ? (void*)root->d_reduce_sum : (void*)films[i]->d_film, npx * 4, ncclFloat,
                                        ncclSum, 0, g_rccl.comms[(size_t)i], films[i]->scene->stream);
        if (nr != ncclSuccess) group_error = std::string("ncclReduce: ") + g_rccl.GetErrorString(nr);
    }
    {
        ncclResult_t nr = g_rccl.GroupEnd();
        if (nr != ncclSuccess && group_error.empty()) group_error = std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(nr);
    }
    if (!group_error.empty()) {
        /* the communicators may be left in an error state: make the next call build new ones */
        for (ncclComm_t c : g_rccl.comms) if (c) (void)g_rccl.CommDestroy(c);
        g_rccl.comms.clear(); g_rccl.devices.clear();
        (void)hipSetDevice(root->scene->device);
        return fail(PT_ERR_DEVICE, group_error);
    }
    for (int i = n - 1; i >= 0; i--) {
        HIP_TRY(hipSetDevice(films[i]->scene->device));
        HIP_TRY(hipStreamSynchronize(films[i]->scene->stream));
    }
    pt_scene* s = root->scene;   /* the current device is the root's again */
    hipLaunchKernelGGL(k_film_split, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s->stream, (const float4*)root->d_reduce_sum, npx, root->d_read_rgb, root->d_read_passes);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rgb_sum, root->d_read_rgb, npx * 12, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(passes, root->d_read_passes, npx * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PT_OK;
}

int pt_film_add(pt_film* f, const float* rgb_sum, const int32_t* passes) {
    if (!f || !rgb_sum || !passes) return fail(PT_ERR_INVALID, "null argument");
    pt_scene* s = f->scene;
    int rc = set_device(s);
    if (rc) return rc;
    size_t n = (size_t)f->width * f->height;
    DevBuf d_rgb, d_p;
    rc = d_rgb.alloc(n * 12); if (rc) return rc;
    rc = d_p.alloc(n * 4); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_rgb.p, rgb_sum, n * 12, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(d_p.p, passes, n * 4, hipMemcpyHostToDevice, s->stream));
    hipLaunchKernelGGL(k_film_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, f->d_film, n, (const float*)d_rgb.p, (const int32_t*)d_p.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return PT_OK;
}

int pt_render_passes(pt_film* f, const pt_camera* cam, const pt_light* lights, int n_lights, const pt_tile_pass* passes,
                     int n_passes, uint64_t seed, pt_render_stats* stats) {
    if (!f || !cam || n_passes < 0 || (n_passes > 0 && !passes) || n_lights < 0 || (n_lights > 0 && !lights))
        return fail(PT_ERR_INVALID, "bad argument");
    if (n_lights > PT_MAX_LIGHTS) return fail(PT_ERR_LIMIT, "more than PT_MAX_LIGHTS lights");
    if (cam->bounce_mode > PT_BOUNCE_LAMBERT) return fail(PT_ERR_INVALID, "unknown bounce_mode");
    if (cam->shadow_mode > PT_SHADOW_BOUNDED) return fail(PT_ERR_INVALID, "unknown shadow_mode");
    pt_scene* s = f->scene;
    int rc = set_device(s);
    if (rc) return rc;
    if (stats) memset(stats, 0, sizeof(*stats));
    struct CallEvents { /* destroyed on every way out */
        hipEvent_t a = nullptr, b = nullptr;
        ~CallEvents() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    if (stats) {
        HIP_TRY(hipEventCreate(&ev.a)); HIP_TRY(hipEventCreate(&ev.b));
        HIP_TRY(hipEventRecord(ev.a, s->stream));
    }
    CamParams cp;
    fill_cam(cp, cam, f->width, f->height);
    /* PassDescriptor::clamp_bounds (reference integrator.cpp:164-183) */
    struct Clamped { pt_tile_pass p; uint32_t bx; size_t padded; };
    std::vector<Clamped> work;
    size_t total_paths = 0, max_padded = 0;
    for (int i = 0; i < n_passes; i++) {
        pt_tile_pass p = passes[i];
        if (p.x < 0) p.x = 0;
        if (p.y < 0) p.y = 0;
        if (p.x > f->width - 1) p.x = f->width - 1;
        if (p.y > f->height - 1) p.y = f->height - 1;
        if (p.w == -1) p.w = f->width;
        if (p.h == -1) p.h = f->height;
        if (p.w > f->width - p.x) p.w = f->width - p.x;
        if (p.h > f->height - p.y) p.h = f->height - p.y;
        if (p.w <= 0 || p.h <= 0 || p.n_samples == 0) continue;
        uint32_t bx = (uint32_t)(p.w + 7) / 8, by = (uint32_t)(p.h + 7) / 8;
        size_t padded = (size_t)bx * by * 64;
        work.push_back({p, bx, padded});
        total_paths += padded * p.n_samples;
        max_padded = std::max(max_padded, padded);
    }
    /* Paths in flight: as many as allowed (a trace launch pays a fixed ramp-up and tail, the
     * latency of its longest rays, so longer queues run closer to the kernels' asymptotic rate:
     * 32 -> 128 -> 256 spp of 1080p per call is +13 % and another +4 %), but the scratch (96 bytes
     * per path + 62 per shadow ray; 152 GB for 2^29 paths and three lights) stays within nine tenths
     * of the memory that is free on the GPU and a batch's shadow rays within what one trace launch
     * accepts. */
    size_t limit = batch_paths_limit();
    {
        const size_t L = (size_t)std::max(n_lights, 1);
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        /* what is free now plus what this film's lanes already hold, less a tenth */
        size_t avail = free_b;
        for (const RenderLane& ln : f->lane) avail += ln.alloc_bytes;
        limit = std::min(limit, (avail / 10 * 9) / (96 + 62 * L));
        limit = std::max<size_t>(limit, 4096);
    }
    /* With two lanes the call's paths go out as (at least) two batches that run on two streams
     * at once, so that one batch's kernels fill the compute units that the other's leave idle
     * (the tail of every trace launch, the memory-bound shade and accumulate kernels).  Each
     * lane gets half of the allowance.  The film kernels still run in batch order. */
    int n_lanes = render_lanes();
    size_t per_batch = limit;
    if (n_lanes == 2) {
        const size_t floor_paths = (size_t)1 << 22; /* below this a second batch costs more in launches than it hides */
        per_batch = std::min(limit / 2, std::max((total_paths + 1) / 2, floor_paths) + max_padded - 1);
        if (max_padded > per_batch || total_paths <= floor_paths) { n_lanes = 1; per_batch = limit; }
    }
    per_batch = std::min(per_batch, (size_t)(PT_MAX_RAYS_PER_CALL / (size_t)std::max(n_lights, 1)));
    /* A pass whose samples do not fit one batch is cut into horizontal bands that do, each with ALL of
     * its samples, rather than into sample ranges of the whole rectangle: the more samples of a screen
     * patch are in flight together, the fewer distinct leaves the rays in flight touch (measured on the
     * 21.5 M-triangle scene at 3840x2160: 16 / 32 / 64 samples per batch = 2 047 / 2 190 / 2 262
     * Mrays/s).  A pixel's samples still reach the film in sample order, so the picture is the same.
     * Bands that are still too large (fewer than 8 rows fit) are cut by samples below. */
    {
        std::vector<Clamped> cut;
        for (const Clamped& c : work) {
            const size_t row_paths = (size_t)c.bx * 64;   /* one row of 8x8 blocks, one sample */
            const size_t block_rows = (size_t)(c.p.h + 7) / 8;
            size_t fit = per_batch / (row_paths * c.p.n_samples);   /* block rows whose samples fit one batch */
            if (c.padded * c.p.n_samples <= per_batch || block_rows <= 1) { cut.push_back(c); continue; }
            if (fit < 1) fit = 1;
            /* equal bands rather than full ones and a remainder */
            const size_t n_bands = (block_rows + fit - 1) / fit;
            const size_t rows_each = (block_rows + n_bands - 1) / n_bands;
            for (size_t r0 = 0; r0 < block_rows; r0 += rows_each) {
                Clamped b = c;
                b.p.y = c.p.y + (int32_t)(r0 * 8);
                b.p.h = (int32_t)std::min<size_t>(rows_each * 8, (size_t)c.p.h - r0 * 8);
                b.padded = (size_t)c.bx * ((size_t)(b.p.h + 7) / 8) * 64;
                cut.push_back(b);
            }
        }
        work.swap(cut);
        for (const Clamped& c : work)
            if (c.padded > per_batch) return fail(PT_ERR_LIMIT, "one row of 8x8-pixel blocks of this rectangle exceeds PT_BATCH_PATHS");
    }
    LaunchCtx ctx[RENDER_LANES] = {{s->stream, s->d_spill}, {nullptr, nullptr}};
    if (n_lanes == 2) {
        rc = ensure_second_lane(s);
        if (rc) return rc;
        ctx[1] = {s->stream2, s->d_spill2};
        /* the second stream starts after whatever the scene's stream already holds */
        HIP_TRY(hipEventRecord(s->ev_order, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream2, s->ev_order, 0));
    }
    uint64_t counts[4] = {0, 0, 0, 0};
    DLight hl[PT_MAX_LIGHTS];
    for (int l = 0; l < n_lights; l++) {
        hl[l].pos = v3(lights[l].position[0], lights[l].position[1], lights[l].position[2]);
        hl[l].color = v3(lights[l].color[0], lights[l].color[1], lights[l].color[2]);
    }
    /* waits for a lane's batch and takes its counters */
    auto harvest = [&](int li) -> int {
        RenderLane& ln = f->lane[li];
        if (!ln.in_flight) return PT_OK;
        ln.in_flight = false;
        HIP_TRY(hipStreamSynchronize(ctx[li].stream));
        const uint32_t* h = ln.h_cnt;
        counts[0] += h[CNT_QUEUE + 0];
        for (int d = 0; d <= 10; d++) {
            counts[1] += h[CNT_QUEUE + d];
            counts[2] += (uint64_t)h[CNT_QUEUE + d + 1] * ln.flight_lights;
            if (ln.flight_lights) counts[3] += h[CNT_SHADOW(d)];
        }
        return PT_OK;
    };
    /* whatever happens, nothing of this call is in flight when it returns: the lanes' copies
     * read `hl` and write the pinned counters */
    struct Drain {
        pt_film* f; LaunchCtx* ctx;
        ~Drain() { for (int li = 0; li < RENDER_LANES; li++) if (f->lane[li].in_flight) { (void)hipStreamSynchronize(ctx[li].stream); f->lane[li].in_flight = false; } }
    } drain{f, ctx};
    /* PT_PATH_UNIT = 1, 2, 4 or 8: side of the screen patch whose samples are numbered together (see SubPass) */
    uint32_t unit_side_log2 = PATH_UNIT_SIDE_LOG2;
    if (const char* e = getenv("PT_PATH_UNIT")) { int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8) unit_side_log2 = v == 1 ? 0u : v == 2 ? 1u : v == 4 ? 2u : 3u; }
    std::vector<SubPass> batch;
    size_t batch_paths = 0;
    int next_lane = 0;
    hipEvent_t last_film = nullptr;
    auto flush = [&]() -> int {
        if (batch.empty()) return PT_OK;
        const int li = next_lane;
        RenderLane& ln = f->lane[li];
        int r2 = harvest(li);
        if (r2) return r2;
        size_t cap = ((batch_paths + ((size_t)1 << 20) - 1) >> 20) << 20;
        /* a call that fills more than half of the allowance takes all of it at once: growing the
         * scratch later means freeing and allocating it anew, which costs seconds at this size */
        if (total_paths > per_batch / 2 && cap > ln.cap_paths) cap = std::max(cap, per_batch);
        r2 = lane_ensure_scratch(&ln, cap, (uint32_t)n_lights, batch.size());
        if (r2) return r2;
        if (n_lights > 0) HIP_TRY(hipMemcpyAsync(ln.d_lights, hl, sizeof(DLight) * n_lights, hipMemcpyHostToDevice, ctx[li].stream));
        ln.h_sub.swap(batch);
        ln.in_flight = true;
        ln.flight_lights = (uint32_t)n_lights;
        r2 = render_batch(f, ln, ctx[li], cp, (uint32_t)n_lights, (uint32_t)batch_paths, seed, last_film);
        if (r2) return r2;
        last_film = ln.film_done;
        next_lane = (next_lane + 1) % n_lanes;
        batch.clear(); batch_paths = 0;
        return PT_OK;
    };
    for (const Clamped& c : work) {
        const pt_tile_pass& p = c.p;
        const size_t padded = c.padded;
        uint32_t done = 0;
        while (done < p.n_samples) {
            size_t room = (per_batch - batch_paths) / padded;
            /* a different rectangle overlapping one already in the batch would race in the film kernel */
            bool overlap = false;
            for (const SubPass& e : batch)
                if (!(e.x == p.x && e.y == p.y && e.w == p.w && e.h == p.h) &&
                    p.x < e.x + e.w && e.x < p.x + p.w && p.y < e.y + e.h && e.y < p.y + p.h) { overlap = true; break; }
            if (room == 0 || overlap || batch.size() >= 16384) { rc = flush(); if (rc) return rc; continue; }
            uint32_t take = (uint32_t)std::min<size_t>(room, p.n_samples - done);
            SubPass sp{};
            sp.x = p.x; sp.y = p.y; sp.w = p.w; sp.h = p.h;
            sp.first_sample = p.first_sample + done; sp.n_samples = take;
            sp.blocks_x = c.bx; sp.padded = (uint32_t)padded; sp.path0 = (uint32_t)batch_paths;
            sp.side_log2 = unit_side_log2;
            sp.group_first = (uint32_t)batch.size();
            for (uint32_t e = 0; e < batch.size(); e++)
                if (batch[e].x == p.x && batch[e].y == p.y && batch[e].w == p.w && batch[e].h == p.h) { sp.group_first = batch[e].group_first; break; }
            batch.push_back(sp);
            batch_paths += padded * take;
            done += take;
        }
    }
    rc = flush();
    if (rc) return rc;
    for (int li = 0; li < RENDER_LANES; li++) { rc = harvest(li); if (rc) return rc; }
    if (s->pending.size() > 8192) { rc = resolve_timing(s); if (rc) return rc; }
    if (stats) {
        HIP_TRY(hipEventRecord(ev.b, s->stream));
        HIP_TRY(hipEventSynchronize(ev.b));
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        stats->camera_samples = counts[0]; stats->closest_rays = counts[1]; stats->shadow_rays = counts[2];
        stats->shadow_rays_traced = counts[3];
        stats->gpu_ms = ms;
    }
    return PT_OK;
}

int pt_cast_rays(pt_scene* s, const pt_camera* cam, const pt_light* lights, int n_lights, uint64_t n, const float* ox,
                 const float* oy, const float* oz, const float* dx, const float* dy, const float* dz, const uint32_t* stream_ids,
                 uint32_t sample, int recursions, int branches, uint64_t seed, float* rgb, pt_render_stats* stats) {
    if (!s || !cam || n_lights < 0 || (n_lights > 0 && !lights)) return fail(PT_ERR_INVALID, "bad argument");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n == 0) return PT_OK;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !rgb) return fail(PT_ERR_INVALID, "null array");
    if (n_lights > PT_MAX_LIGHTS) return fail(PT_ERR_LIMIT, "more than PT_MAX_LIGHTS lights");
    if (recursions < 0 || recursions > 10) return fail(PT_ERR_LIMIT, "recursions outside 0..10");
    if (branches > PT_MAX_BRANCHES) return fail(PT_ERR_LIMIT, "more than PT_MAX_BRANCHES branches");
    if (cam->bounce_mode > PT_BOUNCE_LAMBERT || cam->shadow_mode > PT_SHADOW_BOUNDED) return fail(PT_ERR_INVALID, "unknown bounce_mode or shadow_mode");
    if (n * (uint64_t)std::max(n_lights, 1) > PT_MAX_RAYS_PER_CALL || n > ((uint64_t)1 << 28)) return fail(PT_ERR_LIMIT, "too many rays in one pt_cast_rays call");
    /* the loop over branches at integrator.cpp:77 does not run for branches <= 0: no bounce term */
    if (branches <= 0) { recursions = 0; branches = 1; }
    if (recursions == 0) branches = 1;
    int rc = set_device(s);
    if (rc) return rc;
    /* scratch that lives for the call */
    /* (its destructor waits for the stream first: on an early return copies and kernels that use these
     * buffers, and the host arrays below, may still be queued) */
    struct Scratch { RenderLane ln; uint32_t* d_ids = nullptr; hipStream_t st = nullptr; ~Scratch() { if (st) (void)hipStreamSynchronize(st); lane_free_scratch(&ln); if (ln.h_cnt) (void)hipHostFree(ln.h_cnt); if (ln.film_done) (void)hipEventDestroy(ln.film_done); if (d_ids) (void)hipFree(d_ids); } };
    /* host staging declared before `sc`, so that it outlives the wait in sc's destructor */
    std::vector<float> r((size_t)n), g((size_t)n), bl((size_t)n);
    DLight hl[PT_MAX_LIGHTS];
    const uint32_t n32 = (uint32_t)n;
    Scratch sc;
    sc.st = s->stream;
    rc = lane_ensure_scratch(&sc.ln, (size_t)n, (uint32_t)n_lights, 1);
    if (rc) return rc;
    RenderBuffers& b = sc.ln.rb;
    hipStream_t st = s->stream;
    CamParams cp;
    fill_cam(cp, cam, 1, 1);   /* only sky colour and the two modes are used past the camera rays */
    for (int l = 0; l < n_lights; l++) {
        hl[l].pos = v3(lights[l].position[0], lights[l].position[1], lights[l].position[2]);
        hl[l].color = v3(lights[l].color[0], lights[l].color[1], lights[l].color[2]);
    }
    if (n_lights > 0) HIP_TRY(hipMemcpyAsync(sc.ln.d_lights, hl, sizeof(DLight) * n_lights, hipMemcpyHostToDevice, st));
    if (stream_ids) {
        HIP_TRY(hipMalloc((void**)&sc.d_ids, n * 4));
        HIP_TRY(hipMemcpyAsync(sc.d_ids, stream_ids, n * 4, hipMemcpyHostToDevice, st));
    }
    const float* src[6] = {ox, oy, oz, dx, dy, dz};
    float* dst[6] = {b.q[0].ox, b.q[0].oy, b.q[0].oz, b.q[0].dx, b.q[0].dy, b.q[0].dz};
    const LaunchCtx ctx{st, s->d_spill};
    for (uint64_t i = 0; i < 3 * n; i++) rgb[i] = 0.0f;
    if (stats) stats->camera_samples = n;
    /* One run of the bounce loop per estimate (see BranchOpts); the estimates' radiances are summed. */
    for (int k = 0; k < branches; k++) {
        BranchOpts br;
        br.first_weight = (float)((1.0 / (double)branches) * 0.8);   /* formed in double, narrowed once (integrator.cpp:107) */
        br.stream_off = 256u * (uint32_t)k;
        br.skip_first = k > 0 ? 1 : 0;
        HIP_TRY(hipMemsetAsync(b.cnt, 0, CNT_WORDS * 4, st));
        HIP_TRY(hipMemcpyAsync(b.cnt + CNT_QUEUE, &n32, 4, hipMemcpyHostToDevice, st));
        for (int c = 0; c < 6; c++) HIP_TRY(hipMemcpyAsync(dst[c], src[c], n * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_seed_paths, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n32, sc.d_ids, sample, b);
        HIP_TRY(hipGetLastError());
        rc = enqueue_bounces(s, b, ctx, cp, sc.ln.d_lights, (uint32_t)n_lights, (uint32_t)recursions, seed, br);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(r.data(), b.rad_r, n * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(g.data(), b.rad_g, n * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(bl.data(), b.rad_b, n * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(sc.ln.h_cnt, b.cnt, CNT_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (uint64_t i = 0; i < n; i++) { rgb[3 * i] += r[(size_t)i]; rgb[3 * i + 1] += g[(size_t)i]; rgb[3 * i + 2] += bl[(size_t)i]; }
        if (stats) {
            const uint32_t* h = sc.ln.h_cnt;
            /* the first ray and the first hit's light samples are cast once, by estimate 0 */
            for (int d = k > 0 ? 1 : 0; d <= recursions; d++) {
                stats->closest_rays += h[CNT_QUEUE + d];
                stats->shadow_rays += (uint64_t)h[CNT_QUEUE + d + 1] * (uint64_t)n_lights;
                if (n_lights) stats->shadow_rays_traced += h[CNT_SHADOW(d)];
            }
        }
    }
    return PT_OK;
}

int pt_camera_rays_dev(pt_scene* s, const pt_camera* cam, int width, int height, int x, int y, int w, int h, int jittered,
                       uint32_t sample, uint64_t seed, float* ox, float* oy, float* oz, float* dx, float* dy, float* dz) {
    if (!s || !cam || !ox || !oy || !oz || !dx || !dy || !dz) return fail(PT_ERR_INVALID, "null argument");
    if (width <= 0 || height <= 0 || w <= 0 || h <= 0 || x < 0 || y < 0 || x + w > width || y + h > height)
        return fail(PT_ERR_INVALID, "rectangle outside the image");
    int rc = set_device(s);
    if (rc) return rc;
    CamParams cp;
    fill_cam(cp, cam, width, height);
    RaySoAW out = {ox, oy, oz, dx, dy, dz};
    ScopedTimer tm(s, PT_KERNEL_GENERATE);
    hipLaunchKernelGGL(k_camera_rays, dim3(((unsigned)(w * h) + 255) / 256), dim3(256), 0, s->stream, cp, x, y, w, h, jittered, sample, seed, out);
    HIP_TRY(hipGetLastError());
    return PT_OK;
}

} // extern "C"
