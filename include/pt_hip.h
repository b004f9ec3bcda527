/*
 * pt_hip.h -- C ABI of the MI355X ray-casting library (libpt_hip.so).
 *
 * This is the boundary a host program binds instead of the reference's CPU ray caster.
 * The reference has no FFI of its own; each entry point below names the C++ call it
 * stands in for (paths relative to the reference tree).
 *
 * Conventions: every function returns 0 on success or a negative PT_ERR_* code, never
 * throws and never aborts; pt_last_error() returns a thread-local message for the last
 * failure.  Host arrays passed in are copied before the call returns; the caller keeps
 * ownership.  Handles are not internally locked: one host thread drives one handle at a
 * time.  Pointers named d_* are device pointers on the handle's device.
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_OK               0
#define PT_ERR_INVALID     -1   /* bad argument (null pointer, size out of range, malformed tree) */
#define PT_ERR_DEVICE      -2   /* HIP runtime error; message has the HIP error string          */
#define PT_ERR_NO_DEVICE   -3   /* no usable GPU                                                */
#define PT_ERR_LIMIT       -4   /* scene or request exceeds a documented limit                  */

/* ---- flattened scene (host side produces it; see pt_host.h) ------------------------- */

/* Child reference: bit 31 set = leaf, bit 30 = 0 (reserved for the traversal stack),
 * bits 29..26 = triangle count (1..15, or 0 when the count is stored in the first triangle's
 * `leaf_count`), bits 25..0 = index of the leaf's first triangle.  Bit 31 clear = index of an
 * interior node.  0xffffffff is never a valid reference. */
#define PT_REF_LEAF        0x80000000u
#define PT_REF_COUNT_SHIFT 26
#define PT_REF_COUNT_MASK  0xfu
#define PT_REF_START_MASK  0x03ffffffu
#define PT_MAX_TRIANGLES   ((1u << 26) - 64u)
#define PT_MAX_TREE_DEPTH  20      /* interior levels above any leaf; reference caps at 19 (kdtree.cpp:13) */
#define PT_MAX_LIGHTS      16
#define PT_MAX_RAYS_PER_CALL 0x7fffffffull  /* rays of one ray-cast call: a lane holds its ray's index as a signed
                                               32-bit word whose sign means "idle" */

/* One interior node of the reference's tree (kdTreeNode, kdtree.h:19-36), holding the
 * AABBs of BOTH children so a visit needs one 64-byte load.  The node's own box lives in
 * its parent (the root's in pt_flat_scene).  A box is stored per axis as {minimum, maximum},
 * the order in which the slab test consumes it. */
typedef struct pt_flat_node {
    float    lo[3][2];               /* low_side->aabb:  lo[axis] = {minima[axis], maxima[axis]} */
    float    hi[3][2];               /* high_side->aabb: same layout */
    uint32_t lo_ref, hi_ref;
    uint32_t axis;                   /* split_axis       */
    uint32_t pad;
} pt_flat_node;                      /* 64 bytes */

/* Intersection record, leaf order: what Triangle::ray_test reads (utils.cpp:138-163). */
typedef struct pt_tri_isect {
    float    p0[3], e1[3], e2[3];    /* points[0], edge01, edge02 */
    uint32_t mesh_index;             /* index of this triangle in the input mesh */
    uint32_t leaf_count;             /* triangles in this leaf; valid in a leaf's first record */
    uint32_t leaf_remaining;         /* triangles that follow this one in its leaf (0 = last)  */
} pt_tri_isect;                      /* 48 bytes */

/* Shading record, same order: what Integrator::cast_ray reads after a hit
 * (integrator.cpp:66-74, utils.cpp:165-170). */
typedef struct pt_tri_shade {
    float normal[3];  float plane;   /* Triangle::normal; mean of normal.dot(points[i]) */
    float base_n[3];  float d1_x;    /* base_normal; d1 = edge01.normalized() (integrator.cpp:82), */
    float u_n[3];     float d1_y;    /* u_normal;    one component per record slot              */
    float v_n[3];     float d1_z;    /* v_normal                                                */
} pt_tri_shade;                      /* 64 bytes */

typedef struct pt_flat_scene {
    uint64_t            n_nodes;     /* interior nodes                                  */
    uint64_t            n_tris;
    const pt_flat_node* nodes;
    const pt_tri_isect* tris;
    const pt_tri_shade* shade;
    uint32_t            root_ref;    /* leaf ref when the whole mesh is one leaf        */
    uint32_t            max_depth;   /* interior levels on the longest root-to-leaf path */
    float               root_min[3], root_max[3];
} pt_flat_scene;

/* ---- scene --------------------------------------------------------------------------- */

typedef struct pt_scene pt_scene;

int  pt_device_count(int* n);
/* Brings up the runtime's context on a device ahead of the first scene on it (a few tenths of a
 * second on a cold process), so that a host program can do it on a spare thread while it still
 * reads and builds its scene.  Optional: pt_scene_create does the same implicitly. */
int  pt_device_prepare(int device);
/* Uploads the flattened tree; immutable afterwards.  Replaces the device-less
 * `new kdTree(mesh)` result that Scene holds (integrator.cpp:30).  The arrays are checked
 * (child references in range, depth <= PT_MAX_TREE_DEPTH, leaf ranges inside the triangle
 * array, leaf_remaining counting down to each leaf's end) so that a malformed tree is rejected
 * here instead of faulting a kernel. */
int  pt_scene_create(int device, const pt_flat_scene* flat, pt_scene** out);
void pt_scene_destroy(pt_scene* s);

/* ---- ray casting --------------------------------------------------------------------- */

/* Closest hit for n rays: kdTree::ray_test (kdtree.cpp:577-580) applied to each ray.
 * Directions are used as given (the reference normalises in Ray's constructor, not here).
 * tri_index = mesh triangle index or -1; t,u,v are 0 on a miss.  Visit order, tie-breaking
 * and arithmetic follow kdTreeNode::ray_test (kdtree.cpp:271-396) exactly.
 * Host-pointer form: arrays of n floats each, copied to and from the device.  Calls with
 * n <= 64 (one ray under the mouse, kdTree::ray_test) go through a pinned block that the kernel
 * reads and writes directly, on a stream of their own behind a lock inside the handle: they
 * allocate nothing after the first one and may be made while another thread renders or casts with
 * the same scene (the one exception to the one-thread-per-handle rule). */
int pt_raycast_closest(pt_scene* s, uint64_t n,
                       const float* ox, const float* oy, const float* oz,
                       const float* dx, const float* dy, const float* dz,
                       int32_t* tri_index, float* t, float* u, float* v);
/* Device-pointer form (inputs and outputs already resident in HBM); asynchronous on the
 * scene's stream, ordered after earlier calls on the same scene; pt_scene_sync() waits.
 * The scene's stream is private to the handle (created non-blocking) and is NOT ordered after any
 * stream of the caller: whatever produces the inputs must have finished (a device or stream
 * synchronisation, or an event the caller has waited for) before a *_dev entry point is called,
 * and the outputs may be consumed on other streams only after pt_scene_sync().  The same holds
 * for a film created on a caller's buffer (pt_film_create with d_external). */
int pt_raycast_closest_dev(pt_scene* s, uint64_t n,
                           const float* d_ox, const float* d_oy, const float* d_oz,
                           const float* d_dx, const float* d_dy, const float* d_dz,
                           int32_t* d_tri_index, float* d_t, float* d_u, float* d_v);

/* Light visibility as the integrator evaluates it (integrator.cpp:124-129): visible = 1 iff
 * the unbounded ray hits nothing or the barycentric v of its closest hit exceeds dist
 * (the reference passes its out-parameters in the order u,v,t). */
int pt_raycast_shadow(pt_scene* s, uint64_t n,
                      const float* ox, const float* oy, const float* oz,
                      const float* dx, const float* dy, const float* dz,
                      const float* dist, uint8_t* visible);
int pt_raycast_shadow_dev(pt_scene* s, uint64_t n,
                          const float* d_ox, const float* d_oy, const float* d_oz,
                          const float* d_dx, const float* d_dy, const float* d_dz,
                          const float* d_dist, uint8_t* d_visible);

/* Occlusion bounded at the light: visible = 1 iff no triangle is hit at a ray parameter
 * t < max_t (strictly; t counted in units of the given direction's length).  This is the test
 * the author's note at kdtree.cpp:356-357 asks for; the reference itself never bounds a shadow
 * ray, so this entry point has no counterpart there.  Any-hit: the walk stops at the first
 * triangle inside the bound. */
int pt_raycast_occluded(pt_scene* s, uint64_t n,
                        const float* ox, const float* oy, const float* oz,
                        const float* dx, const float* dy, const float* dz,
                        const float* max_t, uint8_t* visible);
int pt_raycast_occluded_dev(pt_scene* s, uint64_t n,
                            const float* d_ox, const float* d_oy, const float* d_oz,
                            const float* d_dx, const float* d_dy, const float* d_dz,
                            const float* d_max_t, uint8_t* d_visible);

/* pt_raycast_closest_dev for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same
 * arguments, same results bit for bit, asynchronous on the scene's stream; it calls no
 * hipMalloc/hipFree and no allocating library routine.  The scratch the trace kernel needs (ray
 * count, queue heads, stack spill area) lives in a static device array of 44 MiB + 1 KiB that each
 * GPU the library runs on holds until the process ends, which bounds the launch to 1024 workgroups
 * and the call to PT_NOALLOC_MAX_RAYS rays (more: PT_ERR_LIMIT).  All calls on one GPU -- from any
 * scene handle -- share that GPU's array: calls that overlap in time on one GPU race; keep them
 * on one stream or wait (pt_scene_sync) between them. */
#define PT_NOALLOC_MAX_RAYS (1ull << 24)
int pt_raycast_closest_dev_noalloc(pt_scene* s, uint64_t n,
                                   const float* d_ox, const float* d_oy, const float* d_oz,
                                   const float* d_dx, const float* d_dy, const float* d_dz,
                                   int32_t* d_tri_index, float* d_t, float* d_u, float* d_v);

int pt_scene_sync(pt_scene* s);

/* Milliseconds the GPU spent in this scene's kernels of one class since the last call with
 * reset != 0, measured with HIP events on the scene's stream, and the number of launches.
 * Classes: 0 closest-hit, 1 shadow (the any-hit walk that answers light samples at distance >= 1,
 * or the bounded test), 2 generate, 3 shade, 4 accumulate, 5 the closest-hit pass for the shadow
 * rays of nearer lights (one launch per depth, nearly always over an empty queue). */
#define PT_KERNEL_CLOSEST    0
#define PT_KERNEL_SHADOW     1
#define PT_KERNEL_GENERATE   2
#define PT_KERNEL_SHADE      3
#define PT_KERNEL_ACCUMULATE 4
#define PT_KERNEL_SHADOW_EXACT 5
#define PT_KERNEL_CLASSES    6
int pt_scene_kernel_time(pt_scene* s, int kernel_class, int reset, double* ms, uint64_t* launches);

/* Which trace kernels the scene's geometry selected (no counterpart in the reference; for
 * measurement tools): out[0] triangles of a leaf a lane tests per round (1, 2, 3; 0 when leaves are
 * tested by the whole wave from 36-byte component-wise records instead), out[1] 1 if the any-hit
 * walks read quantised 32-byte interior nodes, out[2] 1 if the any-hit walk defers its triangle
 * tests to dense passes, out[3] reserved (0). */
int pt_scene_trace_variant(const pt_scene* s, int32_t out[4]);

/* ---- film and rendering -------------------------------------------------------------- */

/* The fields of Scene that Integrator::perform_pass reads at every pass
 * (integrator.h:24-29); callers may change them between calls (cli_render.cpp:92-100). */
typedef struct pt_camera {
    float origin[3];
    float direction[3];
    float up[3];
    float image_plane_width;
    float plane_of_focus_distance;
    float dof_dispersion;
    float sky_color[3];
    uint32_t bounce_mode;            /* PT_BOUNCE_MIRROR or PT_BOUNCE_LAMBERT */
    uint32_t shadow_mode;            /* PT_SHADOW_REFERENCE or PT_SHADOW_BOUNDED */
} pt_camera;

/* What the bounce ray of a path vertex is (integrator.cpp:76-108).
 * PT_BOUNCE_MIRROR: the reflected direction -- the reference as it stands (:104 overwrites the
 * scattered direction).  This is the mode every parity claim is about.
 * PT_BOUNCE_LAMBERT: the direction :79-84 compute before :104 replaces it (a unit-sphere sample
 * folded onto the triangle's side, in the frame {normal, edge01, normal x edge01}); the
 * reference's "diffuse" look, kept as a flagged extension. */
#define PT_BOUNCE_MIRROR  0u
#define PT_BOUNCE_LAMBERT 1u

/* When a light sample counts as visible (integrator.cpp:124-129).
 * PT_SHADOW_REFERENCE: the reference as it stands -- the unbounded shadow ray hits nothing, or
 * the barycentric v of its closest hit exceeds the distance to the light (the swapped
 * out-parameters of :127).  The mode every parity claim is about.
 * PT_SHADOW_BOUNDED: no triangle between the surface point and the (jittered) light position,
 * i.e. no hit with t < distance; a flagged extension (kdtree.cpp:356-357 notes the missing
 * bound).  Geometry behind a light then no longer shadows it. */
#define PT_SHADOW_REFERENCE 0u
#define PT_SHADOW_BOUNDED   1u

typedef struct pt_light {        /* Light, integrator.h:15-18 */
    float position[3];
    float color[3];
} pt_light;

/* n_samples passes of PassDescriptor(x, y, w, h) (integrator.h:35-43), using sample indices
 * first_sample .. first_sample + n_samples - 1 of the random stream. */
typedef struct pt_tile_pass {
    int32_t  x, y, w, h;         /* w or h = -1: to the edge, as PassDescriptor::clamp_bounds */
    uint32_t first_sample;
    uint32_t n_samples;
} pt_tile_pass;

typedef struct pt_render_stats {
    uint64_t camera_samples;
    uint64_t closest_rays;       /* what `rays_cast++` (kdtree.cpp:578) would count from :57  */
    uint64_t shadow_rays;        /* ... and from integrator.cpp:127                            */
    double   gpu_ms;             /* GPU time of this call, HIP events on the scene's stream    */
    uint64_t shadow_rays_traced; /* of shadow_rays, those actually traced: a light sample whose
                                    weight (integrator.cpp:131-145) is exactly zero adds nothing
                                    whether the light is visible or not, so its ray is counted
                                    (the reference casts it) but not traced                     */
} pt_render_stats;

typedef struct pt_film pt_film;

/* Device-resident Canvas (canvas.h:8-27): per pixel the RGB energy sum and the pass count,
 * stored as 4 floats {r, g, b, passes} so one sum-reduce combines films of several GPUs
 * (Canvas::add_from, canvas.cpp:47-52).  If d_external is not NULL the film uses that
 * device buffer (width*height*4 floats, owned by the caller) instead of allocating. */
int  pt_film_create(pt_scene* s, int width, int height, float* d_external, pt_film** out);
void pt_film_destroy(pt_film* f);
int  pt_film_zero(pt_film* f);                         /* Canvas::zero, canvas.cpp:19-25 */
float* pt_film_device_ptr(pt_film* f);
/* Copies out Canvas::pixels (W*H*3) and Canvas::per_pixel_passes (W*H); waits for rendering. */
int  pt_film_read(pt_film* f, float* rgb_sum, int32_t* per_pixel_passes);
/* Sums n films of the same size, one per GPU (films[0] may be the only one), and copies out the
 * total as pt_film_read does: RenderEngine::rebuild_master_canvas (integrator.cpp:469-487),
 * i.e. Canvas::add_from over all workers (canvas.cpp:47-52).  For n > 1 the sum is ONE RCCL
 * sum-reduce of W*H*4 floats over xGMI to films[0]'s GPU (communicators for the device set are
 * created on first use and kept); the films themselves are left unchanged.  RCCL is loaded on
 * first use, so programs that never call this do not depend on it.  All films must belong to
 * different GPUs; the calling thread must be the only one using them during the call. */
int  pt_films_reduce(pt_film* const* films, int n, float* rgb_sum, int32_t* per_pixel_passes);
/* Adds host accumulators of the same size into the film (Canvas::add_from with a host Canvas,
 * canvas.cpp:47-52): how a render resumes from accumulators saved earlier. */
int  pt_film_add(pt_film* f, const float* rgb_sum, const int32_t* per_pixel_passes);

/* Integrator::perform_pass (integrator.cpp:211-275) for every listed tile pass, as a
 * wavefront path tracer.  The call returns when the passes have been rendered into the film
 * (it waits for its last batch's counters); stats, if not NULL, receives the ray counts and the
 * GPU time.  Pixels touched by several entries are accumulated in list order. */
int pt_render_passes(pt_film* f, const pt_camera* cam, const pt_light* lights, int n_lights,
                     const pt_tile_pass* passes, int n_passes, uint64_t seed,
                     pt_render_stats* stats);

/* Integrator::cast_ray(ray, recursions, branches) (integrator.cpp:51-156) for n rays of the caller's
 * own: the radiance along each, rgb[3*i..3*i+2].  Host arrays; directions are used as given (the
 * reference's Ray constructor normalises, this call does not).  Of the camera only sky_color,
 * bounce_mode and shadow_mode are read.  Ray i draws its light samples from the random stream
 * keyed by stream_ids[i] (NULL: i) where a rendered path has its pixel number, and by `sample`;
 * recursions is 0..10 (perform_pass uses 10).  branches (perform_pass uses 1; at most
 * PT_MAX_BRANCHES): the bounce term of the first hit is the mean of that many sub-paths, each
 * weighted (1.0/branches)*0.8 (:77, :107) and drawing from random streams of its own; 0 or less:
 * no bounce term, as the reference's loop then does not run.  Waits for the result; stats may be
 * NULL. */
#define PT_MAX_BRANCHES 16
int pt_cast_rays(pt_scene* s, const pt_camera* cam, const pt_light* lights, int n_lights, uint64_t n,
                 const float* ox, const float* oy, const float* oz,
                 const float* dx, const float* dy, const float* dz,
                 const uint32_t* stream_ids, uint32_t sample, int recursions, int branches, uint64_t seed,
                 float* rgb, pt_render_stats* stats);

/* The jittered camera rays of sample `sample` for the rectangle, row-major, as SoA device
 * arrays of w*h floats: what integrator.cpp:248-259 builds before calling cast_ray.
 * jittered = 0 gives Integrator::get_ray_for_pixel (integrator.cpp:197-209) instead. */
int pt_camera_rays_dev(pt_scene* s, const pt_camera* cam, int width, int height,
                       int x, int y, int w, int h, int jittered, uint32_t sample, uint64_t seed,
                       float* d_ox, float* d_oy, float* d_oz, float* d_dx, float* d_dy, float* d_dz);

const char* pt_last_error(void);
/* Version string of the library and the GPU architecture it was compiled for. */
const char* pt_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
