/*
 * pt_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  This library restates, on the CPU and in
 * strict IEEE arithmetic, what the reference path tracer computes on its ray-casting
 * path.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
 * load it, and only as the checker.  The product (path_tracing_amd) never links it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this
 * path, and it cannot be compiled in this image (its only third-party dependency,
 * Eigen 3, is absent).  The oracle is therefore pinned by nothing but a careful
 * reading of the reference sources; every function cites the lines it follows.
 *
 * All functions are plain C, take/return POD, and never throw.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* Camera / scene parameters the integrator reads at every pass
 * (reference: Scene fields, integrator.h:20-33). */
typedef struct orc_camera {
    float origin[3];     /* main_camera.origin                    */
    float direction[3];  /* main_camera.direction (used as given) */
    float up[3];         /* scene_up                              */
    float image_plane_width;
    float plane_of_focus_distance;
    float dof_dispersion;
    float sky_color[3];
    uint32_t bounce_mode; /* 0: mirror bounce (the reference as it stands, integrator.cpp:104);
                             1: the Lambertian scatter of :79-84 without :104 (flagged extension) */
    uint32_t shadow_mode; /* 0: the reference's visibility test (integrator.cpp:127-129);
                             1: a light is hidden only by a hit nearer than it (flagged extension) */
} orc_camera;

typedef struct orc_light {
    float position[3];
    float color[3];
} orc_light;

/* Scene = mesh + vertex normals + tree (reference Scene::Scene, integrator.cpp:11-39). */
orc_scene* orc_scene_load_stl(const char* path, int build_threads);
/* n triangles, 9 floats each (p0 p1 p2), same pipeline as the STL loader after the read. */
orc_scene* orc_scene_from_triangles(const float* verts, int64_t n, int build_threads);
void       orc_scene_destroy(orc_scene* s);
int64_t    orc_scene_num_triangles(const orc_scene* s);

/* Per-triangle data as the reference's Triangle holds it (utils.h:94-113), 30 floats per
 * triangle: p0 p1 p2 edge01 edge02 normal base_normal u_normal v_normal, then the plane
 * parameter of project_point_to_given_altitude and two zero pads. */
void orc_scene_triangle_data(const orc_scene* s, float* out30);

/* Tree statistics and a pre-order structural dump.
 * stats[0]=nodes, [1]=leaves, [2]=max depth, [3]=largest leaf, [4]=sum of leaf sizes. */
void    orc_tree_stats(const orc_scene* s, int64_t stats[5]);
/* Pre-order: per node one record of 10 floats {is_leaf, axis, split, min[3], max[3], count}
 * into nodes10 (size = stats[0]*10) and the leaves' mesh indices, in leaf order, appended
 * to leaf_indices (size = stats[4]).  axis/split are 0 for leaves. */
void    orc_tree_dump(const orc_scene* s, float* nodes10, int32_t* leaf_indices);
/* FNV-1a 64-bit hash over the same dump, for meshes too large to ship as a fixture. */
uint64_t orc_tree_hash(const orc_scene* s);

/* Closest hit (reference kdTree::ray_test, kdtree.cpp:577-580).  Directions are used as
 * given unless normalize != 0, in which case they go through the Ray constructor's
 * normalisation first (utils.cpp:20-23).  idx = mesh triangle index or -1; t/u/v are
 * left at 0 on a miss. */
void orc_raycast(const orc_scene* s, int64_t n, const float* origins, const float* dirs,
                 int normalize, int32_t* idx, float* t, float* u, float* v, int threads);
/* Same traversal with counters: box tests and triangle tests per ray (reference visit order). */
void orc_raycast_counts(const orc_scene* s, int64_t n, const float* origins, const float* dirs,
                        int normalize, uint64_t* box_tests, uint64_t* tri_tests);
/* Exhaustive closest hit over all triangles in mesh order (no tree), strict '<'. */
void orc_raycast_brute(const orc_scene* s, int64_t n, const float* origins, const float* dirs,
                       int normalize, int32_t* idx, float* t, float* u, float* v, int threads);
/* The integrator's visibility test for a light sample, including the reference's swapped
 * out-parameters (integrator.cpp:127-129): visible[i] = 1 iff the unbounded shadow ray hits
 * nothing, or the barycentric v of its closest hit exceeds dist[i]. */
void orc_shadow_visible(const orc_scene* s, int64_t n, const float* origins, const float* dirs,
                        const float* dist, uint8_t* visible, int threads);

/* Extension, no counterpart in the reference: visible[i] = 1 iff no triangle, in any node whose
 * box the ray hits, is hit at t < max_t[i].  What camera.shadow_mode = 1 uses for its lights. */
void orc_occlusion_visible(const orc_scene* s, int64_t n, const float* origins, const float* dirs,
                           const float* max_t, uint8_t* visible, int threads);

/* Camera rays.  jittered=0: Integrator::get_ray_for_pixel (integrator.cpp:197-209).
 * jittered=1: the per-pixel ray of Integrator::perform_pass (integrator.cpp:248-259) with
 * the counter-based random stream of sample index `sample`.  Rays for the w*h pixels of the
 * rectangle, row-major, 3 floats each. */
void orc_camera_rays(const orc_camera* cam, int width, int height, int x0, int y0, int w, int h,
                     int jittered, uint32_t sample, uint64_t seed, float* origins, float* dirs);

/* Integrator::perform_pass over a rectangle for samples [first_sample, first_sample+n_samples)
 * (integrator.cpp:211-275), accumulating into rgb_sum (W*H*3) and passes (W*H), which the
 * caller zeroes.  ray_counts (3 words, may be NULL): [0] += closest-hit casts, [1] += shadow casts,
 * [2] += the shadow casts whose light sample has a non-zero weight (bookkeeping for the GPU
 * renderer, which counts the others but does not trace them). */
void orc_render(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights,
                int width, int height, int x0, int y0, int w, int h,
                uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                float* rgb_sum, int32_t* passes, uint64_t* ray_counts, int threads);

/* Same render, additionally counting the work of the reference's tree walk: work_counts[0..6] +=
 * closest casts, shadow casts, box tests and triangle tests of the closest casts, box tests
 * and triangle tests of the shadow casts (reference visit order, no extra pruning), and the
 * shadow casts whose light sample has a non-zero weight (the ones the GPU renderer traces). */
void orc_render_counting(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights,
                         int width, int height, int x0, int y0, int w, int h,
                         uint32_t first_sample, uint32_t n_samples, uint64_t seed,
                         float* rgb_sum, int32_t* passes, uint64_t* work_counts, int threads);

/* The same render on one thread, handing back every ray it casts, in casting order: 8 floats per
 * ray -- origin, direction, light distance (0 for a closest-hit cast), tag (0 closest-hit cast,
 * 1 shadow cast of a light sample with non-zero weight, 2 shadow cast of a zero-weight sample).
 * Returns the number of rays cast; rays beyond `cap` are counted but not stored. */
int64_t orc_render_capture(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights,
                           int width, int height, int x0, int y0, int w, int h,
                           uint32_t first_sample, uint32_t n_samples, uint64_t seed, float* rays8, int64_t cap);

/* Integrator::cast_ray(ray, recursions, branches) (integrator.cpp:51-156) for n given rays (origins
 * and dirs are n*3 floats, directions used as given): rgb = n*3 floats.  ids (NULL: the ray's number)
 * and sample key the random stream where a rendered path has its pixel and sample numbers; sub-path
 * k of a first hit with several branches adds 256*k to its stream numbers.
 * counts = {closest casts, shadow casts, shadow casts of light samples with non-zero weight}. */
void orc_cast_rays(const orc_scene* s, const orc_camera* cam, const orc_light* lights, int n_lights, int64_t n,
                   const float* origins, const float* dirs, const uint32_t* ids, uint32_t sample, int recursions,
                   int branches, uint64_t seed, float* rgb, uint64_t* counts);

/* Canvas::get_pixel for a whole image (canvas.cpp:39-45): rgb8 = W*H*3 bytes. */
void orc_tonemap(const float* rgb_sum, const int32_t* passes, int64_t n_pixels, float gain,
                 uint8_t* rgb8);

/* Random stream pieces, exposed so the device implementation can be checked piecewise. */
void   orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_uniform01(uint32_t word);
float  orc_normal_from_word(uint32_t word);

#ifdef __cplusplus
}
#endif
#endif
