/*
 * pt_host.h -- C interface of the host-side scene preparation (libpt_host.so).
 *
 * Everything the GPU path needs from the CPU before the first ray: reading the mesh,
 * per-vertex normals, the reference's tree, and its flattening into the arrays
 * pt_scene_create() uploads (pt_hip.h).  No GPU code in this library.
 */
#ifndef PT_HOST_H
#define PT_HOST_H

#include <stdint.h>
#include "pt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pth_scene pth_scene;

/* read_stl + compute_barycentric_normals + kdTree build (reference stlreader.cpp:10-43,
 * :62-81, kdtree.cpp:466-569), then flattening.  threads <= 0: one per hardware thread.
 * Returns NULL on an unreadable or malformed file (the reference prints "Couldn't read
 * input file." and carries on with an unusable Scene, integrator.cpp:23-26). */
pth_scene* pth_scene_from_stl(const char* path, int threads);
/* The same, and `mesh_ready(scene, ctx)` is called once, on a thread of the library's, as soon as
 * the per-triangle data (pth_triangle_data, pth_num_triangles) is final -- while the tree is still
 * being built -- so that a caller can fill its own mesh structures beside the build.  The call has
 * returned before pth_scene_from_stl_with does.  No other function may be used on the scene from
 * inside the callback. */
typedef void (*pth_mesh_ready_fn)(const pth_scene* scene, void* ctx);
pth_scene* pth_scene_from_stl_with(const char* path, int threads, pth_mesh_ready_fn mesh_ready, void* ctx);
/* Same from n triangles of 9 floats (p0 p1 p2). */
pth_scene* pth_scene_from_triangles(const float* verts, int64_t n, int threads);
void       pth_scene_destroy(pth_scene* s);

int64_t              pth_num_triangles(const pth_scene* s);
const pt_flat_scene* pth_flat_scene(const pth_scene* s);
/* Seconds spent in the vertex-normal pass, the tree build and the flattening.  With more than
 * one thread the first two run side by side, so their sum exceeds the wall time. */
void pth_timings(const pth_scene* s, double out[3]);

/* Per-triangle data in mesh order, 30 floats each: p0 p1 p2 edge01 edge02 normal base_normal
 * u_normal v_normal plane 0 0 (Triangle, reference utils.h:94-113). */
void pth_triangle_data(const pth_scene* s, float* out30);
/* The same for triangles [first, last) only, on the calling thread: out30 receives
 * (last - first) * 30 floats. */
void pth_triangle_data_range(const pth_scene* s, int64_t first, int64_t last, float* out30);

/* 1 if the library's multi-threaded restatement of std::sort leaves a tie-heavy test array exactly
 * as this C++ library's std::sort does (the builder uses it only then, std::sort otherwise), else 0. */
int pth_parallel_sort_matches_std(void);

/* stats[0]=nodes, [1]=leaves, [2]=max depth, [3]=largest leaf, [4]=sum of leaf sizes. */
void pth_tree_stats(const pth_scene* s, int64_t stats[5]);
/* Pre-order dump of the pointer-free build tree: 10 floats per node {is_leaf, axis, split,
 * min[3], max[3], count} and the leaves' mesh indices in leaf order. */
void pth_tree_dump(const pth_scene* s, float* nodes10, int32_t* leaf_indices);

#ifdef __cplusplus
}
#endif
#endif
