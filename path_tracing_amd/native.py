"""ctypes bindings of the two native libraries.

* ``libpt_host.so`` -- CPU scene preparation (mesh, vertex normals, tree, flattening).
* ``libpt_hip.so``  -- the MI355X ray-casting library (C ABI in ``include/pt_hip.h``).

There is no CPU fallback for the GPU entry points: if ``libpt_hip.so`` is missing or no
GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_PKG, "lib")
HOST_LIB = os.path.join(LIB_DIR, "libpt_host.so")
# PT_HIP_LIB selects another build of the same library (kernel tuning experiments).
HIP_LIB = os.environ.get("PT_HIP_LIB") or os.path.join(LIB_DIR, "libpt_hip.so")

PT_REF_LEAF = 0x80000000
PT_REF_COUNT_SHIFT = 26
PT_REF_START_MASK = 0x03FFFFFF


class FlatNode(C.Structure):
    _fields_ = [("lo", C.c_float * 6), ("hi", C.c_float * 6), ("lo_ref", C.c_uint32), ("hi_ref", C.c_uint32),
                ("axis", C.c_uint32), ("pad", C.c_uint32)]


NODE_DTYPE = np.dtype([("lo", "<f4", (3, 2)), ("hi", "<f4", (3, 2)), ("lo_ref", "<u4"), ("hi_ref", "<u4"), ("axis", "<u4"),
                       ("pad", "<u4")])
ISECT_DTYPE = np.dtype([("p0", "<f4", (3,)), ("e1", "<f4", (3,)), ("e2", "<f4", (3,)), ("mesh_index", "<u4"),
                        ("leaf_count", "<u4"), ("leaf_remaining", "<u4")])
SHADE_DTYPE = np.dtype([("normal", "<f4", (3,)), ("plane", "<f4"), ("base_n", "<f4", (3,)), ("d1_x", "<f4"),
                        ("u_n", "<f4", (3,)), ("d1_y", "<f4"), ("v_n", "<f4", (3,)), ("d1_z", "<f4")])
assert NODE_DTYPE.itemsize == 64 and ISECT_DTYPE.itemsize == 48 and SHADE_DTYPE.itemsize == 64


class FlatScene(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_tris", C.c_uint64), ("nodes", C.c_void_p), ("tris", C.c_void_p),
                ("shade", C.c_void_p), ("root_ref", C.c_uint32), ("max_depth", C.c_uint32),
                ("root_min", C.c_float * 3), ("root_max", C.c_float * 3)]


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("image_plane_width", C.c_float), ("plane_of_focus_distance", C.c_float),
                ("dof_dispersion", C.c_float), ("sky_color", C.c_float * 3), ("bounce_mode", C.c_uint32),
                ("shadow_mode", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3)]


class TilePass(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("w", C.c_int32), ("h", C.c_int32),
                ("first_sample", C.c_uint32), ("n_samples", C.c_uint32)]


class RenderStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("closest_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("gpu_ms", C.c_double), ("shadow_rays_traced", C.c_uint64)]


_host = None
_hip = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise RuntimeError(f"{HOST_LIB} is missing: run `make host` (or __graft_entry__.build())")
        L = C.CDLL(HOST_LIB)
        vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
        L.pth_scene_from_stl.restype = vp
        L.pth_scene_from_stl.argtypes = [C.c_char_p, i32]
        L.pth_scene_from_triangles.restype = vp
        L.pth_scene_from_triangles.argtypes = [vp, i64, i32]
        L.pth_scene_destroy.argtypes = [vp]
        L.pth_num_triangles.restype = i64
        L.pth_num_triangles.argtypes = [vp]
        L.pth_flat_scene.restype = C.POINTER(FlatScene)
        L.pth_flat_scene.argtypes = [vp]
        L.pth_timings.argtypes = [vp, vp]
        L.pth_triangle_data.argtypes = [vp, vp]
        L.pth_tree_stats.argtypes = [vp, vp]
        L.pth_triangle_data_range.argtypes = [vp, i64, i64, vp]
        L.pth_parallel_sort_matches_std.restype = i32
        L.pth_parallel_sort_matches_std.argtypes = []
        L.pth_scene_from_stl_with.restype = vp
        L.pth_scene_from_stl_with.argtypes = [C.c_char_p, i32, vp, vp]
        L.pth_tree_dump.argtypes = [vp, vp, vp]
        _host = L
    return _host


HIP_SYMBOLS = [
    "pt_device_count", "pt_device_prepare", "pt_scene_create", "pt_scene_destroy", "pt_raycast_closest", "pt_raycast_closest_dev",
    "pt_raycast_closest_dev_noalloc",
    "pt_raycast_shadow", "pt_raycast_shadow_dev", "pt_raycast_occluded", "pt_raycast_occluded_dev", "pt_scene_sync", "pt_scene_kernel_time", "pt_scene_trace_variant", "pt_film_create",
    "pt_film_destroy", "pt_film_zero", "pt_film_device_ptr", "pt_film_read", "pt_film_add", "pt_films_reduce", "pt_render_passes",
    "pt_camera_rays_dev", "pt_cast_rays", "pt_last_error", "pt_build_info",
]


def hip_lib():
    """Loads libpt_hip.so; raises if it has not been built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError(f"{HIP_LIB} is missing: run `make hip` (or __graft_entry__.build()); "
                               "there is no CPU fallback for the GPU path")
        if os.environ.get("PT_NO_TORCH") != "1":
            # torch bundles its own HIP runtime (same soname).  Two HIP runtimes in one process
            # cannot both see the GPU, so when torch is going to be used for device memory it has
            # to be loaded first; libpt_hip.so then binds to the runtime torch brought in.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(HIP_LIB)
        vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
        L.pt_device_count.argtypes = [C.POINTER(i32)]
        L.pt_device_prepare.argtypes = [i32]
        L.pt_scene_create.argtypes = [i32, C.POINTER(FlatScene), C.POINTER(vp)]
        L.pt_scene_destroy.argtypes = [vp]
        L.pt_scene_destroy.restype = None
        L.pt_raycast_closest.argtypes = [vp, u64] + [vp] * 10
        L.pt_raycast_closest_dev.argtypes = [vp, u64] + [vp] * 10
        L.pt_raycast_closest_dev_noalloc.argtypes = [vp, u64] + [vp] * 10
        L.pt_raycast_shadow.argtypes = [vp, u64] + [vp] * 8
        L.pt_raycast_shadow_dev.argtypes = [vp, u64] + [vp] * 8
        L.pt_raycast_occluded.argtypes = [vp, u64] + [vp] * 8
        L.pt_raycast_occluded_dev.argtypes = [vp, u64] + [vp] * 8
        L.pt_scene_sync.argtypes = [vp]
        L.pt_scene_kernel_time.argtypes = [vp, i32, i32, C.POINTER(C.c_double), C.POINTER(u64)]
        L.pt_scene_trace_variant.argtypes = [vp, vp]
        L.pt_film_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
        L.pt_film_destroy.argtypes = [vp]
        L.pt_film_destroy.restype = None
        L.pt_film_zero.argtypes = [vp]
        L.pt_film_device_ptr.argtypes = [vp]
        L.pt_film_device_ptr.restype = vp
        L.pt_film_read.argtypes = [vp, vp, vp]
        L.pt_film_add.argtypes = [vp, vp, vp]
        L.pt_films_reduce.argtypes = [vp, i32, vp, vp]
        L.pt_render_passes.argtypes = [vp, C.POINTER(Camera), vp, i32, vp, i32, u64, C.POINTER(RenderStats)]
        L.pt_camera_rays_dev.argtypes = [vp, C.POINTER(Camera), i32, i32, i32, i32, i32, i32, i32, u32, u64] + [vp] * 6
        L.pt_cast_rays.argtypes = [vp, C.POINTER(Camera), vp, i32, u64] + [vp] * 7 + [u32, i32, i32, u64, vp, C.POINTER(RenderStats)]
        L.pt_last_error.restype = C.c_char_p
        L.pt_build_info.restype = C.c_char_p
        _hip = L
    return _hip


class PtError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = hip_lib().pt_last_error()
        raise PtError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
