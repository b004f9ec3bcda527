"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")


class Camera(C.Structure):
    _fields_ = [("origin", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3),
                ("image_plane_width", C.c_float), ("plane_of_focus_distance", C.c_float),
                ("dof_dispersion", C.c_float), ("sky_color", C.c_float * 3), ("bounce_mode", C.c_uint32),
                ("shadow_mode", C.c_uint32)]


class Light(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("color", C.c_float * 3)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pt_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        vp, i64, i32, u32, u64 = C.c_void_p, C.c_int64, C.c_int, C.c_uint32, C.c_uint64
        L.orc_scene_load_stl.restype = vp
        L.orc_scene_load_stl.argtypes = [C.c_char_p, i32]
        L.orc_scene_from_triangles.restype = vp
        L.orc_scene_from_triangles.argtypes = [vp, i64, i32]
        L.orc_scene_destroy.argtypes = [vp]
        L.orc_scene_num_triangles.restype = i64
        L.orc_scene_num_triangles.argtypes = [vp]
        L.orc_scene_triangle_data.argtypes = [vp, vp]
        L.orc_tree_stats.argtypes = [vp, vp]
        L.orc_tree_dump.argtypes = [vp, vp, vp]
        L.orc_tree_hash.restype = u64
        L.orc_tree_hash.argtypes = [vp]
        L.orc_raycast.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, vp, i32]
        L.orc_raycast_counts.argtypes = [vp, i64, vp, vp, i32, vp, vp]
        L.orc_raycast_brute.argtypes = [vp, i64, vp, vp, i32, vp, vp, vp, vp, i32]
        L.orc_shadow_visible.argtypes = [vp, i64, vp, vp, vp, vp, i32]
        L.orc_occlusion_visible.argtypes = [vp, i64, vp, vp, vp, vp, i32]
        L.orc_camera_rays.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, u32, u64, vp, vp]
        L.orc_render.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, u32, u32, u64, vp, vp, vp, i32]
        L.orc_render_counting.argtypes = L.orc_render.argtypes
        L.orc_render_capture.restype = i64
        L.orc_render_capture.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, u32, u32, u64, vp, i64]
        L.orc_cast_rays.argtypes = [vp, vp, vp, i32, i64, vp, vp, vp, u32, i32, i32, u64, vp, vp]
        L.orc_tonemap.argtypes = [vp, vp, i64, C.c_float, vp]
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_uniform01.restype = C.c_double
        L.orc_uniform01.argtypes = [u32]
        L.orc_normal_from_word.restype = C.c_float
        L.orc_normal_from_word.argtypes = [u32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_camera(origin, direction, up=(0, 0, 1), image_plane_width=1.0, plane_of_focus_distance=1.0,
                dof_dispersion=0.0, sky_color=(0, 0, 0), bounce_mode=0, shadow_mode=0) -> Camera:
    c = Camera()
    c.origin[:] = [float(np.float32(x)) for x in origin]
    c.direction[:] = [float(np.float32(x)) for x in direction]
    c.up[:] = [float(np.float32(x)) for x in up]
    c.image_plane_width = image_plane_width
    c.plane_of_focus_distance = plane_of_focus_distance
    c.dof_dispersion = dof_dispersion
    c.sky_color[:] = list(sky_color)
    c.bounce_mode = bounce_mode   # 0: mirror (the reference as it stands), 1: Lambertian scatter
    c.shadow_mode = shadow_mode   # 0: the reference's visibility test, 1: occlusion bounded at the light
    return c


def make_lights(lights):
    arr = (Light * max(len(lights), 1))()
    for i, (pos, col) in enumerate(lights):
        arr[i].position[:] = [float(np.float32(x)) for x in pos]
        arr[i].color[:] = [float(np.float32(x)) for x in col]
    return arr


class Scene:
    """Mesh + vertex normals + tree, as the reference's Scene constructor builds them."""

    def __init__(self, tris=None, stl_path=None, threads=1):
        L = lib()
        if stl_path is not None:
            self.h = L.orc_scene_load_stl(stl_path.encode(), threads)
        else:
            t = _f32(tris).reshape(-1, 9)
            self.h = L.orc_scene_from_triangles(_p(t), t.shape[0], threads)
        if not self.h:
            raise RuntimeError("oracle: could not load scene")
        self.n = int(L.orc_scene_num_triangles(self.h))

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def triangle_data(self):
        out = np.zeros((self.n, 30), dtype=np.float32)
        lib().orc_scene_triangle_data(self.h, _p(out))
        return out

    def tree_stats(self):
        st = np.zeros(5, dtype=np.int64)
        lib().orc_tree_stats(self.h, _p(st))
        return dict(nodes=int(st[0]), leaves=int(st[1]), max_depth=int(st[2]), largest_leaf=int(st[3]),
                    leaf_triangles=int(st[4]))

    def tree_dump(self):
        st = self.tree_stats()
        nodes = np.zeros((st["nodes"], 10), dtype=np.float32)
        leaves = np.zeros(max(st["leaf_triangles"], 1), dtype=np.int32)
        lib().orc_tree_dump(self.h, _p(nodes), _p(leaves))
        return nodes, leaves[:st["leaf_triangles"]]

    def tree_hash(self):
        return int(lib().orc_tree_hash(self.h))

    def _cast(self, fn, o, d, normalize, threads):
        o = _f32(o).reshape(-1, 3)
        d = _f32(d).reshape(-1, 3)
        n = o.shape[0]
        idx = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        u = np.empty(n, dtype=np.float32)
        v = np.empty(n, dtype=np.float32)
        fn(self.h, n, _p(o), _p(d), int(normalize), _p(idx), _p(t), _p(u), _p(v), threads)
        return idx, t, u, v

    def raycast(self, o, d, normalize=False, threads=8):
        return self._cast(lib().orc_raycast, o, d, normalize, threads)

    def raycast_brute(self, o, d, normalize=False, threads=8):
        return self._cast(lib().orc_raycast_brute, o, d, normalize, threads)

    def raycast_counts(self, o, d, normalize=False):
        o = _f32(o).reshape(-1, 3)
        d = _f32(d).reshape(-1, 3)
        nb = C.c_uint64(0)
        nt = C.c_uint64(0)
        lib().orc_raycast_counts(self.h, o.shape[0], _p(o), _p(d), int(normalize), C.byref(nb), C.byref(nt))
        return nb.value, nt.value

    def shadow_visible(self, o, d, dist, threads=8, bounded=False):
        """The reference's light-visibility test, or (bounded, an extension) no hit at t < dist."""
        o = _f32(o).reshape(-1, 3)
        d = _f32(d).reshape(-1, 3)
        dist = _f32(dist)
        vis = np.empty(o.shape[0], dtype=np.uint8)
        fn = lib().orc_occlusion_visible if bounded else lib().orc_shadow_visible
        fn(self.h, o.shape[0], _p(o), _p(d), _p(dist), _p(vis), threads)
        return vis

    def render(self, cam, lights, width, height, rect=None, first_sample=0, n_samples=1, seed=1, threads=8,
               rgb=None, passes=None, count_work=False):
        """Returns (rgb_sum, passes, counts).  counts = [closest casts, shadow casts, shadow casts of
        light samples with non-zero weight] or, with count_work, [closest casts, shadow casts, closest
        box tests, closest triangle tests, shadow box tests, shadow triangle tests, shadow casts of
        light samples with non-zero weight]."""
        x0, y0, w, h = rect if rect is not None else (0, 0, -1, -1)
        if rgb is None:
            rgb = np.zeros((height, width, 3), dtype=np.float32)
        if passes is None:
            passes = np.zeros((height, width), dtype=np.int32)
        counts = np.zeros(7 if count_work else 3, dtype=np.uint64)
        larr = make_lights(lights)
        fn = lib().orc_render_counting if count_work else lib().orc_render
        fn(self.h, C.byref(cam), larr, len(lights), width, height, x0, y0, w, h, first_sample,
                         n_samples, seed, _p(rgb), _p(passes), _p(counts), threads)
        return rgb, passes, counts


    def render_capture(self, cam, lights, width, height, rect=None, first_sample=0, n_samples=1, seed=1):
        """Every ray the render casts, in casting order: array [n, 8] of origin, direction, light
        distance, tag (0 closest-hit cast, 1 shadow cast with non-zero weight, 2 with zero weight)."""
        x0, y0, w, h = rect if rect is not None else (0, 0, -1, -1)
        larr = make_lights(lights)
        args = (self.h, C.byref(cam), larr, len(lights), width, height, x0, y0, w, h, first_sample, n_samples, seed)
        n = int(lib().orc_render_capture(*args, None, 0))
        out = np.empty((n, 8), dtype=np.float32)
        lib().orc_render_capture(*args, _p(out), n)
        return out

    def cast_rays(self, cam, lights, o, d, ids=None, sample=0, recursions=10, seed=1, branches=1):
        """Integrator::cast_ray for given rays: (rgb [n,3], [closest casts, shadow casts, weighted shadow casts])."""
        o = _f32(o).reshape(-1, 3)
        d = _f32(d).reshape(-1, 3)
        rgb = np.zeros((o.shape[0], 3), dtype=np.float32)
        counts = np.zeros(3, dtype=np.uint64)
        idarr = None if ids is None else np.ascontiguousarray(ids, dtype=np.uint32)
        larr = make_lights(lights)
        lib().orc_cast_rays(self.h, C.byref(cam), larr, len(lights), o.shape[0], _p(o), _p(d),
                            _p(idarr) if idarr is not None else None, sample, recursions, branches, seed, _p(rgb), _p(counts))
        return rgb, counts


def camera_rays(cam, width, height, rect=None, jittered=False, sample=0, seed=1):
    x0, y0, w, h = rect if rect is not None else (0, 0, width, height)
    o = np.empty((h * w, 3), dtype=np.float32)
    d = np.empty((h * w, 3), dtype=np.float32)
    lib().orc_camera_rays(C.byref(cam), width, height, x0, y0, w, h, int(jittered), sample, seed, _p(o), _p(d))
    return o, d


def tonemap(rgb, passes, gain=255.0):
    rgb = _f32(rgb).reshape(-1, 3)
    passes = np.ascontiguousarray(passes, dtype=np.int32).reshape(-1)
    out = np.empty((rgb.shape[0], 3), dtype=np.uint8)
    lib().orc_tonemap(_p(rgb), _p(passes), rgb.shape[0], gain, _p(out))
    return out


def philox(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(c), _p(k), _p(out))
    return out


def cli_camera(angle=1.0, altitude=0.2, distance=5.0, z_offset=0.0, dof_aperture=0.0, dof_distance=1.0, bounce_mode=0,
               shadow_mode=0):
    """Camera exactly as the reference's cli_render sets it up (cli_render.cpp:92-100)."""
    f32 = np.float32
    a = f32(angle)
    # -distance * Vec(cos(angle), sin(angle), 0.0): the double scalar is narrowed to float.
    base = np.array([f32(np.cos(np.float64(a))), f32(np.sin(np.float64(a))), f32(0.0)], dtype=np.float32)
    origin = f32(-distance) * base
    direction = -origin
    direction[2] = f32(np.float64(direction[2]) + z_offset)
    z = direction[0] * direction[0] + (direction[1] * direction[1] + direction[2] * direction[2])
    direction = direction / np.sqrt(f32(z), dtype=np.float32)
    origin = origin + np.array([0.0, 0.0, altitude], dtype=np.float32)
    return make_camera(origin, direction, (0, 0, 1), image_plane_width=0.75, plane_of_focus_distance=dof_distance,
                       dof_dispersion=dof_aperture, bounce_mode=bounce_mode, shadow_mode=shadow_mode)


def cli_lights():
    """The three lights of cli_render.cpp:89-91."""
    f32 = np.float32
    def col(c):
        return [float(f32(9.0) * f32(x)) for x in c]
    return [((0, 0, 3), col((0.8, 0.5, 0.25))), ((-2, 2, 4), col((0.25, 0.8, 0.25))),
            ((-2, -2, 4), col((0.25, 0.25, 0.8)))]
