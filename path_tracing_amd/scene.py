"""Python face of the host-side scene preparation and of the GPU handles.

``HostScene`` wraps libpt_host.so (mesh -> vertex normals -> tree -> flat arrays).
``GpuScene`` / ``Film`` wrap the C ABI of libpt_hip.so.  Device memory handed to the
``*_dev`` entry points is owned by torch tensors: torch is the allocator, not the compute.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import native
from .native import Camera, Light, RenderStats, TilePass, check


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


def cli_camera(angle=1.0, altitude=0.2, distance=5.0, z_offset=0.0, dof_aperture=0.0, dof_distance=1.0,
               bounce_mode=0, shadow_mode=0) -> Camera:
    """Camera as the reference's cli_render derives it from its flags (cli_render.cpp:92-100)."""
    f32 = np.float32
    a = f32(angle)
    base = np.array([f32(np.cos(np.float64(a))), f32(np.sin(np.float64(a))), f32(0.0)], dtype=np.float32)
    origin = f32(-distance) * base
    direction = -origin
    direction[2] = f32(np.float64(direction[2]) + z_offset)
    z = direction[0] * direction[0] + (direction[1] * direction[1] + direction[2] * direction[2])
    direction = direction / np.sqrt(f32(z), dtype=np.float32)
    origin = origin + np.array([0.0, 0.0, altitude], dtype=np.float32)
    return make_camera(origin, direction, (0, 0, 1), image_plane_width=0.75,
                       plane_of_focus_distance=dof_distance, dof_dispersion=dof_aperture, bounce_mode=bounce_mode,
                       shadow_mode=shadow_mode)


def cli_lights():
    """The three lights cli_render hard-codes (cli_render.cpp:89-91)."""
    f32 = np.float32

    def col(c):
        return [float(f32(9.0) * f32(x)) for x in c]

    return [((0, 0, 3), col((0.8, 0.5, 0.25))), ((-2, 2, 4), col((0.25, 0.8, 0.25))),
            ((-2, -2, 4), col((0.25, 0.25, 0.8)))]


class HostScene:
    """Mesh + vertex normals + tree + flattened arrays, all on the CPU."""

    def __init__(self, tris=None, stl_path=None, threads=0):
        L = native.host_lib()
        if stl_path is not None:
            self.h = L.pth_scene_from_stl(stl_path.encode(), threads)
        else:
            t = _f32(tris).reshape(-1, 9)
            self.h = L.pth_scene_from_triangles(_p(t), t.shape[0], threads)
        if not self.h:
            raise RuntimeError("Couldn't read input file." if stl_path else "scene build failed")
        self.n = int(L.pth_num_triangles(self.h))

    def close(self):
        if getattr(self, "h", None):
            native.host_lib().pth_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def flat(self) -> native.FlatScene:
        return native.host_lib().pth_flat_scene(self.h).contents

    def flat_arrays(self):
        """Views (copies) of the flattened arrays as numpy structured arrays."""
        f = self.flat

        def view(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros(0, dtype=dt)
            buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt, count=n).copy()

        return (view(f.nodes, f.n_nodes, native.NODE_DTYPE), view(f.tris, f.n_tris, native.ISECT_DTYPE),
                view(f.shade, f.n_tris, native.SHADE_DTYPE))

    def timings(self):
        out = np.zeros(3, dtype=np.float64)
        native.host_lib().pth_timings(self.h, _p(out))
        return dict(normals_s=float(out[0]), build_s=float(out[1]), flatten_s=float(out[2]))

    def triangle_data(self):
        out = np.zeros((self.n, 30), dtype=np.float32)
        native.host_lib().pth_triangle_data(self.h, _p(out))
        return out

    def tree_stats(self):
        st = np.zeros(5, dtype=np.int64)
        native.host_lib().pth_tree_stats(self.h, _p(st))
        return dict(nodes=int(st[0]), leaves=int(st[1]), max_depth=int(st[2]), largest_leaf=int(st[3]),
                    leaf_triangles=int(st[4]))

    def tree_dump(self):
        st = self.tree_stats()
        nodes = np.zeros((st["nodes"], 10), dtype=np.float32)
        leaves = np.zeros(max(st["leaf_triangles"], 1), dtype=np.int32)
        native.host_lib().pth_tree_dump(self.h, _p(nodes), _p(leaves))
        return nodes, leaves[:st["leaf_triangles"]]


class GpuScene:
    """The flattened tree resident in HBM (pt_scene)."""

    def __init__(self, host: HostScene, device: int = 0):
        L = native.hip_lib()
        self.host = host
        self.device = device
        h = C.c_void_p()
        flat = host.flat
        check(L.pt_scene_create(device, C.byref(flat), C.byref(h)), "pt_scene_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            native.hip_lib().pt_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(native.hip_lib().pt_scene_sync(self.h), "pt_scene_sync")

    def kernel_time(self, kernel_class: int, reset: bool = False):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        check(native.hip_lib().pt_scene_kernel_time(self.h, kernel_class, int(reset), C.byref(ms), C.byref(n)),
              "pt_scene_kernel_time")
        return ms.value, n.value

    def trace_variant(self):
        """Which trace kernels this scene's geometry selected (pt_scene_trace_variant)."""
        out = np.zeros(4, dtype=np.int32)
        check(native.hip_lib().pt_scene_trace_variant(self.h, _p(out)), "pt_scene_trace_variant")
        return dict(tris_per_round=int(out[0]), leafwise=int(out[0]) == 0, qnodes=bool(out[1]), dense=bool(out[2]))

    # -- host-pointer entry points (numpy in, numpy out) ---------------------------------
    def raycast_closest(self, origins, dirs):
        o = _f32(origins).reshape(-1, 3)
        d = _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        idx = np.empty(n, dtype=np.int32)
        t = np.empty(n, dtype=np.float32)
        u = np.empty(n, dtype=np.float32)
        v = np.empty(n, dtype=np.float32)
        check(native.hip_lib().pt_raycast_closest(self.h, n, *[_p(c) for c in cols], _p(idx), _p(t), _p(u), _p(v)),
              "pt_raycast_closest")
        return idx, t, u, v

    def raycast_shadow(self, origins, dirs, dist, bounded=False):
        """Visibility per ray: the reference's test, or (bounded) no hit nearer than dist."""
        o = _f32(origins).reshape(-1, 3)
        d = _f32(dirs).reshape(-1, 3)
        dist = _f32(dist).reshape(-1)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        vis = np.empty(n, dtype=np.uint8)
        L = native.hip_lib()
        name = "pt_raycast_occluded" if bounded else "pt_raycast_shadow"
        check(getattr(L, name)(self.h, n, *[_p(c) for c in cols], _p(dist), _p(vis)), name)
        return vis

    # -- device-pointer entry points (torch tensors on this device) ----------------------
    def raycast_closest_dev(self, rays, out_idx, out_t, out_u, out_v, noalloc=False):
        """rays: six float32 CUDA tensors (ox, oy, oz, dx, dy, dz) of equal length.
        noalloc selects the entry point whose scratch is a static device array."""
        n = rays[0].numel()
        ptrs = [C.c_void_p(r.data_ptr()) for r in rays]
        fn = native.hip_lib().pt_raycast_closest_dev_noalloc if noalloc else native.hip_lib().pt_raycast_closest_dev
        check(fn(
            self.h, n, *ptrs, C.c_void_p(out_idx.data_ptr()), C.c_void_p(out_t.data_ptr()),
            C.c_void_p(out_u.data_ptr()), C.c_void_p(out_v.data_ptr())), "pt_raycast_closest_dev")

    def raycast_shadow_dev(self, rays, dist, out_vis, bounded=False):
        n = rays[0].numel()
        ptrs = [C.c_void_p(r.data_ptr()) for r in rays]
        name = "pt_raycast_occluded_dev" if bounded else "pt_raycast_shadow_dev"
        check(getattr(native.hip_lib(), name)(
            self.h, n, *ptrs, C.c_void_p(dist.data_ptr()), C.c_void_p(out_vis.data_ptr())), name)

    def cast_rays(self, cam: Camera, lights, origins, dirs, stream_ids=None, sample=0, recursions=10, seed=1, branches=1):
        """Radiance along each given ray (Integrator::cast_ray); returns (rgb [n,3], RenderStats)."""
        o = _f32(origins).reshape(-1, 3)
        d = _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, k]) for k in range(3)] + [np.ascontiguousarray(d[:, k]) for k in range(3)]
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.uint32)
        rgb = np.zeros((n, 3), dtype=np.float32)
        st = RenderStats()
        larr = make_lights(lights)
        check(native.hip_lib().pt_cast_rays(self.h, C.byref(cam), larr, len(lights), n, *[_p(c) for c in cols],
                                            _p(ids) if ids is not None else None, sample, recursions, branches, seed, _p(rgb), C.byref(st)),
              "pt_cast_rays")
        return rgb, st

    def camera_rays_dev(self, cam: Camera, width, height, rect, jittered, sample, seed, out6):
        x, y, w, h = rect
        ptrs = [C.c_void_p(t.data_ptr()) for t in out6]
        check(native.hip_lib().pt_camera_rays_dev(self.h, C.byref(cam), width, height, x, y, w, h, int(jittered),
                                                  sample, seed, *ptrs), "pt_camera_rays_dev")


class Film:
    """Device-resident accumulation canvas (pt_film): W*H*{r,g,b,passes} float32."""

    def __init__(self, scene: GpuScene, width: int, height: int, external=None):
        L = native.hip_lib()
        self.scene = scene
        self.width, self.height = width, height
        self._external = external  # keeps the torch tensor alive
        h = C.c_void_p()
        ext = C.c_void_p(external.data_ptr()) if external is not None else None
        check(L.pt_film_create(scene.h, width, height, ext, C.byref(h)), "pt_film_create")
        self.h = h
        self.zero()

    def close(self):
        if getattr(self, "h", None):
            native.hip_lib().pt_film_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def zero(self):
        check(native.hip_lib().pt_film_zero(self.h), "pt_film_zero")

    def render(self, cam: Camera, lights, passes, seed=1, want_stats=True):
        """passes: list of (x, y, w, h, first_sample, n_samples)."""
        larr = make_lights(lights)
        parr = (TilePass * max(len(passes), 1))()
        for i, p in enumerate(passes):
            parr[i].x, parr[i].y, parr[i].w, parr[i].h, parr[i].first_sample, parr[i].n_samples = p
        st = RenderStats()
        check(native.hip_lib().pt_render_passes(self.h, C.byref(cam), larr, len(lights), parr, len(passes), seed,
                                                C.byref(st) if want_stats else None), "pt_render_passes")
        return st if want_stats else None

    def read(self):
        rgb = np.empty((self.height, self.width, 3), dtype=np.float32)
        passes = np.empty((self.height, self.width), dtype=np.int32)
        check(native.hip_lib().pt_film_read(self.h, _p(rgb), _p(passes)), "pt_film_read")
        return rgb, passes

    def add(self, rgb, passes):
        """Adds host accumulators (as returned by read()) into the film."""
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        passes = np.ascontiguousarray(passes, dtype=np.int32)
        if rgb.shape != (self.height, self.width, 3) or passes.shape != (self.height, self.width):
            raise ValueError("accumulator shapes do not match the film")
        check(native.hip_lib().pt_film_add(self.h, _p(rgb), _p(passes)), "pt_film_add")
