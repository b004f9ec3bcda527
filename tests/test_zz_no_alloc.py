"""pt_raycast_closest_dev_noalloc: the closest-hit entry point whose scratch is a static device
array (no hipMalloc in the call).  Its output must equal pt_raycast_closest_dev's bit for bit, on
two calls in a row, the second with a smaller input (the array keeps the first call's contents),
and requests beyond its limit must be refused."""
import ctypes as C

import numpy as np
import pytest

from conftest import bits, orc, pta, rays_sphere
from path_tracing_amd import native

pytestmark = pytest.mark.gpu


def _to_dev(o, d):
    import torch
    cols = [o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2]]
    return [torch.from_numpy(np.ascontiguousarray(c)).cuda() for c in cols]


def _outputs(n):
    import torch
    return [torch.full((n,), -7, dtype=torch.int32, device="cuda")] + \
           [torch.full((n,), -7.0, dtype=torch.float32, device="cuda") for _ in range(3)]


def test_noalloc_closest_matches_regular_entry_point(scene_medium):
    import torch
    gpu = scene_medium.gpu
    for n, seed in ((300000, 11), (5000, 12)):       # the second call is the smaller one
        o, d = rays_sphere(n, seed=seed)
        rays = _to_dev(o, d)
        want = _outputs(n)
        got = _outputs(n)
        torch.cuda.synchronize()
        gpu.raycast_closest_dev(rays, *want)
        gpu.sync()
        gpu.raycast_closest_dev(rays, *got, noalloc=True)
        gpu.sync()
        for g, w, name in zip(got, want, ("index", "t", "u", "v")):
            g, w = g.cpu().numpy(), w.cpu().numpy()
            assert np.array_equal(bits(g), bits(w)), (n, name)
        idx = got[0].cpu().numpy()
        assert (idx >= -1).all() and (idx >= 0).mean() > 0.3
        # and against the oracle on a sample, so that two equal wrong answers cannot pass
        sel = np.random.default_rng(seed).choice(n, 2000, replace=False)
        ref = scene_medium.oracle.raycast(np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]))
        for g, w in zip(got, ref):
            assert np.array_equal(bits(g.cpu().numpy()[sel]), bits(w))


def test_noalloc_limit_and_empty(scene_medium):
    L = native.hip_lib()
    gpu = scene_medium.gpu
    fake = [C.c_void_p(0x1000)] * 10
    assert L.pt_raycast_closest_dev_noalloc(gpu.h, (1 << 24) + 1, *fake) == -4
    assert b"PT_NOALLOC_MAX_RAYS" in L.pt_last_error()
    assert L.pt_raycast_closest_dev_noalloc(gpu.h, 0, *fake) == 0
    assert L.pt_raycast_closest_dev_noalloc(None, 4, *fake) == -1
