"""The jittered camera rays of Integrator::perform_pass (reference integrator.cpp:211-259) held to what the
reference's lines imply, by statistics computed here in numpy on nothing but the rays that come out:

  - the origin is displaced along camera_right and camera_up by two independent N(0, dof_dispersion)
    amounts (:254-257);
  - the direction is tilted by the same amounts divided by the plane-of-focus distance and is NOT
    renormalised (:258-259), so origin + plane_of_focus_distance * direction does not depend on the
    displacement: it is camera origin + plane_of_focus_distance * (the unit direction through the
    jittered pixel position), at distance plane_of_focus_distance from the camera exactly;
  - the pixel position is jittered by U(-0.5, 0.5) in x and y (:248-249), with dx = ipw * (x + U - W/2) / W
    and dy = -ipw * (y + U - H/2) / W (aspect_ratio / H = 1 / W).

The oracle's rays are analysed on the CPU, the GPU's rays are bit-identical to the oracle's
(tests/test_gpu_raycast.py::test_camera_rays_bit_exact)."""
import numpy as np

from conftest import orc

W, H, SAMPLES = 48, 32, 96
DOF, POF, IPW = 0.1, 4.5, 0.75


def unit(v):
    return v / np.linalg.norm(v)


def test_jittered_camera_rays_statistics():
    cam = orc.cli_camera(dof_aperture=DOF, dof_distance=POF)
    cam_o = np.array(cam.origin[:], np.float64)
    cam_d = np.array(cam.direction[:], np.float64)
    right = unit(np.cross(cam_d, np.array(cam.up[:], np.float64)))       # :215-217
    up = unit(np.cross(right, cam_d))                                     # :218-219
    a, b, jx, jy, focus_r = [], [], [], [], []
    ys, xs = np.mgrid[0:H, 0:W]
    for s in range(SAMPLES):
        o, d = orc.camera_rays(cam, W, H, jittered=True, sample=s, seed=3)
        o = o.astype(np.float64).reshape(H, W, 3)
        d = d.astype(np.float64).reshape(H, W, 3)
        a.append((o - cam_o) @ right)
        b.append((o - cam_o) @ up)
        through = (o + POF * d - cam_o) / POF                             # the unit direction through the jittered pixel position
        focus_r.append(np.linalg.norm(through, axis=2))
        # undo the normalisation of Ray's constructor: direction + offset has component 1 along the camera direction
        scale = through @ cam_d
        dx = (through @ right) / scale
        dy = (through @ up) / scale
        jx.append(dx * W / IPW - (xs - W / 2.0))
        jy.append(-dy * W / IPW - (ys - H / 2.0))
    a, b, jx, jy, focus_r = (np.concatenate([v.ravel() for v in q]) for q in (a, b, jx, jy, focus_r))
    n = len(a)
    assert n == W * H * SAMPLES
    # aperture: two independent Gaussians of standard deviation dof_dispersion
    for g in (a, b):
        assert abs(g.mean()) < 5 * DOF / np.sqrt(n)
        assert abs(g.std() / DOF - 1) < 0.01
        assert abs(((g / DOF) ** 4).mean() - 3.0) < 0.1                   # a Gaussian's fourth moment
    assert abs(np.corrcoef(a, b)[0, 1]) < 0.01
    # the plane of focus is met at the same point whatever the displacement
    assert np.abs(focus_r - 1.0).max() < 1e-5
    # anti-aliasing jitter: uniform on (-0.5, 0.5), independent in x and y and of the aperture
    for j in (jx, jy):
        assert j.min() > -0.5 - 1e-3 and j.max() < 0.5 + 1e-3
        assert abs(j.mean()) < 5 / np.sqrt(12 * n) + 1e-4
        assert abs(j.std() * np.sqrt(12) - 1) < 0.01
    assert abs(np.corrcoef(jx, jy)[0, 1]) < 0.01 and abs(np.corrcoef(jx, a)[0, 1]) < 0.01 and abs(np.corrcoef(jy, b)[0, 1]) < 0.01
    # different samples of one pixel differ, and the same (pixel, sample, seed) repeats
    o1, d1 = orc.camera_rays(cam, W, H, jittered=True, sample=5, seed=3)
    o2, d2 = orc.camera_rays(cam, W, H, jittered=True, sample=5, seed=3)
    o3, _ = orc.camera_rays(cam, W, H, jittered=True, sample=6, seed=3)
    assert np.array_equal(o1, o2) and np.array_equal(d1, d2) and not np.array_equal(o1, o3)
