/*
 * shade_core.h -- per-lane camera-ray generation and surface shading of the integrator.
 *
 * Follows, operation by operation, Integrator::perform_pass (reference integrator.cpp:211-275),
 * Integrator::get_ray_for_pixel (:197-209) and the hit branch of Integrator::cast_ray
 * (:64-148).  3-vector arithmetic is strict binary32 in the evaluation order Eigen uses for
 * fixed-size vectors: dot = a0*b0 + (a1*b1 + a2*b2); normalize divides each component by
 * sqrtf(dot(a,a)) and leaves a zero vector unchanged; vector/scalar is a true division.
 *
 * Compiled by hipcc for the kernels and by g++ for the CPU-side lane emulation used in tests.
 */
#ifndef PT_SHADE_CORE_H
#define PT_SHADE_CORE_H

#include "rng_core.h"

namespace ptc {

struct V3f { float x, y, z; };
PT_HD V3f v3(float x, float y, float z) { V3f r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD V3f vadd(V3f a, V3f b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3f vsub(V3f a, V3f b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3f vscale(V3f a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3f vscale_l(float s, V3f a) { return v3(s * a.x, s * a.y, s * a.z); }
PT_HD V3f vdiv(V3f a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
PT_HD float vdot(V3f a, V3f b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
PT_HD V3f vcross(V3f a, V3f b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
PT_HD float vnorm(V3f a) { return pt_sqrtf(vdot(a, a)); }
PT_HD V3f vunit(V3f a) {
    float z = vdot(a, a);
    if (z > 0.0f) { float n = pt_sqrtf(z); return v3(a.x / n, a.y / n, a.z / n); }
    return a;
}

/* Camera state of one render call; right/up/aspect are derived once on the host
 * (reference integrator.cpp:215-220). */
struct CamParams {
    V3f origin, direction, right, up;
    float aspect;
    float image_plane_width, plane_of_focus_distance, dof_dispersion;
    V3f sky;
    int width, height;
    uint32_t bounce_mode;
    uint32_t shadow_mode;
};

PT_HD void cam_basis(CamParams& c, V3f scene_up) {
    c.right = vunit(vcross(c.direction, scene_up));
    c.up = vunit(vcross(c.right, c.direction));
    c.aspect = (float)c.height / (float)c.width;
}

/* Integrator::get_ray_for_pixel: integer W/2 and H/2, all-float arithmetic. */
PT_HD void camera_ray_plain(const CamParams& c, int x, int y, V3f& o, V3f& d) {
    float dx = c.image_plane_width * (float)(x - c.width / 2) / (float)c.width;
    float dy = -c.image_plane_width * (float)(y - c.height / 2) * c.aspect / (float)c.height;
    V3f offset = vadd(vscale(c.right, dx), vscale(c.up, dy));
    o = c.origin;
    d = vunit(vadd(c.direction, offset));
}

/* Integrator::perform_pass, per pixel: the offsets are formed in double (the jitter is a
 * double) and narrowed on assignment; the aperture shift moves the origin and tilts the
 * direction, which is not renormalised afterwards. */
PT_HD void camera_ray_jittered(const CamParams& c, int x, int y, uint32_t sample, uint64_t seed, V3f& o, V3f& d) {
    Rng4 w = rng_block(seed, (uint32_t)(y * c.width + x), sample, 0u);
    double jx = uniform01(w.w[0]) - 0.5, jy = uniform01(w.w[1]) - 0.5;
    float dx = (float)((double)c.image_plane_width * ((double)x + jx - (double)c.width / 2.0) / (double)(float)c.width);
    float dy = (float)((double)(-c.image_plane_width) * ((double)y + jy - (double)c.height / 2.0) * (double)c.aspect /
                       (double)(float)c.height);
    V3f offset = vadd(vscale(c.right, dx), vscale(c.up, dy));
    o = c.origin;
    d = vunit(vadd(c.direction, offset));
    float dof_x = normal_from_word(w.w[2]) * c.dof_dispersion;
    float dof_y = normal_from_word(w.w[3]) * c.dof_dispersion;
    o = vadd(o, vscale_l(dof_x, c.right));
    o = vadd(o, vscale_l(dof_y, c.up));
    d = vsub(d, vscale_l(dof_x / c.plane_of_focus_distance, c.right));
    d = vsub(d, vscale_l(dof_y / c.plane_of_focus_distance, c.up));
}

struct ShadeTri {
    V3f normal; float plane;
    V3f base_n, u_n, v_n;
    V3f d1; /* edge01.normalized() */
};

struct SurfaceFrame {
    V3f n;     /* Phong-interpolated normal          (:66-67) */
    V3f hit;   /* hit point lifted 1e-3 off the plane (:69-72) */
    V3f refl;  /* mirror direction                    (:73-74) */
};

PT_HD SurfaceFrame surface_frame(V3f o, V3f d, float t, float u, float v, const ShadeTri& tr) {
    SurfaceFrame f;
    f.n = vunit(vadd(vadd(tr.base_n, vscale_l(u, tr.u_n)), vscale_l(v, tr.v_n)));
    V3f hit = vadd(o, vscale_l(t, d));
    /* Triangle::project_point_to_given_altitude, reference utils.cpp:165-170 */
    float parameter = vdot(tr.normal, hit);
    float change = (tr.plane - parameter) + 1e-3f;
    f.hit = vadd(hit, vscale_l(change, tr.normal));
    f.refl = vunit(vsub(d, vscale_l(2.0f * vdot(f.n, d), f.n)));
    return f;
}

/* The bounce ray: Ray(hit, reflection), whose constructor normalises again (:104-105). */
PT_HD void bounce_ray(const SurfaceFrame& f, V3f& o, V3f& d) { o = f.hit; d = vunit(f.refl); }

/* The bounce ray of the Lambertian mode (:79-84, i.e. the reference without :104): a unit-sphere
 * sample (utils.cpp:177-185: three standard normals, normalised) with its first component
 * made non-negative, taken in the triangle's frame {normal, d1, normal x d1}.  Stream 1 + depth
 * (+ stream_off, see BranchOpts in pt_hip.hip) of the path's random stream supplies the three normals. */
PT_HD void bounce_ray_lambert(const SurfaceFrame& f, const ShadeTri& tr, uint64_t seed, uint32_t pixel, uint32_t sample,
                              uint32_t depth, uint32_t stream_off, V3f& o, V3f& d) {
    Rng4 w = rng_block(seed, pixel, sample, 1u + depth + stream_off);
    V3f local = vunit(v3(normal_from_word(w.w[0]), normal_from_word(w.w[1]), normal_from_word(w.w[2])));
    local.x = __builtin_fabsf(local.x);
    V3f d2 = vcross(tr.normal, tr.d1);
    V3f scatter = vadd(vadd(vscale_l(local.x, tr.normal), vscale_l(local.y, tr.d1)), vscale_l(local.z, d2));
    o = f.hit;
    d = vunit(scatter);
}

struct LightSample {
    V3f dir;      /* normalised shadow-ray direction (Ray constructor) */
    float dist;   /* |to_light| */
    V3f weight;   /* colour / dist^2 * (lambert + phong^16): what the light adds if visible */
};

PT_HD float max0(float x) { return 0.0f < x ? x : 0.0f; } /* real_max(0.0, x) */
PT_HD float sq(float x) { return x * x; }

/* One light of the loop at :112-148 for path vertex `depth`. */
PT_HD LightSample light_sample(const SurfaceFrame& f, V3f light_pos, V3f light_color, uint64_t seed, uint32_t pixel,
                               uint32_t sample, uint32_t depth, uint32_t light, uint32_t stream_off = 0u) {
    Rng4 w = rng_block(seed, pixel, sample, 16u + 16u * depth + light + stream_off);
    V3f jitter = v3(0.2f * normal_from_word(w.w[0]), 0.2f * normal_from_word(w.w[1]), 0.2f * normal_from_word(w.w[2]));
    V3f to_light = vsub(vadd(jitter, light_pos), f.hit);
    LightSample s;
    s.dir = vunit(to_light);
    s.dist = vnorm(to_light);
    V3f contribution = vdiv(light_color, s.dist * s.dist);
    float lambert = max0(vdot(f.n, to_light) / s.dist);
    float phong = max0(vdot(f.refl, to_light) / s.dist);
    phong = sq(sq(sq(sq(phong))));
    s.weight = vscale(contribution, lambert + phong);
    return s;
}

} // namespace ptc

#endif
