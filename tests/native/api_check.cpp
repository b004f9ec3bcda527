/*
 * api_check.cpp -- exercises the C++ scene/render API (include/pt_render.h) the way the
 * reference's front ends do, beyond what cli_render touches: kdTree::ray_test one ray at a
 * time with the hit triangle pointer (interactive.cpp picks objects that way),
 * Integrator::get_ray_for_pixel, Integrator::perform_pass on a rectangle, Canvas accessors.
 * TEST HARNESS ONLY.  Prints plain numbers (hex floats) that tests/test_gpu_cpp_api.py compares
 * with the CPU oracle.
 *   api_check model.stl width height
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pt_render.h"

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    int W = atoi(argv[2]), H = atoi(argv[3]);
    Scene* scene = new Scene(argv[1]);
    if (scene->mesh == nullptr) return 1;
    scene->lights->push_back(Light({Vec(0, 0, 3), 9.0 * Vec(0.8, 0.5, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, 2, 4), 9.0 * Vec(0.25, 0.8, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, -2, 4), 9.0 * Vec(0.25, 0.25, 0.8)}));
    scene->camera_image_plane_width = 0.5 * 1.5;
    Real angle = 1.0;
    scene->main_camera.origin = -5.0 * Vec(cos(angle), sin(angle), 0.0);
    scene->main_camera.direction = -scene->main_camera.origin;
    scene->main_camera.direction.normalize();
    scene->main_camera.origin += Vec(0.0, 0.0, 0.2);

    int deepest = 0, biggest = 0;
    scene->tree->root->get_stats(deepest, biggest);
    printf("stats %d %d %zu\n", deepest, biggest, scene->mesh->size());

    Integrator* in = new Integrator(W, H, scene, 0);
    in->seed = 5;
    /* pixel-centre rays through the single-ray entry point */
    for (int y = 0; y < H; y += 7)
        for (int x = 0; x < W; x += 5) {
            Ray r = in->get_ray_for_pixel(x, y);
            Real t = 0, u = 0, v = 0;
            const Triangle* tri = nullptr;
            bool hit = scene->tree->ray_test(r, t, u, v, &tri);
            long idx = hit ? (long)(tri - scene->mesh->data()) : -1;
            printf("ray %d %d %08x %08x %08x %08x %08x %08x hit %ld %08x %08x %08x\n", x, y, bits(r.origin(0)), bits(r.origin(1)),
                   bits(r.origin(2)), bits(r.direction(0)), bits(r.direction(1)), bits(r.direction(2)), idx, bits(hit ? t : 0.0f),
                   bits(hit ? u : 0.0f), bits(hit ? v : 0.0f));
        }
    /* Integrator::cast_ray on a few pixel-centre rays, re-seeded as the reference's engine can be */
    in->engine.seed(9);
    for (int k = 0; k < 6; k++) {
        Ray r = in->get_ray_for_pixel(5 + 9 * k, 4 + 7 * k);
        Color c = in->cast_ray(r, k == 5 ? 2 : 10, 1);
        printf("cast %d %08x %08x %08x %08x %08x %08x rgb %08x %08x %08x\n", k, bits(r.origin(0)), bits(r.origin(1)), bits(r.origin(2)),
               bits(r.direction(0)), bits(r.direction(1)), bits(r.direction(2)), bits(c(0)), bits(c(1)), bits(c(2)));
    }
    in->seed = 5;
    /* two passes over a rectangle, one over the whole frame */
    in->perform_pass(PassDescriptor(8, 4, 24, 16));
    in->perform_pass(PassDescriptor(8, 4, 24, 16));
    in->perform_pass();
    printf("passes %d\n", in->passes);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            Color* c = in->canvas->pixel_ptr(x, y);
            unsigned char rgb[3];
            in->canvas->get_pixel(x, y, rgb);
            printf("px %d %d %d %08x %08x %08x %d %d %d\n", x, y, *in->canvas->per_pixel_passes_ptr(x, y), bits((*c)(0)), bits((*c)(1)),
                   bits((*c)(2)), rgb[0], rgb[1], rgb[2]);
        }
    delete in;

    /* RenderEngine with rectangles that overlap one another: a full-frame pass, four tiles, the
     * full frame again, the first tile again.  No pixel may see a sample index twice: the engine
     * numbers the passes 0; 1, 2, 3, 4; 5; 6.  Meanwhile this thread keeps casting single rays
     * (the reference's ray_test may be called while workers render). */
    RenderEngine* engine = new RenderEngine(W, H, scene);
    engine->seed = 5;
    engine->issue_pass_desc(PassDescriptor());
    const int tw = W / 2, th = H / 2;
    for (int k = 0; k < 4; k++) engine->issue_pass_desc(PassDescriptor((k & 1) * tw, (k >> 1) * th, tw, th));
    engine->issue_pass_desc(PassDescriptor());
    engine->issue_pass_desc(PassDescriptor(0, 0, tw, th));
    Ray probe = Ray(scene->main_camera.origin, scene->main_camera.direction);
    Real t0 = 0, u0 = 0, v0 = 0;
    const Triangle* tri0 = nullptr;
    bool hit0 = scene->tree->ray_test(probe, t0, u0, v0, &tri0);
    int same = 0, casts = 0;
    do {
        Real t = 0, u = 0, v = 0;
        const Triangle* tri = nullptr;
        bool hit = scene->tree->ray_test(probe, t, u, v, &tri);
        same += (hit == hit0 && tri == tri0 && bits(t) == bits(t0) && bits(u) == bits(u0) && bits(v) == bits(v0)) ? 1 : 0;
        casts++;
    } while (casts < 200);
    engine->sync();
    printf("concurrent %d %d %d\n", casts, same, hit0 ? 1 : 0);
    int total = engine->rebuild_master_canvas();
    printf("engine %d %d %d %s\n", total, engine->render_errors, engine->mixed_rectangles ? 1 : 0, engine->last_reduce_how.c_str());
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            Color* c = engine->master_canvas->pixel_ptr(x, y);
            printf("epx %d %d %d %08x %08x %08x\n", x, y, *engine->master_canvas->per_pixel_passes_ptr(x, y), bits((*c)(0)), bits((*c)(1)), bits((*c)(2)));
        }
    delete engine;
    delete scene;
    return 0;
}
