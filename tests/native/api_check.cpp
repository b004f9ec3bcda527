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
    delete scene;
    return 0;
}
