/*
 * surface_probe.cpp -- compile-time check of include/pt_render.h against the public surface of the
 * reference's headers that a front end can reach: every member a caller may name (reference
 * integrator.h:15-127, canvas.h:8-27, kdtree.h:15, 64-85, utils.h:10-27, 64-130, visualizer.h:10-38,
 * minus the SDL display) exists here with the reference's type, and every member function takes and
 * returns what the reference's does.  TEST HARNESS ONLY: it is compiled (tests/test_abi.py), never run.
 */
#include <string>
#include <type_traits>
#include <vector>

#include "pt_render.h"

#define MEMBER(S, m, T) static_assert(std::is_same<decltype(S::m), T>::value, #S "::" #m " has the reference's type")
#define METHOD(S, m, SIG) static_assert(std::is_same<decltype(static_cast<SIG>(&S::m)), SIG>::value, #S "::" #m)

/* utils.h */
static_assert(std::is_same<Real, float>::value, "Real = float unless DOUBLE_PRECISION (utils.h:19-24)");
static_assert(std::is_same<Color, Vec>::value || sizeof(Color) == sizeof(Vec), "Color is a 3-vector like Vec");
MEMBER(Ray, origin, Vec);
MEMBER(Ray, direction, Vec);
MEMBER(Triangle, edge01, Vec); MEMBER(Triangle, edge02, Vec); MEMBER(Triangle, normal, Vec); MEMBER(Triangle, aabb, AABB);
MEMBER(Triangle, base_normal, Vec); MEMBER(Triangle, u_normal, Vec); MEMBER(Triangle, v_normal, Vec); MEMBER(Triangle, assigned_normal, Vec);
static_assert(std::is_same<decltype(Triangle::points), Vec[3]>::value, "Triangle::points");
static_assert(std::is_same<decltype(&override_thread_count), void (*)(int)>::value, "override_thread_count");
static_assert(std::is_same<decltype(&get_optimal_thread_count), int (*)()>::value, "get_optimal_thread_count");
static_assert(std::is_same<decltype(&start_performance_counter), void (*)()>::value, "start_performance_counter");
static_assert(std::is_same<decltype(&print_performance_counter), void (*)()>::value, "print_performance_counter");
static_assert(std::is_same<decltype(&format_seconds_as_hms), std::string (*)(double, int)>::value, "format_seconds_as_hms");
static_assert(std::is_same<decltype(rays_cast), long long>::value && std::is_same<decltype(triangle_tests), long long>::value, "counters");

/* canvas.h */
MEMBER(Canvas, width, int); MEMBER(Canvas, height, int); MEMBER(Canvas, size, int); MEMBER(Canvas, gain, Real);
MEMBER(Canvas, pixels, Color*); MEMBER(Canvas, per_pixel_passes, int*); MEMBER(Canvas, depth_buffer, Real*);
static_assert(std::is_constructible<Canvas, int, int>::value, "Canvas(int, int)");
METHOD(Canvas, zero, void (Canvas::*)());
METHOD(Canvas, pixel_ptr, Color* (Canvas::*)(int, int));
METHOD(Canvas, per_pixel_passes_ptr, int* (Canvas::*)(int, int));
METHOD(Canvas, depth_ptr, Real* (Canvas::*)(int, int));
METHOD(Canvas, get_pixel, void (Canvas::*)(int, int, uint8_t*));
METHOD(Canvas, add_from, void (Canvas::*)(Canvas*));
METHOD(Canvas, save, int (Canvas::*)(std::string));

/* kdtree.h */
static_assert(std::is_constructible<kdTree, std::vector<Triangle>*>::value, "kdTree(vector<Triangle>*)");
METHOD(kdTree, ray_test, bool (kdTree::*)(const Ray&, Real&, Real&, Real&, const Triangle**));
static_assert(std::is_pointer<decltype(kdTree::root)>::value, "kdTree::root");

/* integrator.h */
MEMBER(Light, position, Vec); MEMBER(Light, color, Color);
MEMBER(Scene, mesh, std::vector<Triangle>*); MEMBER(Scene, lights, std::vector<Light>*); MEMBER(Scene, tree, kdTree*);
MEMBER(Scene, main_camera, Ray); MEMBER(Scene, scene_up, Vec); MEMBER(Scene, camera_image_plane_width, Real);
MEMBER(Scene, plane_of_focus_distance, Real); MEMBER(Scene, dof_dispersion, Real); MEMBER(Scene, sky_color, Color);
static_assert(std::is_constructible<Scene, std::string>::value, "Scene(std::string)");
MEMBER(PassDescriptor, start_x, int); MEMBER(PassDescriptor, start_y, int); MEMBER(PassDescriptor, width, int); MEMBER(PassDescriptor, height, int);
static_assert(std::is_default_constructible<PassDescriptor>::value && std::is_constructible<PassDescriptor, int, int, int, int>::value, "PassDescriptor");
METHOD(PassDescriptor, clamp_bounds, void (PassDescriptor::*)(int, int));
MEMBER(Integrator, scene, Scene*); MEMBER(Integrator, canvas, Canvas*); MEMBER(Integrator, passes, int); MEMBER(Integrator, last_pass_seconds, double);
static_assert(std::is_constructible<Integrator, int, int, Scene*>::value, "Integrator(int, int, Scene*)");
METHOD(Integrator, cast_ray, Color (Integrator::*)(const Ray&, int, int));
METHOD(Integrator, get_ray_for_pixel, Ray (Integrator::*)(int, int));
METHOD(Integrator, perform_pass, void (Integrator::*)(PassDescriptor));
MEMBER(RenderMessage, do_die, bool); MEMBER(RenderMessage, desc, PassDescriptor);
MEMBER(RenderThread, parent, RenderEngine*); MEMBER(RenderThread, integrator, Integrator*);
MEMBER(RenderEngine, width, int); MEMBER(RenderEngine, height, int); MEMBER(RenderEngine, scene, Scene*);
MEMBER(RenderEngine, master_canvas, Canvas*); MEMBER(RenderEngine, total_passes_issued, int);
MEMBER(RenderEngine, tile_width, int); MEMBER(RenderEngine, tile_height, int);
MEMBER(RenderEngine, workers, std::vector<RenderThread*>);
MEMBER(RenderEngine, master_lock, pthread_mutex_t); MEMBER(RenderEngine, total_passes_completed, int);
static_assert(std::is_constructible<RenderEngine, int, int, Scene*>::value, "RenderEngine(int, int, Scene*)");
METHOD(RenderEngine, issue_pass_desc, void (RenderEngine::*)(PassDescriptor));
METHOD(RenderEngine, perform_full_pass, void (RenderEngine::*)());
METHOD(RenderEngine, perform_full_passes, void (RenderEngine::*)(int));
METHOD(RenderEngine, sync, void (RenderEngine::*)());
METHOD(RenderEngine, kill_workers, void (RenderEngine::*)());
METHOD(RenderEngine, rebuild_master_canvas, int (RenderEngine::*)());
METHOD(RenderEngine, zero, void (RenderEngine::*)());

/* visualizer.h (the progress bar; the SDL display is not provided) */
static_assert(std::is_abstract<ProgressReporter>::value && std::has_virtual_destructor<ProgressReporter>::value, "ProgressReporter");
static_assert(std::is_base_of<ProgressReporter, ProgressBar>::value && std::is_constructible<ProgressBar, RenderEngine*>::value, "ProgressBar(RenderEngine*)");
METHOD(ProgressBar, init, bool (ProgressBar::*)());
METHOD(ProgressBar, main_loop, void (ProgressBar::*)());

int main() { return 0; }
