/*
 * pt_render.h -- the reference renderer's C++ scene/camera/render API, backed by the GPU.
 *
 * Source-compatible with what the reference's front ends use from utils.h, kdtree.h,
 * canvas.h, integrator.h and visualizer.h (ProgressBar), so that cli_render.cpp, main.cpp
 * and animate.cpp keep working when they include this header instead:
 *
 *   Vec / Color / Real, Ray, Triangle, Light, Scene, kdTree, PassDescriptor, Canvas,
 *   Integrator, RenderEngine, ProgressReporter / ProgressBar,
 *   override_thread_count, get_optimal_thread_count, format_seconds_as_hms, rays_cast, triangle_tests,
 *   start_performance_counter, print_performance_counter.
 *
 * Differences that a caller can observe:
 *   - Workers are GPUs, not CPU threads: get_optimal_thread_count() is the number of
 *     visible MI355X devices (override_thread_count limits it).
 *   - Random numbers come from a counter-based stream keyed by (seed, pixel, sample index),
 *     not from a per-thread std::mt19937 seeded from std::random_device, so renders are
 *     reproducible; RenderEngine::seed selects the stream.
 *   - rays_cast is maintained (the reference left its increment commented out,
 *     kdtree.cpp:578), so ProgressBar's "Perf: ...Mr/s" is a real number.
 *   - kdTree::ray_test hands back a pointer into Scene::mesh, not into a leaf's private copy.
 *   - The SDL display (ProgressDisplay) is not provided.
 * Everything below the API goes through the C ABI of pt_hip.h; there is no CPU ray caster here.
 */
#ifndef PT_RENDER_H
#define PT_RENDER_H

#include <pthread.h>
#include <semaphore.h>
#include <stdint.h>

#include <cmath>
#include <list>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "pt_hip.h"
#include "pt_host.h"

#define FLOAT_INF (1e100)

typedef float Real;

/* 3-vector with the slice of Eigen::Vector3f's interface the reference uses (utils.h:21).
 * Evaluation order matches Eigen's fixed-size kernels: dot = a0*b0 + (a1*b1 + a2*b2). */
struct Vec {
    float v[3];
    Vec() { v[0] = v[1] = v[2] = 0.0f; }
    template <class A, class B, class C>
    Vec(A x, B y, C z) { v[0] = (float)x; v[1] = (float)y; v[2] = (float)z; }
    float operator()(int i) const { return v[i]; }
    float& operator()(int i) { return v[i]; }
    float dot(const Vec& o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
    Vec cross(const Vec& o) const {
        return Vec(v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]);
    }
    float squaredNorm() const { return dot(*this); }
    float norm() const { return sqrtf(dot(*this)); }
    void normalize() { float z = squaredNorm(); if (z > 0.0f) { float n = sqrtf(z); v[0] /= n; v[1] /= n; v[2] /= n; } }
    Vec normalized() const { Vec r = *this; r.normalize(); return r; }
    Vec operator-() const { return Vec(-v[0], -v[1], -v[2]); }
    Vec operator+(const Vec& o) const { return Vec(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
    Vec operator-(const Vec& o) const { return Vec(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]); }
    Vec& operator+=(const Vec& o) { v[0] += o.v[0]; v[1] += o.v[1]; v[2] += o.v[2]; return *this; }
    Vec& operator-=(const Vec& o) { v[0] -= o.v[0]; v[1] -= o.v[1]; v[2] -= o.v[2]; return *this; }
    /* a double scalar is narrowed to float first, as Eigen promotes scalars of float expressions */
    Vec operator*(double s) const { float f = (float)s; return Vec(v[0] * f, v[1] * f, v[2] * f); }
    Vec operator/(double s) const { float f = (float)s; return Vec(v[0] / f, v[1] / f, v[2] / f); }
};
inline Vec operator*(double s, const Vec& a) { float f = (float)s; return Vec(f * a.v[0], f * a.v[1], f * a.v[2]); }
typedef Vec Color;

inline static Real real_min(Real x, Real y) { return x < y ? x : y; } /* utils.h:29-31 */
inline static Real real_max(Real x, Real y) { return x < y ? y : x; } /* utils.h:33-35 */

struct Ray { /* utils.h:64-71 */
    Vec origin, direction;
    Ray();
    Ray(Vec origin, Vec direction); /* normalises the direction (utils.cpp:20-23) */
    Real distance_along_ray(Vec p) const;
};

struct AABB { Vec minima, maxima; };

struct Triangle { /* utils.h:94-113; filled by the scene loader */
    Vec points[3];
    Vec edge01, edge02;
    Vec normal;
    AABB aabb;
    Vec base_normal, u_normal, v_normal;
    Vec assigned_normal;
};

extern long long rays_cast; /* kdtree.h:15: closest-hit plus shadow casts, summed over all GPUs */
extern long long triangle_tests; /* utils.h:12: the reference never counts (utils.cpp:139 is commented out); stays 0 here too */

void start_performance_counter();   /* utils.cpp:217-229: a wall-clock stopwatch; print_... writes the seconds to cout */
void print_performance_counter();

void override_thread_count(int thread_count);
int get_optimal_thread_count();
std::string format_seconds_as_hms(double seconds, int width);

struct kdTreeNode { /* only what callers touch: the statistics walk (kdtree.cpp:411-420) */
    int deepest, biggest;
    void get_stats(int& deepest_depth, int& biggest_set) const;
};

class kdTree { /* kdtree.h:64-85 */
    friend struct Scene;
    friend struct RenderEngine;
    friend struct Integrator;
    friend struct RenderThread;
    std::vector<Triangle>* all_triangles;
    pth_scene* built;                 /* mesh, tree and flattened arrays on the host */
    std::vector<pt_scene*> on_device; /* uploaded lazily, one per GPU */
    std::vector<pthread_mutex_t*> device_mutex; /* one per uploaded scene: a pt_scene is driven by one thread at a time */
    std::vector<pthread_mutex_t*> upload_mutex; /* one per GPU: held while its scene is being created, so GPUs upload side by side */
    pthread_mutex_t lock;
    kdTree(std::vector<Triangle>* all_triangles, pth_scene* prebuilt); /* adopts a finished build */
    void lock_device(int device);
    void unlock_device(int device);
public:
    kdTreeNode* root;
    kdTree(std::vector<Triangle>* all_triangles); /* builds from the triangles' corner points */
    ~kdTree();
    /* Closest hit of one ray, as a one-ray batch on GPU 0.  Outputs are written only on a hit.
     * As in the reference, it may be called from any thread while workers render. */
    bool ray_test(const Ray& ray, Real& hit_parameter, Real& hit_u, Real& hit_v, const Triangle** hit_triangle = nullptr);
    pt_scene* device_scene(int device); /* NULL on failure; pt_last_error() has the reason */
};

struct Light { Vec position; Color color; }; /* integrator.h:15-18 */

struct Scene { /* integrator.h:20-33 */
    std::vector<Triangle>* mesh;
    std::vector<Light>* lights;
    kdTree* tree;
    Ray main_camera;
    Vec scene_up;
    Real camera_image_plane_width;
    Real plane_of_focus_distance;
    Real dof_dispersion;
    Color sky_color;
    bool lambertian_bounces; /* extension, default false: bounce by integrator.cpp:79-84 instead of the mirror direction of :104 */
    bool bounded_shadow_rays; /* extension, default false: a light is hidden only by geometry nearer than it (kdtree.cpp:356-357) */
    Scene(std::string path);
    ~Scene();
};

struct PassDescriptor { /* integrator.h:35-43 */
    int start_x, start_y;
    int width, height; /* -1: full extent */
    PassDescriptor();
    PassDescriptor(int start_x, int start_y, int width, int height);
    void clamp_bounds(int max_width, int max_height);
};

class Canvas { /* canvas.h:8-27 */
public:
    int width, height, size;
    Real gain;
    Color* pixels;
    int* per_pixel_passes;
    Real* depth_buffer;
    Canvas(int width, int height);
    ~Canvas();
    void zero();
    Color* pixel_ptr(int x, int y);
    int* per_pixel_passes_ptr(int x, int y);
    Real* depth_ptr(int x, int y);
    void get_pixel(int x, int y, uint8_t* dest);
    void add_from(Canvas* other);
    int save(std::string path); /* 8-bit RGB PNG; 0 on success, -1 on failure */
    /* The accumulators themselves (energy sums and pass counts, no tone mapping), so that a
     * long render can be stopped and continued; 0 on success, -1 on failure (load: also when
     * the file's size differs from this canvas). */
    int save_accumulators(std::string path);
    int load_accumulators(std::string path);
};

struct RenderEngine;

/* integrator.h:45-60; renders on one GPU.  Several Integrators (the engine's worker of that GPU
 * among them) may share a GPU: their render calls take turns on that GPU's scene. */
struct Integrator {
    Scene* scene;
    Canvas* canvas;   /* host copy of the film, refreshed by perform_pass / download() */
    int passes;
    double last_pass_seconds;
    uint64_t seed;
    int device;
    pt_film* film;
    uint32_t next_sample;
    Integrator(int width, int height, Scene* scene, int device = 0);
    ~Integrator();
    Ray get_ray_for_pixel(int x, int y);
    /* Radiance along one ray of the caller's own (integrator.h:54, integrator.cpp:51-156), traced
     * on this integrator's GPU as a one-path batch.  recursions 0..10.  As in the reference, the
     * bounce term of the first hit is the mean of `branches` sub-paths, each weighted
     * (1.0/branches)*0.8 (at most PT_MAX_BRANCHES; 0 or less: no bounce term).  Each call draws from
     * a stream of its own (seed, call number), so repeated calls give independent samples. */
    Color cast_ray(const Ray& ray, int recursions, int branches);
    /* integrator.h:50-51 has a public std::mt19937 `engine` that callers may re-seed
     * (engine.seed(k)); this stands in for it: seeding it sets `seed` and restarts the
     * call numbering of cast_ray. */
    struct SeedHandle { Integrator* owner; void seed(uint64_t value); } engine;
    uint32_t cast_calls;
    void perform_pass(PassDescriptor desc = PassDescriptor());
    int download(); /* film -> canvas */
};

struct RenderMessage {
    bool do_die;
    PassDescriptor desc;
    uint32_t sample; /* index in the random stream of the first pass of the run */
    uint32_t count;  /* passes of this rectangle in the run, with consecutive sample indices */
};

struct RenderThread { /* integrator.h:67-89: one per GPU */
    pthread_t thread;
    RenderEngine* parent;
    volatile bool is_running;
    volatile PassDescriptor currently_processing;
    pthread_mutex_t integrator_lock;
    Integrator* integrator;
    RenderThread(RenderEngine* parent, int device);
    ~RenderThread();
    /* queue_locked: the caller holds parent->messages_lock (only RenderEngine::perform_full_passes does) */
    void send_message(RenderMessage message, bool queue_locked = false);
    void kill_immediately();
    static void* render_thread_main(void* cookie);
};

struct RenderEngine { /* integrator.h:91-127 */
    int width, height;
    Scene* scene;
    Canvas* master_canvas;
    int total_passes_issued;
    int tile_width, tile_height;
    uint64_t seed;

    std::vector<RenderThread*> workers;
    sem_t passes_semaphore;          /* posted once per finished queue entry (reference field; sync() does not wait on it) */
    int semaphore_passes_pending;
    pthread_cond_t passes_cond;      /* signalled under master_lock whenever total_passes_completed grows */
    sem_t messages_semaphore;
    pthread_mutex_t messages_lock;
    std::list<RenderMessage> messages;
    pthread_mutex_t master_lock;
    int total_passes_completed;
    std::map<std::pair<std::pair<int, int>, std::pair<int, int> >, uint32_t> samples_issued;

    RenderEngine(int width, int height, Scene* scene);
    ~RenderEngine();
    void issue_pass_desc(PassDescriptor desc);
    /* `count` passes of one rectangle as ONE queue entry (perform_full_passes issues a tile's
     * passes this way: half a million queue entries for 1000 samples of a 1080p frame would cost
     * more host time than the GPUs need for a tenth of the render) */
    void issue_pass_run(PassDescriptor desc, int count, bool queue_locked = false);
    void perform_full_pass();
    void perform_full_passes(int pass_count);
    void sync();
    void kill_workers();
    /* Sums the workers' films into master_canvas and returns the number of passes in it, or -1 if
     * the films could not be summed (then master_canvas is zero and last_error says why). */
    int rebuild_master_canvas();
    void zero();
    /* What went wrong, if anything: render calls that failed on a GPU (their passes are counted as
     * completed so that sync() and the progress bar end, but they are not in the picture), and the
     * way the last rebuild_master_canvas summed the films ("rccl", "host", or "host after rccl
     * failed").  Front ends check render_errors after sync() and exit non-zero. */
    int render_errors;
    std::string last_error;
    std::string last_reduce_how;
    uint32_t next_global_sample; /* see issue_pass_desc */
    bool mixed_rectangles;
    /* Continues a render: zero(), then the canvas' accumulators become the first worker's film
     * and every rectangle's sample stream starts after the samples the canvas already holds
     * (its largest pass count).  0 on success. */
    int resume_from(Canvas* saved);
    uint32_t sample_base;
};

class ProgressReporter { /* visualizer.h:10-15 */
public:
    virtual ~ProgressReporter() {}
    virtual bool init() = 0;
    virtual void main_loop() = 0;
};

class ProgressBar : public ProgressReporter { /* visualizer.h:30-38 */
    RenderEngine* engine;
    struct timeval* start;
public:
    ProgressBar(RenderEngine* engine);
    ~ProgressBar();
    bool init();
    void main_loop();
};

#endif
