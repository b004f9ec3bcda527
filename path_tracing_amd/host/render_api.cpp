/*
 * render_api.cpp -- the reference's Scene / RenderEngine / Integrator / Canvas API on top of
 * the C ABI of pt_hip.h (see include/pt_render.h for the contract and the differences).
 *
 * One RenderThread per GPU pulls tile passes from the engine's shared queue, exactly as the
 * reference's CPU workers do (integrator.cpp:315-359), but takes as many queued passes as it
 * can find in one go and hands them to pt_render_passes() as one batch: the GPU needs
 * millions of rays in flight, a single 64x64 tile pass has 4096.
 */
#include "pt_render.h"

#include <sys/time.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <thread>

using namespace std;

long long rays_cast = 0;
long long triangle_tests = 0;

static std::chrono::steady_clock::time_point g_perf_counter_start;
void start_performance_counter() { g_perf_counter_start = std::chrono::steady_clock::now(); }
void print_performance_counter() { cout << std::chrono::duration<double>(std::chrono::steady_clock::now() - g_perf_counter_start).count(); }

/* ------------------------------------------------------------------- utils */

Ray::Ray() : origin(0, 0, 0), direction(0, 0, 1) {}
Ray::Ray(Vec _origin, Vec _direction) : origin(_origin), direction(_direction) { direction.normalize(); }
Real Ray::distance_along_ray(Vec p) const { return direction.dot(p - origin); }

static bool thread_count_is_overridden = false;
static int overridden_thread_count = 0;

void override_thread_count(int thread_count) { /* utils.cpp:190-194: zero means automatic */
    thread_count_is_overridden = thread_count != 0;
    overridden_thread_count = thread_count;
}

/* Workers are GPUs here: the number of visible devices, or the override if smaller. */
int get_optimal_thread_count() {
    int n = 0;
    if (pt_device_count(&n) != PT_OK || n <= 0) return 0;
    if (thread_count_is_overridden && overridden_thread_count > 0) return min(n, overridden_thread_count);
    return n;
}

string format_seconds_as_hms(double seconds, int width) { /* utils.cpp:231-264 */
    string s;
    bool is_negative = seconds < 0;
    if (is_negative) seconds *= -1;
    int total_seconds = (int)ceil(seconds);
    int second_count = total_seconds % 60, minute_count = (total_seconds / 60) % 60, hour_count = total_seconds / 3600;
    s = to_string(second_count);
    if (s.size() == 1) s = "0" + s;
    s = to_string(minute_count) + ":" + s;
    if (total_seconds >= 3600) {
        if (s.size() == 4) s = "0" + s;
        s = to_string(hour_count) + ":" + s;
    }
    if (is_negative) s = "-" + s;
    while ((int)s.size() < width) s = " " + s;
    return s;
}

/* ------------------------------------------------------------ mesh and tree */

static Vec vec3(const float* p) { return Vec(p[0], p[1], p[2]); }

/* Triangle fields as the reference's constructor and set_normals leave them. */
static void fill_triangles(const pth_scene* built, vector<Triangle>& tris) {
    int64_t n = pth_num_triangles(built);
    tris.resize((size_t)n);
    /* 156 bytes per triangle to fill: a handful of threads, each its own slice, a block at a time */
    const int n_threads = (int)max<int64_t>(1, min<int64_t>(8, n / 65536));
    vector<thread> pool;
    for (int w = 0; w < n_threads; w++)
        pool.emplace_back([&, w]() {
            const int64_t block = 4096;
            vector<float> data((size_t)block * 30);
            const int64_t end = n * (w + 1) / n_threads;
            for (int64_t at = n * w / n_threads; at < end; at += block) {
                const int64_t m = min(block, end - at);
                pth_triangle_data_range(built, at, at + m, data.data());
                for (int64_t i = 0; i < m; i++) {
                    const float* d = &data[(size_t)i * 30];
                    Triangle& t = tris[(size_t)(at + i)];
                    for (int k = 0; k < 3; k++) t.points[k] = vec3(d + 3 * k);
                    t.edge01 = vec3(d + 9); t.edge02 = vec3(d + 12); t.normal = vec3(d + 15);
                    t.base_normal = vec3(d + 18); t.u_normal = vec3(d + 21); t.v_normal = vec3(d + 24);
                    for (int a = 0; a < 3; a++) {
                        t.aabb.minima(a) = real_min(real_min(t.points[0](a), t.points[1](a)), t.points[2](a));
                        t.aabb.maxima(a) = real_max(real_max(t.points[0](a), t.points[1](a)), t.points[2](a));
                    }
                    t.assigned_normal = t.normal;
                }
            }
        });
    for (thread& th : pool) th.join();
}

void kdTreeNode::get_stats(int& deepest_depth, int& biggest_set) const {
    if (deepest > deepest_depth) deepest_depth = deepest;
    if (biggest > biggest_set) biggest_set = biggest;
}

static void finish_tree(kdTree* t, pth_scene* built, kdTreeNode*& root) {
    int64_t st[5];
    pth_tree_stats(built, st);
    root = new kdTreeNode();
    root->deepest = (int)st[2];
    root->biggest = (int)st[3];
    (void)t;
}

kdTree::kdTree(vector<Triangle>* _all_triangles) : all_triangles(_all_triangles), built(nullptr), root(nullptr) {
    pthread_mutex_init(&lock, NULL);
    vector<float> soup(all_triangles->size() * 9);
    for (size_t i = 0; i < all_triangles->size(); i++)
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) soup[i * 9 + k * 3 + a] = (*all_triangles)[i].points[k](a);
    built = pth_scene_from_triangles(soup.data(), (int64_t)all_triangles->size(), 0);
    if (built) { fill_triangles(built, *all_triangles); finish_tree(this, built, root); }
}

kdTree::kdTree(vector<Triangle>* _all_triangles, pth_scene* prebuilt) : all_triangles(_all_triangles), built(prebuilt), root(nullptr) {
    pthread_mutex_init(&lock, NULL);
    if (built) finish_tree(this, built, root);
}

kdTree::~kdTree() {
    for (pt_scene* s : on_device) pt_scene_destroy(s);
    for (pthread_mutex_t* m : device_mutex) if (m) { pthread_mutex_destroy(m); delete m; }
    for (pthread_mutex_t* m : upload_mutex) if (m) { pthread_mutex_destroy(m); delete m; }
    if (built) pth_scene_destroy(built);
    delete root;
    pthread_mutex_destroy(&lock);
}

pt_scene* kdTree::device_scene(int device) {
    if (!built || device < 0) return nullptr;
    pthread_mutex_lock(&lock);
    if ((int)on_device.size() <= device) {
        on_device.resize((size_t)device + 1, nullptr);
        device_mutex.resize((size_t)device + 1, nullptr);
        upload_mutex.resize((size_t)device + 1, nullptr);
    }
    if (!device_mutex[(size_t)device]) { device_mutex[(size_t)device] = new pthread_mutex_t; pthread_mutex_init(device_mutex[(size_t)device], NULL); }
    if (!upload_mutex[(size_t)device]) { upload_mutex[(size_t)device] = new pthread_mutex_t; pthread_mutex_init(upload_mutex[(size_t)device], NULL); }
    pt_scene* s = on_device[(size_t)device];
    pthread_mutex_t* creating = upload_mutex[(size_t)device];
    pthread_mutex_unlock(&lock);
    if (s) return s;
    /* created outside the tree's lock (other GPUs' scenes, and calls that only look one up, go on
     * meanwhile); the GPU's own upload lock makes it happen once */
    pthread_mutex_lock(creating);
    pthread_mutex_lock(&lock);
    s = on_device[(size_t)device];
    pthread_mutex_unlock(&lock);
    if (!s) {
        if (pt_scene_create(device, pth_flat_scene(built), &s) != PT_OK) s = nullptr;
        pthread_mutex_lock(&lock);
        on_device[(size_t)device] = s;
        pthread_mutex_unlock(&lock);
    }
    pthread_mutex_unlock(creating);
    return s;
}

/* Held around every call that drives the device's pt_scene (the handles are not locked inside). */
void kdTree::lock_device(int device) {
    pthread_mutex_lock(&lock);
    pthread_mutex_t* m = device >= 0 && device < (int)device_mutex.size() ? device_mutex[(size_t)device] : nullptr;
    pthread_mutex_unlock(&lock);
    if (m) pthread_mutex_lock(m);
}
void kdTree::unlock_device(int device) {
    pthread_mutex_lock(&lock);
    pthread_mutex_t* m = device >= 0 && device < (int)device_mutex.size() ? device_mutex[(size_t)device] : nullptr;
    pthread_mutex_unlock(&lock);
    if (m) pthread_mutex_unlock(m);
}

bool kdTree::ray_test(const Ray& ray, Real& hit_parameter, Real& hit_u, Real& hit_v, const Triangle** hit_triangle) {
    pt_scene* s = device_scene(0);
    if (!s) return false;
    int32_t idx = -1;
    float t = 0, u = 0, v = 0;
    float o[3] = {ray.origin(0), ray.origin(1), ray.origin(2)}, d[3] = {ray.direction(0), ray.direction(1), ray.direction(2)};
    /* a cast of up to 64 rays is serialised inside the handle and independent of rendering */
    int rc = pt_raycast_closest(s, 1, &o[0], &o[1], &o[2], &d[0], &d[1], &d[2], &idx, &t, &u, &v);
    __sync_fetch_and_add(&rays_cast, 1);
    if (rc != PT_OK || idx < 0) return false;
    hit_parameter = t; hit_u = u; hit_v = v;
    if (hit_triangle != nullptr) *hit_triangle = &(*all_triangles)[(size_t)idx];
    return true;
}

/* ------------------------------------------------------------------- scene */

Scene::Scene(string path) : main_camera(Vec(-1, 0, 0), Vec(1, 0, 0)) { /* integrator.cpp:11-39 */
    scene_up = Vec(0, 0, 1);
    camera_image_plane_width = 1.0;
    plane_of_focus_distance = 1.0;
    dof_dispersion = 0.0;
    sky_color = Vec(0, 0, 0);
    lambertian_bounces = false;
    bounded_shadow_rays = false;
    mesh = nullptr; lights = nullptr; tree = nullptr;
    /* the GPUs' contexts come up on a spare thread while this one reads the file and builds the tree */
    pthread_t warm;
    bool warming = pthread_create(&warm, nullptr, [](void*) -> void* {
        int n = get_optimal_thread_count();
        for (int i = 0; i < n; i++) (void)pt_device_prepare(i);
        return nullptr;
    }, nullptr) == 0;
    /* The host-side mesh is filled in beside the tree build, as soon as the vertex normals exist.  Its
     * memory (156 bytes per triangle, cleared by the vector on one thread) is made ready even earlier,
     * from the triangle count in the file's header, while the file is still being read. */
    mesh = new vector<Triangle>();
    struct Early { vector<Triangle>* mesh; thread sizing; } early{mesh, thread()};
    {
        uint32_t count = 0;
        FILE* fp = fopen(path.c_str(), "rb");
        const bool have = fp && fseek(fp, 80, SEEK_SET) == 0 && fread(&count, 4, 1, fp) == 1 && fseek(fp, 0, SEEK_END) == 0 &&
                          ftell(fp) >= 84 + (long)count * 50;
        if (fp) fclose(fp);
        if (have && count >= (1u << 18)) early.sizing = thread([count, m = mesh]() { m->resize(count); });
    }
    pth_scene* built = pth_scene_from_stl_with(path.c_str(), 0, [](const pth_scene* ready, void* ctx) {
        Early* e = (Early*)ctx;
        if (e->sizing.joinable()) e->sizing.join();
        fill_triangles(ready, *e->mesh);
    }, &early);
    if (early.sizing.joinable()) early.sizing.join();   /* the file turned out to be unreadable */
    if (warming) pthread_join(warm, nullptr);
    if (built == nullptr) {
        cout << "Couldn't read input file." << endl;
        delete mesh;
        mesh = nullptr;
        return;
    }
    /* adopt the already built tree instead of building it a second time from the points, and send
     * its flat arrays up to the GPUs */
    tree = new kdTree(mesh, built);
    const auto t0 = chrono::steady_clock::now();
    {   /* one thread per GPU */
        vector<thread> up;
        kdTree* t = tree;
        for (int i = 0, n = get_optimal_thread_count(); i < n; i++) up.emplace_back([t, i]() { (void)t->device_scene(i); });
        for (thread& th : up) th.join();
    }
    if (getenv("PT_BUILD_TIMING")) fprintf(stderr, "[scene] upload %.2f s\n", chrono::duration<double>(chrono::steady_clock::now() - t0).count());
    cout << "Read in " << mesh->size() << " triangles." << endl;
    lights = new vector<Light>();
}

Scene::~Scene() {
    delete tree;
    delete mesh;
    delete lights;
}

/* ------------------------------------------------------------------ canvas */

PassDescriptor::PassDescriptor() : start_x(0), start_y(0), width(-1), height(-1) {}
PassDescriptor::PassDescriptor(int sx, int sy, int w, int h) : start_x(sx), start_y(sy), width(w), height(h) {}
void PassDescriptor::clamp_bounds(int max_width, int max_height) { /* integrator.cpp:164-183 */
    if (start_x < 0) start_x = 0;
    if (start_y < 0) start_y = 0;
    if (start_x > max_width - 1) start_x = max_width - 1;
    if (start_y > max_height - 1) start_y = max_height - 1;
    if (width == -1) width = max_width;
    if (height == -1) height = max_height;
    if (width > max_width - start_x) width = max_width - start_x;
    if (height > max_height - start_y) height = max_height - start_y;
}

Canvas::Canvas(int _width, int _height) : width(_width), height(_height), gain(255.0) {
    size = width * height;
    pixels = new Color[size];
    per_pixel_passes = new int[size];
    depth_buffer = new Real[size];
}
Canvas::~Canvas() { delete[] pixels; delete[] per_pixel_passes; delete[] depth_buffer; }
void Canvas::zero() {
    for (int i = 0; i < size; i++) { pixels[i] = Vec(0, 0, 0); per_pixel_passes[i] = 0; depth_buffer[i] = 0.0; }
}
Color* Canvas::pixel_ptr(int x, int y) { return pixels + (x + y * width); }
int* Canvas::per_pixel_passes_ptr(int x, int y) { return per_pixel_passes + (x + y * width); }
Real* Canvas::depth_ptr(int x, int y) { return depth_buffer + (x + y * width); }
void Canvas::get_pixel(int x, int y, uint8_t* dest) { /* canvas.cpp:39-45 */
    Color c = pixels[x + y * width] / real_max(per_pixel_passes[x + y * width], 1.0);
    for (int i = 0; i < 3; i++) dest[i] = (uint8_t)real_max(0.0, real_min(255.0, (Real)(c(i) * gain)));
}
void Canvas::add_from(Canvas* other) { /* canvas.cpp:47-52 */
    for (int i = 0; i < size; i++) { pixels[i] += other->pixels[i]; per_pixel_passes[i] += other->per_pixel_passes[i]; }
}

static void png_chunk(FILE* fp, const char* type, const unsigned char* data, size_t len) {
    unsigned char hdr[8] = {(unsigned char)(len >> 24), (unsigned char)(len >> 16), (unsigned char)(len >> 8), (unsigned char)len,
                            (unsigned char)type[0], (unsigned char)type[1], (unsigned char)type[2], (unsigned char)type[3]};
    fwrite(hdr, 1, 8, fp);
    if (len) fwrite(data, 1, len, fp);
    uLong crc = crc32(0L, hdr + 4, 4);
    if (len) crc = crc32(crc, data, (uInt)len);
    unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
    fwrite(c, 1, 4, fp);
}

/* 8-bit RGB, no interlace, like the reference's libpng call (canvas.cpp:55-131); written with
 * zlib alone.  Returns 0 on success and -1 on any failure, never throws. */
static const char ACC_MAGIC[8] = {'P', 'T', 'A', 'C', 'C', '0', '1', 0};

int Canvas::save_accumulators(string path) {
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return -1;
    int32_t dims[2] = {width, height};
    bool ok = fwrite(ACC_MAGIC, 1, 8, fp) == 8 && fwrite(dims, 4, 2, fp) == 2;
    for (int i = 0; ok && i < size; i++) {
        float rgb[3] = {pixels[i](0), pixels[i](1), pixels[i](2)};
        ok = fwrite(rgb, 4, 3, fp) == 3;
    }
    ok = ok && fwrite(per_pixel_passes, sizeof(int), (size_t)size, fp) == (size_t)size;
    return (fclose(fp) == 0 && ok) ? 0 : -1;
}

int Canvas::load_accumulators(string path) {
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return -1;
    char magic[8];
    int32_t dims[2];
    bool ok = fread(magic, 1, 8, fp) == 8 && memcmp(magic, ACC_MAGIC, 8) == 0 && fread(dims, 4, 2, fp) == 2 &&
              dims[0] == width && dims[1] == height;
    for (int i = 0; ok && i < size; i++) {
        float rgb[3];
        ok = fread(rgb, 4, 3, fp) == 3;
        if (ok) pixels[i] = Color(rgb[0], rgb[1], rgb[2]);
    }
    ok = ok && fread(per_pixel_passes, sizeof(int), (size_t)size, fp) == (size_t)size;
    fclose(fp);
    return ok ? 0 : -1;
}

int Canvas::save(string path) {
    FILE* fp = fopen(path.c_str(), "wb");
    if (!fp) return -1;
    vector<unsigned char> raw((size_t)height * ((size_t)width * 3 + 1));
    for (int y = 0; y < height; y++) {
        unsigned char* row = &raw[(size_t)y * ((size_t)width * 3 + 1)];
        row[0] = 0;
        for (int x = 0; x < width; x++) get_pixel(x, y, row + 1 + 3 * x);
    }
    uLongf clen = compressBound((uLong)raw.size());
    vector<unsigned char> comp(clen);
    int status = -1;
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) == Z_OK) {
        static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
        fwrite(sig, 1, 8, fp);
        unsigned char ihdr[13] = {(unsigned char)(width >> 24), (unsigned char)(width >> 16), (unsigned char)(width >> 8), (unsigned char)width,
                                  (unsigned char)(height >> 24), (unsigned char)(height >> 16), (unsigned char)(height >> 8), (unsigned char)height,
                                  8, 2, 0, 0, 0};
        png_chunk(fp, "IHDR", ihdr, 13);
        png_chunk(fp, "IDAT", comp.data(), clen);
        png_chunk(fp, "IEND", nullptr, 0);
        status = ferror(fp) ? -1 : 0;
    }
    if (fclose(fp) != 0) status = -1;
    return status;
}

/* -------------------------------------------------------------- integrator */

static void fill_camera(pt_camera& c, const Scene* s) {
    for (int k = 0; k < 3; k++) {
        c.origin[k] = s->main_camera.origin(k); c.direction[k] = s->main_camera.direction(k);
        c.up[k] = s->scene_up(k); c.sky_color[k] = s->sky_color(k);
    }
    c.image_plane_width = s->camera_image_plane_width;
    c.plane_of_focus_distance = s->plane_of_focus_distance;
    c.dof_dispersion = s->dof_dispersion;
    c.bounce_mode = s->lambertian_bounces ? PT_BOUNCE_LAMBERT : PT_BOUNCE_MIRROR;
    c.shadow_mode = s->bounded_shadow_rays ? PT_SHADOW_BOUNDED : PT_SHADOW_REFERENCE;
}

static void fill_lights(vector<pt_light>& out, const Scene* s) {
    out.clear();
    if (!s->lights) return;
    for (const Light& l : *s->lights) {
        pt_light p;
        for (int k = 0; k < 3; k++) { p.position[k] = l.position(k); p.color[k] = l.color(k); }
        out.push_back(p);
    }
}

static double now_seconds() {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

Integrator::Integrator(int width, int height, Scene* _scene, int _device)
    : scene(_scene), passes(0), last_pass_seconds(0), seed(1), device(_device), film(nullptr), next_sample(0), cast_calls(0) {
    engine.owner = this;
    canvas = new Canvas(width, height);
    canvas->zero();
    pt_scene* ds = scene && scene->tree ? scene->tree->device_scene(device) : nullptr;
    if (ds && pt_film_create(ds, width, height, nullptr, &film) != PT_OK) film = nullptr;
    if (!film) cerr << "Integrator: no GPU film (" << pt_last_error() << ")" << endl;
}

Integrator::~Integrator() {
    if (film) pt_film_destroy(film);
    delete canvas;
}

Ray Integrator::get_ray_for_pixel(int x, int y) { /* integrator.cpp:197-209 */
    Vec camera_right = scene->main_camera.direction.cross(scene->scene_up);
    camera_right.normalize();
    Vec camera_up = camera_right.cross(scene->main_camera.direction);
    camera_up.normalize();
    Real aspect_ratio = canvas->height / (Real)canvas->width;
    Real dx = scene->camera_image_plane_width * (x - canvas->width / 2) / (Real)canvas->width;
    Real dy = -scene->camera_image_plane_width * (y - canvas->height / 2) * aspect_ratio / (Real)canvas->height;
    Vec offset = camera_right * dx + camera_up * dy;
    return Ray(scene->main_camera.origin, scene->main_camera.direction + offset);
}

void Integrator::SeedHandle::seed(uint64_t value) { owner->seed = value; owner->cast_calls = 0; }

Color Integrator::cast_ray(const Ray& ray, int recursions, int branches) {
    pt_scene* ds = scene && scene->tree ? scene->tree->device_scene(device) : nullptr;
    if (!ds) return Color(0, 0, 0);
    pt_camera cam;
    fill_camera(cam, scene);
    vector<pt_light> lights;
    fill_lights(lights, scene);
    float o[3] = {ray.origin(0), ray.origin(1), ray.origin(2)}, d[3] = {ray.direction(0), ray.direction(1), ray.direction(2)};
    /* streams of caller-made rays are kept apart from those of pixels (which are below 2^31) */
    uint32_t id = 0x80000000u | (cast_calls++ & 0x7fffffffu);
    float rgb[3] = {0, 0, 0};
    pt_render_stats st;
    scene->tree->lock_device(device);
    int rc = pt_cast_rays(ds, &cam, lights.data(), (int)lights.size(), 1, &o[0], &o[1], &o[2], &d[0], &d[1], &d[2], &id, 0,
                          min(max(recursions, 0), 10), min(branches, PT_MAX_BRANCHES), seed, rgb, &st);
    scene->tree->unlock_device(device);
    if (rc != PT_OK) { cerr << "Integrator::cast_ray: " << pt_last_error() << endl; return Color(0, 0, 0); }
    __sync_fetch_and_add(&rays_cast, (long long)(st.closest_rays + st.shadow_rays));
    return Color(rgb[0], rgb[1], rgb[2]);
}

/* One pass over the rectangle on this integrator's GPU, then the film is copied back so that
 * canvas reflects it (callers such as interactive.cpp read canvas right after perform_pass). */
void Integrator::perform_pass(PassDescriptor desc) {
    if (!film) return;
    double t0 = now_seconds();
    pt_camera cam;
    fill_camera(cam, scene);
    vector<pt_light> lights;
    fill_lights(lights, scene);
    pt_tile_pass p = {desc.start_x, desc.start_y, desc.width, desc.height, next_sample++, 1};
    pt_render_stats st;
    scene->tree->lock_device(device);
    int rc = pt_render_passes(film, &cam, lights.data(), (int)lights.size(), &p, 1, seed, &st);
    scene->tree->unlock_device(device);
    last_pass_seconds = now_seconds() - t0;
    if (rc != PT_OK) { cerr << "Integrator::perform_pass: " << pt_last_error() << endl; return; } /* a failed pass is not counted */
    __sync_fetch_and_add(&rays_cast, (long long)(st.closest_rays + st.shadow_rays));
    download();
    passes++;
}

int Integrator::download() {
    if (!film) return -1;
    vector<float> rgb((size_t)canvas->size * 3);
    scene->tree->lock_device(device);
    int rc = pt_film_read(film, rgb.data(), canvas->per_pixel_passes);
    scene->tree->unlock_device(device);
    if (rc != PT_OK) return -1;
    for (int i = 0; i < canvas->size; i++) canvas->pixels[i] = Vec(rgb[3 * (size_t)i], rgb[3 * (size_t)i + 1], rgb[3 * (size_t)i + 2]);
    return 0;
}

/* ----------------------------------------------------------- render engine */

RenderThread::RenderThread(RenderEngine* _parent, int device) : parent(_parent), is_running(false) {
    pthread_mutex_init(&integrator_lock, NULL);
    integrator = new Integrator(parent->width, parent->height, parent->scene, device);
    integrator->seed = parent->seed;
    pthread_create(&thread, nullptr, RenderThread::render_thread_main, (void*)this);
}

RenderThread::~RenderThread() {
    pthread_mutex_destroy(&integrator_lock);
    delete integrator;
}

void RenderThread::send_message(RenderMessage message, bool queue_locked) {
    if (!queue_locked) pthread_mutex_lock(&parent->messages_lock); /* else this very thread holds it */
    parent->messages.push_back(message);
    if (!queue_locked) pthread_mutex_unlock(&parent->messages_lock);
    sem_post(&parent->messages_semaphore);
}

void RenderThread::kill_immediately() { send_message(RenderMessage({true, PassDescriptor(), 0, 0})); }

/* Paths (pixels x samples, rectangles padded to 8x8 blocks) one worker hands to one render call.
 * 2^27 keeps a GPU at about 96 % of the rate of the largest batches while its scratch stays at
 * 38 GB instead of 152 GB -- which a short render feels, because the driver clears that memory
 * when the process ends (0.2 s against up to 1.3 s).  PT_ENGINE_BATCH_PATHS overrides it. */
static size_t engine_batch_paths() {
    const char* e = getenv("PT_ENGINE_BATCH_PATHS");
    size_t v = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 27;
    return max<size_t>(v, 4096);
}

void* RenderThread::render_thread_main(void* cookie) {
    RenderThread* self = (RenderThread*)cookie;
    RenderEngine* eng = self->parent;
    const size_t max_paths = engine_batch_paths();
    vector<pt_tile_pass> batch;
    vector<pt_light> lights;
    while (true) {
        sem_wait(&eng->messages_semaphore);
        bool die = false;
        size_t taken = 0;    /* passes in this batch */
        size_t finished = 0; /* queue entries used up by this batch (each owes one post of passes_semaphore) */
        size_t paths = 0;
        batch.clear();
        pthread_mutex_lock(&eng->messages_lock);
        /* The entry that woke us, then further ones already queued (each has a semaphore count of
         * its own, consumed with sem_trywait), until the batch holds this worker's share of the
         * queued work or max_paths paths.  A run that does not fit is split: the rest stays at the
         * head of the queue with its semaphore count. */
        size_t queued_paths = 0;
        for (const RenderMessage& m : eng->messages) {
            if (m.do_die) continue;
            PassDescriptor d = m.desc;
            d.clamp_bounds(eng->width, eng->height);
            queued_paths += (size_t)((d.width + 7) / 8) * (size_t)((d.height + 7) / 8) * 64 * m.count;
            if (queued_paths > max_paths * eng->workers.size()) break;
        }
        const size_t share = min(max_paths, max<size_t>((queued_paths + eng->workers.size() - 1) / max<size_t>(eng->workers.size(), 1), (size_t)1 << 22));
        while (true) {
            if (eng->messages.empty()) break; /* queue was cleared by kill_workers */
            RenderMessage& m = eng->messages.front();
            if (m.do_die) {
                if (batch.empty()) { eng->messages.pop_front(); die = true; }
                else sem_post(&eng->messages_semaphore); /* leave it queued: render what we hold first */
                break;
            }
            PassDescriptor d = m.desc;
            d.clamp_bounds(eng->width, eng->height);
            const size_t per_pass = max<size_t>((size_t)((d.width + 7) / 8) * (size_t)((d.height + 7) / 8) * 64, 1);
            size_t room = paths < share ? (share - paths) / per_pass : 0;
            if (room == 0 && batch.empty()) room = 1; /* a single pass larger than the share still goes out (and may be refused below) */
            if (room == 0) { sem_post(&eng->messages_semaphore); break; }
            const uint32_t take = (uint32_t)min<size_t>(room, m.count);
            /* a run that continues the previous entry (same rectangle, next sample) extends it */
            if (!batch.empty() && batch.back().x == m.desc.start_x && batch.back().y == m.desc.start_y &&
                batch.back().w == m.desc.width && batch.back().h == m.desc.height &&
                batch.back().first_sample + batch.back().n_samples == m.sample)
                batch.back().n_samples += take;
            else
                batch.push_back({m.desc.start_x, m.desc.start_y, m.desc.width, m.desc.height, m.sample, take});
            taken += take;
            paths += per_pass * take;
            if (take < m.count) { /* split: the remainder keeps this entry and its wake-up */
                m.sample += take; m.count -= take;
                sem_post(&eng->messages_semaphore);
                break;
            }
            eng->messages.pop_front();
            finished++;
            if (paths >= share || sem_trywait(&eng->messages_semaphore) != 0) break;
        }
        pthread_mutex_unlock(&eng->messages_lock);
        if (die) break;
        if (batch.empty()) continue;

        self->is_running = true;
        self->currently_processing.start_x = batch[0].x; self->currently_processing.start_y = batch[0].y;
        self->currently_processing.width = batch[0].w; self->currently_processing.height = batch[0].h;
        pthread_mutex_lock(&self->integrator_lock);
        Integrator* in = self->integrator;
        double t0 = now_seconds();
        if (in->film) {
            pt_camera cam;
            fill_camera(cam, eng->scene); /* camera and lights are re-read for every batch */
            fill_lights(lights, eng->scene);
            pt_render_stats st;
            eng->scene->tree->lock_device(in->device);
            int rc = pt_render_passes(in->film, &cam, lights.data(), (int)lights.size(), batch.data(), (int)batch.size(), eng->seed, &st);
            eng->scene->tree->unlock_device(in->device);
            if (rc == PT_OK) {
                __sync_fetch_and_add(&rays_cast, (long long)(st.closest_rays + st.shadow_rays));
                in->passes += (int)taken;
            } else {
                /* the passes still count as completed below (sync() and the progress bar must end),
                 * but the failure is on record and they are not in Integrator::passes */
                string why = pt_last_error();
                cerr << "render worker (GPU " << in->device << "): " << why << endl;
                pthread_mutex_lock(&eng->master_lock);
                eng->render_errors++;
                eng->last_error = why;
                pthread_mutex_unlock(&eng->master_lock);
            }
        } else {
            pthread_mutex_lock(&eng->master_lock);
            eng->render_errors++;
            eng->last_error = "worker has no film on its GPU";
            pthread_mutex_unlock(&eng->master_lock);
        }
        in->last_pass_seconds = now_seconds() - t0;
        pthread_mutex_unlock(&self->integrator_lock);
        self->is_running = false;

        pthread_mutex_lock(&eng->master_lock);
        eng->total_passes_completed += (int)taken;
        pthread_cond_broadcast(&eng->passes_cond);
        pthread_mutex_unlock(&eng->master_lock);
        for (size_t i = 0; i < finished; i++) sem_post(&eng->passes_semaphore);
    }
    return nullptr;
}

RenderEngine::RenderEngine(int _width, int _height, Scene* _scene) : width(_width), height(_height), scene(_scene), seed(1) {
    sem_init(&passes_semaphore, 0, 0);
    pthread_cond_init(&passes_cond, NULL);
    pthread_mutex_init(&master_lock, NULL);
    sem_init(&messages_semaphore, 0, 0);
    pthread_mutex_init(&messages_lock, NULL);
    master_canvas = new Canvas(width, height);
    tile_width = width;
    tile_height = height;
    total_passes_issued = 0;
    total_passes_completed = 0;
    semaphore_passes_pending = 0;
    sample_base = 0;
    render_errors = 0;
    next_global_sample = 0;
    mixed_rectangles = false;
    int n = get_optimal_thread_count();
    /* PT_ENGINE_WORKERS=k: k workers dealt to the GPUs in turn (more than one per GPU when k exceeds
     * their number).  Renders no faster -- workers of one GPU take turns on it -- but runs the
     * several-worker machinery (shared queue, split runs, one film per worker, their sum) on a
     * machine with a single GPU. */
    int k = n;
    if (const char* e = getenv("PT_ENGINE_WORKERS")) { int v = atoi(e); if (v > 0 && n > 0) k = min(v, 64); }
    for (int i = 0; i < k; i++) workers.push_back(new RenderThread(this, i % n));
    if (workers.empty()) cerr << "RenderEngine: no GPU available (" << pt_last_error() << ")" << endl;
}

RenderEngine::~RenderEngine() {
    sync();
    for (auto worker : workers) worker->send_message(RenderMessage({true, PassDescriptor(), 0, 0}));
    for (auto worker : workers) pthread_join(worker->thread, nullptr);
    for (auto worker : workers) delete worker;
    sem_destroy(&passes_semaphore);
    pthread_cond_destroy(&passes_cond);
    pthread_mutex_destroy(&master_lock);
    sem_destroy(&messages_semaphore);
    pthread_mutex_destroy(&messages_lock);
    delete master_canvas;
}

void RenderEngine::issue_pass_desc(PassDescriptor desc) { issue_pass_run(desc, 1); }

void RenderEngine::issue_pass_run(PassDescriptor desc, int count, bool queue_locked) {
    if (workers.empty() || count <= 0) return;
    PassDescriptor key = desc;
    key.clamp_bounds(width, height);
    /* The random stream is keyed by (pixel, sample index), so a pixel must never see an index
     * twice.  While every rectangle issued so far is either identical to or disjoint from every
     * other (a fixed tiling, the normal case), each rectangle counts its own passes 0, 1, 2, ...:
     * every pixel then gets consecutive indices.  The first rectangle that overlaps a different
     * one switches the engine to ONE counter for all passes, started above every index handed
     * out before. */
    auto rect = make_pair(make_pair(key.start_x, key.start_y), make_pair(key.width, key.height));
    auto found = samples_issued.find(rect);
    if (found == samples_issued.end()) {
        if (!mixed_rectangles)
            for (const auto& kv : samples_issued) {
                int ox = kv.first.first.first, oy = kv.first.first.second, ow = kv.first.second.first, oh = kv.first.second.second;
                if (key.start_x < ox + ow && ox < key.start_x + key.width && key.start_y < oy + oh && oy < key.start_y + key.height) { mixed_rectangles = true; break; }
            }
        found = samples_issued.insert(make_pair(rect, 0u)).first;
    }
    uint32_t sample;
    if (mixed_rectangles) {
        for (const auto& kv : samples_issued) next_global_sample = max(next_global_sample, kv.second);
        sample = sample_base + next_global_sample;
        next_global_sample += (uint32_t)count;
        found->second = next_global_sample;
    } else {
        sample = sample_base + found->second;
        found->second += (uint32_t)count;
    }
    int worker = total_passes_issued % (int)workers.size();
    pthread_mutex_lock(&master_lock);
    total_passes_issued += count;
    pthread_mutex_unlock(&master_lock);
    workers[(size_t)worker]->send_message(RenderMessage({false, desc, sample, (uint32_t)count}), queue_locked);
    semaphore_passes_pending++;
}

void RenderEngine::perform_full_pass() {
    for (int next_y = 0; next_y < height; next_y += tile_height)
        for (int next_x = 0; next_x < width; next_x += tile_width)
            issue_pass_desc(PassDescriptor(next_x, next_y, tile_width, tile_height));
}

static Real global_tile_center_x, global_tile_center_y;
static inline Real square(Real x) { return x * x; }
static bool tile_compare(const pair<int, int>& a, const pair<int, int>& b) {
    Real a_distance = square(a.first - global_tile_center_x) + square(a.second - global_tile_center_y);
    Real b_distance = square(b.first - global_tile_center_x) + square(b.second - global_tile_center_y);
    return a_distance < b_distance;
}

void RenderEngine::perform_full_passes(int pass_count) { /* integrator.cpp:428-449: centre-out order */
    /* The reference lists every tile pass_count times and sorts the list stably by distance from
     * the centre, which leaves each tile's passes next to one another: the same order as sorting
     * the tiles and issuing each one's passes as a run. */
    vector<pair<int, int>> tile_spots;
    for (int next_y = 0; next_y < height; next_y += tile_height)
        for (int next_x = 0; next_x < width; next_x += tile_width) tile_spots.push_back(pair<int, int>(next_x, next_y));
    global_tile_center_x = (width - tile_width) / 2.0;
    global_tile_center_y = (height - tile_height) / 2.0;
    stable_sort(tile_spots.begin(), tile_spots.end(), tile_compare);
    /* the queue is held while the runs go in, so that the first worker to wake sees all of them and
     * sizes its first batch by the whole job instead of by the few entries queued so far */
    pthread_mutex_lock(&messages_lock);
    for (auto spot : tile_spots) issue_pass_run(PassDescriptor(spot.first, spot.second, tile_width, tile_height), pass_count, true);
    pthread_mutex_unlock(&messages_lock);
}

/* Waits until every pass issued so far has been rendered.  Completion is counted in passes, under
 * master_lock: a run of passes may be rendered in parts by several workers, so "its last part has
 * been taken" would not mean "all of it is done". */
void RenderEngine::sync() {
    pthread_mutex_lock(&master_lock);
    while (total_passes_completed < total_passes_issued && !workers.empty()) pthread_cond_wait(&passes_cond, &master_lock);
    pthread_mutex_unlock(&master_lock);
    while (sem_trywait(&passes_semaphore) == 0) {} /* kept for callers that look at it; nothing waits on it */
    semaphore_passes_pending = 0;
}

/* Drops every pass that has not started yet and stops the workers; passes already on a GPU
 * finish first.  No more passes may be issued afterwards. */
void RenderEngine::kill_workers() {
    pthread_mutex_lock(&messages_lock);
    int dropped = 0;
    for (const RenderMessage& m : messages) if (!m.do_die) dropped += (int)m.count;
    messages.clear();
    pthread_mutex_unlock(&messages_lock);
    pthread_mutex_lock(&master_lock);
    total_passes_issued -= dropped; /* they will never complete */
    pthread_cond_broadcast(&passes_cond);
    pthread_mutex_unlock(&master_lock);
    for (auto worker : workers) worker->kill_immediately();
    for (auto worker : workers) pthread_join(worker->thread, nullptr);
    for (auto worker : workers) delete worker;
    workers.clear();
    semaphore_passes_pending = 0;
}

/* Sums the workers' films into master_canvas (Canvas::add_from per worker, as the reference
 * does) after copying each GPU's film to the host.  May be called while workers run: each
 * film is read between two batches. */
int RenderEngine::rebuild_master_canvas() {
    master_canvas->zero();
    int total_passes = 0;
    /* Several GPUs: one RCCL sum-reduce of the films on the device side (pt_films_reduce), the
     * workers' own canvases are then not refreshed.  PT_FILMS_REDUCE=host keeps the host-side
     * sum below, which is also the fallback if the collective cannot be set up. */
    const char* how = getenv("PT_FILMS_REDUCE");
    const bool forced_rccl = how && strcmp(how, "rccl") == 0;
    bool one_film_per_gpu = true; /* the collective takes one film per GPU */
    for (size_t a = 0; a < workers.size(); a++)
        for (size_t b = 0; b < a; b++)
            if (workers[a]->integrator->device == workers[b]->integrator->device) one_film_per_gpu = false;
    const bool want_rccl = how ? forced_rccl : (workers.size() > 1 && one_film_per_gpu);
    last_reduce_how = "host";
    if (forced_rccl && !one_film_per_gpu) {
        /* refused before any lock is taken: the collective sums one film per GPU */
        last_error = "device-side film reduce failed: one film per GPU required (several workers share a GPU)";
        cerr << "rebuild_master_canvas: " << last_error << endl;
        return -1;
    }
    if (want_rccl && !workers.empty()) {
        vector<pt_film*> films;
        for (auto worker : workers) {
            pthread_mutex_lock(&worker->integrator_lock);
            if (worker->integrator->film) films.push_back(worker->integrator->film);
            total_passes += worker->integrator->passes;
        }
        int rc = PT_ERR_DEVICE;
        string why = "a worker has no film";
        if (films.size() == workers.size()) {
            vector<float> rgb((size_t)master_canvas->size * 3);
            /* every distinct GPU once, in ascending order (the per-GPU locks are not recursive) */
            vector<int> devices;
            for (auto worker : workers) devices.push_back(worker->integrator->device);
            sort(devices.begin(), devices.end());
            devices.erase(unique(devices.begin(), devices.end()), devices.end());
            for (int d : devices) scene->tree->lock_device(d);
            rc = pt_films_reduce(films.data(), (int)films.size(), rgb.data(), master_canvas->per_pixel_passes);
            if (rc != PT_OK) why = pt_last_error();
            for (size_t k = devices.size(); k-- > 0;) scene->tree->unlock_device(devices[k]);
            if (rc == PT_OK)
                for (int i = 0; i < master_canvas->size; i++)
                    master_canvas->pixels[i] = Vec(rgb[3 * (size_t)i], rgb[3 * (size_t)i + 1], rgb[3 * (size_t)i + 2]);
        }
        for (auto worker : workers) pthread_mutex_unlock(&worker->integrator_lock);
        if (rc == PT_OK) { last_reduce_how = "rccl"; return total_passes; }
        last_error = "device-side film reduce failed: " + why;
        cerr << "rebuild_master_canvas: " << last_error << (forced_rccl ? "" : "; summing on the host") << endl;
        master_canvas->zero();
        if (forced_rccl) return -1; /* the collective was asked for by name: no silent substitute */
        last_reduce_how = "host after rccl failed";
        total_passes = 0;
    }
    for (auto worker : workers) {
        pthread_mutex_lock(&worker->integrator_lock);
        worker->integrator->download();
        master_canvas->add_from(worker->integrator->canvas);
        total_passes += worker->integrator->passes;
        pthread_mutex_unlock(&worker->integrator_lock);
    }
    return total_passes;
}

void RenderEngine::zero() {
    sync();
    total_passes_issued = 0;
    total_passes_completed = 0;
    samples_issued.clear(); /* a zeroed canvas starts its sample stream again: frames are reproducible */
    sample_base = 0;
    next_global_sample = 0;
    mixed_rectangles = false;
    for (auto worker : workers) {
        if (worker->integrator->film) pt_film_zero(worker->integrator->film);
        worker->integrator->canvas->zero();
    }
}

int RenderEngine::resume_from(Canvas* saved) {
    if (!saved || saved->width != width || saved->height != height || workers.empty()) return -1;
    zero();
    Integrator* in = workers[0]->integrator;
    if (!in->film) return -1;
    vector<float> rgb((size_t)saved->size * 3);
    vector<int32_t> passes((size_t)saved->size);
    int most = 0;
    for (int i = 0; i < saved->size; i++) {
        for (int c = 0; c < 3; c++) rgb[(size_t)i * 3 + c] = saved->pixels[i](c);
        passes[(size_t)i] = saved->per_pixel_passes[i];
        most = max(most, saved->per_pixel_passes[i]);
    }
    if (pt_film_add(in->film, rgb.data(), passes.data()) != PT_OK) return -1;
    sample_base = (uint32_t)most;
    return 0;
}

/* ------------------------------------------------------------ progress bar */

ProgressBar::ProgressBar(RenderEngine* _engine) : engine(_engine) {
    start = new struct timeval;
    gettimeofday(start, NULL);
}
ProgressBar::~ProgressBar() { delete start; }
bool ProgressBar::init() { return true; }

void ProgressBar::main_loop() { /* visualizer.cpp:152-179 */
    int completed, issued;
    double elapsed, last_print = -1.0;
    do {
        struct timeval stop, result;
        gettimeofday(&stop, NULL);
        timersub(&stop, start, &result);
        elapsed = result.tv_sec + result.tv_usec * 1e-6;
        pthread_mutex_lock(&engine->master_lock);
        completed = engine->total_passes_completed;
        pthread_mutex_unlock(&engine->master_lock);
        issued = engine->total_passes_issued;
        /* the line is redrawn every 50 ms as in the reference, but completion is looked for every
         * 2 ms: a render that takes a second should not wait 50 ms to be noticed */
        if (elapsed - last_print >= 0.05 || completed >= issued || engine->workers.empty()) {
            last_print = elapsed;
            double completion = issued ? completed / (double)issued : 1.0;
            double total_time = elapsed / real_max(1e-4, completion);
            double remaining = total_time - elapsed;
            string str_elapsed = format_seconds_as_hms(elapsed, 7);
            string str_remaining = format_seconds_as_hms(remaining, 7);
            string str_total_time = format_seconds_as_hms(total_time, 7);
            printf("\r[\033[93m%6.2f%%\033[0m] \033[94mElapsed:\033[0m %s   \033[94mRemaining:\033[0m %s   \033[94mTotal:\033[0m %s   \033[94mTile samples:\033[0m %i/%i",
                   100.0 * completion, str_elapsed.c_str(), str_remaining.c_str(), str_total_time.c_str(), completed, issued);
            fflush(stdout);
        }
        if (completed < issued) usleep(engine->workers.empty() ? 0 : 2000);
        if (engine->workers.empty()) break;
    } while (completed < issued);
    double performance = rays_cast / real_max(1e-9, elapsed);
    printf(" \033[92mDone!\033[0m Perf: %.1fMr/s\n", performance * 1e-6);
}
