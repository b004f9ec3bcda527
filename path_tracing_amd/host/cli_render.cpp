/*
 * cli_render -- command-line renderer with the reference's flags and defaults
 * (reference cli_render.cpp:16-35), rendering on the GPUs through pt_render.h.
 *
 * The reference parses with Boost.program_options; this parser accepts the same spellings
 * (--name value, --name=value, one positional STL path) without that dependency.  Added:
 * --gpus N (same as --threads: how many GPUs to use, 0 = all), --seed N, and --checkpoint FILE /
 * --resume FILE (stop a long render and continue it later from its accumulators).  --display
 * is accepted and ignored (there is no SDL preview).
 */
#include <sys/time.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "pt_render.h"

using namespace std;

struct Option { const char* name; const char* def; const char* help; bool flag; };

/* PT_CLI_TIMING=1: wall-clock seconds of each stage on stderr (where an end-to-end run spends its time). */
static double wall_now() { struct timeval tv; gettimeofday(&tv, NULL); return tv.tv_sec + tv.tv_usec * 1e-6; }
static double stage_t0 = 0;
static void stage(const char* name) {
    if (!getenv("PT_CLI_TIMING")) return;
    double t = wall_now();
    if (stage_t0 > 0) fprintf(stderr, "[timing] %-28s %.3f s\n", name, t - stage_t0);
    else fprintf(stderr, "[timing] main() entered at %.3f\n", t);
    stage_t0 = t;
}

static const Option OPTIONS[] = {
    {"help", nullptr, "Produce help message.", true},
    {"stl", nullptr, "Input STL file.", false},
    {"output", "output.png", "Output PNG file.", false},
    {"samples", "10", "Number of samples.", false},
    {"width", "1920", "Width of rendered image.", false},
    {"height", "1080", "Height of rendered image.", false},
    {"display", nullptr, "Display render progress graphically (ignored: no SDL preview).", true},
    {"progressive", nullptr, "Render passes progressively.", false},
    {"threads", "0", "Number of GPUs. (0 for all)", false},
    {"gpus", nullptr, "Same as --threads.", false},
    {"seed", "1", "Seed of the random stream.", false},
    {"lambertian", nullptr, "Bounce diffusely (reference integrator.cpp:79-84) instead of mirror-like.", true},
    {"bounded-shadows", nullptr, "Let only geometry nearer than a light hide it (the reference does not bound shadow rays).", true},
    {"checkpoint", nullptr, "Also write the accumulators (energy sums, pass counts) to this file.", false},
    {"resume", nullptr, "Continue from accumulators written by --checkpoint (same size, same seed).", false},
    {"angle", "1", "Camera angle.", false},
    {"camera-altitude", "0.2", "Camera altitude.", false},
    {"camera-distance", "5", "Camera distance from (origin + camera_altitude * zhat).", false},
    {"camera-z-facing-offset", "0", "This option shouldn't exist.", false},
    {"dof-aperture", "0", "Depth of field Gaussian aperture standard deviation. Use 0.0 to disable DoF.", false},
    {"dof-distance", "1", "Distance to the plane of focus.", false},
    {"tile-width", "64", "Width of a rendering tile in pixels.", false},
    {"tile-height", "64", "Height of a rendering tile in pixels.", false},
};

static void print_help() {
    cout << "Allowed options:" << "\n";
    for (const Option& o : OPTIONS) {
        string left = string("  --") + o.name + (o.flag ? "" : " arg");
        if (o.def) left += string(" (=") + o.def + ")";
        while (left.size() < 40) left += " ";
        cout << left << o.help << "\n";
    }
    cout << "\n";
}

int main(int argc, char** argv) {
    map<string, string> vm;
    vector<string> inputs;
    for (const Option& o : OPTIONS) if (o.def) vm[o.name] = o.def;
    for (int i = 1; i < argc; i++) {
        string a = argv[i];
        if (a.rfind("--", 0) != 0) { inputs.push_back(a); continue; }
        string name = a.substr(2), value;
        bool has_value = false;
        size_t eq = name.find('=');
        if (eq != string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); has_value = true; }
        const Option* opt = nullptr;
        for (const Option& o : OPTIONS) if (name == o.name) opt = &o;
        if (!opt) { cout << "unrecognised option '--" << name << "'" << endl; return 1; }
        if (opt->flag) { vm[name] = "1"; continue; }
        if (!has_value) {
            if (i + 1 >= argc) { cout << "the required argument for option '--" << name << "' is missing" << endl; return 1; }
            value = argv[++i];
        }
        if (name == "stl") inputs.push_back(value); else vm[name] = value;
    }
    if (vm.count("help")) { print_help(); return 1; }
    if (inputs.empty()) {
        cout << "No input STL specified." << endl;
        cout << "Example usage: cli_render input.stl --output output.png" << endl;
        return 1;
    }
    if (inputs.size() > 1) {
        cout << "Too many inputs specified -- currently only one input STL is supported." << endl;
        return 1;
    }
    if (vm.count("gpus")) vm["threads"] = vm["gpus"];
    string path = inputs[0];

    cout << "input        = " << path << endl;
    for (string key : {"output", "samples", "width", "height", "threads", "angle", "camera-altitude", "dof-aperture", "dof-distance", "tile-width", "tile-height"}) {
        if (key == "dof-distance" and atof(vm["dof-aperture"].c_str()) == 0) continue;
        cout << key;
        for (int i = (int)key.size(); i < 13; i++) cout << " ";
        cout << "= " << vm[key] << endl;
    }

    override_thread_count(atoi(vm["threads"].c_str()));

    stage("start");
    auto scene = new Scene(path);
    stage("scene (read, normals, tree)");
    if (scene->mesh == nullptr) return 1; /* the reference carries on with an unusable scene here */
    scene->lights->push_back(Light({Vec(0, 0, 3), 9.0 * Vec(0.8, 0.5, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, 2, 4), 9.0 * Vec(0.25, 0.8, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, -2, 4), 9.0 * Vec(0.25, 0.25, 0.8)}));
    scene->camera_image_plane_width = 0.5 * 1.5;
    Real angle = atof(vm["angle"].c_str());
    scene->main_camera.origin = -atof(vm["camera-distance"].c_str()) * Vec(cos(angle), sin(angle), 0.0);
    scene->main_camera.direction = -scene->main_camera.origin;
    scene->main_camera.direction(2) += atof(vm["camera-z-facing-offset"].c_str());
    scene->main_camera.direction.normalize();
    scene->main_camera.origin += Vec(0.0, 0.0, atof(vm["camera-altitude"].c_str()));
    scene->plane_of_focus_distance = atof(vm["dof-distance"].c_str());
    scene->dof_dispersion = atof(vm["dof-aperture"].c_str());
    scene->lambertian_bounces = vm.count("lambertian") != 0;
    scene->bounded_shadow_rays = vm.count("bounded-shadows") != 0;

    auto engine = new RenderEngine(atoi(vm["width"].c_str()), atoi(vm["height"].c_str()), scene);
    if (engine->workers.empty()) { cout << "No GPU available." << endl; return 1; }
    stage("engine (GPU init, upload)");
    engine->seed = strtoull(vm["seed"].c_str(), nullptr, 10);
    engine->tile_width = atoi(vm["tile-width"].c_str());
    engine->tile_height = atoi(vm["tile-height"].c_str());

    if (vm.count("resume")) {
        Canvas saved(engine->width, engine->height);
        if (saved.load_accumulators(vm["resume"]) != 0 || engine->resume_from(&saved) != 0) {
            cout << "Couldn't resume from: " << vm["resume"] << endl;
            return 1;
        }
        cout << "Resumed after " << engine->sample_base << " samples from: " << vm["resume"] << endl;
    }

    ProgressReporter* pr = new ProgressBar(engine);
    pr->init();
    int samples_count = atoi(vm["samples"].c_str());
    if (vm.count("progressive")) {
        int progressive_count = max(1, atoi(vm["progressive"].c_str()));
        for (int i = 0; i < samples_count / progressive_count; i++) engine->perform_full_passes(progressive_count);
        engine->perform_full_passes(samples_count % progressive_count);
    } else
        engine->perform_full_passes(samples_count);
    pr->main_loop();
    delete pr;
    engine->sync();
    stage("render");
    if (engine->render_errors) { cout << "Render failed: " << engine->last_error << endl; return 1; }
    if (engine->rebuild_master_canvas() < 0) { cout << "Couldn't sum the films: " << engine->last_error << endl; return 1; }
    if (getenv("PT_FILMS_REDUCE")) cerr << "films summed by: " << engine->last_reduce_how << endl;
    stage("rebuild_master_canvas");
    string output_path = vm["output"];
    if (engine->master_canvas->save(output_path) != 0) { cout << "Couldn't write: " << output_path << endl; return 1; }
    cout << "Wrote to: " << output_path << endl;
    stage("save PNG");
    if (vm.count("checkpoint")) {
        if (engine->master_canvas->save_accumulators(vm["checkpoint"]) != 0) { cout << "Couldn't write: " << vm["checkpoint"] << endl; return 1; }
        cout << "Accumulators written to: " << vm["checkpoint"] << endl;
    }
    /* Like the reference (cli_render.cpp:127-128) the engine and the scene are not torn down piece
     * by piece: the picture is on disk, and handing ~150 GB of render scratch back allocation by
     * allocation costs 0.6 s that process exit does not.  PT_CLI_TEARDOWN=1 runs the destructors. */
    if (getenv("PT_CLI_TEARDOWN")) {
        delete engine;
        delete scene;
        stage("teardown");
        return 0;
    }
    if (getenv("PT_CLI_TIMING")) fprintf(stderr, "[timing] leaving main() at %.3f\n", wall_now());
    cout.flush();
    fflush(stdout);
    fflush(stderr);
    _exit(0);
}
