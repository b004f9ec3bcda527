/*
 * animate -- multi-frame driver: one scene, resident on the GPUs, rendered from a camera and a
 * plane of focus that change per frame (reference animate.cpp:26-48: zero(), set camera and
 * focus distance, perform_full_passes, sync, rebuild_master_canvas, save frameNNNN.png).
 *
 * The reference hard-codes its numbers; they are the defaults here (1366x768, 100 frames of 100
 * samples, angle 1.0 with the orbit line commented out, focus 3.5 + frame/33, aperture 0.1,
 * output under massive/), each with a flag.  The geometry is uploaded once: per frame only the
 * camera block changes, which the workers re-read for every batch.
 */
#include <sys/stat.h>
#include <sys/time.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <map>
#include <string>

#include "pt_render.h"

using namespace std;

static double wall() {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + tv.tv_usec * 1e-6;
}

int main(int argc, char** argv) {
    map<string, string> vm = {
        {"frames", "100"}, {"samples", "100"}, {"width", "1366"}, {"height", "768"},
        {"angle", "1"}, {"angle-step", "0"}, {"focus-start", "3.5"}, {"focus-step", "0.030303030303030304"},
        {"dof-aperture", "0.1"}, {"camera-distance", "5"}, {"camera-altitude", "0.2"},
        {"tile-width", "64"}, {"tile-height", "64"}, {"output-dir", "massive"}, {"gpus", "0"}, {"seed", "1"},
    };
    string path;
    for (int i = 1; i < argc; i++) {
        string a = argv[i];
        if (a.rfind("--", 0) != 0) { path = a; continue; }
        string name = a.substr(2), value;
        size_t eq = name.find('=');
        if (eq != string::npos) { value = name.substr(eq + 1); name = name.substr(0, eq); }
        else if (i + 1 < argc) value = argv[++i];
        else { cout << "the required argument for option '--" << name << "' is missing" << endl; return 1; }
        if (name == "help" || !vm.count(name)) {
            cout << "usage: animate model.stl";
            for (auto& kv : vm) cout << " [--" << kv.first << " " << kv.second << "]";
            cout << endl;
            return 1;
        }
        vm[name] = value;
    }
    if (path.empty()) { cout << "No input STL specified." << endl; return 1; }

    override_thread_count(atoi(vm["gpus"].c_str()));
    auto scene = new Scene(path);
    if (scene->mesh == nullptr) return 1;
    scene->lights->push_back(Light({Vec(0, 0, 3), 9.0 * Vec(0.8, 0.5, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, 2, 4), 9.0 * Vec(0.25, 0.8, 0.25)}));
    scene->lights->push_back(Light({Vec(-2, -2, 4), 9.0 * Vec(0.25, 0.25, 0.8)}));
    scene->camera_image_plane_width = 0.5 * 1.5;
    scene->dof_dispersion = atof(vm["dof-aperture"].c_str());

    double t_start = wall();
    auto engine = new RenderEngine(atoi(vm["width"].c_str()), atoi(vm["height"].c_str()), scene);
    if (engine->workers.empty()) { cout << "No GPU available." << endl; return 1; }
    engine->seed = strtoull(vm["seed"].c_str(), nullptr, 10);
    engine->tile_width = atoi(vm["tile-width"].c_str());
    engine->tile_height = atoi(vm["tile-height"].c_str());
    mkdir(vm["output-dir"].c_str(), 0777);

    int frames = atoi(vm["frames"].c_str()), samples = atoi(vm["samples"].c_str());
    for (int frame = 0; frame < frames; frame++) {
        double t_frame = wall();
        engine->zero();
        Real angle = atof(vm["angle"].c_str()) + frame * atof(vm["angle-step"].c_str());
        scene->main_camera.origin = -atof(vm["camera-distance"].c_str()) * Vec(cos(angle), sin(angle), 0.0);
        scene->main_camera.direction = -scene->main_camera.origin;
        scene->main_camera.direction.normalize();
        scene->main_camera.origin += Vec(0.0, 0.0, atof(vm["camera-altitude"].c_str()));
        scene->plane_of_focus_distance = atof(vm["focus-start"].c_str()) + frame * atof(vm["focus-step"].c_str());

        engine->perform_full_passes(samples);
        engine->sync();
        if (engine->render_errors) { cout << "Render failed: " << engine->last_error << endl; return 1; }
        if (engine->rebuild_master_canvas() < 0) { cout << "Couldn't sum the films: " << engine->last_error << endl; return 1; }

        char output_path[4096];
        snprintf(output_path, sizeof output_path, "%s/frame%04i.png", vm["output-dir"].c_str(), frame);
        if (engine->master_canvas->save(output_path) != 0) { cout << "Couldn't write: " << output_path << endl; return 1; }
        cout << "frame " << frame << ": " << output_path << " in " << (wall() - t_frame) << " s" << endl;
    }
    delete engine;
    cout << "Total time to render: " << format_seconds_as_hms(wall() - t_start, 0) << endl;
    cout << "Rays cast: " << rays_cast << endl;
    delete scene;
    return 0;
}
