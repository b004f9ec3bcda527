"""Latency of small render calls (one sample of a frame, one tile): what a progressive front end sees."""
import time, numpy as np, sys, os
sys.path.insert(0, os.getcwd())
import path_tracing_amd as pta
tris = pta.meshgen.standin_scene(8)
host = pta.HostScene(tris, threads=16)
gpu = pta.GpuScene(host, 0)
film = pta.Film(gpu, 1920, 1080)
cam = pta.cli_camera(dof_aperture=0.1, dof_distance=4.5); lights = pta.cli_lights()
for name, rect, spp in (("full frame, 1 spp", (0, 0, -1, -1), 1), ("full frame, 4 spp", (0, 0, -1, -1), 4), ("64x64 tile, 1 spp", (512, 512, 64, 64), 1), ("64x64 tile, 16 spp", (512, 512, 64, 64), 16)):
    for _ in range(3):
        film.render(cam, lights, [rect + (0, spp)], seed=1)
    gpu.sync()
    t0 = time.perf_counter(); n = 20
    for k in range(n):
        film.render(cam, lights, [rect + (k * spp, spp)], seed=1)
    gpu.sync()
    print("%-22s %.3f ms per call" % (name, (time.perf_counter() - t0) / n * 1e3))
