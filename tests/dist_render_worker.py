"""One rank of a sample-partitioned render with the HIP renderer (helper of
tests/test_gpu_distributed.py; started once per rank with RANK / WORLD_SIZE / MASTER_* set).

Every rank renders its own range of sample indices into its own film on the GPU it is given
(--device; on a one-GPU box all ranks share device 0), the films are summed to rank 0 with
path_tracing_amd.distributed.reduce_film -- over `gloo` on host copies here, the same call that
bench.py makes over RCCL -- and rank 0 saves the sum."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--level", type=int, default=3)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--total-samples", type=int, required=True)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import path_tracing_amd as pta
    from path_tracing_amd.distributed import partition_samples, reduce_film

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(a.device)
    host = pta.HostScene(pta.meshgen.standin_scene(a.level), threads=2)
    gpu = pta.GpuScene(host, a.device)
    film_t = torch.zeros(a.width * a.height * 4, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    film = pta.Film(gpu, a.width, a.height, external=film_t)
    first, count = partition_samples(a.total_samples, world, rank)
    tiles = [(x, y, 64, 64, first, count) for y in range(0, a.height, 64) for x in range(0, a.width, 64)]
    st = film.render(pta.cli_camera(dof_aperture=0.1, dof_distance=4.5), pta.cli_lights(), tiles, seed=a.seed)
    gpu.sync()
    torch.cuda.synchronize()
    host_film = film_t.cpu()
    reduce_film(host_film, dst=0)
    counts = torch.tensor([st.camera_samples, st.closest_rays, st.shadow_rays], dtype=torch.int64)
    dist.reduce(counts, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.savez(a.out, film=host_film.numpy(), counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
