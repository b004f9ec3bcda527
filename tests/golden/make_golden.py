"""Regenerates tests/golden/standin_level3.npz from the CPU oracle.

The reference holds no golden vectors for its ray-casting path and cannot be compiled in
this image, so these vectors come from the oracle (oracle/pt_oracle.cpp) and pin it against
regressions; they do not pin it against the reference ("parity unpinned").
Run: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orc  # noqa: E402
import path_tracing_amd as pta  # noqa: E402
from conftest import all_ray_sets  # noqa: E402

tris = pta.meshgen.standin_scene(3)
s = orc.Scene(tris)
sets = all_ray_sets(tris, 4000)
o = np.concatenate([v[0][:4000] for v in sets.values()])
d = np.concatenate([v[1][:4000] for v in sets.values()])
idx, t, u, v = s.raycast(o, d)
cam = orc.cli_camera(dof_aperture=0.1, dof_distance=4.5)
rgb, passes, counts = s.render(cam, orc.cli_lights(), 64, 48, None, 0, 2, seed=1)
np.savez_compressed(os.path.join(HERE, "standin_level3.npz"), tris=tris, tree_hash=np.uint64(s.tree_hash()),
                    origins=o, dirs=d, idx=idx, t=t, u=u, v=v, render_rgb=rgb, render_counts=counts)
print("rays", len(o), "hit fraction", float((idx >= 0).mean()), "render counts", counts.tolist())
