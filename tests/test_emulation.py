"""The per-lane device logic (path_tracing_amd/csrc/traverse_core.h), compiled for the CPU and
run one ray at a time over the flattened arrays, against the recursive oracle: bit-exact."""
import ctypes as C

import numpy as np
import pytest

from conftest import all_ray_sets, assert_hits_equal, load_emul, orc, pta


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def emul_closest(E, host, o, d):
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    n = len(o)
    idx = np.empty(n, np.int32)
    t = np.empty(n, np.float32)
    u = np.empty(n, np.float32)
    v = np.empty(n, np.float32)
    pos = np.empty(n, np.int32)
    flat = host.flat
    worst = E.emul_raycast_closest(C.byref(flat), n, _p(o), _p(d), _p(idx), _p(t), _p(u), _p(v), _p(pos))
    return (idx, t, u, v), worst, pos


@pytest.mark.parametrize("which", ["scene_tiny", "scene_small", "scene_medium"])
def test_lane_walk_bit_exact(which, request):
    sc = request.getfixturevalue(which)
    E = load_emul()
    for name, (o, d) in all_ray_sets(sc.tris).items():
        got, worst, pos = emul_closest(E, sc.host, o, d)
        assert 0 <= worst <= 3 * 20, "stack bound"
        assert_hits_equal(got, sc.oracle.raycast(o, d), f"{which}/{name}")
        # leaf-order position and mesh index agree through the permutation
        _, tris, _ = sc.host.flat_arrays()
        hit = pos >= 0
        assert np.array_equal(tris["mesh_index"][pos[hit]].astype(np.int32), got[0][hit])


def test_lane_shadow(scene_small, scene_medium):
    E = load_emul()
    for sc in (scene_small, scene_medium):
        for name, (o, d) in all_ray_sets(sc.tris, 6000).items():
            rng = np.random.default_rng(3)
            dist = np.where(rng.random(len(o)) < 0.5, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            vis = np.empty(len(o), np.uint8)
            flat = sc.host.flat
            o32 = np.ascontiguousarray(o, np.float32)
            d32 = np.ascontiguousarray(d, np.float32)
            worst = E.emul_raycast_shadow(C.byref(flat), len(o), _p(o32), _p(d32), _p(dist), _p(vis))
            assert worst >= 0
            assert np.array_equal(vis, sc.oracle.shadow_visible(o, d, dist)), name


def test_lane_shadow_on_quantised_nodes(scene_tiny, scene_small, scene_medium):
    """The any-hit walks over 16-bit interior boxes (only nodes whose children are both interior; a
    node with a leaf child keeps its exact record) give the oracle's visibility bit for bit, for the
    reference's test and for the bounded one; and the walk really uses them."""
    E = load_emul()
    for sc in (scene_tiny, scene_small, scene_medium):
        walked = np.zeros(4, np.uint64)
        for name, (o, d) in all_ray_sets(sc.tris, 6000).items():
            rng = np.random.default_rng(3)
            dist = np.where(rng.random(len(o)) < 0.5, rng.random(len(o)), 1.0 + 4.0 * rng.random(len(o))).astype(np.float32)
            vis = np.empty(len(o), np.uint8)
            flat = sc.host.flat
            o32 = np.ascontiguousarray(o, np.float32)
            d32 = np.ascontiguousarray(d, np.float32)
            rc = E.emul_raycast_shadow_q(C.byref(flat), len(o), _p(o32), _p(d32), _p(dist), _p(vis), _p(walked))
            if sc is scene_tiny and rc == -2:
                continue                      # a tree this small may have no node with two interior children
            assert rc == 0
            assert np.array_equal(vis, sc.oracle.shadow_visible(o, d, dist)), name
            idx, t, _, _ = sc.oracle.raycast(o, d)
            bound = np.where(idx >= 0, t * np.float32(1.5), np.float32(1.0)).astype(np.float32)
            bound[::3] = t[::3] * np.float32(0.5)
            assert E.emul_raycast_occluded_q(C.byref(flat), len(o), _p(o32), _p(d32), _p(bound), _p(vis)) == 0
            assert np.array_equal(vis, sc.oracle.shadow_visible(o, d, bound, bounded=True)), name + " bounded"
        if sc is not scene_tiny:
            assert walked[0] > 0 and walked[1] > walked[2] > 0      # mostly q-node records, exact ones near the leaves


def test_quantised_boxes_enclose_exact_ones(scene_medium):
    """Every q-node's decoded boxes contain the exact child boxes (the property the walk rests on),
    and are at most one grid step larger on every side."""
    nodes, _, _ = scene_medium.host.flat_arrays()
    flat = scene_medium.host.flat
    lo, hi = np.array(flat.root_min[:], np.float32), np.array(flat.root_max[:], np.float32)
    step = ((hi.astype(np.float64) - lo) / 65535.0).astype(np.float32)
    both = ((nodes["lo_ref"] & 0x80000000) == 0) & ((nodes["hi_ref"] & 0x80000000) == 0)
    assert 0 < both.sum() < len(nodes)
    for child in ("lo", "hi"):
        box = nodes[child][both]                                # [n, 3, 2]
        qmin = np.floor((box[:, :, 0].astype(np.float64) - lo) / step).clip(0, 65535)
        qmax = np.ceil((box[:, :, 1].astype(np.float64) - lo) / step).clip(0, 65535)
        dmin = (qmin * step.astype(np.float64) + lo)
        dmax = (qmax * step.astype(np.float64) + lo)
        assert (dmin <= box[:, :, 0] + 1e-6).all() and (dmax >= box[:, :, 1] - 1e-6).all()
        assert (box[:, :, 0] - dmin <= 1.01 * step).all() and (dmax - box[:, :, 1] <= 1.01 * step).all()


def test_oversized_leaf():
    """Leaves with more than 15 triangles keep their count in the first triangle record."""
    from conftest import oversized_leaf_mesh, rays_over_grid
    t = oversized_leaf_mesh(6)
    host = pta.HostScene(t)
    oracle = orc.Scene(t)
    assert host.tree_stats() == oracle.tree_stats() and host.tree_stats()["largest_leaf"] == 40
    nodes, tris, _ = host.flat_arrays()
    refs = np.concatenate([nodes["lo_ref"], nodes["hi_ref"]])
    leaf_refs = refs[(refs & 0x80000000) != 0]
    assert len(leaf_refs) == 36 and ((leaf_refs >> 26) & 0xF == 0).all()    # count field 0: read it from the record
    assert (tris["leaf_count"][leaf_refs & 0x03FFFFFF] == 40).all()
    E = load_emul()
    o, d = rays_over_grid(5000, 6 * 1.5)
    got, _, _ = emul_closest(E, host, o, d)
    assert_hits_equal(got, oracle.raycast(o, d))
    assert 0.2 < (got[0] >= 0).mean() < 1.0
