"""Animated scenes: GPUKernel::rotatePrimitives + compactBoxes(false) per frame (reference:
apps/scenes/science/MoleculeScene.cpp:75-81, GPUKernel.cpp:1378-1460 and :1151-1281).

With the HIP engine the rotation runs on the resident scene (solr_hip_rotate_primitives) and the host
scene store follows lazily.  The bar: after any number of such steps the device holds, bit for bit, the
primitives and the node list the host route (rotate on the host, flatten, upload) produces; the frames
rendered on the way match the oracle on the host-route scene; and the host store, once it catches up,
equals the host route's."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes_extra  # noqa: E402
from helpers import assert_parity, compare_frames, gpu_frame, oracle_frame  # noqa: E402

pytestmark = pytest.mark.gpu

STEPS = [((0.0, 0.0, 0.0), (0.0, 0.02, 0.0)), ((10.0, -20.0, 30.0), (0.11, -0.07, 0.05)),
         ((0.0, 0.0, 0.0), (0.0, 0.0, 0.3)), ((-500.0, 100.0, 0.0), (-0.2, 0.4, 0.0)),
         ((0.0, 0.0, 0.0), (1.5, 0.02, -0.6))]

SCENES = [
    ("molecule", dict(atoms=400, width=96, height=64, iterations=2)),
    ("height_field", dict(n=20, width=96, height=64)),
    ("cornell", dict(width=96, height=64, iterations=3)),
    ("primitives_mix", dict()),
    ("sticks", dict()),
]


def _build(solr, spec, engine):
    name, kw = spec
    k = solr.Kernel(engine=engine, deterministic_seed=1)
    (getattr(solr.scenes, name, None) or getattr(scenes_extra, name))(k, **kw)
    return k


def _node_rows(flat):
    """the reference's flattened boxes in the engine's node-record layout (scene_layout.h)"""
    b = flat.boxes
    rows = np.zeros((len(b), 2, 4), np.float32)
    rows[:, 0, :3] = b["min"]
    rows[:, 0, 3] = b["max"][:, 2]
    rows[:, 1, :2] = b["max"][:, :2]
    rows[:, 1, 2] = b["nbPrimitives"].view(np.float32)
    rows[:, 1, 3] = b["indexForNextBox"][:, 0].copy().view(np.float32)
    return rows


def _same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def _same_records(a, b):
    """field by field: the structs carry padding that no one initialises"""
    return len(a) == len(b) and all(_same_bits(a[f], b[f]) for f in a.dtype.names)


def _primitive_geometry(rows):
    """p0 p1 p2 n0 n1 n2 of the device's primitive records (rows 0, 2, 3, 4, 5, 6; .xyz)"""
    return rows[:, [0, 2, 3, 4, 5, 6], :3]


@pytest.mark.parametrize("grouping", [True, False], ids=["grouped", "ungrouped"])
@pytest.mark.parametrize("spec", SCENES, ids=[s[0] for s in SCENES])
def test_device_rotation_equals_host_rotation(solr, oracle, spec, grouping):
    hip = solr.hip_lib()
    hip.solr_hip_set_variant(0 if grouping else 5)      # 5: no grouping nodes, from the next upload on
    try:
        _device_rotation_equals_host_rotation(solr, oracle, spec, grouping)
    finally:
        hip.solr_hip_set_variant(0)


def _device_rotation_equals_host_rotation(solr, oracle, spec, grouping):
    hip = solr.hip_lib()
    k = _build(solr, spec, "hip")
    gpu_frame(k)                                   # uploads the scene
    order_free = hip.solr_hip_order_free_nodes()   # > 0 where the primary rays walk the order-free lists
    frames = []
    for n, (center, angles) in enumerate(STEPS):
        k.rotate_primitives(center, angles)
        assert k.pending_rotations() == n + 1, "the rotation took the host route"
        frames.append(gpu_frame(k))
    assert hip.solr_hip_device_rotations() == len(STEPS)
    assert hip.solr_hip_order_free_nodes() == order_free, "the order-free lists did not follow the rotations"
    nodes = k.device_nodes(exact=True)
    prims = k.device_primitives()
    walk = k.device_nodes(exact=False)
    assert k.pending_rotations() == len(STEPS)     # reading the device does not wake the host store
    caught_up = k.flat_scene()                     # this does
    assert k.pending_rotations() == 0
    pp2, ids2, rgb2 = gpu_frame(k)                 # fresh upload of the replayed host scene
    walk_fresh = k.device_nodes(exact=False)
    k.finalize()

    h = _build(solr, spec, "host-only")
    for n, (center, angles) in enumerate(STEPS):
        h.rotate_primitives(center, angles)
        assert h.pending_rotations() == 0
        if n in (0, len(STEPS) - 1):
            pp, ids, rgb = frames[n]
            flat = h.flat_scene()
            flat.randoms = caught_up.randoms       # a store that never rendered has not drawn its random buffer
            si, ppi, eye, direction, view_angles = h.frame_parameters()
            opp, oids, orgb, _, status = oracle.render(flat, si, ppi, eye, direction, view_angles)
            assert status == 0
            assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    flat = h.flat_scene()
    # the host store after its lazy replay == the host route
    assert _same_records(caught_up.boxes, flat.boxes)
    assert _same_records(caught_up.primitives, flat.primitives)
    # the resident scene == what the host route would upload
    assert nodes.shape[0] == len(flat.boxes)
    assert _same_bits(nodes, _node_rows(flat))
    fp = flat.primitives
    want = np.stack([fp["p0"], fp["p1"], fp["p2"], fp["n0"], fp["n1"], fp["n2"]], axis=1)
    assert _same_bits(_primitive_geometry(prims), want)
    # the engine's own walk-order list keeps the shape it was given at upload; refitted, every node still
    # holds its children (a fresh upload may group differently, the frames are the same either way)
    assert walk.shape[0] > 0 and np.isfinite(walk[:, 0, :3]).all()
    pp, ids, rgb = frames[-1]
    res = compare_frames(pp, ids, rgb, pp2, ids2, rgb2)
    assert res["ids_all_equal"] and res["max_ulp"] == 0 and res["rgb_max_diff"] == 0, res
    assert walk_fresh.shape[0] > 0
    if not grouping:
        # without grouping nodes the walk-order list is the reference's list minus its single-child chains:
        # the same nodes whether refitted in place or uploaded afresh, and then the same bits
        assert walk.shape == walk_fresh.shape and _same_bits(walk, walk_fresh)
    h.finalize()


def test_touching_the_scene_store_ends_the_fast_path(solr, oracle):
    k = _build(solr, SCENES[0], "hip")
    gpu_frame(k)
    k.rotate_primitives((0.0, 0.0, 0.0), (0.0, 0.1, 0.0))
    assert k.pending_rotations() == 1
    gpu_frame(k)
    # a change of the scene store: the next rotation has to see it, so it runs on the host and is uploaded
    k.L.SolR_SetPrimitiveMaterial(3, 5)
    assert k.pending_rotations() == 0
    k.rotate_primitives((0.0, 0.0, 0.0), (0.0, 0.1, 0.0))
    assert k.pending_rotations() == 0
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    # ... after which the scene is resident and unchanged again (reading the flattened arrays is no change)
    k.rotate_primitives((0.0, 0.0, 0.0), (0.0, 0.1, 0.0))
    assert k.pending_rotations() == 1
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)   # catches the host store up
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    k.finalize()


def test_rotation_with_frames_in_flight_and_materials_update(solr, oracle):
    """a retag of the primitives (materials change) rebuilds the arena from the host images: they must
    have been brought up to date from the device first"""
    hip = solr.hip_lib()
    k = _build(solr, SCENES[0], "hip")
    hip.solr_hip_set_frames_in_flight(2)
    try:
        gpu_frame(k)
        for _ in range(3):
            k.rotate_primitives((0.0, 0.0, 0.0), (0.03, 0.1, 0.0))
            gpu_frame(k)
        assert k.pending_rotations() == 3
        k.L.SolR_SetMaterial(5, 0.9, 0.1, 0.1, 0.0, 0.0, 0.0, 0, 0, 0, 0.0, 0.0, -1, -1, -1, -1, -1, -1, -1, 1.0, 200.0,
                             0.0, 0.0, 0.0, 0.0, 0)
        still_pending = k.pending_rotations()
        pp, ids, rgb = gpu_frame(k)
        opp, oids, orgb, _, status = oracle_frame(k, oracle)
        assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
        assert still_pending in (0, 3)
    finally:
        hip.solr_hip_set_frames_in_flight(1)
        k.finalize()


def test_exact_node_list_follows_when_a_frame_walks_it(solr, oracle):
    """rotations refit the walk-order list; the reference's own list is refitted when a frame walks it
    (variant 3: walk the list exactly as uploaded; the box-debug view does the same)"""
    hip = solr.hip_lib()
    k = _build(solr, SCENES[0], "hip")
    try:
        gpu_frame(k)
        for center, angles in STEPS[:3]:
            k.rotate_primitives(center, angles)
        fast = gpu_frame(k)
        hip.solr_hip_set_variant(3)
        exact = gpu_frame(k)
        assert k.pending_rotations() == 3
        res = compare_frames(*exact, *fast)
        assert res["ids_all_equal"] and res["max_ulp"] == 0 and res["rgb_max_diff"] == 0, res
        k.set_scene_info(renderBoxes=1)
        boxes = gpu_frame(k)
        opp, oids, orgb, _, status = oracle_frame(k, oracle)      # the host store catches up here
        assert_parity(compare_frames(*boxes, opp, oids, orgb))
    finally:
        hip.solr_hip_set_variant(0)
        k.finalize()


def test_rotation_requests_the_engine_cannot_serve_change_nothing(solr):
    import ctypes as C
    hip = solr.hip_lib()
    k = _build(solr, SCENES[0], "hip")
    try:
        gpu_frame(k)
        before = k.device_primitives().copy()
        f3 = C.c_float * 3
        zero, one = f3(0, 0, 0), f3(1, 1, 1)
        turn_c, turn_s = f3(1.0, float(np.cos(0.3)), 1.0), f3(0.0, float(np.sin(0.3)), 0.0)
        served = hip.solr_hip_device_rotations()
        assert hip.solr_hip_rotate_primitives(None, one, zero, 50000.0) == 0          # no centre
        assert hip.solr_hip_rotate_primitives(zero, turn_c, turn_s, 2.0e6) == 0       # seeds would not commute
        assert hip.solr_hip_rotate_primitives(zero, turn_c, turn_s, -1.0) == 0
        hip.solr_hip_set_movable(None, 0)                                             # flags for another scene size
        assert hip.solr_hip_rotate_primitives(zero, turn_c, turn_s, 50000.0) == 0
        assert hip.solr_hip_device_rotations() == served
        assert _same_bits(k.device_primitives(), before)
        k.check(0, "refused rotations are not errors")
        # too small a buffer for the read-back
        small = np.zeros((4, 4), np.float32)
        assert hip.solr_hip_read_primitives(small.ctypes.data, 4) == -1
        assert hip.solr_hip_read_nodes(1, small.ctypes.data, 4) == -1
        # the host store notices nothing of all this: its own rotation still goes through (by the host route,
        # the flags being gone) and the frame is right
        k.rotate_primitives((0.0, 0.0, 0.0), (0.0, 0.3, 0.0))
        assert k.pending_rotations() == 0
        gpu_frame(k)
        k.rotate_primitives((0.0, 0.0, 0.0), (0.0, 0.3, 0.0))     # uploaded again, flags and all
        assert k.pending_rotations() == 1
    finally:
        k.finalize()


@pytest.mark.parametrize("spec", [SCENES[0], SCENES[3]], ids=[SCENES[0][0], SCENES[3][0]])
def test_long_animation_is_fetched_not_replayed(solr, spec):
    """beyond a handful of pending rotations the host store takes the primitives from the device instead of
    replaying them one by one: same bits, constant cost"""
    k = _build(solr, spec, "hip")
    gpu_frame(k)
    n = 0
    for lap in range(5):
        for center, angles in STEPS:
            k.rotate_primitives(center, angles)
            n += 1
    assert k.pending_rotations() == n == 25
    fetched = k.flat_scene()
    assert k.pending_rotations() == 0
    k.finalize()
    h = _build(solr, spec, "host-only")
    for lap in range(5):
        for center, angles in STEPS:
            h.rotate_primitives(center, angles)
    eager = h.flat_scene()
    h.finalize()
    assert _same_records(fetched.boxes, eager.boxes) and _same_records(fetched.primitives, eager.primitives)


def _list_invariants(nodes):
    """nested skip pointers; every inner node holds the nodes of its subtree; returns the leaves' rows"""
    n = nodes.shape[0]
    count = nodes[:, 1, 2].view(np.int32)
    skip = nodes[:, 1, 3].view(np.int32)
    lo = nodes[:, 0, :3]
    hi = np.stack([nodes[:, 1, 0], nodes[:, 1, 1], nodes[:, 0, 3]], axis=1)
    assert (skip >= 1).all() and (np.arange(n) + skip <= n).all()
    stack = []
    for i in range(n):
        while stack and i >= stack[-1] + skip[stack[-1]]:
            stack.pop()
        if stack:
            p = stack[-1]
            assert i + skip[i] <= p + skip[p], "skip pointers are not nested"
            assert (lo[i] >= lo[p]).all() and (hi[i] <= hi[p]).all(), "node %d is not inside node %d" % (i, p)
        assert (count[i] > 0) == (skip[i] == 1) or count[i] == 0
        stack.append(i)
    return nodes[count > 0]


@pytest.mark.parametrize("spec", [s for s in SCENES if s[0] in ("molecule", "height_field", "cornell")], ids=lambda s: s[0])
def test_order_free_lists_are_eight_orders_of_one_hierarchy(solr, spec):
    """the eight order-free lists hold the same leaves - those of the reference's list, bit for bit - under nodes
    that hold their subtrees, before and after rotations on the device (refitted in place)"""
    k = _build(solr, spec, "hip")
    gpu_frame(k)
    hip = solr.hip_lib()
    assert hip.solr_hip_order_free_nodes() > 0
    for round_ in range(2):
        exact = k.device_nodes(exact=True)
        reference_leaves = exact[exact[:, 1, 2].view(np.int32) > 0]
        reference_leaves = reference_leaves[:, :, :].copy()
        reference_leaves[:, 1, 3] = np.int32(1).view(np.float32)
        want = np.sort(np.ascontiguousarray(reference_leaves).view(np.uint32).reshape(len(reference_leaves), -1), axis=0)
        sizes = set()
        for octant in range(8):
            nodes = k.device_nodes(order_free=octant)
            sizes.add(nodes.shape[0])
            leaves = _list_invariants(nodes)
            got = np.sort(np.ascontiguousarray(leaves).view(np.uint32).reshape(len(leaves), -1), axis=0)
            assert got.shape == want.shape and np.array_equal(got, want), "octant %d holds other leaves" % octant
        assert len(sizes) == 1
        for center, angles in STEPS[:2]:
            k.rotate_primitives(center, angles)
        gpu_frame(k)
    k.finalize()


def test_order_free_lists_built_between_rotations(solr, oracle):
    """the default timing: the lists are built before the second frame after an upload - here after the first
    rotation ran on the device, i.e. from host images that are fetched back first - and refitted by the next one;
    every frame is the one a fresh upload of the rotated scene renders"""
    hip = solr.hip_lib()
    saved = os.environ.pop("SOLR_HIP_FREE_AFTER", None)
    try:
        spec = SCENES[0]
        k = _build(solr, spec, "hip")
        gpu_frame(k)
        assert hip.solr_hip_order_free_nodes() == 0
        frames = []
        for n, (center, angles) in enumerate(STEPS[:3]):
            k.rotate_primitives(center, angles)
            assert k.pending_rotations() == n + 1
            frames.append(gpu_frame(k))
            assert hip.solr_hip_order_free_nodes() > 0
        k.flat_scene()                              # the host store replays the rotations
        assert k.pending_rotations() == 0
        fresh = gpu_frame(k)                        # a fresh upload of the rotated scene
        k.finalize()
        pp, ids, rgb = frames[-1]
        res = compare_frames(pp, ids, rgb, fresh[0], fresh[1], fresh[2])
        assert res["ids_all_equal"] and res["max_ulp"] == 0 and res["rgb_max_diff"] == 0, res
    finally:
        if saved is not None:
            os.environ["SOLR_HIP_FREE_AFTER"] = saved


def test_sorted_copies_of_the_order_free_lists_follow_a_rotation(solr, oracle):
    """A scene whose lists are long enough for the three-bank node loop (more than 1 024 nodes): the copies of the
    order-free lists with sorted bounds (rt_device.h SOLR_ORDER_SORTED / _REVERSED: the node loop without its min / max
    where a wave's rays share the list's octant) are made again after a rotation on the device refitted the lists.  After
    every rotation the frame is, bit for bit, the frame without them (variant 12) - and the frame a fresh upload of the
    rotated scene renders"""
    hip = solr.hip_lib()
    k = _build(solr, ("molecule", dict(atoms=2500, width=160, height=104, iterations=2)), "hip")
    try:
        gpu_frame(k)
        gpu_frame(k)                                 # (the order-free lists arrive with the second frame)
        assert hip.solr_hip_order_free_nodes() > 1024 // 8
        for n, (center, angles) in enumerate(STEPS[:3]):
            k.rotate_primitives(center, angles)
            hip.solr_hip_set_variant(0)
            sorted_frame = [np.array(a, copy=True) for a in gpu_frame(k)]
            hip.solr_hip_set_variant(12)
            plain = [np.array(a, copy=True) for a in gpu_frame(k)]
            hip.solr_hip_set_variant(0)
            assert _same_bits(sorted_frame[0], plain[0]) and np.array_equal(sorted_frame[1], plain[1]), n
            assert np.array_equal(sorted_frame[2], plain[2]), n
            assert hip.solr_hip_order_free_nodes() > 0
        k.flat_scene()                               # the host store replays the rotations
        fresh = gpu_frame(k)                         # a fresh upload of the rotated scene
        res = compare_frames(sorted_frame[0], sorted_frame[1], sorted_frame[2], fresh[0], fresh[1], fresh[2])
        assert res["ids_all_equal"] and res["max_ulp"] == 0 and res["rgb_max_diff"] == 0, res
    finally:
        hip.solr_hip_set_variant(0)
        k.finalize()
