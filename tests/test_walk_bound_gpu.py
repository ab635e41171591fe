"""The walk's own ceiling (include/solr_hip.h solr_hip_walk_bound; SURVEY.md 8d's "measured empty-traversal upper
bound"): a frame that records its walks is a frame like any other - same bits - and the replay of those walks with the
node loop alone visits leaves, leaves nothing out, and takes less time than the frame."""
import ctypes as C

import numpy as np
import pytest

from helpers import device_frame

pytestmark = pytest.mark.gpu


def _args(solr, k):
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    return si, objects, ppi, eye, direction, angles


def _walk_bound(solr, args, repeats=5):
    si, objects, ppi, eye, direction, angles = args
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    ms, stats = (C.c_double * 3)(), (C.c_ulonglong * 4)()
    status = solr.hip_lib().solr_hip_walk_bound(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles),
                                                repeats, ms, stats)
    return status, list(ms), [int(x) for x in stats]


@pytest.mark.parametrize("scene", ["cornell", "height_field", "molecule"])
def test_a_recorded_frame_is_the_frame_and_its_walks_replay(solr, scene):
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    W, H = 320, 200
    if scene == "cornell":
        solr.scenes.cornell(k, width=W, height=H, iterations=3)
    elif scene == "height_field":
        solr.scenes.height_field(k, n=48, width=W, height=H)
    else:
        solr.scenes.molecule(k, atoms=1500, width=W, height=H)
    try:
        for _ in range(3):
            k.render()
        args = _args(solr, k)
        si, objects, ppi, eye, direction, angles = args
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        plain = device_frame(solr, si)
        status, ms, stats = _walk_bound(solr, args)
        k.check(status, "solr_hip_walk_bound")
        recorded = device_frame(solr, si)          # the frame solr_hip_walk_bound rendered while recording
        assert np.array_equal(recorded[0].view(np.uint32), plain[0].view(np.uint32))
        assert np.array_equal(recorded[1], plain[1]) and np.array_equal(recorded[2], plain[2])
        walks, left_out, entries, workgroups = stats
        tiles = ((W + 7) // 8) * ((H + 7) // 8)
        assert workgroups >= tiles
        assert walks >= 2 * tiles * 0.5          # at least a closest-hit and a shadow walk for the tiles that see the scene
        assert left_out == 0, "walks the replay left out (more than %d per wave, or not through the node loop)" % 16
        assert entries > W * H * 0.5             # the replay entered leaves
        assert 0.0 < ms[2] <= ms[1]
        # the frame after: the engine is back to plain frames
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        again = device_frame(solr, si)
        assert np.array_equal(again[0].view(np.uint32), plain[0].view(np.uint32))
        k.check(0, "after the replay")
    finally:
        k.finalize()


def _lists(solr):
    out = (C.c_ulonglong * 6)()
    solr.hip_lib().solr_hip_walk_bound_lists(out)
    return [int(x) for x in out]


@pytest.mark.parametrize("scene", ["height_field", "layered_terrain"])
def test_bounce_rays_of_a_long_triangle_list_walk_order_free_and_the_frame_is_the_reference_s(solr, oracle, scene):
    """closestHitWalk's checked form (rt_device.h): bounce rays are |direction| = 1 - rayEpsilon long, and for a ray
    shorter than 1 the reference's cut-off - slab parameter against closest DISTANCE - makes the winner depend on the
    order of the leaves whenever a second hit lies within 1 / |direction| of the first.  The long-list triangle kernels
    walk such rays order-free and repeat, in the reference's order, the lanes that have such a rival.  Here: (1) the
    bounce walks really do take the order-free lists (the walk record says which list every walk took); (2) the frame
    is the oracle's - ids, depth, RGB8 exact, colour <= 1 ULP - and bit for bit the frame without order-free lists
    (variant 6).  `layered_terrain` is the adversarial case: mirror sheets a few per cent of the bounce distance apart,
    at a grazing angle, so that nearly every bounce hit has a rival."""
    import scenes_extra as X
    from helpers import assert_parity_pinned, gpu_frame, oracle_frame
    hip = solr.hip_lib()
    W, H = 160, 120
    frames = []
    try:
        # (a small frame is as long as its longest tile: left to itself the engine keeps such rays in the reference's
        # order, solr_hip_set_short_ray_lists; the third frame is that - the same bits again)
        hip.solr_hip_set_tile_scheduling(0)       # raster order: one workgroup, one primary walk per tile
        for variant, short in ((0, 1), (6, 1), (0, 0)):
            hip.solr_hip_set_variant(variant)
            hip.solr_hip_set_short_ray_lists(short)
            k = solr.Kernel(engine="hip")
            if scene == "height_field":
                solr.scenes.height_field(k, n=48, width=W, height=H, iterations=3)
                k.set_camera((0.0, -1000.0, -15000.0), look_at=(0.0, -3000.0, 0.0))   # along the terrain
            else:
                X.layered_terrain(k, width=W, height=H)
            for _ in range(2):
                k.render()          # (the lists arrive with the second frame)
            pp, ids, rgb = gpu_frame(k)
            k.check(0, "render")
            frames.append((np.array(pp, copy=True), np.array(ids, copy=True), np.array(rgb, copy=True)))
            if variant == 0 and short == 0:
                status, ms, stats = _walk_bound(solr, _args(solr, k), repeats=1)
                k.check(status, "solr_hip_walk_bound")
                closest_ref, closest_free, _, _, general, unclassified = _lists(solr)
                assert general == 0 and unclassified == 0
                assert closest_free == ((W + 7) // 8) * ((H + 7) // 8), _lists(solr)      # the primary walks and no other
            if variant == 0 and short == 1:
                assert hip.solr_hip_short_ray_lists() == 1
                assert hip.solr_hip_order_free_nodes() > 1024          # a long list: the deep kernels
                misround = np.zeros((H, W), np.uint8)
                opp, oids, orgb, _, status = oracle_frame(k, oracle, misround=misround)
                assert status == 0
                res = assert_parity_pinned((pp, ids, rgb), (opp, oids, orgb), misround, 2, scene)
                bounced = (oids[..., 1] > 1).mean()
                assert bounced > (0.4 if scene == "layered_terrain" else 0.05), (bounced, res)
                status, ms, stats = _walk_bound(solr, _args(solr, k), repeats=1)
                k.check(status, "solr_hip_walk_bound")
                closest_ref, closest_free, shadow_ref, shadow_free, general, unclassified = _lists(solr)
                tiles = ((W + 7) // 8) * ((H + 7) // 8)
                assert general == 0 and unclassified == 0, _lists(solr)
                bounce_free = closest_free - tiles          # every tile's primary walk is order-free
                assert bounce_free > 0.3 * tiles, _lists(solr)
                # second walks (reference order) happen, and are the minority even here
                assert closest_ref <= bounce_free, _lists(solr)
                if scene == "layered_terrain":
                    assert closest_ref > 0, _lists(solr)
                assert shadow_ref == 0 and shadow_free > 0, _lists(solr)
            k.finalize()
    finally:
        hip.solr_hip_set_variant(0)
        hip.solr_hip_set_short_ray_lists(-1)
        hip.solr_hip_set_tile_scheduling(1)
    for other in frames[1:]:
        assert np.array_equal(frames[0][1], other[1]) and np.array_equal(frames[0][2], other[2])
        assert np.array_equal(frames[0][0].view(np.uint32), other[0].view(np.uint32))


def test_scenes_outside_the_lean_kernels_are_refused(solr):
    import scenes_extra as X
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    X.textured(k, width=96, height=64)
    try:
        k.render()
        status, _, _ = _walk_bound(solr, _args(solr, k), repeats=1)
        assert status == -1
        buf = C.create_string_buffer(512)
        assert hip.solr_hip_last_error(buf, 512) == -1 and b"no recording instantiation" in buf.value
        hip.solr_hip_clear_error()
        assert k.render().any()
    finally:
        hip.solr_hip_clear_error()
        k.finalize()
