"""GPU parity tests: the HIP path (host mirror -> C-ABI -> kernels) against the CPU oracle AS PINNED on the
same flattened scene.  The bar (BASELINE.json north_star): primitive ids exact, RGB8 exact, float
colour channels <= 1 ULP.  The oracle's transcendentals are libm's binary32 routines (what the reference's host
build calls); the engine evaluates them in binary64 and rounds once.  libm's results are within an ULP of that but
not always equal to it: where a frame depends on more than pow (sin / cos of procedural spheres; atan2 / asin of
sphere and skybox UVs) the comparison is helpers.assert_frame_pinned - every pixel at the bar except a COUNTED
handful, each of which the oracle itself marks as having gone through a libm result that is not the correctly rounded
value (oracle_set_misround_mask); a pixel outside the bar without that mark fails.  No test here switches the oracle
to the engine's transcendentals."""
import importlib

import numpy as np
import pytest

import scenes_extra as X
from helpers import (assert_frame_pinned, assert_parity, assert_pass_parity, compare_frames, device_frame, gpu_frame,
                     oracle_frame)

pytestmark = pytest.mark.gpu
solr_mod = importlib.import_module("sol-r_amd")


def both(solr, oracle, build, **kw):
    k = solr.Kernel(engine="hip")
    build(k, **kw)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    assert status == 0
    return k, compare_frames(pp, ids, rgb, opp, oids, orgb), counts


@pytest.mark.parametrize("iterations", [1, 3])
def test_cornell(solr, oracle, iterations):
    k, res, counts = both(solr, oracle, solr.scenes.cornell, width=160, height=120, iterations=iterations)
    k.finalize()
    assert_parity(res)


@pytest.mark.parametrize("size", [(1, 1), (8, 8), (13, 7), (65, 9), (200, 3)])
def test_image_sizes_that_do_not_fill_the_8x8_tiles(solr, oracle, size):
    k, res, _ = both(solr, oracle, solr.scenes.cornell, width=size[0], height=size[1], iterations=2)
    k.finalize()
    assert_parity(res)


# pixels of the 96 x 64 every-primitive frame that may lie outside the bar, each marked by the oracle: the procedural
# sphere's surface is displaced by cos / sin of the hit point (GI:235-246), a sinf / cosf result one ULP off moves the
# normal and the bounce ray by an ULP - the colour by a few ULP, RGB8 by at most two steps.  Measured: 9.
PROCEDURAL_EXCEPTIONS = 12


def test_every_primitive_type(solr, oracle):
    k = solr.Kernel(engine="hip")
    X.primitives_mix(k)
    frame = gpu_frame(k)
    res = assert_frame_pinned(k, oracle, frame, PROCEDURAL_EXCEPTIONS, "every primitive type", marked_bounds=(None, 2))
    k.finalize()
    print(res)
    assert res["depth_max_ulp"] == 0 and res["pixels_with_a_misrounded_libm_result"] > 0


def test_every_primitive_type_without_the_procedural_sphere(solr, oracle):
    def build(k, **kw):
        X.primitives_mix(k, **kw)
    k = solr.Kernel(engine="hip")
    X.primitives_mix(k, timestamp=0)
    # make the procedural material plain: then the frame has no transcendental but pow
    k.L.SolR_SetMaterial(3, 0.3, 0.8, 0.3, 0.0, 0.0, 0.0, 0, 0, 0, 0.0, 0.0, -1, -1, -1, -1, -1, -1, -1, 0.4, 30.0, 0.0,
                         0.0, 500000.0, 50000.0, 0)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    k.finalize()
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))


@pytest.mark.parametrize("scene", [X.triangles_only, X.sticks, X.lone_light])
def test_specialised_kernels(solr, oracle, scene):
    """each of these scenes selects a different instantiation of the renderer"""
    k, res, _ = both(solr, oracle, scene)
    k.finalize()
    assert_parity(res)


def test_all_triangle_mode(solr, oracle):
    k, res, _ = both(solr, oracle, X.triangles_only, extendedGeometry=0)
    k.finalize()
    assert_parity(res)


def test_double_sided_triangles(solr, oracle):
    k, res, _ = both(solr, oracle, X.triangles_only, doubleSidedTriangles=1)
    k.finalize()
    assert_parity(res)


@pytest.mark.parametrize("info", [
    dict(graphicsLevel=solr_mod.glNoShading), dict(graphicsLevel=solr_mod.glPhong),
    dict(graphicsLevel=solr_mod.glPhongAndBlinn), dict(graphicsLevel=solr_mod.glReflectionsAndRefractions),
    dict(gradientBackground=1), dict(atmosphericEffect=solr_mod.aeFog, viewDistance=30000.0),
    dict(cameraType=solr_mod.ctOrthographic), dict(cameraType=solr_mod.ctAntialiazed),
    dict(cameraType=solr_mod.ctVR), dict(cameraType=solr_mod.ctAnaglyph, eyeSeparation=350.0),
    dict(cameraType=solr_mod.ctPanoramic),
    dict(shadowIntensity=0.4), dict(renderBoxes=1),
    dict(frameBufferType=solr_mod.ftBGR), dict(bgColor=(0.2, 0.3, 0.4, 0.1)), dict(draftMode=1),
])
def test_scene_info_modes(solr, oracle, info):
    size = dict(width=64, height=64) if info.get("frameBufferType") else dict(width=96, height=64)
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, iterations=3, **size, **info)
    pp, ids, rgb = gpu_frame(k)
    if info.get("draftMode"):
        # CudaKernel.cpp:287-291: the engine class renders iteration 0 unshaded in draft mode
        k.set_scene_info(graphicsLevel=solr.glNoShading)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    k.finalize()
    assert status == 0
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))


# pixels of the 96 x 64 textured frame that may show the neighbouring texel (helpers.assert_parity_pinned, marked_bounds):
# sphere and skybox UVs are atan2 / asin of the hit point, and glibc's atan2f / asinf land on the other side of a texel
# boundary on a few pixels of a frame.  Measured: 5 with the textured skybox, 2 without.
TEXEL_EXCEPTIONS = 8


@pytest.mark.parametrize("skybox", [True, False], ids=["with-skybox", "without-skybox"])
def test_textures(solr, oracle, skybox):
    k = solr.Kernel(engine="hip")
    X.textured(k, skybox=skybox)
    frame = gpu_frame(k)
    res = assert_frame_pinned(k, oracle, frame, TEXEL_EXCEPTIONS, "textured scene", marked_bounds=(None, None))
    k.finalize()
    print(res)
    assert res["ids_all_equal"] and res["depth_max_ulp"] == 0


def progressive(solr, oracle, build, passes, **info):
    """frames with pathTracingIteration = 0, 1, 2, ...: the engine keeps its buffers on the device.  Every pass is
    compared twice: with the oracle's pass over the ENGINE's previous buffers - one pass, one rounding of the
    running sum, held to the bar (<= 1 ULP) right here - and with the oracle handed its own previous frame, whose
    running sum drifts by a rounding per accumulated sample (the figures returned; callers allow 2 ULP)."""
    k = solr.Kernel(engine="hip")
    build(k, **info)
    opp = oids = None
    previous = None
    worst = None
    for it in passes:
        k.set_scene_info(pathTracingIteration=it, maxPathTracingIterations=max(passes) + 1)
        pp, ids, rgb = gpu_frame(k)
        if previous is not None:
            assert_pass_parity(k, oracle, (pp, ids, rgb), previous, what="pass %d over the engine's previous buffers" % it)
        previous = (pp.copy(), ids.copy())
        opp, oids, orgb, counts, status = oracle_frame(k, oracle, pp=opp, ids=oids)
        assert status == 0, "oracle read outside the random buffer"
        res = compare_frames(pp, ids, rgb, opp, oids, orgb)
        res["iteration"] = it
        if worst is None or res["max_ulp"] > worst["max_ulp"] or not res["ids_all_equal"]:
            worst = res
        if not res["ids_all_equal"]:
            break
    k.finalize()
    return worst


def test_progressive_refinement_passes(solr, oracle):
    # CRT:121-123, 454-458, 540-549: iterations 1..10 re-render with deeper bounce limits, skipping finished pixels
    res = progressive(solr, oracle, solr.scenes.cornell, range(0, 11), width=96, height=64, iterations=1)
    print(res)
    assert_parity(res)


def test_accumulation_passes(solr, oracle):
    # CRT:470-479, 515-522, 550-562 + GI:969-976: jittered, accumulated samples beyond iteration 10
    res = progressive(solr, oracle, solr.scenes.cornell, range(0, 15), width=64, height=48, iterations=1)
    print(res)
    assert_parity(res, max_ulp=2)   # the running sum adds one rounding per accumulated sample


@pytest.mark.parametrize("mode", [solr_mod.aiBasic, solr_mod.aiFull, solr_mod.aiRandomIllumination])
def test_advanced_illumination(solr, oracle, mode):
    res = progressive(solr, oracle, solr.scenes.cornell, range(8, 13), width=64, height=48, iterations=1,
                      advancedIllumination=mode)
    print(res)
    assert_parity(res, max_ulp=2)


@pytest.mark.parametrize("pp", [dict(type=solr_mod.ppe_ambientOcclusion, param1=0.0, param2=10.0, param3=0),
                                dict(type=solr_mod.ppe_depthOfField, param1=12000.0, param2=20.0, param3=16),
                                dict(type=solr_mod.ppe_radiosity, param1=0.0, param2=4000.0, param3=12),
                                dict(type=solr_mod.ppe_cartoon, param1=9000.0, param2=0.0, param3=0)] +
                               [dict(type=solr_mod.ppe_filter, param1=0.0, param2=0.0, param3=f) for f in range(7)],
                         ids=lambda pp: "type%d-%d" % (pp["type"], pp["param3"]))
def test_post_processing(solr, oracle, pp):
    k = solr.Kernel(engine="hip")
    k.set_post_processing(**pp)
    solr.scenes.cornell(k, width=96, height=64, iterations=2)
    pp_, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    k.finalize()
    assert status == 0
    assert_parity(compare_frames(pp_, ids, rgb, opp, oids, orgb))


@pytest.mark.parametrize("param2", [10.0, 120.0, 500.0, 2500.0, 6000.0])
def test_ambient_occlusion_taps_that_share_a_depth_are_one_comparison(solr, oracle, param2):
    """k_ambientOcclusion on a frame wide enough for tiles whose 256 tap offsets are the same for every pixel (x and y
    inside one binade): the workgroup dedupes the offsets and a pixel compares once per DISTINCT offset, weighted by
    the number of taps that share it - four offsets for param2 = 10 (cfg4's), dozens for 120 and 500, the plain loop
    beyond 128 (2500); 6000: taps that reach 48 pixels, a window that does not fit LDS - every tap gathered from memory.
    The count is an integer either way: the oracle's image, every pixel"""
    k = solr.Kernel(engine="hip")
    k.set_post_processing(type=solr_mod.ppe_ambientOcclusion, param1=0.0, param2=param2, param3=0)
    solr.scenes.cornell(k, width=640, height=208, iterations=1)
    pp_, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    # ... and the order the workgroups take the tiles in (the frame's heavy tiles first; variant 9: a fixed stride of
    # tiles per workgroup) changes no pixel
    solr.hip_lib().solr_hip_set_variant(9)
    _, _, rgb_stride = gpu_frame(k)
    solr.hip_lib().solr_hip_set_variant(0)
    k.finalize()
    assert status == 0
    assert_parity(compare_frames(pp_, ids, rgb, opp, oids, orgb))
    assert np.array_equal(rgb, rgb_stride)
    assert len(np.unique(rgb.reshape(-1, 3), axis=0)) > 50


def test_ambient_occlusion_tile_order_on_a_tall_and_a_wide_frame(solr, oracle):
    """the lists of heavy and light tile rows / columns (k_ambientOcclusion): frames with more tile rows than threads
    of a workgroup would list at once (2 100 rows: 263 tile rows), with one tile column, and with one tile row"""
    for W, H in ((40, 2100), (1500, 8), (32, 8)):
        k = solr.Kernel(engine="hip")
        k.set_post_processing(type=solr_mod.ppe_ambientOcclusion, param1=0.0, param2=10.0, param3=0)
        solr.scenes.cornell(k, width=W, height=H, iterations=1)
        pp_, ids, rgb = gpu_frame(k)
        opp, oids, orgb, counts, status = oracle_frame(k, oracle)
        k.finalize()
        assert status == 0, (W, H)
        assert np.array_equal(rgb, orgb), (W, H)


def test_row_strips_equal_the_full_frame(solr, oracle):
    """what each rank of a multi-GPU run renders: rows [first, first+count) with its own buffers"""
    W, H = 96, 70
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=3)
    full = k.render()
    hip = solr.hip_lib()
    assembled = np.zeros_like(full)
    for rank in range(3):
        first, count, _ = solr.strip_rows(rank, 3, H)
        hip.solr_hip_set_strip(first, count)
        img = k.render()                       # d2h_bitmap places the strip at its rows
        assembled[first:first + count] = img[first:first + count]
        opp, oids, orgb, _, _ = oracle_frame(k, oracle, first_row=first, nb_rows=count)
        assert np.array_equal(img[first:first + count], orgb)
    hip.solr_hip_set_strip(0, -1)
    k.finalize()
    assert np.array_equal(assembled, full)


def test_an_empty_strip_renders_nothing(solr):
    """more processes than rows to share out (strip_rows gives the trailing ranks no row): such a rank
    must not render the whole frame into its strip-sized buffer"""
    import ctypes as C
    W, H = 64, 17
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    full = k.render()
    hip = solr.hip_lib()
    try:
        first, count, per = solr.strip_rows(7, 8, H)
        assert count == 0 and per == 3
        hip.solr_hip_set_strip(first, count)
        guard = np.full((per + 2, W, 3), 0xAB, np.uint8)   # what a strip-sized send buffer of the gather holds
        image = guard.copy()
        si, ppi, eye, direction, angles = k.frame_parameters()
        flat = k.flat_scene()
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        k.check(0, "empty strip")
        # d2h places `count` rows at row `first` of a full-size image: nothing may be written
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(image.ctypes.data), None)
        k.check(0, "empty strip read-back")
        assert np.array_equal(image, guard)
    finally:
        hip.solr_hip_set_strip(0, -1)
    again = k.render()
    k.finalize()
    assert np.array_equal(again, full)


def test_frames_in_flight_with_a_moving_camera_render_every_tile(solr):
    """The cost-ordered launch sorts the tiles while the frame on the other stream is still storing its
    costs (k_orderTiles): the order must stay a permutation, or a tile is skipped and shows the picture of
    two frames earlier.  Frames are issued back to back, no host synchronisation between them, with the
    camera moving so that a stale tile differs; the last frames are compared with the same views rendered
    in raster order one at a time."""
    import ctypes as C
    W, H = 512, 384     # 3072 tiles: the sort runs over several batches per thread
    k = solr.Kernel(engine="hip")
    X.primitives_mix(k)
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    k.render()
    hip = solr.hip_lib()
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    def view(i):
        e = eye.copy()
        e[0] += 90.0 * i
        e[1] += 40.0 * (i % 7)
        return e

    def render(i):
        e = view(i)
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(e), fp(direction), fp(angles))

    nb = 200  # the order is refreshed every 64th frame: several sorts overlap a running frame
    try:
        hip.solr_hip_set_tile_scheduling(0)
        hip.solr_hip_set_frames_in_flight(1)
        expected = {}
        for i in (nb - 2, nb - 1):
            render(i)
            expected[i] = device_frame(solr, si)
        for flights in (2, 3):
            hip.solr_hip_set_tile_scheduling(2)     # cost order forced
            hip.solr_hip_set_frames_in_flight(flights)
            for i in range(nb - 1):
                render(i)
            # frame nb-2 is the newest on its buffer set: read it, then issue the last one
            k.check(0, "frames in flight")
            pp, ids, rgb = device_frame(solr, si)
            assert np.array_equal(ids, expected[nb - 2][1]), flights
            assert np.array_equal(pp.view(np.uint32), expected[nb - 2][0].view(np.uint32)), flights
            render(nb - 1)
            pp, ids, rgb = device_frame(solr, si)
            assert np.array_equal(rgb, expected[nb - 1][2]), flights
            assert np.array_equal(ids, expected[nb - 1][1]), flights
            assert np.array_equal(pp.view(np.uint32), expected[nb - 1][0].view(np.uint32)), flights
    finally:
        hip.solr_hip_set_tile_scheduling(1)
        hip.solr_hip_set_frames_in_flight(1)
        k.finalize()


def test_ray_census_matches_the_oracle(solr, oracle):
    import ctypes as C
    k = solr.Kernel(engine="hip")
    X.primitives_mix(k)
    k.render()
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    counts = (C.c_ulonglong * 8)()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    solr.hip_lib().solr_hip_render_counting(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction),
                                            fp(angles), counts)
    k.check(0, "census")
    _, _, _, ocounts, status = oracle_frame(k, oracle)
    k.finalize()
    assert [int(c) for c in counts[:4]] == ocounts
    assert counts[4] * 64 >= counts[2] and counts[5] * 64 >= counts[3]


def test_engine_survives_scene_changes(solr, oracle):
    """dirty-flag protocol: new primitives / materials / camera between frames re-upload what changed"""
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=64, height=48, iterations=2)
    k.render()
    k.L.SolR_RotatePrimitives(0, 0, 0.0, 0.0, 0.0, 0.0, 0.4, 0.1)
    k.set_camera((500.0, 200.0, -14000.0), angles=(0.0, 0.1, 0.0))
    k.L.SolR_SetMaterial(0, 0.9, 0.1, 0.1, 0.0, 0.7, 0.0, 0, 0, 0, 0.0, 0.0, -1, -1, -1, -1, -1, -1, -1, 1.0, 100.0, 0.0,
                         0.0, 500000.0, 50000.0, 0)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    k.finalize()
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))


def test_cost_ordered_launch_changes_no_pixel(solr, oracle):
    """Tile scheduling (solr_hip_set_tile_scheduling): frames launched most-expensive-tile-first are
    bit-identical to frames launched in raster order, and still equal to the oracle."""
    import ctypes as C
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    X.sticks(k, width=200, height=120)
    try:
        hip.solr_hip_set_tile_scheduling(0)
        pp0, ids0, rgb0 = gpu_frame(k)
        k.check(0, "raster-order frame")
        flat = k.flat_scene()
        si, ppi, eye, direction, angles = k.frame_parameters()
        si.pathTracingIteration = 0
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        hip.solr_hip_set_tile_scheduling(2)
        for i in range(4):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            hip.solr_hip_synchronize()
            k.check(0, "cost-ordered frame %d" % i)
            if i >= 1:
                assert hip.solr_hip_tile_scheduling_active() == 1
            pp, ids, rgb = device_frame(solr, si)
            assert np.array_equal(pp.view(np.uint32), pp0.view(np.uint32)), i
            assert np.array_equal(ids, ids0) and np.array_equal(rgb, rgb0), i
        hip.solr_hip_set_tile_scheduling(1)   # automatic: whatever it decides, the frame is the same
        for i in range(3):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            hip.solr_hip_synchronize()
            pp, ids, rgb = device_frame(solr, si)
            assert np.array_equal(pp.view(np.uint32), pp0.view(np.uint32)) and np.array_equal(ids, ids0), i
        opp, oids, orgb, counts, status = oracle_frame(k, oracle)
        assert status == 0
        assert_parity(compare_frames(pp, ids, rgb0, opp, oids, orgb))
    finally:
        hip.solr_hip_set_tile_scheduling(1)
        k.finalize()


def test_blinn_exponents_outside_the_lean_pow_domain(solr, oracle):
    """pow_f (rt_device.h) has a lean path for exponents in (0, 4096] and sends everything else to the
    library routine: specular powers 0 and 10 000 take that road."""
    def build(k, **kw):
        k.initialize(width=96, height=64, nbRayIterations=2, **kw)
        a = k.add_material(0.8, 0.3, 0.2, reflection=0.3, specValue=1.0, specPower=0.0)
        b = k.add_material(0.2, 0.7, 0.9, specValue=0.8, specPower=10000.0)
        c = k.add_material(0.5, 0.5, 0.5, specValue=0.6, specPower=60.0)
        k.add_primitive(solr.ptSphere, (-2500.0, 0.0, 0.0), size=(2000.0, 0, 0), material=a)
        k.add_primitive(solr.ptSphere, (2500.0, 0.0, 0.0), size=(2000.0, 0, 0), material=b)
        k.add_primitive(solr.ptXZPlane, (0.0, -2500.0, 0.0), size=(9000.0, 0.0, 9000.0), material=c)
        solr.scenes.add_light(k, position=(3000.0, 8000.0, -8000.0))
        k.compact_boxes(True)
        k.set_camera((100.0, 500.0, -12000.0))
        return k
    k, res, _ = both(solr, oracle, build)
    k.finalize()
    assert_parity(res, rgb_max_diff=1)


def test_two_frames_in_flight(solr, oracle):
    """solr_hip_set_frames_in_flight(2): first-pass frames alternate between two streams and buffer sets;
    every frame read back is the frame just rendered, and refinement passes continue on the set of
    the pass before them."""
    import ctypes as C
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    X.sticks(k, width=136, height=72)
    try:
        hip.solr_hip_set_frames_in_flight(2)
        assert hip.solr_hip_get_frames_in_flight() == 2
        pp0, ids0, rgb0 = gpu_frame(k)
        k.check(0, "first frame")
        opp, oids, orgb, counts, status = oracle_frame(k, oracle)
        assert status == 0
        assert_parity(compare_frames(pp0, ids0, rgb0, opp, oids, orgb))
        flat = k.flat_scene()
        si, ppi, eye, direction, angles = k.frame_parameters()
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        # a burst of first-pass frames, no synchronisation in between, from two camera positions
        eye2 = eye.copy()
        eye2[0] += 700.0
        si.pathTracingIteration = 0
        for i in range(7):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye if i % 2 == 0 else eye2),
                                fp(direction), fp(angles))
        k.check(0, "burst")
        pp, ids, rgb = device_frame(solr, si)   # frame 6: camera 1
        assert np.array_equal(pp.view(np.uint32), pp0.view(np.uint32)) and np.array_equal(ids, ids0)
        assert np.array_equal(rgb, rgb0)
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye2), fp(direction), fp(angles))
        pp2, ids2, _ = device_frame(solr, si)  # frame 7: camera 2, the other buffer set
        opp2, oids2, _, _, status = oracle.render(flat, si, ppi, eye2, direction, angles, nthreads=4)
        assert status == 0
        assert np.array_equal(ids2, oids2) and not np.array_equal(ids2, ids0)
        # refinement passes 1..3 continue where pass 0 of camera 2 left off
        pp_ref, ids_ref = opp2, oids2
        for it in (1, 2, 3):
            si.pathTracingIteration = it
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye2), fp(direction), fp(angles))
            pp_ref, ids_ref, _, _, status = oracle.render(flat, si, ppi, eye2, direction, angles, pp=pp_ref,
                                                          ids=ids_ref, nthreads=4)
            assert status == 0
        ppn, idsn, _ = device_frame(solr, si)
        assert np.array_equal(idsn, ids_ref)
        from helpers import ulp_distance
        assert int(ulp_distance(ppn[..., :3], pp_ref[..., :3]).max()) <= 1
    finally:
        hip.solr_hip_set_frames_in_flight(1)
        k.finalize()


def test_post_processing_and_strips_with_two_frames_in_flight(solr, oracle):
    """The neighbourhood post-process and a row strip on both buffer sets of the two-frame mode."""
    import ctypes as C
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    k.set_post_processing(type=solr_mod.ppe_ambientOcclusion, param1=0.0, param2=10.0, param3=0)
    solr.scenes.cornell(k, width=96, height=64, iterations=2)
    try:
        hip.solr_hip_set_frames_in_flight(2)
        pp0, ids0, rgb0 = gpu_frame(k)
        opp, oids, orgb, counts, status = oracle_frame(k, oracle)
        assert status == 0
        assert_parity(compare_frames(pp0, ids0, rgb0, opp, oids, orgb))
        flat = k.flat_scene()
        si, ppi, eye, direction, angles = k.frame_parameters()
        si.pathTracingIteration = 0
        objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        for i in range(3):   # sets 1, 0, 1
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            pp, ids, rgb = device_frame(solr, si)
            assert np.array_equal(rgb, rgb0) and np.array_equal(ids, ids0), i
            assert np.array_equal(pp.view(np.uint32), pp0.view(np.uint32)), i
        # a strip of rows 16..47 on each set
        hip.solr_hip_set_strip(16, 32)
        for i in range(2):
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            k.check(0, "strip frame")
            spp = np.zeros((32, 96, 8), np.float32)
            hip.solr_hip_d2h_postprocessing(C.c_void_p(spp.ctypes.data))
            assert np.array_equal(spp.view(np.uint32), pp0[16:48].view(np.uint32)), i
    finally:
        hip.solr_hip_set_strip(0, -1)
        hip.solr_hip_set_frames_in_flight(1)
        k.finalize()


def test_anaglyph_camera_through_refinement_and_accumulation_passes(solr, oracle):
    # k_anaglyphRenderer, CRT:840-950: two traces per pixel, plain store then plain accumulation
    res = progressive(solr, oracle, solr.scenes.cornell, range(0, 14), width=64, height=48, iterations=2,
                      cameraType=solr_mod.ctAnaglyph, eyeSeparation=420.0)
    assert_parity(res, max_ulp=2)


def test_textures_in_the_all_triangle_mode(solr, oracle):
    # extendedGeometry == 0 tests every primitive as a triangle (GI:743-747) and textures it as one
    # (intersectionShader's else branch): the kernel chosen for that mode has to carry the texture tier
    k = solr.Kernel(engine="hip")
    X.textured(k, width=96, height=64, skybox=False, extendedGeometry=0)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, counts, status = oracle_frame(k, oracle)
    assert status == 0
    res = compare_frames(pp, ids, rgb, opp, oids, orgb)
    assert (ids[..., 0] >= 0).mean() > 0.02, "no triangle in view"
    assert_parity(res)
    k.finalize()


def test_a_texture_id_that_was_never_loaded_does_not_fault(solr):
    # A material that names a texture nobody loaded.  Through the host protocol realignTexturesAndMaterials
    # gives it a 0 x 0 mapping (no texel is fetched; `x % 0` is guarded on the device); uploaded as
    # GPUKernel::setMaterial leaves it - the id next to the 40000 x 40000 "computed texture" mapping - the
    # reference's mappers would read gigabytes past the atlas, and h2d_materials drops the id instead.
    k = solr.Kernel(engine="hip")
    k.initialize(width=64, height=48, nbRayIterations=2)
    k.set_texture(0, np.full((8, 8, 3), 200, np.uint8))
    wall = k.add_material(0.8, 0.4, 0.2, diffuseTextureId=7)
    ball = k.add_material(0.2, 0.6, 0.9, diffuseTextureId=7, normalTextureId=300, reflection=0.3)
    k.add_primitive(solr_mod.ptXYPlane, (0, 0, 4000), size=(9000, 6000, 0), material=wall)
    k.add_primitive(solr_mod.ptSphere, (0, 0, 0), size=(1500, 0, 0), material=ball)
    t = k.add_primitive(solr_mod.ptTriangle, (-3000, -2000, 500), (-1000, -2000, 0), (-2000, 1500, 300), material=ball)
    k.set_texture_coordinates(t, (0.1, 0.1), (0.9, 0.2), (0.4, 0.9))
    solr_mod.scenes.add_light(k)
    k.compact_boxes(True)
    k.set_scene_info(skyboxMaterialId=wall, skyboxSize=30000)
    k.set_camera((0.0, 0.0, -12000.0))
    first = gpu_frame(k)
    k.check(0, "render")
    second = gpu_frame(k)
    k.check(0, "render")
    assert (first[1][..., 0] >= 0).mean() > 0.2
    assert np.array_equal(first[0].view(np.uint32), second[0].view(np.uint32))
    k.finalize()


def test_texture_tables_that_point_outside_the_atlas_are_refused(solr):
    """fetchTexel indexes the atlas with textureOffset + texel index, unchecked (as the reference does, which
    reads whatever is there): an offset beyond the uploaded atlas, or a textured material with no atlas at all,
    must end in an error the caller can read, not in a GPU memory fault."""
    import ctypes as C
    k = solr.Kernel(engine="hip")
    X.textured(k, width=64, height=48, skybox=False)
    good = k.render()
    k.check(0, "textured frame")
    hip = solr.hip_lib()
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    occ = 1 | (1 << 32)   # vec2i {1, 1} by value
    mats = np.array(flat.materials, copy=True)
    textured = [i for i in range(len(mats)) if mats["textureIds"][i][0] >= 0 and mats["textureMapping"][i][0] > 0]
    assert textured, "the scene has no textured material"
    bad = mats.copy()
    bad["textureOffset"][textured[0]][0] = 1 << 30
    try:
        hip.h2d_materials(occ, C.c_void_p(bad.ctypes.data), len(bad))
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        text = C.create_string_buffer(512)
        assert hip.solr_hip_last_error(text, 512) != 0
        assert b"outside the uploaded atlas" in text.value, text.value
        hip.solr_hip_clear_error()
        # the tables as they were: renders again, same picture
        hip.h2d_materials(occ, C.c_void_p(mats.ctypes.data), len(mats))
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
        assert hip.solr_hip_last_error(None, 0) == 0
        pp, ids, rgb = device_frame(solr, si)
        assert np.array_equal(rgb, good)
    finally:
        hip.solr_hip_clear_error()
        k.finalize()


def _stereo_scene(k, **info):
    solr_mod.scenes.cornell(k, **info)
    # a look-at point off the z = 0 plane: the eyes' distance is eyeSeparation x look-at depth / focus depth
    k.set_camera((200.0, 100.0, -15000.0), look_at=(0.0, 0.0, 4000.0))


def test_3d_vision_camera_through_refinement_and_accumulation_passes(solr, oracle):
    # cameraType ctVR is dispatched to k_3DVisionRenderer (CRT:1737-1755, 953-1043): side-by-side eyes whose
    # distance follows the depth the frame before stored for the focus pixel; plain store, then accumulation
    res = progressive(solr, oracle, _stereo_scene, range(0, 14), width=96, height=64, iterations=2,
                      cameraType=solr_mod.ctVR, eyeSeparation=380.0)
    print(res)
    assert_parity(res, max_ulp=2)


def test_3d_vision_camera_reads_the_focus_depth_of_the_frame_before(solr, oracle):
    k = solr.Kernel(engine="hip")
    _stereo_scene(k, width=96, height=64, iterations=2, cameraType=solr_mod.ctVR, eyeSeparation=380.0)
    first = gpu_frame(k)            # focus depth 0: nothing has been rendered yet
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert_parity(compare_frames(*first, opp, oids, orgb))
    second = gpu_frame(k)           # now the depth the first frame left behind
    opp2, oids2, orgb2, _, status = oracle_frame(k, oracle, pp=opp, ids=oids)
    assert_parity(compare_frames(*second, opp2, oids2, orgb2))
    assert not np.array_equal(first[2], second[2]), "the eyes did not move with the focus depth"
    k.finalize()


def _panorama_inside_the_room(k, **info):
    solr_mod.scenes.cornell(k, **info)
    k.set_camera((150.0, -300.0, -1200.0), look_at=(150.0, -300.0, 2800.0))
    k.set_post_processing(type=solr_mod.ppe_none, param2=0.002)   # strength of the depth-of-field jitter


def test_fisheye_camera_through_refinement_and_accumulation_passes(solr, oracle):
    # k_fishEyeRenderer, CRT:741-815: 360 degrees across the image width from inside the room, the
    # depth-of-field jitter of the accumulation passes reads the depth the first pass stored.  The per-pixel
    # cos / sin of the turn are libm binary32 in the oracle and binary64-rounded-once in the engine.
    res = progressive(solr, oracle, _panorama_inside_the_room, range(0, 14), width=128, height=48, iterations=2,
                      cameraType=solr_mod.ctPanoramic)
    print(res)
    assert_parity(res, max_ulp=2)


@pytest.mark.parametrize("scene", ["cornell", "primitives_mix"])
def test_short_paths_of_plain_primitives_change_nothing(solr, oracle, scene):
    """plain spheres and axis planes are classified at upload (PrimKind in the primitive's tag) and take a short
    path through both walks; SOLR_HIP_NO_KINDS sends every primitive through the general tests: same frame, bit
    for bit, and both are the oracle's"""
    import os
    build = solr.scenes.cornell if scene == "cornell" else X.primitives_mix
    frames = []
    for no_kinds in (False, True):
        if no_kinds:
            os.environ["SOLR_HIP_NO_KINDS"] = "1"
        try:
            k = solr.Kernel(engine="hip")
            if scene == "cornell":
                build(k, width=160, height=120, iterations=3)
            else:
                build(k)
            pp, ids, rgb = gpu_frame(k)
            if not no_kinds:
                if scene == "cornell":
                    assert_frame_pinned(k, oracle, (pp, ids, rgb), 2, scene)
                else:   # libm's binary32 sinf / cosf on the procedural sphere: test_every_primitive_type
                    assert_frame_pinned(k, oracle, (pp, ids, rgb), PROCEDURAL_EXCEPTIONS, scene, marked_bounds=(None, 2))
            frames.append((np.array(pp, copy=True), np.array(ids, copy=True), np.array(rgb, copy=True)))
            k.finalize()
        finally:
            os.environ.pop("SOLR_HIP_NO_KINDS", None)
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))
    assert np.array_equal(frames[0][1], frames[1][1]) and np.array_equal(frames[0][2], frames[1][2])


def test_ambient_occlusion_on_a_strip_reads_the_neighbours_rows(solr, oracle):
    """Multi-GPU frames with the ambient-occlusion kernel: a strip's taps reach into the rows of the ranks above and
    below.  With their depths handed over (solr_hip_set_depth_halo - what the RCCL exchange of cudaRender delivers
    between ranks) the strips assemble to the frame one GPU renders, bit for bit; without them the rows next to a
    seam differ, as they do between the devices of the reference's own split."""
    import ctypes as C
    W, H, reach = 192, 96, 18          # param2 = 2000: taps up to 16 * 2000 * 0.005 / 10 = 16 pixels away, + 2
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    k.set_post_processing(type=solr_mod.ppe_ambientOcclusion, param1=0.0, param2=2000.0, param3=0)
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
    try:
        pp, ids, full = gpu_frame(k)
        opp, oids, orgb, _, status = oracle_frame(k, oracle)
        assert status == 0
        assert_parity(compare_frames(pp, ids, full, opp, oids, orgb))
        full = np.array(full, copy=True)
        depth = np.ascontiguousarray(pp[..., 3], dtype=np.float32)
        seams_seen = 0
        for rank in range(3):
            first, count, _ = solr.strip_rows(rank, 3, H)
            hip.solr_hip_set_strip(first, count)
            alone = np.array(k.render()[first:first + count], copy=True)
            seams_seen += int(not np.array_equal(alone, full[first:first + count]))
            above = np.ascontiguousarray(depth[max(0, first - reach):first])
            below = np.ascontiguousarray(depth[first + count:first + count + reach])
            hip.solr_hip_set_depth_halo(fp(above) if len(above) else None, len(above),
                                        fp(below) if len(below) else None, len(below))
            k.check(0, "solr_hip_set_depth_halo")
            img = k.render()
            k.check(0, "strip with halo")
            assert np.array_equal(img[first:first + count], full[first:first + count]), rank
            hip.solr_hip_set_depth_halo(None, 0, None, 0)
        assert seams_seen >= 2, "the taps never crossed a seam: the test would prove nothing"
    finally:
        hip.solr_hip_set_depth_halo(None, 0, None, 0)
        hip.solr_hip_set_strip(0, -1)
        k.finalize()


def _tie_scene(k, width=160, height=120, mirror=False, jitter=0.0, **info):
    """Primitives that tie: every sphere, triangle and cylinder is in the scene twice (and once more with another
    material), at the same place - equal hit distances bit for bit.  The reference keeps the one it visits
    first; which one that is follows from its flatten order, not from the order of insertion."""
    rng = solr_mod.scenes.LCG(11)
    k.initialize(width=width, height=height, nbRayIterations=3 if mirror else 2, **info)
    mats = [k.add_material(0.9, 0.2, 0.2, specValue=0.3, specPower=40.0), k.add_material(0.2, 0.9, 0.2, reflection=0.4),
            k.add_material(0.2, 0.3, 0.9, specValue=0.8, specPower=100.0)]
    if mirror:
        # a mirror floor under everything: most of the frame is bounce rays among the tied primitives
        floor = k.add_material(0.8, 0.8, 0.8, reflection=0.85)
        k.add_primitive(solr_mod.ptXZPlane, (0, -5200, 2000), size=(14000, 0, 12000), material=floor)
    for copy in range(3):
        rng = solr_mod.scenes.LCG(11)       # the same geometry again
        u = rng.uniform
        dz = copy * jitter                  # ... or a hair behind it: distances that differ in their last bits
        for i in range(40):
            c = (u(-6000, 6000), u(-4000, 4000), u(-3000, 6000) + dz)
            k.add_primitive(solr_mod.ptSphere, c, size=(u(300, 900), 0, 0), material=mats[(i + copy) % 3])
        for i in range(60):
            p0 = (u(-7000, 7000), u(-4500, 4500), u(0, 7000) + dz)
            p1 = (p0[0] + u(-1500, 1500), p0[1] + u(-1500, 1500), p0[2] + u(-800, 800))
            p2 = (p0[0] + u(-1500, 1500), p0[1] + u(-1500, 1500), p0[2] + u(-800, 800))
            t = k.add_primitive(solr_mod.ptTriangle, p0, p1, p2, material=mats[(i + 2 * copy) % 3])
            k.set_normals(t, (0, 0, -1), (0.1, 0, -1), (0, 0.1, -1))
        for i in range(20):
            a = (u(-6000, 6000), u(-4000, 4000), u(-2000, 5000) + dz)
            b = (a[0] + u(-1200, 1200), a[1] + u(-1200, 1200), a[2] + u(-1200, 1200))
            k.add_primitive(solr_mod.ptCylinder, a, b, size=(u(80, 250), 0, 0), material=mats[(i + copy) % 3])
    k.add_primitive(solr_mod.ptXYPlane, (0, 0, 9000), size=(12000, 8000, 0), material=mats[0])
    X._light(k)
    k.compact_boxes(True)
    if mirror:
        k.set_camera((300.0, 3000.0, -15000.0), look_at=(0.0, -4000.0, 0.0), angles=(0.02, -0.03, 0.0))
    else:
        k.set_camera((300.0, 200.0, -15000.0), look_at=(0.0, 0.0, 0.0), angles=(0.02, -0.03, 0.0))
    return k


@pytest.mark.parametrize("mirror,jitter", [(False, 0.0), (True, 0.0), (True, 0.01), (False, 0.004)],
                         ids=["primary", "bounce", "bounce-near-ties", "primary-near-ties"])
def test_order_free_walk_resolves_ties_as_the_reference_does(solr, oracle, mirror, jitter):
    """Primary rays walk a hierarchy of the engine's own over the scene's leaves, in an order of its own (DESIGN.md
    section 4): equal distances must go to the primitive the reference visits first, and nothing may depend on
    the order otherwise.  A scene in which every primitive exists three times is rendered with the order-free
    lists (the default), without them (variant 6) and by the oracle: ids, depth and RGB8 identical, float
    colour within 1 ULP of the oracle and bit for bit between the two engine forms.  "bounce": the same over a
    mirror floor - unit-length rays take the lists in the checked form (lanes whose best hit has a rival within
    the margin walk the reference's order afterwards), and here nearly every hit has one.  "near-ties": the
    copies lie a hundredth of a unit behind one another, distances of thousands that differ in their last bits or
    not at all - whatever the reference's strict comparison and cut-off make of them has to come out."""
    hip = solr.hip_lib()
    frames = []
    try:
        for variant in (0, 6):
            hip.solr_hip_set_variant(variant)
            k = solr.Kernel(engine="hip")
            _tie_scene(k, mirror=mirror, jitter=jitter)
            pp, ids, rgb = gpu_frame(k)
            assert (hip.solr_hip_order_free_nodes() > 0) == (variant == 0)
            assert hip.solr_hip_order_free_shadows() == (1 if variant == 0 else 0)   # nothing transparent here
            if variant == 0:
                opp, oids, orgb, _, status = oracle_frame(k, oracle)
                assert status == 0
                res = compare_frames(pp, ids, rgb, opp, oids, orgb)
                assert_parity(res)
                hit = oids[..., 0] >= 0
                assert hit.mean() > 0.3
            frames.append((np.array(pp, copy=True), np.array(ids, copy=True), np.array(rgb, copy=True)))
            k.finalize()
    finally:
        hip.solr_hip_set_variant(0)
    assert np.array_equal(frames[0][1], frames[1][1]) and np.array_equal(frames[0][2], frames[1][2])
    assert np.array_equal(frames[0][0].view(np.uint32), frames[1][0].view(np.uint32))


def test_order_free_lists_arrive_with_the_second_frame(solr, oracle):
    """the default: the first frame after an upload walks the reference's order everywhere, the lists are built
    before the second (SOLR_HIP_FREE_AFTER, 2 unless set) - and nobody can tell from the frames"""
    import os
    hip = solr.hip_lib()
    saved = os.environ.pop("SOLR_HIP_FREE_AFTER", None)
    try:
        k = solr.Kernel(engine="hip")
        solr.scenes.molecule(k, atoms=300, width=160, height=120)
        first = gpu_frame(k)
        assert hip.solr_hip_order_free_nodes() == 0
        second = gpu_frame(k)
        assert hip.solr_hip_order_free_nodes() > 0
        third = gpu_frame(k)
        opp, oids, orgb, _, status = oracle_frame(k, oracle)
        assert status == 0
        assert_parity(compare_frames(first[0], first[1], first[2], opp, oids, orgb))
        for frame in (second, third):
            assert np.array_equal(frame[0].view(np.uint32), first[0].view(np.uint32))
            assert np.array_equal(frame[1], first[1]) and np.array_equal(frame[2], first[2])
        k.finalize()
    finally:
        if saved is not None:
            os.environ["SOLR_HIP_FREE_AFTER"] = saved


@pytest.mark.gpu
@pytest.mark.parametrize("flights", [2, 3])
def test_frames_in_flight_through_the_frame_protocol(solr, flights):
    """GPUKernel::setFramesInFlight: render_begin starts the read-back of its frame behind the kernel, render_end
    delivers the image of the frame `flights - 1` calls back while the newer ones render.  Same frames, bit for bit,
    as the reference's one-at-a-time protocol: with a moving camera (first-pass frames rotate over the engine's
    buffer sets) and through refinement and accumulation passes (which stay on one set: its second RGB image takes
    every other pass, so that no pass waits for the copy of the pass before)."""
    import ctypes as C
    W, H = 192, 120
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2, maxPathTracingIterations=40)
    L = k.L

    def delivered():
        ptr = L.SolRx_GetBitmap()
        return np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3).copy()

    def frames(n, moving):
        for i in range(n):
            if moving:
                k.set_camera((300.0 * i, 0.0, -15000.0))
                k.set_scene_info(pathTracingIteration=0)
            else:
                k.set_scene_info(pathTracingIteration=i)
            yield i

    try:
        for moving in (True, False):
            n = 9 if moving else 14
            L.SolRx_SetFramesInFlight(1)
            expected = []
            for i in frames(n, moving):
                expected.append(k.render().copy())
            assert any(not np.array_equal(expected[0], e) for e in expected[1:])
            L.SolRx_SetFramesInFlight(flights)
            lag = flights - 1
            caller = np.zeros((H, W, 3), np.uint8)
            for i in frames(n, moving):
                if i % 2:
                    assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0     # ... which also copies what was delivered
                else:
                    assert L.SolRx_Render(0.0) == 0
                if i >= lag:
                    assert np.array_equal(delivered(), expected[i - lag]), (moving, i)
                    if i % 2:
                        assert np.array_equal(caller, expected[i - lag])
            assert L.SolRx_FlushFrames() == 0
            assert np.array_equal(delivered(), expected[n - 1]), moving
            k.check(0, "frames in flight")
        # picking still sees the newest frame's ids
        L.SolRx_SetFramesInFlight(1)
        k.set_scene_info(pathTracingIteration=0)
        k.render()
        ids = k.primitive_ids().copy()
        L.SolRx_SetFramesInFlight(flights)
        L.SolRx_Render(0.0)
        assert np.array_equal(k.primitive_ids(), ids)
    finally:
        L.SolRx_SetFramesInFlight(1)
        k.finalize()


@pytest.mark.parametrize("scene", ["molecule-deep-lists", "mesh-with-mirrors", "cornell", "sticks"])
def test_the_engines_default_state_frame_after_frame(solr, oracle, scene):
    """tests/conftest.py builds the order-free lists with the FIRST frame of a scene so that one frame per test walks them.
    Here the engine is left as a host gets it (SOLR_HIP_FREE_AFTER unset: the first frame after an upload walks the
    reference's order, the lists - and their thin and sorted copies - arrive with the second), and four frames of a scene
    with a camera move in between are each the oracle's frame: lists long enough for the three-bank node loop, a mesh
    with mirrors (bounce rays), the Cornell room (thin leaves), spheres and cylinders on a short list"""
    import os
    import scenes_extra
    hip = solr.hip_lib()
    saved = os.environ.pop("SOLR_HIP_FREE_AFTER", None)
    try:
        k = solr.Kernel(engine="hip")
        if scene == "molecule-deep-lists":
            solr.scenes.molecule(k, atoms=2500, width=160, height=104, iterations=2)
        elif scene == "mesh-with-mirrors":
            solr.scenes.height_field(k, n=40, width=160, height=104)
        elif scene == "cornell":
            solr.scenes.cornell(k, width=160, height=104, iterations=3)
        else:
            scenes_extra.sticks(k, width=128, height=80)
        eye = np.array(k.frame_parameters()[2], np.float32)
        for frame in range(4):
            if frame == 2:                      # the camera moves: the resident scene and its lists stay
                k.set_camera((float(eye[0]) + 700.0, float(eye[1]) + 150.0, float(eye[2])))
            pp, ids, rgb = gpu_frame(k)
            if scene in ("molecule-deep-lists", "mesh-with-mirrors", "cornell"):
                assert (hip.solr_hip_order_free_nodes() > 0) == (frame >= 1), frame
            # (the oracle as pinned: at most two pixels behind a mis-rounded powf outside the bar, helpers)
            assert_frame_pinned(k, oracle, (pp, ids, rgb), 2, "%s, frame %d of the default state" % (scene, frame))
        k.finalize()
    finally:
        if saved is not None:
            os.environ["SOLR_HIP_FREE_AFTER"] = saved


@pytest.mark.gpu
def test_run_kernel_delivers_into_the_callers_array_and_getbitmap_follows(solr):
    """SolR_RunKernel (SolRStub.cpp:154-164: render, copy to m_bitmap, copy m_bitmap to the caller) delivers the device's
    image straight into the caller's array; m_bitmap - what getBitmap / SolRx_GetBitmap return - is brought up to date
    when somebody asks.  Same bytes in both as the two-step protocol gives, frame after frame, through a reshape, and
    when the device goes away with the newest image only in the caller's hands."""
    import ctypes as C
    W, H = 200, 136
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    L = k.L

    def bitmap(w, h):
        ptr = L.SolRx_GetBitmap()
        return np.frombuffer((C.c_ubyte * (w * h * 3)).from_address(ptr), np.uint8).reshape(h, w, 3).copy()

    try:
        for i in range(4):
            k.set_camera((250.0 * i, 0.0, -15000.0))
            assert L.SolRx_Render(0.0) == 0
            two_steps = bitmap(W, H)
            caller = np.full((H, W, 3), 7, np.uint8)
            assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0
            assert np.array_equal(caller, two_steps), i
            if i % 2:                        # asked at once, or only after the next frame
                assert np.array_equal(bitmap(W, H), two_steps), i
        assert two_steps.any()
        # the caller holds the only copy when the device is released: m_bitmap is fetched before it goes
        k.set_camera((900.0, 0.0, -15000.0))
        caller = np.zeros((H, W, 3), np.uint8)
        assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0
        assert not np.array_equal(caller, two_steps)
        k.L.SolRx_SetGpuCount(1)             # (a no-op for one device; getBitmap below is what fetches)
        assert np.array_equal(bitmap(W, H), caller)
        assert L.SolR_RunKernel(0.0, None) == -1
        k.check(0, "SolR_RunKernel")
    finally:
        k.finalize()


def test_deep_bounces_keep_three_stack_slots_in_lds_and_the_rest_in_hbm(solr, oracle):
    """A frame that may bounce more than SOLR_LDS_STACK_SLOTS = 3 times (ten here: what every accumulation pass asks for)
    used to size the per-lane colour stack in LDS for it - 67 dwords, nine waves per CU.  The lean kernels now have an
    instantiation (F_STACK) that keeps three slots in LDS and the deeper ones in a per-pixel buffer in HBM: the same
    frame bit for bit as with the whole stack in LDS (variant 7), the oracle's frame, with two frames in flight (a
    buffer per set) and on strips"""
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=200, height=136, iterations=10)
    try:
        hip.solr_hip_set_variant(0)
        frame = [np.array(a, copy=True) for a in gpu_frame(k)]
        deep = int((frame[1][..., 1] > 3).sum())
        assert deep > 50, "hardly a ray of this frame goes beyond the LDS slots (%d)" % deep
        assert_frame_pinned(k, oracle, frame, 2, "ten bounces, deep slots in HBM")
        hip.solr_hip_set_variant(7)
        whole = [np.array(a, copy=True) for a in gpu_frame(k)]
        hip.solr_hip_set_variant(0)
        assert np.array_equal(frame[0].view(np.uint32), whole[0].view(np.uint32))
        assert np.array_equal(frame[1], whole[1]) and np.array_equal(frame[2], whole[2])
        # the deep slots are never zeroed: with NaNs in all of them before the launch (variant 10) the frame is the same -
        # no lane reads a slot the frame has not written
        hip.solr_hip_set_variant(10)
        poisoned = [np.array(a, copy=True) for a in gpu_frame(k)]
        hip.solr_hip_set_variant(0)
        assert np.array_equal(frame[0].view(np.uint32), poisoned[0].view(np.uint32))
        assert np.array_equal(frame[1], poisoned[1]) and np.array_equal(frame[2], poisoned[2])
        # two frames in flight: every buffer set has deep slots of its own
        hip.solr_hip_set_frames_in_flight(2)
        for i in range(6):
            again = gpu_frame(k)
            assert np.array_equal(again[0].view(np.uint32), frame[0].view(np.uint32)), i
            assert np.array_equal(again[2], frame[2]), i
        hip.solr_hip_set_frames_in_flight(1)
        # a strip: the deep buffer is the strip's
        hip.solr_hip_set_strip(40, 56)
        k.render()
        k.check(0, "strip")
        pp = np.zeros((56, 200, 8), np.float32)
        import ctypes as C
        hip.solr_hip_d2h_postprocessing(C.c_void_p(pp.ctypes.data))
        assert np.array_equal(pp.view(np.uint32), frame[0][40:96].view(np.uint32))
        hip.solr_hip_set_strip(0, -1)
    finally:
        hip.solr_hip_set_variant(0)
        hip.solr_hip_set_frames_in_flight(1)
        hip.solr_hip_set_strip(0, -1)
        k.finalize()


def _room_with_a_view(k, width=200, height=136, iterations=3, angles=(0.0, 0.0, 0.0), eye=(0.0, 0.0, -15000.0), **info):
    """the Cornell room seen from `eye` with the camera turned by `angles`"""
    solr_mod.scenes.cornell(k, width=width, height=height, iterations=iterations, **info)
    k.set_camera(eye, angles=angles)


@pytest.mark.parametrize("view", ["straight", "turned", "from-a-corner", "outside-the-room", "deep"],)
def test_thin_leaves_of_plain_planes_change_nothing(solr, oracle, view):
    """The reference gives a wall the box p0 +- size in all three axes - half the room - and every ray entered all six
    walls.  Long rays without a zero direction component walk a copy of the lists in which such leaves are as thin as
    their planes (rt_device.h tightRay): the frame is the frame on the reference's boxes (variant 8) bit for bit, and
    the oracle's - straight on (a column and a row of rays have a zero component: those waves take the reference's
    boxes), turned, from a corner of the room, from outside it, ten bounces deep."""
    hip = solr.hip_lib()
    kw = {"straight": dict(), "turned": dict(angles=(0.21, -0.37, 0.11)),
          "from-a-corner": dict(eye=(-17000.0, 30000.0, -17000.0), angles=(0.5, 0.7, 0.0)),
          "outside-the-room": dict(eye=(3000.0, 2000.0, -45000.0)), "deep": dict(iterations=10, angles=(0.1, 0.2, 0.0))}[view]
    k = solr.Kernel(engine="hip")
    _room_with_a_view(k, **kw)
    try:
        hip.solr_hip_set_variant(0)
        thin = [np.array(a, copy=True) for a in gpu_frame(k)]
        assert_frame_pinned(k, oracle, thin, 2, "thin leaves, " + view)
        hip.solr_hip_set_variant(8)
        fat = [np.array(a, copy=True) for a in gpu_frame(k)]
        assert np.array_equal(thin[0].view(np.uint32), fat[0].view(np.uint32))
        assert np.array_equal(thin[1], fat[1]) and np.array_equal(thin[2], fat[2])
        # ... and the thin copies were there to be walked (the census counts leaf tests on the reference's list: the
        # walk record says which list a walk took)
        hip.solr_hip_set_variant(0)
    finally:
        hip.solr_hip_set_variant(0)
        k.finalize()


def test_thin_leaves_follow_a_rotation_on_the_device(solr, oracle):
    """device-side rotations refit the lists and rebuild the leaf records: the thin copies are made again from the
    rotated planes (a rotated axis plane is still tested as an axis plane at its new p0)"""
    import ctypes as C
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    _room_with_a_view(k, angles=(0.1, 0.3, 0.0))
    try:
        k.render()
        for step in range(3):
            k.rotate_primitives((0.0, 1000.0, 0.0), (0.02 * (step + 1), 0.05, -0.03))
            hip.solr_hip_set_variant(0)
            thin = [np.array(a, copy=True) for a in gpu_frame(k)]
            hip.solr_hip_set_variant(8)
            fat = [np.array(a, copy=True) for a in gpu_frame(k)]
            hip.solr_hip_set_variant(0)
            assert np.array_equal(thin[0].view(np.uint32), fat[0].view(np.uint32)), step
            assert np.array_equal(thin[1], fat[1]) and np.array_equal(thin[2], fat[2]), step
        assert_frame_pinned(k, oracle, thin, 2, "thin leaves after rotations")
    finally:
        hip.solr_hip_set_variant(0)
        k.finalize()


@pytest.mark.gpu
def test_one_frame_at_a_time_the_image_leaves_in_bands_while_the_kernel_renders(solr):
    """render_begin / render_end and SolR_RunKernel one frame at a time (solr_hip_stream_next_image; csrc/renderer.h,
    ImageStreaming): the frame's waves count themselves into tile rows and bands, every band's copy waits for the band's
    word instead of the kernel.  The bytes that arrive - in m_bitmap, in the caller's array - are the device's image as
    d2h_bitmap reads it after the kernel: with a moving camera, through refinement and
    accumulation passes, around a frame with a neighbourhood post-process (not streamed), through a reshape (new
    counters), after frames in flight were on and off again."""
    import ctypes as C
    hip = solr.hip_lib()
    W, H = 200, 136
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    solr.scenes.cornell(k, width=W, height=H, iterations=2, maxPathTracingIterations=40)
    L = k.L

    def delivered(w, h):
        ptr = L.SolRx_GetBitmap()
        return np.frombuffer((C.c_ubyte * (w * h * 3)).from_address(ptr), np.uint8).reshape(h, w, 3).copy()

    def on_device(w, h):
        """the image of the frame rendered last, read back behind the kernel (d2h_bitmap)"""
        rgb, ids = np.zeros((h, w, 3), np.uint8), np.zeros((h, w, 4), np.int32)
        si = k.frame_parameters()[0]
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data))
        k.check(0, "solr_hip_d2h")
        return rgb

    def in_bands():
        return hip.solr_hip_stream_next_image(-2)

    try:
        if hip.solr_hip_stream_next_image(0) != 1:
            pytest.skip("SOLR_HIP_NO_IMAGE_STREAMING=1")
        assert L.SolRx_Render(0.0) == 0      # (the frame that brings the device up is not streamed)
        # (a frame whose longest tile the cost-ordered launch would split into quadrant waves is not streamed either -
        # this small one would be, now and then; the 1080p test below runs with the order as the engine chooses it)
        hip.solr_hip_set_tile_scheduling(0)
        # a moving camera; both entry points
        frames = []
        for i in range(8):
            k.set_camera((250.0 * i, 40.0 * i, -15000.0))
            k.set_scene_info(pathTracingIteration=0)
            before = in_bands()
            if i % 2:
                caller = np.full((H, W, 3), 7, np.uint8)
                assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0
                streamed = caller
            else:
                assert L.SolRx_Render(0.0) == 0
                streamed = delivered(W, H)
            assert in_bands() == before + 1, i
            assert np.array_equal(streamed, on_device(W, H)), i
            assert np.array_equal(delivered(W, H), streamed), i      # (getBitmap follows SolR_RunKernel's frame)
            frames.append(streamed.copy())
        assert frames[0].any() and not np.array_equal(frames[0], frames[7])
        # the C ABI's own route with the primitive ids (d2h_bitmap hands both over): the same image, the same ids
        for i in range(4):
            k.set_camera((250.0 * i, 40.0 * i, -15000.0))
            k.set_scene_info(pathTracingIteration=0)
            assert L.SolRx_Render(0.0) == 0                    # (the host protocol's frame: its parameters are what follows)
            si, ppi, eye, direction, angles = k.frame_parameters()
            flat = k.flat_scene()
            objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            before = in_bands()
            assert hip.solr_hip_stream_next_image(2) == 1
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            rgb, ids = np.zeros((H, W, 3), np.uint8), np.full((H, W, 4), -7, np.int32)
            assert hip.solr_hip_d2h_streamed(C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data)) == 1
            assert in_bands() == before + 1
            rgb2, ids2 = np.zeros((H, W, 3), np.uint8), np.zeros((H, W, 4), np.int32)
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb2.ctypes.data), C.c_void_p(ids2.ctypes.data))
            assert np.array_equal(rgb, rgb2) and np.array_equal(ids, ids2) and np.array_equal(rgb, frames[i]), i
            # asked for the image only, the ids are not to be had in bands: 0, nothing copied
            hip.solr_hip_stream_next_image(1)
            hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles))
            assert hip.solr_hip_d2h_streamed(C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data)) == 0
            assert hip.solr_hip_d2h_streamed_image(C.c_void_p(rgb.ctypes.data)) == 1 and np.array_equal(rgb, rgb2)
        k.check(0, "the ids in bands")
        # the counters only ever grow and are zeroed when a row's count nears 2^32 (4.4 million 1080p frames): variant 14
        # zeroes them every third frame
        hip.solr_hip_set_variant(14)
        before = in_bands()
        for i in range(8):
            k.set_camera((250.0 * i, 40.0 * i, -15000.0))
            assert L.SolRx_Render(0.0) == 0
            assert np.array_equal(delivered(W, H), frames[i]), i
            assert np.array_equal(delivered(W, H), on_device(W, H)), i
        assert in_bands() == before + 8
        hip.solr_hip_set_variant(0)
        # refinement and accumulation passes
        passes = []
        before = in_bands()
        for i in range(14):
            k.set_scene_info(pathTracingIteration=i)
            assert L.SolRx_Render(0.0) == 0
            passes.append(delivered(W, H))
            assert np.array_equal(passes[i], on_device(W, H)), i
        assert in_bands() == before + 14
        assert any(not np.array_equal(passes[0], e) for e in passes[1:])
        for i in range(14):                                            # ... the same passes, the other entry point
            k.set_scene_info(pathTracingIteration=i)
            assert np.array_equal(k.render(), passes[i]), i
        # a frame whose image a post-process kernel writes is not streamed; the one after it is
        k.set_scene_info(pathTracingIteration=0)
        k.set_post_processing(solr.ppe_ambientOcclusion, 0.0, 4000.0, 40)
        before = in_bands()
        assert L.SolRx_Render(0.0) == 0
        assert in_bands() == before
        ao = delivered(W, H)
        assert np.array_equal(ao, on_device(W, H))
        k.set_post_processing(solr.ppe_none)
        assert L.SolRx_Render(0.0) == 0
        assert in_bands() == before + 1
        assert np.array_equal(delivered(W, H), passes[0]) and not np.array_equal(ao, passes[0])
        # frames in flight on, off: one at a time streams again
        L.SolRx_SetFramesInFlight(2)
        for _ in range(3):
            assert L.SolRx_Render(0.0) == 0
        L.SolRx_SetFramesInFlight(1)
        before = in_bands()
        assert L.SolRx_Render(0.0) == 0
        assert in_bands() == before + 1 and np.array_equal(delivered(W, H), passes[0])
        # another size: the counters are made anew (and a frame with fewer than sixteen tile rows is not streamed)
        k.set_scene_info(width=320, height=240)
        before = in_bands()
        for i in range(3):
            k.set_camera((100.0 * i, 0.0, -15000.0))
            assert L.SolRx_Render(0.0) == 0
            assert np.array_equal(delivered(320, 240), on_device(320, 240)), i
        assert in_bands() == before + 3
        k.set_scene_info(width=160, height=96)
        before = in_bands()
        assert L.SolRx_Render(0.0) == 0
        assert in_bands() == before and np.array_equal(delivered(160, 96), on_device(160, 96))
        k.check(0, "image streaming")
    finally:
        hip.solr_hip_set_variant(0)
        hip.solr_hip_set_tile_scheduling(1)
        L.SolRx_SetFramesInFlight(1)
        k.finalize()


@pytest.mark.gpu
@pytest.mark.parametrize("scene", ["cornell", "height_field", "molecule"])
def test_full_size_frames_leave_in_bands_with_the_same_bytes(solr, scene):
    """... at 1920x1080, forty frames, both entry points in turn: the Cornell box under a camera that moves and the
    molecule (launched band after band, by cost inside a band), and the 100k-triangle mesh, whose horizon tiles the
    cost-ordered launch renders as four quadrant waves each and first of all - such a frame keeps that order and is read
    back behind the kernel"""
    import ctypes as C
    hip = solr.hip_lib()
    W, H = 1920, 1080
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    kw = dict(width=W, height=H)
    if scene == "cornell":
        kw["iterations"] = 3
    getattr(solr.scenes, scene)(k, **kw)
    L = k.L
    try:
        if hip.solr_hip_stream_next_image(0) != 1:
            pytest.skip("SOLR_HIP_NO_IMAGE_STREAMING=1")
        before = hip.solr_hip_stream_next_image(-2)
        different = 0
        last = None
        plain, ids = np.zeros((H, W, 3), np.uint8), np.zeros((H, W, 4), np.int32)
        caller = np.zeros((H, W, 3), np.uint8)
        for i in range(40):
            if scene == "cornell":
                k.set_camera((40.0 * i, 10.0 * i, -15000.0))
            if i % 2:
                assert L.SolR_RunKernel(0.0, caller.ctypes.data) == 0
                streamed = caller
            else:
                assert L.SolRx_Render(0.0) == 0
                ptr = L.SolRx_GetBitmap()
                streamed = np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3)
            si = k.frame_parameters()[0]
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(plain.ctypes.data), C.c_void_p(ids.ctypes.data))
            assert np.array_equal(streamed, plain), i
            if last is not None and not np.array_equal(last, plain):
                different += 1
            last = plain.copy()
        in_bands = hip.solr_hip_stream_next_image(-2) - before
        assert (in_bands <= 8) if scene == "height_field" else (in_bands >= 36), in_bands
        assert last.any() and (scene != "cornell" or different > 30)
        k.check(0, "image streaming at 1080p")
    finally:
        k.finalize()


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["SolRx_Render", "SolR_RunKernel"])
def test_a_band_whose_word_never_comes_is_copied_when_the_kernel_has_ended(solr, entry):
    """variant 13: the waves of a streamed frame write no band's word.  The host, which watches the words, also watches the
    event behind the kernel: when that has fired everything the kernel wrote is in memory, and the bands are copied then -
    the right bytes, no wait without end, whatever becomes of a word."""
    import ctypes as C
    import time
    hip = solr.hip_lib()
    W, H = 200, 136
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    solr.scenes.cornell(k, width=W, height=H, iterations=2)

    def frame(x):
        k.set_camera((x, 0.0, -15000.0))
        if entry == "SolR_RunKernel":
            out = np.zeros((H, W, 3), np.uint8)
            assert k.L.SolR_RunKernel(0.0, out.ctypes.data) == 0
        else:
            assert k.L.SolRx_Render(0.0) == 0
            ptr = k.L.SolRx_GetBitmap()
            out = np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3).copy()
        rgb, ids = np.zeros((H, W, 3), np.uint8), np.zeros((H, W, 4), np.int32)
        si = k.frame_parameters()[0]
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb.ctypes.data), C.c_void_p(ids.ctypes.data))
        assert np.array_equal(out, rgb), x
        return out

    try:
        frame(0.0)
        if hip.solr_hip_stream_next_image(0) != 1:
            pytest.skip("SOLR_HIP_NO_IMAGE_STREAMING=1")
        hip.solr_hip_set_tile_scheduling(0)
        before = hip.solr_hip_stream_next_image(-2)
        first = frame(100.0)
        hip.solr_hip_set_variant(13)
        t0 = time.perf_counter()
        second = frame(200.0)
        assert time.perf_counter() - t0 < 2.0
        hip.solr_hip_set_variant(0)
        third = frame(300.0)                                             # (the counters are a frame out of step: the same way out)
        assert hip.solr_hip_stream_next_image(-2) == before + 3
        assert not np.array_equal(first, second) and not np.array_equal(second, third)
        k.check(0, "a band without its word")
    finally:
        hip.solr_hip_set_variant(0)
        hip.solr_hip_set_tile_scheduling(1)
        k.finalize()
