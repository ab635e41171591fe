"""The oracle against hand-derived known answers.

The reference ships no tests or golden vectors for this path.  Whole frames
of the oracle are pinned against the reference's own OpenCL renderer in test_reference_opencl.py
(GPU, image level); here each restated function is pinned, on CPU, to results that follow from the
reference's formulas by hand.  Every expected value below is derived in the comment next to it from
the cited reference lines.
"""
import ctypes as C
import importlib
import math

import numpy as np
import pytest

solr = importlib.import_module("sol-r_amd")


def scene_info(**kw):
    si = solr.SceneInfo()
    si.size_x, si.size_y = 8, 8
    si.graphicsLevel = solr.glFull
    si.nbRayIterations = 1
    si.transparentColor = 2.0
    si.viewDistance = 50000.0
    si.shadowIntensity = 1.0
    si.extendedGeometry = 1
    si.skyboxMaterialId = solr.MATERIAL_NONE
    si.geometryEpsilon = 0.001
    si.rayEpsilon = 0.05
    for k, v in kw.items():
        setattr(si, k, v)
    return si


def prim(ptype, p0=(0, 0, 0), p1=(0, 0, 0), p2=(0, 0, 0), n0=(0, 0, 0), n1=(0, 0, 0), n2=(0, 0, 0), size=(0, 0, 0),
         material=0, index=0):
    a = np.zeros(1, solr.PRIMITIVE_DTYPE)
    a["p0"], a["p1"], a["p2"], a["n0"], a["n1"], a["n2"], a["size"] = p0, p1, p2, n0, n1, n2, size
    a["type"], a["materialId"], a["index"] = ptype, material, index
    return a


def materials(n=4, **fields):
    m = np.zeros(n, solr.MATERIAL_DTYPE)
    m["color"] = (0.5, 0.5, 0.5, 0.0)
    m["textureIds"] = (-1, -1, -1, -1)
    m["advancedTextureIds"] = (-1, -1, -1, -1)
    for k, v in fields.items():
        m[k] = v
    return m


def fa(*v):
    return np.array(v, dtype=np.float32)


def intersect(oracle, si, p, mats, origin, direction, shadows=0):
    L = oracle.lib()
    o, d = fa(*origin), fa(*direction)
    inter, normal, areas = fa(0, 0, 0), fa(0, 0, 0), fa(0, 0, 0)
    shadow = C.c_float(0)
    hit = L.oracle_primitive_intersection(C.addressof(si), p.ctypes.data, mats.ctypes.data, None, o.ctypes.data,
                                          d.ctypes.data, shadows, inter.ctypes.data, normal.ctypes.data,
                                          areas.ctypes.data, C.addressof(shadow))
    return hit, inter, normal, areas, shadow.value


# ---------------------------------------------------------------- boxIntersection (GI:52-79)
def box(lo, hi):
    b = np.zeros(1, solr.BOX_DTYPE)
    b["min"], b["max"] = lo, hi
    return b


BOX_CASES = [
    ((-10, 0, 0), (20, 0, 0), 1e9, 1),    # tmin = (-1+10)/20 = .45, tmax = .55, y/z slabs: dir 0 -> inv 1: (-1-0)*1=-1..1
    ((-10, 0, 0), (20, 0, 0), 0.4, 0),    # tmin .45 is not < t1 = .4
    ((-10, 5, 0), (20, 0, 0), 1e9, 0),    # y slab: inv.y = 1 (zero direction rule, GI:38-40): tymin = -6, tymax = -4 < tmin .45
    ((10, 0, 0), (-20, 0, 0), 1e9, 1),    # negative direction uses parameters[1] for tmin (GI:56)
    ((0, 0, 0), (1, 1, 1), 1e9, 1),       # origin inside: tmin -1, tmax 1 > t0 = 0
    ((-10, 0, 0), (-20, 0, 0), 1e9, 0),   # box behind the ray: tmax = (-1+10)/-20 < 0
]


def test_box_zero_direction_quirk(oracle):
    # A ray exactly parallel to z: x and y slabs are evaluated with inv = 1 (GI:38-40), giving [-1, 1];
    # the z slab gives [4, 6]; tzmin 4 > tmax 1 -> miss (GI:71) although the ray geometrically pierces the box.
    b = box((-1, -1, 4), (1, 1, 6))
    L = oracle.lib()
    o, d1, d2 = fa(0, 0, 0), fa(0, 0, 1), fa(1e-3, 1e-3, 1)   # keep the arrays alive across the calls
    assert L.oracle_box_intersection(b.ctypes.data, o.ctypes.data, d1.ctypes.data, 0.0, 1e9) == 0
    assert L.oracle_box_intersection(b.ctypes.data, o.ctypes.data, d2.ctypes.data, 0.0, 1e9) == 1


@pytest.mark.parametrize("origin,direction,t1,expected", BOX_CASES)
def test_box_intersection(oracle, origin, direction, t1, expected):
    b = box((-1, -1, -1), (1, 1, 1))
    o, d = fa(*origin), fa(*direction)
    assert oracle.lib().oracle_box_intersection(b.ctypes.data, o.ctypes.data, d.ctypes.data, 0.0, t1) == expected


# ---------------------------------------------------------------- sphereIntersection (GI:220-284)
def test_sphere_front_hit(oracle):
    si = scene_info()
    p = prim(solr.ptSphere, p0=(0, 0, 0), size=(2, 2, 2))
    # dir normalised = (0,0,1); a = 2, b = 2*dot((0,0,-10),(0,0,1)) = -20, c = 100-4 = 96, d = 400-4*96 = 16, r = 4
    # t1 = (20-4)/2 = 8, t2 = 12 -> t = 8 -> intersection (0,0,-2), normal (0,0,-1), opaque -> shadowIntensity 1
    hit, inter, normal, areas, shadow = intersect(oracle, si, p, materials(), (0, 0, -10), (0, 0, 5))
    assert hit == 1
    assert np.array_equal(inter, fa(0, 0, -2))
    assert np.array_equal(normal, fa(0, 0, -1))
    assert shadow == 1.0


def test_sphere_from_inside_flips_normal(oracle):
    si = scene_info()
    p = prim(solr.ptSphere, p0=(0, 0, 0), size=(2, 2, 2))
    # origin at the centre: b = 0, c = -4, d = 16, r = 4, t1 = -2 <= eps -> t = t2 = 2, back = true -> normal * -1
    hit, inter, normal, _, _ = intersect(oracle, si, p, materials(), (0, 0, 0), (0, 0, 1))
    assert hit == 1
    assert np.array_equal(inter, fa(0, 0, 2))
    assert np.array_equal(normal, fa(0, 0, -1))


def test_sphere_miss_and_behind(oracle):
    si = scene_info()
    p = prim(solr.ptSphere, p0=(0, 0, 0), size=(2, 2, 2))
    assert intersect(oracle, si, p, materials(), (0, 3, -10), (0, 0, 1))[0] == 0   # d = 0*0 - 4*(109-4) < 0
    assert intersect(oracle, si, p, materials(), (0, 0, 10), (0, 0, 1))[0] == 0    # both roots negative (GI:240)


def test_sphere_transparent_shadow_intensity(oracle):
    si = scene_info()
    p = prim(solr.ptSphere, p0=(0, 0, 0), size=(2, 2, 2))
    mats = materials(transparency=0.5)
    # head-on: dot(dir, normal) = -1 -> shadowIntensity = 1 - |r| = 0 (GI:280-281)
    hit, _, _, _, shadow = intersect(oracle, si, p, mats, (0, 0, -10), (0, 0, 1))
    assert hit == 1 and shadow == 0.0


# ---------------------------------------------------------------- planeIntersection (GI:424-567)
def test_xy_plane_both_sides(oracle):
    si = scene_info()
    p = prim(solr.ptXYPlane, p0=(0, 0, 5), n0=(0, 0, 1), size=(2, 2, 0))
    # from z < plane with dir.z > 0: second branch (GI:529-540): normal negated, hit at (.5,.5,5)
    hit, inter, normal, _, shadow = intersect(oracle, si, p, materials(), (0.5, 0.5, 0), (0, 0, 1))
    assert hit == 1 and np.array_equal(inter, fa(0.5, 0.5, 5)) and np.array_equal(normal, fa(0, 0, -1))
    assert shadow == 1.0
    # from z > plane with dir.z < 0: first branch, normal as stored
    hit, inter, normal, _, _ = intersect(oracle, si, p, materials(), (0.5, 0.5, 10), (0, 0, -1))
    assert hit == 1 and np.array_equal(normal, fa(0, 0, 1))
    # outside the rectangle: |x - p0.x| = 3 is not < size.x = 2
    assert intersect(oracle, si, p, materials(), (3, 0, 0), (0, 0, 1))[0] == 0
    # moving away
    assert intersect(oracle, si, p, materials(), (0, 0, 0), (0, 0, -1))[0] == 0


def test_plane_colour_key_transparency(oracle):
    # GI:561: (r+g+b)/3 >= transparentColor makes the plane fully transparent; the viewer's default 0 hides every plane
    p = prim(solr.ptXZPlane, p0=(0, 0, 0), n0=(0, 1, 0), size=(5, 0, 5))
    assert intersect(oracle, scene_info(transparentColor=0.0), p, materials(), (0, 3, 0), (0, -1, 0))[0] == 0
    assert intersect(oracle, scene_info(transparentColor=0.51), p, materials(), (0, 3, 0), (0, -1, 0))[0] == 1
    assert intersect(oracle, scene_info(transparentColor=0.5), p, materials(), (0, 3, 0), (0, -1, 0))[0] == 0


def test_yz_plane_chessboard_lights(oracle):
    # GI:486-490: an emissive YZ plane only exists where int(|z|) % 4000 < 2000 and int(|y|) % 4000 < 2000
    p = prim(solr.ptYZPlane, p0=(0, 0, 0), n0=(1, 0, 0), size=(0, 10000, 10000))
    mats = materials(innerIllumination=(1.0, 0, 0, 0))
    si = scene_info()
    assert intersect(oracle, si, p, mats, (5, 100, 100), (-1, 0, 0))[0] == 1
    assert intersect(oracle, si, p, mats, (5, 2500, 100), (-1, 0, 0))[0] == 0
    assert intersect(oracle, si, p, mats, (5, 100, 6100), (-1, 0, 0))[0] == 0


def test_checkerboard_is_single_sided(oracle):
    p = prim(solr.ptCheckboard, p0=(0, 0, 0), n0=(0, 1, 0), size=(5, 0, 5))
    si = scene_info()
    assert intersect(oracle, si, p, materials(), (0, 3, 0), (0, -1, 0))[0] == 1
    assert intersect(oracle, si, p, materials(), (0, -3, 0), (0, 1, 0))[0] == 0   # no second branch (GI:435-446)


# ---------------------------------------------------------------- triangleIntersection (GI:575-659)
TRI = dict(p0=(0, 0, 0), p1=(1, 0, 0), p2=(0, 1, 0), n0=(0, 0, -1), n1=(0, 0, -1), n2=(0, 0, -1))


def test_triangle_hit_and_areas(oracle):
    si = scene_info()
    p = prim(solr.ptTriangle, **TRI)
    # P = dir x E03 = (0,0,1)x(0,1,0) = (-1,0,0); det = E01.P = -1; T = (.25,.25,-1); a = T.P/det = .25
    # Q = T x E01 = (0*0-(-1)*0, (-1)*1-.25*0, .25*0-.25*1) = (0,-1,-.25); b = dir.Q/det = .25; t = E03.Q/det = 1
    hit, inter, normal, areas, _ = intersect(oracle, si, p, materials(), (0.25, 0.25, -1), (0, 0, 1))
    assert hit == 1
    assert np.array_equal(inter, fa(0.25, 0.25, 0))
    # sub-triangle areas (GI:632-634): opposite p0: .5*|v1 x v2| = .25, opposite p1: .125, opposite p2: .125
    assert np.array_equal(areas, fa(0.25, 0.125, 0.125))
    # interpolated normal (0,0,-1); dot(dir, normal) = -1 is not > 0 -> kept
    assert np.array_equal(normal, fa(0, 0, -1))


def test_triangle_normal_faces_the_ray(oracle):
    p = prim(solr.ptTriangle, **TRI)
    hit, _, normal, _, _ = intersect(oracle, scene_info(), p, materials(), (0.25, 0.25, 1), (0, 0, -1))
    assert hit == 1 and np.array_equal(normal, fa(0, 0, 1))   # r = dot(dir, n) = 1 > 0 -> normal *= -1 (GI:653-654)


def test_triangle_second_half_of_the_parallelogram_is_rejected(oracle):
    # a + b > 1 enters GI:601-617 where E21 = p1 - p1 = 0 => det_ = 0 < geometryEpsilon => miss
    p = prim(solr.ptTriangle, **TRI)
    assert intersect(oracle, scene_info(), p, materials(), (0.75, 0.75, -1), (0, 0, 1))[0] == 0
    assert intersect(oracle, scene_info(), p, materials(), (0.49, 0.49, -1), (0, 0, 1))[0] == 1


def test_triangle_double_sided_shadow_quirk(oracle):
    # GI:643-647: with doubleSidedTriangles every shadow test misses (dangling else)
    p = prim(solr.ptTriangle, **TRI)
    si = scene_info(doubleSidedTriangles=1)
    assert intersect(oracle, si, p, materials(), (0.25, 0.25, -1), (0, 0, 1), shadows=1)[0] == 0
    assert intersect(oracle, si, p, materials(), (0.25, 0.25, 1), (0, 0, -1), shadows=1)[0] == 0
    assert intersect(oracle, si, p, materials(), (0.25, 0.25, -1), (0, 0, 1), shadows=0)[0] == 1


# ---------------------------------------------------------------- cylinderIntersection (GI:293-349)
def test_cylinder_side_hit(oracle):
    # axis (0,1,0) from y=-1 to y=1, radius .5 (size.x = size.y = .5), centre p2 = origin
    p = prim(solr.ptCylinder, p0=(0, -1, 0), p1=(0, 1, 0), p2=(0, 0, 0), n1=(0, 1, 0), size=(0.5, 0.5, 0.5))
    # dir (1,0,0): n = dir x axis = (0,0,1), ln = 1; d = |O_C.n| = 0; O = O_C x axis = (-5,1,0)x(0,1,0) = (0,0,-5)
    # t = -O.n/ln = 5; O' = normalize(n x axis) = (-1,0,0); s = |sqrt(.25)/dot(dir,O')| = .5; t1 = 4.5 -> x = -0.5
    hit, inter, normal, _, shadow = intersect(oracle, scene_info(), p, materials(), (-5, 0, 0), (1, 0, 0))
    assert hit == 1 and np.array_equal(inter, fa(-0.5, 0, 0)) and np.array_equal(normal, fa(-1, 0, 0))
    assert shadow == 1.0
    # beyond the caps: both candidate points fail the scale test (GI:330-340)
    assert intersect(oracle, scene_info(), p, materials(), (-5, 3, 0), (1, 0, 0))[0] == 0
    # parallel to the axis: ln = 0 (GI:303)
    assert intersect(oracle, scene_info(), p, materials(), (0, -5, 0), (0, 1, 0))[0] == 0


# ---------------------------------------------------------------- ellipsoidIntersection (GI:159-212)
def test_ellipsoid_axis_hit(oracle):
    p = prim(solr.ptEllipsoid, p0=(0, 0, 0), size=(2, 1, 1))
    # along x from -10: a = 1/4, b = 2*(-10)/4 = -5, c = 100/4 - 1 = 24, d = 25 - 24 = 1 -> t = (5-1)/(.5) = 8 -> x = -2
    hit, inter, normal, _, shadow = intersect(oracle, scene_info(), p, materials(), (-10, 0, 0), (3, 0, 0))
    assert hit == 1 and np.array_equal(inter, fa(-2, 0, 0)) and np.array_equal(normal, fa(-1, 0, 0)) and shadow == 1.0
    # b == 0 rejects (GI:181): origin at the centre
    assert intersect(oracle, scene_info(), p, materials(), (0, 0, 0), (1, 0, 0))[0] == 0


# ---------------------------------------------------------------- vectorRotation / makeColor
def test_vector_rotation(oracle):
    L = oracle.lib()
    v, c, a = fa(1, 2, 3), fa(0, 0, 0), fa(0, 0, 0)
    L.oracle_vector_rotation(v.ctypes.data, c.ctypes.data, a.ctypes.data)
    assert np.array_equal(v, fa(1, 2, 3))              # cos 0 = 1, sin 0 = 0: exact identity
    v, a = fa(1, 0, 0), fa(0, 0, math.pi / 2)
    L.oracle_vector_rotation(v.ctypes.data, c.ctypes.data, a.ctypes.data)
    # Z step (VU:136-137): x' = x cos - y sin = cosf(pi/2), y' = x sin + y cos = sinf(pi/2) = 1
    assert np.allclose(v, (0, 1, 0), atol=1e-7) and v[1] == 1.0
    v, c, a = fa(2, 0, 0), fa(1, 0, 0), fa(0, 0, math.pi)   # rotation about a centre (VU:118-120,139-141)
    L.oracle_vector_rotation(v.ctypes.data, c.ctypes.data, a.ctypes.data)
    assert np.allclose(v, (0, 0, 0), atol=1e-6)


@pytest.mark.parametrize("colour,expected", [
    ((0.0, 0.5, 1.0), (0, 127, 255)),          # (uint8)(c * 255.f): truncation, GS:159-161
    ((0.999, 1.5, -0.25), (254, 255, 0)),      # clamp to [0,1] first, GS:135-140
    ((1 / 255 * 3, 0.99999994, 2.0), (3, 254, 255)),
])
def test_make_color_rgb(oracle, colour, expected):
    si = scene_info()
    out = np.zeros(12, np.uint8)
    c = fa(*colour)
    oracle.lib().oracle_make_color(C.addressof(si), c.ctypes.data, out.ctypes.data, 1)
    assert tuple(out[3:6]) == expected and out[:3].sum() == 0


def test_make_color_bgr_mirrors_the_row(oracle):
    si = scene_info(frameBufferType=solr.ftBGR, size_x=4, size_y=4)
    out = np.zeros(4 * 4 * 3, np.uint8)
    # index 1 -> y = 1/4 = 0, x = 1 -> i = (0+1)*4 - 1 - 1 = 2 (GS:147-150), channels stored B,G,R
    c = fa(1.0, 0.5, 0.0)
    oracle.lib().oracle_make_color(C.addressof(si), c.ctypes.data, out.ctypes.data, 1)
    assert tuple(out[6:9]) == (0, 127, 255) and out.sum() == 382


# ---------------------------------------------------------------- walks on a hand-built tree
def two_sphere_scene(oracle, second_at=10.0, transparency=0.0):
    """box 0 (lights, empty), then two leaves holding one sphere each."""
    boxes = np.zeros(3, solr.BOX_DTYPE)
    # loose in x/y on purpose: for a ray parallel to z the x/y slabs use inv = 1 (GI:38-40), i.e. the
    # "t interval" of those slabs is [lo - o, hi - o] and has to overlap the z interval for a hit
    boxes["min"] = [(-50000,) * 3, (-50, -50, 4), (-50, -50, second_at - 1)]
    boxes["max"] = [(50000,) * 3, (50, 50, 6), (50, 50, second_at + 1)]
    boxes["nbPrimitives"] = [0, 1, 1]
    boxes["startIndex"] = [0, 0, 1]
    boxes["indexForNextBox"] = [(1, 0), (1, 0), (1, 0)]
    prims = np.zeros(2, solr.PRIMITIVE_DTYPE)   # (np.concatenate would re-pack the padded record to 120 bytes)
    prims[0] = prim(solr.ptSphere, p0=(0, 0, 5), size=(1, 1, 1), index=7, material=1)[0]
    prims[1] = prim(solr.ptSphere, p0=(0, 0, second_at), size=(1, 1, 1), index=9, material=2)[0]
    mats = materials(4)
    mats["transparency"][1] = transparency
    flat = solr.FlatScene(boxes, prims, np.zeros(0, solr.LIGHT_DTYPE), 0, mats, np.zeros(0, np.float32),
                          np.zeros(0, np.uint8))
    return oracle.Scene(flat)


def closest(oracle, scene, si, origin, target, iteration=0, current_material=-2):
    L = oracle.lib()
    o, t = fa(*origin), fa(*target)
    p = C.c_int(-1)
    inter, normal, areas = fa(0, 0, 0), fa(0, 0, 0), fa(0, 0, 0)
    hit = L.oracle_closest_hit(C.byref(scene.c), C.addressof(si), o.ctypes.data, t.ctypes.data, iteration,
                               current_material, C.byref(p), inter.ctypes.data, normal.ctypes.data, areas.ctypes.data)
    return hit, p.value, inter, normal


def test_closest_hit_takes_the_nearest(oracle):
    s = two_sphere_scene(oracle)
    hit, p, inter, normal = closest(oracle, s, scene_info(), (0, 0, 0), (0, 0, 1))
    assert hit == 1 and p == 0 and np.array_equal(inter, fa(0, 0, 4)) and np.array_equal(normal, fa(0, 0, -1))
    hit, p, inter, _ = closest(oracle, s, scene_info(), (0, 0, 20), (0, 0, 19))
    assert hit == 1 and p == 1 and np.array_equal(inter, fa(0, 0, 11))


def test_closest_hit_tie_keeps_the_first_visited(oracle):
    # two identical spheres: the second distance is not < minDistance (GI:751) -> flattened index 0 wins
    s = two_sphere_scene(oracle, second_at=5.0)
    hit, p, _, _ = closest(oracle, s, scene_info(), (0, 0, 0), (0, 0, 1))
    assert hit == 1 and p == 0


def test_closest_hit_min_distance_shrinks_with_the_bounce(oracle):
    # GI:674: iteration >= 2 limits hits to viewDistance / (iteration + 1)
    s = two_sphere_scene(oracle)
    si = scene_info(viewDistance=9.0)
    assert closest(oracle, s, si, (0, 0, 0), (0, 0, 1), iteration=0)[0] == 1      # distance 4 < 9
    assert closest(oracle, s, si, (0, 0, 0), (0, 0, 1), iteration=2)[0] == 0      # distance 4 is not < 9/3


def shadow(oracle, scene, si, lamp, origin, light_id=-1, object_id=-1, iteration=0):
    colour, lamp_, origin_ = fa(0, 0, 0), fa(*lamp), fa(*origin)
    v = oracle.lib().oracle_shadow(C.byref(scene.c), C.addressof(si), lamp_.ctypes.data, origin_.ctypes.data,
                                   light_id, iteration, object_id, colour.ctypes.data)
    return v, colour


def test_shadow_opaque_occluder(oracle):
    s = two_sphere_scene(oracle)
    si = scene_info()
    v, colour = shadow(oracle, s, si, lamp=(0, 0, 20), origin=(0, 0, 0))
    assert v == 1.0 and not colour.any()                     # first occluder already saturates (GI:815)
    assert shadow(oracle, s, si, lamp=(0, 0, 3), origin=(0, 0, 0))[0] == 0.0    # lamp in front of both: l < |O_L| fails
    assert shadow(oracle, s, si, lamp=(10, 0, 0), origin=(0, 0, 0))[0] == 0.0   # nothing on the way


def test_shadow_excludes_by_original_index(oracle):
    # GI:829 compares Primitive.index (7 and 9 here) with lightId / objectId
    s = two_sphere_scene(oracle)
    si = scene_info()
    assert shadow(oracle, s, si, (0, 0, 20), (0, 0, 0), light_id=7, object_id=9)[0] == 0.0
    assert shadow(oracle, s, si, (0, 0, 20), (0, 0, 0), light_id=7)[0] == 1.0
    assert shadow(oracle, s, si, (0, 0, 20), (0, 0, 0), object_id=0)[0] == 1.0   # flattened index 0 excludes nothing


def test_shadow_through_transparent_sphere(oracle):
    # first sphere transparency .5, hit head-on: shadowIntensity = 1-|dot| = 0 -> ratio 0; the second, opaque one adds 1
    s = two_sphere_scene(oracle, transparency=0.5)
    v, colour = shadow(oracle, s, scene_info(), (0, 0, 20), (0, 0, 0))
    assert v == 1.0 and not colour.any()
    # shadowIntensity .4 < 1: the walk stops after the first opaque hit contributes 1 * .4 and clamps (GI:906)
    v, _ = shadow(oracle, s, scene_info(shadowIntensity=0.4), (0, 0, 20), (0, 0, 0), light_id=7)
    assert v == pytest.approx(0.4)


def test_rounded_transcendentals_switch(oracle):
    """oracle_set_rounded_transcendentals: off by default (libm's binary32 routines, what the reference's host
    code calls), on = evaluated in binary64 and rounded once (what the engine does, so that frames with
    procedural spheres or sphere / skybox UV maps compare exactly).  A frame without transcendentals but the
    Blinn power may move by an ULP where powf is not the correctly rounded value, never more."""
    lib = oracle.lib()
    assert lib.oracle_get_rounded_transcendentals() == 0
    k = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k, width=48, height=32, iterations=2)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    a = oracle.render(flat, si, ppi, eye, direction, angles)
    try:
        lib.oracle_set_rounded_transcendentals(1)
        assert lib.oracle_get_rounded_transcendentals() == 1
        b = oracle.render(flat, si, ppi, eye, direction, angles)
    finally:
        lib.oracle_set_rounded_transcendentals(0)
    k.finalize()
    assert lib.oracle_get_rounded_transcendentals() == 0
    assert np.array_equal(a[1], b[1])
    ulp = np.abs(a[0][..., :3].view(np.int32).astype(np.int64) - b[0][..., :3].view(np.int32).astype(np.int64))
    assert ulp.max() <= 4 and (ulp > 0).mean() < 0.05
