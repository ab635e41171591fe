"""Host logic (scene store, box-tree builder, frame protocol) on the device-less engine.

Expected values follow the reference's builder (solr/engines/GPUKernel.cpp, cited per test);
no GPU is touched: engine "host-only" stores scenes and refuses to render."""
import numpy as np
import pytest


@pytest.fixture()
def kernel(solr):
    k = solr.Kernel(engine="host-only")
    k.initialize(width=64, height=48)
    yield k
    k.finalize()


def nested(boxes):
    """skip pointers describe nested intervals (what the ballot-only walk relies on)"""
    ends = []
    n = len(boxes)
    for i, b in enumerate(boxes):
        skip = int(b["indexForNextBox"][0])
        if skip < 1 or i + skip > n:
            return False
        while ends and ends[-1] <= i:
            ends.pop()
        if ends and i + skip > ends[-1]:
            return False
        ends.append(i + skip)
    return True


def test_set_primitive_derives_what_the_intersections_need(solr, kernel):
    k = kernel
    m = k.add_material(0.5, 0.5, 0.5)
    k.add_primitive(solr.ptSphere, (1, 2, 3), size=(7, 8, 9), material=m)            # GPUKernel.cpp:569-575
    k.add_primitive(solr.ptCylinder, (0, 0, 0), (0, 10, 0), size=(2, 5, 6), material=m)  # :583-612
    k.add_primitive(solr.ptXYPlane, (0, 0, 5), size=(1, 2, 3), material=m)           # :622-630
    k.add_primitive(solr.ptYZPlane, (5, 0, 0), size=(1, 2, 3), material=m)
    k.add_primitive(solr.ptXZPlane, (0, 5, 0), size=(1, 2, 3), material=m)
    k.add_primitive(solr.ptTriangle, (0, 0, 0), (2, 0, 0), (0, 3, 0), material=m)    # :650-668
    k.add_primitive(solr.ptEllipsoid, (0, 0, 0), size=(1, 2, 3), material=m)
    light = k.add_material(1, 1, 1, innerIllumination=2.0)
    k.add_primitive(solr.ptSphere, (100, 100, 100), size=(1, 0, 0), material=light)
    k.compact_boxes(True)
    prims = {int(p["index"]): p for p in k.flat_scene().primitives}
    assert tuple(prims[0]["size"]) == (7, 7, 7)                       # sphere: w replicated
    cyl = prims[1]
    assert tuple(cyl["n1"]) == (0, 1, 0) and tuple(cyl["p2"]) == (0, 5, 0) and tuple(cyl["size"]) == (2, 2, 2)
    assert tuple(prims[2]["n0"]) == (0, 0, 1) and tuple(prims[3]["n0"]) == (1, 0, 0) and tuple(prims[4]["n0"]) == (0, 1, 0)
    for i in (2, 3, 4):
        assert tuple(prims[i]["n1"]) == tuple(prims[i]["n0"]) == tuple(prims[i]["n2"])
    # triangle: normalize(p1-p0) x normalize(p2-p0) = (1,0,0) x (0,1,0) = (0,0,1)
    assert tuple(prims[5]["n0"]) == (0, 0, 1) and tuple(prims[5]["n2"]) == (0, 0, 1)
    assert tuple(prims[6]["size"]) == (1, 2, 3)


def test_add_rectangle_emits_six_planes_in_reference_order(solr, kernel):
    k = kernel
    m = k.add_material()
    last = k.L.SolRx_AddRectangle(1.0, 2.0, 3.0, 10.0, 20.0, 30.0, m)      # GPUKernel.cpp:1711-1739
    assert last == 5
    light = k.add_material(innerIllumination=1.0)
    k.add_primitive(solr.ptSphere, (0, 0, 0), size=(1, 0, 0), material=light)
    k.compact_boxes(True)
    prims = {int(p["index"]): p for p in k.flat_scene().primitives}
    expect = [(solr.ptXYPlane, (1, 2, 33)), (solr.ptXYPlane, (1, 2, -27)), (solr.ptYZPlane, (-9, 2, 3)),
              (solr.ptYZPlane, (11, 2, 3)), (solr.ptXZPlane, (1, 22, 3)), (solr.ptXZPlane, (1, -18, 3))]
    for i, (t, p0) in enumerate(expect):
        assert int(prims[i]["type"]) == t and tuple(prims[i]["p0"]) == p0 and tuple(prims[i]["size"]) == (10, 20, 30)


def test_material_record(solr, kernel):
    k = kernel
    idx = k.add_material(0.1, 0.2, 0.3, noise=0.4, reflection=0.5, refraction=1.33, procedural=True, wireframe=True,
                         wireframeWidth=3, transparency=0.6, opacity=0.7, specValue=0.8, specPower=90.0,
                         specCoef=0.25, innerIllumination=1.5, illuminationDiffusion=11.0,
                         illuminationPropagation=12.0, fastTransparency=True)
    light = k.add_material(innerIllumination=1.0)
    k.add_primitive(solr.ptSphere, (0, 0, 0), size=(1, 0, 0), material=light)
    k.compact_boxes(True)
    m = k.flat_scene().materials[idx]                                  # GPUKernel.cpp:1806-1847
    assert np.allclose(m["color"], (0.1, 0.2, 0.3, 0.0))              # color.w is zeroed, the noise goes to innerIllumination.w
    assert np.allclose(m["innerIllumination"], (1.5, 11.0, 12.0, 0.4))
    assert np.allclose(m["specular"], (0.8, 90.0, 0.0, 0.25))
    assert np.allclose([m["reflection"], m["refraction"], m["transparency"], m["opacity"]], (0.5, 1.33, 0.6, 0.7))
    assert tuple(m["attributes"]) == (1, 1, 2, 3)                      # wireframe with width != 0 -> 2
    assert tuple(m["textureIds"]) == (-1, -1, -1, -1)
    assert tuple(m["textureMapping"]) == (40000, 40000, -1, 3)         # "computed texture" defaults (:1893-1896)
    # and back out through SolR_GetMaterial (SolRStub.cpp:427-476)
    import ctypes as C
    d = [C.c_double() for _ in range(14)]
    n = [C.c_int() for _ in range(11)]
    r = C.byref
    status = k.L.SolR_GetMaterial(idx, r(d[0]), r(d[1]), r(d[2]), r(d[3]), r(d[4]), r(d[5]), r(n[0]), r(n[1]), r(n[2]),
                                  r(d[6]), r(d[7]), r(n[3]), r(n[4]), r(n[5]), r(n[6]), r(n[7]), r(n[8]), r(n[9]),
                                  r(d[8]), r(d[9]), r(d[10]), r(d[11]), r(d[12]), r(d[13]), r(n[10]))
    assert status == 0
    assert np.allclose([x.value for x in d], (0.1, 0.2, 0.3, 0.4, 0.5, 1.33, 0.6, 0.7, 0.8, 90.0, 0.25, 1.5, 11.0, 12.0))
    assert [x.value for x in n] == [1, 0, 3, -1, -1, -1, -1, -1, -1, -1, 1]   # wireframe reads 0: the stored code is 2
    assert k.L.SolR_GetMaterial(5000, *([None] * 25)) == -1


def test_lights_box_comes_first_and_spans_the_view_distance(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    flat = k.flat_scene()
    b0 = flat.boxes[0]                                                 # GPUKernel.cpp:1179-1190
    assert tuple(b0["min"]) == (-50000,) * 3 and tuple(b0["max"]) == (50000,) * 3
    assert int(b0["nbPrimitives"]) == 1 and int(b0["startIndex"]) == 0
    assert len(flat.lights) == 1 and flat.nb_lamps == 1
    li = flat.lights[0]
    lamp = flat.primitives[0]
    assert int(li["primitiveId"]) == int(lamp["index"]) == 28 and tuple(li["location"]) == tuple(lamp["p0"])
    assert li["color"][3] == 2.0                                       # intensity = innerIllumination.x (:1225)
    assert k.L.SolR_GetLight(0) == 28 and k.L.SolR_GetLight(1) == -1


def test_flattened_tree_invariants(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    flat = k.flat_scene()
    boxes, prims = flat.boxes, flat.primitives
    assert nested(boxes)
    # every primitive is streamed exactly once ("Lost primitives on the way" check, GPUKernel.cpp:1268)
    assert sorted(int(i) for i in prims["index"]) == list(range(29))
    covered = np.zeros(len(prims), int)
    for b in boxes:
        n, s = int(b["nbPrimitives"]), int(b["startIndex"])
        if n:
            covered[s:s + n] += 1
            assert int(b["indexForNextBox"][0]) >= 1
            if not (tuple(b["min"]) == (-50000,) * 3):
                for p in prims[s:s + n]:                               # leaf bounds hold the primitive (:741-839)
                    r = p["size"][0] if int(p["type"]) == solr.ptSphere else p["size"]
                    assert np.all(b["min"] <= p["p0"] - r) and np.all(b["max"] >= p["p0"] + r)
        else:
            assert int(b["startIndex"]) >= 1                           # inner nodes store their depth (:1102)
    assert np.all(covered == 1)
    # tree depth: 29 primitives, divide by 4 while > 2 (GPUKernel.cpp:1056-1075): 29 -> 7 -> 1: depth 2
    assert k.L.SolRx_GetTreeDepth() == 2


def test_tree_depth_formula(solr):
    for n, depth in ((1, 1), (3, 1), (12, 2), (47, 2), (48, 3), (200, 4)):
        k = solr.Kernel(engine="host-only")
        k.initialize(width=16, height=16)
        m = k.add_material()
        for i in range(n - 1):
            k.add_primitive(solr.ptSphere, (i * 100.0, 0, 0), size=(10, 0, 0), material=m)
        k.add_primitive(solr.ptSphere, (0, 500, 0), size=(10, 0, 0), material=k.add_material(innerIllumination=1.0))
        k.compact_boxes(True)
        expect, c = 0, n
        while True:                                                    # do { ++depth; n /= 4 } while (n > 2)
            expect += 1
            c //= 4
            if c <= 2:
                break
        assert k.L.SolRx_GetTreeDepth() == expect == depth, n
        assert nested(k.flat_scene().boxes)
        k.finalize()


def same_records(a, b):
    """field-wise equality (padding bytes of the C records are unspecified)"""
    return a.shape == b.shape and all(np.array_equal(a[n], b[n]) for n in a.dtype.names)


def test_builder_is_deterministic_and_rebuildable(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    a = k.flat_scene()
    k.compact_boxes(True)                                              # rebuilding must not duplicate primitives
    b = k.flat_scene()
    assert same_records(a.boxes, b.boxes) and same_records(a.primitives, b.primitives)
    k2 = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k2, width=64, height=48, iterations=1)
    c = k2.flat_scene()
    assert same_records(a.boxes, c.boxes) and same_records(a.primitives, c.primitives)
    assert same_records(a.materials, c.materials) and same_records(a.lights, c.lights)


def test_rotate_primitives_keeps_lights_and_refits_boxes(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    before = {int(p["index"]): p.copy() for p in k.flat_scene().primitives}
    k.L.SolR_RotatePrimitives(0, 0, 0.0, 0.0, 0.0, 0.0, 0.3, 0.0)       # about Y; compactBoxes(false) inside
    after = {int(p["index"]): p for p in k.flat_scene().primitives}
    assert tuple(after[28]["p0"]) == tuple(before[28]["p0"])           # the lamp lives in the top-level box: not rotated
    assert not np.allclose(after[0]["p0"], before[0]["p0"])            # sphere at (2200,0,0) moved
    assert np.isclose(np.linalg.norm(after[0]["p0"]), 2200.0, rtol=1e-6)
    assert nested(k.flat_scene().boxes)


def test_deterministic_randoms(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    with pytest.raises(solr.SolrError):
        k.render()                                                     # fills the random buffer before failing
    r = k.flat_scene().randoms
    assert len(r) == solr.MAX_BITMAP_SIZE
    # 5e-6 * (k % 2000 - 1000): the reference's distribution (GPUKernel.cpp:2726)
    assert r.min() >= -0.005 and r.max() < 0.005 and len(np.unique(r)) > 1000
    steps = np.round(r / 0.000005)
    assert np.allclose(steps * 0.000005, r, atol=1e-9)


def test_host_only_engine_refuses_to_render(solr, kernel):
    k = kernel
    solr.scenes.cornell(k, width=64, height=48, iterations=1)
    with pytest.raises(solr.SolrError, match="host-only"):
        k.render()


def test_primitive_accessors(solr, kernel):
    import ctypes as C
    k = kernel
    m = k.add_material()
    i = k.add_primitive(solr.ptSphere, (1, 2, 3), size=(4, 0, 0), material=m)
    vals = [C.c_double() for _ in range(12)]
    mat = C.c_int()
    assert k.L.SolR_GetPrimitive(i, *[C.byref(v) for v in vals], C.byref(mat)) == 0
    assert [v.value for v in vals[:3]] == [1, 2, 3] and vals[9].value == 4 and mat.value == m
    assert k.L.SolR_GetPrimitive(99, *[C.byref(v) for v in vals], C.byref(mat)) == -1
    k.L.SolR_SetPrimitiveMaterial(i, 7)
    assert k.L.SolR_GetPrimitiveMaterial(i) == 7
    x, y, z = C.c_double(), C.c_double(), C.c_double()
    k.L.SolR_GetPrimitiveCenter(i, C.byref(x), C.byref(y), C.byref(z))
    assert (x.value, y.value, z.value) == (1, 2, 3)


def test_textures_are_packed_into_one_atlas(solr, kernel):
    k = kernel
    t0 = np.arange(4 * 2 * 3, dtype=np.uint8).reshape(2, 4, 3)
    t1 = (200 + np.arange(2 * 2 * 3, dtype=np.uint8)).reshape(2, 2, 3)
    k.set_texture(0, t0)
    k.set_texture(1, t1)
    m = k.add_material(diffuseTextureId=1)
    k.add_primitive(solr.ptSphere, (0, 0, 0), size=(1, 0, 0), material=m)
    k.add_primitive(solr.ptSphere, (0, 9, 0), size=(1, 0, 0), material=k.add_material(innerIllumination=1.0))
    k.compact_boxes(True)
    flat = k.flat_scene()
    assert bytes(flat.textures[:24]) == t0.tobytes() and bytes(flat.textures[24:36]) == t1.tobytes()
    mat = flat.materials[m]                                            # GPUKernel.cpp:1865-1889 / 2280-2300
    assert tuple(mat["textureMapping"]) == (2, 2, -1, 3) and int(mat["textureOffset"][0]) == 24
    # SolR_GetTexture hands the pixels back with the first and third channel swapped (SolRStub.cpp:351-373)
    back = np.zeros_like(t1)
    assert k.L.SolR_GetTexture(1, back.ctypes.data) == 0 and np.array_equal(back, t1[..., ::-1])
    assert k.L.SolR_GetTexture(7, back.ctypes.data) == 1
    assert k.L.SolR_RotatePrimitive(0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0) == 0 and k.L.SolR_RecompileKernels(None) == 0


def test_key_frames_are_blended_primitive_by_primitive(solr, kernel):
    # GPUKernel::morphPrimitives (reference GPUKernel.cpp:1513-1572): frames 1 .. n-2 = first + frame / n * (last - first)
    k = kernel
    light = k.add_material(innerIllumination=1.0)
    m = k.add_material(0.5, 0.5, 0.5)
    k.L.SolRx_SetNbFrames(4)
    for frame, x in ((0, 0.0), (3, 400.0)):
        k.L.SolRx_SetFrame(frame)
        for i in range(40):
            k.add_primitive(solr.ptSphere, (x + 100.0 * i, 10.0 * i, 0.0), size=(20.0 + frame, 0, 0), material=m)
        k.add_primitive(solr.ptSphere, (0.0, 5000.0, 0.0), size=(10.0, 0, 0), material=light, movable=0)
        k.compact_boxes(True)
    k.L.SolRx_MorphPrimitives()
    for frame in (1, 2):
        k.L.SolRx_SetFrame(frame)
        k.compact_boxes(False)              # the flattened arrays are shared: stream this frame's
        prims = k.flat_scene().primitives
        spheres = prims[prims["index"] < 40]
        spheres = spheres[np.argsort(spheres["index"])]
        r = np.float32(frame) / np.float32(4)
        want_x = (np.float32(100.0) * np.arange(40, dtype=np.float32)) + r * np.float32(400.0)
        assert np.allclose(spheres["p0"][:, 0], want_x, rtol=0, atol=1e-3)
        assert np.allclose(spheres["size"][:, 0], 20.0 + r * 3.0)
    k.L.SolRx_SetFrame(0)


def _flat_digest(flat):
    import hashlib
    h = hashlib.sha256()
    for a in (flat.boxes, flat.primitives, flat.lights):
        for name in a.dtype.names:   # field by field: the records have padding bytes
            h.update(np.ascontiguousarray(a[name]).tobytes())
    return h.hexdigest()[:16]


@pytest.mark.parametrize("name,kw,boxes,prims,digest", [
    ("cornell", dict(width=64, height=64, iterations=2), 75, 29, "bb2bfdce9fac275a"),
    ("molecule", dict(width=64, height=64, atoms=20000), 184169, 39999, "a83ea5ee46fd3f99"),
    ("height_field", dict(width=64, height=64, n=96), 55117, 18433, "8b8f8939456a1dd0"),
])
def test_flattened_tree_is_what_the_map_based_builder_produced(solr, name, kw, boxes, prims, digest):
    """The builder's per-level std::map (as in the reference) was replaced by OrderedMap (hash index +
    one sort per level).  These digests were taken from the std::map build: node order, skip pointers,
    bounds, primitive order and light records are unchanged, bit for bit."""
    k = solr.Kernel(engine="host-only")
    getattr(solr.scenes, name)(k, **kw)
    flat = k.flat_scene()
    assert (len(flat.boxes), len(flat.primitives)) == (boxes, prims)
    assert _flat_digest(flat) == digest
