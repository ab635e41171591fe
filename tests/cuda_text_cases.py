"""The cases of tests/test_cuda_text_model.py, and the process that runs them.

    python tests/cuda_text_cases.py            -> one JSON document on stdout

Every case renders a small frame (or a sequence of passes) twice - with oracle/solr_oracle.c in dialect 0, built with
a counter on every dialect switch (make -C oracle coverage), and with tests/cuda_text_model.py, the independent
reading of the CUDA text - and reports whether the float frame buffer, the primitive ids and the bitmap are the same
bits.  The document ends with the counters: how often each switch was evaluated in each dialect.  It runs in a
process of its own because the counting build must be the first oracle library the process loads."""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

solr = importlib.import_module("sol-r_amd")
import scenes_extra as X          # noqa: E402
import cuda_text_model as M       # noqa: E402

W, H = 24, 16


def two_lamps_and_a_sky(k, width=W, height=H, iterations=2, **info):
    """a few primitives under a plain-coloured skybox bright enough for the sky word of CRT:274-275 (r + g + b > 2.5),
    a transparent and reflective sphere between the lamp and the floor (tinted shadows, the deferred reflection ray),
    two lamps (the lamp loop of GI:956-961 applies lamp 0 twice)"""
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    sky = k.add_material(0.9, 0.85, 0.95)
    floor = k.add_material(0.6, 0.5, 0.4, specValue=0.3, specPower=20.0)
    glass = k.add_material(0.8, 0.9, 1.0, reflection=0.7, refraction=1.3, transparency=0.5, opacity=0.3, specValue=0.9,
                           specPower=120.0)
    matte = k.add_material(0.3, 0.7, 0.4, specValue=0.2, specPower=10.0, noise=0.01)
    k.add_primitive(solr.ptXZPlane, (0, -3000, 0), size=(9000, 0, 9000), material=floor)
    k.add_primitive(solr.ptSphere, (0, 0, 0), size=(1800, 0, 0), material=glass)
    k.add_primitive(solr.ptSphere, (2600, -1800, 1500), size=(1000, 0, 0), material=matte)
    k.add_primitive(solr.ptCone, (-3800, -3000, 500), (-3000, 1200, 900), size=(500, 0, 0), material=matte)
    # a dimly glowing sphere in view: Lambert's term of an emissive material is innerIllumination.x + n.l (GI:987)
    ember = k.add_material(0.9, 0.4, 0.2, innerIllumination=0.25, specValue=0.1, specPower=10.0)
    k.add_primitive(solr.ptSphere, (-2500, 1500, -500), size=(900, 0, 0), material=ember)
    X._light(k, pos=(1500.0, 9000.0, -2500.0))
    X._light(k, pos=(-6000.0, 7000.0, -7000.0), intensity=1.0)
    k.compact_boxes(True)
    k.set_camera((0.0, 1200.0, -12000.0), look_at=(0.0, -500.0, 0.0), angles=(0.0, 0.0, 0.0))
    k.set_scene_info(skyboxMaterialId=sky)
    return k


def grids(k, width=W, height=H, iterations=2, **info):
    """planes with the wireframe pattern of GI:458-460 (attributes.z == 2: only the lines of a 100-unit grid are
    there, TM:449-456), in front of the camera, under it and beside it; a mirror sphere to see them by reflection"""
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    wire = [k.add_material(c[0], c[1], c[2], wireframe=True, wireframeWidth=w_, specValue=0.3, specPower=30.0)
            for c, w_ in (((0.9, 0.3, 0.2), 40), ((0.2, 0.8, 0.3), 55), ((0.3, 0.4, 0.9), 70))]
    back = k.add_material(0.5, 0.5, 0.6, specValue=0.2, specPower=10.0)
    mirror = k.add_material(0.7, 0.7, 0.7, reflection=0.6, specValue=0.8, specPower=100.0)
    k.add_primitive(solr.ptXYPlane, (130, 70, 900), size=(760, 520, 0), material=wire[0])
    k.add_primitive(solr.ptXZPlane, (60, -430, 300), size=(830, 0, 700), material=wire[1])
    k.add_primitive(solr.ptYZPlane, (-640, 30, 300), size=(0, 540, 700), material=wire[2])
    k.add_primitive(solr.ptXYPlane, (0, 0, 1500), size=(3000, 3000, 0), material=back)
    k.add_primitive(solr.ptSphere, (200, -100, 200), size=(230, 0, 0), material=mirror)
    X._light(k, pos=(500.0, 900.0, -1500.0))
    k.compact_boxes(True)
    k.set_camera((30.0, 20.0, -1400.0), look_at=(0.0, 0.0, 0.0), angles=(0.0, 0.0, 0.0), w=1100.0)
    return k


def mix(k, **info):
    info.setdefault("width", W)
    info.setdefault("height", H)
    return X.primitives_mix(k, **info)


def cornell(k, **info):
    info.setdefault("iterations", 3)
    info.setdefault("width", W)
    info.setdefault("height", H)
    return solr.scenes.cornell(k, **info)


def sticks(k, **info):
    return X.sticks(k, width=W, height=H, **info)


def triangles(k, **info):
    return X.triangles_only(k, width=W, height=H, **info)


PP = lambda t, p1=0.0, p2=0.0, p3=0: dict(type=t, param1=p1, param2=p2, param3=p3)   # noqa: E731

# name -> (scene builder, SceneInfo changes, post-processing, passes)
CASES = {
    "cornell, 3 bounces (glass: refraction, opacity, deferred reflection)": (cornell, {}, None, [0]),
    "every primitive type, two lamps": (mix, {}, None, [0]),
    "double-sided triangles (GI:638-648)": (mix, dict(doubleSidedTriangles=1), None, [0]),
    "every primitive tested as a triangle (GI:743-747)": (mix, dict(extendedGeometry=0), None, [0]),
    "triangle mesh": (triangles, {}, None, [0]),
    "triangle mesh, double-sided": (triangles, dict(doubleSidedTriangles=1), None, [0]),
    "cylinders": (sticks, {}, None, [0]),
    "box-debug view (GI:693-696, CRT:404)": (mix, dict(renderBoxes=1), None, [0]),
    "no shading (GI:953, 1075-1078)": (mix, dict(graphicsLevel=M.glNoShading), None, [0]),
    "Phong only": (mix, dict(graphicsLevel=M.glPhong), None, [0]),
    "Phong and Blinn, no bounces": (mix, dict(graphicsLevel=M.glPhongAndBlinn), None, [0]),
    "bounces without shadows": (mix, dict(graphicsLevel=M.glReflectionsAndRefractions), None, [0]),
    "gradient background (CRT:278-286)": (mix, dict(gradientBackground=1, bgColor=(0.3, 0.5, 0.7, 0.2)), None, [0]),
    "fog (CRT:391-398)": (mix, dict(atmosphericEffect=M.aeFog, viewDistance=24000.0), None, [0]),
    "soft shadows (shadowIntensity 0.4)": (mix, dict(shadowIntensity=0.4), None, [0]),
    "orthographic camera (axis-parallel rays: GI:38-40)": (mix, dict(cameraType=M.ctOrthographic), None, [0]),
    "five-ray camera (CRT:504-514, 534-535)": (cornell, dict(cameraType=M.ctAntialiazed, iterations=2), None, [0]),
    "anaglyph camera (CRT:840-926), passes 0-12": (cornell, dict(cameraType=M.ctAnaglyph, eyeSeparation=350.0, iterations=1),
                                                   None, [0, 1, 10, 11, 12]),
    "fish-eye camera (CRT:741-813), passes 0-12": (cornell, dict(cameraType=M.ctPanoramic, iterations=1), PP(0, 0.0, 0.002),
                                                  [0, 1, 10, 11, 12]),
    "3D-vision camera (CRT:953-1043), passes 0-12": (cornell, dict(cameraType=M.ctVR, iterations=1), PP(0, 9000.0),
                                                    [0, 1, 10, 11, 12]),
    "BGR frame buffer (GS:144-154)": (cornell, dict(frameBufferType=M.ftBGR, height=W, iterations=1), None, [0]),
    "refinement and accumulation passes 0-13 (CRT:454-458, 470-479, 515-522, 550-562; lamp jitter GI:969-976)":
        (mix, dict(iterations=1), PP(0, 11000.0), list(range(0, 14))),
    "global illumination, full (CRT:157-174, 317-378), passes 0-13": (mix, dict(advancedIllumination=M.aiFull, iterations=1),
                                                                     None, [0, 5, 10, 11, 12, 13]),
    "global illumination, basic": (mix, dict(advancedIllumination=M.aiBasic, iterations=1), None, [0, 10, 11]),
    "random illumination (CRT:527-532)": (mix, dict(advancedIllumination=M.aiRandomIllumination, timestamp=7), None, [0, 11]),
    "sky of one colour, two lamps, tinted shadows": (two_lamps_and_a_sky, {}, None, [0]),
    "sky of one colour under global illumination": (two_lamps_and_a_sky, dict(advancedIllumination=M.aiFull, iterations=1),
                                                    None, [0, 10, 11, 12]),
    "depth of field (CRT:1081-1120: second index + 1000)": (cornell, dict(iterations=1), PP(M.ppe_depthOfField, 9000.0, 6000.0, 12),
                                                            [0, 10, 11, 12]),
    "ambient occlusion (CRT:1128-1181)": (cornell, dict(iterations=1), PP(M.ppe_ambientOcclusion, 0.0, 900.0), [0, 11, 12]),
    "radiosity (CRT:1189-1228)": (mix, dict(iterations=1), PP(M.ppe_radiosity, 0.0, 400.0, 10), [0, 11, 12]),
    "cartoon (CRT:1341-1358)": (cornell, dict(iterations=1), PP(M.ppe_cartoon, 9000.0), [0]),
}


def lamps_without_primitives(flat):
    """the light list as a host would fill it for lamps that are not primitives of the scene (LightInformation with
    primitiveId -1): GI:969-976 jitters their centres in accumulation passes all the same"""
    flat.lights["primitiveId"] = -1


def skewed_plane_normals(flat):
    """planeIntersection returns the primitive's n0 (GI:431), whatever it is: not the axis the plane is named after"""
    for i, p in enumerate(flat.primitives):
        if p["type"] in (M.ptXYPlane, M.ptYZPlane, M.ptXZPlane, M.ptCheckboard):
            n = np.array(p["n0"], np.float32) + np.array([0.25, -0.35, 0.15], np.float32)
            flat.primitives["n0"][i] = n / np.float32(np.sqrt((n * n).sum()))


def view_noise(flat):
    """Material.color.w, the view noise of CRT:257-267 (no entry of the flat API sets it)"""
    flat.materials["color"][:, 3] = np.where(np.arange(len(flat.materials)) % 2 == 0, 0.004, 0.0).astype(np.float32)


def large_randoms(flat):
    """four times the reference's random numbers: the global-illumination ray's direction normal + 100 r (CRT:164-166)
    can then point below the surface and is turned round (CRT:168-170)"""
    flat.randoms = (flat.randoms * np.float32(4)).astype(np.float32)


CASES["volume camera (CRT:592-713, GI:1088-1265), passes 0-12"] = (cornell, dict(cameraType=M.ctVolumeRendering, iterations=1),
                                                                    PP(0, 3000.0, 8.0), [0, 1, 2, 10, 11, 12])
CASES["volume camera, hits nearer than the threshold left out (GI:1170)"] = (cornell, dict(cameraType=M.ctVolumeRendering, iterations=1),
                                                                             PP(0, 15500.0, 6.0), [0])
CASES["volume camera, every primitive type, no threshold"] = (mix, dict(cameraType=M.ctVolumeRendering), PP(0, 0.0, 5.0), [0])
CASES["volume camera without shading (GI:1174)"] = (mix, dict(cameraType=M.ctVolumeRendering, graphicsLevel=M.glNoShading),
                                                    PP(0, 0.0, 5.0), [0])
CASES["volume camera with random illumination (CRT:677-682)"] = (cornell, dict(cameraType=M.ctVolumeRendering, iterations=1,
                                                                                advancedIllumination=M.aiRandomIllumination,
                                                                                timestamp=5), PP(0, 0.0, 12.0), [0, 11])
CASES["wireframe grids (GI:458-460, 470-472, 491-493, 525-527; TM:449-456)"] = (grids, {}, None, [0])
CASES["view noise, passes 0-12 (CRT:257-267)"] = (mix, dict(iterations=2), None, [0, 1, 2, 11, 12], view_noise)
CASES["global illumination with large random numbers (CRT:168-170)"] = (mix, dict(advancedIllumination=M.aiFull, iterations=1),
                                                                         None, [0, 10, 11], large_randoms)
CASES["3D-vision camera with random illumination (CRT:1020-1025)"] = (cornell, dict(cameraType=M.ctVR, iterations=1,
                                                                       advancedIllumination=M.aiRandomIllumination, timestamp=3),
                                                                      PP(0, 9000.0), [0, 1])
CASES["lamps that are no primitives, passes 0-12 (GI:961, 969-976)"] = (mix, dict(iterations=1), None, [0, 10, 11, 12], lamps_without_primitives)
CASES["planes whose n0 is not their axis (GI:431)"] = (mix, {}, None, [0], skewed_plane_normals)
for f, name in enumerate(("emboss", "find edges", "sharpen", "blur", "motion blur", "subtle sharpen", "none (param3 = 6)")):
    CASES["filter: %s (CRT:1236-1333)" % name] = (cornell, dict(iterations=1), PP(M.ppe_filter, 0.0, 0.0, f), [0, 12] if f == 3 else [0])


class Prepared:
    """a case's inputs, kept after its scene is gone: the flattened arrays and the frame parameters of every pass"""

    def __init__(self, name):
        build, info, pp, passes = CASES[name][:4]
        tweak = CASES[name][4] if len(CASES[name]) > 4 else None
        info = dict(info)
        k = solr.Kernel(engine="host-only")
        if pp:
            k.set_post_processing(**pp)
        info.setdefault("maxPathTracingIterations", 20)
        build(k, **info)
        try:
            self.flat = k.flat_scene()
            if build is two_lamps_and_a_sky:
                # a material without a texture gets the host's 40000 x 40000 "computed texture" mapping, with which the
                # reference's skyboxMapping (GI:133-147) reads gigabytes past its atlas; 0 x 0 is what
                # realignTexturesAndMaterials gives it on the way to the device
                self.flat.materials["textureMapping"][k.frame_parameters()[0].skyboxMaterialId] = 0
            if tweak:
                tweak(self.flat)
            self.frames = []
            for it in passes:
                k.set_scene_info(pathTracingIteration=it)
                self.frames.append(k.frame_parameters())
        finally:
            k.finalize()

    def oracle_passes(self, oracle):
        out, pp, ids = [], None, None
        for si, ppi, eye, direction, angles in self.frames:
            pp, ids, rgb, counts, status = oracle.render(self.flat, si, ppi, eye, direction, angles, pp=pp, ids=ids, nthreads=1)
            out.append((pp, ids, rgb, status))
        return out

    def model_passes(self):
        out, pp, ids = [], None, None
        for si, ppi, eye, direction, angles in self.frames:
            focus = 0.0
            if si.cameraType == M.ctVR and pp is not None:
                idx = si.size_x // 2 * si.size_y // 2                # CRT:972, the integer expression as written
                focus = float(pp[idx // si.size_x, idx % si.size_x, 3])
            pp, ids, rgb = M.render(si, ppi, self.flat, eye, direction, angles, pp=pp, ids=ids, focus_depth=focus)
            out.append((pp, ids, rgb))
        return out


def compare(oracle_out, model_out, frames):
    res = {"passes": [], "same": True}
    for (opp, oids, orgb, status), (mpp, mids, mrgb), frame in zip(oracle_out, model_out, frames):
        same = {"status": int(status),
                "frame_buffer": bool(np.array_equal(opp.view(np.int32)[..., :7], mpp.view(np.int32)[..., :7])),
                "ids": bool(np.array_equal(oids, mids)), "bitmap": bool(np.array_equal(orgb, mrgb)),
                "hit_pixels": int((mids[..., 0] >= 0).sum()), "lit_pixels": int((mpp[..., :3].sum(axis=-1) > 0).sum()),
                "pass": int(frame[0].pathTracingIteration)}
        if not same["frame_buffer"]:
            bad = np.argwhere(opp.view(np.int32)[..., :7] != mpp.view(np.int32)[..., :7])
            y, x = int(bad[0][0]), int(bad[0][1])
            same["first_difference"] = {"pixel": [x, y], "oracle": [float(v) for v in opp[y, x]], "model": [float(v) for v in mpp[y, x]],
                                        "differing_values": int(len(bad))}
        if not same["ids"]:
            bad = np.argwhere(oids != mids)
            y, x = int(bad[0][0]), int(bad[0][1])
            same["first_id_difference"] = {"pixel": [x, y], "oracle": oids[y, x].tolist(), "model": mids[y, x].tolist()}
        res["passes"].append(same)
        res["same"] = res["same"] and status == 0 and same["frame_buffer"] and same["ids"] and same["bitmap"]
    return res


def texture_maps_case(oracle):
    """Sites of TM:30-73 (normalMap zeroes z, specularMap sets x y z) and TM:260-264 (the bump map leaves the opacity
    alone): intersectionShader on a textured XY plane through the oracle's probe entry, expected values from the text
    of cubeMapping (TM:385-441) and the three maps, in binary32."""
    F = np.float32
    L = oracle.lib()
    si = solr.SceneInfo()
    si.size_x = si.size_y = 8
    si.extendedGeometry = 1
    si.viewDistance = 50000.0
    tw, th = 4, 4
    rng = np.random.RandomState(3)
    atlas = rng.randint(0, 256, size=4 * tw * th * 3 + 64).astype(np.uint8)
    mats = np.zeros(2, solr.MATERIAL_DTYPE)
    mats["color"][0] = (0.1, 0.2, 0.3, 0.0)
    mats["specular"][0] = (0.5, 20.0, 0.25, 0.0)
    mats["textureMapping"][0] = (tw, th, 0, 3)
    mats["textureIds"][0] = (0, 1, 2, 3)                       # diffuse, normal, bump, specular
    mats["textureOffset"][0] = (0, tw * th * 3, 2 * tw * th * 3, 3 * tw * th * 3)
    mats["advancedTextureIds"][0] = (-1, -1, -1, -1)
    prims = np.zeros(1, solr.PRIMITIVE_DTYPE)
    prims["type"], prims["p0"], prims["size"], prims["n0"], prims["materialId"] = M.ptXYPlane, (10.0, 20.0, 5.0), (2.0, 2.0, 0.0), (0, 0, 1), 0
    inter = np.array([[10.5, 21.25, 5.0]], np.float32)
    areas = np.zeros((1, 3), np.float32)
    at = np.array([[0.5, 0.25, 1.1, 0.75]], np.float32)
    color, bump, spec, adv = np.zeros((1, 4), np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    L.oracle_probe_intersection_shader(1, C.byref(si), p(prims), p(mats), p(atlas), p(inter), p(areas), p(at), p(color), p(bump), p(spec), p(adv))
    # TM:385-397: u = (int)(x - p0.x + size.x) = (int)2.5 = 2, v = (int)(y - p0.y + size.y) = (int)3.25 = 3; % 4
    u, v = int(F(10.5) - F(10.0) + F(2.0)) % tw, int(F(21.25) - F(20.0) + F(2.0)) % th
    index = ((v * tw + u) * 3) % (tw * th * 3)
    expect_color = [F(atlas[index + c]) / F(256) for c in range(3)]
    strength = F(10) * F(int(atlas[2 * tw * th * 3 + index]) + int(atlas[2 * tw * th * 3 + index + 1]) + int(atlas[2 * tw * th * 3 + index + 2])) / F(768)   # TM:54
    i = tw * th * 3 + index
    expect_bump = [F(0) - strength * (F(atlas[i]) / F(256) - F(0.5)), F(0) - strength * (F(atlas[i + 1]) / F(256) - F(0.5)), F(0)]   # TM:37-39
    i = 3 * tw * th * 3 + index
    expect_spec = [F(atlas[i]) / F(256), F(1000) * F(atlas[i + 1]) / F(256), F(atlas[i + 2]) / F(256)]       # TM:70-72
    return {"same": bool(all(F(a) == b for a, b in zip(color[0, :3], expect_color)) and
                         all(F(a) == b for a, b in zip(bump[0], expect_bump)) and
                         all(F(a) == b for a, b in zip(spec[0, :3], expect_spec)) and
                         list(at[0]) == [F(0.5), F(0.25), F(1.1), F(0.75)]),      # attributes untouched (TM:260-264)
            "color": [float(x) for x in color[0, :3]], "expected_color": [float(x) for x in expect_color],
            "bump": [float(x) for x in bump[0]], "expected_bump": [float(x) for x in expect_bump],
            "specular": [float(x) for x in spec[0, :3]], "expected_specular": [float(x) for x in expect_spec]}


def _textured(tw=4, th=4, maps=(0, -1, -1, -1), seed=5, color=(0.1, 0.2, 0.3, 0.5)):
    rng = np.random.RandomState(seed)
    atlas = rng.randint(1, 255, size=4 * tw * th * 3 + 64).astype(np.uint8)
    mats = np.zeros(2, solr.MATERIAL_DTYPE)
    mats["color"][0] = color
    mats["specular"][0] = (0.5, 20.0, 0.25, 0.0)
    mats["textureMapping"][0] = (tw, th, 0, 3)
    mats["textureIds"][0] = maps
    mats["textureOffset"][0] = (0, tw * th * 3, 2 * tw * th * 3, 3 * tw * th * 3)
    mats["advancedTextureIds"][0] = (-1, -1, -1, -1)
    return atlas, mats


def _si():
    si = solr.SceneInfo()
    si.size_x = si.size_y = 8
    si.extendedGeometry = 1
    si.viewDistance = 50000.0
    si.transparentColor = 2.0
    si.geometryEpsilon = 0.001
    return si


def _intersect(oracle, si, prim, mats, atlas, origin, direction, shadows=0):
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    o, d = np.array(origin, np.float32), np.array(direction, np.float32)
    inter, normal, areas = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    shadow = C.c_float(0)
    hit = oracle.lib().oracle_primitive_intersection(C.addressof(si), p(prim), p(mats), p(atlas) if atlas is not None else None,
                                                     p(o), p(d), shadows, p(inter), p(normal), p(areas), C.addressof(shadow))
    return hit, inter, normal, shadow.value


def function_cases(oracle):
    """single functions through the oracle's entry points, expected values worked out here from the CUDA text"""
    F = np.float32
    out = {}
    p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
    si = _si()

    # --- a plane with a normal map: planeIntersection hands ITS OWN normal to cubeMapping (GI:556), whose normalMap
    # subtracts the texel's red and green from x and y and zeroes z (TM:37-39); strength 3 without a bump map (TM:421);
    # the shadow intensity is the colour's w (GI:558), here the material's
    tw = th = 4
    atlas, mats = _textured(maps=(0, 1, -1, -1))
    prim = np.zeros(1, solr.PRIMITIVE_DTYPE)
    prim["type"], prim["p0"], prim["size"], prim["n0"], prim["materialId"] = M.ptXYPlane, (10.0, 20.0, 5.0), (2.0, 2.0, 0.0), (0, 0, 1), 0
    hit, inter, normal, shadow = _intersect(oracle, si, prim, mats, atlas, (10.5, 21.25, 0.0), (0.0, 0.0, 1.0))
    u, v = int(F(10.5) - F(10.0) + F(2.0)) % tw, int(F(21.25) - F(20.0) + F(2.0)) % th            # TM:385-397
    index = ((v * tw + u) * 3) % (tw * th * 3)
    i = tw * th * 3 + index
    # seen from z < plane with dir.z > 0: the second branch negates n0 first (GI:531) -> (0, 0, -1)
    expect = [F(-0.0) - F(3) * (F(atlas[i]) / F(256) - F(0.5)), F(-0.0) - F(3) * (F(atlas[i + 1]) / F(256) - F(0.5)), F(0)]
    out["normal map on a plane's own normal (GI:556, TM:37-39)"] = {
        "same": bool(hit == 1 and all(F(a) == b for a, b in zip(normal, expect)) and F(shadow) == F(0.5)),
        "normal": [float(x) for x in normal], "expected": [float(x) for x in expect], "shadow": shadow}

    # --- ptCamera from behind: the second branch exists for it as for ptXYPlane (GI:514-541); every ptCamera hit goes
    # through cubeMapping (GI:551), which with a 0 x 0 mapping returns the material's colour (TM:399): shadow = colour.w
    mats0 = np.zeros(2, solr.MATERIAL_DTYPE)
    mats0["color"][0] = (0.2, 0.3, 0.4, 0.75)
    mats0["textureIds"][0] = (-1, -1, -1, -1)
    cam = np.zeros(1, solr.PRIMITIVE_DTYPE)
    cam["type"], cam["p0"], cam["size"], cam["n0"], cam["materialId"] = M.ptCamera, (0.0, 0.0, 5.0), (2.0, 2.0, 0.0), (0, 0, 1), 0
    hit_b, inter_b, normal_b, shadow_b = _intersect(oracle, si, cam, mats0, None, (0.5, 0.5, 0.0), (0.0, 0.0, 1.0))
    hit_f, inter_f, normal_f, shadow_f = _intersect(oracle, si, cam, mats0, None, (0.5, 0.5, 9.0), (0.0, 0.0, -1.0))
    out["camera plane from both sides (GI:514-541, 551-559)"] = {
        "same": bool(hit_b == 1 and list(normal_b) == [F(-0.0), F(-0.0), F(-1)] and list(inter_b) == [F(0.5), F(0.5), F(5)] and
                     F(shadow_b) == F(0.75) and hit_f == 1 and list(normal_f) == [F(0), F(0), F(1)]),
        "back": [int(hit_b)] + [float(x) for x in normal_b], "front": [int(hit_f)] + [float(x) for x in normal_f]}

    # --- intersectionShader: ptCone shares the case label of ptCylinder and ptSphere (GS:46-58): a textured cone is
    # mapped like a textured cylinder at the same place; neither shows the material's colour
    atlas, mats = _textured(tw=8, th=8, maps=(0, -1, -1, -1), seed=9)
    res = {}
    for t in (M.ptCone, M.ptCylinder):
        pr = np.zeros(1, solr.PRIMITIVE_DTYPE)
        pr["type"], pr["p0"], pr["p1"], pr["size"], pr["materialId"] = t, (0.0, 0.0, 0.0), (0.0, 4.0, 0.0), (1.0, 1.0, 1.0), 0
        inter = np.array([[0.6, 1.3, -0.8]], np.float32)
        areas, at = np.zeros((1, 3), np.float32), np.array([[0.5, 0.25, 1.1, 0.75]], np.float32)
        color, bump, spec, adv = np.zeros((1, 4), np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32)
        oracle.lib().oracle_probe_intersection_shader(1, C.byref(si), p(pr), p(mats), p(atlas), p(inter), p(areas), p(at), p(color), p(bump), p(spec), p(adv))
        res[t] = [float(x) for x in color[0, :3]]
    texels = {tuple(float(F(atlas[j + c]) / F(256)) for c in range(3)) for j in range(0, 8 * 8 * 3, 3)}
    out["a textured cone is mapped like a cylinder (GS:46-58)"] = {
        "same": bool(res[M.ptCone] == res[M.ptCylinder] and tuple(res[M.ptCone]) in texels), "cone": res[M.ptCone], "cylinder": res[M.ptCylinder]}

    # --- triangleUVMapping with a bump map: bumpMap only sets the strength (TM:45-57), nothing in the mapper writes
    # attributes but the reflection and transparency maps (TM:260-276), which this material has not
    atlas, mats = _textured(tw=8, th=8, maps=(0, -1, 2, -1), seed=11)
    tri = np.zeros(1, solr.PRIMITIVE_DTYPE)
    tri["type"], tri["p0"], tri["p1"], tri["p2"], tri["materialId"] = M.ptTriangle, (0, 0, 0), (4, 0, 0), (0, 4, 0), 0
    tri["vt0"], tri["vt1"], tri["vt2"] = (0.1, 0.1), (0.9, 0.1), (0.1, 0.9)
    inter = np.array([[1.0, 1.0, 0.0]], np.float32)
    areas = np.array([[4.0, 2.0, 2.0]], np.float32)
    at = np.array([[0.5, 0.25, 1.1, 0.75]], np.float32)
    color, bump, spec, adv = np.zeros((1, 4), np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32)
    oracle.lib().oracle_probe_intersection_shader(1, C.byref(si), p(tri), p(mats), p(atlas), p(inter), p(areas), p(at), p(color), p(bump), p(spec), p(adv))
    texels = {tuple(float(F(atlas[j + c]) / F(256)) for c in range(3)) for j in range(0, 8 * 8 * 3, 3)}
    out["a triangle's bump map leaves the opacity alone (TM:260-276)"] = {
        "same": bool(tuple(float(x) for x in color[0, :3]) in texels and list(at[0]) == [F(0.5), F(0.25), F(1.1), F(0.75)]),
        "attributes": [float(x) for x in at[0]], "color": [float(x) for x in color[0, :3]]}
    return out


def main():
    from oracle import loader
    loader.use_coverage_build()
    L = loader.lib()
    L.oracle_set_dialect(0)
    L.oracle_set_rounded_transcendentals(1)
    only = sys.argv[1:]
    out = {"cases": {}}
    prepared, model_out, hits_of = {}, {}, {}
    for name in CASES:
        if only and not any(o in name for o in only):
            continue
        loader.site_hits(reset=True)
        prepared[name] = Prepared(name)
        model_out[name] = prepared[name].model_passes()
        out["cases"][name] = compare(prepared[name].oracle_passes(loader), model_out[name], prepared[name].frames)
        hits_of[name] = [h[0] for h in loader.site_hits(reset=True)]
        assert all(h[1] == 0 for h in loader.site_hits()), "a case ran in the OpenCL dialect"
    texture_name = "texture maps on a plane (TM:30-73, 385-441)"
    out["cases"][texture_name] = texture_maps_case(loader)
    hits_of[texture_name] = [h[0] for h in loader.site_hits(reset=True)]
    function_names = []
    for fname, res in function_cases(loader).items():
        out["cases"][fname] = res
        function_names.append(fname)
    function_hits = [h[0] for h in loader.site_hits(reset=True)]
    for fname in function_names:
        hits_of[fname] = function_hits           # (evaluated together; a flip re-runs them all)
    nb_sites = len(hits_of[texture_name])
    out["site_hits"] = [sum(h[s] for h in hits_of.values()) for s in range(nb_sites)]
    # every switch in turn reads the OTHER dialect: some case that evaluates it must then differ from the model
    out["flipped"] = []
    for site in range(nb_sites):
        L.oracle_flip_site(site)
        noticed, tried = None, []
        for name in sorted((n for n in hits_of if hits_of[n][site] > 0), key=lambda n: -hits_of[n][site]):
            tried.append(name)
            if name == texture_name:
                differs = not texture_maps_case(loader)["same"]
            elif name in function_names:
                differs = not function_cases(loader)[name]["same"]
            else:
                differs = not compare(prepared[name].oracle_passes(loader), model_out[name], prepared[name].frames)["same"]
            if differs:
                noticed = name
                break
        out["flipped"].append({"site": site, "noticed_by": noticed, "cases_that_evaluate_it": len(tried) if noticed else tried})
    L.oracle_flip_site(-1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
