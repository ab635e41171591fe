"""Feature-coverage scenes for the parity tests (built through the flat SolR_* API)."""
import importlib
import math

import numpy as np

solr = importlib.import_module("sol-r_amd")
S = solr.scenes


def _light(k, pos=(6000.0, 7000.0, -9000.0), intensity=2.0):
    return S.add_light(k, position=pos, intensity=intensity)


def primitives_mix(k, width=96, height=64, iterations=3, lamps=2, **info):
    """Every primitive type the walks dispatch on, plus the material switches they look at.  lamps=1: without the second
    lamp (the lamp loop of primitiveShader differs between the reference's two engines only when there are several)."""
    rng = S.LCG(7)
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    plain = k.add_material(0.7, 0.3, 0.2, specValue=0.5, specPower=50.0)
    mirror = k.add_material(0.2, 0.6, 0.8, reflection=0.6, specValue=1.0, specPower=200.0)
    glass = k.add_material(0.9, 0.95, 1.0, reflection=0.8, refraction=1.2, transparency=0.6, opacity=0.2,
                           specValue=1.0, specPower=150.0)
    proc = k.add_material(0.3, 0.8, 0.3, procedural=True, specValue=0.4, specPower=30.0)
    wire1 = k.add_material(1.0, 1.0, 0.2, wireframe=True, wireframeWidth=0)
    wire2 = k.add_material(0.2, 1.0, 1.0, wireframe=True, wireframeWidth=30)
    fast = k.add_material(0.8, 0.2, 0.8, transparency=0.5, refraction=1.0, fastTransparency=True)
    noisy = k.add_material(0.6, 0.6, 0.6, noise=0.02)
    glow = k.add_material(1.0, 0.9, 0.6, innerIllumination=0.4 if lamps > 1 else 0.0)   # (an emissive wall is a lamp too)
    k.add_primitive(solr.ptSphere, (-3000, 500, 0), size=(1500, 0, 0), material=mirror)
    k.add_primitive(solr.ptSphere, (800, -500, -2500), size=(1100, 0, 0), material=glass)
    k.add_primitive(solr.ptSphere, (3200, 900, 500), size=(1300, 0, 0), material=proc)
    k.add_primitive(solr.ptSphere, (-800, 2600, 1500), size=(700, 0, 0), material=fast)
    k.add_primitive(solr.ptSphere, (-1200, 2500, 3500), size=(700, 0, 0), material=fast)
    k.add_primitive(solr.ptEllipsoid, (0, -2500, 1000), size=(2200, 700, 1200), material=plain)
    k.add_primitive(solr.ptCylinder, (-4500, -3000, -1000), (-2500, 2500, 500), size=(350, 0, 0), material=noisy)
    k.add_primitive(solr.ptCone, (4500, -3500, 0), (3800, 1500, -800), size=(400, 0, 0), material=mirror)
    for i in range(6):   # a small fan of triangles with interpolated normals
        a0, a1 = 0.6 * i, 0.6 * (i + 1)
        p0 = (0.0, 3500.0, 2500.0)
        p1 = (2500.0 * math.cos(a0), 3500.0 + 900.0 * math.sin(3 * a0), 2500.0 + 2500.0 * math.sin(a0))
        p2 = (2500.0 * math.cos(a1), 3500.0 + 900.0 * math.sin(3 * a1), 2500.0 + 2500.0 * math.sin(a1))
        t = k.add_primitive(solr.ptTriangle, p0, p1, p2, material=plain if i % 2 else mirror)
        k.set_normals(t, (0.1, -1, 0.2), (0.3 * math.cos(a0), -1, 0.3 * math.sin(a0)),
                      (0.3 * math.cos(a1), -1, 0.3 * math.sin(a1)))
    k.add_primitive(solr.ptXYPlane, (0, 0, 9000), size=(9000, 6000, 0), material=plain)
    k.add_primitive(solr.ptYZPlane, (-9000, 0, 3000), size=(0, 6000, 6000), material=wire2)
    k.add_primitive(solr.ptYZPlane, (9000, 0, 3000), size=(0, 6000, 6000), material=glow)
    k.add_primitive(solr.ptXZPlane, (0, 6000, 3000), size=(9000, 0, 6000), material=wire1)
    k.add_primitive(solr.ptCheckboard, (0, -4500, 2000), size=(9000, 0, 7000), material=noisy)
    _light(k, intensity=2.0 if lamps > 1 else 0.8)
    if lamps > 1:
        _light(k, pos=(-7000.0, 5000.0, -6000.0), intensity=1.0)
    k.compact_boxes(True)
    k.set_camera((200.0, 300.0, -14000.0), look_at=(0.0, 0.0, 0.0), angles=(0.05, -0.1, 0.02))
    return k


def _texture(seed, w, h):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    base[..., 0] = (base[..., 0] // 2 + (xx * 255 // max(w - 1, 1)) // 2).astype(np.uint8)
    base[..., 1] = (base[..., 1] // 2 + (yy * 255 // max(h - 1, 1)) // 2).astype(np.uint8)
    return base


def textured(k, width=96, height=64, iterations=2, skybox=True, **info):
    """Texture tier: diffuse / normal / bump / specular / reflection / transparency / ambient-occlusion
    maps on a sphere, an axis plane and triangles; Mandelbrot and Julia materials; textured skybox."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    for i, (w, h) in enumerate([(64, 32), (32, 32), (32, 16), (16, 16), (24, 24), (20, 12), (18, 18), (128, 64)]):
        k.set_texture(i, _texture(100 + i, w, h), texture_type=i % 7)
    sky = k.add_material(0.1, 0.2, 0.4, diffuseTextureId=7)
    m_sphere = k.add_material(1, 1, 1, diffuseTextureId=0, normalTextureId=1, bumpTextureId=2, specularTextureId=3,
                              specValue=0.5, specPower=40.0, reflection=0.3)
    m_plane = k.add_material(1, 1, 1, diffuseTextureId=1, ambientOcclusionTextureId=4, reflectionTextureId=5,
                             reflection=0.5)
    m_tri = k.add_material(1, 1, 1, diffuseTextureId=4, transparencyTextureId=6, transparency=0.4, refraction=1.1,
                           procedural=True)
    m_mandel = k.add_material(0.9, 0.5, 0.2, diffuseTextureId=solr.TEXTURE_MANDELBROT)
    m_julia = k.add_material(0.3, 0.6, 0.9, diffuseTextureId=solr.TEXTURE_JULIA)
    s = k.add_primitive(solr.ptSphere, (-2500, 0, 0), size=(2000, 0, 0), material=m_sphere)
    k.set_texture_coordinates(s, (0, 0), (1.0, 1.0), (0, 0))     # sphereUVMapping scales by vt1 (TM:302-303)
    k.add_primitive(solr.ptXZPlane, (0, -2500, 0), size=(8000, 0, 8000), material=m_plane)
    k.add_primitive(solr.ptXYPlane, (0, 0, 6000), size=(8000, 5000, 0), material=m_mandel)
    k.add_primitive(solr.ptYZPlane, (7000, 0, 0), size=(0, 5000, 6000), material=m_julia)
    for i in range(4):
        p0 = (1000.0 + 900 * i, -1500.0, -1500.0 + 500 * i)
        p1 = (2800.0 + 900 * i, -1200.0, -1000.0 + 500 * i)
        p2 = (1500.0 + 900 * i, 1800.0, -500.0 + 500 * i)
        t = k.add_primitive(solr.ptTriangle, p0, p1, p2, material=m_tri)
        k.set_texture_coordinates(t, (0.1, 0.1), (0.9, 0.2), (0.4, 0.95))
    _light(k)
    k.compact_boxes(True)
    if skybox:
        k.set_scene_info(skyboxMaterialId=sky, skyboxSize=40000)
    k.set_camera((0.0, 500.0, -13000.0), angles=(0.03, 0.04, 0.0))
    return k


def triangles_only(k, width=80, height=60, iterations=2, n=6, dim=1.0, backdrop=False, **info):
    """Small height field: the all-triangle code path (extendedGeometry may be switched off).  n=7: 98 triangles and the
    lamp - BASELINE configs[2] in small; dim < 1: darker materials, no pixel's colour leaves [0, 1]."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    m = [k.add_material(dim * (0.3 + 0.1 * i), dim * 0.5, dim * (0.8 - 0.1 * i), reflection=0.3 * (i % 2)) for i in range(4)]
    def pt(i, j):
        return ((i / n - 0.5) * 9000.0, 900.0 * math.sin(1.7 * i) * math.cos(1.3 * j) - 1500.0, (j / n - 0.5) * 9000.0)
    for i in range(n):
        for j in range(n):
            a, b, c, d = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            t = k.add_primitive(solr.ptTriangle, a, b, c, material=m[(i + j) % 4])
            k.set_normals(t, (0, 1, 0), (0.2, 1, 0), (0.2, 1, 0.2))
            k.add_primitive(solr.ptTriangle, a, c, d, material=m[(i + j) % 4])
    if backdrop:   # (two triangles: no camera ray misses, and the frame stays with the lean kernel of triangles)
        wall = k.add_material(0.3 * dim, 0.35 * dim, 0.4 * dim)
        k.add_primitive(solr.ptTriangle, (-40000, -30000, 9000), (40000, -30000, 9000), (40000, 40000, 9000), material=wall)
        k.add_primitive(solr.ptTriangle, (-40000, -30000, 9000), (40000, 40000, 9000), (-40000, 40000, 9000), material=wall)
    lm = k.add_material(1, 1, 1, innerIllumination=2.0)
    k.add_primitive(solr.ptTriangle, (5000, 6000, -5000), (5300, 6000, -5000), (5000, 6300, -5000), material=lm)
    k.compact_boxes(True)
    k.set_camera((0.0, 2500.0, -12000.0), angles=(0.2, 0.0, 0.0))
    return k


def layered_terrain(k, n=36, layers=3, gap=40.0, width=160, height=120, iterations=3, reflection=0.6, **info):
    """A mirror terrain in `layers` sheets `gap` apart, seen at a grazing angle: every hit has a second hit a few per cent
    farther along the ray - what the reference's cut-off (slab parameter of a ray shorter than 1 against the closest
    DISTANCE, rt_device.h closestHitWalk) makes depend on the order of the leaves - and 2 n n layers triangles make
    the list long enough for the kernels that walk bounce rays order-free, checked."""
    rng = S.LCG(77)
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    mats = [k.add_material(*S._wall_color(rng), reflection=reflection, specValue=0.5, specPower=60.0) for _ in range(5)]
    def pt(i, j, layer):
        u, v = i / n - 0.5, j / n - 0.5
        return (u * 18000.0, 700.0 * math.sin(9.0 * u + 0.4) * math.cos(7.0 * v) - 2500.0 - gap * layer, v * 18000.0)
    for layer in range(layers):
        for i in range(n):
            for j in range(n):
                a, b, c, d = pt(i, j, layer), pt(i + 1, j, layer), pt(i + 1, j + 1, layer), pt(i, j + 1, layer)
                m = mats[(i // 3 + j // 3 + layer) % 5]
                t = k.add_primitive(solr.ptTriangle, a, b, c, material=m)
                k.set_normals(t, (0, 1, 0), (0.1, 1, 0), (0.1, 1, 0.1))
                t = k.add_primitive(solr.ptTriangle, a, c, d, material=m)
                k.set_normals(t, (0, 1, 0), (0.1, 1, 0.1), (0, 1, 0.1))
    _light(k, pos=(6000.0, 7000.0, -9000.0))
    k.compact_boxes(True)
    k.set_camera((0.0, -1200.0, -12000.0), look_at=(0.0, -2400.0, 0.0))
    return k


def sticks(k, width=80, height=60, iterations=2, backdrop=False, dim=1.0, **info):
    """Molecule-like: spheres joined by cylinders.  backdrop: a wall behind them, so that no camera ray misses; dim < 1:
    darker materials with a smaller highlight, so that no pixel's colour leaves [0, 1] (BASELINE configs[3] in small)."""
    rng = S.LCG(99)
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    mats = [k.add_material(*[dim * c for c in S._wall_color(rng)], specValue=0.8 * dim, specPower=100.0) for _ in range(6)]
    prev = None
    for a in range(40):
        p = (3500.0 * math.cos(0.5 * a) + rng.uniform(-200, 200), -3000.0 + 150.0 * a, 3500.0 * math.sin(0.5 * a))
        k.add_primitive(solr.ptSphere, p, size=(250.0 + 10 * (a % 7), 0, 0), material=mats[a % 6])
        if prev:
            k.add_primitive(solr.ptCylinder, prev, p, size=(80.0, 0, 0), material=mats[(a + 1) % 6])
        prev = p
    if backdrop:   # (a sphere: the frame stays with the lean kernel of spheres + cylinders)
        k.add_primitive(solr.ptSphere, (0, 0, 40000), size=(30000, 0, 0),
                        material=k.add_material(0.3 * dim, 0.35 * dim, 0.4 * dim, specValue=0.1, specPower=50.0))
    _light(k, pos=(-5000.0, 5000.0, -15000.0))
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -14000.0))
    return k


def lone_light(k, width=16, height=16, iterations=1, **info):
    """Nothing but the lamp: every ray misses or hits the emissive sphere."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, **info)
    k.add_primitive(solr.ptSphere, (0, 0, 0), size=(2500, 0, 0), material=k.add_material(1, 1, 1, innerIllumination=1.0))
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0))
    return k
