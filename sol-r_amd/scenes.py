"""Synthetic scenes for the configurations named in BASELINE.json / SURVEY.md section 8(d).

Every scene is built through the same flat API a reference host would use
(SolR_AddMaterial / SolR_SetMaterial / SolR_AddPrimitive / SolR_SetPrimitive /
SolR_CompactBoxes), so the box tree and primitive order are the ones the
reference's builder produces for these calls.  All randomness comes from a
32-bit LCG with a fixed seed (the reference uses rand()).
"""
import math

from . import (ptSphere, ptCylinder, ptTriangle, ptXYPlane, ptYZPlane, ptXZPlane, glFull)


class LCG:
    """Numerical Recipes LCG; deterministic stand-in for the reference's rand()."""

    def __init__(self, seed):
        self.state = seed & 0xFFFFFFFF

    def next(self):
        self.state = (self.state * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.state >> 8

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next() % 100000) / 100000.0


def _wall_color(rng):
    # createRandomMaterials: 0.2 + rand() % 600 / 1000 (apps/scenes/Scene.cpp:386-388)
    return [0.2 + (rng.next() % 600) / 1000.0 for _ in range(3)]


def add_light(k, position=(8000.0, 8000.0, -8000.0), radius=10.0, intensity=2.0):
    """Emissive sphere with the reference's DEFAULT_LIGHT_MATERIAL parameters
    (apps/scenes/Scene.cpp:628-633): white, innerIllumination = (2, 10*viewDistance, viewDistance)."""
    m = k.add_material(1.0, 1.0, 1.0, innerIllumination=intensity)
    return k.add_primitive(ptSphere, position, size=(radius, 0, 0), material=m, movable=0)


def add_room(k, rng, center=(0.0, 15000.0, 0.0), half=(20000.0, 20000.0, 20000.0)):
    """Six axis-aligned rectangles, the primitives GPUKernel::addRectangle emits
    (solr/engines/GPUKernel.cpp:1711-1739), one material per wall.  Floor at y = -5000
    like the viewer's ground height (CornellBoxScene.cpp:40)."""
    x, y, z = center
    w, h, d = half
    mats = [k.add_material(*_wall_color(rng), specValue=0.1, specPower=200.0) for _ in range(6)]
    ids = [
        k.add_primitive(ptXYPlane, (x, y, z + d), size=(w, h, d), material=mats[0]),
        k.add_primitive(ptXYPlane, (x, y, z - d), size=(w, h, d), material=mats[1]),
        k.add_primitive(ptYZPlane, (x - w, y, z), size=(w, h, d), material=mats[2]),
        k.add_primitive(ptYZPlane, (x + w, y, z), size=(w, h, d), material=mats[3]),
        k.add_primitive(ptXZPlane, (x, y + h, z), size=(w, h, d), material=mats[4]),
        k.add_primitive(ptXZPlane, (x, y - h, z), size=(w, h, d), material=mats[5]),
    ]
    return ids


def cornell(k, width=512, height=512, iterations=1, extra_spheres=16, glass=2, seed=2017, room=True, **scene_info):
    """BASELINE configs[0]/[1]: Cornell box, about 30 primitives (spheres + planes).

    Four reflective spheres of radius 2000 at (+-2200, 0, 0), (0, +-2200, 0)
    (apps/scenes/experiments/CornellBoxScene.cpp:42-50), a ring of small spheres and two
    glass spheres, a six-plane room, one emissive sphere.  Camera as the viewer's default
    (apps/solrViewer.cpp:178-181): eye (0, 0, -15000), look-at origin, angles (0,0,0,6400)."""
    rng = LCG(seed)
    scene_info.setdefault("graphicsLevel", glFull)
    k.initialize(width=width, height=height, nbRayIterations=iterations, **scene_info)
    for cx, cy in ((2200.0, 0.0), (-2200.0, 0.0), (0.0, 2200.0), (0.0, -2200.0)):
        c = _wall_color(rng)
        m = k.add_material(c[0], c[1], c[2], reflection=0.5, specValue=1.0, specPower=234.0)
        k.add_primitive(ptSphere, (cx, cy, 0.0), size=(2000.0, 0, 0), material=m)
    for i in range(extra_spheres):
        a = 2.0 * math.pi * i / max(extra_spheres, 1)
        c = _wall_color(rng)
        m = k.add_material(c[0], c[1], c[2], reflection=0.25 if i % 2 else 0.0, specValue=0.5, specPower=100.0)
        k.add_primitive(ptSphere, (9000.0 * math.cos(a), -4400.0, 9000.0 * math.sin(a)), size=(600.0, 0, 0),
                        material=m)
    for i in range(glass):
        m = k.add_material(0.9, 0.95, 1.0, reflection=1.0, refraction=1.1, transparency=0.7, specValue=1.0,
                           specPower=200.0)
        k.add_primitive(ptSphere, (-5000.0 + 10000.0 * i, 3500.0, -6000.0), size=(1200.0, 0, 0), material=m)
    if room:
        add_room(k, rng)
    add_light(k)
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0))
    return k


def height_field(k, n=224, width=1920, height=1080, iterations=2, seed=1234, **scene_info):
    """BASELINE configs[2]: triangle mesh with per-vertex normals, 2*n*n triangles
    (n = 224 -> 100 352), scaled to +-10000 like apps/scenes ObjScene, one light at
    (8000, 8000, -8000) (ObjScene.cpp:98-106)."""
    rng = LCG(seed)
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    scale = 10000.0
    phase = [rng.uniform(0.0, 6.28) for _ in range(4)]

    def f(u, v):
        return 1500.0 * (math.sin(3.0 * u + phase[0]) * math.cos(2.0 * v + phase[1]) +
                         0.5 * math.sin(7.0 * u + 5.0 * v + phase[2]))

    def point(i, j):
        u = (i / n) * 2.0 - 1.0
        v = (j / n) * 2.0 - 1.0
        return (u * scale, f(u * 3.0, v * 3.0) - 3000.0, v * scale)

    def normal(i, j):
        e = 1.0
        a, b = point(i - e, j), point(i + e, j)
        c, d = point(i, j - e), point(i, j + e)
        tx = (b[0] - a[0], b[1] - a[1], b[2] - a[2])
        tz = (d[0] - c[0], d[1] - c[1], d[2] - c[2])
        nx = tz[1] * tx[2] - tz[2] * tx[1]
        ny = tz[2] * tx[0] - tz[0] * tx[2]
        nz = tz[0] * tx[1] - tz[1] * tx[0]
        return (nx, ny, nz)

    mats = [k.add_material(*_wall_color(rng), reflection=0.3 if m % 4 == 0 else 0.0, specValue=0.6, specPower=80.0)
            for m in range(8)]
    pts = [[point(i, j) for j in range(n + 1)] for i in range(n + 1)]
    nrm = [[normal(i, j) for j in range(n + 1)] for i in range(n + 1)]
    for i in range(n):
        for j in range(n):
            m = mats[((i // 28) + (j // 28)) % len(mats)]
            a, b, c, d = pts[i][j], pts[i + 1][j], pts[i + 1][j + 1], pts[i][j + 1]
            na, nb, nc, nd = nrm[i][j], nrm[i + 1][j], nrm[i + 1][j + 1], nrm[i][j + 1]
            t = k.add_primitive(ptTriangle, a, b, c, material=m)
            k.set_normals(t, na, nb, nc)
            t = k.add_primitive(ptTriangle, a, c, d, material=m)
            k.set_normals(t, na, nc, nd)
    add_light(k)
    k.compact_boxes(True)
    k.set_camera((0.0, 4000.0, -15000.0))
    return k


def molecule(k, atoms=50000, width=1920, height=1080, iterations=3, seed=4321, **scene_info):
    """BASELINE configs[3]: synthetic molecule - spheres of radius 30..60 (scaled) on a
    jittered helix plus half-bond cylinders of radius 10 as io/PDBReader.cpp:616-695 emits
    them, 119 flat element materials (PDBReader.cpp:57,392-399), light at
    (-5000, 5000, -15000) (apps/scenes/science/MoleculeScene.cpp:83-89)."""
    rng = LCG(seed)
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    mats = [k.add_material(*_wall_color(rng), specValue=0.8, specPower=100.0) for _ in range(119)]
    scale = 2.0
    turns = max(atoms // 400, 1)
    prev = None
    for a in range(atoms):
        t = a / atoms
        ang = 2.0 * math.pi * turns * t
        rad = 3000.0 + 2500.0 * math.sin(9.0 * math.pi * t)
        p = (rad * math.cos(ang) + rng.uniform(-300, 300), (t - 0.5) * 16000.0 + rng.uniform(-300, 300),
             rad * math.sin(ang) + rng.uniform(-300, 300))
        e = rng.next() % 119
        r = (30.0 + (rng.next() % 31)) * scale
        k.add_primitive(ptSphere, p, size=(r, 0, 0), material=mats[e])
        if prev is not None and a % 2 == 0:
            mid = tuple((p[i] + prev[0][i]) * 0.5 for i in range(3))
            k.add_primitive(ptCylinder, prev[0], mid, size=(10.0 * scale, 0, 0), material=mats[prev[1]])
            k.add_primitive(ptCylinder, mid, p, size=(10.0 * scale, 0, 0), material=mats[e])
        prev = (p, e)
    add_light(k, position=(-5000.0, 5000.0, -15000.0))
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0))
    return k


def irt_model(k, path, width=1920, height=1080, iterations=3, scale=5000.0, floor=True, **scene_info):
    """A scene around an .irt model file (reference: FileMarshaller::loadFromFile, the viewer's
    "load scene"): material slots for the file's ids first - the loader overwrites them - then the
    model scaled to `scale` units of height, a floor and the default light.  medias/irt/test.irt of the
    reference is the sample of the format."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    for _ in range(64):                      # the ids a file may carry; test.irt uses 0..15
        k.add_material(0.5, 0.5, 0.5)
    ground = k.add_material(0.55, 0.5, 0.45, reflection=0.2, specValue=0.5, specPower=100.0)
    n = k.load_from_file(path, scale)
    if floor:
        k.add_primitive(ptXZPlane, (0.0, -0.5 * scale - 50.0, 0.0), size=(20000.0, 0.0, 20000.0), material=ground,
                        movable=0)
    add_light(k, position=(-6000.0, 9000.0, -9000.0))
    k.compact_boxes(True)
    k.set_camera((0.0, 0.3 * scale, -3.2 * scale), look_at=(0.0, 0.0, 0.0))
    return k


def obj_model(k, path, width=1920, height=1080, iterations=3, scale=5000.0, **scene_info):
    """A scene around a Wavefront OBJ model (reference: OBJReader::loadModelFromFile; the viewer's
    Cornell-box scene loads medias/obj/cornell.obj this way): the model auto-scaled to `scale` and
    centred, its MTL materials from id 0, one light inside."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    for _ in range(64):                      # ids for the MTL file's materials, overwritten by the loader
        k.add_material(0.5, 0.5, 0.5)
    ground = k.load_obj_model(path, material_id=0, auto_scale=True, scale=scale, auto_center=True)
    add_light(k, position=(0.0, 0.35 * scale, -0.1 * scale), radius=0.02 * scale)
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -1.6 * scale), look_at=(0.0, 0.0, 0.0))
    return ground


def swc_morphology(k, path, width=1920, height=1080, iterations=3, scale=40.0, **scene_info):
    """A scene around an SWC neuron morphology (reference: SWCReader::loadMorphologyFromFile, its SwcScene):
    spheres and cylinders along the sample points, one material, a light."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    m = k.add_material(0.85, 0.75, 0.35, specValue=0.6, specPower=80.0)
    n = k.load_swc_morphology(path, scale=(scale, scale, scale, scale), material_id=m)
    add_light(k, position=(-6000.0, 9000.0, -12000.0))
    k.compact_boxes(True)
    k.set_camera((0.0, 2000.0, -16000.0), look_at=(0.0, 2000.0, 0.0))
    return n


def pdb_molecule(k, path, width=1920, height=1080, iterations=3, geometry_type=3, scale=200.0, **scene_info):
    """A scene around a PDB molecule (reference: PDBReader::loadAtomsFromFile as its MoleculeScene calls it:
    atom size 100, stick size 10, scale 200, materials by element; light at (-5000, 5000, -15000))."""
    k.initialize(width=width, height=height, nbRayIterations=iterations, graphicsLevel=glFull, **scene_info)
    for _ in range(1100):                    # ids the reader writes or refers to (0..118, 1000, 1010)
        k.add_material(0.5, 0.5, 0.5, specValue=1.0, specPower=100.0)
    k.load_molecule(path, geometry_type=geometry_type, atom_size=100.0, stick_size=10.0, material_type=0, scale=scale)
    add_light(k, position=(-5000.0, 5000.0, -15000.0), radius=1.0)
    k.compact_boxes(True)
    k.set_camera((0.0, 0.0, -15000.0), look_at=(0.0, 0.0, 0.0))
    return k
