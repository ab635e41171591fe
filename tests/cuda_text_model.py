"""An independent model of the CUDA engine's per-pixel path, written from the reference's CUDA text alone.

TEST INFRASTRUCTURE.  oracle/solr_oracle.c restates the same text in C and is pinned, bit for bit, to outputs of the
reference's OpenCL engine wherever the two engines say the same thing.  Where they do not - the statements the
oracle keeps under its `g_cl` switches - its dialect 0 (the CUDA form, what the product is held to) used to be
pinned by reading only: the CUDA engine cannot be built here.  This file is the second
reading.  Every function below follows the cited lines of
    CRT  /root/reference/solr/engines/cuda/CudaRayTracer.cu
    GI   .../GeometryIntersections.cuh      GS  .../GeometryShaders.cuh
    VU   .../VectorUtils.cuh                TM  .../TextureMapping.cuh       HM  .../helper_math.h
one statement at a time, in plain Python over numpy binary32 scalars (every operation rounds to binary32 as the host
build of that text would: products and sums in source order, no contraction, `normalize` = v * (1.f / sqrtf(dot)),
HM:62-65, 1309-1313).  It was written without looking at the oracle's code and shares nothing with it but the record
layouts of include/solr_types.h; tests/test_cuda_text_model.py holds the oracle in dialect 0 to it bit for bit, and
counts (a build of the oracle with a counter on every switch) that every CUDA arm was executed on the way.

Scope: what those switches touch - untextured spheres, cylinders, cones, ellipsoids, triangles, axis planes and the
checkerboard (wireframe and chessboard-light masks included), both walks, the shader, the bounce loop with its
deferred reflection and global-illumination rays, the standard / orthographic / five-ray / anaglyph / fish-eye /
3D-vision / volume cameras, k_default and the five post-processing kernels.  Texture maps are left to the function-level
cases of tests/test_cuda_text_model.py.  Transcendentals (pow of the Blinn term, cos / sin of the procedural sphere
and the fish-eye camera) are taken in binary64 and rounded once, which is the oracle's
oracle_set_rounded_transcendentals(1) form; the camera's cosf / sinf come from the same libm the oracle calls."""
import ctypes
import math

import numpy as np

F = np.float32
NB_MAX_ITERATIONS = 10                      # Consts.h:29
NB_MAX_MATERIALS = 65506 + 30               # Consts.h:35
MAX_BITMAP_SIZE = 1920 * 1080               # Consts.h:39-41
MATERIAL_NONE = -1                          # Consts.h:44
TEXTURE_NONE = -1                           # Consts.h:45
PI = F(3.14159265358979323846)              # Consts.h:51
STANDARD_LUNINANCE_STRENGTH = F(0.1)        # Consts.h:52
SKYBOX_LUNINANCE_STRENGTH = F(0.2)          # Consts.h:53
# types.h: PrimitiveType, CameraType, GraphicsLevel, AdvancedIllumination, PostProcessingEffect
ptSphere, ptCylinder, ptTriangle, ptCheckboard, ptCamera, ptXYPlane, ptYZPlane, ptXZPlane = range(8)
ptMagicCarpet, ptEnvironment, ptEllipsoid, ptQuad, ptCone = 8, 9, 10, 11, 12
ctPerspective, ctOrthographic, ctAnaglyph, ctVR, ctPanoramic, ctAntialiazed, ctVolumeRendering = range(7)
glNoShading, glPhong, glPhongAndBlinn, glReflectionsAndRefractions, glFull = range(5)
aiNone, aiBasic, aiFull, aiRandomIllumination = range(4)
aeNone, aeFog = 0, 1
ftRGB, ftBGR = 0, 1
ppe_none, ppe_depthOfField, ppe_ambientOcclusion, ppe_radiosity, ppe_filter, ppe_cartoon = range(6)

_libm = ctypes.CDLL("libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = ctypes.c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [ctypes.c_float]


def cosf(a):
    return F(_libm.cosf(float(a)))


def sinf(a):
    return F(_libm.sinf(float(a)))


# ---- helper_math.h ---------------------------------------------------------------------------------------------
def v(x, y, z):
    return (F(x), F(y), F(z))


def add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def scale(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def neg(a):
    return (-a[0], -a[1], -a[2])


def dot(a, b):                                   # HM:1248-1251
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def length(a):                                   # HM:1291-1294
    return np.sqrt(dot(a, a))


def normalize(a):                                # HM:1309-1313 with HM:62-65
    inv = F(1.0) / np.sqrt(dot(a, a))
    return (a[0] * inv, a[1] * inv, a[2] * inv)


def cross(b, c):                                 # VU:46-53
    return (b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0])


def reflection(i, n):                            # VU:63-66: r = i - 2.f * dot(i, n) * n
    k = F(2.0) * dot(i, n)
    return sub(i, scale(n, k))


def refraction(incident, n1, normal, n2):        # VU:76-90
    refracted = incident
    if n2 != F(0):
        eta = n1 / n2
        c1 = -dot(incident, normal)
        cs2 = F(1) - eta * eta * (F(1) - c1 * c1)
        if cs2 >= F(0):
            refracted = add(scale(incident, eta), scale(normal, eta * c1 - np.sqrt(cs2)))
    return refracted


def project(A, B):                               # VU:95-98
    return scale(B, dot(A, B) / dot(B, B))


def saturate(c):                                 # VU:31-43
    return tuple(F(1) if (F(0) if x < F(0) else x) > F(1) else (F(0) if x < F(0) else x) for x in c)


def vector_rotation(p, center, angles):          # VU:104-142
    cx, cy, cz = cosf(angles[0]), cosf(angles[1]), cosf(angles[2])
    sx, sy, sz = sinf(angles[0]), sinf(angles[1]), sinf(angles[2])
    return _rotate(p, center, cx, cy, cz, sx, sy, sz)


def _rotate(p, center, cx, cy, cz, sx, sy, sz):
    x, y, z = p[0] - center[0], p[1] - center[1], p[2] - center[2]
    ry = y * cx - z * sx
    rz = y * sx + z * cx
    y, z = ry, rz
    rz = z * cy - x * sy
    rx = z * sy + x * cy
    z, x = rz, rx
    rx = x * cz - y * sz
    ry = x * sz + y * cz
    return (rx + center[0], ry + center[1], rz + center[2])


def to_int(x):
    """C's (int) of a float: truncation towards zero (the scenes of the tests stay far inside int's range)"""
    return int(x)


# ---- the scene as the kernels get it ------------------------------------------------------------------------------
class Scene:
    def __init__(self, flat, randoms=None):
        self.boxes = flat.boxes
        self.prims = flat.primitives
        self.lights = flat.lights
        self.materials = flat.materials
        self.randoms = np.asarray(flat.randoms if randoms is None else randoms, np.float32)
        # plain Python views of the records: field access on numpy records is the slow part
        self.B = [dict(lo=v(*b["min"]), hi=v(*b["max"]), n=int(b["nbPrimitives"]), start=int(b["startIndex"]),
                       skip=int(b["indexForNextBox"][0])) for b in self.boxes]
        self.P = [dict(p0=v(*p["p0"]), p1=v(*p["p1"]), p2=v(*p["p2"]), n0=v(*p["n0"]), n1=v(*p["n1"]), n2=v(*p["n2"]),
                       size=v(*p["size"]), type=int(p["type"]), index=int(p["index"]), mat=int(p["materialId"]))
                  for p in self.prims]
        self.M = [dict(illum=tuple(F(x) for x in m["innerIllumination"]), color=tuple(F(x) for x in m["color"]),
                       specular=tuple(F(x) for x in m["specular"]), reflection=F(m["reflection"]),
                       refraction=F(m["refraction"]), transparency=F(m["transparency"]), opacity=F(m["opacity"]),
                       attributes=tuple(int(x) for x in m["attributes"]), texture=int(m["textureIds"][0]),
                       mapping=tuple(int(x) for x in m["textureMapping"]),
                       ao_texture=int(m["advancedTextureIds"][2])) for m in self.materials]
        self.L = [dict(prim=int(l["primitiveId"]), mat=int(l["materialId"]), location=v(*l["location"]),
                       color=tuple(F(x) for x in l["color"])) for l in self.lights]

    def rnd(self, i):
        return F(self.randoms[i])

    ZERO_MATERIAL = dict(illum=(F(0),) * 4, color=(F(0),) * 4, specular=(F(0),) * 4, reflection=F(0), refraction=F(0),
                         transparency=F(0), opacity=F(0), attributes=(0, 0, 0, 0), texture=0, mapping=(0, 0, 0, 0), ao_texture=0)

    def material(self, i):
        """the reference's array has NB_MAX_MATERIALS + 1 records, zeros beyond the active ones (GPUKernel.cpp:319-370)"""
        return self.M[i] if 0 <= i < len(self.M) else self.ZERO_MATERIAL


class Ray:
    pass


def compute_ray_attributes(origin, direction):   # GI:36-44
    r = Ray()
    r.origin, r.direction = origin, direction
    with np.errstate(all="ignore"):
        r.inv = tuple((F(1) / d) if d != F(0) else F(1) for d in direction)
    r.signs = tuple(1 if i < F(0) else 0 for i in r.inv)
    return r


def box_intersection(box, r, t0, t1):            # GI:52-79
    par = (box["lo"], box["hi"])
    tmin = (par[r.signs[0]][0] - r.origin[0]) * r.inv[0]
    tmax = (par[1 - r.signs[0]][0] - r.origin[0]) * r.inv[0]
    tymin = (par[r.signs[1]][1] - r.origin[1]) * r.inv[1]
    tymax = (par[1 - r.signs[1]][1] - r.origin[1]) * r.inv[1]
    if tmin > tymax or tymin > tmax:
        return False
    if tymin > tmin:
        tmin = tymin
    if tymax < tmax:
        tmax = tymax
    tzmin = (par[r.signs[2]][2] - r.origin[2]) * r.inv[2]
    tzmax = (par[1 - r.signs[2]][2] - r.origin[2]) * r.inv[2]
    if tmin > tzmax or tzmin > tmax:
        return False
    if tzmin > tmin:
        tmin = tzmin
    if tzmax < tmax:
        tmax = tzmax
    return bool(tmin < t1 and tmax > t0)


def skybox_mapping(si, s, origin, target):       # GI:87-151 for a skybox material whose texture mapping is 0 x 0
    """Such a material never gets as far as a texel: u < textureMapping.x (GI:133) fails for every u.  What is
    left of the function is the material's colour - through every one of its early returns as well."""
    m = s.material(si.skyboxMaterialId)
    assert m["mapping"][0] == 0 and m["mapping"][1] == 0, "textured skyboxes are outside this model"
    return m["color"][:3]


# ---- primitive tests ----------------------------------------------------------------------------------------------
# each returns None (miss) or (intersection, normal, shadowIntensity[, areas])
def sphere_intersection(si, s, p, r):            # GI:220-284
    eps = F(si.geometryEpsilon)
    back = False
    O_C = sub(r.origin, p["p0"])
    d_ = normalize(r.direction)
    a = F(2) * dot(d_, d_)
    b = F(2) * dot(O_C, d_)
    c = dot(O_C, O_C) - (p["size"][0] * p["size"][0])
    d = b * b - F(2) * a * c
    if d <= F(0) or a == F(0):
        return None
    rt = np.sqrt(d)
    t1 = (-b - rt) / a
    t2 = (-b + rt) / a
    if t1 <= eps and t2 <= eps:
        return None
    if t1 <= eps:
        t = t2
        back = True
    elif t2 <= eps:
        t = t1
    else:
        t = t1 if t1 < t2 else t2
    if t < eps:
        return None
    inter = add(r.origin, scale(d_, t))
    m = s.material(p["mat"])
    if m["attributes"][1] == 0:
        normal = sub(inter, p["p0"])
    else:
        # procedural: GI:269-273, cos / sin of (int timestamp + float coordinate), in double here (module docstring)
        ts = F(si.timestamp)
        nc = (p["p0"][0] + F(0.008) * p["size"][0] * F(math.cos(float(ts + inter[0]))),
              p["p0"][1] + F(0.008) * p["size"][1] * F(math.sin(float(ts + inter[1]))),
              p["p0"][2] + F(0.008) * p["size"][2] * F(math.sin(float(F(math.cos(float(ts + inter[2])))))))
        normal = sub(inter, nc)
    normal = normalize(normal)
    if back:
        normal = scale(normal, F(-1))
    rr = dot(d_, normal)
    shadow = (F(1) - abs(rr)) if m["transparency"] != F(0) else F(1)
    return inter, normal, shadow


def ellipsoid_intersection(si, s, p, r):         # GI:159-212
    eps = F(si.geometryEpsilon)
    O_C = sub(r.origin, p["p0"])
    d_ = normalize(r.direction)
    sx, sy, sz = p["size"]
    a = ((d_[0] * d_[0]) / (sx * sx)) + ((d_[1] * d_[1]) / (sy * sy)) + ((d_[2] * d_[2]) / (sz * sz))
    b = ((F(2) * O_C[0] * d_[0]) / (sx * sx)) + ((F(2) * O_C[1] * d_[1]) / (sy * sy)) + ((F(2) * O_C[2] * d_[2]) / (sz * sz))
    c = ((O_C[0] * O_C[0]) / (sx * sx)) + ((O_C[1] * O_C[1]) / (sy * sy)) + ((O_C[2] * O_C[2]) / (sz * sz)) - F(1)
    d = (b * b) - (F(4) * a * c)
    if d < F(0) or a == F(0) or b == F(0) or c == F(0):
        return None
    d = np.sqrt(d)
    t1 = (-b + d) / (F(2) * a)
    t2 = (-b - d) / (F(2) * a)
    if t1 <= eps and t2 <= eps:
        return None
    if t1 <= eps:
        t = t2
    elif t2 <= eps:
        t = t1
    else:
        t = t1 if t1 < t2 else t2
    if t < eps:
        return None
    inter = add(r.origin, scale(d_, t))
    n = sub(inter, p["p0"])
    n = (F(2) * n[0] / (sx * sx), F(2) * n[1] / (sy * sy), F(2) * n[2] / (sz * sz))
    return inter, normalize(n), F(1)


def cylinder_intersection(si, s, p, r):          # GI:293-349; the cone's is the same text, GI:358-416
    eps = F(si.geometryEpsilon)
    O_C = sub(r.origin, p["p0"])
    d_ = r.direction
    n = cross(d_, p["n1"])
    ln = length(n)
    if ln < eps and ln > -eps:
        return None
    n = normalize(n)
    d = abs(dot(O_C, n))
    if d > p["size"][1]:
        return None
    O = cross(O_C, p["n1"])
    t = -dot(O, n) / ln
    if t < F(0):
        return None
    O = normalize(cross(n, p["n1"]))
    with np.errstate(all="ignore"):
        s_ = abs(np.sqrt(p["size"][0] * p["size"][0] - d * d) / dot(d_, O))
    t1 = t - s_
    t2 = t + s_
    inter = add(r.origin, scale(d_, t1))
    scale1 = dot(sub(inter, p["p0"]), p["n1"])
    scale2 = dot(sub(inter, p["p1"]), p["n1"])
    if scale1 < eps or scale2 > eps:
        inter = add(r.origin, scale(d_, t2))
        scale1 = dot(sub(inter, p["p0"]), p["n1"])
        scale2 = dot(sub(inter, p["p1"]), p["n1"])
        if scale1 < eps or scale2 > eps:
            return None
    V = sub(inter, p["p2"])
    normal = normalize(sub(V, project(V, p["n1"])))
    return inter, normal, F(1)


def wire_frame_mapping(x, y, width):             # TM:449-456
    X, Y = abs(to_int(x)), abs(to_int(y))
    return (X % 100 <= width) or (Y % 100 <= width)


def plane_intersection(si, s, p, r, reverse=False):   # GI:424-567 (untextured materials)
    m = s.material(p["mat"])
    collision = False
    rev = F(-1) if reverse else F(1)
    normal = p["n0"]
    o, d, p0, size = r.origin, r.direction, p["p0"], p["size"]
    ix = iy = iz = F(0)
    t = p["type"]
    with np.errstate(all="ignore"):
        if t in (ptMagicCarpet, ptCheckboard):
            iy = p0[1]
            y = o[1] - p0[1]
            if rev * d[1] < F(0) and rev * o[1] > rev * p0[1]:
                ix = o[0] + y * d[0] / -d[1]
                iz = o[2] + y * d[2] / -d[1]
                collision = abs(ix - p0[0]) < size[0] and abs(iz - p0[2]) < size[2]
        elif t == ptXZPlane:
            y = o[1] - p0[1]
            if rev * d[1] < F(0) and rev * o[1] > rev * p0[1]:
                ix = o[0] + y * d[0] / -d[1]
                iy = p0[1]
                iz = o[2] + y * d[2] / -d[1]
                collision = abs(ix - p0[0]) < size[0] and abs(iz - p0[2]) < size[2]
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(ix, iz, m["attributes"][3])
            if not collision and rev * d[1] > F(0) and rev * o[1] < rev * p0[1]:
                normal = neg(normal)
                ix = o[0] + y * d[0] / -d[1]
                iy = p0[1]
                iz = o[2] + y * d[2] / -d[1]
                collision = abs(ix - p0[0]) < size[0] and abs(iz - p0[2]) < size[2]
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(ix, iz, m["attributes"][3])
        elif t == ptYZPlane:
            x = o[0] - p0[0]
            if rev * d[0] < F(0) and rev * o[0] > rev * p0[0]:
                ix = p0[0]
                iy = o[1] + x * d[1] / -d[0]
                iz = o[2] + x * d[2] / -d[0]
                collision = abs(iy - p0[1]) < size[1] and abs(iz - p0[2]) < size[2]
                if m["illum"][0] != F(0):        # chessboard-like lights, GI:486-490
                    collision = collision and (to_int(abs(iz)) % 4000 < 2000 and to_int(abs(iy)) % 4000 < 2000)
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(iy, iz, m["attributes"][3])
            if not collision and rev * d[0] > F(0) and rev * o[0] < rev * p0[0]:
                normal = neg(normal)
                ix = p0[0]
                iy = o[1] + x * d[1] / -d[0]
                iz = o[2] + x * d[2] / -d[0]
                collision = abs(iy - p0[1]) < size[1] and abs(iz - p0[2]) < size[2]
                if m["illum"][0] != F(0):
                    collision = collision and (to_int(abs(iz)) % 4000 < 2000 and to_int(abs(iy)) % 4000 < 2000)
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(iy, iz, m["attributes"][3])
        elif t in (ptXYPlane, ptCamera):
            z = o[2] - p0[2]
            if rev * d[2] < F(0) and rev * o[2] > rev * p0[2]:
                iz = p0[2]
                ix = o[0] + z * d[0] / -d[2]
                iy = o[1] + z * d[1] / -d[2]
                collision = abs(ix - p0[0]) < size[0] and abs(iy - p0[1]) < size[1]
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(ix, iy, m["attributes"][3])
            if not collision and rev * d[2] > F(0) and rev * o[2] < rev * p0[2]:
                normal = neg(normal)
                iz = p0[2]
                ix = o[0] + z * d[0] / -d[2]
                iy = o[1] + z * d[1] / -d[2]
                collision = abs(ix - p0[0]) < size[0] and abs(iy - p0[1]) < size[1]
                if m["attributes"][2] == 2:
                    collision = collision and wire_frame_mapping(ix, iy, m["attributes"][3])
    if not collision:
        return None, (ix, iy, iz), normal
    shadow = F(1)
    assert t != ptCamera and m["texture"] == TEXTURE_NONE, "textured planes are outside this model"
    color = m["color"]
    if (color[0] + color[1] + color[2]) / F(3) >= F(si.transparentColor):      # GI:561
        return None, (ix, iy, iz), normal
    return ((ix, iy, iz), normal, shadow), (ix, iy, iz), normal


def triangle_intersection(si, s, p, r, processing_shadows):   # GI:575-659
    eps = F(si.geometryEpsilon)
    E01 = sub(p["p1"], p["p0"])
    E03 = sub(p["p2"], p["p0"])
    P = cross(r.direction, E03)
    det = dot(E01, P)
    if abs(det) < eps:
        return None
    T = sub(r.origin, p["p0"])
    a = dot(T, P) / det
    if a < F(0) or a > F(1):
        return None
    Q = cross(T, E01)
    b = dot(r.direction, Q) / det
    if b < F(0) or b > F(1):
        return None
    if (a + b) > F(1):
        E23 = sub(p["p0"], p["p1"])
        E21 = sub(p["p1"], p["p1"])
        P_ = cross(r.direction, E21)
        det_ = dot(E23, P_)
        if abs(det_) < eps:
            return None
        with np.errstate(all="ignore"):
            T_ = sub(r.origin, p["p2"])
            a_ = dot(T_, P_) / det_
            if a_ < F(0):
                return None
            Q_ = cross(T_, E23)
            b_ = dot(r.direction, Q_) / det_
            if b_ < F(0):
                return None
    t = dot(E03, Q) / det
    if t < F(0):
        return None
    inter = add(r.origin, scale(r.direction, t))
    v0, v1, v2 = sub(p["p0"], inter), sub(p["p1"], inter), sub(p["p2"], inter)
    areas = (F(0.5) * length(cross(v1, v2)), F(0.5) * length(cross(v0, v2)), F(0.5) * length(cross(v0, v1)))
    with np.errstate(all="ignore"):
        wn = add(add(scale(p["n0"], areas[0]), scale(p["n1"], areas[1])), scale(p["n2"], areas[2]))
        k = areas[0] + areas[1] + areas[2]
        normal = normalize((wn[0] / k, wn[1] / k, wn[2] / k))
    if si.doubleSidedTriangles:
        N = normalize(r.direction)
        # GI:643-647 as C parses it: the `else` belongs to the inner `if`
        if processing_shadows:
            if dot(N, normal) <= F(0):
                return None
            elif dot(N, normal) >= F(0):
                return None
    d_ = normalize(r.direction)
    rr = dot(d_, normal)
    if rr > F(0):
        normal = scale(normal, F(-1))
    return inter, normal, F(1), areas


def _dispatch_closest(si, s, p, r):              # the switch of GI:712-747
    """-> (hit?, intersection, normal, areas): like the reference's locals, intersection and normal keep what the
    last test wrote even on a miss (the distance of GI:749 is computed before `i` is looked at)"""
    areas = v(0, 0, 0)
    if si.extendedGeometry:
        t = p["type"]
        if t in (ptEnvironment, ptSphere):
            h = sphere_intersection(si, s, p, r)
        elif t == ptCylinder or t == ptCone:
            h = cylinder_intersection(si, s, p, r)
        elif t == ptEllipsoid:
            h = ellipsoid_intersection(si, s, p, r)
        elif t == ptTriangle:
            h = triangle_intersection(si, s, p, r, False)
            if h:
                return True, h[0], h[1], h[3]
            return False, None, None, areas
        else:
            h, inter, normal = plane_intersection(si, s, p, r)
            return (h is not None), inter, normal, areas
        if h:
            return True, h[0], h[1], areas
        return False, None, None, areas
    h = triangle_intersection(si, s, p, r, False)
    if h:
        return True, h[0], h[1], h[3]
    return False, None, None, areas


def intersection_with_primitives(si, s, origin, target, iteration, current_material, color_box):   # GI:667-772
    """-> (hit?, closestPrimitive, closestIntersection, closestNormal, closestAreas); color_box is a 3-list, in/out"""
    vd = F(si.viewDistance)
    min_distance = vd if iteration < 2 else vd / F(iteration + 1)
    r = compute_ray_attributes(origin, sub(target, origin))
    found, best = False, (None, None, None, None)
    c = 0
    n_boxes = len(s.B)
    while c < n_boxes:
        box = s.B[c]
        if box_intersection(box, r, F(0), min_distance):
            if si.renderBoxes != 0:
                m = s.material((box["start"] % 4294967296) % NB_MAX_MATERIALS)    # GI:695 (unsigned NB_MAX_MATERIALS)
                for k in range(3):
                    color_box[k] = color_box[k] + m["color"][k] / F(200)
            else:
                for k in range(box["n"]):
                    p = s.P[box["start"] + k]
                    m = s.material(p["mat"])
                    if m["attributes"][0] == 0 or (m["attributes"][0] == 1 and current_material != p["mat"]):
                        hit, inter, normal, areas = _dispatch_closest(si, s, p, r)
                        if hit:
                            distance = length(sub(inter, r.origin))
                            if distance > F(si.geometryEpsilon) and distance < min_distance:
                                min_distance = distance
                                best = (box["start"] + k, inter, normal, areas)
                                found = True
            c += 1
        else:
            c += box["skip"]
    return (found,) + best


def process_shadows(si, s, lamp_center, origin, light_id, iteration, object_id):   # GI:798-908
    """-> (shadow, (r, g, b) of the shadow colour)"""
    result = F(0)
    color = [F(0), F(0), F(0)]
    direction = sub(lamp_center, origin)
    r = compute_ray_attributes(add(origin, scale(normalize(direction), F(si.rayEpsilon))), direction)
    vd = F(si.viewDistance)
    min_distance = vd if iteration < 2 else vd / F(iteration + 1)
    limit = F(si.shadowIntensity)
    c = 0
    while result < limit and c < len(s.B):
        box = s.B[c]
        if box_intersection(box, r, F(0), min_distance):
            k = 0
            while result < limit and k < box["n"]:
                p = s.P[box["start"] + k]
                m = s.material(p["mat"])
                if p["index"] != light_id and p["index"] != object_id and m["attributes"][0] == 0:
                    hit = None
                    if si.extendedGeometry:
                        t = p["type"]
                        if t == ptSphere:
                            hit = sphere_intersection(si, s, p, r)
                        elif t == ptEllipsoid:
                            hit = ellipsoid_intersection(si, s, p, r)
                        elif t == ptCylinder or t == ptCone:
                            hit = cylinder_intersection(si, s, p, r)
                        elif t == ptTriangle:
                            hit = triangle_intersection(si, s, p, r, True)
                        elif t == ptCamera:
                            hit = None
                        else:
                            hit = plane_intersection(si, s, p, r)[0]
                    else:
                        hit = triangle_intersection(si, s, p, r, True)
                    if hit:
                        inter, normal, shadow = hit[0], hit[1], hit[2]
                        O_I = sub(inter, r.origin)
                        O_L = r.direction
                        l = length(O_I)
                        if l > F(si.geometryEpsilon) and l < length(O_L):
                            ratio = shadow * limit
                            if m["transparency"] != F(0):
                                O_L = normalize(O_L)
                                a = abs(dot(O_L, normal))
                                rr = F(1) if m["transparency"] == F(0) else (F(1) - m["transparency"])
                                ratio = ratio * (rr * a)
                                for q in range(3):
                                    color[q] = color[q] + ratio * (F(0.3) - F(0.3) * m["color"][q])
                            result = result + ratio
                k += 1
            c += 1
        else:
            c += box["skip"]
    result = max(F(0), min(result, limit))
    return result, tuple(color)


def intersection_shader(si, s, p, inter):        # GS:36-124 for untextured materials
    m = s.material(p["mat"])
    c = [m["color"][0], m["color"][1], m["color"][2]]
    assert m["texture"] == TEXTURE_NONE, "textured materials are outside this model"
    if si.extendedGeometry and p["type"] == ptCheckboard:
        vd = F(si.viewDistance)
        x = to_int(vd + ((inter[0] - p["p0"][0]) / p["size"][0]))
        z = to_int(vd + ((inter[2] - p["p0"][2]) / p["size"][0]))
        # C's % keeps the sign of the dividend; x and z are positive here (viewDistance dominates)
        if (x % 2 == 0 and z % 2 == 0) or (x % 2 != 0 and z % 2 != 0):
            c = [F(1) - c[0], F(1) - c[1], F(1) - c[2]]
    return tuple(c)


class ShaderState:
    """the in/out arguments of primitiveShader that live across bounces (CRT:103, 120, 100)"""

    def __init__(self):
        self.closest_color = [F(0), F(0), F(0)]
        self.total_blinn = [F(0), F(0), F(0), F(0)]      # rBlinn
        self.shadow_intensity = F(0)


def primitive_shader(si, s, index, origin, normal, object_id, inter, iteration, st):   # GI:916-1080
    """-> (colour, normal): colour is what the function returns, the normal is in/out (GI:944-945)"""
    p = s.P[object_id]
    m = s.material(p["mat"])
    lamps = [F(0), F(0), F(0)]
    st.shadow_intensity = F(0)
    spec = (m["specular"][0], m["specular"][1], m["specular"][2])
    ic = intersection_shader(si, s, p, inter)
    normal = normalize(add(normal, v(0, 0, 0)))          # bumpNormal stays zero without a bump map
    if m["attributes"][2] == 1:
        return ic, normal
    cc = st.closest_color
    if si.graphicsLevel > glNoShading:
        for k in range(3):
            cc[k] = cc[k] * m["illum"][0]
        n_lights = len(s.L)
        for cpt in range(n_lights):
            lamp = (si.pathTracingIteration % n_lights) if si.pathTracingIteration >= NB_MAX_ITERATIONS else 0
            li = s.L[lamp]
            if li["prim"] != p["index"]:
                center = li["location"]
                t = (index + si.timestamp) % (MAX_BITMAP_SIZE - 3)
                lm = s.material(li["mat"])
                if si.pathTracingIteration >= NB_MAX_ITERATIONS:
                    a = lm["illum"][1] * F(10) * F(si.pathTracingIteration) / F(si.maxPathTracingIterations)
                    center = (center[0] + s.rnd(t) * a, center[1] + s.rnd(t + 1) * a, center[2] + s.rnd(t + 2) * a)
                light_ray = sub(center, inter)
                light_ray_length = length(light_ray)
                if light_ray_length < lm["illum"][2]:
                    shadow_color = (F(0), F(0), F(0))
                    light_ray = normalize(light_ray)
                    lambert = m["illum"][0] + dot(normal, light_ray)
                    if lambert > F(0) and si.graphicsLevel > 3 and iteration < 4 and m["illum"][0] == F(0):
                        st.shadow_intensity, shadow_color = process_shadows(si, s, center, inter, li["prim"], iteration,
                                                                            object_id)
                    if si.graphicsLevel > glNoShading:
                        photon = np.sqrt(light_ray_length / lm["illum"][2])
                        photon = F(1) if photon > F(1) else photon
                        photon = F(0) if photon < F(0) else photon
                        lambert = lambert * ((-m["transparency"]) if lambert < F(0) else F(1))
                        if li["mat"] != MATERIAL_NONE:
                            lambert = lambert * lm["illum"][0]
                        else:
                            lambert = lambert * li["color"][3]
                        if m["illum"][3] != F(0):
                            lambert = lambert * (F(1) + s.rnd(t) * m["illum"][3] * F(100))
                        lambert = lambert * (F(1) - st.shadow_intensity)
                        lambert = lambert + F(si.backgroundColor[3])
                        lambert = lambert * (F(1) - photon)
                        for k in range(3):
                            lamps[k] = lamps[k] + (lambert * li["color"][k] - shadow_color[k])
                        if si.graphicsLevel > 1 and st.shadow_intensity < F(si.shadowIntensity):
                            view_ray = normalize(sub(inter, origin))
                            blinn_dir = sub(light_ray, view_ray)
                            temp = np.sqrt(dot(blinn_dir, blinn_dir))
                            if temp != F(0):
                                blinn_dir = scale(blinn_dir, F(1) / temp)
                                term = dot(blinn_dir, normal)
                                term = F(0) if term < F(0) else term
                                term = spec[0] * F(math.pow(float(term), float(spec[1])))
                                term = term * (F(1) - photon)
                                # float4 * float * float, left to right (HM operator*)
                                for k in range(4):
                                    st.total_blinn[k] = st.total_blinn[k] + li["color"][k] * li["color"][3] * term
                                st.total_blinn[3] = spec[2]
            for k in range(3):
                cc[k] = cc[k] + ic[k] * lamps[k]
            assert m["ao_texture"] == TEXTURE_NONE
            cc[:] = saturate(cc)
            st.total_blinn[:] = saturate(st.total_blinn)
    else:
        cc[:] = list(ic)
    return tuple(cc), normal


# ---- launchRayTracing, CRT:69-408 -----------------------------------------------------------------------------------
def launch_ray_tracing(si, s, index, ray_origin, ray_target, dof_in):
    """-> (colour xyz, depthOfField, primitiveXYId as [x, y, z, w] with None where the function does not write)"""
    inter_color = (F(0), F(0), F(0))
    closest_inter = v(0, 0, 0)
    normal = v(0, 0, 0)
    closest_prim = -1
    carryon = True
    ro_o, ro_d = ray_origin, ray_target
    initial_refraction = F(1)
    iteration = 0
    ident = [-1, None, 0, 0]
    current_material = -2
    contributions = [F(0)] * (NB_MAX_ITERATIONS + 1)
    colors = [(F(0), F(0), F(0)) for _ in range(NB_MAX_ITERATIONS + 1)]
    recursive_blinn = [F(0), F(0), F(0)]
    st = ShaderState()
    color_box = [F(0), F(0), F(0)]
    latest = ray_origin
    ray_length = F(0)
    vd = F(si.viewDistance)
    dof = vd
    reflected_rays = -1
    rr_o = rr_d = None
    reflected_ratio = F(0)
    pt_o = pt_d = None
    pt_ratio = F(0)
    pt_color = (F(0), F(0), F(0))
    use_gi = False
    reflected_target = None     # uninitialised in the reference until a hit writes it
    current_max = 1 if si.graphicsLevel < glReflectionsAndRefractions else si.nbRayIterations + si.pathTracingIteration
    current_max = NB_MAX_ITERATIONS if current_max > NB_MAX_ITERATIONS else current_max
    eps_ray = F(si.rayEpsilon)
    areas = v(0, 0, 0)
    while iteration < current_max and ray_length < vd and carryon:
        hit, prim, inter, nrm, ar = intersection_with_primitives(si, s, ro_o, ro_d, iteration, current_material, color_box)
        carryon = hit
        if hit:
            closest_prim, closest_inter, normal, areas = prim, inter, nrm, ar
            p = s.P[closest_prim]
            m = s.material(p["mat"])
            current_material = p["mat"]
            att = [m["reflection"], m["transparency"], m["refraction"], m["opacity"]]
            if iteration == 0:
                colors[0] = (F(0), F(0), F(0))
                contributions[0] = F(1)
                latest = closest_inter
                dof = length(sub(closest_inter, ray_origin))
                if m["illum"][0] == F(0) and si.advancedIllumination in (aiBasic, aiFull):
                    t = (index + si.pathTracingIteration * 100 + si.timestamp) % (MAX_BITMAP_SIZE - 3)
                    pt_o = add(closest_inter, scale(normal, eps_ray))
                    d = (normal[0] + F(100) * s.rnd(t), normal[1] + F(100) * s.rnd(t + 1), normal[2] + F(100) * s.rnd(t + 2))
                    cos_theta = dot(normalize(d), normal)
                    if cos_theta < F(0):
                        d = neg(d)
                    pt_d = add(d, closest_inter)
                    pt_ratio = (F(1) - att[1]) * abs(cos_theta)
                    use_gi = True
                ident[0] = p["index"]
            st.total_blinn[3] = att[1]
            col, normal = primitive_shader(si, s, index, ro_o, normal, closest_prim, closest_inter, iteration, st)
            colors[iteration] = col
            ident[2] = to_int(F(ident[2]) + m["illum"][0] * F(256))          # int += float: CRT:190
            segment = length(sub(closest_inter, latest))
            latest = closest_inter
            transparency = att[1]
            a = F(0)
            if att[1] != F(0):
                refr = att[2]
                if initial_refraction == refr:
                    refr = F(1)
                    ln = segment * (att[3] * (F(1) - transparency))
                    ray_length = ray_length + ln
                    ray_length = vd if ray_length > vd else ray_length
                    a = ray_length / vd
                    colors[iteration] = tuple(c - a for c in colors[iteration])
                O_E = normalize(sub(closest_inter, ro_o))
                reflected_target = refraction(O_E, refr, normal, initial_refraction)
                contributions[iteration] = transparency - a
                initial_refraction = refr
                if reflected_rays == -1 and att[0] != F(0):
                    rdir = reflection(O_E, normal)
                    rr_o = add(closest_inter, scale(rdir, eps_ray))
                    rr_d = add(closest_inter, rdir)
                    reflected_ratio = att[0]
                    reflected_rays = iteration
            elif att[0] != F(0):
                O_E = normalize(sub(closest_inter, ro_o))
                reflected_target = reflection(O_E, normal)
                contributions[iteration] = att[0]
            else:
                carryon = False
                contributions[iteration] = F(1)
            # rBlinn /= (iteration + 1): float4 / int -> float division by (float)(iteration + 1)
            for k in range(4):
                st.total_blinn[k] = st.total_blinn[k] / F(iteration + 1)
            for k in range(3):
                recursive_blinn[k] = st.total_blinn[k] if st.total_blinn[k] > recursive_blinn[k] else recursive_blinn[k]
            if reflected_target is None:
                assert not carryon, "an uninitialised reflectedTarget would be used"
                reflected_target = v(0, 0, 0)
            ro_o = add(closest_inter, scale(reflected_target, eps_ray))
            ro_d = add(closest_inter, reflected_target)
            if si.pathTracingIteration != 0 and m["color"][3] != F(0):
                ratio = m["color"][3]
                ratio = ratio * (F(1000) if att[1] == F(0) else F(1))
                rindex = (index + si.timestamp) % (MAX_BITMAP_SIZE - 3)
                ro_d = (ro_d[0] + s.rnd(rindex) * ratio, ro_d[1] + s.rnd(rindex + 1) * ratio, ro_d[2] + s.rnd(rindex + 2) * ratio)
        else:
            bg = tuple(F(x) for x in si.backgroundColor)
            if si.skyboxMaterialId != MATERIAL_NONE:                         # CRT:271-276
                colors[iteration] = skybox_mapping(si, s, ro_o, ro_d)
                rad = colors[iteration][0] + colors[iteration][1] + colors[iteration][2]
                ident[2] = to_int(F(ident[2]) + ((rad * F(256)) if rad > F(2.5) else F(0)))
            elif si.gradientBackground:
                d = normalize(sub(ro_d, ro_o))
                angle = F(0.5) - dot(v(0, 1, 0), d)
                angle = F(1) if angle > F(1) else angle
                colors[iteration] = tuple((F(1) - angle) * c for c in bg[:3])
            else:
                colors[iteration] = bg[:3]
            contributions[iteration] = F(1)
        iteration += 1

    areas = v(0, 0, 0)
    if si.graphicsLevel >= glReflectionsAndRefractions and reflected_rays != -1:
        hit, prim, inter, nrm, ar = intersection_with_primitives(si, s, rr_o, rr_d, reflected_rays, current_material, color_box)
        if hit:
            closest_prim, closest_inter, normal = prim, inter, nrm
            col, normal = primitive_shader(si, s, index, rr_o, normal, closest_prim, closest_inter, reflected_rays, st)
            colors[reflected_rays] = tuple(c + k * reflected_ratio for c, k in zip(colors[reflected_rays], col))
            ident[3] = to_int(st.shadow_intensity * F(255))

    test = True
    if si.advancedIllumination in (aiBasic, aiFull) and si.pathTracingIteration >= NB_MAX_ITERATIONS:
        if use_gi and si.advancedIllumination == aiFull:
            hit, prim, inter, nrm, ar = intersection_with_primitives(si, s, pt_o, pt_d, 30, MATERIAL_NONE, color_box)
            if hit:
                closest_prim, closest_inter, normal = prim, inter, nrm
                p = s.P[closest_prim]
                if p["mat"] != MATERIAL_NONE:
                    m = s.material(p["mat"])
                    if m["illum"][0] == F(0):
                        colors[0] = tuple(c * m["illum"][0] * pt_ratio for c in m["color"][:3])
                        test = False
                    else:
                        colors[0] = tuple(c * pt_ratio for c in m["color"][:3])
                if test:
                    pt_ratio = pt_ratio * STANDARD_LUNINANCE_STRENGTH
                    m = s.material(p["mat"])
                    if m["illum"][0] == F(0):
                        colors[0] = tuple(c - F(si.shadowIntensity) for c in colors[0])
                    else:
                        pt_color, normal = primitive_shader(si, s, index, pt_o, normal, closest_prim, closest_inter,
                                                            iteration, st)
            elif si.skyboxMaterialId != MATERIAL_NONE:                       # CRT:363-367
                pt_color = skybox_mapping(si, s, pt_o, pt_d)
                pt_ratio = pt_ratio * SKYBOX_LUNINANCE_STRENGTH
        elif si.skyboxMaterialId != MATERIAL_NONE:                           # CRT:371-375
            # (pathTracingRay is uninitialised here when no first hit set it up: the colour does not depend on it)
            pt_color = skybox_mapping(si, s, pt_o, pt_d)
            pt_ratio = pt_ratio * SKYBOX_LUNINANCE_STRENGTH
        if test:
            colors[0] = tuple(c + k * pt_ratio for c, k in zip(colors[0], pt_color))

    if test:
        for i in range(iteration - 2, -1, -1):
            colors[i] = tuple(colors[i][k] * (F(1) - contributions[i]) + colors[i + 1][k] * contributions[i] for k in range(3))
        inter_color = tuple(colors[0][k] + recursive_blinn[k] for k in range(3))
    else:
        inter_color = colors[0]

    D1 = vd * F(0.95)
    if si.atmosphericEffect == aeFog and dof > D1:
        D2 = vd * F(0.05)
        a = dof - D1
        b = F(1) - (a / D2)
        bg = tuple(F(x) for x in si.backgroundColor)
        inter_color = tuple(inter_color[k] * b + bg[k] * (F(1) - b) for k in range(3))
    ident[1] = iteration
    inter_color = tuple(inter_color[k] - color_box[k] for k in range(3))
    return inter_color, dof, ident


# ---- the camera kernels ---------------------------------------------------------------------------------------------
AA_ROTATED_GRID = ((F(3), F(5)), (F(5), F(-3)), (F(-3), F(-5)), (F(-5), F(3)))     # CRT:450


class Frame:
    """the per-pixel buffers a kernel reads and writes: pp[H][W] = [cx, cy, cz, cw, sx, sy, sz, sw], ids[H][W][4]"""

    def __init__(self, W, H, pp=None, ids=None):
        self.W, self.H = W, H
        self.pp = np.zeros((H, W, 8), np.float32) if pp is None else np.array(pp, np.float32)
        self.ids = np.zeros((H, W, 4), np.int32) if ids is None else np.array(ids, np.int32)


def _skip(si, frame, x, y):                      # CRT:454-458 and its copies in the other camera kernels
    it = si.pathTracingIteration
    return it > frame.ids[y, x, 1] and frame.ids[y, x, 3] == 0 and it > 0 and it <= NB_MAX_ITERATIONS


def _apply_ids(frame, x, y, ident):
    for k in range(4):
        if ident[k] is not None:
            frame.ids[y, x, k] = ident[k]


def _store(si, frame, x, y, color, dof, accumulate_max):
    """CRT:537-562 (accumulate_max) or the plain store of the other camera kernels (CRT:798-812, 911-925, 1028-1042)"""
    it = si.pathTracingIteration
    pp = frame.pp[y, x]
    if it == 0:
        pp[3] = dof
    if it <= NB_MAX_ITERATIONS:
        pp[0:3] = color
        if accumulate_max:
            pp[4:7] = color
    elif accumulate_max:
        z = frame.ids[y, x, 2]
        for k in range(3):
            pp[4 + k] = max(F(pp[4 + k]), color[k]) if z > 0 else color[k]
            pp[k] = F(pp[k]) + F(pp[4 + k])
    else:
        for k in range(3):
            pp[k] = F(pp[k]) + color[k]


def standard_renderer(si, ppi, s, frame, origin, direction, angles):   # CRT:437-563 (one device, whole frame)
    W, H = si.size_x, si.size_y
    origin, direction = v(*origin), v(*direction)
    angles = tuple(F(a) for a in angles)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            if _skip(si, frame, x, y):
                continue
            ro, rd = origin, direction
            center = origin if si.cameraType == ctVR else v(0, 0, 0)
            if ppi.type != ppe_depthOfField and si.pathTracingIteration >= NB_MAX_ITERATIONS:    # CRT:470-479
                a = F(ppi.param1) / F(20000)
                rindex = index + si.timestamp % (MAX_BITMAP_SIZE - 2)
                w = F(frame.pp[y, x, 3])
                ro = (ro[0] + s.rnd(rindex) * w * a, ro[1] + s.rnd(rindex + 1) * w * a, ro[2])
            dof = F(0)
            if si.cameraType == ctOrthographic:
                dx = ro[2] * F(0.001) * F(x - (W // 2))
                dy = -ro[2] * F(0.001) * F(y - (H // 2))
                rd = (dx, dy, rd[2])
                ro = (dx, dy, ro[2])
            else:
                ratio = F(W) / F(H)
                step_x = ratio * angles[3] / F(W)
                step_y = angles[3] / F(H)
                rd = (rd[0] - step_x * F(x - (W // 2)), rd[1] + step_y * F(y - (H // 2)), rd[2])
            ro = vector_rotation(ro, center, angles)
            rd = vector_rotation(rd, center, angles)
            color = [F(0), F(0), F(0)]
            r_o, r_d = ro, rd
            if si.cameraType == ctAntialiazed:
                for I in range(4):
                    r_o = (r_o[0] + AA_ROTATED_GRID[I][0], r_o[1] + AA_ROTATED_GRID[I][1], r_o[2])
                    c, dof, ident = launch_ray_tracing(si, s, index, r_o, r_d, dof)
                    _apply_ids(frame, x, y, ident)
                    color = [color[k] + c[k] for k in range(3)]
            elif si.pathTracingIteration >= NB_MAX_ITERATIONS:
                g = AA_ROTATED_GRID[si.pathTracingIteration % 4]
                r_d = (r_d[0] + g[0], r_d[1] + g[1], r_d[2])
            c, dof, ident = launch_ray_tracing(si, s, index, r_o, r_d, dof)
            _apply_ids(frame, x, y, ident)
            color = [color[k] + c[k] for k in range(3)]
            if si.advancedIllumination == aiRandomIllumination:
                rindex = (index + si.timestamp) % MAX_BITMAP_SIZE
                color = [color[k] + F(si.backgroundColor[k]) * s.rnd(rindex) * F(5) for k in range(3)]
            if si.cameraType == ctAntialiazed:
                color = [c_ / F(5) for c_ in color]
            _store(si, frame, x, y, color, dof, True)


def fish_eye_renderer(si, ppi, s, frame, origin, direction, angles):   # CRT:741-813
    W, H = si.size_x, si.size_y
    angles = tuple(F(a) for a in angles)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            if _skip(si, frame, x, y):
                continue
            ro, rd = v(*origin), v(*direction)
            if si.pathTracingIteration >= NB_MAX_ITERATIONS:
                rindex = (index + si.timestamp) % (MAX_BITMAP_SIZE - 3)
                a = F(si.pathTracingIteration) / F(si.maxPathTracingIterations)
                w = F(frame.pp[y, x, 3])
                rd = (rd[0] + s.rnd(rindex) * w * F(ppi.param2) * a, rd[1] + s.rnd(rindex + 1) * w * F(ppi.param2) * a,
                      rd[2] + s.rnd(rindex + 2) * w * F(ppi.param2) * a)
            dof = F(0)
            step_y = angles[3] / F(H)
            rd = (rd[0], rd[1] + step_y * F(y - (H // 2)), rd[2])
            step_x = F(2) * PI / F(W)
            turn = angles[1] + step_x * F(x)
            # vectorRotation with angles (0, turn, 0): cos 0 = 1, sin 0 = 0 exactly; cos / sin of the turn in double
            # (module docstring: cosf is specified to an error bound only, both sides take the rounded double)
            rd = _rotate(rd, ro, F(1), F(math.cos(float(turn))), F(1), F(0), F(math.sin(float(turn))), F(0))
            c, dof, ident = launch_ray_tracing(si, s, index, ro, rd, dof)
            _apply_ids(frame, x, y, ident)
            _store(si, frame, x, y, list(c), dof, False)


def anaglyph_renderer(si, ppi, s, frame, origin, direction, angles):   # CRT:840-926
    W, H = si.size_x, si.size_y
    angles = tuple(F(a) for a in angles)
    origin, direction = v(*origin), v(*direction)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            if _skip(si, frame, x, y):
                continue
            center = origin if si.cameraType == ctVR else v(0, 0, 0)
            dof = F(0)
            ratio = F(W) / F(H)
            step_x = ratio * angles[3] / F(W)
            step_y = angles[3] / F(H)
            sides = []
            for sign in (F(-1), F(1)):
                eo = ((origin[0] - F(si.eyeSeparation)) if sign < 0 else (origin[0] + F(si.eyeSeparation)), origin[1], origin[2])
                ed = (direction[0] - step_x * F(x - (W // 2)), direction[1] + step_y * F(y - (H // 2)), direction[2])
                eo = vector_rotation(eo, center, angles)
                ed = vector_rotation(ed, center, angles)
                c, dof, ident = launch_ray_tracing(si, s, index, eo, ed, dof)
                _apply_ids(frame, x, y, ident)
                sides.append(c)
            left, right = sides
            r1 = left[0] * F(0.299) + left[1] * F(0.587) + left[2] * F(0.114)
            color = [r1 + F(0), F(0) + right[1], F(0) + right[2]]
            _store(si, frame, x, y, color, dof, False)


def vision_renderer(si, ppi, s, frame, origin, direction, angles, focus_depth):   # CRT:953-1043
    """focus_depth: postProcessingBuffer[size.x / 2 * size.y / 2].colorInfo.w as the frame before left it (the
    reference reads it while the pixel's own thread may be writing it: CRT:972)"""
    W, H = si.size_x, si.size_y
    angles = tuple(F(a) for a in angles)
    origin, direction = v(*origin), v(*direction)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            if _skip(si, frame, x, y):
                continue
            with np.errstate(all="ignore"):
                focus = abs(F(focus_depth) - origin[2])
                eye_separation = F(si.eyeSeparation) * (direction[2] / focus)
            center = origin if si.cameraType == ctVR else v(0, 0, 0)
            dof = F(ppi.param1)
            half = W // 2
            ratio = F(W) / F(H)
            step_x = ratio * angles[3] / F(W)
            step_y = angles[3] / F(H)
            if x < half:
                eo = (origin[0] + eye_separation, origin[1], origin[2])
                ed = (direction[0] - step_x * F(x - (W // 2) + half // 2) + F(si.eyeSeparation),
                      direction[1] + step_y * F(y - (H // 2)), direction[2])
            else:
                eo = (origin[0] - eye_separation, origin[1], origin[2])
                ed = (direction[0] - step_x * F(x - (W // 2) - half // 2) - F(si.eyeSeparation),
                      direction[1] + step_y * F(y - (H // 2)), direction[2])
            eo = vector_rotation(eo, center, angles)
            ed = vector_rotation(ed, center, angles)
            c, dof, ident = launch_ray_tracing(si, s, index, eo, ed, dof)
            _apply_ids(frame, x, y, ident)
            color = list(c)
            if si.advancedIllumination == aiRandomIllumination:
                rindex = (index + si.timestamp) % MAX_BITMAP_SIZE
                color = [color[k] + F(si.backgroundColor[k]) * s.rnd(rindex) * F(5) for k in range(3)]
            _store(si, frame, x, y, color, dof, False)


# ---- k_default and the post-processing kernels ------------------------------------------------------------------------
def make_color(si, color, bitmap, index):        # GS:132-165
    c = [F(1) if x > F(1) else x for x in color[:3]]
    c = [F(0) if x < F(0) else x for x in c]
    flat = bitmap.reshape(-1)
    if si.frameBufferType == ftBGR:
        y = index // si.size_y
        x = index % si.size_x
        i = ((y + 1) * si.size_y - x - 1) * 3
        flat[i], flat[i + 1], flat[i + 2] = to_int(c[2] * F(255)), to_int(c[1] * F(255)), to_int(c[0] * F(255))
    else:
        i = index * 3
        flat[i], flat[i + 1], flat[i + 2] = to_int(c[0] * F(255)), to_int(c[1] * F(255)), to_int(c[2] * F(255))


def _divisor(si):
    return F(si.pathTracingIteration - NB_MAX_ITERATIONS + 1) if si.pathTracingIteration > NB_MAX_ITERATIONS else None


def k_default(si, frame, bitmap):                # CRT:1057-1073
    d = _divisor(si)
    for y in range(si.size_y):
        for x in range(si.size_x):
            c = [F(a) for a in frame.pp[y, x, 0:3]]
            if d is not None:
                c = [a / d for a in c]
            make_color(si, c, bitmap, y * si.size_x + x)


def k_depth_of_field(si, ppi, s, frame, bitmap):     # CRT:1081-1120
    W, H = si.size_x, si.size_y
    wh = W * H
    d = _divisor(si)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            c = [F(0), F(0), F(0)]
            depth = abs(F(frame.pp[y, x, 3]) - F(ppi.param1)) / F(si.viewDistance)
            for i in range(ppi.param3):
                ix = i % wh
                iy = (i + 1000) % wh
                xx = to_int(F(x) + depth * s.rnd(ix) * F(ppi.param2))
                yy = to_int(F(y) + depth * s.rnd(iy) * F(ppi.param2))
                if 0 <= xx < W and 0 <= yy < H:
                    li = yy * W + xx
                    if 0 <= li < wh:
                        c = [c[k] + F(frame.pp[yy, xx, k]) for k in range(3)]
                else:
                    c = [c[k] + F(frame.pp[y, x, k]) for k in range(3)]
            c = [a / F(ppi.param3) for a in c]
            if d is not None:
                c = [a / d for a in c]
            make_color(si, c, bitmap, index)


def k_ambient_occlusion(si, ppi, s, frame, bitmap):  # CRT:1128-1181
    W, H = si.size_x, si.size_y
    wh = W * H
    d = _divisor(si)
    for y in range(H):
        for x in range(W):
            occ = F(0)
            c = [F(a) for a in frame.pp[y, x, 0:3]]
            depth = F(frame.pp[y, x, 3])
            i = 0
            cnt = F(0)
            for X in range(-16, 16, 2):
                for Y in range(-16, 16, 2):
                    ix = i % wh
                    iy = (i + 100) % wh
                    i += 1
                    cnt = cnt + F(1)
                    xx = to_int(F(x) + (F(X) * F(ppi.param2) * s.rnd(ix) / F(10)))
                    yy = to_int(F(y) + (F(Y) * F(ppi.param2) * s.rnd(iy) / F(10)))
                    if 0 <= xx < W and 0 <= yy < H:
                        if F(frame.pp[yy, xx, 3]) >= depth:
                            occ = occ + F(1)
                    else:
                        occ = occ + F(1)
            occ = occ / cnt
            occ = occ + F(0.3)
            if occ < F(1):
                c = [a * occ for a in c]
            if d is not None:
                c = [a / d for a in c]
            make_color(si, saturate(c), bitmap, y * W + x)


def k_radiosity(si, ppi, s, frame, bitmap):      # CRT:1189-1228
    W, H = si.size_x, si.size_y
    wh = W * H
    it = si.pathTracingIteration
    div = (it - NB_MAX_ITERATIONS + 1) if it > NB_MAX_ITERATIONS else 1
    for y in range(H):
        for x in range(W):
            c = [F(0), F(0), F(0)]
            for i in range(ppi.param3):
                ix = (i + it) % wh
                iy = (i + 100 + it) % wh
                xx = to_int(F(x) + s.rnd(ix) * F(ppi.param2))
                yy = to_int(F(y) + s.rnd(iy) * F(ppi.param2))
                c = [c[k] + F(frame.pp[y, x, k]) for k in range(3)]
                if 0 <= xx < W and 0 <= yy < H:
                    z = F(frame.ids[yy, xx, 2])
                    c = [c[k] + F(frame.pp[yy, xx, k]) * z / F(256) for k in range(3)]
            c = [a / F(ppi.param3) for a in c]
            c = [a / F(div) for a in c]
            make_color(si, saturate(c), bitmap, y * W + x)


FILTER_SIZE = ((3, 3), (5, 5), (3, 3), (3, 3), (5, 5), (5, 5))                       # CRT:1249
FILTER_FACTORS = ((1.0, 128.0), (1.0, 0.0), (1.0, 0.0), (1.0, 0.0), (0.2, 0.0), (0.125, 0.0))   # CRT:1251-1252
FILTER_INFO = (                                                                       # CRT:1254-1289
    ((-1, -1, 0, 0, 0), (-1, 0, 1, 0, 0), (0, 1, 1, 0, 0), (0, 0, 0, 0, 0), (0, 0, 0, 0, 0)),
    ((0, 0, 0, 0, 0), (0, 0, 0, 0, 0), (-1, -1, 2, 0, 0), (0, 0, 0, 0, 0), (0, 0, 0, 0, 0)),
    ((-1, -1, -1, 0, 0), (-1, 9, -1, 0, 0), (-1, -1, -1, 0, 0), (0, 0, 0, 0, 0), (0, 0, 0, 0, 0)),
    ((0, 0.2, 0, 0, 0), (0.2, 0.2, 0.2, 0, 0), (0, 0.2, 0, 0, 0), (0, 0, 0, 0, 0), (0, 0, 0, 0, 0)),
    ((1, 0, 0, 0, 0), (0, 1, 0, 0, 0), (0, 0, 1, 0, 0), (0, 0, 0, 1, 0), (0, 0, 0, 0, 1)),
    ((-1, -1, -1, -1, -1), (-1, 2, 2, 2, -1), (-1, 2, 8, 2, -1), (-1, 2, 2, 2, -1), (-1, -1, -1, -1, -1)))


def k_filter(si, ppi, frame, bitmap):            # CRT:1236-1333
    W, H = si.size_x, si.size_y
    d = _divisor(si)
    f = ppi.param3
    for y in range(H):
        for x in range(W):
            lc = [F(0), F(0), F(0)]
            color = [F(0), F(0), F(0)]
            if 0 <= f < 6:      # `param3 < NB_FILTERS` compares as unsigned in the reference: a negative one fails
                for fx in range(FILTER_SIZE[f][0]):
                    for fy in range(FILTER_SIZE[f][1]):
                        image_x = (x - FILTER_SIZE[f][0] // 2 + fx + W) % W
                        image_y = (y - FILTER_SIZE[f][1] // 2 + fy + H) % H
                        c = [F(a) for a in frame.pp[image_y, image_x, 0:3]]
                        if d is not None:
                            c = [a / d for a in c]
                        k = F(FILTER_INFO[f][fx][fy])
                        lc = [lc[q] + c[q] * k for q in range(3)]
                fa, fb = F(FILTER_FACTORS[f][0]), F(FILTER_FACTORS[f][1])
                color = [color[q] + min(max(fa * lc[q] + fb / F(255), F(0)), F(1)) for q in range(3)]
            make_color(si, saturate(color), bitmap, y * W + x)


def k_cartoon(si, ppi, frame, bitmap):           # CRT:1341-1358
    for y in range(si.size_y):
        for x in range(si.size_x):
            with np.errstate(all="ignore"):
                depth = F(si.viewDistance) / abs(F(frame.pp[y, x, 3]) - F(ppi.param1))
            make_color(si, saturate([depth, depth, depth]), bitmap, y * si.size_x + x)


# ---- k_volumeRenderer, CRT:592-713 ----------------------------------------------------------------------------------
def intersections_with_primitives(si, ppi, s, index, origin, target):   # GI:1088-1265 (VOLUME_RENDERING_NORMALS is not
    """-> float4 colour.  `target` is ray.direction, a point."""         #  defined: GI:22, Consts.h:57)
    MAXDEPTH = 10                                                        # GI:1104
    r = compute_ray_attributes(origin, sub(target, origin))              # GI:1095-1098
    vd = F(si.viewDistance)
    # GI:1105-1112; one element more than the text declares: the shift of GI:1198-1200 starts at j = MAXDEPTH - 1 and
    # writes colors[MAXDEPTH], which nothing reads
    colors = [[F(0), F(0), F(0), vd] for _ in range(MAXDEPTH + 1)]
    nb_intersections = 0
    c = 0
    n_boxes = len(s.B)
    while c < n_boxes:                                                   # GI:1120-1221
        box = s.B[c]
        if box_intersection(box, r, F(0), vd):
            for k in range(box["n"]):
                p = s.P[box["start"] + k]
                m = s.material(p["mat"])
                hit, inter, normal, areas = _dispatch_closest(si, s, p, r)     # the switch of GI:1132-1166 is GI:712-747's
                if hit:
                    dist = length(sub(inter, r.origin))                  # GI:1169
                    if dist > F(ppi.param1):                             # GI:1170
                        nb_intersections += 1
                        color = (m["color"][0], m["color"][1], m["color"][2])
                        if si.graphicsLevel != glNoShading:              # GI:1174-1192 (the product of GI:1176 is overwritten)
                            st = ShaderState()
                            st.closest_color = [m["color"][0], m["color"][1], m["color"][2]]
                            color, normal = primitive_shader(si, s, index, r.origin, normal, box["start"] + k, inter, 0, st)
                        for i in range(MAXDEPTH):                        # GI:1193-1215
                            if dist < colors[i][3]:
                                a = dot(normalize(sub(target, origin)), normal)
                                for j in range(MAXDEPTH - 1, i - 1, -1):
                                    colors[j + 1] = list(colors[j])
                                colors[i] = [color[0] * np.abs(a), color[1] * np.abs(a), color[2] * np.abs(a), dist]
                                break
            c += 1
        else:
            c += box["skip"]
    bgw = F(si.backgroundColor[3])
    color = [colors[0][k] * bgw for k in range(4)]                       # GI:1227
    if nb_intersections > 0:                                             # GI:1228-1262
        N = 0
        D = colors[0][3]
        precision = 500
        step = vd / F(precision)
        alpha = F(1) / F(ppi.param2)
        cc = 0
        i = 0
        while i < precision and cc < MAXDEPTH - 1:
            if D > colors[cc][3] and N == 0:
                for k in range(3):
                    color[k] = color[k] + colors[cc][k] * alpha
            D = D + step
            if D >= colors[cc + 1][3]:
                cc += 1
            i += 1
        color[3] = F(0)
        # GI:1254: normalize(color) returns a value nobody takes
    color[3] = colors[0][3]                                              # GI:1263
    return color


def volume_renderer(si, ppi, s, frame, origin, direction, angles):      # CRT:592-713 for cameraType ctVolumeRendering
    """(CRT:1777: the only camera type that reaches this kernel, so its orthographic, VR and five-ray branches are dead)"""
    W, H = si.size_x, si.size_y
    origin, direction = v(*origin), v(*direction)
    angles = tuple(F(a) for a in angles)
    for y in range(H):
        for x in range(W):
            index = y * W + x
            if _skip(si, frame, x, y):                                   # CRT:610-615
                continue
            ro, rd = origin, direction
            center = v(0, 0, 0)
            if ppi.type != ppe_depthOfField and si.pathTracingIteration >= NB_MAX_ITERATIONS:    # CRT:626-633
                a = F(ppi.param1) / F(20000)
                rindex = index + si.timestamp % (MAX_BITMAP_SIZE - 2)
                w = F(frame.pp[y, x, 3])
                ro = (ro[0] + s.rnd(rindex) * w * a, ro[1] + s.rnd(rindex + 1) * w * a, ro[2])
            dof = F(0)                                                   # CRT:635: nothing assigns it afterwards
            ratio = F(W) / F(H)                                          # CRT:645-651
            step_x = ratio * angles[3] / F(W)
            step_y = angles[3] / F(H)
            rd = (rd[0] - step_x * F(x - (W // 2)), rd[1] + step_y * F(y - (H // 2)), rd[2])
            ro = vector_rotation(ro, center, angles)                     # CRT:654-655
            rd = vector_rotation(rd, center, angles)
            g = AA_ROTATED_GRID[si.pathTracingIteration % 4]             # CRT:670-671: every pass
            r_d = (rd[0] + g[0], rd[1] + g[1], rd[2])
            _apply_ids(frame, x, y, [-1, 1, 0, None])                    # CRT:59-61
            c = intersections_with_primitives(si, ppi, s, index, ro, r_d)
            color = [F(0) + c[k] for k in range(3)]                      # CRT:657, 673
            if si.advancedIllumination == aiRandomIllumination:          # CRT:677-682
                rindex = (index + si.timestamp) % MAX_BITMAP_SIZE
                color = [color[k] + F(si.backgroundColor[k]) * s.rnd(rindex) * F(5) for k in range(3)]
            _store(si, frame, x, y, color, dof, True)                    # CRT:686-712


def render(si, ppi, flat, origin, direction, angles, pp=None, ids=None, randoms=None, focus_depth=0.0):
    """One cudaRender (CRT:1680-1908 for one device): the camera kernel the dispatch picks, then the post-processing
    kernel.  -> (pp (H, W, 8) float32, ids (H, W, 4) int32, bitmap (H, W, 3) uint8)"""
    s = Scene(flat, randoms)
    frame = Frame(si.size_x, si.size_y, pp, ids)
    if si.cameraType == ctAnaglyph:
        anaglyph_renderer(si, ppi, s, frame, origin, direction, angles)
    elif si.cameraType == ctVR:
        vision_renderer(si, ppi, s, frame, origin, direction, angles, focus_depth)
    elif si.cameraType == ctPanoramic:
        fish_eye_renderer(si, ppi, s, frame, origin, direction, angles)
    elif si.cameraType == ctVolumeRendering:
        volume_renderer(si, ppi, s, frame, origin, direction, angles)
    else:
        standard_renderer(si, ppi, s, frame, origin, direction, angles)
    bitmap = np.zeros((si.size_y, si.size_x, 3), np.uint8)
    if ppi.type == ppe_depthOfField:
        k_depth_of_field(si, ppi, s, frame, bitmap)
    elif ppi.type == ppe_ambientOcclusion:
        k_ambient_occlusion(si, ppi, s, frame, bitmap)
    elif ppi.type == ppe_radiosity:
        k_radiosity(si, ppi, s, frame, bitmap)
    elif ppi.type == ppe_filter:
        k_filter(si, ppi, frame, bitmap)
    elif ppi.type == ppe_cartoon:
        k_cartoon(si, ppi, frame, bitmap)
    else:
        k_default(si, frame, bitmap)
    return frame.pp, frame.ids, bitmap
