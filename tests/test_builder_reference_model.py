"""The host box-tree builder against a line-by-line model of the REFERENCE'S builder.

The flattened tree decides results: its depth-first order is the order of the primitive tests (first visited
wins a tie, GeometryIntersections.cuh:751; shadows accumulate in it).  sol-r_amd/host/GPUKernel.cpp restates
the reference's builder with other containers, so a self-consistent but wrong tree would pass every
render test.  This file therefore re-derives the tree INDEPENDENTLY, in plain Python with numpy binary32
scalars, one statement of solr/engines/GPUKernel.cpp at a time (cited on the right), and the builder must
produce exactly that: node for node the bounds, primitive counts, start indices and skip pointers, the
primitive order, the lamps and the light list.  What the model spells out, because the reference does:

  * level 0: cell of p0 in a 6400^3 grid over the scene extent, key 1 + 1000 (X 6400^2 + Y 6400 + Z) in
    UNSIGNED 32-bit arithmetic - it wraps (:938-941);
  * level d >= 1: cell of the child's centre in a grid of `nbBoxes`^3 (the primitive count, divided by 4 per
    level), key X n^2 + Y n + Z + 1 in SIGNED 32-bit arithmetic - it wraps too, to negative values, and the
    std::map is keyed by unsigned int (GPUKernel.h:75): those sort LAST (:1011-1017);
  * emissive primitives: a level-0 cell is still created for them (empty, bounds left at the +-1e6 seed,
    centre 0) but they are listed in box 0 of the top level (:954-975);
  * the flattening treats the FIRST entry of the top level as the lamp box: bounds forced to +-viewDistance, its
    list emitted as primitives, then recursed into AS IF it held child keys (:1169-1252);
  * the scene extent a fresh kernel starts from is whatever its allocation holds - the members are only set by
    cleanup() (:394-399) - zeros in the mirror, so the grid spans the primitives and the origin;
  * boxes without entries are not emitted; skip pointer = size of the subtree (:1096, 1143).
The first scene below is small enough to follow by hand; its expected tree is also written out literally.
"""
import numpy as np
import pytest

F = np.float32
AABB_MAGIC_NUMBER = 6400          # GPUKernel.cpp:69


def i32(v):
    """wrap to a signed 32-bit int (what the reference's int arithmetic does on overflow in practice)"""
    v = int(v) & 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def u32(v):
    return int(v) & 0xFFFFFFFF


def cast_int(f):
    """static_cast<int>(float): truncation toward zero"""
    return int(np.trunc(np.float64(f)))


class Box:
    def __init__(self, vd):
        self.lo = [F(vd)] * 3            # :952-958 / :1021-1026: seeded inside out
        self.hi = [F(-vd)] * 3
        self.entries = []
        self.center = [F(0)] * 3


class ReferenceBuilder:
    """GPUKernel::compactBoxes(true) + streamDataToGPU of a fresh kernel, statement by statement"""

    def __init__(self, solr, view_distance, primitives, emissive):
        self.solr = solr
        self.vd = F(view_distance)
        self.prims = primitives            # index -> dict(type, p0, p1, p2, size, material)
        self.emissive = emissive           # material id -> bool
        # the scene extent: m_minPos / m_maxPos are plain members the constructor does not set and only cleanup()
        # (the destructor, a viewer's scene switch) resets to +-viewDistance (:394-399): a fresh kernel driven
        # through the flat API starts from what the allocation holds - zeros here, as in the mirror - and
        # setPrimitive grows it (:671-678)
        self.min = [F(0)] * 3
        self.max = [F(0)] * 3
        for p in primitives.values():
            for a in range(3):
                self.min[a] = min(F(p["p0"][a]), self.min[a])
                self.max[a] = max(F(p["p0"][a]), self.max[a])
        self.levels = {}                   # depth -> {key: Box}
        self.wrapped = False

    def level(self, d):
        return self.levels.setdefault(d, {})

    def steps(self, box_size):
        s = [(self.max[a] - self.min[a]) / F(box_size) for a in range(3)]      # :920-922, :998-1000
        return [F(1) if v == 0 else v for v in s]                               # :924-926

    def cell(self, point, s):
        return [cast_int((F(point[a]) - self.min[a]) / s[a]) for a in range(3)]   # :938-940, :1011-1013

    def update_bounding_box(self, box):                                          # :741-839
        solr = self.solr
        box.lo, box.hi = [F(1000000)] * 3, [F(-1000000)] * 3
        for p in box.entries:
            prim = self.prims[p]
            if prim["type"] == solr.ptTriangle:
                c0 = [min(min(F(prim["p0"][a]), F(prim["p1"][a])), F(prim["p2"][a])) for a in range(3)]
                c1 = [max(max(F(prim["p0"][a]), F(prim["p1"][a])), F(prim["p2"][a])) for a in range(3)]
            elif prim["type"] == solr.ptCylinder:
                c0 = [min(F(prim["p0"][a]), F(prim["p1"][a])) for a in range(3)]
                c1 = [max(F(prim["p0"][a]), F(prim["p1"][a])) for a in range(3)]
            else:
                c0 = c1 = [F(v) for v in prim["p0"]]
            p0 = [c0[a] if c0[a] <= c1[a] else c1[a] for a in range(3)]
            p1 = [c0[a] if c0[a] > c1[a] else c1[a] for a in range(3)]
            size = [F(v) for v in prim["size"]]
            if prim["type"] in (solr.ptCylinder, solr.ptSphere, solr.ptCone):   # :795-807: the radius on every axis
                size = [size[0]] * 3
            p0 = [p0[a] - size[a] for a in range(3)]
            p1 = [p1[a] + size[a] for a in range(3)]
            for a in range(3):
                if p0[a] < box.lo[a]:
                    box.lo[a] = p0[a]
                if p1[a] > box.hi[a]:
                    box.hi[a] = p1[a]
        box.center = [(box.lo[a] + box.hi[a]) / F(2) for a in range(3)]          # :834-836

    def update_outer_bounding_box(self, box, depth):                            # :843-892
        box.lo, box.hi = [self.vd] * 3, [-self.vd] * 3
        for key in box.entries:
            child = self.level(depth).setdefault(key, Box(0))                   # operator[]: creates what is missing
            for a in range(3):
                if box.lo[a] > child.lo[a]:
                    box.lo[a] = child.lo[a]
                if box.hi[a] < child.hi[a]:
                    box.hi[a] = child.hi[a]
        box.center = [(box.lo[a] + box.hi[a]) / F(2) for a in range(3)]

    def compact_boxes(self):
        n = len(self.prims)
        tree_depth = 2                                                           # :198, the constructor's value
        self.level(tree_depth).setdefault(0, Box(self.vd))                      # :1049 resetBox(levels[depth][0])
        tree_depth, nb = 0, n                                                    # :1056-1062
        while nb > 2:
            tree_depth += 1
            nb //= 4
        # processBoxes(6400), :917-992
        s = self.steps(AABB_MAGIC_NUMBER)
        for p in sorted(self.prims):                                             # the std::map of primitives, by index
            prim = self.prims[p]
            X, Y, Z = (u32(v) for v in self.cell(prim["p0"], s))                 # unsigned int X = static_cast<int>(...)
            B = u32(1 + 1000 * u32(u32(X * AABB_MAGIC_NUMBER) * AABB_MAGIC_NUMBER + Y * AABB_MAGIC_NUMBER + Z))
            if B not in self.level(0):
                self.level(0)[B] = Box(self.vd)                                  # :947-960
            if self.emissive[prim["material"]]:
                self.level(tree_depth).setdefault(0, Box(self.vd)).entries.append(p)   # :966-969
            else:
                self.level(0)[B].entries.append(p)                               # :977
        for key in sorted(self.level(0)):
            self.update_bounding_box(self.level(0)[key])                         # :985-987
        # outer levels, :1068-1076
        tree_depth, nb = 0, n
        while True:
            tree_depth += 1
            self.process_outer_boxes(nb, tree_depth)
            nb //= 4
            if not nb > 2:
                break
        self.tree_depth = tree_depth

    def process_outer_boxes(self, box_size, depth):                              # :994-1039
        s = self.steps(box_size)
        below = self.level(depth - 1)
        for key in sorted(below):
            X, Y, Z = self.cell(below[key].center, s)
            B = i32(i32(i32(X * box_size) * box_size) + i32(Y * box_size) + Z)   # int arithmetic: it wraps
            B = i32(B + 1)
            self.wrapped = self.wrapped or B < 0
            B = u32(B)                                                           # std::map<unsigned int, ...>
            box = self.level(depth).setdefault(B, Box(self.vd))
            box.lo, box.hi = [self.vd] * 3, [-self.vd] * 3
            box.entries.append(key)
        for key in sorted(self.level(depth)):
            self.update_outer_bounding_box(self.level(depth)[key], depth - 1)

    def stream(self):                                                            # :1151-1281
        self.nodes, self.order, self.lamps = [], [], []
        top = self.level(self.tree_depth)
        for rank, key in enumerate(sorted(top)):
            box = top[key]
            at = len(self.nodes)
            node = dict(lo=list(box.lo), hi=list(box.hi), nb=0, start=self.tree_depth, skip=0)
            self.nodes.append(node)
            if rank == 0:                                                        # :1181: begin() is the lamp box
                node["lo"], node["hi"] = [-self.vd] * 3, [self.vd] * 3
                node["nb"], node["start"] = len(box.entries), 0
                for p in box.entries:
                    self.order.append(p)
                    self.lamps.append(p)
            if self.tree_depth > 0:
                self.recurse(self.tree_depth - 1, box.entries)                   # :1252, for the lamp box too
            node["skip"] = len(self.nodes) - at

    def recurse(self, depth, keys):                                              # :1085-1149
        for key in keys:
            box = self.level(depth).setdefault(key, Box(0))                      # operator[] again
            if not box.entries:
                continue
            at = len(self.nodes)
            node = dict(lo=list(box.lo), hi=list(box.hi), nb=len(box.entries) if depth == 0 else 0,
                        start=len(self.order) if depth == 0 else depth, skip=1)
            self.nodes.append(node)
            if depth == 0:
                self.order.extend(box.entries)
            else:
                self.recurse(depth - 1, box.entries)
                node["skip"] = len(self.nodes) - at


# ---- scenes: (name, [(type, p0, p1, p2, size, emissive)]) ----------------------------------------------------
def _tiny(solr):
    S = solr.ptSphere
    return [(S, (-3000, 0, 0), 0, 0, (500, 0, 0), False), (S, (3000, 0, 0), 0, 0, (700, 0, 0), False),
            (S, (0, 2000, 1000), 0, 0, (300, 0, 0), False), (S, (8000, 8000, -8000), 0, 0, (10, 0, 0), True)]


def _mixed(solr):
    rng = np.random.default_rng(5)
    out = []
    for i in range(37):
        p0 = tuple(float(v) for v in rng.uniform(-9000, 9000, 3))
        if i % 5 == 0:
            p1 = tuple(p0[a] + float(rng.uniform(-900, 900)) for a in range(3))
            out.append((solr.ptCylinder, p0, p1, 0, (80, 0, 0), False))
        elif i % 5 == 1:
            p1 = tuple(p0[a] + float(rng.uniform(-900, 900)) for a in range(3))
            p2 = tuple(p0[a] + float(rng.uniform(-900, 900)) for a in range(3))
            out.append((solr.ptTriangle, p0, p1, p2, (0, 0, 0), False))
        elif i % 5 == 2:
            out.append((solr.ptXZPlane, p0, 0, 0, (2000, 0, 1500), False))
        else:
            out.append((solr.ptSphere, p0, 0, 0, (float(rng.uniform(50, 600)), 0, 0), i == 18))
    out.append((solr.ptSphere, (8000, 8000, -8000), 0, 0, (10, 0, 0), True))
    return out


def _spheres(n, seed, lights):
    def make(solr):
        rng = np.random.default_rng(seed)
        out = [(solr.ptSphere, tuple(float(v) for v in rng.uniform(-20000, 20000, 3)), 0, 0,
                (float(rng.uniform(20, 300)), 0, 0), False) for _ in range(n)]
        if n >= 1500:
            # the level-1 key X n^2 + ... passes 2^31 between X = 536 and 537 for n = 2001 (537 * 2001^2 > 2^31):
            # a row of spheres across those cells (a cell is 40000 / 2001 wide), four level-1 cells per level-2 cell
            cell = 40000.0 / (n + lights)
            for X in range(530, 545):
                out.append((solr.ptSphere, (-20000.0 + (X + 0.5) * cell, 100.0, 200.0), 0, 0, (5.0, 0, 0), False))
            out = out[15:]
        for k in range(lights):
            out.insert(int(rng.integers(0, len(out))),
                       (solr.ptSphere, tuple(float(v) for v in rng.uniform(-20000, 20000, 3)), 0, 0, (10, 0, 0), True))
        return out
    return make


SCENES = {
    "tiny: 3 spheres + a lamp, one outer level": _tiny,
    "mixed: 38 primitives of five types, two lamps, two outer levels": _mixed,
    "300 spheres + 3 lamps: four outer levels": _spheres(300, 7, 3),
    "2000 spheres + a lamp: outer keys wrap to negative": _spheres(2000, 9, 1),
}


def _build_both(solr, spec, view_distance=50000.0):
    k = solr.Kernel(engine="host-only")
    k.initialize(width=64, height=48, viewDistance=view_distance)
    plain = k.add_material(0.5, 0.5, 0.5)
    glow = k.add_material(1.0, 1.0, 1.0, innerIllumination=2.0)
    prims, emissive = {}, {plain: False, glow: True}
    for i, (t, p0, p1, p2, size, lamp) in enumerate(spec):
        p1 = p1 or (0, 0, 0)
        p2 = p2 or (0, 0, 0)
        idx = k.add_primitive(t, p0, p1, p2, size=size, material=glow if lamp else plain)
        assert idx == i
    k.compact_boxes(True)
    flat = k.flat_scene()
    # the model is fed what the host's setPrimitive stored (it derives a cylinder's size and centre, a sphere's
    # replicated radius ...): the records of the flattened scene, by original index
    for rec in flat.primitives:
        prims[int(rec["index"])] = dict(type=int(rec["type"]), p0=rec["p0"], p1=rec["p1"], p2=rec["p2"],
                                        size=rec["size"], material=int(rec["materialId"]))
    assert len(prims) == len(spec), "the builder lost primitives"
    model = ReferenceBuilder(solr, view_distance, prims, emissive)
    model.compact_boxes()
    model.stream()
    boxes = np.array(flat.boxes, copy=True)
    order = [int(v) for v in flat.primitives["index"]]
    lights = [int(v) for v in flat.lights["primitiveId"]]
    k.finalize()
    return model, boxes, order, lights


@pytest.mark.parametrize("name", list(SCENES))
def test_the_flattened_tree_is_the_reference_builders(solr, name):
    model, boxes, order, lights = _build_both(solr, SCENES[name](solr))
    assert len(boxes) == len(model.nodes), (len(boxes), len(model.nodes))
    assert order == model.order, "primitive order differs"
    assert lights == model.lamps, "light list differs"
    for i, (b, n) in enumerate(zip(boxes, model.nodes)):
        got = (tuple(float(v) for v in b["min"]), tuple(float(v) for v in b["max"]), int(b["nbPrimitives"]),
               int(b["startIndex"]), int(b["indexForNextBox"][0]))
        want = (tuple(float(v) for v in n["lo"]), tuple(float(v) for v in n["hi"]), n["nb"], n["start"], n["skip"])
        assert got == want, "node %d: builder %s, reference model %s" % (i, got, want)
    if "negative" in name:
        assert model.wrapped, "the scene was meant to wrap the outer keys"
        # ... and to have a parent whose children sit on both sides of the wrap: their order shows whether the
        # keys are compared as the unsigned values the reference's map holds
        mixed = [b for b in model.level(2).values()
                 if any(k >= 2 ** 31 for k in b.entries) and any(k < 2 ** 31 for k in b.entries)]
        assert mixed, "no level-2 box with children on both sides of the key wrap"
    if "four outer levels" in name:
        assert model.tree_depth == 4
    if "two outer levels" in name:
        assert model.tree_depth == 2


def test_tiny_scene_by_hand(solr):
    """Followed by hand through GPUKernel.cpp.  Extent: min (-3000, 0, -8000), max (8000, 8000, 1000) - the four
    centres and the origin the extent starts from.  Level 0: a cell is (11000, 8000, 9000) / 6400 =
    (1.71875, 1.25, 1.40625) wide.
      primitive 0, p0 = (-3000, 0, 0):   cell (0, 0, int(8000 / 1.40625) = 5688)  ->  key 1 + 1000 * 5688 = 5688001
      primitive 1, p0 = (3000, 0, 0):    cell (int(6000 / 1.71875) = 3490, 0, 5688)
                                         ->  key (1 + 1000 (3490 * 6400^2 + 5688)) mod 2^32 = 1009175233
      primitive 2, p0 = (0, 2000, 1000): cell (1745, 1600, 6400)  ->  key 10725377 (wrapped as well)
      primitive 3 (the lamp), p0 = (8000, 8000, -8000): cell (6400, 6400, 0), key 2976382977: the cell is created
           and stays empty (bounds left at the inside-out +-1e6 seed, centre 0); the primitive goes to box 0 of
           the top level.
    4 primitives: 4 > 2 -> depth 1, 4 / 4 = 1 stops: ONE outer level on a grid of 4^3 cells of (2750, 2000, 2250):
      cell of primitive 0, centre (-3000, 0, 0):  (0, 0, int(8000 / 2250) = 3)  ->  key 0 + 0 + 3 + 1 = 4
      the empty cell, centre (0, 0, 0):           (int(3000 / 2750) = 1, 0, 3)  ->  key 16 + 0 + 3 + 1 = 20
      cell of primitive 2, centre (0, 2000, 1000): (1, 1, int(9000 / 2250) = 4) ->  key 16 + 4 + 4 + 1 = 25
      cell of primitive 1, centre (3000, 0, 0):    (int(6000 / 2750) = 2, 0, 3) ->  key 32 + 0 + 3 + 1 = 36
    Flattened in key order 0, 4, 20, 25, 36:
      node 0  the lamp box: +-viewDistance, 1 primitive from index 0 (primitive 3); recursing into "key 3" of
              level 0 finds an empty box: skip 1
      node 1  box 4 = bounds of primitive 0's cell, inner node (start = depth 1), skip 2
      node 2     its cell: sphere 0, centre -3000 +- 500, one primitive from index 1
      node 3  box 20: only child is the empty cell, whose inside-out 1e6 seed never wins a comparison against
              the box's own inside-out +-viewDistance seed (:874-885): the node is emitted with min > max and no
              child - no ray enters it
      node 4  box 25, node 5 its cell: sphere 2 (radius 300), primitive index 2
      node 6  box 36, node 7 its cell: sphere 1 (radius 700), primitive index 3
    so the primitives are streamed in the order 3, 0, 2, 1."""
    model, boxes, order, lights = _build_both(solr, _tiny(solr))
    assert sorted(k for k, b in model.level(0).items() if b.entries or k > 3) == [5688001, 10725377, 1009175233,
                                                                                   2976382977]
    assert sorted(model.level(1)) == [0, 4, 20, 25, 36]
    assert order == [3, 0, 2, 1] and lights == [3]
    vd = 50000.0
    expected = [
        # min                    max                     nb start skip
        ((-vd, -vd, -vd), (vd, vd, vd), 1, 0, 1),
        ((-3500, -500, -500), (-2500, 500, 500), 0, 1, 2),
        ((-3500, -500, -500), (-2500, 500, 500), 1, 1, 1),
        ((vd, vd, vd), (-vd, -vd, -vd), 0, 1, 1),
        ((-300, 1700, 700), (300, 2300, 1300), 0, 1, 2),
        ((-300, 1700, 700), (300, 2300, 1300), 1, 2, 1),
        ((2300, -700, -700), (3700, 700, 700), 0, 1, 2),
        ((2300, -700, -700), (3700, 700, 700), 1, 3, 1),
    ]
    assert len(boxes) == len(expected)
    for b, (lo, hi, nb, start, skip) in zip(boxes, expected):
        assert tuple(float(v) for v in b["min"]) == lo and tuple(float(v) for v in b["max"]) == hi
        assert (int(b["nbPrimitives"]), int(b["startIndex"]), int(b["indexForNextBox"][0])) == (nb, start, skip)
