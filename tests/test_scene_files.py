"""Scene files: the .irt container (reference: solr/io/FileMarshaller.cpp; sol-r_amd/host/FileMarshaller.*).

Golden vector: the reference's own sample medias/irt/test.irt - in full where /root/reference exists
(here), and as tests/golden/model_subset.irt (every 10th triangle, made by
tests/golden/make_irt_fixture.py) everywhere.  The expectations below are derived from the file's bytes
with numpy, independently of the loader."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import assert_parity, compare_frames, gpu_frame, oracle_frame  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SUBSET = os.path.join(HERE, "golden", "model_subset.irt")
FULL = "/root/reference/medias/irt/test.irt"

f4, i4 = np.float32, np.int32
RECORD = np.dtype({"names": ["belongs", "movable", "p0", "p1", "p2", "n0", "n1", "n2", "size", "type", "materialId",
                             "vt0", "vt1", "vt2"],
                   "formats": ["u1", "u1"] + [(f4, 3)] * 7 + [i4, i4] + [(f4, 2)] * 3,
                   "offsets": [0, 1, 4, 16, 28, 40, 52, 64, 76, 88, 92, 96, 104, 112], "itemsize": 160})


def parse(path):
    d = open(path, "rb").read()
    version, = struct.unpack("<Q", d[:8])
    n, = struct.unpack("<Q", d[120:128])
    records = np.frombuffer(d, RECORD, count=n, offset=128)
    at = 128 + 160 * n
    nb_textures, = struct.unpack("<Q", d[at:at + 8])
    assert nb_textures == 0
    nb_materials, = struct.unpack("<Q", d[at + 8:at + 16])
    at += 16
    materials = {}
    for _ in range(nb_materials):
        mid, = struct.unpack("<Q", d[at:at + 8])
        materials[mid] = d[at + 8:at + 8 + 176]
        at += 184
    assert at == len(d)
    return version, records, materials


def expected_geometry(records, scale):
    """loadFromFile: setPrimitive(center + p), then every primitive times scale / height, all binary32"""
    pts = np.stack([records["p0"], records["p1"], records["p2"]])
    height = np.float32(abs(pts[..., 1].max() - pts[..., 1].min()))
    ratio = np.float32(scale) / height
    zero = np.float32(0.0)
    return [((zero + records[f]) * ratio).astype(f4) for f in ("p0", "p1", "p2")], ratio


def load(solr, path, engine="host-only", scale=5000.0, **kw):
    k = solr.Kernel(engine=engine, deterministic_seed=1)
    solr.scenes.irt_model(k, path, scale=scale, **kw)
    return k


FILES = [SUBSET] + ([FULL] if os.path.exists(FULL) else [])


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p) for p in FILES])
def test_irt_loader_against_the_files_bytes(solr, path):
    version, records, materials = parse(path)
    assert version == 2 and (records["type"] == solr.ptTriangle).all()
    k = load(solr, path, width=64, height=48)
    flat = k.flat_scene()
    prims = flat.primitives
    model = prims[prims["index"] < len(records)]
    assert len(model) == len(records) and len(prims) == len(records) + 2       # + floor + light
    order = np.argsort(model["index"])
    model = model[order]
    (p0, p1, p2), ratio = expected_geometry(records, 5000.0)
    for got, want in ((model["p0"], p0), (model["p1"], p1), (model["p2"], p2)):
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(model["materialId"], records["materialId"])
    assert np.array_equal(model["vt0"], records["vt0"]) and np.array_equal(model["vt2"], records["vt2"])
    # setPrimitiveNormals normalises what the file holds
    n0 = records["n0"] / np.maximum(np.linalg.norm(records["n0"], axis=1, keepdims=True), 1e-30)
    assert np.abs(model["n0"] - n0).max() < 1e-6
    # materials: the file's ids overwritten with the file's bytes (no textures in this file: no id shift)
    for mid, raw in materials.items():
        assert flat.materials[mid].tobytes()[:168] == raw[:168], mid   # 8 bytes of padding follow
    k.finalize()


def test_irt_save_and_load_round_trip(solr, tmp_path):
    _, records, materials = parse(SUBSET)
    k = load(solr, SUBSET, width=64, height=48)
    out = str(tmp_path / "saved.irt")
    assert k.save_to_file(out) == len(records) + 2
    version, saved, saved_materials = parse(out)
    assert version == 2 and len(saved) == len(records)       # the floor and the light are not of the model
    (p0, p1, p2), _ = expected_geometry(records, 5000.0)
    assert np.array_equal(saved["p0"].view(np.uint32), p0.view(np.uint32))
    assert np.array_equal(saved["p2"].view(np.uint32), p2.view(np.uint32))
    assert (saved["belongs"] == 1).all() and (saved["movable"] == 0).all()
    assert set(saved_materials) == set(int(m) for m in np.unique(records["materialId"]))
    for mid, raw in saved_materials.items():
        assert raw[:168] == materials[mid][:168]
    first = k.flat_scene()
    k.finalize()
    # a second generation, loaded at the height the first one has: ratio exactly 1
    height = float(np.float32(abs(max(saved["p0"][:, 1].max(), saved["p1"][:, 1].max(), saved["p2"][:, 1].max()) -
                                  min(saved["p0"][:, 1].min(), saved["p1"][:, 1].min(), saved["p2"][:, 1].min()))))
    k2 = load(solr, out, scale=height, width=64, height=48)
    second = k2.flat_scene()
    a = first.primitives[first.primitives["index"] < len(records)]
    b = second.primitives[second.primitives["index"] < len(records)]
    a, b = a[np.argsort(a["index"])], b[np.argsort(b["index"])]
    for f in ("p0", "p1", "p2", "materialId", "vt0", "vt1", "vt2"):
        assert np.array_equal(np.ascontiguousarray(a[f]).view(np.uint8), np.ascontiguousarray(b[f]).view(np.uint8)), f
    for f in ("n0", "n1", "n2"):     # normalised once more on the way in: the last bit may move
        assert np.abs(a[f] - b[f]).max() < 2e-7, f
    k2.finalize()


def test_irt_loader_refuses_other_versions_and_survives_truncation(solr, tmp_path):
    d = bytearray(open(SUBSET, "rb").read())
    other = str(tmp_path / "v1.irt")
    open(other, "wb").write(struct.pack("<Q", 1) + bytes(d[8:]))     # the OpenCL engine's format
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    solr.scenes.cornell(k, width=32, height=32)
    before = k.flat_scene().primitives
    assert k.load_from_file(other, 1000.0) == len(before)
    k.compact_boxes(True)
    after = k.flat_scene().primitives
    assert len(after) == len(before) and np.array_equal(after["p0"], before["p0"])          # untouched
    cut = str(tmp_path / "cut.irt")
    open(cut, "wb").write(bytes(d[:128 + 160 * 10 + 77]))
    k.load_from_file(cut, 1000.0)
    k.compact_boxes(True)
    ids = np.unique(k.flat_scene().primitives["index"])
    assert set(range(len(before) + 10)) <= set(ids.tolist()) and len(before) + 10 not in ids   # the ten whole records
    k.finalize()


@pytest.mark.gpu
def test_irt_model_renders_like_the_oracle(solr, oracle):
    k = load(solr, SUBSET, engine="hip", width=160, height=120, iterations=3)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert status == 0
    model_pixels = (ids[..., 0] >= 0) & (ids[..., 0] < 2096)
    assert model_pixels.mean() > 0.05, "the model is not in view"
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    k.finalize()


# ---- Wavefront OBJ / MTL (reference: solr/io/OBJReader.cpp; sol-r_amd/host/OBJReader.*) --------------
# Golden vector: the reference's medias/obj/cornell.obj + cornell.mtl (the model its Cornell-box scene
# loads), kept as data under tests/golden/.  Expectations come from an independent reading of the file.
OBJ = os.path.join(HERE, "golden", "cornell.obj")


def parse_obj(path):
    v, vt, vn, faces, material = [None], [None], [None], [], None
    for line in open(path):
        w = line.split()
        if not w:
            continue
        if w[0] == "v":
            v.append((f4(w[1]), f4(w[2]), -f4(w[3])))
        elif w[0] == "vn":
            vn.append((f4(w[1]), f4(w[2]), -f4(w[3])))
        elif w[0] == "vt":
            t = [f4(w[1]), f4(w[2])]
            t = [f4(abs(c) - int(abs(c))) if c < 0 else c for c in t]
            vt.append(tuple(t))
        elif w[0] == "usemtl":
            material = w[1]
        elif w[0] == "f":
            corners = [tuple(int(x) if x else 0 for x in (c.split("/") + ["", ""])[:3]) for c in w[1:]]
            faces.append((corners[:3], material))
            if len(corners) == 4:
                faces.append(([corners[3], corners[2], corners[0]], material))
    return v, vt, vn, faces


def parse_mtl(path):
    order, kd = [], {}
    for line in open(path):
        w = line.split()
        if w and w[0] == "newmtl":
            order.append(w[1])
        if w and w[0] == "Kd":
            kd[order[-1]] = tuple(f4(x) for x in w[1:4])
    return order, kd


def test_obj_reader_against_the_files_text(solr):
    v, vt, vn, faces = parse_obj(OBJ)
    order, kd = parse_mtl(OBJ[:-4] + ".mtl")
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    ground = solr.scenes.obj_model(k, OBJ, width=64, height=48, scale=5000.0)
    flat = k.flat_scene()
    prims = flat.primitives
    model = prims[prims["index"] < len(faces)]
    model = model[np.argsort(model["index"])]
    assert len(model) == len(faces) == 38 and (model["type"] == solr.ptTriangle).all()

    pts = np.array(v[1:], f4)
    lo, hi = pts.min(0), pts.max(0)
    os_ = max(hi[0] - lo[0], max(hi[1] - lo[1], hi[2] - lo[2]))
    s = f4(5000.0) / f4(os_)
    centre = ((lo + hi) / f4(2.0)).astype(f4)
    assert abs(ground - float(-(s * (hi[1] - lo[1])) / f4(2.0))) < 1e-3

    def place(p):                                       # position + scale * (-centre + p), binary32
        return (f4(0.0) + s * (-centre + np.array(p, f4))).astype(f4)

    ids = {name: n for n, name in enumerate(order)}     # material ids in MTL order from materialId = 0
    for n, (corners, material) in enumerate(faces):
        for field, (vi, ti, ni) in zip(("p0", "p1", "p2"), corners):
            assert np.array_equal(model[field][n].view(np.uint32), place(v[vi]).view(np.uint32)), (n, field)
        for field, (vi, ti, ni) in zip(("vt0", "vt1", "vt2"), corners):
            assert tuple(model[field][n]) == tuple(vt[ti]), (n, field)
        for field, (vi, ti, ni) in zip(("n0", "n1", "n2"), corners):
            want = np.array(vn[ni], f4)
            assert np.abs(model[field][n] - want / np.linalg.norm(want)).max() < 1e-6, (n, field)
        assert model["materialId"][n] == ids[material]
    for name, n in ids.items():
        assert tuple(flat.materials["color"][n][:3]) == kd[name]
        # Ks 0.33 0.33 0.33 -> specular value, 100 x power, coefficient (OBJReader.cpp:188-190)
        assert np.allclose(flat.materials["specular"][n], (0.33, 33.0, 0.0, 0.33))
    k.finalize()


def test_obj_reader_quads_missing_indices_and_absent_files(solr, tmp_path):
    path = str(tmp_path / "quad.obj")
    open(path, "w").write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt -1.25 0.5\nf 1/1 2 3 4\nf 1 2 9 5 6\n")
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    solr.scenes.cornell(k, width=32, height=32)
    before = len(k.flat_scene().primitives)
    k.load_obj_model(path, material_id=3, auto_scale=False, scale=10.0, auto_center=False)
    k.load_obj_model(str(tmp_path / "absent.obj"), material_id=3)      # nothing is added, as in the reference
    k.compact_boxes(True)
    prims = k.flat_scene().primitives
    new = prims[(prims["index"] >= before) & (prims["index"] < before + 3)]
    new = new[np.argsort(new["index"])]
    assert len(new) == 3 and before + 3 not in prims["index"]
    # the quad: corners 0 1 2, then 3 2 0 (OBJReader.cpp:735-752); z is negated, scale 10, no centring
    assert new["p0"][0].tolist() == [0.0, 0.0, 0.0] and new["p1"][0].tolist() == [10.0, 0.0, 0.0]
    assert new["p2"][0].tolist() == [10.0, 10.0, 0.0]
    assert new["p0"][1].tolist() == [0.0, 10.0, 0.0] and new["p1"][1].tolist() == [10.0, 10.0, 0.0]
    assert new["p2"][1].tolist() == [0.0, 0.0, 0.0]
    assert new["vt0"][0].tolist() == [0.25, 0.5]          # -1.25 -> the fraction of its magnitude
    assert new["vt1"][0].tolist() == [0.0, 0.0]           # no texture coordinate given: number 0, zeros
    # five corners: the first three only; vertex 9 does not exist and reads as the origin
    assert new["p2"][2].tolist() == [0.0, 0.0, 0.0] and new["p1"][2].tolist() == [10.0, 0.0, 0.0]
    assert (new["materialId"] == 3).all()
    k.finalize()


@pytest.mark.gpu
def test_obj_model_renders_like_the_oracle(solr, oracle):
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    solr.scenes.obj_model(k, OBJ, width=160, height=120, iterations=3)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert status == 0
    assert ((ids[..., 0] >= 0) & (ids[..., 0] < 38)).mean() > 0.5, "the model is not in view"
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    k.finalize()


# ---- SWC neuron morphologies (reference: solr/io/SWCReader.cpp; sol-r_amd/host/SWCReader.*) -----------
# Golden vector: the reference's medias/swc/02a_pyramidal2aFI.CNG.swc, kept as data (pyramidal.swc).
SWC = os.path.join(HERE, "golden", "pyramidal.swc")


def read_swc_like_the_reference(path):
    """drop a line, take seven blank-separated words (across line ends), repeat while the stream is good"""
    text = open(path, "rb").read().decode("latin-1")
    pos, good, records = 0, True, []
    while good:
        nl = text.find("\n", pos)
        if nl < 0:
            good = False            # getline ran into the end
            pos = len(text)
        else:
            pos = nl + 1
        words = []
        for _ in range(7):
            while pos < len(text) and text[pos] in " \t\r\n\v\f":
                pos += 1
            if pos >= len(text):
                good = False
                words.append("")
                continue
            end = pos
            while end < len(text) and text[end] not in " \t\r\n\v\f":
                end += 1
            words.append(text[pos:end])
            pos = end
        if words[0] != "#":
            records.append(words)
    return records


def _atoi(s):
    import re
    m = re.match(r"\s*[+-]?\d+", s)
    return int(m.group(0)) if m else 0


def _atof(s):
    import re
    m = re.match(r"\s*[+-]?(\d+\.?\d*([eE][+-]?\d+)?|\.\d+([eE][+-]?\d+)?)", s)
    return float(m.group(0)) if m else 0.0


def test_swc_reader_against_the_files_text(solr):
    scale = 40.0
    samples = {}
    for w in read_swc_like_the_reference(SWC):
        samples[_atoi(w[0])] = dict(x=f4(scale * _atof(w[2])), y=f4(scale * _atof(w[3])), z=f4(scale * _atof(w[4])),
                                    r=f4(scale * _atof(w[5])), parent=_atoi(w[6]))
    nb_samples = len(samples)
    expected = []                                        # (type, p0, p1, size.x) in the order they are added
    for sid in sorted(samples):
        a = samples[sid]
        if a["parent"] == -1:
            expected.append((solr.ptSphere, (a["x"], a["y"], a["z"]), (0, 0, 0), f4(a["r"] * f4(1.5))))
            continue
        b = samples.setdefault(a["parent"], dict(x=f4(0), y=f4(0), z=f4(0), r=f4(0), parent=0))
        if b["parent"] == -1:
            continue
        expected.append((solr.ptCylinder, (a["x"], a["y"], a["z"]), (b["x"], b["y"], b["z"]), a["r"]))
        expected.append((solr.ptSphere, (b["x"], b["y"], b["z"]), (0, 0, 0), b["r"]))

    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    n = solr.scenes.swc_morphology(k, SWC, width=64, height=48, scale=scale)
    assert n == len(samples)         # the reader's map, parents that came into being included
    assert nb_samples in (n, n - 1)
    prims = k.flat_scene().primitives
    model = prims[prims["index"] < len(expected)]
    model = model[np.argsort(model["index"])]
    assert len(model) == len(expected) > 3000
    for got, (ptype, p0, p1, radius) in zip(model, expected):
        assert got["type"] == ptype
        assert tuple(got["p0"]) == tuple(f4(c) for c in p0)
        if ptype == solr.ptCylinder:
            assert tuple(got["p1"]) == tuple(f4(c) for c in p1)
        assert got["size"][0] == radius
    k.finalize()


@pytest.mark.gpu
def test_swc_morphology_renders_like_the_oracle(solr, oracle):
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    solr.scenes.swc_morphology(k, SWC, width=160, height=120, iterations=2)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert status == 0
    assert (ids[..., 0] >= 0).mean() > 0.02, "the neuron is not in view"
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    k.finalize()


# ---- PDB molecules (reference: solr/io/PDBReader.cpp; sol-r_amd/host/PDBReader.*) ---------------------
# Golden vector: the reference's medias/pdb/1BNA.pdb (a B-DNA dodecamer, 486 atoms), kept as data.  The
# expectations restate the reader with plain column slices and numpy binary32 arithmetic.
PDB = os.path.join(HERE, "golden", "1BNA.pdb")
ELEMENTS = os.path.join(os.path.dirname(HERE), "sol-r_amd", "host", "pdb_elements.txt")


def _element_tables():
    colours, radii = [], []
    for line in open(ELEMENTS):
        w = line.split()
        if w and w[0] == "colour":
            colours.append((w[1].upper(), int(w[2]), int(w[3]), int(w[4])))
        elif w and w[0] == "radius":
            radii.append((w[1], f4(w[2])))
    return colours, radii


def _squeeze(s):
    return s.replace(" ", "")


def parse_pdb(path):
    atoms = []
    for raw in open(path, newline=""):
        line = raw.rstrip("\n")
        if not line.startswith("ATOM"):
            continue
        assert len(line) >= 80
        atoms.append(dict(id=_atoi(_squeeze(line[7:11])), code=_squeeze(line[13:17]), chain=ord(line[21]) - 64,
                          residue=_atoi(_squeeze(line[23:26])), x=f4(_atof(_squeeze(line[31:37]))),
                          y=f4(_atof(_squeeze(line[39:45]))), z=f4(-f4(_atof(_squeeze(line[47:53])))),
                          element=_squeeze(line[77:79])))
    return atoms


def expected_molecule(solr, geometry_type, scale=200.0, atom_size=100.0, stick_size=10.0):
    colours, radii = _element_tables()
    atoms = parse_pdb(PDB)
    for n, a in enumerate(atoms):
        a["material"] = next(i for i, c in enumerate(colours) if c[0] == a["element"])
        a["w"] = next(r for name, r in radii if name == a["element"])
        a["backbone"] = len(a["code"]) == 1 or geometry_type in (4, 5)
        a["key"] = a["id"] if (geometry_type == 2 or (geometry_type == 3 and a["residue"] % 2 == 0)) else n + 1
    by_key = {}
    for a in atoms:
        by_key[a["key"]] = a                      # a later atom with the same key replaces the earlier one
    order = [by_key[k_] for k_ in sorted(by_key)]
    pos = np.array([[a["x"], a["y"], a["z"]] for a in atoms], f4)   # the extent is taken over every atom read
    mn, mx = pos.min(0), pos.max(0)
    centre = ((mn + mx) / f4(2)).astype(f4)
    s = (f4(scale) / (mx - mn)).astype(f4)
    spread = ((s * f4(2)) * f4(30)).astype(f4)

    def place(p):
        return (spread * (np.array(p, f4) - centre)).astype(f4)

    out = []
    for a in order:
        radius, stick = a["w"], a["w"]
        if geometry_type == 2:
            radius = stick = f4(stick_size)
        elif geometry_type == 3:
            radius, stick = f4(a["w"] / f4(2)), f4(f4(stick_size) / f4(2))
        if geometry_type in (2, 3):
            for b in order:
                if b is a or b["backbone"] != a["backbone"]:
                    continue
                d = np.array([a["x"] - b["x"], a["y"] - b["y"], a["z"] - b["z"]], f4)
                dist = np.sqrt(f4(f4(d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]))
                if dist < f4(1.7):
                    half = [f4((a[c] + b[c]) / f4(2)) for c in "xyz"]
                    out.append((solr.ptCylinder, place([a["x"], a["y"], a["z"]]), place(half), f4(s[0] * stick),
                                a["material"] if geometry_type == 2 else 1010))
        material = 11 if geometry_type == 3 else a["material"]
        out.append((solr.ptSphere, place([a["x"], a["y"], a["z"]]), None, f4(s[0] * stick), material))
    return out, colours


@pytest.mark.parametrize("geometry_type", [0, 2, 3], ids=["atoms", "sticks", "atoms-and-sticks"])
def test_pdb_reader_against_the_files_text(solr, geometry_type):
    expected, colours = expected_molecule(solr, geometry_type)
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    solr.scenes.pdb_molecule(k, PDB, width=64, height=48, geometry_type=geometry_type)
    flat = k.flat_scene()
    prims = flat.primitives
    model = prims[prims["index"] < len(expected)]
    model = model[np.argsort(model["index"])]
    assert len(model) == len(expected) and len(np.unique(prims["index"])) == len(expected) + 1     # + the light
    assert len(expected) == 486 if geometry_type == 0 else len(expected) > 486 + 800      # half-bonds on top
    for got, (ptype, p0, p1, size, material) in zip(model, expected):
        assert got["type"] == ptype and got["materialId"] == material
        assert np.array_equal(got["p0"].view(np.uint32), p0.view(np.uint32))
        if p1 is not None:
            assert np.array_equal(got["p1"].view(np.uint32), p1.view(np.uint32))
        assert got["size"][0] == size
    for i, (_, r, g, b) in enumerate(colours):          # materials 0.. take the element colours
        assert np.array_equal(flat.materials["color"][i][:3], np.array([r, g, b], f4) / f4(255))
    k.finalize()


@pytest.mark.gpu
def test_pdb_molecule_renders_like_the_oracle(solr, oracle):
    k = solr.Kernel(engine="hip", deterministic_seed=1)
    solr.scenes.pdb_molecule(k, PDB, width=160, height=120, iterations=3)
    pp, ids, rgb = gpu_frame(k)
    opp, oids, orgb, _, status = oracle_frame(k, oracle)
    assert status == 0
    assert (ids[..., 0] >= 0).mean() > 0.05, "the molecule is not in view"
    assert_parity(compare_frames(pp, ids, rgb, opp, oids, orgb))
    k.finalize()
