"""GPUKernel::compactBoxes(true) on the device (sol-r_amd/csrc/solr_tree.hip, solr_hip_build_tree) against the
host builder: the flattened node list - bounds, primitive counts, start indices, skip pointers - and the order
in which the primitives are streamed must be the host builder's, bit for bit, which tests/
test_builder_reference_model.py in turn holds to a line-by-line model of the reference's builder.  Scenes: the
small ones of that test (key wraps, several lamps, five primitive types) and BASELINE's cfg2 (100 352
triangles) and cfg3 (50 000 atoms, 100k spheres and cylinders)."""
import ctypes as C
import importlib
import time

import numpy as np
import pytest

import test_builder_reference_model as M

pytestmark = pytest.mark.gpu


def _device_tree(solr, flat, view_distance):
    hip = solr.hip_lib()
    hip.solr_hip_build_tree_message.restype = C.c_char_p
    n = len(flat.primitives)
    by_index = np.zeros(n, flat.primitives.dtype)
    by_index[flat.primitives["index"]] = flat.primitives            # the scene's primitives in index order
    assert np.array_equal(np.sort(flat.primitives["index"]), np.arange(n))
    emissive = (flat.materials["innerIllumination"][by_index["materialId"], 0] != 0).astype(np.uint8)
    p0 = by_index["p0"]
    lo = np.minimum(p0.min(axis=0), 0).astype(np.float32)           # a fresh kernel's extent: primitives and the origin
    hi = np.maximum(p0.max(axis=0), 0).astype(np.float32)
    boxes = np.zeros(12 * n + 64, flat.boxes.dtype)
    order = np.zeros(n, np.int32)
    nb_boxes, nb_lamps = C.c_int(), C.c_int()
    t0 = time.perf_counter()
    depth = hip.solr_hip_build_tree(C.c_void_p(by_index.ctypes.data), C.c_void_p(emissive.ctypes.data), n,
                                    C.c_void_p(lo.ctypes.data), C.c_void_p(hi.ctypes.data), C.c_float(view_distance),
                                    C.c_void_p(boxes.ctypes.data), len(boxes), C.c_void_p(order.ctypes.data),
                                    C.byref(nb_boxes), C.byref(nb_lamps))
    seconds = time.perf_counter() - t0
    assert depth >= 1, (depth, hip.solr_hip_build_tree_message())
    return boxes[: nb_boxes.value], order, nb_lamps.value, depth, seconds


def _same_tree(boxes, order, nb_lamps, flat):
    assert len(boxes) == len(flat.boxes)
    for field in ("min", "max", "nbPrimitives", "startIndex"):
        assert np.array_equal(boxes[field], flat.boxes[field]), field
    assert np.array_equal(boxes["indexForNextBox"][:, 0], flat.boxes["indexForNextBox"][:, 0])
    assert np.array_equal(order, flat.primitives["index"])
    assert nb_lamps == flat.nb_lamps


@pytest.mark.parametrize("name", list(M.SCENES))
def test_small_scenes(solr, name):
    k = solr.Kernel(engine="hip")
    k.initialize(width=64, height=48, viewDistance=50000.0)
    plain = k.add_material(0.5, 0.5, 0.5)
    glow = k.add_material(1.0, 1.0, 1.0, innerIllumination=2.0)
    for t, p0, p1, p2, size, lamp in M.SCENES[name](solr):
        k.add_primitive(t, p0, p1 or (0, 0, 0), p2 or (0, 0, 0), size=size, material=glow if lamp else plain)
    k.L.SolRx_HostBuild(1)                      # the host builder's tree is the expectation
    k.compact_boxes(True)
    flat = k.flat_scene()
    boxes, order, nb_lamps, depth, _ = _device_tree(solr, flat, 50000.0)
    _same_tree(boxes, order, nb_lamps, flat)
    k.L.SolRx_HostBuild(0)
    k.finalize()


@pytest.mark.parametrize("scene,kw", [("height_field", dict(n=224)), ("molecule", dict(atoms=50000)),
                                      ("cornell", dict())], ids=["cfg2-mesh", "cfg3-molecule", "cfg1-cornell"])
def test_baseline_scenes(solr, scene, kw):
    k = solr.Kernel(engine="hip")
    k.L.SolRx_HostBuild(1)
    getattr(solr.scenes, scene)(k, width=64, height=48, **kw)
    flat = k.flat_scene()
    si = k.frame_parameters()[0]
    boxes, order, nb_lamps, depth, seconds = _device_tree(solr, flat, si.viewDistance)
    _same_tree(boxes, order, nb_lamps, flat)
    boxes, order, nb_lamps, depth, seconds = _device_tree(solr, flat, si.viewDistance)   # warm: allocations aside
    print("%s: %d primitives, %d nodes, depth %d, device build + copies %.1f ms" % (scene, len(order), len(boxes), depth,
                                                                                   seconds * 1e3))
    k.L.SolRx_HostBuild(0)
    k.finalize()
