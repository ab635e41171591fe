"""The order-free node lists built on the device (sol-r_amd/csrc/solr_lists.hip) against the host builder
(solr_scene.hip buildFreeOrderLists): the eight lists - bounds, primitive counts, skip pointers, the inner nodes that
were left out - must be the host's bit for bit, on BASELINE's scenes at their full size, on scenes whose leaves
coincide (twins: the split that halves by position, which only a stable partition on both sides makes the same),
and after the frames rendered from them equal the ones of the reference-order walks (variant 6)."""
import os

import numpy as np
import pytest

import scenes_extra as X

pytestmark = pytest.mark.gpu


def _lists(solr, build, **kw):
    k = solr.Kernel(engine="hip")
    build(k, **kw)
    image = k.render()                 # (tests build the lists with the first frame: conftest.py)
    image = k.render()
    k.check(0, "frames")
    n = solr.hip_lib().solr_hip_order_free_nodes()
    lists = [k.device_nodes(order_free=o).copy() for o in range(8)] if n else []
    lists.append(k.device_nodes(exact=False).copy())     # the walk-order list: its pruning decisions are made there too
    k.finalize()
    return n, lists, image


def _twins(k, width=96, height=64, **info):
    """spheres in identical pairs, each pair in two different grid cells' worth of distance apart so that they are
    leaves of their own - but some exactly on top of each other in separate leaves is what the reference's grid does
    not produce; coinciding CENTRES of different leaves come from equal boxes: cylinders and their end spheres"""
    k.initialize(width=width, height=height, nbRayIterations=2, **info)
    m = k.add_material(0.6, 0.5, 0.4, specValue=0.3, specPower=20.0)
    rng = solr_rng(3)
    for i in range(300):
        c = (rng.uniform(-6000, 6000), rng.uniform(-4000, 4000), rng.uniform(-3000, 6000))
        k.add_primitive(k_solr.ptSphere, c, size=(120.0, 0, 0), material=m)
        k.add_primitive(k_solr.ptSphere, c, size=(120.0, 0, 0), material=m)          # the same box again
        k.add_primitive(k_solr.ptCylinder, c, (c[0] + 1.0, c[1], c[2]), size=(119.5, 0, 0), material=m)
    X._light(k)
    k.compact_boxes(True)
    return k


k_solr = None


def solr_rng(seed):
    return k_solr.scenes.LCG(seed)


@pytest.mark.parametrize("scene,kw", [("cornell", dict(width=160, height=120, iterations=2)),
                                      ("height_field", dict(n=224, width=160, height=120)),
                                      ("molecule", dict(atoms=50000, width=160, height=120)),
                                      ("molecule", dict(atoms=3000, width=160, height=120)),
                                      ("twins", dict())],
                         ids=["cfg1-cornell", "cfg2-mesh", "cfg3-molecule", "small-molecule", "twins"])
def test_device_lists_are_the_hosts(solr, scene, kw):
    global k_solr
    k_solr = solr
    build = _twins if scene == "twins" else getattr(solr.scenes, scene)
    os.environ.pop("SOLR_HIP_LISTS_ON_HOST", None)
    n_dev, dev, image_dev = _lists(solr, build, **kw)
    os.environ["SOLR_HIP_LISTS_ON_HOST"] = "1"
    try:
        n_host, host, image_host = _lists(solr, build, **kw)
    finally:
        os.environ.pop("SOLR_HIP_LISTS_ON_HOST", None)
    assert n_dev == n_host and n_dev > 0, (n_dev, n_host)
    for o in range(9):
        a, b = dev[o].view(np.int32), host[o].view(np.int32)
        assert a.shape == b.shape
        if not np.array_equal(a, b):
            bad = np.argwhere((a != b).any(axis=(1, 2)))[:5].ravel()
            raise AssertionError("list %d (8 = the walk-order list): nodes %s differ, e.g. device %s host %s" % (o, bad, dev[o][bad[0]], host[o][bad[0]]))
    assert np.array_equal(image_dev, image_host)


def _sequence(solr, build, kw, after):
    """upload, `after` frames until the lists are built, the lists; a material more (the arena is laid out again),
    the lists; a rotation (the refit plan wants the host images of the lists), a frame"""
    saved = os.environ.get("SOLR_HIP_FREE_AFTER")
    os.environ["SOLR_HIP_FREE_AFTER"] = str(after)
    try:
        k = solr.Kernel(engine="hip")
        build(k, **kw)
        for _ in range(after + 1):
            k.render()
        k.check(0, "frames")
        n = solr.hip_lib().solr_hip_order_free_nodes()
        first = [k.device_nodes(order_free=o).copy() for o in range(8)]
        k.add_material(0.1, 0.9, 0.2, specValue=0.5, specPower=10.0)
        k.render()
        second = [k.device_nodes(order_free=o).copy() for o in range(8)]
        k.rotate_primitives((0.0, 0.0, 0.0), (0.05, 0.1, 0.0))
        image = k.render().copy()
        k.check(0, "rotated frame")
        third = [k.device_nodes(order_free=o).copy() for o in range(8)]
        k.finalize()
        return n, first, second, third, image
    finally:
        if saved is None:
            os.environ.pop("SOLR_HIP_FREE_AFTER", None)
        else:
            os.environ["SOLR_HIP_FREE_AFTER"] = saved


@pytest.mark.parametrize("scene,kw", [("cornell", dict(width=160, height=120, iterations=2)),
                                      ("molecule", dict(atoms=3000, width=160, height=120)),
                                      ("height_field", dict(n=64, width=160, height=120))],
                         ids=["cornell", "small-molecule", "small-mesh"])
@pytest.mark.parametrize("after", [1, 2], ids=["from-host-rows", "from-the-arena"])
def test_lists_that_stay_on_the_device(solr, scene, kw, after):
    """The device builder leaves its lists on the device (a device-to-device copy into the arena) and, once the scene
    has been rendered, reads the exact list from the arena too: the same lists as the host's, through a second layout
    of the arena (host images fetched from the arena) and a rotation on the device (refit plan from them)."""
    global k_solr
    k_solr = solr
    build = getattr(solr.scenes, scene)
    for name in ("SOLR_HIP_LISTS_ON_HOST", "SOLR_HIP_LISTS_VIA_HOST"):
        os.environ.pop(name, None)
    n_dev, first, second, third, image = _sequence(solr, build, kw, after)
    os.environ["SOLR_HIP_LISTS_ON_HOST"] = "1"
    try:
        n_host, first_h, second_h, third_h, image_h = _sequence(solr, build, kw, after)
    finally:
        os.environ.pop("SOLR_HIP_LISTS_ON_HOST", None)
    assert n_dev == n_host and n_dev > 0
    for name, a, b in (("as built", first, first_h), ("laid out again", second, second_h), ("rotated", third, third_h)):
        for o in range(8):
            assert np.array_equal(a[o].view(np.int32), b[o].view(np.int32)), (name, o)
    assert np.array_equal(image, image_h)


def test_scene_changes_do_not_leak_device_memory(solr):
    """upload, lists on the device (staged, appended, fetched back for a refit plan), rotation, finalize - five
    times: what the device has free afterwards stays where it was after the first round (the builders' pooled
    scratch, the staged lists and their origins, the arena that was re-allocated to take the lists)"""
    import ctypes as C
    runtime = C.CDLL("libamdhip64.so")

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert runtime.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value

    saved = os.environ.get("SOLR_HIP_FREE_AFTER")
    os.environ["SOLR_HIP_FREE_AFTER"] = "2"
    try:
        after = []
        for cycle in range(5):
            k = solr.Kernel(engine="hip")
            solr.scenes.molecule(k, atoms=6000 + 500 * cycle, width=96, height=64)
            for _ in range(3):
                k.render()
            assert solr.hip_lib().solr_hip_order_free_nodes() > 0
            k.rotate_primitives((0.0, 0.0, 0.0), (0.02, 0.03, 0.0))
            k.render()
            k.check(0, "cycle %d" % cycle)
            k.finalize()
            after.append(free_bytes())
    finally:
        if saved is None:
            os.environ.pop("SOLR_HIP_FREE_AFTER", None)
        else:
            os.environ["SOLR_HIP_FREE_AFTER"] = saved
    # (the second round settles what the process keeps between scenes - the runtime's own pools among it)
    assert min(after[2:]) >= after[1] - (16 << 20), [a >> 20 for a in after]
    assert after[1] >= after[0] - (256 << 20), [a >> 20 for a in after]
