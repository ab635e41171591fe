"""Host side of the animated-scene protocol, no GPU: the scene store's lazy replay of rotations that an
engine applied to its resident scene (GPUKernel::syncHost) against the eager host route
(rotatePrimitives as the reference does it, GPUKernel.cpp:1378-1460), and the rules for when the device
route may be taken at all.  The "host-replay" engine is a test double that claims every rotation
(sol-r_amd/host/HipKernel.h ReplayKernel); the real engine's arithmetic is held to the same host route
in tests/test_animation_gpu.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenes_extra  # noqa: E402

STEPS = [((0.0, 0.0, 0.0), (0.0, 0.02, 0.0)), ((10.0, -20.0, 30.0), (0.11, -0.07, 0.05)),
         ((0.0, 0.0, 0.0), (0.0, 0.0, 0.3)), ((-500.0, 100.0, 0.0), (-0.2, 0.4, 0.0))]


def _same(a, b):
    return len(a) == len(b) and all(
        np.array_equal(np.ascontiguousarray(a[f]).view(np.uint8), np.ascontiguousarray(b[f]).view(np.uint8))
        for f in a.dtype.names)


def _build(solr, engine, scene):
    k = solr.Kernel(engine=engine, deterministic_seed=1)
    name, kw = scene
    (getattr(solr.scenes, name, None) or getattr(scenes_extra, name))(k, **kw)
    return k


def _frame(k):
    """one frame of the frame protocol; without a device it renders nothing and says so"""
    k.L.SolRx_Render(0.0)


SCENES = [("molecule", dict(atoms=300, width=32, height=32)), ("primitives_mix", dict(width=32, height=32)),
          ("height_field", dict(n=12, width=32, height=32))]


def test_lazy_replay_equals_the_eager_host_route(solr):
    for scene in SCENES:
        k = _build(solr, "host-replay", scene)
        _frame(k)
        for n, step in enumerate(STEPS):
            k.rotate_primitives(*step)
            assert k.pending_rotations() == n + 1
            _frame(k)
        lazy = k.flat_scene()            # reading the flattened arrays wakes the store ...
        assert k.pending_rotations() == 0
        k.rotate_primitives(*STEPS[0])   # ... without ending the fast path
        assert k.pending_rotations() == 1
        lazy2 = k.flat_scene()
        k.finalize()

        h = _build(solr, "host-only", scene)
        for step in STEPS:
            h.rotate_primitives(*step)
            assert h.pending_rotations() == 0
        eager = h.flat_scene()
        h.rotate_primitives(*STEPS[0])
        eager2 = h.flat_scene()
        h.finalize()
        assert _same(lazy.boxes, eager.boxes) and _same(lazy.primitives, eager.primitives), scene[0]
        assert _same(lazy2.boxes, eager2.boxes) and _same(lazy2.primitives, eager2.primitives), scene[0]
        assert _same(lazy.lights, eager.lights)


def test_device_route_needs_an_uploaded_untouched_scene(solr):
    k = _build(solr, "host-replay", SCENES[0])
    k.rotate_primitives(*STEPS[0])               # nothing uploaded yet
    assert k.pending_rotations() == 0
    _frame(k)
    k.rotate_primitives(*STEPS[1])
    assert k.pending_rotations() == 1
    assert k.L.SolR_GetPrimitiveMaterial(2) >= 0 # a look: replayed, fast path kept
    assert k.pending_rotations() == 0
    k.rotate_primitives(*STEPS[1])
    assert k.pending_rotations() == 1
    k.L.SolR_SetPrimitiveMaterial(2, 3)          # a change: replayed first, then applied, fast path off
    assert k.pending_rotations() == 0
    k.rotate_primitives(*STEPS[2])
    assert k.pending_rotations() == 0
    _frame(k)                                    # uploaded again
    k.rotate_primitives(*STEPS[3])
    assert k.pending_rotations() == 1
    k.compact_boxes(True)                        # a rebuild reads the primitives: replay first
    assert k.pending_rotations() == 0
    got = k.flat_scene()
    k.finalize()

    h = _build(solr, "host-only", SCENES[0])
    h.rotate_primitives(*STEPS[0])
    h.rotate_primitives(*STEPS[1])
    h.rotate_primitives(*STEPS[1])
    h.L.SolR_SetPrimitiveMaterial(2, 3)
    h.rotate_primitives(*STEPS[2])
    h.rotate_primitives(*STEPS[3])
    h.compact_boxes(True)
    want = h.flat_scene()
    h.finalize()
    assert _same(got.boxes, want.boxes) and _same(got.primitives, want.primitives)


def test_movable_flags_follow_the_flattened_order(solr):
    k = _build(solr, "host-only", SCENES[1])
    flat = k.flat_scene()
    ptr, n = solr.C.c_void_p(), solr.C.c_int()
    k.L.SolRx_GetMovable(solr.C.byref(ptr), solr.C.byref(n))
    flags = np.frombuffer((solr.C.c_char * n.value).from_address(ptr.value), np.uint8).copy()
    assert n.value == len(flat.primitives)
    # the lights sit in the first cell and stay put (GPUKernel.cpp:1189-1215), everything else here is movable
    assert flags[: flat.nb_lamps].sum() == 0 and flags[flat.nb_lamps:].all()
    k.finalize()
