"""Generates tests/golden/reference_*.npz: frames rendered by THE REFERENCE'S OWN renderer.

Run on a GPU box where oracle/_ref is built (python -c 'import __graft_entry__ as g; g.build()' in the
container that has /root/reference, then gpurun):  python tests/golden/make_reference_fixtures.py

Each fixture holds the outputs of the reference's OpenCL k_standardRenderer + k_default
(solr/engines/opencl/RayTracer.cl compiled for gfx950, see oracle/Makefile) for one small scene built by
sol-r_amd/scenes.py - float framebuffer, primitive ids, RGB8 - plus a digest of the scene arrays it was
fed, so that tests/test_golden_reference.py can rebuild the same inputs on CPU and compare the oracle
with the reference without a GPU.  Only plane-free scenes: the OpenCL engine's plane shading reads
unwritten .w lanes and is not reproducible (tests/test_reference_opencl.py).  Data only - no reference
source in any form."""
import hashlib
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FIXTURES = {
    "spheres_1_bounce": ("cornell", dict(width=96, height=64, iterations=1, glass=0, room=False)),
    "spheres_3_bounces": ("cornell", dict(width=96, height=64, iterations=3, glass=0, room=False)),
    "triangle_mesh_2_bounces": ("height_field", dict(n=16, width=96, height=64)),
}
CAMERA = dict(eye=(131.0, 77.0, -15000.0), look_at=(57.0, 23.0, 0.0))   # off-axis, unrotated


def build(solr, spec):
    name, kw = spec
    k = solr.Kernel(engine="host-only", deterministic_seed=1)
    getattr(solr.scenes, name)(k, **kw)
    k.set_camera(CAMERA["eye"], look_at=CAMERA["look_at"])
    return k


def scene_digest(flat):
    h = hashlib.sha256()
    for a in (flat.boxes, flat.primitives, flat.lights):
        for field in a.dtype.names:
            h.update(np.ascontiguousarray(a[field]).tobytes())
    return h.hexdigest()


def main():
    solr = importlib.import_module("sol-r_amd")
    from oracle import loader
    assert loader.have_ref(), "oracle/_ref is not built"
    for name, spec in FIXTURES.items():
        k = build(solr, spec)
        flat = k.flat_scene()
        si, ppi, eye, direction, angles = k.frame_parameters()
        d = np.array(direction, np.float32).copy()
        d[0] -= np.float32(3.0)   # the OpenCL engine adds AArotatedGrid[0] = (3, 5) on pass 0 (RayTracer.cl:2526)
        d[1] -= np.float32(5.0)
        frames = [loader.ref_render(flat, si, ppi, eye, d, angles) for _ in range(2)]
        assert all(np.array_equal(a, b) for a, b in zip(*frames)), "reference output is not reproducible: " + name
        pp, ids, rgb = frames[0]
        np.savez_compressed(os.path.join(HERE, "reference_%s.npz" % name), colour=pp[..., :4].astype(np.float32),
                            ids=ids[..., 0].astype(np.int32), rgb=rgb, scene_digest=np.array(scene_digest(flat)))
        print(name, pp.shape, "written")


if __name__ == "__main__":
    main()
