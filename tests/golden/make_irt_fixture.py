"""tests/golden/model_subset.irt: every 10th triangle of the reference's sample scene file
medias/irt/test.irt (20 958 triangles, 16 materials, no textures; written by the reference's
FileMarshaller::saveToFile), re-wrapped in the same container: version, SceneInfo, count, records,
textures, materials, byte for byte the reference's.  Run here, where /root/reference exists:

    python tests/golden/make_irt_fixture.py
"""
import os
import struct

SRC = "/root/reference/medias/irt/test.irt"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model_subset.irt")
KEEP_EVERY = 10
RECORD = 160          # sizeof(CPUPrimitive) in the reference's 64-bit builds
HEADER = 8 + 112      # version + SceneInfo

d = open(SRC, "rb").read()
version, = struct.unpack("<Q", d[:8])
assert version == 2
n, = struct.unpack("<Q", d[HEADER:HEADER + 8])
body = HEADER + 8
tail = d[body + RECORD * n:]
records = [d[body + RECORD * i: body + RECORD * (i + 1)] for i in range(0, n, KEEP_EVERY)]
with open(DST, "wb") as f:
    f.write(d[:HEADER])
    f.write(struct.pack("<Q", len(records)))
    f.write(b"".join(records))
    f.write(tail)
print("%s: %d of %d primitives, %d bytes" % (DST, len(records), n, os.path.getsize(DST)))
