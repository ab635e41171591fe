"""Development aid: what would STREAMED secondary rays buy the node loop?  (bounce / shadow rays
written to a queue, sorted by direction octant and origin cell, walked by a kernel of their own at twice the occupancy.)

Measured on the walk replay (solr_hip_walk_bound: one frame's walks recorded per wave - list, and per lane ray, cut-off,
when a shadow lane was done - and replayed with nothing but the hand-scheduled node loop), before any of the pipeline is
built: the recorded rays are taken to the host, regrouped there and replayed

  * as recorded                                   (the megakernel's grouping: a wave = an 8 x 8 pixel tile)
  * one class of walks at a time                  (primary / bounce / shadow: where the node loop's time is)
  * compacted: the class's live rays packed 64 to a wave in their recorded order (what dropping dead lanes buys)
  * sorted: ... after a sort by (list, direction octant, Morton code of the origin's cell in a 32^3 grid)

each at the renderer's occupancy (the recorded launch's dynamic LDS: 4 waves per SIMD) and with no LDS at all (the
replay kernel holds 64 vector registers: 8 waves per SIMD - what a walk-only kernel could have).  A regrouped shadow ray
cannot keep "done after the wave's n-th leaf visit" (that counts the recorded wave's visits), so the shadow classes are
shown both as recorded and with every lane walking to the end of its list, and regrouping is compared on the latter.

    python tools/ray_regroup.py [height_field|molecule|cornell] [repeats]
"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
solr = importlib.import_module("sol-r_amd")
hip = solr.hip_lib()
scene = sys.argv[1] if len(sys.argv) > 1 else "height_field"
repeats = int(sys.argv[2]) if len(sys.argv) > 2 else 10
W, H = 1920, 1080
SLOTS = 16
HEAD = 16 * (SLOTS + 1)
SLOT_BYTES = HEAD + SLOTS * 64 * 32
NEVER = 0x7fffffff
CLOSEST, SHADOW = 0, 1

k = solr.Kernel(engine="hip")
kw = dict(width=W, height=H)
if scene == "cornell":
    kw["iterations"] = 3
getattr(solr.scenes, scene)(k, **kw)
hip.solr_hip_set_tile_scheduling(0)            # raster order: workgroup b is tile b
for _ in range(6):
    k.render()
flat = k.flat_scene()
si, ppi, eye, direction, angles = k.frame_parameters()
si.pathTracingIteration = 0
objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

hip.solr_hip_walk_records_keep(1)
ms, stats = (C.c_double * 3)(), (C.c_ulonglong * 4)()
k.check(hip.solr_hip_walk_bound(C.byref(si), C.byref(objects), C.byref(ppi), fp(eye), fp(direction), fp(angles), repeats,
                                ms, stats), "solr_hip_walk_bound")
info = (C.c_ulonglong * 4)()
k.check(hip.solr_hip_walk_records_info(info), "solr_hip_walk_records_info")
grid, slot_bytes, lds_recorded, slots = (int(v) for v in info)
assert slot_bytes == SLOT_BYTES and slots == SLOTS, (slot_bytes, slots)
print("# tools/ray_regroup.py %s: %d workgroups recorded, %d walks, dynamic LDS of the recorded launch %d B a wave; node "
      "loop alone as recorded %.4f ms (fastest %.4f)" % (scene, grid, stats[0], lds_recorded, ms[1], ms[2]))

raw = np.zeros((grid, SLOT_BYTES), np.uint8)
k.check(hip.solr_hip_walk_records_copy(C.c_void_p(raw.ctypes.data), grid, 0), "solr_hip_walk_records_copy")
heads = raw[:, :HEAD].copy().view(np.int32).reshape(grid, SLOTS + 1, 4)
lanes = raw[:, HEAD:].copy().view(np.float32).reshape(grid, SLOTS, 64, 2, 4)
del raw
nwalks = np.minimum(heads[:, 0, 0], SLOTS)
j = np.arange(SLOTS)[None, :]
valid = j < nwalks[:, None]                                   # (grid, SLOTS)
kind = heads[:, 1:, 0]
free = heads[:, 1:, 1]
tight = heads[:, 1:, 3] & 1
done = lanes[..., 1, 3].view(np.int32)                       # (grid, SLOTS, 64)
took_part = done >= 0
classes = {
    "primary": valid & (kind == CLOSEST) & (j == 0),
    "bounce": valid & (kind == CLOSEST) & (j > 0),
    "shadow": valid & (kind == SHADOW),
}


def replay(buf, lds):
    n = buf.shape[0]
    k.check(hip.solr_hip_walk_records_copy(C.c_void_p(buf.ctypes.data), n, 1), "records to the device")
    m, s = (C.c_double * 3)(), (C.c_ulonglong * 4)()
    k.check(hip.solr_hip_walk_replay(n, lds, repeats, m, s), "solr_hip_walk_replay")
    return float(m[1]), float(m[2]), int(s[2])


def one_walk_per_group(head_rows, lane_rows):
    """records with one walk per workgroup: head_rows (n, 4) int32, lane_rows (n, 64, 2, 4) float32"""
    n = head_rows.shape[0]
    buf = np.zeros((n, SLOT_BYTES), np.uint8)
    h = buf[:, :HEAD].view(np.int32).reshape(n, SLOTS + 1, 4)
    h[:, 0, 0] = 1
    h[:, 1, :] = head_rows
    buf[:, HEAD:HEAD + 64 * 32] = lane_rows.reshape(n, 64 * 32 // 4).view(np.uint8).reshape(n, 64 * 32)
    return buf


def morton(cells):
    code = np.zeros(cells.shape[0], np.int64)
    for bit in range(5):
        for axis in range(3):
            code |= ((cells[:, axis] >> bit) & 1) << (3 * bit + axis)
    return code


def report(name, buf, note=""):
    rec = replay(buf, -1)
    full = replay(buf, 0)
    print("  %-44s %7d waves   %8.4f ms at the renderer's occupancy   %8.4f ms with no LDS (8 waves a SIMD)   "
          "%10d leaf entries  %s" % (name, buf.shape[0], rec[0], full[0], full[2], note))
    return rec[0], full[0]


# the frame as recorded, at both occupancies
whole = np.zeros((grid, SLOT_BYTES), np.uint8)
whole[:, :HEAD] = heads.reshape(grid, -1).view(np.uint8)
whole[:, HEAD:] = lanes.reshape(grid, -1).view(np.uint8)
report("every walk, as recorded", whole)
# which walks took the node loop without its six min / max (the copies of the order-free lists with sorted bounds, rt_device.h
# SOLR_ORDER_SORTED / _REVERSED: bits 1-2 of a record's fourth word)
form = (heads[:, 1:, 3] >> 1) & 3
print("  of %d walks %d took the sorted form of the node loop, %d the reversed one" % (
    int(valid.sum()), int((valid & (form == 1)).sum()), int((valid & (form == 2)).sum())))
del whole

for name, mask in classes.items():
    b, s = np.nonzero(mask)
    if len(b) == 0:
        print("  %s: no such walks in this frame" % name)
        continue
    head_rows = heads[b, 1 + s, :].copy()
    lane_rows = lanes[b, s].copy()                           # (n, 64, 2, 4)
    part = took_part[b, s]
    live = int(part.sum())
    print("%s walks: %d waves, %d rays, %.1f of 64 lanes alive" % (name, len(b), live, live / float(len(b))))
    report(name + ", as recorded", one_walk_per_group(head_rows, lane_rows))
    if name == "primary":
        continue
    if name == "shadow":
        never = lane_rows.copy()
        d = never[:, :, 1, 3].view(np.int32)
        d[d >= 0] = NEVER
        report(name + ", every lane to the end of its list", one_walk_per_group(head_rows, never))
        lane_rows = never
    # the live rays of the class, one row each, with the list their wave walked
    rays = lane_rows[part]                                   # (live, 2, 4)
    wave_of = np.repeat(np.arange(len(b)), 64).reshape(len(b), 64)[part]
    lists = head_rows[wave_of]                               # (live, 4): kind, free list?, octant, thin copy?
    dirs = rays[:, 1, :3]
    octant = (dirs[:, 0] < 0).astype(np.int64) | ((dirs[:, 1] < 0).astype(np.int64) << 1) | ((dirs[:, 2] < 0).astype(np.int64) << 2)
    origins = rays[:, 0, :3].astype(np.float64)
    lo, hi = origins.min(axis=0), origins.max(axis=0)
    cells = np.clip(((origins - lo) / np.maximum(hi - lo, 1e-9) * 32.0).astype(np.int64), 0, 31)
    list_key = (lists[:, 1].astype(np.int64) << 1) | (lists[:, 3].astype(np.int64) & 1)

    def pack(order, label):
        r = rays[order]
        lk = list_key[order]
        oc = octant[order]
        # a wave holds rays of one list (and, sorted, of one octant): start a new wave where the key changes
        key = lk * 8 + (oc if label == "sorted" else 0)
        change = np.nonzero(np.diff(key))[0] + 1
        starts = np.concatenate(([0], change))
        ends = np.concatenate((change, [len(key)]))
        waves_h, waves_l = [], []
        for a, e in zip(starts, ends):
            n = e - a
            nw = (n + 63) // 64
            block = np.zeros((nw * 64, 2, 4), np.float32)
            block[:, 1, :3] = 1.0
            block[:, 1, 3].view(np.int32)[:] = -1
            block[:n] = r[a:e]
            waves_l.append(block.reshape(nw, 64, 2, 4))
            hrow = np.zeros((nw, 4), np.int32)
            hrow[:, 0] = lists[order[a], 0]
            hrow[:, 1] = lists[order[a], 1]
            hrow[:, 3] = lists[order[a], 3] & 1      # (the generic form of the node loop: regrouped rays share no octant)
            # the octant list of each wave's first ray (any of the eight gives the same result)
            hrow[:, 2] = oc[a:e][::64][:nw]
            waves_h.append(hrow)
        return one_walk_per_group(np.concatenate(waves_h), np.concatenate(waves_l))

    recorded_order = np.argsort(list_key, kind="stable")
    compacted = pack(recorded_order, "compacted")
    report(name + ", compacted (recorded order)", compacted)
    sorted_order = np.lexsort((morton(cells), octant, list_key))
    regrouped = pack(sorted_order, "sorted")
    report(name + ", sorted by octant and origin cell", regrouped)
    if len(b) < 16384:
        # fewer waves than the chip holds at once: alone, the class takes as long as its longest wave.  Its THROUGHPUT
        # cost - what it adds to a frame that keeps the chip full - from copies of it side by side
        copies = max(2, 32768 // len(b))
        as_recorded = one_walk_per_group(head_rows, lane_rows)
        for label, buf in ((", as recorded", as_recorded), (", compacted", compacted), (", sorted", regrouped)):
            rec, full = report("%s x %d%s" % (name, copies, label), np.tile(buf, (copies, 1)))
            print("      -> %.4f / %.4f ms per copy" % (rec / copies, full / copies))

hip.solr_hip_walk_records_release()
hip.solr_hip_walk_records_keep(0)
k.finalize()
