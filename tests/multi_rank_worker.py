"""One rank of tests/test_multi_rank_gpu.py (its own process; no torch anywhere).

    python multi_rank_worker.py <rank> <world> <directory> <loopback | rccl>

The engine's multi-rank code through the C ABI alone, as INTEGRATION.md section 4 shows it to a C++ host: strips,
solr_hip_comm_init, solr_hip_gather_strips / _ids, solr_hip_balance_strips and the depth-halo exchange inside
cudaRender - with `world` ranks.  Transport "rccl": the real library, one GPU per rank (skipped on a one-GPU box).
Transport "loopback": tests/loopback_rccl.c, all ranks on GPU 0.  Either way rank 0 first renders every frame alone,
whole, and the assembled frames must equal those bit for bit.

The ranks are given DIFFERENT random buffers on purpose (a host seeds its own from the clock, GPUKernel.cpp:89): the
frames only assemble because solr_hip_comm_init makes rank 0's buffer everybody's and the halo height is agreed."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def put(path, data):
    with open(path + ".part", "wb") as f:
        f.write(data)
    os.rename(path + ".part", path)


def get(path, seconds=60.0):
    deadline = time.time() + seconds
    while not os.path.exists(path):
        if time.time() > deadline:
            raise RuntimeError("waited %.0f s for %s" % (seconds, path))
        time.sleep(0.005)
    with open(path, "rb") as f:
        return f.read()


def main():
    rank, world, directory, transport = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if transport == "loopback":
        assert os.environ.get("SOLR_HIP_RCCL_LIBRARY"), "the parent test names the stand-in library"
    device = rank if transport == "rccl" else 0
    solr = importlib.import_module("sol-r_amd")
    hip = solr.hip_lib()
    W, H = 192, 136
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    def barrier(name):
        put(os.path.join(directory, "at.%s.%d" % (name, rank)), b"1")
        for r in range(world):
            get(os.path.join(directory, "at.%s.%d" % (name, r)))

    # ---- the scene, the same on every rank but for the random buffer
    k = solr.Kernel(engine="hip", device=device, deterministic_seed=1000 + 17 * rank)
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    k.render()
    k.check(0, "first frame")
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    plain = solr.PostProcessingInfo(solr.ppe_none, 0.0, 0.0, 0)
    # taps up to 16 * 1700 * 0.005 / 10 = 13.6 pixels away: rows of the neighbouring strips (whose randoms differ
    # in their extremes from rank to rank: the exact reach is what has to be agreed)
    occlusion = solr.PostProcessingInfo(solr.ppe_ambientOcclusion, 0.0, 1700.0, 0)

    def render(pp=plain, dx=0.0):
        e = eye.copy()
        e[0] += dx
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(pp), fp(e), fp(direction), fp(angles))

    def whole_frame(pp=plain, dx=0.0, ids=False):
        render(pp, dx)
        rgb = np.zeros((H, W, 3), np.uint8)
        pid = np.zeros((H, W, 4), np.int32)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(rgb.ctypes.data), C.c_void_p(pid.ctypes.data) if ids else None)
        k.check(0, "reference frame")
        return (rgb, pid) if ids else rgb

    def gathered():
        image = np.zeros((H, W, 3), np.uint8)
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        return image

    def my_strip():
        first, count = C.c_int(), C.c_int()
        hip.solr_hip_get_strip(C.byref(first), C.byref(count))
        return first.value, count.value

    # rank 0's references: every frame of the scenario, whole, on one GPU, BEFORE there is a communicator
    if rank == 0:
        ref_plain, ref_ids = whole_frame(ids=True)
        ref_moved = [whole_frame(dx=100.0 * i) for i in range(9)]
        ref_occlusion = whole_frame(occlusion)
        ref_occlusion_narrow = whole_frame(solr.PostProcessingInfo(solr.ppe_ambientOcclusion, 0.0, 700.0, 0))

    # ---- the communicator: rank 0's id travels through a file ("whatever channel the host has")
    uid = C.create_string_buffer(128)
    if rank == 0:
        assert hip.solr_hip_comm_unique_id(uid) == 0
        put(os.path.join(directory, "uid.bin"), uid.raw)
    else:
        uid = C.create_string_buffer(get(os.path.join(directory, "uid.bin")), 128)
    assert hip.solr_hip_comm_init(rank, world, uid) == 0
    k.check(0, "solr_hip_comm_init")
    assert hip.solr_hip_comm_ranks() == world
    shared_seed = int(hip.solr_hip_comm_shared_seed())
    assert (shared_seed != 0) == (world > 1)
    first, count, _ = solr.strip_rows(rank, world, H)
    hip.solr_hip_set_strip(first, count)
    report = {"rank": rank, "equal_strip": [first, count], "shared_seed": shared_seed,
              "communicators": int(hip.solr_hip_comm_count())}

    # 1. equal strips, one frame, image and ids
    render()
    assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_plain), "equal strips do not assemble to the one-GPU frame"
    assert hip.solr_hip_gather_ids(0) == 0
    if rank == 0:
        ids = np.zeros((H, W, 4), np.int32)
        assert hip.solr_hip_d2h_gathered_ids(C.c_void_p(ids.ctypes.data)) == 0
        assert np.array_equal(ids, ref_ids), "primitive-id strips do not assemble to the one-GPU buffer"
    k.check(0, "equal strips")

    # 2. three frames in flight, a gather behind every frame on its stream, no host wait in between
    hip.solr_hip_set_frames_in_flight(3)
    for i in range(9):
        render(dx=100.0 * i)
        assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_moved[8]), "frames in flight: the last assembled frame is not frame 8"
    hip.solr_hip_set_frames_in_flight(1)
    k.check(0, "frames in flight")

    # 2b. the delivered frame: one page-locked host image shared by the ranks' processes (solr_hip_image_share) - every
    # rank copies its strip over its own link behind its kernel, the host lags two frames, rank 0's wait returns when
    # every strip of that frame has landed; and the same frames through the gathered route (rank 0 copies the assembled
    # frame).  Eighteen frames: the ring of six is gone round three times
    hip.solr_hip_image_wait.restype = C.c_void_p
    name = ("/solr_ranks_%s" % os.path.basename(directory)).encode()
    if rank == 0:
        assert hip.solr_hip_image_share(name, rank, world) == 0
    barrier("shared image created")
    if rank != 0:
        assert hip.solr_hip_image_share(name, rank, world) == 0
    barrier("shared image open")
    hip.solr_hip_set_frames_in_flight(2)

    def image_of(ticket):
        ptr = hip.solr_hip_image_wait(ticket)
        assert ptr, "solr_hip_image_wait"
        return np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3).copy()

    for round_, route in enumerate(("strips", "gathered", "strips")):
        tickets, seen = [], []
        for i in range(18):
            render(dx=100.0 * (i % 9))
            assert hip.solr_hip_gather_strips(0) == 0
            t = hip.solr_hip_d2h_image_async() if route == "strips" else hip.solr_hip_d2h_gathered_async()
            assert t >= 0, (route, t)        # (with a shared ring every rank takes a ticket: they keep counting alike)
            tickets.append(t)
            if len(tickets) > 2 and tickets[-3] >= 0:
                seen.append(image_of(tickets[-3]))
        for t in tickets[-2:]:
            if t >= 0:
                seen.append(image_of(t))
        if rank == 0:
            assert len(seen) == 18
            for i, got in enumerate(seen):
                bad = np.flatnonzero((got != ref_moved[i % 9]).any(axis=(1, 2)))
                if len(bad):
                    print("DEBUG frame", i, route, "matches of rows 68..:", [j for j in range(9) if np.array_equal(got[68:], ref_moved[j][68:])],
                          "rows 68.. equal to rows 0..68 of frame:", [j for j in range(9) if np.array_equal(got[68:], ref_moved[j][:68])],
                          "mean abs diff", float(np.abs(got[68:].astype(int) - ref_moved[i % 9][68:].astype(int)).mean()),
                          "differing pixels", int((got[68:] != ref_moved[i % 9][68:]).any(axis=2).sum()), flush=True)
                assert len(bad) == 0, "delivered frame %d (%s) is not the one-GPU frame: rows %d..%d differ (%d rows; zero rows %d)" % (
                    i, route, bad[0], bad[-1], len(bad), int((~got.any(axis=(1, 2))).sum()))
        k.check(0, "delivered frames, " + route)
        barrier("delivered %s %d" % (route, round_))
    hip.solr_hip_set_frames_in_flight(1)

    # 3. ambient occlusion across the strips: agreed halo height, rank 0's random buffer on every rank
    render(occlusion)
    assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_occlusion), "ambient occlusion on strips differs from the one-GPU frame"
    # (another param2: a new agreement, fewer rows)
    render(solr.PostProcessingInfo(solr.ppe_ambientOcclusion, 0.0, 700.0, 0))
    assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_occlusion_narrow)
    k.check(0, "ambient occlusion on strips")

    # 4. cost-balanced strips: the same table on every rank, the same frame
    for _ in range(3):
        render()
        assert hip.solr_hip_gather_strips(0) == 0
    assert hip.solr_hip_balance_strips() == 0
    k.check(0, "solr_hip_balance_strips")
    report["balanced_strip"] = list(my_strip())
    render()
    assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_plain), "balanced strips do not assemble to the one-GPU frame"
    # (a strip cut without knowing the taps' reach may be lower than it, and the exchange trades rows with the next
    # rank only: this ambient-occlusion frame is rendered for the reach it records, not compared)
    render(occlusion)
    assert hip.solr_hip_gather_strips(0) == 0
    # ... and again after the ambient-occlusion frame: strips are now cut on multiples of the taps' reach
    assert hip.solr_hip_balance_strips() == 0
    report["balanced_strip_with_reach"] = list(my_strip())
    render(occlusion)
    assert hip.solr_hip_gather_strips(0) == 0
    if rank == 0:
        assert np.array_equal(gathered(), ref_occlusion), "ambient occlusion on strips cut on the taps' reach"
    k.check(0, "balanced strips")

    # 5. nobody waits for a rank in trouble.  The last rank renders a strip that is not the table's: it still sends
    # as many rows as the root expects (zeros), keeps an error and returns -1; the others carry on.
    if world > 1:
        victim = world - 1
        mine = my_strip()
        if rank == victim:
            hip.solr_hip_set_strip(mine[0], max(mine[1] - 8, 0))
        render()
        rc = hip.solr_hip_gather_strips(0)
        if rank == victim:
            assert rc == -1 and hip.solr_hip_last_error(None, 0) != 0
        else:
            assert rc == 0
        if rank == 0:
            image = gathered()
            table = json.loads(get(os.path.join(directory, "strip.%d" % victim)).decode()) if victim else list(mine)
            assert np.array_equal(image[:table[0]], ref_plain[:table[0]])
            assert not image[table[0]:table[0] + table[1]].any(), "a rank that could not send its rows sends zeros"
        if rank == victim:
            put(os.path.join(directory, "strip.%d" % victim), json.dumps(list(mine)).encode())
        # the same with the halo exchange of an ambient-occlusion frame and with the blocking all-reduce of the
        # balance: the rank in an error state takes part in both, everybody gets -1 from the balance
        render(occlusion)
        rc = hip.solr_hip_gather_strips(0)
        assert (rc == -1) == (rank == victim)
        assert hip.solr_hip_balance_strips() == -1, "a failed rank fails the balance on every rank"
        assert hip.solr_hip_last_error(None, 0) != 0
        hip.solr_hip_clear_error()
        # the table is unchanged; the victim takes its strip back and the next frame is whole again
        hip.solr_hip_set_strip(*mine)
        render()
        assert hip.solr_hip_gather_strips(0) == 0
        if rank == 0:
            assert np.array_equal(gathered(), ref_plain), "the frame after the trouble"
        k.check(0, "after the trouble")

    put(os.path.join(directory, "report.%d" % rank), json.dumps(report).encode())
    barrier("end")
    hip.solr_hip_set_strip(0, -1)
    hip.solr_hip_comm_finalize()
    k.finalize()
    print("MULTI_RANK_OK %d of %d (%s)" % (rank, world, transport), flush=True)


if __name__ == "__main__":
    main()
