"""The multi-GPU path of the C ABI (include/solr_hip.h: solr_hip_strip_rows, solr_hip_comm_*,
solr_hip_gather_strips): row strips gathered with RCCL called from the library itself, no torch.

CPU: the strip arithmetic equals the launcher's (sol-r_amd.strip_rows).  GPU: a communicator of ONE rank on the
box's one GPU - RCCL's send-to-self inside a group - carries every frame of the loop: full frame, a strip,
several frames in flight; the gathered image must be the rendered one.  (N > 1 needs as many GPUs as ranks; the
world-size-2 logic of the partition and the assembly is covered on CPU by tests/test_strips_gloo.py.)"""
import ctypes as C

import numpy as np
import pytest


def test_strip_rows_of_the_library_are_the_launchers(solr):
    hip = solr.hip_lib()
    first, count, per = C.c_int(), C.c_int(), C.c_int()
    for height in (1, 7, 17, 45, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            covered = []
            for rank in range(world):
                hip.solr_hip_strip_rows(rank, world, height, C.byref(first), C.byref(count), C.byref(per))
                assert (first.value, count.value, per.value) == solr.strip_rows(rank, world, height)
                covered += list(range(first.value, first.value + count.value))
            assert covered == list(range(height))


def test_gather_without_a_communicator_is_refused(solr, have_gpu):
    hip = solr.hip_lib()
    hip.solr_hip_clear_error()
    assert hip.solr_hip_gather_strips(0) == -1
    assert hip.solr_hip_last_error(None, 0) != 0
    hip.solr_hip_clear_error()


@pytest.mark.gpu
@pytest.mark.parametrize("per_flight", [0, 1], ids=["one communicator", "one communicator per flight"])
def test_native_rccl_gather_with_one_rank(solr, per_flight):
    """both communicator modes (solr_hip_comm_set_per_flight): RCCL orders the operations of ONE communicator, so the
    first N > 1 run can A/B whether gathers of frames in flight on different streams serialise; with real RCCL and one
    rank both carry the same frames"""
    W, H = 192, 136
    hip = solr.hip_lib()
    hip.solr_hip_gathered_frame.restype = C.c_void_p
    hip.solr_hip_image_wait.restype = C.c_void_p
    hip.solr_hip_comm_set_per_flight(per_flight)
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    full = k.render()
    k.check(0, "first frame")
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    si.pathTracingIteration = 0
    objects = solr.Vec4i(len(flat.boxes), len(flat.primitives), flat.nb_lamps, len(flat.lights))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731

    def render(dx=0.0):
        e = eye.copy()
        e[0] += dx
        hip.solr_hip_render(C.byref(si), C.byref(objects), C.byref(ppi), fp(e), fp(direction), fp(angles))

    uid = C.create_string_buffer(128)
    try:
        assert hip.solr_hip_comm_unique_id(uid) == 0
        assert hip.solr_hip_comm_init(0, 1, uid) == 0
        k.check(0, "communicator")
        assert hip.solr_hip_comm_count() == (4 if per_flight else 1) and hip.solr_hip_comm_ranks() == 1
        # the whole frame
        render()
        assert hip.solr_hip_gather_strips(0) == 0
        image = np.zeros((H, W, 3), np.uint8)
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        assert np.array_equal(image, full)
        # a strip: lands at its rows of the assembled frame, the other rows keep the frame before
        image[:] = 0
        hip.solr_hip_set_strip(40, 32)
        render(dx=700.0)
        moved = np.zeros((H, W, 3), np.uint8)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(moved.ctypes.data), None)     # places the strip at its rows
        assert hip.solr_hip_gather_strips(0) == 0
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        assert np.array_equal(image[40:72], moved[40:72]) and not np.array_equal(moved[40:72], full[40:72])
        assert np.array_equal(image[:40], full[:40]) and np.array_equal(image[72:], full[72:])
        hip.solr_hip_set_strip(0, -1)
        # three frames in flight, a gather behind every frame on its own stream, no host wait in between
        hip.solr_hip_set_frames_in_flight(3)
        shifts = [100.0 * i for i in range(9)]
        for dx in shifts:
            render(dx)
            assert hip.solr_hip_gather_strips(0) == 0
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        hip.solr_hip_set_frames_in_flight(1)
        render(shifts[-1])
        expected = np.zeros((H, W, 3), np.uint8)
        hip.solr_hip_d2h(C.byref(si), C.c_void_p(expected.ctypes.data), None)
        assert np.array_equal(image, expected)
        k.check(0, "gather loop")
        # the delivered frame, pipelined: the assembled frame of every gather copied to a page-locked image behind
        # it, the host two frames behind (solr_hip_d2h_gathered_async; what bench.py's headline times at N > 1)
        hip.solr_hip_set_frames_in_flight(2)
        tickets, seen = [], []
        for dx in shifts:
            render(dx)
            assert hip.solr_hip_gather_strips(0) == 0
            tickets.append(hip.solr_hip_d2h_gathered_async())
            assert tickets[-1] >= 0
            if len(tickets) > 2:
                ptr = hip.solr_hip_image_wait(tickets[-3])
                assert ptr
                seen.append(np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3).copy())
        for t in tickets[-2:]:
            ptr = hip.solr_hip_image_wait(t)
            seen.append(np.frombuffer((C.c_ubyte * (W * H * 3)).from_address(ptr), np.uint8).reshape(H, W, 3).copy())
        hip.solr_hip_set_frames_in_flight(1)
        for dx, got in zip(shifts, seen):
            render(dx)
            hip.solr_hip_d2h(C.byref(si), C.c_void_p(expected.ctypes.data), None)
            assert np.array_equal(got, expected), dx
        k.check(0, "pipelined delivery")
    finally:
        hip.solr_hip_set_strip(0, -1)
        hip.solr_hip_set_frames_in_flight(1)
        hip.solr_hip_comm_finalize()
        hip.solr_hip_comm_set_per_flight(0)
        hip.solr_hip_clear_error()
        k.finalize()


# ---- cost-balanced strips (solr_hip_balanced_strips, solr_hip_strip_row_costs, solr_hip_set_strip_table,
#      solr_hip_balance_strips) ------------------------------------------------------------------------------------

def _check_partition(strips, height, align=8):
    at = 0
    for first, count in strips:
        assert count >= 0
        if count:
            assert first == at and first % align == 0
            at += count
    assert at == height


def test_balanced_strips_partition_the_frame_and_even_out_the_cost(solr):
    rng = np.random.RandomState(5)
    for height in (8, 45, 200, 1080, 2160):
        for world in (1, 2, 3, 8):
            for trial in range(6):
                cost = rng.gamma(0.3, 1000.0, size=height).astype(np.float32)
                if trial == 0:
                    cost[:] = 1.0
                if trial == 1:
                    cost[: height // 3] = 0.0
                strips = solr.balanced_strips(cost, world)
                _check_partition(strips, height)
                blocks = (height + 7) // 8
                if blocks >= world:
                    assert all(count > 0 for _, count in strips), (height, world, strips)
                    # every boundary sits at the block edge nearest its share of the total - or as near as the
                    # one block every rank is owed allows: no strip is more than a block (and the owed ones) over
                    floor = 1e-3 * float(cost.astype(np.float64).sum()) / height
                    block = [float(cost[b * 8:(b + 1) * 8].astype(np.float64).sum()) + floor * len(cost[b * 8:(b + 1) * 8])
                             for b in range(blocks)]
                    share = sum(block) / world
                    worst = max(sum(block[f // 8:(f + c + 7) // 8]) for f, c in strips)
                    assert worst <= share + 2.0 * max(block) + 1e-3 * share, (height, world, trial)
    # equal costs, and no costs at all (a frame not rendered yet): the equal split to within a tile
    for cost in (np.ones(1080, np.float32), np.zeros(1080, np.float32)):
        strips = solr.balanced_strips(cost, 8)
        assert all(abs(count - 135) <= 8 for _, count in strips)
    # a frame whose cost sits in its top tile rows (the mesh's horizon): those rows are shared out one tile each
    cost = np.zeros(1080, np.float32)
    cost[:56] = 1.0e6
    strips = solr.balanced_strips(cost, 8)
    assert [c for _, c in strips[:7]] == [8] * 7 and strips[7] == (56, 1024)
    # strips no lower than the reach of the ambient-occlusion taps: cut on multiples of it
    rng = np.random.RandomState(9)
    cost = rng.gamma(0.3, 1000.0, size=2160).astype(np.float32)
    for align in (24, 40, 272):
        strips = solr.balanced_strips(cost, 8, align=align)
        _check_partition(strips, 2160, align=align)
        if (2160 + align - 1) // align >= 8:
            assert all(count >= min(align, 2160 - first) for first, count in strips)
    hip = solr.hip_lib()
    assert hip.solr_hip_balanced_strips(None, 10, 2, 8, None, None) == -1


def test_strip_table_has_to_partition_the_frame(solr):
    hip = solr.hip_lib()
    hip.solr_hip_clear_error()
    assert solr.set_strip_table([(0, 40), (40, 24), (64, 72)], 136) == 0
    assert solr.set_strip_table([(0, 40), (40, 0), (40, 96)], 136) == 0      # an empty strip is a strip
    for bad in ([(0, 40), (48, 88)], [(0, 40), (40, 90)], [(40, 96), (0, 40)], [(0, 40), (40, -1)]):
        assert solr.set_strip_table(bad, 136) == -1
        assert hip.solr_hip_last_error(None, 0) != 0
        hip.solr_hip_clear_error()
    assert solr.set_strip_table(None, 0) == 0


@pytest.mark.gpu
def test_balanced_strips_render_the_same_frame(solr):
    """rows' costs from a rendered frame, strips of equal cost from them, every strip rendered on its own: the
    strips add up to the frame, and they differ from the equal ones where the cost does"""
    W, H = 320, 200
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    solr.scenes.height_field(k, n=48, width=W, height=H)
    try:
        full = k.render()
        full = k.render()
        k.check(0, "frames")
        cost = solr.strip_row_costs(H)
        assert cost.shape == (H,) and (cost >= 0).all() and cost.sum() > 0
        world = 4
        strips = solr.balanced_strips(cost, world)
        _check_partition(strips, H)
        assert strips != [solr.strip_rows(r, world, H)[:2] for r in range(world)]   # sky above, terrain below
        shares = [cost[f:f + c].sum() / cost.sum() for f, c in strips]
        assert max(shares) < 0.25 + cost.reshape(-1, 8).sum(axis=1).max() / cost.sum() + 1e-6
        assembled = np.zeros_like(full)
        for first, count in strips:
            hip.solr_hip_set_strip(first, count)
            image = k.render()
            assembled[first:first + count] = image[first:first + count]
            # the costs of a strip are reported at the strip's rows only
            part = solr.strip_row_costs(H)
            assert part[first:first + count].sum() > 0 and part[:first].sum() == 0 and part[first + count:].sum() == 0
        assert np.array_equal(assembled, full)
    finally:
        hip.solr_hip_set_strip(0, -1)
        k.finalize()


@pytest.mark.gpu
def test_balance_strips_with_one_rank(solr):
    """solr_hip_balance_strips through RCCL with a communicator of one rank: the all-reduce of the rows' costs,
    the table, this rank's strip (the whole frame) - and the gather still assembles the frame"""
    W, H = 192, 136
    hip = solr.hip_lib()
    k = solr.Kernel(engine="hip")
    solr.scenes.cornell(k, width=W, height=H, iterations=2)
    uid = C.create_string_buffer(128)
    try:
        full = k.render()
        hip.solr_hip_clear_error()
        assert hip.solr_hip_balance_strips() == -1          # no communicator yet
        hip.solr_hip_clear_error()
        assert hip.solr_hip_comm_unique_id(uid) == 0
        assert hip.solr_hip_comm_init(0, 1, uid) == 0
        hip.solr_hip_set_strip(40, 32)                       # whatever it was: one rank gets the frame
        k.render()
        assert hip.solr_hip_balance_strips() == 0
        k.check(0, "balance")
        again = k.render()
        assert np.array_equal(again, full)
        assert hip.solr_hip_gather_strips(0) == 0
        image = np.zeros((H, W, 3), np.uint8)
        assert hip.solr_hip_d2h_gathered(C.c_void_p(image.ctypes.data)) == 0
        assert np.array_equal(image, full)
    finally:
        hip.solr_hip_set_strip(0, -1)
        hip.solr_hip_comm_finalize()
        hip.solr_hip_clear_error()
        k.finalize()
