"""The N > 1 path on CPU: two processes (gloo), each renders its row strip of the frame, one gather
assembles the image on rank 0.  The strips are rendered by the CPU oracle here (no GPU in this
container); what is under test is the partition arithmetic and the gather that bench.py uses
(sol-r_amd.strip_rows / gather_strips) and that an N-strip frame equals the 1-strip frame."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(worker, args, world):
    """`world` fresh processes running worker(rank, *args).  The parent does not import torch: torch brings its own
    copy of the HIP runtime, and a process that has the engine library's copy initialised already (the GPU tests
    of the same pytest run on a GPU box) corrupts its heap on exit with both."""
    import multiprocessing
    ctx = multiprocessing.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(rank,) + tuple(args)) for rank in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    solr = importlib.import_module("sol-r_amd")
    from oracle import loader
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k, width=width, height=height, iterations=2)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    first, count, rows_per_rank = solr.strip_rows(rank, world, height)
    _, _, rgb, counts, status = loader.render(flat, si, ppi, eye, direction, angles, first_row=first, nb_rows=count,
                                              nthreads=2)
    assert status == 0
    strip = torch.from_numpy(rgb.reshape(-1).copy())
    image = solr.gather_strips(dist, torch, strip, rows_per_rank, width, height, rank, world)
    rays = torch.tensor([counts[0] + counts[1]], dtype=torch.int64)
    dist.all_reduce(rays)
    if rank == 0:
        np.savez(out_path, image=image.numpy(), rays=rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


# 45: the last strip is one row shorter; (4, 3): strips of 2 rows leave the third process without a row
@pytest.mark.parametrize("height,world", [(48, 2), (45, 2), (4, 3)])
def test_strips_equal_the_full_frame(solr, oracle, tmp_path, height, world):
    width = 64
    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000) + height
    _spawn(_worker, (world, port, width, height, out), world)
    got = np.load(out)
    k = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k, width=width, height=height, iterations=2)
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    _, _, full, counts, status = oracle.render(flat, si, ppi, eye, direction, angles, nthreads=2)
    assert status == 0
    assert got["image"].shape == (height, width, 3)
    assert np.array_equal(got["image"], full)
    assert int(got["rays"][0]) == counts[0] + counts[1]


def test_strip_rows_cover_the_frame_exactly(solr):
    for height in (1, 7, 45, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            rows = [solr.strip_rows(r, world, height) for r in range(world)]
            covered = []
            for first, count, per in rows:
                assert per == rows[0][2] and count <= per
                covered += list(range(first, first + count))
            assert covered == list(range(height))


def _pipeline_worker(rank, world, port, width, height, nb_frames, out_path, pipelined=True):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    solr = importlib.import_module("sol-r_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sg = solr.StripGather(dist, torch, width, height, rank, world, pipelined=pipelined)
    images = []
    for i in range(nb_frames):
        buf = sg.buffer(i)                       # what bench.py binds as the engine's bitmap
        rows = buf[: sg.nb_rows * width * 3].reshape(sg.nb_rows, width, 3)
        y = torch.arange(sg.first_row, sg.first_row + sg.nb_rows).reshape(-1, 1, 1)
        x = torch.arange(width).reshape(1, -1, 1)
        c = torch.arange(3).reshape(1, 1, -1)
        rows.copy_(((y * 7 + x * 3 + c * 5 + i * 11) % 251).to(torch.uint8))   # stands in for the render
        sg.submit(i)
        if not pipelined:                        # in-order: frame i is assembled when submit returns
            img = sg.image(i)
            if rank == 0:
                images.append(img.clone().numpy())
        elif i >= 1:                             # frame i-1 completes while frame i is in flight
            img = sg.image(i - 1)
            if rank == 0:
                images.append(img.clone().numpy())
    if pipelined:
        img = sg.image(nb_frames - 1)
        if rank == 0:
            images.append(img.clone().numpy())
    if rank == 0:
        np.savez(out_path, images=np.stack(images))
    sg.drain()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,pipelined", [(48, True), (45, True), (45, False)])
def test_pipelined_gather_assembles_every_frame(solr, tmp_path, height, pipelined):
    """StripGather (what bench.py runs at N > 1): in-order on one buffer, or two buffers in flight;
    every frame assembled intact."""
    width, world, nb_frames = 40, 2, 5
    out = str(tmp_path / "frames.npz")
    port = 31500 + (os.getpid() % 2000) + height + (7 if pipelined else 0)
    _spawn(_pipeline_worker, (world, port, width, height, nb_frames, out, pipelined), world)
    images = np.load(out)["images"]
    assert images.shape == (nb_frames, height, width, 3)
    y = np.arange(height).reshape(-1, 1, 1)
    x = np.arange(width).reshape(1, -1, 1)
    c = np.arange(3).reshape(1, 1, -1)
    for i in range(nb_frames):
        assert np.array_equal(images[i], ((y * 7 + x * 3 + c * 5 + i * 11) % 251).astype(np.uint8)), i


def _balanced_worker(rank, world, port, width, height, out_path):
    """the cost-balanced form of the loop: equal strips first, the rows' costs summed over
    the ranks, the same balanced table on every rank, the new strips rendered and gathered.  The cost of a row
    is taken from the oracle's frame here (pixels that hit something) - on the GPU it is the tiles' durations."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    solr = importlib.import_module("sol-r_amd")
    from oracle import loader
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k, width=width, height=height, iterations=2, room=False)
    k.set_camera((0.0, 0.0, -15000.0), look_at=(0.0, 3500.0, 0.0))    # the spheres in the lower half of the frame
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    first, count, _ = solr.strip_rows(rank, world, height)
    cost = torch.zeros(height, dtype=torch.float32)
    if count > 0:
        _, ids, _, _, status = loader.render(flat, si, ppi, eye, direction, angles, first_row=first, nb_rows=count,
                                             nthreads=2)
        assert status == 0
        cost[first:first + count] = torch.from_numpy((ids[..., 0] >= 0).sum(axis=1).astype(np.float32))
    dist.all_reduce(cost)
    table = solr.balanced_strips(cost.numpy(), world)
    first, count = table[rank]
    rgb = np.zeros((0, width, 3), np.uint8)
    if count > 0:
        _, _, rgb, _, status = loader.render(flat, si, ppi, eye, direction, angles, first_row=first, nb_rows=count,
                                             nthreads=2)
        assert status == 0
    strip = torch.from_numpy(rgb.reshape(-1).copy())
    image = solr.gather_strips(dist, torch, strip, 0, width, height, rank, world, strips=table)
    if rank == 0:
        np.savez(out_path, image=image.numpy(), table=np.array(table), cost=cost.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height,world", [(64, 2), (72, 3)])
def test_balanced_strips_equal_the_full_frame(solr, oracle, tmp_path, height, world):
    width = 64
    out = str(tmp_path / "balanced.npz")
    port = 31500 + (os.getpid() % 2000) + height
    _spawn(_balanced_worker, (world, port, width, height, out), world)
    got = np.load(out)
    k = solr.Kernel(engine="host-only")
    solr.scenes.cornell(k, width=width, height=height, iterations=2, room=False)
    k.set_camera((0.0, 0.0, -15000.0), look_at=(0.0, 3500.0, 0.0))    # the spheres in the lower half of the frame
    flat = k.flat_scene()
    si, ppi, eye, direction, angles = k.frame_parameters()
    _, _, full, _, status = oracle.render(flat, si, ppi, eye, direction, angles, nthreads=2)
    assert status == 0
    assert np.array_equal(got["image"], full)
    table, cost = got["table"], got["cost"]
    assert table[:, 1].sum() == height and (table[:, 0] % 8 == 0).all()
    # without the room the spheres sit in the middle rows: the balanced strips are not the equal ones, and no
    # strip carries more than its share plus a tile row
    assert [tuple(t) for t in table] != [solr.strip_rows(r, world, height)[:2] for r in range(world)]
    shares = [cost[f:f + c].sum() for f, c in table]
    assert max(shares) <= cost.sum() / world + cost.reshape(-1, 8).sum(axis=1).max() + 1e-3


# ---- what the ranks of bench.py agree on (world size 2, gloo) ---------------------------------------------------
def _bench_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import json
    import torch
    import torch.distributed as dist
    import bench_dist as bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 1 is the slower one in regions 0 and 2, rank 0 in region 1; rank 1's kernel is the slower
    regions = [[0.010, 0.030, 0.011], [0.020, 0.012, 0.015]][rank]
    second = [[0.5, 0.1], [0.2, 0.3]][rank]
    got = bench.reduce_over_ranks(dist, torch, regions, second, [0.25, 0.40][rank], [1000, 2345][rank])
    # only rank 1 fails to open the shared image: both must decide to fall back; nobody failing: nobody falls back
    fallback = bench.agree(dist, torch, rank == 1)
    none = bench.agree(dist, torch, False)
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump({"reduced": got, "fallback": fallback, "none": none, "median": bench.median(got[0])}, f)
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_of_the_bench_agree_on_times_rays_and_the_delivery_route(tmp_path):
    """bench.py at N > 1: a region's time is the slowest rank's, region by region (so that the median region is a
    region that every rank finished); rays add up; one rank that cannot share the host image sends all of them to the
    gathered route - the N > 1 decisions, run with two gloo processes"""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "bench_agreement")
    _spawn(_bench_worker, (2, port, out), 2)
    a, b = (json.load(open("%s.%d" % (out, r))) for r in (0, 1))
    assert a == b                                   # every rank holds the same figures
    regions, second, kernel_ms, rays = a["reduced"]
    assert regions == [0.020, 0.030, 0.015] and second == [0.5, 0.3] and kernel_ms == 0.40 and rays == 3345
    assert a["median"] == 0.020 and a["fallback"] is True and a["none"] is False
