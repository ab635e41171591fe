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


@pytest.mark.parametrize("height", [48, 45])   # 45: the last strip is one row shorter
def test_two_strips_equal_the_full_frame(solr, oracle, tmp_path, height):
    import torch.multiprocessing as mp
    width, world = 64, 2
    out = str(tmp_path / "gathered.npz")
    port = 29500 + (os.getpid() % 2000) + height
    mp.spawn(_worker, args=(world, port, width, height, out), nprocs=world, join=True)
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
