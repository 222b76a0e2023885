"""N > 1 path on CPU: two gloo ranks partition the iteration set, each produces
its radiance sum (the oracle stands in for the GPU renderer here: no GPU in
this container), one reduce combines them, and the root's image equals the
single-process render of all iterations up to fp32 addition order."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, default_camera_axes, scene_arrays

W, H, SPP, DEPTH = 24, 16, 6, 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import oracle_binding as ob
    from pathtracer_amd import distributed as ptd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = ob.OracleScene(scene_arrays("Box"))
    cam = ob.make_camera(*default_camera_axes(), W, H)
    local = np.zeros((H, W, 4), np.float32)
    its = list(ptd.iteration_indices(SPP, rank, world))
    assert len(its) == ptd.local_iterations(SPP, rank, world)
    for it in its:
        ob.render(sc, cam, it, it + 1, DEPTH, 1, local)
    t = torch.from_numpy(local)
    dist.barrier()
    ptd.reduce_sum_to_root(t, 0)
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_partition_and_reduce(tmp_path, ob):
    import torch.multiprocessing as mp

    out = str(tmp_path / "sum.npy")
    mp.start_processes(_worker, args=(2, _free_port(), out), nprocs=2, join=True, start_method="spawn")
    got = np.load(out)
    sc = ob.OracleScene(scene_arrays("Box"))
    ref, _ = ob.render(sc, ob.make_camera(*default_camera_axes(), W, H), 0, SPP, DEPTH, 1)
    assert np.all(got[..., 3] == SPP)
    assert np.allclose(got[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)


def test_partition_arithmetic():
    from pathtracer_amd import distributed as ptd

    for spp in (0, 1, 7, 256, 1023):
        for world in (1, 2, 3, 8):
            parts = [list(ptd.iteration_indices(spp, r, world)) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(spp))
            assert [len(p) for p in parts] == [ptd.local_iterations(spp, r, world) for r in range(world)]
    with pytest.raises(ValueError):
        ptd.local_iterations(4, 2, 2)
    m = ptd.mean_image(np.array([[[2.0, 4.0, 6.0, 2.0], [0, 0, 0, 0]]], np.float32))
    assert m.tolist() == [[[1.0, 2.0, 3.0], [0.0, 0.0, 0.0]]]
