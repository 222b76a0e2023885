"""Run in a fresh process (torch first, like bench.py's multi-GPU path): the
renderer's sum buffer aliased as a torch tensor and one RCCL collective on it
in a world of one rank.  Prints OK on success."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    import pathtracer_amd as pt
    from pathtracer_amd import distributed as ptd

    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", "Box.npz"))
    scene = pt.Scene.from_arrays(**{k: z[k] for k in z.files})
    r = pt.Renderer(scene, pt.Camera(resolution=(96, 64)), max_depth=5, device=0)
    r.set_partition(0, 1)
    r.draw(4)
    r.synchronize()
    host = r.read_sum()
    t = ptd.sum_buffer_as_tensor(r, "cuda:0")
    assert t.shape == (64, 96, 4) and t.dtype == torch.float32 and t.is_cuda
    assert t.data_ptr() == r.sum_device_ptr(), "not an alias"
    assert np.array_equal(t.cpu().numpy(), host), "aliased contents differ"
    dist.all_reduce(t, op=dist.ReduceOp.SUM)  # RCCL on the renderer's own memory
    ptd.reduce_sum_to_root(t, 0)
    torch.cuda.synchronize()
    assert np.array_equal(r.read_sum(), host), "a one-rank reduce changed the buffer"
    dist.barrier()
    dist.destroy_process_group()
    r.close()
    print("OK")


if __name__ == "__main__":
    main()
