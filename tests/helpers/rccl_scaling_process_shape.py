"""Run in a fresh process with the shape the multi-GPU benchmark's ranks have: torch.distributed (gloo) is
imported and initialised FIRST, then the library, then the library's own RCCL communicator is made from an id
that travelled through torch.distributed (distributed.native_comm), in a world of one rank.  Checks that the
process ends up with ONE HIP runtime and ONE RCCL mapped, that the RCCL the library reduces with calls into
the HIP runtime the library itself is bound to, and that render -> reduce -> render -> reduce gives the image
of a straight render.  Prints the mapped libraries and OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def mapped(fragment):
    with open("/proc/self/maps") as f:
        return sorted({line.split()[-1] for line in f if fragment in line and "/" in line.split()[-1]})


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29531")
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=0, world_size=1)
    import pathtracer_amd as pt
    from pathtracer_amd import distributed as ptd

    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", "Box.npz"))
    scene = pt.Scene.from_arrays(**{k: z[k] for k in z.files})
    straight = pt.Renderer(scene, pt.Camera(resolution=(96, 64)), max_depth=5, device=0)
    straight.draw(9)
    want = straight.read_sum()
    straight.close()

    comm = ptd.native_comm(0, 1, 0)
    paths = pt.comm_runtime_paths()
    print("library:", paths)
    hip, rccl = mapped("libamdhip64"), mapped("librccl")
    print("mapped HIP runtimes:", hip)
    print("mapped RCCL copies:", rccl)
    assert len(hip) == 1, "two HIP runtimes in one process: streams of one mean nothing to the other"
    assert len(rccl) == 1, "two RCCL copies in one process"
    assert os.path.realpath(paths["hip"]) == os.path.realpath(paths["rccl_hip"]), paths
    assert os.path.realpath(paths["rccl"]) == os.path.realpath(rccl[0]), (paths, rccl)

    r = pt.Renderer(scene, pt.Camera(resolution=(96, 64)), max_depth=5, device=0)
    r.set_partition(0, 1)
    r.draw(4)
    pt.reduce_sum([r], [comm], root=0)
    r.draw(5)
    pt.reduce_sum([r], [comm], root=0)
    assert np.array_equal(r.read_sum(), want), "render -> reduce -> render -> reduce differs from a straight render"
    r.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print("OK")


if __name__ == "__main__":
    main()
