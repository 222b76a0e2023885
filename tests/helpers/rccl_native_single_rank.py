"""Run in a fresh process that never imports torch: the library's own RCCL path (pt_comm_*, pt_reduce_sum)
in a world of one rank -- communicator from a unique id, and one made for all devices of the process --
on a renderer's sum buffer.  Prints OK on success."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import pathtracer_amd as pt

    assert "torch" not in sys.modules
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", "Box.npz"))
    scene = pt.Scene.from_arrays(**{k: z[k] for k in z.files})
    straight = pt.Renderer(scene, pt.Camera(resolution=(96, 64)), max_depth=5, device=0)
    straight.draw(9)
    want = straight.read_sum()
    straight.close()

    uid = pt.comm_unique_id()
    assert len(uid) == pt.COMM_ID_BYTES and any(uid)
    comm = pt.Comm(uid, 0, 1, device=0)
    r = pt.Renderer(scene, pt.Camera(resolution=(96, 64)), max_depth=5, device=0)
    r.set_partition(0, 1)
    r.draw(4)
    host = r.read_sum()
    pt.reduce_sum([r], [comm], root=0)  # ncclReduce on the renderer's stream, in place
    assert np.array_equal(r.read_sum(), host), "a one-rank reduce changed the buffer"
    r.draw(5)  # the second batch stream has to wait for the reduce
    pt.reduce_sum([r], [comm], root=0)
    assert np.array_equal(r.read_sum(), want), "render -> reduce -> render -> reduce differs from a straight render"
    try:
        pt.reduce_sum([r], [comm], root=1)
    except pt.PtError as e:
        assert "root" in str(e)
    else:
        raise AssertionError("a root outside the communicator was accepted")
    comm.close()

    (c0,) = pt.Comm.for_devices([0])
    pt.reduce_sum([r], [c0], root=0)
    r.synchronize()
    assert np.array_equal(r.read_sum(), want)
    small = pt.Renderer(scene, pt.Camera(resolution=(64, 64)), max_depth=5, device=0)
    try:
        pt.reduce_sum([r, small], [c0, c0], root=0)
    except pt.PtError as e:
        assert "sizes differ" in str(e)
    else:
        raise AssertionError("renderers of different sizes were accepted")
    small.close()
    c0.close()
    r.close()
    print("OK")


if __name__ == "__main__":
    main()
