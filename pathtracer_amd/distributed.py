"""Sample-space partitioning across GPUs: one process per GPU, scene replicated,
rank g of G renders iterations g, g+G, g+2G, ... and the per-GPU radiance sums
are combined with ONE reduce (RCCL over xGMI when the backend is "nccl").

The reference is single-GPU (one GL context); what makes this partition exact
is that a sample depends only on (pixel, iteration index)
(pathtracer.frag.glsl:866) and that accumulation here is a float SUM rather
than the reference's 8-bit running average (pathtracer.frag.glsl:900-904).

The reduce is the library's own (`pt_reduce_sum`: ncclReduce on the renderer's
stream); this module only bootstraps its communicator in a process group that
already exists (`native_comm`: rank 0's 128-byte id travels through
`torch.distributed.broadcast_object_list`, any backend -- gloo will do, torch
never touches the GPU).  `sum_buffer_as_tensor` / `reduce_sum_to_root` are the
older route through torch's own RCCL binding, kept for callers whose tensors
live in torch anyway; it needs torch imported before pathtracer_amd (torch's
wheel brings its own HIP runtime, and the first one loaded is the one both use).
"""
from __future__ import annotations

import numpy as np


def local_iterations(total_spp: int, rank: int, world: int) -> int:
    """How many of the iterations 0..total_spp-1 fall to `rank`."""
    if not (0 <= rank < world):
        raise ValueError("need 0 <= rank < world")
    return max(0, (total_spp - rank + world - 1) // world)


def iteration_indices(total_spp: int, rank: int, world: int) -> range:
    return range(rank, total_spp, world)


def native_comm(rank: int, world: int, device: int):
    """A library-owned RCCL communicator for this rank of an initialised torch.distributed group."""
    import torch.distributed as dist

    import pathtracer_amd as pt

    box = [pt.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    return pt.Comm(box[0], rank, world, device)


class _DeviceArray:
    """Exposes a raw device pointer through __cuda_array_interface__ so that
    torch can alias it without a copy."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def sum_buffer_as_tensor(renderer, device):
    """The renderer's device-resident sum buffer (H x W x 4 float32) as a torch
    tensor that aliases it."""
    import torch

    if not torch.cuda.is_available():
        # torch's wheel carries its own HIP runtime; whichever of torch and this library is loaded first decides
        # which runtime the process uses, and torch only finds the GPUs with its own.
        raise RuntimeError("torch sees no GPU: in a multi-GPU process import torch (and set the device) before "
                           "pathtracer_amd, as bench.py does")
    arr = _DeviceArray(renderer.sum_device_ptr(), (renderer.height, renderer.width, 4))
    return torch.as_tensor(arr, device=device)


def reduce_sum_to_root(tensor, root: int = 0):
    """One SUM reduce of the radiance-sum buffer into `root` (in place)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM)
    return tensor


def mean_image(sum_rgba: np.ndarray) -> np.ndarray:
    """sum / count per pixel; the alpha lane carries the sample count."""
    cnt = np.maximum(sum_rgba[..., 3:4], 1.0)
    return sum_rgba[..., :3] / cnt
