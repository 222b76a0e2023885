"""Drives oracle/_ref/lp_harness: the reference's own fragment shader on Mesa
llvmpipe.  Container-only test tooling (needs /root/reference and Mesa)."""
import os
import re
import struct
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HARNESS = os.path.join(os.path.dirname(HERE), "_ref", "lp_harness")


def available() -> bool:
    ref = os.environ.get("PT_REFERENCE_DIR", "/root/reference")
    return os.path.exists(HARNESS) and os.path.exists(ref + "/src/shaders/pathtracer/pathtracer.frag.glsl")


def write_scene(path, arrays, eye, forward, up, right):
    a = arrays
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", a["vertices"].shape[0], a["indices"].shape[0], a["lights"].shape[0],
                            a["materials"].shape[0]))
        for k, ty in (("vertices", np.float32), ("indices", np.float32), ("vertex_data", np.float32),
                      ("lights", np.float32), ("materials", np.float32), ("material_map", np.uint32)):
            f.write(np.ascontiguousarray(a[k], ty).tobytes())
        f.write(np.concatenate([eye, forward, up, right]).astype(np.float32).tobytes())


def run(arrays, cam, width, height, mode, iters=1, depth=10):
    """cam = (eye, forward, up, right).  For the debug modes 'ids' / 'tn', `depth` is the iteration index.
    Returns (array HxWx4, seconds); for mode 'post' the array is a pair (float image, what post.frag drew as uint8)."""
    with tempfile.TemporaryDirectory() as tmp:
        sp, op = os.path.join(tmp, "scene.bin"), os.path.join(tmp, "out.bin")
        write_scene(sp, arrays, *cam)
        out = subprocess.run([HARNESS, sp, op, str(width), str(height), mode, str(iters), str(depth)],
                             check=True, capture_output=True, text=True)
        secs = float(re.search(r"seconds=([0-9.]+)", out.stdout).group(1))
        if mode == "post":
            raw = np.fromfile(op, np.uint8)
            n = width * height * 4
            img = (raw[:4 * n].view(np.float32).reshape(height, width, 4).copy(),
                   raw[4 * n:].reshape(height, width, 4).copy())
        else:
            dt = np.uint8 if mode == "rgba8" else np.float32
            img = np.fromfile(op, dt).reshape(height, width, 4)
    return img, secs
