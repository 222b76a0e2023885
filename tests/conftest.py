import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = os.path.join(ROOT, "tests")
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(TESTS, "golden")
SCENES = os.path.join(GOLDEN, "scenes")
LLVMPIPE = os.path.join(GOLDEN, "llvmpipe")
REFERENCE = os.environ.get("PT_REFERENCE_DIR", "/root/reference")

ALL_SCENES = sorted(os.path.splitext(f)[0] for f in os.listdir(SCENES) if f.endswith(".npz"))
CONFIG_SCENES = ["Box", "Veggie", "Box_Transmission", "Box_TransmissionSphere", "Box_Anisotropic", "AnisotropicTest"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_arrays(name):
    z = np.load(os.path.join(SCENES, name + ".npz"))
    return {k: z[k] for k in z.files}


def default_camera_axes():
    """eye, forward, up, right of the pose main.cpp:126 hard-codes, as bit
    patterns dumped from glm (oracle/_ref/ref_glm_dump basis 0 1.5 15 0 -0.25 0)."""
    eye = np.array([0.0, 1.5, 15.0], np.float32)
    fwd = np.array([0x00000000, 0xBDED52DC, 0xBF7E467E], np.uint32).view(np.float32)
    up = np.array([0x00000000, 0x3F7E467F, 0xBDED52DD], np.uint32).view(np.float32)
    right = np.array([0x3F7FFFFF, 0x00000000, 0x00000000], np.uint32).view(np.float32)
    return eye, fwd, up, right


def hit_fixture(name, w=256, h=256, it=0):
    return np.load(os.path.join(LLVMPIPE, f"hit_{name}_{w}x{h}_it{it}.npz"))


def image_fixture(name, depth=8):
    return np.load(os.path.join(LLVMPIPE, f"image_{name}_256x256_spp256_depth{depth}.npz"))


def box_filter(img, k=16):
    h, w = img.shape[:2]
    return img[: h - h % k, : w - w % k].reshape(h // k, k, w // k, k, -1).mean(axis=(1, 3))


def rel_mse(img, ref, eps=0.01):
    """mean over pixels and channels of (I - R)^2 / (R^2 + eps)."""
    img = np.asarray(img, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.mean((img - ref) ** 2 / (ref ** 2 + eps)))


@pytest.fixture(scope="session")
def pt():
    import pathtracer_amd

    return pathtracer_amd


@pytest.fixture(scope="session")
def ob():
    import oracle_binding

    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def have_reference():
    return os.path.isdir(os.path.join(REFERENCE, "assets"))
