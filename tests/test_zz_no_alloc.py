"""pt_read_tonemapped_noalloc (staging in a device array of the library, no allocation in the call) against
pt_read_tonemapped, through the C ABI: byte for byte, on two calls in a row, the second on a smaller image;
sizes beyond the staging array are refused."""
import ctypes

import numpy as np
import pytest

from conftest import scene_arrays

pytestmark = pytest.mark.gpu


def _renderer(pt, name, w, h, **kw):
    return pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), pt.Camera(resolution=(w, h)), **kw)


def _free_device_bytes():
    """hipMemGetInfo of the current device, from the HIP runtime the library itself is bound to (the soname
    resolves to the copy that is already loaded)."""
    hip = ctypes.CDLL("libamdhip64.so.7")
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def test_noalloc_tonemapped_readback_equals_the_allocating_one(pt):
    big = _renderer(pt, "Box", 640, 360, max_depth=6)
    big.draw(8)
    want_big = big.read_tonemapped()
    big.synchronize()
    free_before = _free_device_bytes()
    got_big = big.read_tonemapped_noalloc()
    assert _free_device_bytes() == free_before  # the call left device memory as it found it
    assert want_big.any()
    assert np.array_equal(got_big, want_big)

    # a smaller image right after: the staging array still holds the larger call's bytes behind it
    small = _renderer(pt, "Veggie", 96, 64, max_depth=6)
    small.draw(8)
    want_small = small.read_tonemapped()
    got_small = small.read_tonemapped_noalloc()
    assert want_small.any()
    assert np.array_equal(got_small, want_small)

    # and the first renderer again, after more samples and after a resize
    big.draw(4)
    assert np.array_equal(big.read_tonemapped_noalloc(), big.read_tonemapped())
    big.resize(131, 67)
    big.draw(3)
    assert np.array_equal(big.read_tonemapped_noalloc(), big.read_tonemapped())
    big.close()
    small.close()


def test_noalloc_readback_refuses_images_beyond_its_staging_array(pt):
    r = _renderer(pt, "Box", 2560, 1440, max_depth=2)
    r.draw(1)
    assert np.array_equal(r.read_tonemapped_noalloc(), r.read_tonemapped())  # the largest size it takes
    r.resize(2560, 1441)
    with pytest.raises(pt.PtError, match="exceed the staging array"):
        r.read_tonemapped_noalloc()
    assert r.read_tonemapped().shape == (1441, 2560, 3)
    r.close()
