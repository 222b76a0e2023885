"""Host camera (camera.cpp) and light construction (scene.h:27-38): the bits
that reach the integrator must be the bits the reference's glm code produces.
Golden patterns were dumped from the reference's vendored glm
(oracle/ref_glm_dump.cpp); where oracle/_ref is built, random poses are
compared against it directly."""
import math
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, default_camera_axes

GLM_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_glm_dump")
needs_glm = pytest.mark.skipif(not os.path.exists(GLM_DUMP), reason="oracle/_ref/ref_glm_dump not built")


def _glm(*args):
    out = subprocess.run([GLM_DUMP, *[repr(float(a)) if not isinstance(a, str) else a for a in args]], check=True,
                         capture_output=True, text=True).stdout
    rows = {}
    for line in out.splitlines():
        w = line.split()
        if len(w) == 1:
            return np.array([int(w[0], 16)], np.uint32).view(np.float32)[0]
        rows[w[0]] = np.array([int(x, 16) for x in w[1:]], np.uint32)
    return rows


def test_default_pose_basis_bits(pt):
    cam = pt.Camera()  # main.cpp:126
    cam.update()
    eye, fwd, up, right = default_camera_axes()
    assert np.array_equal(cam.eye, eye)
    assert np.array_equal(cam.forward.view(np.uint32), fwd.view(np.uint32))
    assert np.array_equal(cam.up.view(np.uint32), up.view(np.uint32))
    assert np.array_equal(cam.right.view(np.uint32), right.view(np.uint32))
    assert cam.right[0] == np.float32(0.99999994)  # not 1: v * (1/sqrt(dot)) rounding
    assert cam.resolution == (1280, 720)


@needs_glm
def test_basis_matches_glm_for_random_poses(pt):
    rng = np.random.default_rng(5)
    for _ in range(25):
        eye = rng.uniform(-20, 20, 3).astype(np.float32)
        focus = rng.uniform(-3, 3, 3).astype(np.float32)
        cam = pt.Camera(eye, focus, (64, 64))
        cam.update()
        g = _glm("basis", *eye, *focus)
        assert np.array_equal(cam.forward.view(np.uint32), g["forward"])
        assert np.array_equal(cam.up.view(np.uint32), g["up"])
        assert np.array_equal(cam.right.view(np.uint32), g["right"])


def _orbit_reference(eye, focus, dx, dy):
    """camera.cpp:38-76 in numpy float32 (glibc float math, as glm calls it)."""
    f = np.float32
    off = (eye - focus).astype(f)
    dist = f(np.sqrt(f(f(off[0] * off[0] + off[1] * off[1]) + off[2] * off[2])))
    d = (off / dist).astype(f)
    theta = f(math.acos(d[1])) if False else np.arccos(d[1], dtype=f)
    sin_t = np.sin(theta, dtype=f)
    gamma = np.arccos(f(d[0] / sin_t), dtype=f) if sin_t != 0 else np.arctan(f(d[2] / d[0]), dtype=f)
    if d[2] < 0:
        gamma = f(math.pi * 2.0 - float(gamma))
    theta = f(theta + f(dy))
    gamma = f(gamma + f(dx))
    theta = max(f(10.0 * math.pi / 180.0), min(theta, f(170.0 * math.pi / 180.0)))
    out = np.array([np.cos(gamma, dtype=f) * np.sin(theta, dtype=f), np.cos(theta, dtype=f),
                    np.sin(gamma, dtype=f) * np.sin(theta, dtype=f)], f)
    return (focus + out * dist).astype(f)


def test_orbit_zoom_pan(pt):
    eye0 = np.array([0, 1.5, 15], np.float32)
    focus = np.array([0, -0.25, 0], np.float32)
    cam = pt.Camera(eye0, focus, (32, 32))
    cam.update()
    dist0 = np.linalg.norm(eye0 - focus)

    cam.orbit(0.3, -0.2)
    e = cam.eye
    assert abs(np.linalg.norm(e - focus) - dist0) < 1e-4          # orbit keeps the distance
    assert np.allclose(e, _orbit_reference(eye0, focus, 0.3, -0.2), atol=2e-5)
    assert np.array_equal(cam.focus, focus)

    cam.orbit(0.0, -10.0)                                           # clamp at 10 degrees from the pole
    d = (cam.eye - focus) / np.linalg.norm(cam.eye - focus)
    assert abs(math.degrees(math.acos(d[1])) - 10.0) < 1e-3
    cam.orbit(0.0, 10.0)
    d = (cam.eye - focus) / np.linalg.norm(cam.eye - focus)
    assert abs(math.degrees(math.acos(d[1])) - 170.0) < 1e-3

    cam = pt.Camera(eye0, focus, (32, 32))
    cam.update()
    cam.zoom(0.5)
    assert abs(np.linalg.norm(cam.eye - focus) - 0.5 * dist0) < 1e-4
    cam.update()
    r, u = cam.right.copy(), cam.up.copy()
    e1, f1 = cam.eye.copy(), cam.focus.copy()
    cam.pan(0.25, -0.5)
    shift = r * np.float32(0.25) + u * np.float32(-0.5)
    assert np.allclose(cam.eye, e1 + shift, atol=1e-6) and np.allclose(cam.focus, f1 + shift, atol=1e-6)
    cam.move([1, 2, 3])
    assert np.allclose(cam.focus, f1 + shift + [1, 2, 3], atol=1e-5)
    cam.look_at([0, 0, 0], [1, 1, 1])
    assert cam.eye.tolist() == [1, 1, 1] and cam.focus.tolist() == [0, 0, 0]
    cam.look_at([0, 1, 0])
    assert cam.eye.tolist() == [1, 1, 1] and cam.focus.tolist() == [0, 1, 0]


def test_default_light_bits(pt):
    """Scene::Light(vec3(4), (0,1.95,0), (3.14/2,0,0), (1.25,1.25,1)) as the
    reference's glm builds it (main.cpp:122): 5 x vec4 = 80 bytes."""
    l = pt.make_light([4, 4, 4], [0, 1.95, 0], [3.14 / 2, 0, 0], [1.25, 1.25, 1])
    exp = [0x40800000, 0x40800000, 0x40800000, 0x00000000,
           0x3FA00000, 0x00000000, 0x00000000, 0x00000000,
           0x00000000, 0x3A82762A, 0x3F9FFFFD, 0x00000000,
           0x00000000, 0xBF7FFFFB, 0x3A50BD10, 0x00000000,
           0x00000000, 0x3FF9999A, 0x00000000, 0x3F800000]
    assert l.view(np.uint32).tolist() == exp
    s = pt.Scene()
    s.set_default_light()
    assert np.array_equal(s.arrays()["lights"][0].view(np.uint32), l.view(np.uint32))


@needs_glm
def test_light_matrix_matches_glm(pt):
    rng = np.random.default_rng(11)
    for _ in range(20):
        p, r, s = rng.uniform(-3, 3, 3), rng.uniform(-3.2, 3.2, 3), rng.uniform(0.2, 3, 3)
        p, r, s = (x.astype(np.float32) for x in (p, r, s))
        l = pt.make_light([1, 2, 3], p, r, s)
        g = _glm("light", *p, *r, *s)
        got = l.view(np.uint32)[4:].reshape(4, 4)
        for c in range(4):
            assert np.array_equal(got[c], g[f"col{c}"]), (c, p, r, s)


def test_set_lights_roundtrip(pt):
    s = pt.Scene()
    two = np.stack([pt.make_light([1, 1, 1], [0, 2, 0], [0, 0, 0], [1, 1, 1]),
                    pt.make_light([2, 3, 4], [1, 2, 3], [0.1, 0.2, 0.3], [2, 2, 1])])
    s.set_lights(two)
    assert np.array_equal(s.arrays()["lights"], two)
    s.set_lights(np.zeros((0, 20), np.float32))
    assert s.arrays()["lights"].shape == (0, 20)
