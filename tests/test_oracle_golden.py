"""Pins the oracle (oracle/pt_oracle.c) to the reference itself: the golden
fixtures under tests/golden/llvmpipe/ are what the reference's own fragment
shader produced on Mesa llvmpipe (tests/golden/make_llvmpipe_fixtures.py).

Bit-exact: primary-ray directions, hit t, hit codes (rng + raycast + triangle
test + light test).  Statistical: converged images (sin/cos/pow precision is
driver-defined in GLSL, so paths diverge after the first bounce)."""
import os

import numpy as np
import pytest

from conftest import (ALL_SCENES, LLVMPIPE, box_filter, default_camera_axes, hit_fixture, image_fixture, rel_mse,
                      scene_arrays)


def _camera(ob, w, h):
    return ob.make_camera(*default_camera_axes(), w, h)


@pytest.mark.parametrize("name", ALL_SCENES)
def test_primary_hit_codes_256(ob, name):
    fx = hit_fixture(name)
    sc = ob.OracleScene(scene_arrays(name))
    ids, t, dirs = ob.primary(sc, _camera(ob, 256, 256), 0)
    assert np.array_equal(ids, fx["ids"]), f"{int((ids != fx['ids']).sum())} hit codes differ from the reference"
    if "dirs_bits" in fx.files:
        assert np.array_equal(dirs.view(np.uint32), fx["dirs_bits"])
        geo = fx["ids"] >= 0
        assert np.array_equal(t.view(np.uint32)[geo], fx["t_bits"][geo])


@pytest.mark.parametrize("w,h,it", [(256, 256, 7), (192, 108, 3)])
def test_primary_other_iteration_and_aspect(ob, w, h, it):
    fx = hit_fixture("Box", w, h, it)
    sc = ob.OracleScene(scene_arrays("Box"))
    ids, t, dirs = ob.primary(sc, _camera(ob, w, h), it)
    assert np.array_equal(ids, fx["ids"])
    assert np.array_equal(dirs.view(np.uint32), fx["dirs_bits"])
    geo = fx["ids"] >= 0
    assert np.array_equal(t.view(np.uint32)[geo], fx["t_bits"][geo])


@pytest.mark.parametrize("name,stride", [("Box", 1), ("Veggie", 12)])
def test_primary_hit_codes_1080p(ob, name, stride):
    """Benchmark resolution; Veggie on every 12th row to bound CPU time."""
    fx = hit_fixture(name, 1920, 1080)["ids"]
    sc = ob.OracleScene(scene_arrays(name))
    cam = _camera(ob, 1920, 1080)
    if stride == 1:
        ids, _, _ = ob.primary(sc, cam, 0, want_t=False, want_dirs=False)
        assert np.array_equal(ids, fx)
    else:
        _, _, dirs = ob.primary_dirs_only(sc, cam, 0)
        rows = np.arange(0, 1080, stride)
        d = dirs[rows].reshape(-1, 3)
        o = np.tile(default_camera_axes()[0], (d.shape[0], 1))
        _, ids = ob.trace(sc, o, d)
        assert np.array_equal(ids.reshape(len(rows), 1920), fx[rows])


def test_surface_normal_of_primary_hit(ob):
    """Interpolated, not renormalised normal (frag:434-450) against the dump of
    the reference's own intersect()."""
    fx = hit_fixture("Box")
    sc = ob.OracleScene(scene_arrays("Box"))
    a = sc.a
    ids, t, dirs = ob.primary(sc, _camera(ob, 256, 256), 0)
    geo = ids >= 0
    eye = default_camera_axes()[0]
    # restate frag:439-450 in float32 numpy on the oracle's t
    tri = a["indices"][ids[geo]].astype(np.int64)
    v0, v1, v2 = (a["vertices"][tri[:, k]] for k in range(3))
    p = (eye + dirs[geo] * t[geo][:, None]).astype(np.float32)

    def dot(x, y):
        return (x[:, 0] * y[:, 0] + (x[:, 1] * y[:, 1] + x[:, 2] * y[:, 2])).astype(np.float32)

    e0, e1, ep = v1 - v0, v2 - v0, p - v0
    e00, e01, e11, e0p, e1p = dot(e0, e0), dot(e0, e1), dot(e1, e1), dot(e0, ep), dot(e1, ep)
    inv = (np.float32(1) / (e00 * e11 - e01 * e01)).astype(np.float32)
    v = ((e11 * e0p - e01 * e1p) * inv).astype(np.float32)
    w = ((e00 * e1p - e01 * e0p) * inv).astype(np.float32)
    u = (np.float32(1) - v - w).astype(np.float32)
    n = a["vertex_data"].reshape(-1, 3, 6)[tri[:, 3], :, :3]
    nrm = ((u[:, None] * n[:, 0] + v[:, None] * n[:, 1]) + w[:, None] * n[:, 2]).astype(np.float32)
    ref = fx["normal_bits"][geo]
    assert np.array_equal(nrm.view(np.uint32), ref)  # bit for bit: one rounding per operation, sums left to right


def test_rng_matches_uint32_definition(ob):
    """frag:297-304 restated with numpy uint32 wrap-around."""
    K = np.uint32(1103515245)
    rng = np.random.default_rng(1)
    for seed in [np.array([0, 0], np.uint32), np.array([1, 2], np.uint32), np.array([0xFFFFFFFF, 7], np.uint32),
                 *rng.integers(0, 2 ** 32, size=(16, 2), dtype=np.uint64).astype(np.uint32)]:
        vals, end = ob.rng_stream(seed, 8)
        s = seed.copy()
        with np.errstate(over="ignore"):
            for k in range(8):
                s = s + np.uint32(1)
                q = K * ((s >> np.uint32(1)) ^ s[::-1])
                n = K * (q[0] ^ (q[1] >> np.uint32(3)))
                assert vals[k] == np.float32(n) * np.float32(2.0 ** -32)
        assert np.array_equal(end, s)


def test_rng_known_defects(ob):
    """Zero seeds on the first column / row (frag:866) make pixel (0,0) start
    from state (0,0) in every iteration; the stream can return exactly 1.0."""
    v, _ = ob.rng_stream(np.array([0, 0], np.uint32), 2)
    w, _ = ob.rng_stream(np.array([0, 0], np.uint32), 2)
    assert np.array_equal(v, w)
    assert np.float32(0xFFFFFF80) * np.float32(2.0 ** -32) == np.float32(1.0)


def test_light_inverse_reproduces_light_mask(ob):
    fx = hit_fixture("Box")["ids"]
    assert int((fx == -2).sum()) == 123  # light pixels the reference sees at 256x256
    lights = scene_arrays("Box")["lights"]
    inv, ok = ob.light_inverse(lights[0, 4:])
    assert ok
    m = lights[0, 4:].reshape(4, 4).T.astype(np.float64)
    assert np.allclose(inv.reshape(4, 4).T.astype(np.float64) @ m, np.eye(4), atol=1e-6)


def test_two_overlapping_lights_in_both_array_orders(ob):
    """The shader's loop over lights (frag:418-430) sets t = 0 at the first light it accepts, so the light that a
    ray is reported to hit is the first one of the ARRAY it meets, not the nearest.  Fixture: the reference's
    hit codes for Box with two rectangles that overlap on screen, in both orders of the array (in 77 pixels the
    winner changes with the order)."""
    path = os.path.join(LLVMPIPE, "two_lights_Box_256x256.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    z = np.load(path)
    ab = z["lights_ab"]
    assert int(((z["ids_ab"] == -2) & (z["ids_ba"] == -2)).sum()) == 77
    for name, lights in (("ab", ab), ("ba", ab[::-1])):
        a = dict(scene_arrays("Box"))
        a["lights"] = np.ascontiguousarray(lights, np.float32)
        ids, _, _ = ob.primary(ob.OracleScene(a), _camera(ob, 256, 256), 0)
        assert np.array_equal(ids, z["ids_" + name]), name


@pytest.mark.parametrize("name,depth,spp", [("Box", 8, 24), ("Box", 5, 24), ("Box_Anisotropic", 8, 24),
                                            ("AnisotropicTest", 8, 24), ("Box_Transmission", 8, 12)])
def test_image_statistics_against_reference(ob, name, depth, spp):
    """Oracle at `spp` samples vs the reference's 256-spp float image: channel
    means within 3 % and the 16x16-filtered relMSE below what `spp` samples
    of this estimator leave (floor 2e-4 at 256 spp, scaling 1/spp)."""
    path = os.path.join(LLVMPIPE, f"image_{name}_256x256_spp256_depth{depth}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    ref = np.load(path)["mean_rgb"]
    sc = ob.OracleScene(scene_arrays(name))
    s, st = ob.render(sc, _camera(ob, 256, 256), 0, spp, depth, nthreads=os.cpu_count() or 1)
    img = s[..., :3] / spp
    assert np.all(s[..., 3] == spp)
    m_img, m_ref = img.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
    assert np.all(np.abs(m_img - m_ref) <= 0.03 * m_ref), (m_img, m_ref)
    floor = 2.0e-4 * 256.0 / spp
    assert rel_mse(box_filter(img), box_filter(ref)) <= 3.0 * floor


def test_rgba8_feedback_accumulator_matches_reference(ob):
    """The as-shipped accumulator (renderer.cpp:21, frag:900-904): 8-bit running
    average.  Mean level within 3 % of the reference's own RGBA8 texture."""
    path = os.path.join(LLVMPIPE, "rgba8_Box_256x256_spp64_depth10.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    ref = np.load(path)["rgba8"]
    sc = ob.OracleScene(scene_arrays("Box"))
    acc = ob.render_rgba8(sc, _camera(ob, 256, 256), 0, 16, 10, nthreads=os.cpu_count() or 1)
    # after 16 iterations the 8-bit average has not drifted yet: compare with a float render
    s, _ = ob.render(sc, _camera(ob, 256, 256), 0, 16, 10, nthreads=os.cpu_count() or 1)
    f = np.clip(s[..., :3] / 16, 0, 1).mean(axis=(0, 1))
    a = acc[..., :3].astype(np.float64).mean(axis=(0, 1)) / 255.0
    assert np.all(np.abs(a - f) <= 0.08 * f + 0.004)
    assert np.all(acc[..., 3] == 255)
    assert ref.shape == acc.shape


def test_tonemap_against_what_the_reference_s_post_pass_drew(ob):
    """post.frag.glsl:9-18 run by llvmpipe on a 64-spp float image of Box.obj into an RGBA8 target
    (fixture), against oracle_tonemap on the same float image quantised the way GL stores UNORM8:
    identical but for texels whose value sits on a rounding boundary (pow differs in the last bits)."""
    fx = np.load(os.path.join(LLVMPIPE, "post_Box_256x256_spp64_depth8.npz"))
    mine = np.floor(np.clip(ob.tonemap(fx["mean_rgb"]), 0, 1) * 255 + 0.5).astype(int)
    d = np.abs(mine - fx["post_rgba8"][..., :3].astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.999, (d.max(), (d == 0).mean())
    assert np.all(fx["post_rgba8"][..., 3] == 255)


def test_tonemap(ob):
    x = np.array([[0.0, 0.5, 1.0], [4.0, 0.01, 100.0]], np.float32)
    y = ob.tonemap(x)
    exp = (x / (1 + x)) ** (1 / 2.2)
    assert np.allclose(y, exp, rtol=1e-6)
