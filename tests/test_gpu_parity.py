"""Parity of the HIP path (through the C ABI) on a real MI355X.

Bit-exact gates: primary hit codes / t / ray directions against the reference's
own shader (llvmpipe fixtures) and against the oracle; BVH extend against the
oracle's linear scan on arbitrary rays.  Statistical gates: BSDF samples
within float tolerance of the oracle; converged images against the
reference's float-accumulated images."""
import os

import numpy as np
import pytest

from conftest import (ALL_SCENES, CONFIG_SCENES, LLVMPIPE, ROOT, box_filter, default_camera_axes, hit_fixture,
                      rel_mse, scene_arrays)

pytestmark = pytest.mark.gpu


def _scene(pt, name):
    return pt.Scene.from_arrays(**scene_arrays(name))


def _renderer(pt, name, w, h, **kw):
    return pt.Renderer(_scene(pt, name), pt.Camera(resolution=(w, h)), **kw)


# ------------------------------------------------------------------ primary hits

@pytest.mark.parametrize("use_bvh", [True, False], ids=["bvh", "scan"])
@pytest.mark.parametrize("name", ALL_SCENES)
def test_primary_hit_buffer_bit_exact_vs_reference_256(pt, name, use_bvh):
    fx = hit_fixture(name)
    r = _renderer(pt, name, 256, 256, use_bvh=use_bvh)
    ids, t, dirs = r.read_hit_ids(0, True, True)
    r.close()
    assert np.array_equal(ids, fx["ids"]), f"{int((ids != fx['ids']).sum())} hit codes differ"
    if "dirs_bits" in fx.files:
        assert np.array_equal(dirs.view(np.uint32), fx["dirs_bits"])
        geo = fx["ids"] >= 0
        assert np.array_equal(t.view(np.uint32)[geo], fx["t_bits"][geo])


@pytest.mark.parametrize("name", ALL_SCENES)
def test_render_loop_extend_differs_only_on_hits_at_infinity(pt, ob, name):
    """The render loop's traversal (without the parallel-plane pass of the
    readbacks) may differ from the reference's buffer only where the reference
    reports a hit at t = +inf (ray exactly parallel to a triangle's plane),
    and there it reports a miss."""
    fx = hit_fixture(name)["ids"]
    r = _renderer(pt, name, 256, 256)
    r.set_exact_readbacks(False)
    ids, t, _ = r.read_hit_ids(0, True)
    r.close()
    bad = ids != fx
    assert bad.sum() <= 2
    if bad.any():
        sc = ob.OracleScene(scene_arrays(name))
        _, ot, _ = ob.primary(sc, ob.make_camera(*default_camera_axes(), 256, 256), 0)
        assert np.all(np.isinf(ot[bad])) and np.all(ids[bad] == -1)


@pytest.mark.parametrize("name", ["Box", "Veggie"])
def test_primary_hit_buffer_bit_exact_vs_reference_1080p(pt, name):
    fx = hit_fixture(name, 1920, 1080)["ids"]
    r = _renderer(pt, name, 1920, 1080)
    ids, _, _ = r.read_hit_ids(0)
    r.close()
    assert np.array_equal(ids, fx)


@pytest.mark.parametrize("name", ["Box_Transmission", "Box_TransmissionSphere", "Box_Anisotropic", "AnisotropicTest"])
def test_primary_hit_buffer_bit_exact_vs_oracle_1080p(pt, ob, name):
    """The other benchmark scenes at the benchmark size, against the oracle (which the 256x256 and the
    Box / Veggie 1920x1080 fixtures pin to the reference): codes, t bits and direction bits."""
    sc = ob.OracleScene(scene_arrays(name))
    oid, ot, od = ob.primary(sc, ob.make_camera(*default_camera_axes(), 1920, 1080), 0)
    r = _renderer(pt, name, 1920, 1080)
    ids, t, dirs = r.read_hit_ids(0, True, True)
    r.close()
    assert np.array_equal(ids, oid)
    assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
    assert np.array_equal(dirs.view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("w,h,it", [(256, 256, 7), (192, 108, 3)])
def test_primary_other_iteration_and_aspect(pt, w, h, it):
    fx = hit_fixture("Box", w, h, it)
    r = _renderer(pt, "Box", w, h)
    ids, t, dirs = r.read_hit_ids(it, True, True)
    r.close()
    assert np.array_equal(ids, fx["ids"])
    assert np.array_equal(dirs.view(np.uint32), fx["dirs_bits"])
    geo = fx["ids"] >= 0
    assert np.array_equal(t.view(np.uint32)[geo], fx["t_bits"][geo])


@pytest.mark.parametrize("name,w,h,it", [("Box", 256, 256, 0), ("Veggie", 256, 256, 0), ("Box_Transmission", 256, 256, 0),
                                         ("Box_Anisotropic", 256, 256, 0), ("Box", 256, 256, 7), ("Box", 192, 108, 3)])
def test_primary_hit_normals_bit_exact_vs_reference(pt, name, w, h, it):
    """Intersection::normal as the reference's own intersect() leaves it (frag:434-460; fixture: a debug
    main() writing vec4(t, normal)): interpolated and not renormalised on surfaces, the light's normal
    on the light, zero where nothing is hit -- bit for bit on every pixel."""
    fx = hit_fixture(name, w, h, it)
    r = _renderer(pt, name, w, h)
    nrm = r.read_hit_normals(it)
    r.close()
    bad = (nrm.view(np.uint32) != fx["normal_bits"]).any(axis=2)
    assert not bad.any(), f"{int(bad.sum())} normals differ"
    assert (fx["ids"] <= -2).any() and np.abs(nrm[fx["ids"] <= -2]).max() > 0.9  # light pixels carry the light's normal


def test_primary_matches_oracle_at_odd_sizes(pt, ob):
    """Ragged sizes (not multiples of the wave or block size) and 1x1."""
    for name, w, h, it in [("Box", 1, 1, 0), ("Box", 63, 5, 2), ("Veggie", 257, 3, 9), ("TEST", 130, 70, 1)]:
        sc = ob.OracleScene(scene_arrays(name))
        oid, ot, od = ob.primary(sc, ob.make_camera(*default_camera_axes(), w, h), it)
        r = _renderer(pt, name, w, h)
        ids, t, dirs = r.read_hit_ids(it, True, True)
        r.close()
        assert np.array_equal(ids, oid)
        assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
        assert np.array_equal(dirs.view(np.uint32), od.view(np.uint32))


# ------------------------------------------------------------------ extend on arbitrary rays

def _secondary_like_rays(arrays, n, seed):
    """Origins on random triangles (offset like frag:897), random directions,
    plus axis-parallel and edge-grazing directions."""
    rng = np.random.default_rng(seed)
    tri = arrays["indices"][rng.integers(0, arrays["indices"].shape[0], n), :3].astype(np.int64)
    v = arrays["vertices"]
    b = rng.dirichlet([1, 1, 1], n).astype(np.float32)
    p = b[:, :1] * v[tri[:, 0]] + b[:, 1:2] * v[tri[:, 1]] + b[:, 2:] * v[tri[:, 2]]
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    k = n // 10
    d[:k] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * rng.choice([-1.0, 1.0], (k, 1)).astype(np.float32)
    # aim a tenth of the rays exactly at a vertex of another triangle (edge / vertex grazing)
    tgt = v[arrays["indices"][rng.integers(0, arrays["indices"].shape[0], k), 0].astype(np.int64)]
    g = tgt - p[k:2 * k]
    nrm = np.linalg.norm(g, axis=1, keepdims=True)
    d[k:2 * k] = np.where(nrm > 0, g / np.maximum(nrm, 1e-20), d[k:2 * k])
    o = (p + d * np.float32(1e-4)).astype(np.float32)
    return o, d.astype(np.float32)


@pytest.mark.parametrize("name", CONFIG_SCENES + ["TEST", "metal_rough_smooth", "cube"])
def test_bvh_extend_equals_linear_scan_of_oracle(pt, ob, name):
    a = scene_arrays(name)
    n = 200_000 if a["indices"].shape[0] < 600 else 60_000
    o, d = _secondary_like_rays(a, n, 3)
    osc = ob.OracleScene(a)
    ot, oid = ob.trace(osc, o, d)
    # Rays lying exactly in a triangle's plane give st = 0/0 in the reference;
    # from there on its result depends on scan order, not on distance.  Only
    # the synthetic in-plane rays (first two tenths) can be of that kind.
    nan_st = ob.poisoned(osc, o, d)
    assert not nan_st[2 * (n // 10):].any()
    ok = ~nan_st
    r = _renderer(pt, name, 64, 64)
    t, ids = r.trace_rays(o, d)
    r.close()
    bad = (ids != oid) & ok
    assert not bad.any(), f"{int(bad.sum())} of {int(ok.sum())} closest hits differ from the linear scan"
    assert np.array_equal(t.view(np.uint32)[ok], ot.view(np.uint32)[ok])


@pytest.mark.parametrize("profile", [0, 8, 32, 32 | 8])
def test_bvh_extend_on_instanced_lattice(pt, ob, profile):
    """38 890 triangles, tree depth > 16: a deeper traversal stack, and a node array too large for LDS
    (the kernel keeps the top of the tree there and reads the rest from global memory; of each stack the
    first 16 entries are in LDS and the rest in an overflow buffer -- with profile bit 3 only the first 4,
    so that nearly every ray goes through the overflow path).  A render must not depend on it either.
    Profile bit 5: the same tree as four-wide nodes (two levels per record, up to three deferred children per
    step): same hits, same image."""
    lat = _scene(pt, "Veggie").instance_lattice(10, 3)
    a = lat.arrays()
    assert lat.bvh_info()["stack_entries"] > 16
    o, d = _secondary_like_rays(a, 20_000, 8)
    eye = default_camera_axes()[0]
    tgt = a["vertices"][np.random.default_rng(2).integers(0, a["vertices"].shape[0], 20_000)]
    d2 = (tgt - eye).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o = np.concatenate([o, np.tile(eye, (20_000, 1))]).astype(np.float32)
    d = np.concatenate([d, d2]).astype(np.float32)
    osc = ob.OracleScene(a)
    ot, oid = ob.trace(osc, o, d)
    ok = ~ob.poisoned(osc, o, d)  # see test_bvh_extend_equals_linear_scan_of_oracle
    assert ok[4000:].all()
    r = pt.Renderer(lat, pt.Camera(resolution=(64, 64)), profile=profile)
    t, ids = r.trace_rays(o, d)
    st = r.stats()
    r.draw(3)
    img = r.read_sum()
    r.close()
    assert st["stack_entries"] > 16 and st["n_nodes"] * 64 > 64 * 1024
    assert np.array_equal(ids[ok], oid[ok])
    assert np.array_equal(t.view(np.uint32)[ok], ot.view(np.uint32)[ok])
    if profile:
        r = pt.Renderer(lat, pt.Camera(resolution=(64, 64)))
        r.draw(3)
        ref = r.read_sum()
        r.close()
        assert np.array_equal(img, ref) and np.isfinite(img).all()


def test_bvh_extend_on_the_ten_million_triangle_scene(pt, ob):
    """The BVH-stress configuration (Veggie instanced on a 19^3 lattice, 9 876 970 triangles): about a
    hundred rays through the 5.5 M-node tree (top of the tree in LDS, the rest from memory) against the
    oracle's linear scan over all triangles, bit for bit."""
    lat = _scene(pt, "Veggie").instance_lattice(10, 19)
    a = lat.arrays()
    assert a["indices"].shape[0] == 9_876_970
    o, d = _secondary_like_rays(a, 60, 21)
    eye = default_camera_axes()[0]
    tgt = a["vertices"][np.random.default_rng(5).integers(0, a["vertices"].shape[0], 36)]
    d2 = (tgt - eye).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o = np.concatenate([o, np.tile(eye, (36, 1))]).astype(np.float32)
    d = np.concatenate([d, d2]).astype(np.float32)
    osc = ob.OracleScene(a)
    ot, oid = ob.trace(osc, o, d)
    ok = ~ob.poisoned(osc, o, d)
    assert ok[12:].all()
    r = pt.Renderer(lat, pt.Camera(resolution=(64, 64)))
    t, ids = r.trace_rays(o, d)
    st = r.stats()
    r.close()
    # the same tree as four-wide nodes (profile bit 5): the same hits
    r = pt.Renderer(lat, pt.Camera(resolution=(64, 64)), profile=32)
    t4, ids4 = r.trace_rays(o, d)
    r.close()
    assert np.array_equal(ids4, ids) and np.array_equal(t4.view(np.uint32), t.view(np.uint32))
    assert st["n_triangles"] == 9_876_970 and st["bvh_depth"] <= st["stack_entries"]
    assert (oid[ok] >= 0).sum() > 40
    assert np.array_equal(ids[ok], oid[ok])
    assert np.array_equal(t.view(np.uint32)[ok], ot.view(np.uint32)[ok])


# ------------------------------------------------------------------ BSDF stage

def _bsdf_inputs(n, seed):
    rng = np.random.default_rng(seed)
    mats = np.zeros((n, 8), np.float32)
    mats[:, :3] = rng.uniform(0, 1, (n, 3))
    kind = rng.integers(0, 6, n)
    mats[:, 3] = np.where(kind == 0, 1.0, rng.uniform(0.02, 0.95, n))          # roughness (1 = diffuse)
    mats[:, 4] = np.where(kind == 1, 1.0, np.where(kind == 2, rng.uniform(0, 1, n), 0.0))  # metallic
    mats[:, 5] = rng.choice([1.1, 1.33, 1.5, 2.0], n)
    mats[:, 6] = np.where(kind == 1, rng.choice([0.0, 0.5, 1.0], n), 0.0)      # anisotropy
    mats[:, 7] = np.where(kind >= 4, rng.choice([0.5, 1.0], n), 0.0)           # transmission
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm *= rng.uniform(0.9, 1.0, (n, 1)).astype(np.float32)                     # interpolated normals are not unit
    wo = rng.normal(size=(n, 3)).astype(np.float32)
    wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    xi = rng.uniform(0, 1, (n, 2)).astype(np.float32)
    seeds = rng.integers(0, 2 ** 32, (n, 2), dtype=np.uint64).astype(np.uint32)
    return mats, nrm.astype(np.float32), wo, xi, seeds


def test_sample_surface_matches_oracle(pt, ob):
    """frag:841-858 per lobe: same lobe choice and RNG consumption (exact),
    direction / pdf / value within float tolerance.  The tolerance covers
    sinf/cosf/powf implementation differences and the few-ulp differences they
    seed in long expressions: rtol 2e-3 on pdf and f, 2e-4 absolute on wi."""
    n = 40_000
    mats, nrm, wo, xi, seeds = _bsdf_inputs(n, 21)
    r = _renderer(pt, "Box", 16, 16)
    wi, pdf, f, flags, so = r.sample_surface(mats, nrm, wo, xi, seeds)
    r.close()
    n_lobe_mismatch = n_bad = n_tight = n_checked = 0
    for i in range(n):
        owi, opdf, of, early, oseed = ob.sample_surface(mats[i], nrm[i], wo[i], xi[i], seeds[i])
        assert np.array_equal(oseed, so[i]), "RNG consumption differs"
        if bool(early) != bool(flags[i] & 1):
            n_lobe_mismatch += 1  # a threshold compare on a value that differs in the last ulp
            continue
        if early:
            continue
        n_checked += 1
        if not np.isfinite(opdf) or not np.all(np.isfinite(of)) or not np.all(np.isfinite(owi)):
            continue  # NaN frames are covered by test_sample_surface_degenerate_frames
        ok = (np.allclose(wi[i], owi, atol=2e-4, rtol=2e-3) and np.isclose(pdf[i], opdf, rtol=2e-3, atol=1e-6) and
              np.allclose(f[i], of, rtol=2e-3, atol=1e-6))
        n_bad += 0 if ok else 1
        # the same at a tenth of the tolerance: what the kernel's own operation order costs (pow5 by
        # multiplication, a reciprocal reused where the shader divides twice) shows up here
        tight = (np.allclose(wi[i], owi, atol=2e-5, rtol=2e-4) and np.isclose(pdf[i], opdf, rtol=2e-4, atol=1e-6) and
                 np.allclose(f[i], of, rtol=2e-4, atol=1e-6))
        n_tight += 0 if tight else 1
    print(f"BSDF samples outside rtol 2e-3: {n_bad} of {n_checked}; outside rtol 2e-4: {n_tight}; "
          f"lobe mismatches {n_lobe_mismatch}")
    assert n_checked > 0.8 * n
    assert n_lobe_mismatch <= n * 2e-4
    # Ill-conditioned samples (grazing microfacet normals, near-TIR refraction) amplify ulp differences.  The
    # library's shading code divides and takes square roots with the hardware's 1-ulp instructions, the oracle
    # with libm's correctly rounded ones; measured: 89 of 32 891 samples outside rtol 2e-3 and 116 outside 2e-4
    # (a build with correctly rounded operations throughout: 4 and 7).  The bounds: 0.4 % and 0.5 % of the samples.
    assert n_bad <= n * 4e-3, f"{n_bad} of {n_checked} BSDF samples outside tolerance"
    assert n_tight <= n * 5e-3, f"{n_tight} of {n_checked} BSDF samples outside a tenth of the tolerance"


def test_sample_surface_degenerate_frames(pt, ob):
    """Normal parallel to the fixed tangent (0,0,1) makes the metallic frame NaN
    in the reference (frag:169-174,816); grazing wo gives the early returns."""
    mats = np.array([[.8, .8, .8, .05, 1, 1.5, 0, 0]] * 4 + [[.8, .8, .8, .3, 0, 1.5, 0, 1]] * 2, np.float32)
    nrm = np.array([[0, 0, 1], [0, 0, -1], [0, 1, 0], [1, 0, 0], [0, 1, 0], [0, 1, 0]], np.float32)
    wo = np.array([[0, .6, .8], [0, .6, .8], [1, 0, 0], [0, 1, 0], [1, 0, 0], [0, -1, 0]], np.float32)
    xi = np.full((6, 2), 0.37, np.float32)
    seeds = np.tile(np.array([[12345, 678]], np.uint32), (6, 1))
    r = _renderer(pt, "Box", 16, 16)
    wi, pdf, f, flags, so = r.sample_surface(mats, nrm, wo, xi, seeds)
    r.close()
    for i in range(6):
        owi, opdf, of, early, oseed = ob.sample_surface(mats[i], nrm[i], wo[i], xi[i], seeds[i])
        assert np.array_equal(oseed, so[i])
        assert bool(early) == bool(flags[i] & 1)
        if not early:
            assert np.array_equal(np.isnan(owi), np.isnan(wi[i]))
            assert np.isnan(opdf) == np.isnan(pdf[i])
    assert np.isnan(wi[0]).all() and np.isnan(wi[1]).all()  # the z-facing wall quirk


# ------------------------------------------------------------------ converged images

IMAGE_CASES = [("Box", 8), ("Box", 5), ("Box_Anisotropic", 8), ("AnisotropicTest", 8), ("Box_Transmission", 8),
               ("Box_TransmissionSphere", 8), ("Veggie", 8)]


@pytest.mark.parametrize("name,depth", IMAGE_CASES)
def test_converged_image_against_reference(pt, name, depth):
    """256 spp at 256x256 against the reference shader's float-accumulated
    image (non-finite samples zeroed) at the same spp, depth and iteration
    indices.  Tolerances (relMSE = mean (I-R)^2/(R^2+0.01)):
      * per-pixel relMSE <= 2x the noise floor, the floor being the relMSE
        between two independent 256-spp renders of our own (iterations
        0..255 vs 256..511);
      * 16x16-box-filtered relMSE <= 1e-3;
      * every channel mean within 1 %."""
    path = os.path.join(LLVMPIPE, f"image_{name}_256x256_spp256_depth{depth}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    ref = np.load(path)["mean_rgb"].astype(np.float64)
    r = _renderer(pt, name, 256, 256, max_depth=depth)
    r.draw(256)
    s1 = r.read_sum()
    r.draw(256)
    s2 = r.read_sum() - s1
    st = r.stats()
    r.close()
    assert np.all(s1[..., 3] == 256)
    a, b = s1[..., :3] / 256.0, s2[..., :3] / 256.0
    floor = rel_mse(b, a)
    assert rel_mse(a, ref) <= 2.0 * floor + 1e-4, (rel_mse(a, ref), floor)
    assert rel_mse(box_filter(a), box_filter(ref)) <= 1e-3
    ma, mr = a.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mr) <= 0.01 * mr), (ma, mr)
    assert 3.0 < st["segments"] / st["paths"] < 5.0


def _reference_image(name, w, h, spp, depth):
    """A float image the reference's shader accumulated on llvmpipe (non-finite samples zeroed), stored as
    float16 with the values float16 does not hold to 1e-3 beside it in full precision."""
    path = os.path.join(LLVMPIPE, f"image_{name}_{w}x{h}_spp{spp}_depth{depth}.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    z = np.load(path)
    img = z["mean_rgb_f16"].astype(np.float32)
    img.ravel()[z["exact_index"]] = z["exact_value"]
    return img.astype(np.float64)


@pytest.mark.parametrize("name", ["Box", "Veggie", "Box_Transmission", "Box_TransmissionSphere", "Box_Anisotropic",
                                  "AnisotropicTest"])
def test_image_against_reference_at_1920x1080(pt, name):
    """The benchmark configurations themselves (the two headline scenes and the four scenes of the transmission
    and anisotropy configurations): 1920x1080, depth 8, through the renderer's default path (all bounces of a
    batch in one launch, except Veggie: separate kernels on two streams),
    against the reference shader's float image of the same iterations 0..15 (llvmpipe takes half a minute to
    six minutes per scene at 16 spp).  Tolerances (relMSE = mean (I-R)^2/(R^2+0.01)), the noise floor being the relMSE between
    two independent renders of our own (iterations 0..15 against 16..31):
      * per-pixel relMSE <= 2x the floor;
      * 16x16-box-filtered relMSE <= 2x the filtered floor, and <= 3 x 2e-4 x 256/spp (2e-4 is the filtered
        floor the reference shows against itself at 256 spp; variance goes with 1/spp);
      * every channel mean within 1 %.
    Both sides draw the same random numbers and make the same intersections, so the observed distances are
    far below the floor: they measure the paths that last-bit differences in sin / cos / pow sent elsewhere."""
    w, h, spp, depth = 1920, 1080, 16, 8
    ref = _reference_image(name, w, h, spp, depth)
    r = _renderer(pt, name, w, h, max_depth=depth)
    r.draw(spp)
    s1 = r.read_sum()
    r.draw(spp)
    s2 = r.read_sum() - s1
    r.close()
    assert np.all(s1[..., 3] == spp)
    a, b = s1[..., :3] / float(spp), s2[..., :3] / float(spp)
    floor, floor_f = rel_mse(b, a), rel_mse(box_filter(b), box_filter(a))
    got, got_f = rel_mse(a, ref), rel_mse(box_filter(a), box_filter(ref))
    print(f"{name} 1080p {spp} spp: per-pixel relMSE {got:.4g} (floor {floor:.4g}), 16x16-filtered {got_f:.4g} (floor {floor_f:.4g})")
    assert got <= 2.0 * floor + 1e-4, (got, floor)
    assert got_f <= 2.0 * floor_f + 1e-5, (got_f, floor_f)
    assert got_f <= 3.0 * 2.0e-4 * 256.0 / spp
    # the same paths, not merely the same statistics: observed 4e-7 .. 7e-6 on the six scenes
    assert got <= 1e-3 * floor, (got, floor)
    ma, mr = a.mean(axis=(0, 1)), ref.mean(axis=(0, 1))
    assert np.all(np.abs(ma - mr) <= 0.01 * mr), (ma, mr)


@pytest.mark.parametrize("name", ["Box_IOR", "Box_Metallic", "Box_MetallicColor", "Box_SmoothSphere", "TEST", "cube",
                                  "materialTest", "metal_rough_smooth", "shadowTest", "smoothCube", "sphere",
                                  "transmissionBox"])
def test_image_against_reference_on_the_other_assets(pt, name):
    """The bundled assets that are no benchmark configuration -- the metallic sweeps (the stochastic choice
    between the metal and the dielectric lobe, frag:846), index-of-refraction and smooth-shading scenes, the material
    tests: 256x256, depth 8, iterations 0..63 against the reference shader's float image of the same iterations.
    Same gates as at 1920x1080: per-pixel relMSE <= 2x the noise floor (two independent renders of our own) and
    far below it (the same paths: <= a hundredth of the floor), 16x16-filtered relMSE <= 2x its floor, channel
    means within 1 % (or, for a nearly black frame, within 1e-4 absolute).
    Box_SmoothSphere is the exception to "the same paths": its blue metal sphere has roughness 0.005, alpha =
    roughness^2 = 2.5e-5, where the shader's float arithmetic breaks down -- a visible-normal sample of a metal
    should weigh at most 1, single samples here weigh up to 733 in the reference's image and 248 in ours -- and
    what comes out depends on the platform's sin / cos / pow / reciprocal to the last bit (llvmpipe's polynomials
    there, libm-accurate functions here).  4.7 % of its pixels hold a sample that differs (2.1 % when our
    division and square root are correctly rounded instead of the hardware's 1 ulp).  Red and green, which the
    sphere does not reflect, agree like in every other scene; the blue channel's distribution differs (the same
    mean within the noise, 0.108 against 0.100 ... 0.127 over six sample sets of ours, but more of it in extreme
    samples there: clipped at 10 per pixel the reference's blue mean is 4 % below ours, outside our own spread
    of 1.5 %).  Gated for this scene: relMSE against the floors, red and green means within 1 %, blue within 8 %."""
    w, h, spp, depth = 256, 256, 64, 8
    ref = _reference_image(name, w, h, spp, depth)
    r = _renderer(pt, name, w, h, max_depth=depth)
    r.draw(spp)
    s1 = r.read_sum()
    r.draw(spp)
    s2 = r.read_sum() - s1
    r.close()
    assert np.all(s1[..., 3] == spp)
    a, b = s1[..., :3] / float(spp), s2[..., :3] / float(spp)
    floor, floor_f = rel_mse(b, a), rel_mse(box_filter(b), box_filter(a))
    got, got_f = rel_mse(a, ref), rel_mse(box_filter(a), box_filter(ref))
    print(f"{name} 256x256 {spp} spp: per-pixel relMSE {got:.4g} (floor {floor:.4g}), 16x16-filtered {got_f:.4g} (floor {floor_f:.4g})")
    assert got <= 2.0 * floor + 1e-4, (got, floor)
    if name != "Box_SmoothSphere":
        assert got <= 1e-2 * floor + 1e-9, (got, floor)
    assert got_f <= 2.0 * floor_f + 1e-5, (got_f, floor_f)
    close = (np.abs(a - ref) <= 1e-2 * np.abs(ref) + 1e-5).all(axis=2).mean()
    print(f"   pixels within 1 % of the reference's: {close:.4f}")
    assert close >= (0.94 if name == "Box_SmoothSphere" else 0.98), close
    # (means with every pixel clipped at 10: a single pixel of the smooth sphere's frame holds up to 733 in the
    # reference's image and 248 in ours -- a tenth of the channel's mean in one sample)
    ma, mr = np.minimum(a, 10.0).mean(axis=(0, 1)), np.minimum(ref, 10.0).mean(axis=(0, 1))
    tol = np.array([0.01, 0.01, 0.08 if name == "Box_SmoothSphere" else 0.01])
    assert np.all(np.abs(ma - mr) <= tol * mr + 1e-4), (ma, mr)


def test_faces_without_usemtl_scatter_what_the_reference_scatters(pt):
    """cube.obj has no usemtl: its faces carry material id 0xFFFFFFFF (scene.h:133) and the shader's material
    fetches (frag:105-126) go out of range.  On llvmpipe those fetches return zeros -- an all-zero material,
    through which no path carries anything on -- and the image is the directly visible light on black.
    The same 64 iterations at 128x128, depth 8, here: light pixels equal the reference's, everything else is zero
    on both sides."""
    w, h, spp, depth = 128, 128, 64, 8
    ref = _reference_image("cube", w, h, spp, depth)
    r = _renderer(pt, "cube", w, h, max_depth=depth)
    r.draw(spp)
    s = r.read_sum()
    st = r.stats()
    r.close()
    a = s[..., :3] / float(spp)
    assert ref.max() > 0 and (ref == 0).mean() > 0.9  # the fixture is what the docstring says it is
    assert np.array_equal(a == 0, ref == 0)
    assert np.allclose(a, ref, rtol=2e-3, atol=0)  # the fixture is stored as float16
    # (a zero roughness makes the specular lobe's density 0/0: those paths end as non-finite samples, which the
    # fixture's accumulation and this renderer both count as zero)
    assert st["paths"] == w * h * spp


@pytest.mark.parametrize("name", CONFIG_SCENES)
def test_short_render_agrees_with_the_oracle_pixel_by_pixel(pt, ob, name):
    """The same paths on both sides (the RNG is integer-exact, the intersections bit-exact): after 8 samples
    per pixel at depth 8 nearly every pixel of the sum agrees with the oracle's to float noise; the ones that
    do not are paths that a last-bit difference in the BSDF arithmetic sent elsewhere (measured: 0.35 % of
    Box's pixels, none in the other five scenes).  Channel means within 1e-4."""
    w, h, spp, depth = 192, 108, 8, 8
    r = _renderer(pt, name, w, h, max_depth=depth)
    r.draw(spp)
    s = r.read_sum()
    r.close()
    osum, _ = ob.render(ob.OracleScene(scene_arrays(name)), ob.make_camera(*default_camera_axes(), w, h), 0, spp, depth,
                        min(16, os.cpu_count() or 1))
    assert np.all(s[..., 3] == spp) and np.all(osum[..., 3] == spp)
    close = np.isclose(s[..., :3], osum[..., :3], rtol=1e-3, atol=1e-4).all(axis=2).mean()
    m, om = s[..., :3].mean(axis=(0, 1)), osum[..., :3].mean(axis=(0, 1))
    print(f"{name}: {close:.4f} of the pixels agree; channel means {m} vs {om}")
    # measured: Box 0.9965 (1-ulp division and square root in the shading code against libm's in the oracle; 0.9993
    # with correctly rounded ones), the other five scenes 1.0
    assert close >= 0.995, f"only {close:.4f} of the pixels agree with the oracle"
    assert np.all(np.abs(m - om) <= 1e-4 * om + 1e-7)


def test_image_equivalence_bvh_vs_scan(pt):
    """Same closest hits => the two extend kernels give identical images."""
    out = []
    for use_bvh in (True, False):
        r = _renderer(pt, "Box_Transmission", 128, 128, max_depth=8, use_bvh=use_bvh)
        r.draw(16)
        out.append(r.read_sum())
        r.close()
    assert np.array_equal(out[0], out[1])


# ------------------------------------------------------------------ lights

def _box_with_lights(pt, lights):
    a = dict(scene_arrays("Box"))
    a["lights"] = np.asarray(lights, np.float32).reshape(-1, 20)
    return a


def test_two_lights_and_no_light_use_the_general_light_test(pt, ob):
    """With one light the distance to it travels with the ray and limits the
    traversal; any other number of lights goes through the reference's loop over
    lights (frag:418-430) when the ray finishes.  Both must give the reference's
    hit codes (light k = -(2 + k)) and identical images from the BVH kernel and
    the linear-scan kernel."""
    first = scene_arrays("Box")["lights"].reshape(-1, 20)[0]
    second = pt.make_light((2.0, 1.0, 0.5), (-0.6, 0.4, 0.9), (0.3, 0.8, 0.0), (0.7, 0.5, 1.0))
    for lights in ([first, second], [second, first], np.zeros((0, 20), np.float32)):
        a = _box_with_lights(pt, lights)
        sc = ob.OracleScene(a)
        oid, ot, _ = ob.primary(sc, ob.make_camera(*default_camera_axes(), 160, 120), 3)
        sums = []
        for use_bvh in (True, False):
            r = pt.Renderer(pt.Scene.from_arrays(**a), pt.Camera(resolution=(160, 120)), max_depth=6, use_bvh=use_bvh)
            ids, t, _ = r.read_hit_ids(3, True)
            assert np.array_equal(ids, oid)
            assert np.array_equal(t.view(np.uint32), ot.view(np.uint32))
            r.draw(8)
            sums.append(r.read_sum())
            r.close()
        assert np.array_equal(sums[0], sums[1])
        n_lights = len(lights)
        if n_lights == 2:
            assert (oid == -2).any() and (oid == -3).any()
            assert sums[0][..., :3].max() > 0
        else:
            assert not sums[0][..., :3].any()


def test_two_overlapping_lights_against_the_reference(pt):
    """Two rectangles that overlap on screen, in both orders of the light array, against the reference's hit codes
    (the first light of the array that a ray meets wins, frag:418-430: in 77 pixels the winner changes with the
    order), and the image of one order against the reference shader's float image of the same iterations."""
    path = os.path.join(LLVMPIPE, "two_lights_Box_256x256.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    z = np.load(path)
    ab = z["lights_ab"]
    for name, lights in (("ab", ab), ("ba", ab[::-1])):
        a = _box_with_lights(pt, list(lights))
        r = pt.Renderer(pt.Scene.from_arrays(**a), pt.Camera(resolution=(256, 256)), max_depth=8)
        ids, _, _ = r.read_hit_ids(0)
        assert np.array_equal(ids, z["ids_" + name]), name
        if name == "ba":
            spp = int(z["spp"])
            r.draw(spp)
            s1 = r.read_sum()
            r.draw(spp)
            s2 = r.read_sum() - s1
            img, other, ref = s1[..., :3] / spp, s2[..., :3] / spp, z["mean_rgb_ba"].astype(np.float64)
            floor, got = rel_mse(other, img), rel_mse(img, ref)
            floor_f, got_f = rel_mse(box_filter(other), box_filter(img)), rel_mse(box_filter(img), box_filter(ref))
            print(f"two lights: per-pixel relMSE {got:.4g} (floor {floor:.4g}), filtered {got_f:.4g} (floor {floor_f:.4g})")
            assert got <= 1e-2 * floor and got_f <= 1e-2 * floor_f
            assert np.all(np.abs(img.mean(axis=(0, 1)) - ref.mean(axis=(0, 1))) <= 0.01 * ref.mean(axis=(0, 1)))
        r.close()


def test_one_light_limit_equals_general_light_test(pt):
    """The same one-light scene rendered through the BVH kernel (light distance
    as the traversal limit) and through the scan kernel (light tested when the
    ray finishes): identical sums, at a size whose queue mapping is by tiles and
    at one whose mapping is by strips with a ragged tail."""
    for w, h in [(128, 64), (131, 67)]:
        out = []
        for use_bvh in (True, False):
            r = _renderer(pt, "Veggie", w, h, max_depth=8, use_bvh=use_bvh)
            r.draw(6)
            out.append(r.read_sum())
            r.close()
        assert np.array_equal(out[0], out[1])
        assert np.all(out[0][..., 3] == 6)


# ------------------------------------------------------------------ the reference's own 8-bit accumulator

def test_rgba8_running_average_against_the_reference_texture(pt, ob):
    """emulate_rgba8 keeps the reference's as-shipped accumulator beside the float sum: an 8-bit texel
    updated as (accum*it + sample)/(it+1) and re-quantised every iteration (frag:900-904,
    renderer.cpp:21,25), NaN samples included.  Against the texture the reference shader itself left
    after 64 iterations at depth 10 (llvmpipe fixture): the BSDF's sin/cos differ in the last bits
    between the two implementations, so texels may differ by a rounding step -- required: >= 90 %
    identical, >= 99.9 % within 2 of 255, channel means within 1 %.  Against the oracle's emulation
    on a small frame: >= 99.5 % identical, none further than 2."""
    ref = np.load(os.path.join(LLVMPIPE, "rgba8_Box_256x256_spp64_depth10.npz"))["rgba8"]
    r = _renderer(pt, "Box", 256, 256, max_depth=10, emulate_rgba8=True)
    r.draw(64)
    tex = r.read_rgba8()
    s = r.read_sum()
    st = r.stats()
    r.close()
    assert tex.shape == ref.shape and np.all(tex[..., 3] == 255)
    d = np.abs(tex[..., :3].astype(int) - ref[..., :3].astype(int))
    assert (d == 0).mean() >= 0.90 and (d <= 2).mean() >= 0.999, ((d == 0).mean(), (d <= 2).mean())
    mt, mr = tex[..., :3].mean(axis=(0, 1)), ref[..., :3].mean(axis=(0, 1))
    assert np.all(np.abs(mt - mr) <= 0.01 * mr), (mt, mr)
    assert np.all(s[..., 3] == 64) and np.isfinite(s).all()
    assert 0 < st["nan_samples"] < 200  # the fixed-tangent frame of frag:171 goes NaN on z-facing walls

    a = scene_arrays("Box")
    acc = ob.render_rgba8(ob.OracleScene(a), ob.make_camera(*default_camera_axes(), 64, 48), 0, 24, 10,
                          nthreads=os.cpu_count() or 1)
    r = pt.Renderer(pt.Scene.from_arrays(**a), pt.Camera(resolution=(64, 48)), max_depth=10, emulate_rgba8=True,
                    iterations_per_batch=5)
    r.draw(24)
    small = r.read_rgba8()
    with pytest.raises(pt.PtError):
        r.set_partition(0, 2)  # a running average is not a sum
    with pytest.raises(pt.PtError):
        r.save_state("/tmp/never_written.bin")
    r.close()
    d = np.abs(small.astype(int) - acc.astype(int))
    assert (d == 0).mean() >= 0.995 and d.max() <= 2

    plain = _renderer(pt, "Box", 32, 32)
    with pytest.raises(pt.PtError):
        plain.read_rgba8()
    plain.close()
