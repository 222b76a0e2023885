"""Renderer semantics on the GPU (renderer.h:32-38 surface): progressive
accumulation, reset / resize / camera changes, batch independence,
determinism, the sample-space partition used for multi-GPU runs, and
size-independent properties at the benchmark's full size."""
import os

import numpy as np
import pytest

from conftest import hit_fixture, scene_arrays

pytestmark = pytest.mark.gpu


def _renderer(pt, name, w, h, **kw):
    return pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), pt.Camera(resolution=(w, h)), **kw)


def test_progressive_accumulation_is_independent_of_call_and_batch_split(pt):
    sums = []
    for batch, calls in [(0, [12]), (1, [12]), (5, [7, 5]), (3, [1] * 12)]:
        r = _renderer(pt, "Box", 96, 64, max_depth=6, iterations_per_batch=batch)
        for c in calls:
            r.draw(c)
        sums.append(r.read_sum())
        assert r.stats()["iterations"] == 12
        r.close()
    for s in sums[1:]:
        assert np.array_equal(s, sums[0])
    assert np.all(sums[0][..., 3] == 12)


def test_two_streams_render_the_image_of_one(pt):
    # batches alternate between two sets of buffers on two streams; the accumulate kernels are ordered,
    # so the sum is the same bit for bit as with every kernel in sequence on one stream
    out = []
    for streams, calls in [(1, [5] * 6), (2, [5] * 6), (2, [30]), (2, [1] * 30)]:
        r = _renderer(pt, "Box_Transmission", 160, 96, max_depth=7, iterations_per_batch=5, batch_streams=streams)
        for c in calls:
            r.draw(c)
        out.append((r.read_sum(), r.stats()))
        r.close()
    for s, st in out[1:]:
        assert np.array_equal(s, out[0][0])
        assert st["segments"] == out[0][1]["segments"] and st["hits_shaded"] == out[0][1]["hits_shaded"]
    # a reset or a camera change between batches of different streams is ordered against both
    r = _renderer(pt, "Box", 128, 64, max_depth=5, iterations_per_batch=2)
    r.draw(6)
    r.reset()
    r.draw(3)
    a = r.read_sum()
    r.close()
    r = _renderer(pt, "Box", 128, 64, max_depth=5, iterations_per_batch=2, batch_streams=1)
    r.draw(3)
    b = r.read_sum()
    r.close()
    assert np.array_equal(a, b) and np.all(a[..., 3] == 3)


@pytest.mark.parametrize("name,lattice", [("Box", 0), ("Veggie", 0), ("Veggie", 3), ("Box_Transmission", 0),
                                          ("Box_TransmissionSphere", 0)])
def test_all_bounces_in_one_launch_render_what_separate_launches_render(pt, name, lattice):
    """A small batch runs all its bounces in one launch (a wave carries its regions of the queues through
    traversal and shading of every bounce); profile bit 4 asks for one launch per kernel and bounce
    instead.  Same image, same counts -- for a scene that lives in LDS in small blocks (Box), in two
    512-thread blocks per CU (Box_Transmission), in one large block with all its triangles
    (Box_TransmissionSphere) or with the hottest ones (Veggie), and one whose tree is read from memory
    (38 890 triangles) -- and the same again as a batch too large for the one launch."""
    scene = pt.Scene.from_arrays(**scene_arrays(name))
    if lattice:
        scene = scene.instance_lattice(10, lattice)
    out = []
    for profile, width, calls in [(0, 256, [3, 3, 2]), (16, 256, [3, 3, 2]), (1, 256, [8])]:
        r = pt.Renderer(scene, pt.Camera(resolution=(width, 192)), max_depth=7, profile=profile)
        for c in calls:
            r.draw(c)
        out.append((r.read_sum(), r.stats()))
        r.close()
    for s, st in out[1:]:
        assert np.array_equal(s, out[0][0])
        assert st["segments"] == out[0][1]["segments"] and st["hits_shaded"] == out[0][1]["hits_shaded"]
    assert np.isfinite(out[0][0]).all() and np.all(out[0][0][..., 3] == 8)
    # 1080p: batches of 4 samples per pixel (8.3 M paths) go in one launch for every scene, batches of 32 only
    # for a scene that lives in LDS with all its triangles; kernel by kernel they give the same image
    if not lattice:
        big = []
        for profile, calls in ((0, [4] * 8), (1, [32]), (17, [32])):
            r = pt.Renderer(scene, pt.Camera(resolution=(1920, 1080)), max_depth=5, profile=profile)
            for c in calls:
                r.draw(c)
            big.append((r.read_sum(), r.stats()))
            r.close()
        assert np.array_equal(big[0][0], big[1][0]) and np.array_equal(big[0][0], big[2][0])
        assert big[2][1]["launches_bounces"] == 0 and big[2][1]["launches_extend"] == 5
        assert (big[1][1]["launches_bounces"] == 0) == (name == "Veggie")


def test_deterministic_across_runs(pt):
    a = []
    for _ in range(2):
        r = _renderer(pt, "Box_Transmission", 128, 96, max_depth=8)
        r.draw(9)
        a.append(r.read_sum())
        r.close()
    assert np.array_equal(a[0], a[1])


def test_reset_resize_and_camera_change(pt):
    r = _renderer(pt, "Box", 80, 60, max_depth=5)
    r.draw(4)
    first = r.read_sum()
    r.reset()  # renderer.cpp:152-158: iteration counter back to 0
    assert r.stats()["iterations"] == 0
    assert not r.read_sum().any()
    r.draw(4)
    assert np.array_equal(r.read_sum(), first)

    r.resize(40, 30)  # renderer.cpp:160-187
    r.draw(2)
    small = r.read_sum()
    assert small.shape == (30, 40, 4) and np.all(small[..., 3] == 2)
    ids, _, _ = r.read_hit_ids(0)
    assert ids.shape == (30, 40)

    r.resize(80, 60)
    r.camera.orbit(0.4, 0.1)  # main.cpp:92-96: orbit then updateCamera -> reset
    r.update_camera()
    assert r.stats()["iterations"] == 0
    r.draw(4)
    moved = r.read_sum()
    assert not np.array_equal(moved, first)
    r.camera = type(r.camera)(resolution=(80, 60))
    r.update_camera()
    r.draw(4)
    assert np.array_equal(r.read_sum(), first)
    r.close()


def test_mean_and_tonemap_readbacks(pt, ob):
    r = _renderer(pt, "Box", 64, 48, max_depth=8)
    r.draw(32)
    s = r.read_sum()
    m = r.read_radiance()
    assert np.allclose(m[..., :3], s[..., :3] / 32.0, rtol=1e-6)
    tm = r.read_tonemapped()
    r.close()
    assert tm.shape == (48, 64, 3) and tm.dtype == np.uint8
    exp = ob.tonemap(m[..., :3])[::-1]  # file order: top row first
    exp8 = np.clip(exp * 255.0 + 0.5, 0, 255).astype(np.uint8)
    assert np.abs(tm.astype(int) - exp8.astype(int)).max() <= 1
    assert tm.max() > 100  # the light is visible


def test_tonemapped_readback_against_what_the_reference_s_post_pass_drew(pt, tmp_path):
    """post.frag.glsl:9-18 as llvmpipe ran it on a 64-spp float image of Box.obj (fixture: that image and
    the RGBA8 texels the post pass drew from it).  The float image is put into the renderer through a
    checkpoint file (header written by the renderer itself, payload replaced: sum = the image, count = 1);
    pt_read_tonemapped must then give the reference's texels to within one step of 255 (pow differs in
    the last bits), and nearly all of them exactly."""
    from conftest import LLVMPIPE
    fx = np.load(os.path.join(LLVMPIPE, "post_Box_256x256_spp64_depth8.npz"))
    r = _renderer(pt, "Box", 256, 256, max_depth=8)
    path = str(tmp_path / "state.bin")
    r.save_state(path)
    raw = bytearray(open(path, "rb").read())
    payload = np.concatenate([fx["mean_rgb"].astype(np.float32), np.ones((256, 256, 1), np.float32)], axis=2)
    assert len(raw) > payload.nbytes
    raw[len(raw) - payload.nbytes:] = payload.tobytes()
    open(path, "wb").write(bytes(raw))
    r.load_state(path)
    assert np.array_equal(r.read_radiance()[..., :3], fx["mean_rgb"])
    out = r.read_tonemapped()
    r.close()
    ref = fx["post_rgba8"][::-1, :, :3]  # the texture's row 0 is the bottom row, the readback's the top row
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.999, (d.max(), (d == 0).mean())


def test_partition_union_equals_single_renderer(pt):
    """Rank g of G renders iterations g, g+G, ...: the union is the single-GPU
    sample set, so the summed buffers agree up to fp32 addition order."""
    spp, world = 12, 3
    single = _renderer(pt, "Box", 96, 64, max_depth=8)
    single.draw(spp)
    ref = single.read_sum()
    single.close()
    total = np.zeros_like(ref)
    for rank in range(world):
        r = _renderer(pt, "Box", 96, 64, max_depth=8)
        r.set_partition(rank, world)
        r.draw(spp // world)
        total += r.read_sum()
        r.close()
    assert np.all(total[..., 3] == spp)
    assert np.allclose(total[..., :3], ref[..., :3], rtol=1e-5, atol=1e-6)

    # a rank's samples are exactly the corresponding iterations of the single run
    r = _renderer(pt, "Box", 96, 64, max_depth=8)
    r.set_partition(1, 3)
    ids_rank, _, _ = r.read_hit_ids(4)  # read_hit_ids takes the global iteration index
    r.close()
    s = _renderer(pt, "Box", 96, 64, max_depth=8)
    ids_single, _, _ = s.read_hit_ids(4)
    s.close()
    assert np.array_equal(ids_rank, ids_single)


def test_stats_accounting(pt):
    r = _renderer(pt, "Box", 128, 128, max_depth=8, profile=3)
    r.draw(8)
    st = r.stats()
    r.close()
    assert st["paths"] == 128 * 128 * 8 and st["iterations"] == 8
    assert st["paths"] <= st["segments"] <= 8 * st["paths"]
    # per batch: max_depth extend launches, one shade launch fewer (the last segment needs no BSDF)
    assert st["launches_extend"] == st["launches_shade"] + st["launches_raygen"] and st["launches_extend"] >= 8
    assert st["ms_extend"] > 0 and st["ms_shade"] > 0
    assert st["nodes_visited"] > st["segments"] and st["tris_tested"] > 0
    assert st["n_triangles"] == 155 and st["bvh_depth"] <= st["stack_entries"]


def test_depth_one_sees_only_the_light(pt):
    """max_depth 1: a pixel is lit iff its primary ray hits the light."""
    r = _renderer(pt, "Box", 256, 256, max_depth=1)
    r.draw(1)
    s = r.read_sum()
    ids, _, _ = r.read_hit_ids(0)
    r.close()
    lit = s[..., :3].sum(axis=2) > 0
    assert np.array_equal(lit, ids == -2)
    assert np.all(s[..., :3][lit] == 4.0)  # radiance of the default light, throughput 1


def test_scene_without_materials_or_lights(pt):
    """Faces without usemtl carry material 0xFFFFFFFF (scene.h:133); no light
    means a black image, not a crash."""
    a = scene_arrays("cube")
    s = pt.Scene.from_arrays(a["vertices"], a["indices"], a["vertex_data"], np.zeros((0, 20), np.float32),
                             a["materials"], a["material_map"])
    r = pt.Renderer(s, pt.Camera(resolution=(64, 64)), max_depth=4)
    r.draw(4)
    out = r.read_sum()
    r.close()
    assert not out[..., :3].any() and np.all(out[..., 3] == 4)
    r = _renderer(pt, "cube", 64, 64, max_depth=4)  # with the default light
    r.draw(4)
    assert np.isfinite(r.read_sum()).all()
    r.close()


def test_full_size_properties_1080p(pt):
    """Benchmark size (1920x1080): hit buffer equals the reference's; every pixel gets its
    samples; two renders agree bit for bit; light pixels converge to radiance 4
    at depth 1."""
    r = _renderer(pt, "Box", 1920, 1080, max_depth=8)
    ids, _, _ = r.read_hit_ids(0)
    assert np.array_equal(ids, hit_fixture("Box", 1920, 1080)["ids"])
    r.draw(6)
    a = r.read_sum()
    st = r.stats()
    r.reset()
    r.draw(6)
    b = r.read_sum()
    r.close()
    assert np.array_equal(a, b)
    assert np.all(a[..., 3] == 6) and np.isfinite(a).all()
    assert st["paths"] == 1920 * 1080 * 6
    assert 2.0 < st["segments"] / st["paths"] < 4.5  # 16:9 frames more background than 1:1
    mean = a[..., :3].mean(axis=(0, 1)) / 6
    assert 0.02 < mean.min() and mean.max() < 0.12


def test_checkpoint_and_resume(pt, tmp_path):
    """State = sum buffer + iteration counter (the reference's accumulation texture and m_iterationCount,
    renderer.h:21-29): 5 + 7 iterations through a file equal 12 in one go, bit for bit; a file written for
    another camera, depth, size, partition or scene is refused."""
    path = str(tmp_path / "state.bin")
    straight = _renderer(pt, "Box", 96, 64, max_depth=6)
    straight.draw(12)
    ref = straight.read_sum()
    ref_stats = straight.stats()
    straight.close()

    a = _renderer(pt, "Box", 96, 64, max_depth=6)
    a.draw(5)
    a.save_state(path)
    a.close()
    b = _renderer(pt, "Box", 96, 64, max_depth=6)
    b.draw(3)  # overwritten by the load
    b.load_state(path)
    assert b.stats()["iterations"] == 5
    b.draw(7)
    assert np.array_equal(b.read_sum(), ref)
    assert b.stats()["iterations"] == 12 and b.stats()["paths"] == 12 * 96 * 64
    # the counters behind the statistics travel with the file: they cover the whole render, like `paths`
    for key in ("segments", "nan_samples", "hits_shaded"):
        assert b.stats()[key] == ref_stats[key], key
    b.close()

    # partitioned renderers keep their own files
    p1 = _renderer(pt, "Box", 96, 64, max_depth=6)
    p1.set_partition(1, 2)
    p1.draw(3)
    part = str(tmp_path / "rank1.bin")
    p1.save_state(part)
    p1.draw(3)
    want = p1.read_sum()
    p1.load_state(part)
    p1.draw(3)
    assert np.array_equal(p1.read_sum(), want)
    with pytest.raises(pt.PtError):
        p1.load_state(path)  # written by rank 0 of 1
    p1.close()

    for kw, size in [(dict(max_depth=7), (96, 64)), (dict(max_depth=6), (64, 96))]:
        c = _renderer(pt, "Box", *size, **kw)
        with pytest.raises(pt.PtError):
            c.load_state(path)
        c.close()
    c = _renderer(pt, "Box", 96, 64, max_depth=6)
    c.camera.orbit(0.2, 0.0)
    c.update_camera()
    with pytest.raises(pt.PtError):
        c.load_state(path)
    c.close()
    c = _renderer(pt, "Veggie", 96, 64, max_depth=6)
    with pytest.raises(pt.PtError):
        c.load_state(path)
    with pytest.raises(pt.PtError):
        c.load_state(str(tmp_path / "missing.bin"))
    with open(path, "rb") as f:
        blob = f.read()
    with open(path, "wb") as f:
        f.write(blob[:-100])
    c.close()
    d = _renderer(pt, "Box", 96, 64, max_depth=6)
    with pytest.raises(pt.PtError):
        d.load_state(path)  # truncated
    # a file of the first format (no counters, another scene hash) is refused by name, not as a damaged file;
    # one of the second format (no culled-ray counter) still loads
    with open(path, "wb") as f:
        f.write(b"PTAMDST1" + blob[8:])
    with pytest.raises(pt.PtError, match="version-1"):
        d.load_state(path)
    d.close()


@pytest.mark.parametrize("name,eye,focus,size,expect", [
    ("Box", (0.0, 1.5, 15.0), (0.0, -0.25, 0.0), (640, 360), "some"),      # the reference's pose: the box in the middle of a wide frame
    ("Veggie", (6.0, 3.0, 12.0), (1.5, 0.5, 0.0), (333, 187), "some"),     # from the side, odd image size (strips, not tiles)
    ("Box", (0.0, 0.5, 1.0), (0.0, 0.0, -1.0), (256, 256), "none"),        # inside the box: the bounds surround the eye
    ("Box", (0.0, 1.5, 15.0), (0.0, 1.5, 30.0), (256, 144), "all"),        # looking away from everything
    ("Box", (0.0, 1.5, 15.0), (3.5, -0.25, 0.0), (512, 288), "some"),      # the scene half out of the frame
])
def test_primary_rays_left_out_at_generation_change_nothing(pt, name, eye, focus, size, expect):
    """Ray generation leaves out the primary rays of pixels outside the window onto which the scene's bounding
    box and the lights project (they can only escape, frag:876).  With profile bit 2 every ray is made and
    traced (a ray that misses everything may still be the reference's hit at t = inf).  Same sum buffer bit
    for bit, same number of segments (a ray left out is still its path's first segment), and the count of
    rays left out is what the pose says: some, none (eye inside the bounds), all (nothing in view)."""
    w, h = size
    spp, depth = 6, 5

    def render(profile):
        cam = pt.Camera(eye=eye, focus=focus, resolution=(w, h))
        r = pt.Renderer(pt.Scene.from_arrays(**scene_arrays(name)), cam, max_depth=depth, profile=profile)
        r.draw(spp)
        s, st = r.read_sum(), r.stats()
        # one-sample draws afterwards take the same path with small regions
        r.draw(1)
        r.draw(1)
        s2, st2 = r.read_sum(), r.stats()
        r.close()
        return s, st, s2, st2

    s, st, s2, st2 = render(0)
    e, ste, e2, ste2 = render(4)
    assert np.array_equal(s, e) and np.array_equal(s2, e2)
    assert ste["primary_rays_culled"] == 0 and ste2["primary_rays_culled"] == 0
    assert st["segments"] == ste["segments"] and st2["segments"] == ste2["segments"]
    assert st["paths"] == w * h * spp and st2["paths"] == w * h * (spp + 2)
    culled = st["primary_rays_culled"]
    assert st2["primary_rays_culled"] * spp == culled * (spp + 2)  # the same pixels in every iteration
    if expect == "none":
        assert culled == 0
    elif expect == "all":
        assert culled == st["paths"] and not s[..., :3].any()
    else:
        assert 0.1 * st["paths"] < culled < 0.9 * st["paths"]


@pytest.mark.parametrize("depth", [1, 2, 3, 8])
@pytest.mark.parametrize("name", ["Box", "Veggie"])
def test_last_segment_rays_that_cannot_reach_the_light_change_nothing(pt, name, depth):
    """On a path's last allowed segment only reaching the light adds anything (a surface hit there is not shaded,
    an escape is black, frag:874-897): in a one-light scene the rays of that segment that do not point at the
    light are not made.  With profile bit 2 (which shades the last surface too) every ray is made.  Same sum
    buffer bit for bit, same number of segments; rays are skipped from depth 2 on, and none with two lights."""
    w, h, spp = 256, 144, 5

    def render(profile, lights=1):
        scene = pt.Scene.from_arrays(**scene_arrays(name))
        if lights == 2:
            scene.set_lights([pt.make_light((4, 4, 4), (0, 1.95, 0), (3.14 / 2, 0, 0), (1.25, 1.25, 1)),
                              pt.make_light((2, 1, 1), (1.0, 0.5, 1.5), (0.3, 0.2, 0.1), (0.5, 0.5, 1))])
        r = pt.Renderer(scene, pt.Camera(resolution=(w, h)), max_depth=depth, profile=profile)
        r.draw(spp)
        r.draw(1)  # a one-sample draw takes the one-launch path with small regions
        s, st = r.read_sum(), r.stats()
        r.close()
        return s, st

    s, st = render(0)
    e, ste = render(4)
    assert np.array_equal(s, e)
    assert st["segments"] == ste["segments"]
    assert ste["unlit_last_rays_skipped"] == 0
    if depth == 1:
        assert st["unlit_last_rays_skipped"] == 0
    else:
        # most rays of the last segment do not point at the light
        assert 0 < st["unlit_last_rays_skipped"] < st["paths"]
    if depth == 3 and name == "Box":
        s2, st2 = render(0, lights=2)
        e2, ste2 = render(4, lights=2)
        assert np.array_equal(s2, e2) and st2["segments"] == ste2["segments"] and st2["unlit_last_rays_skipped"] == 0

