"""The device BVH builder against the host builder: both must make the same tree
(node array, child codes, leaf-ordered triangle records), so everything rendered
through one equals what is rendered through the other."""
import numpy as np
import pytest

from conftest import hit_fixture, scene_arrays

pytestmark = pytest.mark.gpu


def _tree(pt, scene, builder, res=(48, 32)):
    r = pt.Renderer(scene, pt.Camera(resolution=res), max_depth=6, bvh_builder=builder)
    st = r.stats()
    nodes, tris = r.read_bvh()
    r.close()
    return st, nodes, tris


def _assert_same_tree(pt, scene):
    sh, nh, th = _tree(pt, scene, "host")
    sd, nd, td = _tree(pt, scene, "device")
    assert sh["bvh_builder"] == 0 and sd["bvh_builder"] == 1
    for k in ("n_triangles", "n_nodes", "bvh_depth", "stack_entries"):
        assert sh[k] == sd[k], k
    assert np.array_equal(nh[:, :12], nd[:, :12])  # child boxes, by value
    assert np.array_equal(nh[:, 12:].view(np.int32), nd[:, 12:].view(np.int32))  # child codes
    assert np.array_equal(th.view(np.uint32), td.view(np.uint32))  # triangle records, class bits included
    assert sd["sah_cost"] == pytest.approx(sh["sah_cost"], rel=1e-9)
    return sh


@pytest.mark.parametrize("name", ["Box", "Veggie", "Box_TransmissionSphere", "AnisotropicTest"])
def test_same_tree_as_the_host_builder(pt, name):
    _assert_same_tree(pt, pt.Scene.from_arrays(**scene_arrays(name)))


def test_same_tree_on_an_instanced_scene(pt):
    # 207 370 triangles: nodes too large for one wave at the top, the breadth-first
    # numbering of the first 4096 nodes and the depth-first numbering below it
    lat = pt.Scene.from_arrays(**scene_arrays("Veggie")).instance_lattice(10, 5)
    st = _assert_same_tree(pt, lat)
    assert st["n_nodes"] > 4096


def _soup(triangles, material=(0.5, 0.5, 0.5, 1, 0, 1, 0, 0)):
    tri = np.asarray(triangles, np.float32).reshape(-1, 3, 3)
    n = tri.shape[0]
    v = tri.reshape(-1, 3)
    idx = np.zeros((n, 4), np.float32)
    idx[:, :3] = np.arange(3 * n).reshape(n, 3)
    idx[:, 3] = np.arange(n)
    vd = np.tile(np.array([[0, 0, 1, 0, 0, 0]], np.float32), (3 * n, 1))
    return dict(vertices=v, indices=idx, vertex_data=vd, lights=np.zeros((0, 20), np.float32),
                materials=np.array([material], np.float32), material_map=np.zeros(n, np.uint32))


def test_tiny_scenes(pt):
    one = [[[0, 0, 0], [1, 0, 0], [0, 1, 0]]]
    three = [[[x, 0, 0], [x + 1, 0, 0], [x, 1, 0]] for x in (0.0, 2.0, 5.0)]
    for tris in (one, three, three + [[[9, 0, 0], [10, 0, 0], [9, 1, 1]], [[12, 0, 0], [13, 0, 0], [12, 1, 2]]]):
        _assert_same_tree(pt, pt.Scene.from_arrays(**_soup(tris)))


def test_coinciding_centroids(pt):
    # 40 copies of one triangle among others: the surface-area heuristic cannot separate them,
    # both builders halve them by triangle index
    rng = np.random.default_rng(5)
    base = rng.uniform(-1, 1, (30, 3, 3)).astype(np.float32)
    same = np.tile(np.array([[[0.25, 0.25, 0], [0.75, 0.25, 0], [0.25, 0.5, 0.125]]], np.float32), (40, 1, 1))
    tris = np.concatenate([base[:15], same[:20], base[15:], same[20:]])
    _assert_same_tree(pt, pt.Scene.from_arrays(**_soup(tris)))
    # more of them than one wave sorts: the device builder says so, the host builder goes on
    many = np.tile(same[:1], (300, 1, 1))
    scene = pt.Scene.from_arrays(**_soup(many))
    with pytest.raises(pt.PtError, match="host builder"):
        pt.Renderer(scene, pt.Camera(resolution=(32, 32)), bvh_builder="device")
    pt.Renderer(scene, pt.Camera(resolution=(32, 32)), bvh_builder="host").close()


def test_automatic_choice_and_fall_back(pt):
    rng = np.random.default_rng(3)
    c = rng.uniform(-4, 4, (70000, 1, 3)).astype(np.float32)
    soup = c + rng.uniform(-0.05, 0.05, (70000, 3, 3)).astype(np.float32)
    cam = pt.Camera(resolution=(32, 32))
    small = pt.Renderer(pt.Scene.from_arrays(**scene_arrays("Box")), cam)
    assert small.stats()["bvh_builder"] == 0  # not worth the launches
    small.close()
    big = pt.Renderer(pt.Scene.from_arrays(**_soup(soup)), cam)
    assert big.stats()["bvh_builder"] == 1
    nodes_auto, tris_auto = big.read_bvh()
    big.close()
    # the same soup with 300 copies of one triangle in it: the device builder declines, the host builder takes over
    same = np.tile(np.array([[[0.25, 0.25, 0], [0.75, 0.25, 0], [0.25, 0.5, 0.125]]], np.float32), (300, 1, 1))
    r = pt.Renderer(pt.Scene.from_arrays(**_soup(np.concatenate([soup, same]))), cam)
    st = r.stats()
    assert st["bvh_builder"] == 0 and st["n_triangles"] == 70300
    ids = r.read_hit_ids(0)[0]
    r.close()
    assert ids.shape == (32, 32)
    host = pt.Renderer(pt.Scene.from_arrays(**_soup(soup)), cam, bvh_builder="host")
    nodes_host, tris_host = host.read_bvh()
    host.close()
    assert np.array_equal(nodes_auto[:, :12], nodes_host[:, :12])
    assert np.array_equal(nodes_auto[:, 12:].view(np.int32), nodes_host[:, 12:].view(np.int32))
    assert np.array_equal(tris_auto.view(np.uint32), tris_host.view(np.uint32))


def test_random_soup_same_tree(pt):
    rng = np.random.default_rng(11)
    c = rng.uniform(-4, 4, (20000, 1, 3)).astype(np.float32)
    tris = c + rng.uniform(-0.05, 0.05, (20000, 3, 3)).astype(np.float32)
    _assert_same_tree(pt, pt.Scene.from_arrays(**_soup(tris)))


def test_flat_and_sliver_geometry_same_tree(pt):
    # a regular grid in the plane z = 0 (one centroid axis has no extent, many centroids share coordinates)
    n = 96
    xs, ys = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32))
    p00 = np.stack([xs, ys, np.zeros_like(xs)], -1).reshape(-1, 3)
    a = np.stack([p00, p00 + [1, 0, 0], p00 + [0, 1, 0]], 1)
    b = np.stack([p00 + [1, 0, 0], p00 + [1, 1, 0], p00 + [0, 1, 0]], 1)
    _assert_same_tree(pt, pt.Scene.from_arrays(**_soup(np.concatenate([a, b]))))
    # long slivers crossing the whole scene, zero-area triangles among them
    rng = np.random.default_rng(2)
    c = rng.uniform(-1, 1, (3000, 1, 3)).astype(np.float32)
    d = rng.normal(size=(3000, 1, 3)).astype(np.float32) * 5
    sl = np.concatenate([c - d, c + d, c + rng.normal(size=(3000, 1, 3)).astype(np.float32) * 1e-3], 1)
    sl[::50, 2] = sl[::50, 1]  # degenerate: two corners coincide
    _assert_same_tree(pt, pt.Scene.from_arrays(**_soup(sl)))


def test_render_through_the_device_built_tree(pt):
    ids_ref = hit_fixture("Veggie", 256, 256, 0)["ids"]
    scene = pt.Scene.from_arrays(**scene_arrays("Veggie"))
    sums = []
    for builder in ("host", "device"):
        r = pt.Renderer(scene, pt.Camera(resolution=(256, 256)), max_depth=8, bvh_builder=builder)
        ids = r.read_hit_ids(0)[0]
        assert np.array_equal(ids, ids_ref)
        r.draw(4)
        sums.append(r.read_sum())
        r.close()
    assert np.array_equal(sums[0], sums[1])
