"""OBJ/MTL loading and the Scene material rules (scene.h:79-218).

Where the reference tree is present, the loader is compared with the
reference's own vendored tiny_obj_loader compiled as-is
(oracle/_ref/ref_tinyobj_dump) on every bundled asset and on synthetic edge
cases; the committed scene dumps (tests/golden/scenes) must equal a fresh
load, and the material table must carry the values dumped from the
reference's Scene."""
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import ALL_SCENES, REFERENCE, ROOT, scene_arrays

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_tinyobj_dump")
ASSETS = os.path.join(REFERENCE, "assets")
needs_ref = pytest.mark.skipif(not (os.path.exists(REF_DUMP) and os.path.isdir(ASSETS)),
                               reason="reference tree / oracle/_ref not available")


def _bits(words):
    return np.array([int(w, 16) for w in words], np.uint32).view(np.float32)


def _parse_ref_dump(text):
    lines = text.splitlines()
    pos = 0

    def take():
        nonlocal pos
        pos += 1
        return lines[pos - 1]

    nv = int(take().split()[1])
    verts = np.array([_bits(take().split()) for _ in range(nv)], np.float32).reshape(-1, 3)
    nn = int(take().split()[1])
    normals = np.array([_bits(take().split()) for _ in range(nn)], np.float32).reshape(-1, 3)
    ns = int(take().split()[1])
    shapes = []
    for _ in range(ns):
        nf = int(take().split()[1])
        shapes.append(np.array([[int(x) for x in take().split()] for _ in range(nf)], np.int64).reshape(-1, 7))
    nm = int(take().split()[1])
    mats = []
    for _ in range(nm):
        w = take().split()
        f = _bits(w[:11])
        mats.append({"Kd": f[0:3], "Ka": f[3:6], "Tf": f[6:9], "Ns": f[9], "Ni": f[10], "illum": int(w[11]),
                     "isPBR": int(w[12]), "Pr": _bits(w[13:14])[0], "Pm": _bits(w[14:15])[0],
                     "aniso": _bits(w[15:16])[0]})
    return verts, normals, shapes, mats


def _expected_arrays(verts, normals, shapes, mats):
    """scene.h:92-217 applied to tiny_obj_loader's output, in numpy float32."""
    idx, vdata, mmap = [], [], []
    for faces in shapes:
        first = len(idx)
        for f, row in enumerate(faces):
            idx.append([row[0], row[2], row[4], first + f])
            for k in range(3):
                ni = row[2 * k + 1]
                n = normals[ni] if ni != -1 else np.array([0, 0, -1], np.float32)
                vdata.append([n[0], n[1], n[2], 0, 0, 0])
            mmap.append(row[6] & 0xFFFFFFFF)
    out = []
    f32 = np.float32
    for m in mats:
        if not m["isPBR"]:
            if m["Ns"] < 0 and m["illum"] == 2:
                rough = f32(0)
            else:
                ns = max(f32(0), min(m["Ns"], f32(1000)))
                rough = f32(1) - np.sqrt(f32(ns) / f32(1000), dtype=f32)
            metal = f32(0)
            if m["illum"] == 3:
                metal = f32(f32(f32(m["Ka"][0] + m["Ka"][1]) + m["Ka"][2]) / f32(3))
                if metal < 0:
                    metal = f32(1)
            aniso, trans = f32(0), f32(0)
        else:
            rough, metal, aniso = m["Pr"], m["Pm"], m["aniso"]
            trans = f32(f32(f32(m["Tf"][0] + m["Tf"][1]) + m["Tf"][2]) / f32(3))
        out.append([m["Kd"][0], m["Kd"][1], m["Kd"][2], rough, metal, m["Ni"], aniso, trans])
    return (np.array(idx, np.float32).reshape(-1, 4), np.array(vdata, np.float32).reshape(-1, 6),
            np.array(mmap, np.uint32), np.array(out, np.float32).reshape(-1, 8))


def _compare_with_reference(pt, obj, mtl_root):
    dump = subprocess.run([REF_DUMP, obj, mtl_root], check=True, capture_output=True, text=True).stdout
    verts, normals, shapes, mats = _parse_ref_dump(dump)
    a = pt.Scene(obj, mtl_root).arrays()
    idx, vdata, mmap, materials = _expected_arrays(verts, normals, shapes, mats)
    assert np.array_equal(a["vertices"].view(np.uint32), verts.view(np.uint32))
    assert np.array_equal(a["indices"], idx)
    assert np.array_equal(a["vertex_data"].view(np.uint32), vdata.view(np.uint32))
    assert np.array_equal(a["material_map"], mmap)
    assert np.array_equal(a["materials"].view(np.uint32), materials.view(np.uint32))
    return a


@needs_ref
@pytest.mark.parametrize("name", ALL_SCENES)
def test_loader_equals_reference_tinyobj_on_assets(pt, name):
    a = _compare_with_reference(pt, os.path.join(ASSETS, name + ".obj"), ASSETS + "/")
    dump = scene_arrays(name)
    for k in ("vertices", "indices", "vertex_data", "materials", "material_map"):
        assert np.array_equal(a[k].view(np.uint32), dump[k].view(np.uint32)), f"committed dump of {name} is stale: {k}"


EDGE_OBJ = {
    "quads_and_ngon": "mtllib m.mtl\no A\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\nvn 0 0 1\n"
                      "usemtl red\nf 1//1 2//1 3//1 4//1\nf 1//1 2//1 3//1 4//1 5//1\n",
    "negative_indices": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf -3//-1 -2//-1 -1//-1\n",
    "no_normals_no_material": "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
    "crlf_and_tabs": "mtllib m.mtl\r\no\tA\r\nv\t0 0 0\r\nv 1 0 0\r\nv 0 1 0\r\nusemtl pbr\r\nf 1 2 3\r\n",
    "material_switch_mid_object": "mtllib m.mtl\no A\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nusemtl red\nf 1 2 3\n"
                                  "usemtl pbr\nf 2 4 3\no B\nusemtl red\nf 1 2 4\n",
    "exponents_and_signs": "v 1e-3 -2.5E+1 +3.25\nv 1.0e2 .5 5.\nv 0.1234567890123 7 8\nf 1 2 3\n",
    "texcoord_forms": "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 1 0\nf 1/1/1 2/1/1 3/1/1\nf 1/1 2/1 3/1\n",
    "unknown_material": "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl nosuch\nf 1 2 3\n",
    "missing_mtl_file": "mtllib absent.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n",
    "groups": "v 0 0 0\nv 1 0 0\nv 0 1 0\ng first\nf 1 2 3\ng second extra\nf 3 2 1\n",
    "empty": "",
}
EDGE_MTL = ("newmtl red\nKd 0.5 0 0\nKa 0.2 0.4 0.9\nNs 250\nNi 1.45\nillum 3\n\n"
            "newmtl pbr\nKd 0.1 0.2 0.3\nPr 0.25\nPm 0.5\naniso 0.75\nanisor 0.1\nTf 1 0.5 0.25\nNi 1.33\n"
            "newmtl neg\nNs -5\nillum 2\nKa -1 -1 -1\n")


@needs_ref
@pytest.mark.parametrize("case", sorted(EDGE_OBJ))
def test_loader_equals_reference_tinyobj_on_edge_cases(pt, tmp_path, case):
    (tmp_path / "m.mtl").write_bytes(EDGE_MTL.encode())
    obj = tmp_path / (case + ".obj")
    obj.write_bytes(EDGE_OBJ[case].encode())
    _compare_with_reference(pt, str(obj), str(tmp_path) + "/")


def test_edge_cases_without_reference(pt, tmp_path):
    (tmp_path / "m.mtl").write_bytes(EDGE_MTL.encode())
    obj = tmp_path / "q.obj"
    obj.write_bytes(EDGE_OBJ["quads_and_ngon"].encode())
    a = pt.Scene(str(obj), str(tmp_path) + "/").arrays()
    # quad -> 2 triangles, pentagon -> 3, as fans (v0, v_{k-1}, v_k)
    assert a["indices"][:, :3].astype(int).tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3], [0, 3, 4]]
    assert a["indices"][:, 3].astype(int).tolist() == [0, 1, 2, 3, 4]
    assert a["material_map"].tolist() == [0] * 5
    # legacy rules: Ns 250 -> 1 - sqrt(0.25) = 0.5 ; illum 3 -> metallic = mean(Ka)
    red = a["materials"][0]
    assert red[3] == np.float32(0.5) and abs(red[4] - 0.5) < 1e-6 and red[5] == np.float32(1.45)
    # PBR rules: transmission = mean(Tf)
    pbr = a["materials"][1]
    assert pbr[3] == np.float32(0.25) and pbr[4] == np.float32(0.5) and pbr[6] == np.float32(0.75)
    assert abs(pbr[7] - (1 + 0.5 + 0.25) / 3) < 1e-6
    # Ns < 0 with illum 2 -> roughness 0 ; illum 2 -> metallic 0
    neg = a["materials"][2]
    assert neg[3] == 0 and neg[4] == 0

    obj2 = tmp_path / "n.obj"
    obj2.write_bytes(EDGE_OBJ["no_normals_no_material"].encode())
    b = pt.Scene(str(obj2), str(tmp_path) + "/").arrays()
    assert b["material_map"].tolist() == [0xFFFFFFFF]          # scene.h:133 with material id -1
    assert b["vertex_data"][:, :3].tolist() == [[0, 0, -1]] * 3  # scene.h:119 default normal


def test_missing_obj_is_an_error(pt, tmp_path):
    with pytest.raises(pt.PtError) as e:
        pt.Scene(str(tmp_path / "nope.obj"))
    assert "Cannot open file" in str(e.value)


def test_material_tables_dumped_from_reference_scene():
    """Values the reference's own Scene produced for these assets."""
    f = np.float32
    exp = {
        "Box": [[0, .8, 0, 1, 0, 1.5, 0, 0], [.8, .8, .8, 1, 0, 1.5, 0, 0], [.5, 0, 0, 1, 0, 1.5, 0, 0],
                [.8, .8, .8, 0.0513167, 1, 1.5, 0, 0]],
        "Veggie": [[0, 0, 0, .5, 0, 1.5, 0, 0], [0, .8, 0, 1, 0, 1.5, 0, 0], [.8, .8, .8, 1, 0, 1.5, 0, 0],
                   [.5, 0, 0, 1, 0, 1.5, 0, 0]],
        "Box_Transmission": [[0, .8, 0, 1, 0, 1.5, 0, 0], [0, 0, .5, 1, 0, 1.5, 0, 0], [.8, .8, .8, 1, 0, 1.5, 0, 0],
                             [.5, 0, 0, 1, 0, 1.5, 0, 0], [.8, .8, .8, .05, 0, 1.5, 0, 1]],
        "Box_Anisotropic": [[.8, .8, .8, 1, 0, 1.5, 0, 0], [.5, .8, .5, .5, 1, 1.5, 1, 0],
                            [.8, .5, .5, .1, 1, 1.5, 1, 0], [.5, .5, .8, .75, 1, 1.5, 1, 0]],
    }
    for name, table in exp.items():
        got = scene_arrays(name)["materials"]
        assert got.shape == (len(table), 8)
        assert np.allclose(got, np.array(table, f), atol=1e-6), name


def test_triangle_counts_of_assets():
    counts = {"Box": 155, "Veggie": 1450, "Box_Transmission": 366, "Box_TransmissionSphere": 514,
              "Box_Anisotropic": 46, "AnisotropicTest": 46, "Box_IOR": 118, "Box_Metallic": 2890, "TEST": 514,
              "cube": 12, "sphere": 960}
    for name, n in counts.items():
        a = scene_arrays(name)
        assert a["indices"].shape == (n, 4)
        assert a["vertex_data"].shape == (3 * n, 6)
        assert np.array_equal(a["indices"][:, 3], np.arange(n, dtype=np.float32))
    assert np.all(scene_arrays("cube")["material_map"] == 0xFFFFFFFF)  # no usemtl


def test_default_scene_and_roundtrip(pt, tmp_path):
    s = pt.Scene()
    a = s.arrays()
    assert a["vertices"].tolist() == [[-1, -1, 0], [1, -1, 0], [0, 1, 0]]
    assert a["indices"].tolist() == [[0, 1, 2, 0]]
    assert a["lights"].shape == (1, 20) and a["lights"][0, 0] == 1.0
    box = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Box.npz"))
    p = str(tmp_path / "b.npz")
    box.save_npz(p)
    again = pt.Scene.load_npz(p).arrays()
    for k, v in box.arrays().items():
        assert np.array_equal(v, again[k])


def test_lattice_instancing(pt):
    base = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Veggie.npz"))
    lat = base.instance_lattice(10, 3)
    a = lat.arrays()
    assert a["indices"].shape[0] == 10 + 1440 * 27
    assert np.array_equal(a["indices"][:, 3], np.arange(a["indices"].shape[0], dtype=np.float32))
    assert np.array_equal(a["indices"][:10, :3], base.arrays()["indices"][:10, :3])  # room untouched
    v = a["vertices"][a["indices"][10:, :3].astype(np.int64).reshape(-1)]
    assert v.min() >= -2.0 - 1e-4 and v.max() <= 2.0 + 1e-4
    assert a["materials"].shape == base.arrays()["materials"].shape
    # deterministic
    b = base.instance_lattice(10, 3).arrays()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    info = lat.bvh_info()
    assert info["n_triangles"] == a["indices"].shape[0] and info["max_leaf_size"] <= 4
    assert info["depth"] <= info["stack_entries"]


def test_bvh_build_is_the_same_on_one_thread_and_on_many(pt, monkeypatch):
    """Above 65 536 triangles the subtrees below the top of the tree are built on worker threads and
    spliced in subtree order: the tree must not depend on the number of threads."""
    from conftest import scene_arrays

    lat = pt.Scene.from_arrays(**scene_arrays("Veggie")).instance_lattice(10, 5)
    assert lat.n_triangles == 10 + 1440 * 125
    infos = []
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("PT_BVH_THREADS", threads)
        infos.append(lat.bvh_info())
    for i in infos[1:]:
        for key in ("n_nodes", "n_leaves", "depth", "max_leaf_size", "stack_entries", "n_triangles"):
            assert i[key] == infos[0][key], key
        assert abs(i["sah_cost"] - infos[0]["sah_cost"]) <= 1e-9 * infos[0]["sah_cost"]


def test_small_scenes_get_the_best_of_several_sah_binnings(pt):
    # 16 bins alone give Veggie.obj a tree of SAH cost 13.27; among 8..64 bins the builder finds 11.89
    info = pt.Scene.from_arrays(**scene_arrays("Veggie")).bvh_info()
    assert info["sah_cost"] < 12.0 and info["n_triangles"] == 1450 and info["max_leaf_size"] <= 4
