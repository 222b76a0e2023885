"""The drop-in boundary: include/pathtracer_amd.h vs the shared library.
No GPU needed: only loading, symbol export and host-side error behaviour."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_arrays

HEADER = os.path.join(ROOT, "include", "pathtracer_amd.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(pt):
    declared = _declared_functions()
    assert len(declared) >= 35
    out = subprocess.run(["nm", "-D", "--defined-only", pt.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [f for f in declared if f not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"
    # and the Python binding covers all of them
    assert sorted(pt.EXPORTED_SYMBOLS) == declared


def test_library_has_gfx950_code_object(pt):
    blob = open(pt.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "the shared library carries no gfx950 device code"
    for kernel in (b"k_raygen", b"k_extend_bvh", b"k_shade", b"k_accumulate"):
        assert kernel in blob


def test_no_oracle_in_product(pt):
    """The product must not link or load the test oracle."""
    out = subprocess.run(["ldd", pt.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "pathtracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert "pt_oracle" not in src and "liboracle" not in src and "oracle_binding" not in src, f


def test_version_and_config_defaults(pt):
    assert "gfx950" in pt.version()
    cfg = pt.Config()
    pt._lib.pt_config_default(C.byref(cfg))
    # the literals of main.cpp:126 and pathtracer.frag.glsl:306,874
    assert (cfg.width, cfg.height, cfg.max_depth) == (1280, 720, 10)
    assert cfg.fov_y_degrees == pytest.approx(19.5)
    assert cfg.use_bvh == 1 and cfg.device == -1


def test_error_reporting_without_exceptions_across_the_abi(pt, tmp_path):
    h = C.c_void_p()
    rc = pt._lib.pt_scene_load_obj(os.fsencode(str(tmp_path / "missing.obj")), None, C.byref(h))
    assert rc == -2 and b"Cannot open file" in pt._lib.pt_last_error()
    assert pt._lib.pt_scene_load_obj(None, None, C.byref(h)) == -1
    assert pt._lib.pt_camera_update(None) == -1
    assert pt._lib.pt_render(None, 1) == -1
    assert pt._lib.pt_create(None, None, C.byref(h)) == -1


def test_renderer_creation_fails_loudly_without_a_gpu(pt):
    if pt.device_count() > 0:
        pytest.skip("a GPU is present")
    s = pt.Scene.load_npz(os.path.join(ROOT, "tests/golden/scenes/Box.npz"))
    with pytest.raises(pt.PtError) as e:
        pt.Renderer(s, pt.Camera(resolution=(16, 16)))
    assert "error -3" in str(e.value)  # PT_ERR_DEVICE: no CPU fallback exists


def test_bvh_inspect_on_all_scenes(pt):
    from conftest import ALL_SCENES

    for name in ALL_SCENES:
        s = pt.Scene.from_arrays(**scene_arrays(name))
        info = s.bvh_info()
        n = s.n_triangles
        assert info["n_triangles"] == n
        assert info["max_leaf_size"] <= 4
        assert info["n_nodes"] >= 1 and info["n_leaves"] >= 1
        assert info["depth"] <= info["stack_entries"] <= 64
        assert info["depth"] <= 2 * int(np.ceil(np.log2(max(n, 2)))) + 2, "SAH tree unexpectedly deep"


def test_bvh_inspect_degenerate_inputs(pt):
    # all triangles identical: SAH cannot split, the builder must fall back to median splits
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n = 1000
    idx = np.tile(np.array([[0, 1, 2, 0]], np.float32), (n, 1))
    idx[:, 3] = np.arange(n)
    vd = np.tile(np.array([[0, 0, 1, 0, 0, 0]], np.float32), (3 * n, 1))
    s = pt.Scene.from_arrays(v, idx, vd, np.zeros((0, 20), np.float32), np.zeros((1, 8), np.float32),
                             np.zeros(n, np.uint32))
    info = s.bvh_info()
    assert info["n_triangles"] == n and info["depth"] <= 12
    # empty scene and single triangle
    e = pt.Scene.from_arrays(np.zeros((0, 3)), np.zeros((0, 4)), np.zeros((0, 6)), np.zeros((0, 20)),
                             np.zeros((0, 8)), np.zeros(0, np.uint32))
    assert e.bvh_info()["n_leaves"] == 0
    one = pt.Scene()
    assert one.bvh_info()["n_leaves"] == 1
    # out-of-range vertex index is rejected
    bad = pt.Scene.from_arrays(v, np.array([[0, 1, 7, 0]], np.float32), vd[:3], np.zeros((0, 20)),
                               np.zeros((1, 8)), np.zeros(1, np.uint32))
    with pytest.raises(pt.PtError):
        bad.bvh_info()
