"""Writes tests/golden/scenes/<name>.npz: the six Scene arrays (reference
layout, scene.h:53-62) of every bundled OBJ asset plus the default light
(main.cpp:122), so that GPU-side tests and the benchmark do not need the
reference tree.  The arrays come from this repo's loader, which
tests/test_scene_loader.py checks against the reference's own tiny_obj_loader
(oracle/_ref/ref_tinyobj_dump) on every asset.

Run in the build container:  python tests/golden/make_scene_dumps.py
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import pathtracer_amd as pt  # noqa: E402

ASSETS = os.environ.get("PT_REFERENCE_DIR", "/root/reference") + "/assets"

for obj in sorted(glob.glob(ASSETS + "/*.obj")):
    name = os.path.splitext(os.path.basename(obj))[0]
    s = pt.Scene(obj, ASSETS + "/")
    s.set_default_light()
    out = os.path.join(HERE, "scenes", name + ".npz")
    s.save_npz(out)
    print(name, s.n_triangles, os.path.getsize(out))
