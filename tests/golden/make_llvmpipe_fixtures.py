"""Generates the golden fixtures under tests/golden/llvmpipe/ by running the
reference's own fragment shader on Mesa llvmpipe (oracle/llvmpipe/lp_harness.c).
Container-only: needs /root/reference and Mesa's swrast_dri.so.  The fixtures
are data (inputs: committed scene arrays + default camera; outputs: what the
reference rendered); no reference source is stored.

  python tests/golden/make_llvmpipe_fixtures.py ids      # primary hit buffers
  python tests/golden/make_llvmpipe_fixtures.py images   # converged float images (minutes)
  python tests/golden/make_llvmpipe_fixtures.py timing   # llvmpipe throughput table
  python tests/golden/make_llvmpipe_fixtures.py ids_large  # hit buffers at benchmark sizes, a moved camera (minutes)
  python tests/golden/make_llvmpipe_fixtures.py post     # what post.frag draws from a float image
  python tests/golden/make_llvmpipe_fixtures.py images_all  # 64-spp images of the assets without a 256-spp one
  python tests/golden/make_llvmpipe_fixtures.py two_lights  # Box with two overlapping lights, both array orders
  python tests/golden/make_llvmpipe_fixtures.py image_lattice  # Veggie instanced 2^3 times (11 530 triangles), 256x256, depth 5
  python tests/golden/make_llvmpipe_fixtures.py image_512  # Box at 512x512, depth 5, 64 spp
  python tests/golden/make_llvmpipe_fixtures.py image_filtered_4k  # Veggie at 3840x2160, kept as 16x16 block means (~25 min)
"""
import glob
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "llvmpipe"))
import pathtracer_amd as pt  # noqa: E402
import run_reference as rr  # noqa: E402

OUT = os.path.join(HERE, "llvmpipe")
os.makedirs(OUT, exist_ok=True)


def scene_and_cam(name, w, h):
    s = pt.Scene.load_npz(os.path.join(HERE, "scenes", name + ".npz"))
    cam = pt.Camera(resolution=(w, h))
    cam.update()
    return s.arrays(), (cam.eye, cam.forward, cam.up, cam.right)


def moved_camera(w, h, moves):
    """Camera after a sequence of (op, args) moves, e.g. ("orbit", (0.4, 0.1)); axes as Camera::update leaves them."""
    cam = pt.Camera(resolution=(w, h))
    cam.update()
    for op, args in moves:
        getattr(cam, op)(*args)
        cam.update()
    return cam.eye, cam.forward, cam.up, cam.right


def ids_fixture(name, w, h, iteration, full, cam=None, tag=""):
    a, c = scene_and_cam(name, w, h)
    if cam is not None:
        c = cam
    img, _ = rr.run(a, c, w, h, "ids", 1, iteration)
    ids = img[..., 3].astype(np.int32)
    assert np.array_equal(ids.astype(np.float32), img[..., 3])
    out = {"ids": ids, "iteration": np.int32(iteration)}
    if full:
        tn, _ = rr.run(a, c, w, h, "tn", 1, iteration)
        out["dirs_bits"] = img[..., :3].copy().view(np.uint32)
        out["t_bits"] = tn[..., 0].copy().view(np.uint32)
        out["normal_bits"] = tn[..., 1:].copy().view(np.uint32)
    if cam is not None:
        out["camera"] = np.stack(cam).astype(np.float32)  # eye, forward, up, right handed to the shader
    path = os.path.join(OUT, f"hit_{name}_{w}x{h}_it{iteration}{tag}.npz")
    np.savez_compressed(path, **out)
    print(name, w, h, iteration, "geometry", int((ids >= 0).sum()), "light", int((ids <= -2).sum()),
          "miss", int((ids == -1).sum()), os.path.getsize(path), flush=True)


def image_fixture(name, w, h, spp, depth):
    """Float-accumulated image (non-finite samples zeroed) of iterations 0..spp-1.  Stored as float16 plus, in
    full precision, the few values float16 does not hold to 1e-3 (large or tiny samples)."""
    a, c = scene_and_cam(name, w, h)
    t0 = time.time()
    img, secs = rr.run(a, c, w, h, "float0", spp, depth)
    rgb = img[..., :3].astype(np.float32)
    half = rgb.astype(np.float16)
    back = half.astype(np.float32)
    odd = np.flatnonzero(~(np.abs(back - rgb) <= 1e-3 * np.abs(rgb)).ravel())
    path = os.path.join(OUT, f"image_{name}_{w}x{h}_spp{spp}_depth{depth}.npz")
    np.savez_compressed(path, mean_rgb_f16=half, exact_index=odd.astype(np.int64), exact_value=rgb.ravel()[odd],
                        spp=np.int32(spp), depth=np.int32(depth), llvmpipe_seconds=np.float64(secs))
    print(name, w, h, spp, depth, "mean", rgb.mean(axis=(0, 1), dtype=np.float64), "exact", odd.size,
          "%.1fs" % (time.time() - t0), os.path.getsize(path), flush=True)


def filtered_image_fixture(name, w, h, spp, depth, block=16):
    """The same image at a size whose pixels would not fit in the repository: kept as the means of its block x block
    pixel blocks (what the filtered relMSE and the channel means of the image gate are computed from), float32."""
    a, c = scene_and_cam(name, w, h)
    t0 = time.time()
    img, secs = rr.run(a, c, w, h, "float0", spp, depth)
    rgb = img[..., :3].astype(np.float64)
    assert h % block == 0 and w % block == 0
    means = rgb.reshape(h // block, block, w // block, block, 3).mean(axis=(1, 3)).astype(np.float32)
    path = os.path.join(OUT, f"image_{name}_{w}x{h}_spp{spp}_depth{depth}_block{block}.npz")
    np.savez_compressed(path, block_means=means, block=np.int32(block), channel_means=rgb.mean(axis=(0, 1)),
                        spp=np.int32(spp), depth=np.int32(depth), llvmpipe_seconds=np.float64(secs))
    print(name, w, h, spp, depth, "mean", rgb.mean(axis=(0, 1)), "%.1fs" % (time.time() - t0), os.path.getsize(path), flush=True)


def main():
    what =sys.argv[1] if len(sys.argv) > 1 else "ids"
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "scenes", "*.npz")))
    if what == "ids":
        for n in names:
            ids_fixture(n, 256, 256, 0, full=n in ("Box", "Veggie", "Box_Transmission", "Box_Anisotropic"))
        ids_fixture("Box", 256, 256, 7, full=True)
        ids_fixture("Box", 192, 108, 3, full=True)  # non-square aspect
        for n in ("Box", "Veggie"):
            ids_fixture(n, 1920, 1080, 0, full=False)
    elif what == "ids_large":
        # the remaining benchmark configurations at their own sizes
        for n in ("Box_Transmission", "Box_TransmissionSphere", "Box_Anisotropic", "AnisotropicTest"):
            ids_fixture(n, 1920, 1080, 0, full=False)
        # a pose the mouse callbacks of main.cpp:75-106 could have produced: orbit, then zoom, then pan
        moves = [("orbit", (0.4, 0.1)), ("zoom", (0.7,)), ("pan", (0.3, -0.2))]
        ids_fixture("Box", 640, 360, 2, full=True, cam=moved_camera(640, 360, moves), tag="_moved")
        ids_fixture("Veggie", 3840, 2160, 0, full=False)
    elif what == "post":
        a, c = scene_and_cam("Box", 256, 256)
        (img, post), _ = rr.run(a, c, 256, 256, "post", 64, 8)
        path = os.path.join(OUT, "post_Box_256x256_spp64_depth8.npz")
        np.savez_compressed(path, mean_rgb=img[..., :3].astype(np.float32), post_rgba8=post, spp=np.int32(64),
                            depth=np.int32(8))
        print("post", img[..., :3].mean(axis=(0, 1)), post[..., :3].mean(axis=(0, 1)), os.path.getsize(path), flush=True)
    elif what == "images":
        jobs = [("Box", 8), ("Box", 5), ("Box_Anisotropic", 8), ("AnisotropicTest", 8), ("Box_Transmission", 8),
                ("Box_TransmissionSphere", 8), ("Veggie", 8)]
        for n, depth in jobs:
            a, c = scene_and_cam(n, 256, 256)
            t0 = time.time()
            img, secs = rr.run(a, c, 256, 256, "float0", 256, depth)
            path = os.path.join(OUT, f"image_{n}_256x256_spp256_depth{depth}.npz")
            np.savez_compressed(path, mean_rgb=img[..., :3].astype(np.float32), spp=np.int32(256),
                                depth=np.int32(depth), llvmpipe_seconds=np.float64(secs))
            print(n, depth, "mean", img[..., :3].mean(axis=(0, 1)), "%.1fs" % (time.time() - t0),
                  os.path.getsize(path), flush=True)
        # the as-shipped RGBA8 feedback accumulator
        a, c = scene_and_cam("Box", 256, 256)
        img, secs = rr.run(a, c, 256, 256, "rgba8", 64, 10)
        np.savez_compressed(os.path.join(OUT, "rgba8_Box_256x256_spp64_depth10.npz"), rgba8=img, spp=np.int32(64),
                            depth=np.int32(10))
        print("rgba8 Box mean", img[..., :3].mean(axis=(0, 1)) / 255.0, flush=True)
    elif what == "images_1080p":
        # float images at the benchmark size and depth (iterations 0..15): Box ~1 min, Veggie ~6 min on 8 cores
        for n in sys.argv[2:] or ("Box", "Veggie"):
            image_fixture(n, 1920, 1080, 16, 8)
    elif what == "images_all":
        # every bundled asset that has no 256-spp image: 64 spp at 256 x 256, depth 8 (the metallic sweeps, IOR,
        # smooth shading, the material test scenes ...; ~15 min on 8 cores)
        have = {"Box", "Box_Anisotropic", "AnisotropicTest", "Box_Transmission", "Box_TransmissionSphere", "Veggie"}
        for n in sys.argv[2:] or [n for n in names if n not in have]:
            image_fixture(n, 256, 256, 64, 8)
    elif what == "two_lights":
        # Box with two rectangles that overlap on screen, in both orders of the light array: the shader's loop over
        # lights (frag:418-430) sets t = 0 at the first light it accepts, after which no later light can win -- the
        # first light of the ARRAY that the ray meets is the one reported, not the nearest.  Hit codes of both
        # orders and a float image of one.
        s = pt.Scene.load_npz(os.path.join(HERE, "scenes", "Box.npz"))
        cam = pt.Camera(resolution=(256, 256))
        cam.update()
        c = (cam.eye, cam.forward, cam.up, cam.right)
        a_light = pt.make_light((4, 4, 4), (0, 1.95, 0), (3.14 / 2, 0, 0), (1.25, 1.25, 1))
        b_light = pt.make_light((1, 2, 3), (0.2, 1.93, 0.6), (3.14 / 2, 0, 0), (1.0, 1.0, 1))
        out = {"lights_ab": np.stack([a_light, b_light]).astype(np.float32)}
        for name, lights in (("ab", [a_light, b_light]), ("ba", [b_light, a_light])):
            s.set_lights(lights)
            img, _ = rr.run(s.arrays(), c, 256, 256, "ids", 1, 0)
            out["ids_" + name] = img[..., 3].astype(np.int32)
        img, secs = rr.run(s.arrays(), c, 256, 256, "float0", 16, 8)  # order b, a
        out["mean_rgb_ba"] = img[..., :3].astype(np.float32)
        path = os.path.join(OUT, "two_lights_Box_256x256.npz")
        np.savez_compressed(path, spp=np.int32(16), depth=np.int32(8), **out)
        print("two lights", {k: int((out["ids_ab"] == k).sum()) for k in (-2, -3)},
              {k: int((out["ids_ba"] == k).sum()) for k in (-2, -3)}, out["mean_rgb_ba"].mean(axis=(0, 1)), os.path.getsize(path), flush=True)
    elif what == "image_lattice":
        # Veggie's furniture instanced 2 x 2 x 2 times (11 530 triangles, pt_scene_instance_lattice): the kind of scene
        # whose tree is read from memory, as large as llvmpipe can render -- its shader compiler ends every loop of
        # an invocation once 65 535 iterations have been spent in all of them together (gallivm's guard against
        # hangs), and the shader loops over all triangles for every segment: triangles x depth must stay below that
        # (11 530 x 5 here; with 38 890 triangles the reference's image is black from the second segment on).
        s = pt.Scene.load_npz(os.path.join(HERE, "scenes", "Veggie.npz")).instance_lattice(10, 2)
        # (the default light hangs inside the lattice, where next to no camera path finds it: a second rectangle in
        # the room's open front, facing in, lights the scene; the same two lights in the test)
        s.set_lights([pt.make_light((4, 4, 4), (0, 1.95, 0), (3.14 / 2, 0, 0), (1.25, 1.25, 1)),
                      pt.make_light((3, 3, 3), (0, 0.5, 2.5), (3.14159, 0, 0), (3, 3, 1))])
        cam = pt.Camera(resolution=(256, 256))
        cam.update()
        t0 = time.time()
        img, secs = rr.run(s.arrays(), (cam.eye, cam.forward, cam.up, cam.right), 256, 256, "float0", 16, 5)
        rgb = img[..., :3].astype(np.float32)
        half = rgb.astype(np.float16)
        odd = np.flatnonzero(~(np.abs(half.astype(np.float32) - rgb) <= 1e-3 * np.abs(rgb)).ravel())
        path = os.path.join(OUT, "image_Veggie_lattice2_256x256_spp16_depth5.npz")
        np.savez_compressed(path, mean_rgb_f16=half, exact_index=odd.astype(np.int64), exact_value=rgb.ravel()[odd],
                            spp=np.int32(16), depth=np.int32(5), llvmpipe_seconds=np.float64(secs))
        print("lattice 2", rgb.mean(axis=(0, 1), dtype=np.float64), "%.1fs" % (time.time() - t0), os.path.getsize(path), flush=True)
    elif what == "image_512":
        # the reference's own CPU-runnable configuration: Box, 512 x 512, depth 5, iterations 0..63 (~25 s)
        image_fixture("Box", 512, 512, 64, 5)
    elif what == "image_filtered_4k":
        filtered_image_fixture("Veggie", 3840, 2160, 16, 8)
    elif what == "image_no_usemtl":
        # cube.obj has no usemtl: its faces carry material id 0xFFFFFFFF (scene.h:133), and what such a surface
        # scatters is whatever the driver returns for the out-of-range buffer-texture fetches of frag:105-126
        image_fixture("cube", 128, 128, 64, 8)
    elif what == "timing":
        rows = []
        for n, w, h, spp, depth in [("Box", 512, 512, 64, 5), ("Box", 512, 512, 16, 10), ("Box", 256, 256, 8, 8),
                                    ("Veggie", 256, 256, 8, 8), ("Box_Transmission", 256, 256, 8, 8),
                                    ("Box", 1920, 1080, 2, 8), ("Veggie", 1920, 1080, 1, 8)]:
            a, c = scene_and_cam(n, w, h)
            _, secs = rr.run(a, c, w, h, "rgba8", spp, depth)
            rows.append({"scene": n, "width": w, "height": h, "spp": spp, "depth": depth, "seconds": secs,
                         "mpaths_per_s": w * h * spp / secs / 1e6, "host_cores": os.cpu_count()})
            print(rows[-1], flush=True)
        with open(os.path.join(OUT, "llvmpipe_timing.json"), "w") as f:
            json.dump({"renderer": "Mesa llvmpipe (LLVM 15), reference GLSL as shipped (RGBA8 accumulation)",
                       "measured_in": "build container", "rows": rows}, f, indent=1)


main()
