"""The headless command-line renderer (stand-in for main.cpp): scene file, lights and camera from the
command line, image files out, checkpoint and resume."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "pathtracer_amd", "pt_render")

ROOM = """mtllib room.mtl
v -2 -2 -2
v 2 -2 -2
v 2 2 -2
v -2 2 -2
v -2 -2 2
v 2 -2 2
v 2 2 2
v -2 2 2
usemtl white
f 1 2 3 4
f 1 5 6 2
f 4 3 7 8
usemtl red
f 1 4 8 5
usemtl mirror
f 2 6 7 3
"""
MTL = """newmtl white
Kd 0.8 0.8 0.8
Ns 0
newmtl red
Kd 0.7 0.1 0.1
Ns 0
newmtl mirror
Kd 0.9 0.9 0.9
Pr 0.1
Pm 1.0
"""


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)


def _run(args, cwd):
    out = subprocess.run([EXE, *args], cwd=cwd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    return out.stdout


def test_cli_lights_camera_checkpoint_resume(tmp_path):
    if not os.path.exists(EXE):
        pytest.fail("pathtracer_amd/pt_render is not built")
    (tmp_path / "room.obj").write_text(ROOM)
    (tmp_path / "room.mtl").write_text(MTL)
    common = ["--scene", "room.obj", "--mtl-root", "./", "--width", "96", "--height", "64", "--depth", "5",
              "--orbit", "0.1", "0.05", "--zoom", "1.5",
              "--light", "6", "5", "4", "0", "1.9", "0", "1.57", "0", "0", "1.5", "1.5", "1",
              "--light", "1", "2", "3", "-1.9", "0", "0", "0", "1.57", "0", "1", "1", "1"]
    log = _run(common + ["--spp", "12", "--out", "a.ppm", "--pfm", "a.pfm"], tmp_path)
    assert "10 triangles" in log and "12 spp" in log
    a = _read_pfm(tmp_path / "a.pfm")
    assert a.shape == (64, 96, 3) and np.isfinite(a).all() and a.max() > 0.5  # the lights are seen
    with open(tmp_path / "a.ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"96 64\n"

    _run(common + ["--spp", "5", "--save-state", "s.bin", "--out", "b0.ppm"], tmp_path)
    log = _run(common + ["--spp", "12", "--resume", "s.bin", "--checkpoint-every", "4", "--save-state", "s.bin",
                         "--out", "b.ppm", "--pfm", "b.pfm"], tmp_path)
    assert "resumed s.bin at 5 spp" in log
    assert np.array_equal(_read_pfm(tmp_path / "b.pfm"), a)
    assert (tmp_path / "a.ppm").read_bytes() == (tmp_path / "b.ppm").read_bytes()

    log = _run(common + ["--spp", "12", "--bvh", "device", "--out", "c.ppm", "--pfm", "c.pfm"], tmp_path)
    assert "built on the device" in log
    assert np.array_equal(_read_pfm(tmp_path / "c.pfm"), a)  # same tree, same image

    log = _run(common + ["--spp", "12", "--preview-every", "5", "--out", "p.ppm", "--pfm", "p.pfm"], tmp_path)
    assert "preview at 5 spp" in log and "preview at 10 spp" in log and "preview at 12" not in log
    assert np.array_equal(_read_pfm(tmp_path / "p.pfm"), a)

    # --gpus takes the multi-GPU code path (MultiRenderer, a communicator for all GPUs of the process, one
    # RCCL reduce before every image written); with the one GPU of a test box it must give the same image
    log = _run(common + ["--spp", "12", "--gpus", "1", "--preview-every", "5", "--out", "g.ppm", "--pfm", "g.pfm"], tmp_path)
    assert "on 1 GPUs" in log and "preview at 10 spp" in log
    assert np.array_equal(_read_pfm(tmp_path / "g.pfm"), a)
    assert (tmp_path / "a.ppm").read_bytes() == (tmp_path / "g.ppm").read_bytes()
    many = subprocess.run([EXE, *common, "--spp", "2", "--gpus", "64"], cwd=tmp_path, capture_output=True, text=True, timeout=240)
    assert many.returncode != 0 and "GPU(s) visible" in many.stderr

    dark = _run(["--scene", "room.obj", "--mtl-root", "./", "--width", "32", "--height", "32", "--spp", "2", "--no-light",
                 "--pfm", "d.pfm", "--out", "d.ppm"], tmp_path)
    assert "2 spp" in dark and not _read_pfm(tmp_path / "d.pfm").any()
    bad = subprocess.run([EXE, "--scene", "room.obj", "--mtl-root", "./", "--width", "32", "--height", "32", "--spp", "2",
                          "--resume", "s.bin"], cwd=tmp_path, capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0 and "pt_load_state" in bad.stderr


def _read_png_rgb8(path):
    """A minimal PNG reader of our own (8-bit RGB, no interlace; filter type per scanline honoured for type 0
    only), with every chunk's CRC and the zlib stream's Adler-32 checked by zlib."""
    import struct
    import zlib

    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    at, chunks = 8, []
    while at < len(data):
        n, kind = struct.unpack(">I4s", data[at:at + 8])
        body = data[at + 8:at + 8 + n]
        (crc,) = struct.unpack(">I", data[at + 8 + n:at + 12 + n])
        assert zlib.crc32(kind + body) & 0xFFFFFFFF == crc, f"bad CRC in {kind}"
        chunks.append((kind, body))
        at += 12 + n
    assert [k for k, _ in chunks][0] == b"IHDR" and chunks[-1] == (b"IEND", b"")
    w, h, bits, colour, comp, flt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (bits, colour, comp, flt, lace) == (8, 2, 0, 0, 0)
    raw = zlib.decompress(b"".join(b for k, b in chunks if k == b"IDAT"))  # raises on a wrong Adler-32
    rows = np.frombuffer(raw, np.uint8).reshape(h, 1 + 3 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 3)


def test_cli_writes_png(tmp_path):
    """--out x.png: the tone-mapped image as PNG (the format of the reference's published output/*.png), the same
    bytes as the PPM of the same render; read back with our own checker and, where installed, with PIL."""
    if not os.path.exists(EXE):
        pytest.fail("pathtracer_amd/pt_render is not built")
    (tmp_path / "room.obj").write_text(ROOM)
    (tmp_path / "room.mtl").write_text(MTL)
    # 301 x 223: scanlines of odd length, 201 KB of pixel data = four stored deflate blocks
    common = ["--scene", "room.obj", "--mtl-root", "./", "--width", "301", "--height", "223", "--depth", "5", "--spp", "6"]
    _run(common + ["--out", "a.png"], tmp_path)
    _run(common + ["--out", "a.ppm"], tmp_path)
    with open(tmp_path / "a.ppm", "rb") as f:
        assert f.readline() == b"P6\n" and f.readline() == b"301 223\n" and f.readline() == b"255\n"
        ppm = np.frombuffer(f.read(), np.uint8).reshape(223, 301, 3)
    png = _read_png_rgb8(tmp_path / "a.png")
    assert ppm.any() and np.array_equal(png, ppm)
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png").convert("RGB")), ppm)
