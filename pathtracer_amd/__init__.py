"""MI355X-native path-tracing integrator: Python view of the C ABI.

The product is ``libpathtracer_amd.so`` (C++ host + hand-written gfx950
kernels, ``include/pathtracer_amd.h``).  This module only binds it with ctypes
and mirrors the reference's host classes -- ``Scene`` (src/scene.h), ``Camera``
(src/camera.h) and ``Renderer`` (src/renderer.h) -- so tests and the benchmark
read like a caller of the reference.  There is no Python or CPU fallback: if
the shared library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_AMD_LIB selects another build of the same library (kernel tuning experiments).
LIB_PATH = os.environ.get("PT_AMD_LIB") or os.path.join(_HERE, "libpathtracer_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
        "There is no fallback implementation."
    )

_lib = C.CDLL(LIB_PATH)

PT_OK = 0
HIT_MISS = -1


def hit_light(k: int) -> int:
    """Hit code of light k (PT_HIT_LIGHT in the header)."""
    return -(2 + k)


class PtError(RuntimeError):
    pass


class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", C.POINTER(C.c_float)),
        ("n_vertices", C.c_uint32),
        ("indices", C.POINTER(C.c_float)),
        ("n_triangles", C.c_uint32),
        ("vertex_data", C.POINTER(C.c_float)),
        ("lights", C.POINTER(C.c_float)),
        ("n_lights", C.c_uint32),
        ("materials", C.POINTER(C.c_float)),
        ("n_materials", C.c_uint32),
        ("material_map", C.POINTER(C.c_uint32)),
    ]


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("fov_y_degrees", C.c_float),
        ("device", C.c_int32),
        ("use_bvh", C.c_uint32),
        ("iterations_per_batch", C.c_uint32),
        ("profile", C.c_uint32),
        ("emulate_rgba8", C.c_uint32),
        ("bvh_builder", C.c_uint32),
        ("batch_streams", C.c_uint32),
    ]


class BvhInfo(C.Structure):
    _fields_ = [
        ("n_triangles", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("n_leaves", C.c_uint32),
        ("depth", C.c_uint32),
        ("max_leaf_size", C.c_uint32),
        ("stack_entries", C.c_uint32),
        ("sah_cost", C.c_double),
        ("build_ms", C.c_double),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("iterations", C.c_uint64),
        ("paths", C.c_uint64),
        ("segments", C.c_uint64),
        ("nan_samples", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("n_triangles", C.c_uint32),
        ("n_nodes", C.c_uint32),
        ("bvh_depth", C.c_uint32),
        ("stack_entries", C.c_uint32),
        ("bvh_build_ms", C.c_double),
        ("upload_ms", C.c_double),
        ("ms_raygen", C.c_double),
        ("ms_extend", C.c_double),
        ("ms_shade", C.c_double),
        ("ms_accumulate", C.c_double),
        ("launches_raygen", C.c_uint64),
        ("launches_extend", C.c_uint64),
        ("launches_shade", C.c_uint64),
        ("launches_accumulate", C.c_uint64),
        ("sah_cost", C.c_double),
        ("extend_diag", C.c_uint64 * 8),
        ("hits_shaded", C.c_uint64),
        ("shade_batches", C.c_uint64),
        ("bvh_builder", C.c_uint32),
        ("iterations_per_batch", C.c_uint32),
        ("ms_bounces", C.c_double),
        ("launches_bounces", C.c_uint64),
        ("primary_rays_culled", C.c_uint64),
        ("unlit_last_rays_skipped", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["extend_diag"] = list(self.extend_diag)
        return d


_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/pathtracer_amd.h declares.
_SIGNATURES = {
    "pt_last_error": (C.c_char_p, []),
    "pt_version": (C.c_char_p, []),
    "pt_device_count": (C.c_int, []),
    "pt_scene_load_obj": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    "pt_scene_create_default": (C.c_int, [C.POINTER(_vp)]),
    "pt_scene_from_desc": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(_vp)]),
    "pt_scene_destroy": (None, [_vp]),
    "pt_scene_get_desc": (C.c_int, [_vp, C.POINTER(SceneDesc)]),
    "pt_light_make": (C.c_int, [_f32p, _f32p, _f32p, _f32p, _f32p]),
    "pt_scene_set_lights": (C.c_int, [_vp, _f32p, C.c_uint32]),
    "pt_scene_set_default_light": (C.c_int, [_vp]),
    "pt_scene_instance_lattice": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "pt_camera_create": (C.c_int, [_f32p, _f32p, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    "pt_camera_destroy": (None, [_vp]),
    "pt_camera_look_at": (C.c_int, [_vp, _f32p, _f32p]),
    "pt_camera_move": (C.c_int, [_vp, _f32p]),
    "pt_camera_pan": (C.c_int, [_vp, C.c_float, C.c_float]),
    "pt_camera_orbit": (C.c_int, [_vp, C.c_float, C.c_float]),
    "pt_camera_zoom": (C.c_int, [_vp, C.c_float]),
    "pt_camera_update": (C.c_int, [_vp]),
    "pt_camera_get": (C.c_int, [_vp, _f32p, _u32p]),
    "pt_config_default": (None, [C.POINTER(Config)]),
    "pt_create": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(Config), C.POINTER(_vp)]),
    "pt_destroy": (None, [_vp]),
    "pt_set_camera": (C.c_int, [_vp, _f32p, _f32p, _f32p, _f32p]),
    "pt_resize": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pt_reset": (C.c_int, [_vp]),
    "pt_render": (C.c_int, [_vp, C.c_uint32]),
    "pt_set_partition": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "pt_synchronize": (C.c_int, [_vp]),
    "pt_read_rgba8": (C.c_int, [_vp, C.POINTER(C.c_uint8)]),
    "pt_read_bvh": (C.c_int, [_vp, _f32p, _f32p]),
    "pt_save_state": (C.c_int, [_vp, C.c_char_p]),
    "pt_load_state": (C.c_int, [_vp, C.c_char_p]),
    "pt_read_sum": (C.c_int, [_vp, _f32p]),
    "pt_read_radiance": (C.c_int, [_vp, _f32p]),
    "pt_read_tonemapped": (C.c_int, [_vp, _u8p]),
    "pt_read_tonemapped_noalloc": (C.c_int, [_vp, _u8p]),
    "pt_read_hit_ids": (C.c_int, [_vp, C.c_uint32, _i32p, _f32p, _f32p]),
    "pt_read_hit_normals": (C.c_int, [_vp, C.c_uint32, _f32p]),
    "pt_trace_rays": (C.c_int, [_vp, C.c_uint64, _f32p, _f32p, _f32p, _i32p]),
    "pt_sample_surface": (
        C.c_int,
        [_vp, C.c_uint64, _f32p, _f32p, _f32p, _f32p, _u32p, _f32p, _f32p, _f32p, _u32p, _u32p],
    ),
    "pt_set_exact_readbacks": (C.c_int, [_vp, C.c_uint32]),
    "pt_comm_unique_id": (C.c_int, [_u8p]),
    "pt_comm_create": (C.c_int, [_u8p, C.c_uint32, C.c_uint32, C.c_int32, C.POINTER(_vp)]),
    "pt_comm_create_all": (C.c_int, [_i32p, C.c_uint32, C.POINTER(_vp)]),
    "pt_comm_destroy": (None, [_vp]),
    "pt_comm_runtime_paths": (C.c_int, [C.c_char_p, C.c_size_t]),
    "pt_reduce_sum": (C.c_int, [C.POINTER(_vp), C.POINTER(_vp), C.c_uint32, C.c_uint32]),
    "pt_sum_device_ptr": (C.c_int, [_vp, C.POINTER(_vp)]),
    "pt_get_stats": (C.c_int, [_vp, C.POINTER(Stats)]),
    "pt_bvh_inspect": (C.c_int, [C.POINTER(SceneDesc), C.POINTER(BvhInfo)]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(_lib, _name)  # AttributeError here = the library lacks a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _check(rc: int) -> None:
    if rc != PT_OK:
        raise PtError(f"pathtracer_amd error {rc}: {_lib.pt_last_error().decode(errors='replace')}")


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: Optional[np.ndarray], ty):
    return None if a is None else a.ctypes.data_as(ty)


def device_count() -> int:
    n = _lib.pt_device_count()
    return max(n, 0)


def version() -> str:
    return _lib.pt_version().decode()


def make_light(radiance, position, rotation, scale) -> np.ndarray:
    """Scene::Light(radiance, position, rotation, scale) (scene.h:32-34) -> 20 floats."""
    out = np.zeros(20, np.float32)
    _check(
        _lib.pt_light_make(
            _ptr(_f32(radiance), _f32p), _ptr(_f32(position), _f32p), _ptr(_f32(rotation), _f32p),
            _ptr(_f32(scale), _f32p), _ptr(out, _f32p),
        )
    )
    return out


class Scene:
    """Host scene with the reference's public arrays (scene.h:53-62).

    ``Scene(obj, mtl_root)`` parses OBJ/MTL like ``Scene::Scene(objFilename,
    mtlRoot)``; ``Scene()`` is the reference's default one-triangle scene;
    ``Scene.from_arrays`` wraps arrays that are already in the reference layout
    (used for committed scene dumps).
    """

    def __init__(self, obj: Optional[str] = None, mtl_root: Optional[str] = None, _handle=None):
        self._h = _vp()
        if _handle is not None:
            self._h = _handle
        elif obj is None:
            _check(_lib.pt_scene_create_default(C.byref(self._h)))
        else:
            root = None if mtl_root is None else os.fsencode(mtl_root)
            _check(_lib.pt_scene_load_obj(os.fsencode(obj), root, C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.pt_scene_destroy(h)

    @classmethod
    def from_arrays(cls, vertices, indices, vertex_data, lights, materials, material_map) -> "Scene":
        v, i, vd = _f32(vertices).reshape(-1, 3), _f32(indices).reshape(-1, 4), _f32(vertex_data).reshape(-1, 6)
        l, m = _f32(lights).reshape(-1, 20), _f32(materials).reshape(-1, 8)
        mm = np.ascontiguousarray(material_map, dtype=np.uint32).reshape(-1)
        if vd.shape[0] != 3 * i.shape[0] or mm.shape[0] != i.shape[0]:
            raise ValueError("vertex_data needs 3 records and material_map 1 entry per triangle")
        d = SceneDesc(
            _ptr(v, _f32p), v.shape[0], _ptr(i, _f32p), i.shape[0], _ptr(vd, _f32p), _ptr(l, _f32p), l.shape[0],
            _ptr(m, _f32p), m.shape[0], _ptr(mm, _u32p),
        )
        h = _vp()
        _check(_lib.pt_scene_from_desc(C.byref(d), C.byref(h)))
        return cls(_handle=h)

    @classmethod
    def load_npz(cls, path: str) -> "Scene":
        z = np.load(path)
        return cls.from_arrays(z["vertices"], z["indices"], z["vertex_data"], z["lights"], z["materials"],
                               z["material_map"])

    def save_npz(self, path: str) -> None:
        np.savez_compressed(path, **self.arrays())

    def desc(self) -> SceneDesc:
        d = SceneDesc()
        _check(_lib.pt_scene_get_desc(self._h, C.byref(d)))
        return d

    def arrays(self) -> dict:
        """Copies of m_vertices, m_indices, m_vertexData, m_lights, m_materials, m_materialMap."""
        d = self.desc()

        def grab(p, n, ty=np.float32):
            if not p or n == 0:
                return np.zeros(0, ty)
            return np.ctypeslib.as_array(p, shape=(n,)).astype(ty, copy=True)

        t = d.n_triangles
        return {
            "vertices": grab(d.vertices, 3 * d.n_vertices).reshape(-1, 3),
            "indices": grab(d.indices, 4 * t).reshape(-1, 4),
            "vertex_data": grab(d.vertex_data, 18 * t).reshape(-1, 6),
            "lights": grab(d.lights, 20 * d.n_lights).reshape(-1, 20),
            "materials": grab(d.materials, 8 * d.n_materials).reshape(-1, 8),
            "material_map": grab(d.material_map, t, np.uint32),
        }

    @property
    def n_triangles(self) -> int:
        return self.desc().n_triangles

    def set_lights(self, lights) -> None:
        l = _f32(lights).reshape(-1, 20)
        _check(_lib.pt_scene_set_lights(self._h, _ptr(l, _f32p), l.shape[0]))

    def set_default_light(self) -> None:
        """The light main.cpp:122 installs."""
        _check(_lib.pt_scene_set_default_light(self._h))

    def bvh_info(self) -> dict:
        """Builds and validates the BVH on the host (no GPU needed)."""
        d = self.desc()
        info = BvhInfo()
        _check(_lib.pt_bvh_inspect(C.byref(d), C.byref(info)))
        return {name: getattr(info, name) for name, _ in info._fields_}

    def instance_lattice(self, room_triangles: int, n: int) -> "Scene":
        h = _vp()
        _check(_lib.pt_scene_instance_lattice(self._h, room_triangles, n, C.byref(h)))
        return Scene(_handle=h)


class Camera:
    """Orbit camera (camera.h:9-38).  Axes are valid after update()."""

    def __init__(self, eye=(0.0, 1.5, 15.0), focus=(0.0, -0.25, 0.0), resolution=(1280, 720)):
        self._h = _vp()
        _check(
            _lib.pt_camera_create(_ptr(_f32(eye), _f32p), _ptr(_f32(focus), _f32p), int(resolution[0]),
                                  int(resolution[1]), C.byref(self._h))
        )

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.pt_camera_destroy(h)

    def look_at(self, focus, eye=None) -> None:
        e = None if eye is None else _f32(eye)
        _check(_lib.pt_camera_look_at(self._h, _ptr(_f32(focus), _f32p), _ptr(e, _f32p)))

    def move(self, offset) -> None:
        _check(_lib.pt_camera_move(self._h, _ptr(_f32(offset), _f32p)))

    def pan(self, dx: float, dy: float) -> None:
        _check(_lib.pt_camera_pan(self._h, dx, dy))

    def orbit(self, dx: float, dy: float) -> None:
        _check(_lib.pt_camera_orbit(self._h, dx, dy))

    def zoom(self, scale: float) -> None:
        _check(_lib.pt_camera_zoom(self._h, scale))

    def update(self) -> None:
        _check(_lib.pt_camera_update(self._h))

    def _get(self):
        out = np.zeros(15, np.float32)
        res = np.zeros(2, np.uint32)
        _check(_lib.pt_camera_get(self._h, _ptr(out, _f32p), _ptr(res, _u32p)))
        return out.reshape(5, 3), res

    eye = property(lambda self: self._get()[0][0])
    focus = property(lambda self: self._get()[0][1])
    forward = property(lambda self: self._get()[0][2])
    up = property(lambda self: self._get()[0][3])
    right = property(lambda self: self._get()[0][4])
    resolution = property(lambda self: tuple(int(x) for x in self._get()[1]))


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """The 128 bytes rank 0 hands to the other ranks before everybody calls Comm(...)."""
    buf = np.zeros(COMM_ID_BYTES, np.uint8)
    _check(_lib.pt_comm_unique_id(_ptr(buf, _u8p)))
    return buf.tobytes()


def comm_runtime_paths() -> dict:
    """{"hip": ..., "rccl": ..., "rccl_hip": ...}: the HIP runtime the library is bound to, the RCCL it loaded
    and the HIP runtime that RCCL calls into (pt_comm_runtime_paths)."""
    buf = C.create_string_buffer(4096)
    _check(_lib.pt_comm_runtime_paths(buf, len(buf)))
    return dict(line.split("=", 1) for line in buf.value.decode().splitlines())


class Comm:
    """An RCCL communicator owned by the library (one rank of `world`, on HIP device `device`)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = -1, _handle=None):
        self._h = _vp()
        self.rank, self.world = rank, world
        if _handle is not None:
            self._h = _handle
            return
        if len(unique_id) != COMM_ID_BYTES:
            raise ValueError("the id is the 128 bytes comm_unique_id() returned on rank 0")
        buf = np.frombuffer(unique_id, np.uint8).copy()
        _check(_lib.pt_comm_create(_ptr(buf, _u8p), rank, world, device, C.byref(self._h)))

    @classmethod
    def for_devices(cls, devices) -> list:
        """One communicator per device of this process (ncclCommInitAll)."""
        d = np.ascontiguousarray(devices, np.int32)
        hs = (_vp * len(d))()
        _check(_lib.pt_comm_create_all(_ptr(d, _i32p), len(d), hs))
        return [cls(b"", k, len(d), _handle=_vp(hs[k])) for k in range(len(d))]

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.pt_comm_destroy(h)

    __del__ = close


def reduce_sum(renderers, comms, root: int = 0) -> None:
    """One RCCL reduce(sum) of the renderers' sum buffers into rank `root`'s (in place, asynchronous)."""
    n = len(renderers)
    if n != len(comms) or n == 0:
        raise ValueError("one communicator per renderer")
    rs = (_vp * n)(*[r._h for r in renderers])
    cs = (_vp * n)(*[c._h for c in comms])
    _check(_lib.pt_reduce_sum(rs, cs, n, root))


class Renderer:
    """Progressive renderer with the reference's surface (renderer.h:32-38):
    draw(), reset(), resize(), update_camera(), plus readbacks."""

    def __init__(self, scene: Scene, camera: Optional[Camera] = None, *, max_depth: int = 10,
                 fov_y_degrees: float = 19.5, device: int = -1, use_bvh: bool = True,
                 iterations_per_batch: int = 0, profile: int = 0, emulate_rgba8: bool = False,
                 bvh_builder: str = "auto", batch_streams: int = 2):
        self.scene = scene
        self.camera = camera if camera is not None else Camera()
        w, h = self.camera.resolution
        cfg = Config()
        _lib.pt_config_default(C.byref(cfg))
        cfg.width, cfg.height = w, h
        cfg.max_depth = max_depth
        cfg.fov_y_degrees = fov_y_degrees
        cfg.device = device
        cfg.use_bvh = 1 if use_bvh else 0
        cfg.iterations_per_batch = iterations_per_batch
        cfg.profile = int(profile)  # bit 0: per-launch HIP-event timing; bit 1: traversal counters
        cfg.emulate_rgba8 = 1 if emulate_rgba8 else 0
        if bvh_builder not in ("host", "device", "auto"):
            raise ValueError("bvh_builder is 'host', 'device' or 'auto'")
        cfg.bvh_builder = {"host": 0, "device": 1, "auto": 2}[bvh_builder]
        cfg.batch_streams = int(batch_streams)  # 2: batches alternate between two streams; 1: strictly in sequence
        self.config = cfg
        self._h = _vp()
        d = scene.desc()
        _check(_lib.pt_create(C.byref(d), C.byref(cfg), C.byref(self._h)))
        self.width, self.height = w, h
        self.update_camera()

    def close(self) -> None:
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.pt_destroy(h)

    __del__ = close

    # ---- renderer.h surface
    def draw(self, n: int = 1) -> None:
        """n x Renderer::draw: n more samples per pixel (asynchronous)."""
        _check(_lib.pt_render(self._h, n))

    def reset(self) -> None:
        _check(_lib.pt_reset(self._h))

    def resize(self, width: int, height: int) -> None:
        _check(_lib.pt_resize(self._h, width, height))
        self.width, self.height = width, height

    def update_camera(self) -> None:
        self.camera.update()
        c = self.camera
        _check(
            _lib.pt_set_camera(self._h, _ptr(_f32(c.eye), _f32p), _ptr(_f32(c.forward), _f32p),
                               _ptr(_f32(c.up), _f32p), _ptr(_f32(c.right), _f32p))
        )

    # ---- multi-GPU partition, sync, readbacks
    def set_partition(self, rank: int, world: int) -> None:
        _check(_lib.pt_set_partition(self._h, rank, world))

    def save_state(self, path: str) -> None:
        """Checkpoint: sum buffer + iteration counter (the reference's render state, renderer.h:21-29)."""
        _check(_lib.pt_save_state(self._h, os.fsencode(path)))

    def load_state(self, path: str) -> None:
        """Resume from a checkpoint written for the same scene, camera, depth, size and partition."""
        _check(_lib.pt_load_state(self._h, os.fsencode(path)))

    def synchronize(self) -> None:
        _check(_lib.pt_synchronize(self._h))

    def read_sum(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        _check(_lib.pt_read_sum(self._h, _ptr(out, _f32p)))
        return out

    def read_radiance(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), np.float32)
        _check(_lib.pt_read_radiance(self._h, _ptr(out, _f32p)))
        return out

    def read_bvh(self):
        """(nodes, triangles) as they lie in device memory: nodes n x 16 float32 words (two child
        boxes, then the child codes as int32 bits), triangles T x 16 words in leaf order."""
        st = self.stats()
        nodes = np.zeros((st["n_nodes"], 16), np.float32)
        tris = np.zeros((st["n_triangles"], 16), np.float32)
        _check(_lib.pt_read_bvh(self._h, _ptr(nodes, _f32p), _ptr(tris, _f32p)))
        return nodes, tris

    def read_rgba8(self) -> np.ndarray:
        """The emulated 8-bit accumulation texture (emulate_rgba8=True), H x W x 4, row 0 = bottom row."""
        out = np.zeros((self.height, self.width, 4), np.uint8)
        _check(_lib.pt_read_rgba8(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def read_tonemapped(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 3), np.uint8)
        _check(_lib.pt_read_tonemapped(self._h, _ptr(out, _u8p)))
        return out

    def read_tonemapped_noalloc(self) -> np.ndarray:
        """read_tonemapped through the library's own staging array: no device allocation in the call."""
        out = np.zeros((self.height, self.width, 3), np.uint8)
        _check(_lib.pt_read_tonemapped_noalloc(self._h, _ptr(out, _u8p)))
        return out

    def read_hit_ids(self, iteration: int = 0, want_t: bool = False, want_dirs: bool = False):
        ids = np.zeros((self.height, self.width), np.int32)
        t = np.zeros((self.height, self.width), np.float32) if want_t else None
        dirs = np.zeros((self.height, self.width, 3), np.float32) if want_dirs else None
        _check(_lib.pt_read_hit_ids(self._h, iteration, _ptr(ids, _i32p), _ptr(t, _f32p), _ptr(dirs, _f32p)))
        return ids, t, dirs

    def read_hit_normals(self, iteration: int = 0) -> np.ndarray:
        """Intersection::normal of the primary hits (frag:434-460), H x W x 3."""
        out = np.zeros((self.height, self.width, 3), np.float32)
        _check(_lib.pt_read_hit_normals(self._h, iteration, _ptr(out, _f32p)))
        return out

    def set_exact_readbacks(self, enabled: bool) -> None:
        """False: read_hit_ids / trace_rays use the render loop's extend kernel verbatim."""
        _check(_lib.pt_set_exact_readbacks(self._h, 1 if enabled else 0))

    def trace_rays(self, origins, dirs):
        o, d = _f32(origins).reshape(-1, 3), _f32(dirs).reshape(-1, 3)
        n = o.shape[0]
        t = np.zeros(n, np.float32)
        ids = np.zeros(n, np.int32)
        _check(_lib.pt_trace_rays(self._h, n, _ptr(o, _f32p), _ptr(d, _f32p), _ptr(t, _f32p), _ptr(ids, _i32p)))
        return t, ids

    def sample_surface(self, materials, normals, wo, xi, seeds):
        m, nr, w = _f32(materials).reshape(-1, 8), _f32(normals).reshape(-1, 3), _f32(wo).reshape(-1, 3)
        x = _f32(xi).reshape(-1, 2)
        s = np.ascontiguousarray(seeds, dtype=np.uint32).reshape(-1, 2)
        n = m.shape[0]
        wi = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        f = np.zeros((n, 3), np.float32)
        flags = np.zeros(n, np.uint32)
        so = np.zeros((n, 2), np.uint32)
        _check(
            _lib.pt_sample_surface(self._h, n, _ptr(m, _f32p), _ptr(nr, _f32p), _ptr(w, _f32p), _ptr(x, _f32p),
                                   _ptr(s, _u32p), _ptr(wi, _f32p), _ptr(pdf, _f32p), _ptr(f, _f32p),
                                   _ptr(flags, _u32p), _ptr(so, _u32p))
        )
        return wi, pdf, f, flags, so

    def sum_device_ptr(self) -> int:
        p = _vp()
        _check(_lib.pt_sum_device_ptr(self._h, C.byref(p)))
        return int(p.value)

    def stats(self) -> dict:
        s = Stats()
        _check(_lib.pt_get_stats(self._h, C.byref(s)))
        return s.as_dict()
